"""Multi-GPU: environments are independent, so the path shards with no data-path collective.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Rank r owns the contiguous env range `shard_range(total, world, r)`; env ids -- and hence
every random stream -- are global, so results do not depend on the number of GPUs.  The only exchange
is ONE gather of per-episode results ({return[2], winner, length} per env, 16 bytes) at episode
boundaries: about 1 MB per GPU at 65 536 envs, far below what one xGMI link moves in a millisecond.
`ResultGather` is the preallocated form bench.py times (one pack kernel + one `gather` to rank 0);
`gather_episode_results` is the convenience form that leaves the full result on every rank (all-gather).
"""


def shard_range(total_envs, world_size, rank):
    """Contiguous partition of [0, total_envs): returns (first_env_id, count) of `rank`."""
    base, rem = divmod(int(total_envs), int(world_size))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def pack_episode_results(returns, length, winner):
    """[n, 4] float32: return_p0, return_p1, winner, length (small integers are exact in float32)."""
    import torch
    return torch.cat([returns.to(torch.float32).reshape(-1, 2), winner.to(torch.float32).reshape(-1, 1),
                      length.to(torch.float32).reshape(-1, 1)], dim=1).contiguous()


def gather_episode_results(returns, length, winner, total_envs=None, group=None, count_wins=True):
    """The single collective of the path.  Every rank passes the results of its own shard (tensors on its
    device); returns a dict with the results of ALL envs in global env order plus win counts
    (p0, p1, tie, unfinished).  Works without an initialised process group (single GPU).  count_wins=False skips the
    four host-synchronising reductions (`wins` is then None; call win_counts() on the result when needed)."""
    import torch
    import torch.distributed as dist
    local = pack_episode_results(returns, length, winner)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        home = local.device
        if dist.get_backend(group) == "gloo" and local.is_cuda:
            local = local.cpu()                   # gloo rehearsal of the RCCL path: collectives on host copies
        if total_envs is None:
            cnt = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
            dist.all_reduce(cnt, group=group)
            total_envs = int(cnt.item())
        counts = [shard_range(total_envs, world, r)[1] for r in range(world)]
        assert counts[rank] == local.shape[0], "local shard size does not match shard_range()"
        width = max(counts)
        padded = torch.zeros((width, 4), dtype=torch.float32, device=local.device)
        padded[:local.shape[0]] = local
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)           # ONE RCCL all-gather over xGMI
        full = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).to(home)
    else:
        full = local
    out = dict(returns=full[:, :2], winner=full[:, 2].to(torch.int8), length=full[:, 3].to(torch.int32), wins=None)
    if count_wins:
        out["wins"] = win_counts(out)
    return out


class ResultGather:
    """The path's one exchange with everything preallocated: rank `dst` receives the packed per-env results
    (`EvergladesVecEnv.packed_episode_results()`: float32 [n_local, 4] = return p0, return p1, winner, length) of every rank,
    in global env order, with ONE collective -- `torch.distributed.gather` (RCCL: every rank sends its 16 B/env straight to
    `dst` over its own xGMI link; nothing is sent to ranks that do not need it) -- and no other kernel.  Shards that differ
    by one env (total not divisible by the world size) are padded to a common width.  With the gloo backend (CPU tests, or
    a rehearsal on one GPU) the collective runs on host copies.

        g = ResultGather(n_local, total, device)           # once
        full = g(env.packed_episode_results(out=g.buffer)) # [total, 4] on rank dst, None elsewhere (g.buffer: the send buffer
                                                           # itself, so that the pack kernel writes where the collective reads)
        wins = ResultGather.win_counts(full)               # (p0, p1, ties, unfinished)
    """

    # backends whose torch.distributed process group implements gather(); anything else exchanges with an all-gather
    GATHER_BACKENDS = ("nccl", "gloo", "mpi")

    def __init__(self, n_local, total_envs, device, group=None, dst=0, force=False, collective=None):
        """collective: None = decided HERE, once, from the backend's name ("gather" for nccl (= RCCL) / gloo / mpi, "all_gather"
        for any other backend), or "gather" / "all_gather" given by the caller -- the same on every rank, because every rank
        sees the same backend (or passes the same argument).  __call__ never changes it and catches nothing: an error of
        the collective propagates, so a rank with a broken communicator exits non-zero instead of parking its peers in a
        collective that it has left."""
        import torch
        import torch.distributed as dist
        self.group, self.dst, self.total = group, int(dst), int(total_envs)
        # force: run the collective even in a one-rank group (rehearsal of the N > 1 code path on one GPU)
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.counts = [shard_range(self.total, self.world, r)[1] for r in range(self.world)]
        if self.counts[self.rank] != int(n_local):
            raise ValueError("local shard size %d does not match shard_range(%d, %d, %d) = %d" % (int(n_local), self.total, self.world, self.rank,
                                                                                                  self.counts[self.rank]))
        self.width = max(self.counts)
        self.backend = str(dist.get_backend(group)).lower() if self.on else None
        if collective is None:
            collective = "gather" if (self.backend in self.GATHER_BACKENDS or not self.on) else "all_gather"
        if collective not in ("gather", "all_gather"):
            raise ValueError("collective must be 'gather', 'all_gather' or None")
        self.collective = collective
        self.host = self.on and self.backend == "gloo"
        dev = torch.device("cpu") if self.host else torch.device(device)
        self.home = torch.device(device)
        self.send = torch.zeros((self.width, 4), dtype=torch.float32, device=dev) if self.on else None
        # every buffer the collective touches exists before the first call (rank dst for a gather, every rank for an all-gather)
        receives = self.on and (self.rank == self.dst or collective == "all_gather")
        self.recv = torch.empty((self.world, self.width, 4), dtype=torch.float32, device=dev) if receives else None
        self.parts = list(self.recv.unbind(0)) if self.recv is not None else None
        self.calls = 0
        # where the local rows should be written to spare the copy into the send buffer (None: pass any [n_local, 4] tensor)
        self.buffer = self.send[:self.counts[self.rank]] if (self.on and not self.host) else None

    def __call__(self, packed):
        """packed: float32 [n_local, 4] on this rank's device.  Returns [total, 4] on rank dst (a view of the receive buffer when
        all shards have the same size), None on the other ranks; the input itself without a process group."""
        import torch
        import torch.distributed as dist
        if not self.on:
            return packed
        n = self.counts[self.rank]
        if tuple(packed.shape) != (n, 4):
            raise ValueError("packed must be [%d, 4], got %s" % (n, tuple(packed.shape)))
        if self.buffer is None or packed.data_ptr() != self.buffer.data_ptr():
            self.send[:n].copy_(packed, non_blocking=not self.host)
        if self.collective == "gather":
            dist.gather(self.send, self.parts, dst=self.dst, group=self.group)
        else:
            dist.all_gather(self.parts, self.send, group=self.group)
        self.calls += 1
        if self.rank != self.dst:
            return None
        if all(c == self.width for c in self.counts):
            full = self.recv.view(self.world * self.width, 4)
        else:
            full = torch.cat([self.recv[r, :c] for r, c in enumerate(self.counts)], dim=0)
        return full.to(self.home) if self.host and self.home.type != "cpu" else full

    def rows_per_rank(self, full):
        """Rows of a gathered [total, 4] tensor that hold a finished episode (winner >= 0), per source rank: what the collective
        actually carried from each rank (bench.py reports it for N > 1)."""
        out, at = [], 0
        w = full[:, 2].to("cpu")
        _refuse_poisoned(w)
        for c in self.counts:
            out.append(int((w[at:at + c] >= 0).sum()))
            at += c
        return out

    @staticmethod
    def split(full):
        """dict(returns [total, 2] f32, winner int8, length int32) of a gathered [total, 4] tensor"""
        import torch
        return dict(returns=full[:, :2], winner=full[:, 2].to(torch.int8), length=full[:, 3].to(torch.int32), wins=None)

    @staticmethod
    def win_counts(full):
        w = full[:, 2].to("cpu")
        _refuse_poisoned(w)
        return (int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), int((w < 0).sum()))


class NativeGather:
    """The same exchange WITHOUT torch.distributed: RCCL through the C-ABI (evg_comm_unique_id / evg_comm_init / evg_gather_returns, include/evg.h) -- what a
    consumer without a framework uses (examples/c_client.c), and one launch cheaper on the stream than the torch form: the pack kernel and one grouped RCCL
    send / receive run directly on the caller's stream.

        cid = NativeGather.unique_id() on ONE rank; hand the 128 bytes to every rank (file, socket, a torch store ...)
        g = NativeGather(env, total, world, rank, cid)      # collective: blocks until every rank has called (ncclCommInitRank)
        full = g()                                           # collective, enqueued on torch's current stream: [total, 4] on rank `root`, None elsewhere
    """

    def __init__(self, env, total_envs, world, rank, comm_id, root=0):
        import ctypes as C
        import numpy as np
        import torch
        self.env, self.total, self.world, self.rank, self.root = env, int(total_envs), int(world), int(rank), int(root)
        self.counts = [shard_range(self.total, self.world, r)[1] for r in range(self.world)]
        if self.counts[self.rank] != env.num_envs:
            raise ValueError("this handle has %d envs, shard_range(%d, %d, %d) gives %d" % (env.num_envs, self.total, self.world, self.rank,
                                                                                            self.counts[self.rank]))
        if len(comm_id) != 128:
            raise ValueError("comm_id must be the 128 bytes of NativeGather.unique_id()")
        cnt = np.asarray(self.counts, np.int32)
        buf = C.create_string_buffer(bytes(comm_id), 128)
        env._check(env.L.evg_comm_init(env._h, buf, self.world, self.rank, cnt.ctypes.data_as(C.c_void_p)))
        self.recv = torch.empty((self.total, 4), dtype=torch.float32, device=env.device) if self.rank == self.root else None

    @staticmethod
    def unique_id(library=None):
        import ctypes as C
        from . import _lib
        L = _lib.load(library)
        buf = C.create_string_buffer(128)
        _lib.check(L.evg_comm_unique_id(buf), L)
        return buf.raw

    def __call__(self):
        import ctypes as C
        env = self.env
        env._check(env.L.evg_gather_returns(env._h, self.root, None if self.recv is None else C.c_void_p(self.recv.data_ptr()), env._stream()))
        return self.recv

    def close(self):
        self.env._check(self.env.L.evg_comm_destroy(self.env._h))


def _refuse_poisoned(w):
    """Rows with winner -2 come from a handle whose fault word is set (evg_pack_episode_results poisons them): its results are not valid."""
    if bool((w == -2).any()):
        from ._lib import EvgFault
        raise EvgFault("%d gathered rows are poisoned (winner -2): a rank's handle reported a chunk hand-over fault (evg_check_fault); its results are not "
                       "valid" % int((w == -2).sum()))


def win_counts(gathered):
    """(p0 wins, p1 wins, ties, envs without a finished episode) of a gather_episode_results() result."""
    w = gathered["winner"].to("cpu")
    _refuse_poisoned(w)
    return (int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), int((w < 0).sum()))
