"""Multi-GPU: environments are independent, so the path shards with no data-path collective.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Rank r owns the contiguous env range `shard_range(total, world, r)`; env ids -- and hence
every random stream -- are global, so results do not depend on the number of GPUs.  The only exchange
is ONE gather of per-episode results ({return[2], winner, length} per env, 16 bytes) at episode
boundaries: about 1 MB per GPU at 65 536 envs, far below what one xGMI link moves in a millisecond.
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


def win_counts(gathered):
    """(p0 wins, p1 wins, ties, envs without a finished episode) of a gather_episode_results() result."""
    w = gathered["winner"].to("cpu")
    return (int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), int((w < 0).sum()))
