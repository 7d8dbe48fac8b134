"""CPU test of the N > 1 path with torch.distributed (gloo, world_size 2): envs are sharded by contiguous
global id with no data-path collective, and the single gather of episode results reproduces what one process
computes for all envs.  The shard results here come from the CPU oracle (the checker); on GPUs the same
functions run on the HIP path's device tensors over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

TOTAL, SEED, TURNS = 96, 4711, 150


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rollout(n, first):
    import oracle as om
    om.lib().evo_set_num_threads(1)
    o = om.Oracle(n, seed=SEED, env_id_base=first, auto_reset=True)
    o.reset()
    for _ in range(TURNS):
        o.step(o.random_actions())
    return o.episode_stats()


def _worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import everglades_amd as evg
    first, cnt = evg.shard_range(TOTAL, world, rank)
    st = _rollout(cnt, first)
    g = evg.gather_episode_results(torch.from_numpy(st["returns"]), torch.from_numpy(st["length"]), torch.from_numpy(st["winner"]), TOTAL)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), returns=g["returns"].numpy(), winner=g["winner"].numpy(),
             length=g["length"].numpy(), wins=np.array(g["wins"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_rollout_gathers_to_single_process_result(tmp_path, oracle_mod, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ref = _rollout(TOTAL, 0)
    for r in range(world):                       # all-gather: every rank holds the full result, in global env order
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(g["winner"], ref["winner"]) and np.array_equal(g["length"], ref["length"])
        assert np.array_equal(g["returns"], ref["returns"])
        w = ref["winner"]
        assert g["wins"].tolist() == [int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), int((w < 0).sum())]
        assert g["wins"][:3].sum() == TOTAL


def test_gather_without_process_group_is_identity():
    import everglades_amd as evg
    r = torch.arange(10, dtype=torch.float32).reshape(5, 2)
    g = evg.gather_episode_results(r, torch.full((5,), 150, dtype=torch.int32), torch.tensor([0, 1, 2, 0, -1], dtype=torch.int8))
    assert torch.equal(g["returns"], r) and g["wins"] == (2, 1, 1, 1) and g["length"].tolist() == [150] * 5


def _worker_dst(rank, world, port, outdir, total):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import everglades_amd as evg
    first, cnt = evg.shard_range(total, world, rank)
    st = _rollout(cnt, first)
    packed = evg.distributed.pack_episode_results(torch.from_numpy(st["returns"]), torch.from_numpy(st["length"]), torch.from_numpy(st["winner"]))
    g = evg.ResultGather(cnt, total, "cpu", dst=0)
    for _ in range(2):                           # the buffers are reused from call to call
        full = g(packed)
    assert (full is None) == (rank != 0)
    assert g.collective == "gather" and g.backend == "gloo" and g.calls == 2
    assert (g.recv is None) == (rank != 0)       # nothing is allocated inside a call: the receive buffer exists on rank dst only
    g2 = evg.ResultGather(cnt, total, "cpu", dst=0, collective="all_gather")   # what a backend without gather() runs: decided at construction,
    full2 = g2(packed)                           # by every rank alike; same result on rank 0
    assert g2.collective == "all_gather" and g2.recv is not None
    assert (full2 is None) == (rank != 0) and (rank != 0 or torch.equal(full2, full))
    if rank == 0:
        assert sum(g.rows_per_rank(full)) == total and g.rows_per_rank(full) == g.counts
    if rank == 0:
        d = evg.ResultGather.split(full)
        np.savez(os.path.join(outdir, "dst.npz"), returns=d["returns"].numpy(), winner=d["winner"].numpy(), length=d["length"].numpy(),
                 wins=np.array(evg.ResultGather.win_counts(full)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 96), (3, 97), (8, 203)])
def test_result_gather_to_rank_zero(tmp_path, oracle_mod, world, total):
    """The preallocated form bench.py times (ResultGather: one gather of the packed [n, 4] rows to rank 0), with equal shards and
    with shards that differ by one env -- up to the 8 ranks of BASELINE config 4 (203 envs: three shards of 26, five of 25):
    rank 0 holds what one process computes for all envs, the other ranks receive nothing."""
    port = _free_port()
    mp.spawn(_worker_dst, args=(world, port, str(tmp_path), total), nprocs=world, join=True)
    ref = _rollout(total, 0)
    g = np.load(os.path.join(str(tmp_path), "dst.npz"))
    assert np.array_equal(g["winner"], ref["winner"]) and np.array_equal(g["length"], ref["length"]) and np.array_equal(g["returns"], ref["returns"])
    w = ref["winner"]
    assert g["wins"].tolist() == [int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), int((w < 0).sum())]


def test_result_gather_lets_collective_errors_propagate(monkeypatch):
    """The collective is chosen once, at construction, from the backend; a failure inside it is not caught and does not change
    the choice (a rank that swallowed an RCCL error and moved on to another collective would park its peers for ever)."""
    import everglades_amd as evg
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = evg.ResultGather(5, 5, "cpu", force=True)
        assert g.on and g.collective == "gather" and g.backend == "gloo"
        rows = torch.arange(20, dtype=torch.float32).reshape(5, 4)
        assert torch.equal(g(rows), rows)

        def broken(*a, **k):
            raise RuntimeError("communicator aborted")
        monkeypatch.setattr(dist, "gather", broken)
        with pytest.raises(RuntimeError, match="communicator aborted"):
            g(rows)
        assert g.collective == "gather" and g.calls == 1
        monkeypatch.undo()
        with pytest.raises(ValueError):
            g(rows[:4])                              # a wrong shard size never reaches the collective
        with pytest.raises(ValueError):
            evg.ResultGather(4, 5, "cpu", force=True)
        with pytest.raises(ValueError):
            evg.ResultGather(5, 5, "cpu", force=True, collective="ring")
        # an unknown backend name falls to the all-gather, on every rank alike
        monkeypatch.setattr(dist, "get_backend", lambda group=None: "ucc")
        assert evg.ResultGather(5, 5, "cpu", force=True).collective == "all_gather"
    finally:
        monkeypatch.undo()
        dist.destroy_process_group()
