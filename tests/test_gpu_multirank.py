"""SURVEY 8(e) on the product path with one process per rank: every rank owns a HIP handle for its contiguous shard of global env
ids, plays a persistent rollout with auto-reset and sends its packed episode results to rank 0 with the path's ONE collective
(everglades_amd.ResultGather: pack kernel + gather).  What rank 0 receives, the win bookkeeping (evaluate.py:155-160) and every
shard's final state must equal ONE oracle run of the total size -- i.e. the oracle, not the ranks themselves, checks what the
transport carried.

Transports of the same child program:
  gloo, 2 ranks on GPU 0          runs on every box (the collective on host copies)
  nccl (= RCCL), 1 rank           runs on every box: process-group, device buffers and the RCCL launch of the gather, nothing on the wire
  evg_gather_returns, 1 rank      runs on every box: RCCL through the C-ABI (no torch.distributed), one rank; one rank per GPU like the nccl form below
  nccl, one rank per visible GPU  needs >= 2 GPUs (xGMI); at most 4 ranks unless EVG_TEST_MAX_RANKS=8 (then BASELINE config 4 at its full size).  SKIPPED on the
                                  one-GPU boxes this repository has been developed on: RCCL has not carried a row between two GPUs in any
                                  round, and this is the test that checks it the day a multi-GPU box runs the suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_RANK_CHILD = r"""
import os, sys
rank, world, total, seed, steps = (int(x) for x in sys.argv[1:6])
out_dir, root, backend, ndev = sys.argv[6], sys.argv[7], sys.argv[8], int(sys.argv[9])
sys.path.insert(0, root)
import numpy as np
import torch
import torch.distributed as dist
import everglades_amd as evg
dev = rank % ndev                           # nccl: one GPU per rank; the gloo rehearsal puts every rank on GPU 0 (ndev = 1)
torch.cuda.set_device(dev)
device = torch.device("cuda", dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
elif backend == "gloo":
    dist.init_process_group("gloo", rank=rank, world_size=world)
first, cnt = evg.shard_range(total, world, rank)
env = evg.EvergladesVecEnv(cnt, device=device, seed=seed, env_id_base=first, auto_reset=True)
env.reset()
env.rollout_random(steps, turns_per_launch=150)                        # persistent form: launches of 150 and steps - 150 turns
counts = torch.zeros(4, dtype=torch.int64, device=device)
if backend == "evg":
    # no torch.distributed at all: RCCL through the C-ABI (evg_comm_unique_id / evg_comm_init / evg_gather_returns).  The communicator's id travels in a file.
    import time
    idf = os.path.join(out_dir, "comm_id.bin")
    if rank == 0:
        open(idf + ".tmp", "wb").write(evg.NativeGather.unique_id())
        os.rename(idf + ".tmp", idf)
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 120, "no communicator id from rank 0"
        time.sleep(0.05)
    g = evg.NativeGather(env, total, world, rank, open(idf, "rb").read())
    full = g()                                                         # THE collective of the path: pack kernel + grouped RCCL send / receive to rank 0
    env.packed_episode_results(counts=counts)
    torch.cuda.synchronize(device)
    summed, collective, backend_name = None, "gather", "evg_gather_returns"
    class g_:                                                          # (rows_per_rank of the torch form, for the parent's checks)
        @staticmethod
        def rows_per_rank(full):
            w, out, at = full[:, 2].cpu(), [], 0
            for c in g.counts:
                out.append(int((w[at:at + c] >= 0).sum())); at += c
            return out
    rows_per_rank = g_.rows_per_rank
else:
    g = evg.ResultGather(cnt, total, device, force=True)              # (force: the one-rank group runs the collective too)
    full = g(env.packed_episode_results(out=g.buffer, counts=counts)) # THE collective of the path: pack kernel + gather to rank 0
    summed = counts.clone() if backend == "nccl" else counts.cpu()
    dist.all_reduce(summed)                                            # the ranks' own win counts (bench.py's self-check), outside the path
    torch.cuda.synchronize(device)
    collective, backend_name, rows_per_rank = g.collective, str(dist.get_backend()), g.rows_per_rank
assert (full is None) == (rank != 0)
s = env.get_state()
np.savez(os.path.join(out_dir, "rank%d.npz" % rank), first=first, cnt=cnt, groups=s["groups"], nodes=s["nodes"], health=s["health"], env=s["env"],
         actions=env._actions.cpu().numpy(), totals=env.episode_stats()["totals"], counts=counts.cpu().numpy(), device=dev,
         device_name=torch.cuda.get_device_name(dev))
if rank == 0:
    np.savez(os.path.join(out_dir, "gathered.npz"), rows=full.cpu().numpy(), wins=np.array(evg.ResultGather.win_counts(full)),
             summed=summed.cpu().numpy() if summed is not None else np.zeros(0, np.int64), rows_per_rank=np.array(rows_per_rank(full)), collective=collective, backend=backend_name)
if backend == "evg":
    g.close()
env.close()
if backend != "evg":
    dist.barrier()
    dist.destroy_process_group()
"""


def _gpus():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU)


def _ranks():
    """One rank per visible GPU, at most 4 unless EVG_TEST_MAX_RANKS says otherwise: the GPU boxes of this pool allow six processes of one user on the GPUs
    at once (ranks + this pytest process), and a run that exceeds it is killed as a whole.  EVG_TEST_MAX_RANKS=8 on an 8-GPU node without that guard runs
    BASELINE config 4 at its full size."""
    return min(_gpus(), int(os.environ.get("EVG_TEST_MAX_RANKS", "4")))


def _run_ranks(tmp_path, oracle_mod, backend, world, ndev, total, seed=20261005, steps=170):
    import everglades_amd as evg
    script = tmp_path / "rank_child.py"
    script.write_text(_RANK_CHILD)
    port = 29900 + os.getpid() % 300
    envv = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    # fresh processes, started before anything touches the GPU in them
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(total), str(seed), str(steps), str(tmp_path), ROOT, backend, str(ndev)],
                              env=envv)
             for r in range(world)]
    # ... and meanwhile ONE oracle run over all envs (global ids 0 .. total - 1)
    oracle_mod.lib().evo_set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ora = oracle_mod.Oracle(total, seed=seed, auto_reset=True)
    ora.reset()
    for t in range(steps):
        a = ora.random_actions()
        ora.step_noobs(a)
    ost, os_ = ora.episode_stats(), ora.get_state()
    assert ost["totals"][0] >= total                                     # every env finished at least one episode
    for p in procs:
        assert p.wait(timeout=900) == 0
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    g = np.load(tmp_path / "gathered.npz")
    # the shards: contiguous global ids, every shard's final state and last orders == the oracle's slice
    assert [(int(p["first"]), int(p["cnt"])) for p in parts] == [evg.shard_range(total, world, r) for r in range(world)]
    for r, p in enumerate(parts):
        sl = slice(int(p["first"]), int(p["first"]) + int(p["cnt"]))
        for k in ("groups", "nodes", "health", "env"):
            assert np.array_equal(p[k], os_[k][sl]), ("shard", r, k)
        assert np.array_equal(p["actions"], a[sl]), ("orders of the last turn, shard", r)
    assert np.array_equal(sum(p["totals"] for p in parts), ost["totals"])
    # what the collective delivered to rank 0: the rows of ALL envs in global order == the oracle's episode results
    rows = g["rows"]
    assert rows.shape == (total, 4)
    assert np.array_equal(rows[:, 2].astype(np.int8), ost["winner"]) and np.array_equal(rows[:, 3].astype(np.int32), ost["length"])
    assert np.allclose(rows[:, :2], ost["returns"], rtol=0, atol=1e-4)
    w = ost["winner"]
    want = [int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), int((w < 0).sum())]
    assert g["wins"].tolist() == want and sum(want[:3]) == total                                          # win rule evaluate.py:155-160
    assert g["summed"].size == 0 or g["summed"].tolist() == want                                          # (the all-reduced counts of the torch forms)
    assert g["rows_per_rank"].tolist() == [int(p["cnt"]) for p in parts] and str(g["collective"]) == "gather"
    assert sum(p["counts"] for p in parts).tolist() == want
    return parts, g


def test_ranks_over_gloo_on_one_gpu_vs_oracle(tmp_path, oracle_mod):
    """two ranks sharing GPU 0, uneven shards (3001 + 3000 envs: not a multiple of the 32 envs of a wavefront)"""
    _run_ranks(tmp_path, oracle_mod, "gloo", world=2, ndev=1, total=6001)


def test_one_rank_over_rccl_vs_oracle(tmp_path, oracle_mod):
    """the RCCL code path (process group with a device id, device send / receive buffers, the gather's RCCL launch) with a one-rank group"""
    parts, g = _run_ranks(tmp_path, oracle_mod, "nccl", world=1, ndev=1, total=4099)
    assert str(g["backend"]) == "nccl"


@pytest.mark.skipif(_gpus() < 2,
                    reason="needs >= 2 GPUs: one rank per GPU over RCCL / xGMI (this box has %d); the gloo and one-rank RCCL forms of the same program ran "
                           "instead" % _gpus())
def test_one_rank_per_gpu_over_rccl_vs_oracle(tmp_path, oracle_mod):
    """One rank per visible GPU (_ranks(): at most 4 by default), backend nccl = RCCL over xGMI.  8 ranks: BASELINE config 4 at its full size,
    8 x 65 536 envs; fewer: uneven shards of about 20 000 envs."""
    world = _ranks()
    total = 8 * 65536 if world == 8 else world * 20000 + 3
    parts, g = _run_ranks(tmp_path, oracle_mod, "nccl", world=world, ndev=world, total=total)
    assert str(g["backend"]) == "nccl" and sorted(int(p["device"]) for p in parts) == list(range(world))


def test_one_rank_over_the_abis_own_rccl_gather_vs_oracle(tmp_path, oracle_mod):
    """evg_comm_unique_id / evg_comm_init / evg_gather_returns (RCCL opened by libevg.so itself, no torch.distributed) with a one-rank communicator"""
    parts, g = _run_ranks(tmp_path, oracle_mod, "evg", world=1, ndev=1, total=4099)
    assert str(g["backend"]) == "evg_gather_returns"


@pytest.mark.skipif(_gpus() < 2, reason="needs >= 2 GPUs: one rank per GPU over RCCL / xGMI (this box has %d)" % _gpus())
def test_one_rank_per_gpu_over_the_abis_own_rccl_gather_vs_oracle(tmp_path, oracle_mod):
    """the same through evg_gather_returns, one rank per visible GPU (_ranks()): what a consumer without torch.distributed runs"""
    world = _ranks()
    total = 8 * 65536 if world == 8 else world * 20000 + 3
    parts, g = _run_ranks(tmp_path, oracle_mod, "evg", world=world, ndev=world, total=total)
    assert str(g["backend"]) == "evg_gather_returns" and sorted(int(p["device"]) for p in parts) == list(range(world))
