#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec at 65 536 concurrent DemoMap games per MI355X, random vs random.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E]

One "step" = one pass of the hot path over one batch: on-device random action generation for both
players of every env (the input generator, agents/State_Machine/random_actions.py) + the fused
env-step kernel (game_turn + observations + rewards, auto-reset on).  Inputs and outputs stay in HBM.
For N > 1 the driver launches one rank per GPU (torch.distributed.run); environments shard by
contiguous global id (weak scaling: 65 536 per GPU) and the only collective is the gather of episode
results at episode boundaries (RCCL all-gather over xGMI).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per env-step = read + write of the 1 780 B state
# (health f64[2][100] 1 600 + groups 144 + nodes 33 + turn/status 2) + actions 112 + obs f32 840 + 18.
ALGO_BYTES_PER_ENV_STEP = 4530
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cores():
    """CPUs this job may actually use: the scheduler affinity capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def pmc_traffic(n_local, turns_per_launch):
    """HBM bytes per TURN of the step kernel from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md, calibrated in the same passes).
    Counters cannot be read live from inside this process, so the newest committed figure for this config and launch
    form is reported; None when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            if int(d.get("envs", -1)) != n_local:
                continue
            forms = d.get("forms")
            if forms:
                form = forms["persistent"] if turns_per_launch > 1 else forms["one_launch_per_turn"]
                return float(form["corrected_bytes_per_turn"]), os.path.relpath(f, ROOT)
            if turns_per_launch == 1:
                return float(d["corrected_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def cpu_parity(seed, turns, gpu_stats, n=4096):
    """the CPU restatement plays the same games (same seed, global env ids 0..n-1, same on-device action
    generator contract) for the same number of turns; the per-env results of the last finished episode and the win
    counters must equal the GPU's.  Rank 0, outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as om
    om.lib().evo_set_num_threads(usable_cores())
    o = om.Oracle(n, seed=seed, auto_reset=True)
    o.reset()
    for _ in range(turns):
        o.step_noobs(o.random_actions())
    st = o.episode_stats()
    same = (np.array_equal(st["winner"], gpu_stats["winner"][:n]) and np.array_equal(st["length"], gpu_stats["length"][:n]) and
            np.allclose(st["returns"], gpu_stats["returns"][:n], rtol=0, atol=1e-4))
    wins = [int((st["winner"] == k).sum()) for k in (0, 1, 2)]
    gwins = [int((gpu_stats["winner"][:n] == k).sum()) for k in (0, 1, 2)]
    return {"envs": n, "turns": turns, "equal": bool(same), "cpu_wins_p0_p1_tie": wins, "gpu_wins_p0_p1_tie": gwins}


def cpu_baseline(seed, budget_s=12.0):
    """The CPU oracle (C port of the reference's turn loop, oracle/evg_oracle.c) timed on this box's host
    cores with OpenMP over envs: same workload (random vs random incl. action generation and observations,
    auto-reset), bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as C
    import numpy as np
    import oracle as om
    L = om.lib()
    cores = usable_cores()
    L.evo_set_num_threads(cores)
    n = 8192
    o = om.Oracle(n, seed=seed, auto_reset=True)
    o.reset()
    a = np.zeros((n, 2, 7, 2), np.int32)
    obs = np.zeros((n, 2, 105), np.float64)
    rew = np.zeros((n, 2), np.float64)
    done = np.zeros(n, np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    for _ in range(10):          # warm-up (threads, page faults)
        L.evo_random_actions(o.h, p(a)); L.evo_step(o.h, p(a), p(obs), p(rew), p(done), None, None, None)
    o = om.Oracle(n, seed=seed, auto_reset=True)
    o.reset()
    t0, turns = time.perf_counter(), 0
    while True:
        for _ in range(50):
            L.evo_random_actions(o.h, p(a)); L.evo_step(o.h, p(a), p(obs), p(rew), p(done), None, None, None)
        turns += 50
        dt = time.perf_counter() - t0
        if (dt >= budget_s and turns % 150 == 0) or dt >= 2.5 * budget_s:
            break
    return dict(value=n * turns / dt, unit="env-steps/s", cores=cores, kind="port",
                sample="%d envs x %d turns (random vs random, action generation + step + f64 observations, auto-reset), "
                       "C oracle with OpenMP over envs; the Python reference itself runs 529-554 env-steps/s on one core "
                       "(BASELINE.md, measured in the build container)" % (n, turns))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=450)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--envs", type=int, default=65536, help="concurrent games per GPU")
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--obs-dtype", default="float32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--turns-per-launch", type=int, default=150,
                    help="consecutive turns each wavefront plays per launch of the step kernel (persistent rollout form; "
                         "1 = one launch per turn). Outputs are written every turn in both forms and the results are identical.")
    ap.add_argument("--workload", default="random", choices=["random", "scripted"],
                    help="random: BASELINE metric config (random vs random); scripted: BASELINE config 5 (cycle_rush_turn25 vs swarm)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started directly instead of through torch.distributed.run: launch the ranks as child processes (nothing has
        # touched the GPU yet) and leave with their exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()      # one rank per GPU; wraps only in single-GPU rehearsals
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    import everglades_amd as evg

    n_local = args.envs
    total = n_local * world
    first, cnt = evg.shard_range(total, world, rank)
    assert cnt == n_local
    env = evg.EvergladesVecEnv(n_local, device=device, seed=args.seed, env_id_base=first, obs_dtype=args.obs_dtype, auto_reset=True)
    env.reset()
    stats_dev = env.episode_stats_device()
    period = 150                                   # episode length of random vs random: gather at episode boundaries

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier(device_ids=[dev_index]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(device)

    def run(nsteps, timed, tpl=None):
        """nsteps turns through the native rollout driver (evg_rollout_random / evg_rollout_policies: launches of the step
        kernel with the agents of both seats fused in, enqueued from C on torch's current stream), split at multiples of 150
        turns (the episode length of random vs random) where the episode results are gathered.  Returns the summed
        step-kernel time in ms (HIP events recorded on that stream around the step-kernel launches) and the last gather."""
        tpl = args.turns_per_launch if tpl is None else tpl
        nonlocal turn_counter
        kernel_ms_sum, gathered = 0.0, None
        left = nsteps
        while left > 0:
            chunk = min(left, period - turn_counter % period)
            out = (env.rollout_random(chunk, time_kernel=timed, turns_per_launch=tpl) if args.workload == "random" else
                   env.rollout_policies(chunk, "cycle_rush_turn25", "swarm", time_kernel=timed, fused=True, turns_per_launch=tpl))
            if timed:
                kernel_ms_sum += out[-1] * chunk
            turn_counter += chunk
            left -= chunk
            if turn_counter % period == 0:
                gathered = evg.gather_episode_results(stats_dev["returns"], stats_dev["length"], stats_dev["winner"], total, count_wins=False)
        return kernel_ms_sum, gathered

    turn_counter = 0
    run(args.warmup, False)

    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    kernel_ms_sum, gathered = run(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    step_kernel_ms = kernel_ms_sum / args.steps

    # for reference, outside the timed region: the same rollout with one launch per turn (what env.step() costs per call)
    per_turn_launch = None
    if args.turns_per_launch > 1 and world == 1:
        barrier()
        t1 = time.perf_counter()
        k1, _ = run(150, True, tpl=1)
        barrier()
        d1 = time.perf_counter() - t1
        per_turn_launch = {"env_steps_per_s": total * 150 / d1, "ms_per_step": d1 / 150 * 1e3, "kernel_ms": k1 / 150}

    if rank == 0:
        value = total * args.steps / dt
        achieved = ALGO_BYTES_PER_ENV_STEP * n_local / (step_kernel_ms * 1e-3) / 1e9
        st = env.episode_stats()
        tpl = args.turns_per_launch
        # the committed PMC passes are of the random-vs-random workload
        traffic_turn, traffic_src = pmc_traffic(n_local, tpl) if args.workload == "random" else (None, None)
        traffic = traffic_turn * tpl if traffic_turn else None
        out = {
            "metric": "env-steps/sec at 65536 concurrent DemoMap games, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            "config": {"workload": ("%d concurrent DemoMap games per GPU, random_actions vs random_actions drawn on device (fused into the "
                                    "step kernel, orders written to an [N,2,7,2] tensor; persistent rollout form, see turns_per_launch), auto-reset, obs %s [N,2,105]" if args.workload == "random" else
                                    "%d concurrent DemoMap games per GPU, on-device Cycle_BRush_Turn25 vs SwarmAgent (BASELINE config 5; both bots fused into the "
                                    "step kernel, orders written out; episodes end by BaseCapture after 84-94 turns), auto-reset, obs %s [N,2,105]") % (n_local, args.obs_dtype),
                       "envs_per_gpu": n_local, "total_envs": total,
                       "turns_per_launch": args.turns_per_launch,
                       "one_launch_per_turn": per_turn_launch, "parallelism": "env-sharded x%d" % world,
                       "episodes_finished_rank0": int(st["totals"][0]),
                       "wins_p0_p1_tie_rank0": [int(x) for x in st["totals"][1:]],
                       "gathered_wins_all_ranks": list(evg.win_counts(gathered)) if gathered is not None else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "achieved_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n_local * tpl, "turns_per_launch": tpl,
                         "measured_traffic_GBps": (traffic_turn / (step_kernel_ms * 1e-3) / 1e9) if traffic_turn else None,
                         "note": "achieved uses the ALGORITHMIC bytes of SURVEY 8(d); it exceeds 1.0 of peak when the kernel moves fewer "
                                 "bytes than that accounting (health rows are only touched where combat hits): compare traffic",
                         "kernel": "evg_step_kernel", "kernel_ms": step_kernel_ms, "kernel_ms_is": "launch duration / turns played by the launch",
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.seed)
            if args.workload == "random" and turn_counter <= 1500:
                out["cpu_baseline"]["same_games_as_gpu"] = cpu_parity(args.seed, turn_counter, st)
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
