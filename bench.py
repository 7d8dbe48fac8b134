#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec at 65 536 concurrent DemoMap games per MI355X, random vs random.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E]

One "step" = one pass of the hot path over one batch: on-device random action generation for both
players of every env (the input generator, agents/State_Machine/random_actions.py) + the fused
env-step kernel (game_turn + observations + rewards, auto-reset on).  Inputs and outputs stay in HBM.

Window.  Before anything is timed (and whatever --warmup says) the batch is brought to a DESYNCHRONISED
steady state: during a 150-turn pre-roll env e is restarted at pre-roll turn phase(e) = hash(e) mod 150
(a multiplicative hash of the global env id, so neighbouring envs get unrelated phases), so afterwards the
episode phases of the batch -- and of the 32 envs of every wavefront -- are spread uniformly over 0..149 and
every timed turn sees the episode-average mix of early (few fights) and late (many fights) positions, with
~1/150 of the envs resetting per turn.  The MIX of positions is therefore the same for any K; the launch shape is not:
the K timed turns are ceil(K / 150) launches of the persistent step kernel, and a launch lasts as long as its slowest
wavefront plus a start-up and a host synchronisation, which weigh more on a 20-turn launch (--steps 20: about 16 us per
step) than on 150-turn ones (the default --steps 450: about 14 us per step).  Both shapes are kept under profiles/.

Clock.  The kernel's time is a constant number of shader CYCLES, and the shader clock of a GPU that has just started working
needs 30-40 ms of load to reach the value it then sustains (1.9-2.1 -> 2.35 GHz: profiles/r05_d_default_run_kernel_stats.csv).
So that the timed regions measure the rollout rate and not the governor's ramp, a scratch handle of the same size plays untimed
rollouts for --clock-warmup-ms (default 80) right before the W warm-up steps; the line reports it (timing.clock_warmup_ms), and
--clock-warmup-ms 0 gives the cold figure.

Besides the headline (persistent rollout form) the line carries, under config, three legs that pay one step launch per turn --
what a Gym consumer gets from env.step(): `one_launch_per_turn` (orders drawn inside the step kernel),
`caller_actions_per_turn` (per turn evg_random_actions into a caller tensor, then evg_step(actions): the reference's
loop evaluate.py:143-152 with both policies' output arriving in a tensor) and `learner_vs_bot_per_turn` (the loop the
reference's scripts actually run, evaluate.py:85-93,143-152: a caller on seat 0 whose 7 rows arrive in a tensor -- stand-in:
evg_random_actions_seat --, an on-device bot on seat 1 evaluated INSIDE the step kernel, only the caller's observation
written: evg_step_vs_policy) -- each with its own roofline object.

The stdout line is the COMPACT form of the result (every number, ~3 KB: a driver that keeps only the tail of the output must
still see one whole JSON object); `--details FILE` also writes the full object with its explanatory strings (~8 KB), which is
what tools/bench_set.sh commits under profiles/.

For N > 1 the driver launches one rank per GPU (torch.distributed.run); environments shard by
contiguous global id (weak scaling: 65 536 per GPU) and the only collective is the gather of episode
results (one RCCL gather to rank 0 over xGMI, 16 bytes per env).  Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per env-step = read + write of the 1 780 B state
# (health f64[2][100] 1 600 + groups 144 + nodes 33 + turn/status 2) + actions 112 + obs f32 840 + 18.
# Kept as a NAMED secondary figure only: the kernel does not move these bytes (the state stays on chip in the
# persistent form and health rows are touched only where combat hits), so it is not the roofline numerator.
SURVEY_ALGO_BYTES_PER_ENV_STEP = 4530
# What one env-step MUST write/read with this layout: observations f32 2x105x4 = 840, orders 2x7x2xi32 = 112,
# reward 8 + done 1 + winner 1 + scores 8 + status 1 = 19  ->  971 B, plus the float64 health rows of the groups
# that were hit (64 B read + 64 B written per group row, 96 for group 11) -- see DESIGN.md section 3.
MANDATORY_OUTPUT_BYTES = {"float32": 971, "float64": 971 + 840, "int16": 971 - 420}
# the learner-seat turn (evg_step_vs_policy): one seat's observation 420 (f32), the caller's orders 56 written by its policy and 56 read, the same 19
MANDATORY_OUTPUT_BYTES_LEARNER = {"float32": 420 + 112 + 19, "float64": 840 + 112 + 19, "int16": 210 + 112 + 19}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 4.0   # 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction: what a SIMD sustains with 2 or 4
                                                     # waves on this kernel's instruction mix (tools/micro/inst_rate.hip, profiles/r02_q_*: 4.3-4.8 cycles per
                                                     # instruction per SIMD for mul / f64 / bfe / perm / cndmask / DPP, 2.7 for plain add / xor; the guide's
                                                     # 2-cycle SIMD-32 issue holds for the simplest ops only).  SQ_ACTIVE_INST_VALU = 1 quad-cycle per
                                                     # instruction.
REFERENCE_PYTHON_ENV_STEPS_PER_S = (529, 554)   # the reference's own Python turn loop on ONE core (measured in the build container)
NOMINAL_MS_PER_STEP = 0.02   # for the repeat rule of the timed region only (one rule for every rank, box and batch size)
PHASES = 150                 # episode length of random vs random (server.py:321): the pre-roll spreads phases over it


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


def kernel_source_hash():
    """Identifies the kernel sources the running libevg.so was built from (build() rebuilds it from them): the committed
    counter passes under profiles/ carry the same hash, and figures from another build are not used.  (One definition, shared with
    tools/_prof.py: everglades_amd._lib.kernel_source_hash.)"""
    import everglades_amd
    return everglades_amd._lib.kernel_source_hash()


def committed_counters(kind, n_local, workload, obs_dtype, variant=None):
    """Newest committed counter summary (profiles/*_<kind>.json: tools/pmc_summary.py / tools/sq_summary.py) of THIS build
    (same kernel-source hash), batch size, workload and observation dtype; None when there is none.  Counters cannot be read
    from inside the benchmarked process, so the bench prices its own launches with the per-env-step figures of those passes."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s.json" % kind)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel_source_hash") == kernel_source_hash() and int(d.get("envs", -1)) == n_local and \
                d.get("workload", "random") == workload and d.get("obs_dtype", "float32") == obs_dtype and d.get("variant") == variant:
            d["_file"] = os.path.relpath(f, ROOT)
            return d
    return None


def episode_phase(ids):
    """phase(e) in 0..149 of global env id e (numpy or torch int64 array): Knuth's multiplicative hash, so that the envs of one
    wavefront (32 consecutive ids) get unrelated phases"""
    return (((ids * 2654435761) & 0xFFFFFFFF) >> 8) % PHASES


def desynchronise(env, first_id, workload, rollout):
    """150 turns, one launch per turn; after turn j the envs with phase(global id) = j start a new episode."""
    import torch
    phase = episode_phase(torch.arange(first_id, first_id + env.num_envs, device=env.device, dtype=torch.int64))
    for j in range(PHASES):
        rollout(1, False, 1)
        env.reset(mask=(phase == j).to(torch.uint8))


def cpu_parity(seed, n, steps_after_preroll, gpu_stats, gpu_state):
    """the CPU restatement plays the same games (same seed, global env ids 0..n-1, same on-device action
    generator contract, same desynchronising pre-roll) for the same number of turns; the per-env results of the last finished
    episode, the win counters and the whole final state (groups, nodes, float64 health) must equal the GPU's, for EVERY env
    of rank 0's shard.  Outside the timed region."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as om
    om.lib().evo_set_num_threads(usable_cores())
    o = om.Oracle(n, seed=seed, auto_reset=True)
    o.reset()
    phase = episode_phase(np.arange(n, dtype=np.int64))
    for j in range(PHASES):
        o.step_noobs(o.random_actions())
        o.reset(mask=(phase == j).astype(np.uint8))
    for _ in range(steps_after_preroll):
        o.step_noobs(o.random_actions())
    st = o.episode_stats()
    same = (np.array_equal(st["winner"], gpu_stats["winner"][:n]) and np.array_equal(st["length"], gpu_stats["length"][:n]) and
            np.allclose(st["returns"], gpu_stats["returns"][:n], rtol=0, atol=1e-4) and np.array_equal(st["totals"], gpu_stats["totals"]))
    os_ = o.get_state()
    state_same = all(np.array_equal(os_[k], gpu_state[k][:n]) for k in ("groups", "nodes", "health", "env"))
    wins = [int((st["winner"] == k).sum()) for k in (0, 1, 2)]
    gwins = [int((gpu_stats["winner"][:n] == k).sum()) for k in (0, 1, 2)]
    return {"envs": n, "turns": PHASES + steps_after_preroll, "equal": bool(same and state_same), "episode_results_equal": bool(same),
            "final_state_equal_incl_float64_health": bool(state_same), "cpu_wins_p0_p1_tie": wins, "gpu_wins_p0_p1_tie": gwins}


def cpu_baseline(seed, budget_s=12.0):
    """The CPU oracle (C port of the reference's turn loop, oracle/evg_oracle.c) timed on this box's host
    cores with OpenMP over envs: same workload (random vs random incl. action generation and float64 observations --
    the reference's dtype --, auto-reset), bounded sample."""
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
    return dict(value=n * turns / dt, unit="env-steps/s", cores=cores, kind="port", obs_dtype="float64",
                sample="%d envs x %d turns (random vs random, action generation + step + f64 observations, auto-reset), "
                       "C oracle with OpenMP over envs" % (n, turns),
                # the reference ITSELF (pure Python; it cannot travel to the GPU box): measured in the build container, BASELINE.md section 2
                reference_python_env_steps_per_s=list(REFERENCE_PYTHON_ENV_STEPS_PER_S),
                reference_python_source="BASELINE.md section 2: test_battle.py loop, 1 core of the build container (8 cores: 3208); not re-timed on the GPU box")


def expected_if_wire_free(world, steps):
    """What an N-GPU line should show if the wire cost nothing: N x the committed ONE-rank RCCL rehearsal of the same --steps (process group, pack kernel,
    gather and all-reduce launches all paid, nothing to transfer), next to the committed plain one-GPU line of the same shape.  None when profiles/
    holds no such pair.  A first real multi-GPU run is read against this."""
    def newest(pattern):
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
            try:
                d = json.load(open(f))
            except Exception:
                continue
            if d.get("steps") == steps:
                return d, os.path.relpath(f, ROOT)
        return None, None
    reh, reh_f = newest("*_bench_rehearse_rccl_1rank.json")
    one, one_f = newest("*_bench_driver_shape.json" if steps == 20 else "*_bench_final.json")
    if not reh:
        return None
    dd = reh.get("distributed") or {}
    out = {"value_if_wire_free": world * reh["value"], "per_gpu": reh["value"], "from": "1-rank RCCL rehearsal " + reh_f,
           "collective_us_1rank": dd.get("collective_us"), "step_launches_us_1rank": dd.get("step_launches_us")}
    if dd.get("collective_us") and dd.get("step_launches_us"):
        # both stream times come from ONE run (two single samples of a 0.4 ms region on different boxes differ by +-8 %): the share of the step launches in
        # launches + collective path
        out["weak_scaling_efficiency_if_wire_free"] = dd["step_launches_us"] / (dd["step_launches_us"] + dd["collective_us"])
    if one:
        out.update({"one_gpu_value_same_shape": one["value"], "one_gpu_from": one_f})
    return out


def repeats_for(steps, forced=0):
    """How often the exact K-step region is timed: `forced` when given, else 9 when K x a nominal 0.02 ms < 50 ms (K <= 2 500) and 1 otherwise.  The rule
    depends on nothing a rank measures, so every rank of a multi-GPU job times the same number of regions."""
    return forced if forced > 0 else (9 if steps * NOMINAL_MS_PER_STEP < 50.0 else 1)


def median_regions(region_s):
    """indices of the median timed region(s): one for an odd count, the TWO middle ones for an even count -- `value`, `ms_per_step` and the kernel time are then
    the mean over both (a true median; round 5 took the faster of the two)"""
    order = sorted(range(len(region_s)), key=lambda i: region_s[i])
    n = len(order)
    return [order[n // 2]] if n % 2 else [order[n // 2 - 1], order[n // 2]]


def mean_over(indices, values):
    return sum(values[i] for i in indices) / len(indices)


def _r(x, n=4):
    """numbers of the compact line: n significant digits"""
    if isinstance(x, float):
        y = float("%.*g" % (n, x))
        return int(y) if abs(y) >= 1e6 and y == int(y) else y          # 4074230000 instead of 4074230000.0: the line must stay short
    return x


def compact_line(full):
    """The ONE line printed on stdout: every field of the contract and the numbers of every leg, without the explanatory strings (a driver
    that keeps only the tail of the output must still see a whole JSON object: the full object is ~8 KB, this one ~2.5 KB; --details FILE
    writes the full one, and the lines committed under profiles/ are full ones)."""
    def roof(r):
        if not r:
            return None
        keep = ("bound", "bound_detail", "achieved", "peak", "unit", "frac", "achieved_kernel_events", "frac_kernel_events", "traffic", "kernel_ms",
                "bytes_per_env_step", "bytes_source", "launches_timed",
                "survey_8d_frac", "hbm_proper_frac", "frac_algorithmic",      # (algorithmic bytes = bytes_per_env_step / traffic_over_algorithmic)
                "traffic_over_algorithmic", "note")         # (hbm_proper_source: the *_cycled_pmc_traffic.json beside bytes_source's file)
        o = {k: _r(r[k], 5) for k in keep if k in r and r[k] is not None}
        for k in ("bytes_source", "hbm_proper_source"):        # (files under profiles/)
            if isinstance(o.get(k), str) and o[k].startswith("profiles/"):
                o[k] = o[k][len("profiles/"):]
        if isinstance(o.get("bytes_source"), str) and o["bytes_source"].startswith("mandatory outputs only"):
            o["bytes_source"] = "mandatory outputs only (no PMC pass of this build under profiles/): lower bound"
        if "traffic" not in o:
            o["traffic"] = None
        # (frac: bytes / ms_per_step of the region `value` comes from; *_kernel_events: bytes / HIP-event launch time -- `frac_is` of the full object)
        if r.get("bound_detail") == "fabric":
            o["bound_detail_is"] = "L2<->InfinityCache/HBM; working set cache-resident"
        if "survey_8d_frac" in o:
            o["survey_8d_note"] = "4530 B model; >1: not applicable"
        if "note" in o:
            o["note"] = "working set cache-resident: latency/issue-bound, frac meaningless"
        bm = r.get("beyond_mall")
        if bm:      # (the full object has every pass with its source; the line keeps the fractions)
            o["beyond_mall_fracs"] = {("cycled_%dMB" % v["working_set_MB"]) if "working_set_MB" in v else k[:24]: _r(v["frac_of_8TBps"])
                                      for k, v in bm.items() if isinstance(v, dict) and "frac_of_8TBps" in v}
            w = (bm.get("whole_rounds_one_after_the_other") or {}).get("persistent")
            if w:
                o["beyond_mall_fracs"]["whole_rounds_262144_envs"] = _r(w["frac_of_8TBps"])
        return o

    def leg(l):
        if not l:
            return None
        o = {k: _r(l[k], 4) for k in ("env_steps_per_s", "ms_per_step", "kernel_ms", "turns_per_launch", "parts") if k in l}
        for sub in ("learner_vs_bot", "with_features", "with_features_fused", "with_epsilon"):
            if l.get(sub):
                o[sub] = {k: _r(l[sub][k], 4) for k in ("env_steps_per_s", "kernel_ms", "epsilon") if k in l[sub]}     # (ms_per_step: full object)
        if l.get("roofline"):
            # (bound, peak, unit and byte source: as in the main roofline object)
            o["roofline"] = {k: _r(l["roofline"][k], 4) for k in ("frac", "bytes_per_env_step") if l["roofline"].get(k) is not None}    # (achieved = frac x peak)
        return o

    c = full["config"]
    out = {k: _r(full[k], 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                       "dtype", "data")}
    if "value_cold" in full:            # the protocol of rounds 1-4 (one region, no clock warm-up) beside the median-of-R value; the long region beside both
        out["value_cold"] = _r(full["value_cold"], 5)      # (value_protocol of the full object says what each is)
    if full.get("sustained"):
        out["sustained"] = {k: _r(full["sustained"][k], 5) for k in ("turns", "launches", "region_ms", "ms_per_step", "value", "kernel_ms")}
    obs_name = c["workload"].rsplit("obs ", 1)[-1].split(" ")[0]
    out["config"] = {"workload": "%d concurrent DemoMap games per GPU, %s, persistent rollout form, auto-reset, obs %s [N,2,105]" % (
                         c["envs_per_gpu"], "random_actions vs random_actions drawn on device" if "random_actions" in c["workload"] else
                         "on-device Cycle_BRush_Turn25 vs SwarmAgent fused into the step kernel (BASELINE config 5)", obs_name),
                     "window": "desynchronised: 150-turn pre-roll + 150 settle",
                     **{k: c[k] for k in ("envs_per_gpu", "total_envs", "turns_per_launch", "launch_form", "parallelism", "kernel_source_hash",
                                          "episodes_finished_rank0",
                                          "wins_p0_p1_tie_rank0", "gathered_wins_all_ranks") if k in c},
                     "one_launch_per_turn": leg(c.get("one_launch_per_turn")), "caller_actions_per_turn": leg(c.get("caller_actions_per_turn")),
                     "learner_vs_bot_per_turn": leg(c.get("learner_vs_bot_per_turn")),
                     "learner_smart_actions_vs_bot_per_turn": leg(c.get("learner_smart_actions_vs_bot_per_turn")),
                     "pipelined_halves_per_turn": leg(c.get("pipelined_halves_per_turn")),
                     "obs_float64": leg(c.get("obs_float64")),
                     "without_observations": leg(c.get("without_observations"))}
    out["roofline"] = roof(full["roofline"])
    if "timing" in full:
        t = full["timing"]
        out["timing"] = {"repeats": t["repeats"], "reported": t["reported"], "min_ms_per_step": _r(t["min_ms_per_step"], 5),
                         "max_ms_per_step": _r(t["max_ms_per_step"], 5),
                         "min_value": _r(t["min_value"], 4), "max_value": _r(t["max_value"], 4),
                         "clock_warmup_ms": _r(t.get("clock_warmup", {}).get("ms", 0.0), 3)}
        if "value_cold" not in full:     # (lines of round 5 carried it here)
            out["timing"]["cold_value"] = _r((t.get("clock_warmup", {}).get("cold_region") or {}).get("value"), 5)
    if "roofline_valu_issue" in full:
        v = full["roofline_valu_issue"]
        out["roofline_valu_issue"] = {k: _r(v[k], 5) for k in ("bound", "frac", "valu_insts_per_wave_turn", "source")}      # (achieved / peak: full object)
        out["roofline_valu_issue"]["source"] = str(v["source"]).replace("profiles/", "")
    if "distributed" in full:
        d = dict(full["distributed"])
        d["collective"] = "pack kernel + ONE gather to rank 0 (= closing bracket); win-count self-check after the region"
        for k in ("collective_us_is", "closing_bracket", "rows_expected_per_rank"):
            d.pop(k, None)
        pr = d.get("per_rank") or []          # per rank, as columns (rank = position): an 8-rank line must still fit a driver's tail
        d["per_rank"] = {k: [_r(r[k], 4) for r in pr] for k in ("seconds", "kernel_ms_per_step", "collective_us")}
        # where every rank sat: LOCAL_RANK, device ordinal, PCI address (uuid when there is no PCI address), one name when all are the same
        for k in ("local_rank", "device_index", "pci"):
            if pr and all(k in r for r in pr):
                d["per_rank"][k] = [r[k] for r in pr]
        if pr and all("pci" in r for r in pr) and not all(r["pci"] for r in pr):
            d["per_rank"]["uuid"] = [r.get("uuid") for r in pr]
        names = sorted({r.get("name") for r in pr if r.get("name")})
        if names:
            d["device_names"] = names
        archs = sorted({r.get("gcn_arch") for r in pr if r.get("gcn_arch")})
        if archs:
            d["gcn_archs"] = archs
        if d.get("expected"):
            d["expected"] = {k: _r(v, 5) for k, v in d["expected"].items() if k in ("value_if_wire_free", "per_gpu", "weak_scaling_efficiency_if_wire_free",
                                                                                    "collective_us_1rank")}
        for k in ("collective_us", "step_launches_us", "step_share_of_region"):
            d[k] = _r(d.get(k), 5)
        out["distributed"] = d
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": _r(b["value"], 5), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "obs_dtype": b.get("obs_dtype"),
                               "sample": b["sample"].split(" (random vs random")[0] + ", random vs random + f64 obs; C port of the turn loop, OpenMP"}
        if "reference_python_env_steps_per_s" in b:      # the reference itself (pure Python, one core, build container): BASELINE.md section 2
            out["cpu_baseline"]["reference_python_env_steps_per_s"] = list(b["reference_python_env_steps_per_s"])
            out["cpu_baseline"]["reference_python_is"] = "Python reference, 1 core (BASELINE.md s2)"
        if "same_games_as_gpu" in b:
            g = b["same_games_as_gpu"]
            out["cpu_baseline"]["same_games_as_gpu"] = {k: g[k] for k in ("envs", "turns", "equal", "cpu_wins_p0_p1_tie", "gpu_wins_p0_p1_tie")}
    return out


# =====================================================================================================================================================
# pricing: pure functions of the measured times and the committed counter summaries (unit-tested on the CPU: tests/test_abi_and_host.py)
# =====================================================================================================================================================
def hbm_roofline(pmc, form_key, kernel_ms, turns_per_launch_timed, n_local, obs_dtype, workload="random", source_hash="?", region_ms_per_step=None):
    """roofline object of one launch form.  `pmc`: the committed counter summary of THIS build (committed_counters("pmc_traffic", ...)) or None.
    bytes per env-step = the form's steady-state figure + its state round trip re-scaled to the turns per launch that were timed; without a counter pass
    of this build: the unavoidable output bytes, a lower bound, and `bytes_source` says so.
      achieved / frac                      bytes x envs / the time per step of the region `value` comes from (`region_ms_per_step`: wall clock between the
                                           brackets; launch gaps and the host's synchronisation included) -- the figure that belongs to `value`
      achieved_kernel_events / frac_kernel_events   the same bytes / the launches' own duration (`kernel_ms`: HIP events on their stream), what rocprofv3's
                                           average duration of the kernel agrees with
    When no region time is given (the per-turn legs price their stream time) both pairs are the event figure."""
    mand = MANDATORY_OUTPUT_BYTES_LEARNER[obs_dtype] if form_key == "learner_vs_bot_per_turn" else MANDATORY_OUTPUT_BYTES[obs_dtype]
    t_ms = kernel_ms if region_ms_per_step is None else region_ms_per_step
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": kernel_ms, "time_ms_per_step_priced": t_ms,
         "frac_is": "bytes / ms_per_step of the region `value` comes from" if region_ms_per_step is not None else "bytes / stream time per turn (HIP events)",
         "mandatory_output_bytes_per_env_step": mand}
    form = pmc["forms"].get(form_key) if pmc else None
    if form:
        bpe = form["bytes_per_env_step_steady"] + form["state_round_trip_bytes_per_env"] / turns_per_launch_timed
        r.update({"traffic": bpe * n_local * turns_per_launch_timed, "traffic_unit": "bytes per launch", "bytes_per_env_step": bpe,
                  "bytes_source": pmc["_file"] + " [%s]" % form_key, "kernel_us_rocprof": form.get("kernel_us_per_turn"),
                  "frac_at_rocprof_kernel_time": form.get("frac_of_8TBps"), "ratio_to_mandatory_outputs": bpe / mand})
        # this DESIGN's algorithmic bytes: the outputs a turn must write + every health row combat hit, read once and written once -- the
        # rows written are what WRITE_SIZE shows beyond the outputs (persistent form).  traffic / algorithmic > 1 is over-fetch (whole 128-byte lines for
        # 64-byte rows, straddling rows)
        if form.get("WRITE_SIZE_KB_mean") and form_key == "persistent":
            rows_written = max(0.0, form["WRITE_SIZE_KB_mean"] * 1024.0 / form["turns_per_launch"] / n_local - mand)
            r["algorithmic_bytes_per_env_step"] = mand + 2.0 * rows_written
            r["traffic_over_algorithmic"] = bpe / (mand + 2.0 * rows_written)
    else:
        bpe = mand
        r.update({"traffic": None, "bytes_per_env_step": mand,
                  "bytes_source": "mandatory outputs only: a lower bound (profiles/ holds no PMC pass of this build, hash %s, for %s at %d envs, workload %s, %s "
                                  "observations)" % (source_hash, form_key, n_local, workload, obs_dtype)})
    r["achieved"] = bpe * n_local / (t_ms * 1e-3) / 1e9
    r["frac"] = r["achieved"] / HBM_PEAK_GBS
    if r.get("algorithmic_bytes_per_env_step"):
        # the same time priced with THIS DESIGN's algorithmic bytes instead of the counters' (what a reviewer recomputes: outputs + health rows touched, no
        # over-fetch): frac x algorithmic / traffic
        r["frac_algorithmic"] = r["algorithmic_bytes_per_env_step"] * n_local / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    r["achieved_kernel_events"] = bpe * n_local / (kernel_ms * 1e-3) / 1e9
    r["frac_kernel_events"] = r["achieved_kernel_events"] / HBM_PEAK_GBS
    return r


def round_working_set_bytes(n_local, obs_dtype):
    """what one round of resident workgroups (at most 65 536 envs on a whole MI355X) touches: state + both observation rows + orders + small outputs"""
    obs_b = {"float32": 4, "float64": 8, "int16": 2}[obs_dtype] * 210
    return min(n_local, 65536) * (1773 + obs_b + 112 + 32)


def mark_cache_resident(roof, pmc, n_local, obs_dtype):
    """FETCH_SIZE / WRITE_SIZE count requests between the L2s and the fabric, Infinity-Cache (256 MiB) hits included (MI355X_MICROARCH.md, "HBM"); a persistent
    launch works through its batch in rounds whose working set fits that cache.  `bound` stays the contract's "hbm" (the peak the fraction is taken of);
    `bound_detail` says that what the counters price is then the FABRIC (L2 <-> Infinity Cache / HBM), not DRAM."""
    ws = round_working_set_bytes(n_local, obs_dtype)
    if pmc and ws <= (256 << 20):
        roof["bound_detail"] = "fabric"
        roof["bound_detail_is"] = ("requests between the L2s and the Infinity Cache / HBM.  The working set of a round of resident workgroups (%.0f MB) is "
                                   "inside the 256 MiB Infinity Cache, so these bytes are NOT all DRAM traffic; the counters cannot separate cache hits.  `peak` "
                                   "is the HBM3E figure the contract asks for (8 TB/s); `hbm_proper_frac` is the same kernel made to leave the cache" % (ws / 1e6))
    return roof


def beyond_the_cache(pmc, big, cyc, prod, obs_dtype):
    """The same kernel when its launch does leave the cache: ONE number (`hbm_proper_frac`), one source -- the pass with the LARGEST cycled working set.
      cyc   the diagnostic library's chunked form forced over 262 144 envs: every env of a 723 MB working set (2.7 x the cache) is revisited once per 25-turn
            chunk -- the HBM figure proper
      prod  libevg.so itself, a plan it really launches: 131 071 envs with evg_config.cache_mib raised to 1 024 (ONE chunked launch cycling through 359 MB,
            1.34 x the cache: still mostly cache-resident, reported next to the other).  With the default budget the library refuses such a plan BY DESIGN
      big   the product's whole rounds at 262 144 envs, one after the other
    Returns (beyond_mall object, hbm_proper fields) or (None, {})."""
    if not (pmc and (big or cyc or prod)):
        return None, {}
    obs_b = {"float32": 4, "float64": 8, "int16": 2}[obs_dtype] * 210
    bm = {"infinity_cache_MB": 268}

    def cmp_form(d, k):
        fb, fs = d["forms"][k], pmc["forms"][k]
        return {"envs": d["envs"], "working_set_MB": round(d["envs"] * (1773 + obs_b + 112 + 32) / 1e6),
                "bytes_per_env_step": fb["bytes_per_env_step_steady"], "bytes_per_env_step_at_65536": fs["bytes_per_env_step_steady"],
                "ns_per_env_step": fb["kernel_us_per_turn"] * 1e3 / d["envs"], "ns_per_env_step_at_65536": fs["kernel_us_per_turn"] * 1e3 / 65536,
                "traffic_TBps": fb["traffic_TBps"], "frac_of_8TBps": fb["frac_of_8TBps"], "source": d["_file"]}
    if big:
        bm["whole_rounds_one_after_the_other"] = {k: cmp_form(big, k) for k in ("persistent", "one_launch_per_turn") if k in big["forms"] and k in pmc["forms"]}
    if cyc and "persistent" in cyc["forms"] and "persistent" in pmc["forms"]:
        bm["diag_library_chunked_over_262144_envs"] = cmp_form(cyc, "persistent")
    if prod and "persistent" in prod["forms"] and "persistent" in pmc["forms"]:
        bm["product_library_chunked_over_131071_envs_cache_mib_1024"] = cmp_form(prod, "persistent")
    hp = bm.get("diag_library_chunked_over_262144_envs") or bm.get("product_library_chunked_over_131071_envs_cache_mib_1024")
    fields = {}
    if hp:
        fields = {"hbm_proper_frac": hp["frac_of_8TBps"], "hbm_proper_source": hp["source"],
                  "hbm_proper_is": ("the persistent kernel in a launch that cycles through a working set the Infinity Cache cannot hold (%d envs, %d MB, every env "
                                    "revisited once per 25-turn chunk): the same instruction stream and the same bytes per env-step (%.0f) at %.3f instead of "
                                    "%.3f ns per env-step = %.2f TB/s = %.2f of the HBM peak"
                                    % (hp["envs"], hp["working_set_MB"], hp["bytes_per_env_step"], hp["ns_per_env_step"], hp["ns_per_env_step_at_65536"],
                                       hp["traffic_TBps"], hp["frac_of_8TBps"]))}
    return bm, fields


def valu_roofline(sq, form, n_local, step_kernel_ms):
    """VALU instructions per wave-turn (committed SQ counter pass of this build) against one wave64 instruction per 4 cycles per SIMD"""
    if not sq:
        return None
    k = sq["kernels"][form]
    insts = k["valu_insts_per_wave_turn"] * ((n_local + 31) // 32)
    ach = insts / (step_kernel_ms * 1e-3)
    return {"bound": "valu_issue", "achieved": ach, "peak": VALU_PEAK_WAVE_INSTS_PER_S, "unit": "wave64 VALU instructions/s", "frac": ach / VALU_PEAK_WAVE_INSTS_PER_S,
            "valu_insts_per_wave_turn": k["valu_insts_per_wave_turn"], "wave_cycles_per_wave_turn": k.get("wave_cycles_per_wave_turn"),
            "source": sq["_file"], "peak_is": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction"}


def summarise_regions(region_s, kernel_ms, collective_ms, steps, total):
    """The R timed regions of exactly K steps -> what the line reports: the median region (mean of the two middle ones for an even R), min / max beside it."""
    med = median_regions(region_s)
    dt = mean_over(med, region_s)
    return {"median_indices": med, "seconds": dt, "kernel_ms_sum": mean_over(med, kernel_ms),
            "collective_ms": None if collective_ms[0] is None else mean_over(med, collective_ms),
            "timing": {"repeats": len(region_s), "reported": "median region" if len(med) == 1 else "mean of the two middle regions",
                       "region_ms": [x * 1e3 for x in region_s], "min_ms_per_step": min(region_s) / steps * 1e3, "max_ms_per_step": max(region_s) / steps * 1e3,
                       "min_value": total * steps / max(region_s), "max_value": total * steps / min(region_s),
                       "rule": "R = 9 when K x 0.02 ms (nominal) < 50 ms, else 1 (--repeats N forces N); every region = exactly K steps between barrier + "
                               "synchronize brackets"}}


def device_identity(torch, dev_index, local_rank):
    """What proves on which GPU a rank sat: name, PCI address, uuid (whatever this torch build exposes), LOCAL_RANK and the device ordinal."""
    p = torch.cuda.get_device_properties(dev_index)
    pci = None
    if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = "%04x:%02x:%02x" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
    return {"local_rank": int(local_rank), "device_index": int(dev_index), "name": p.name, "pci": pci, "uuid": str(getattr(p, "uuid", "")) or None,
            "gcn_arch": getattr(p, "gcnArchName", None), "compute_units": int(p.multi_processor_count)}


def check_distinct_devices(identities, backend, rehearsal):
    """N ranks of a real run must sit on N distinct GPUs.  The identities compared are the hardware ones this torch build exposes (uuid, PCI address; the
    ordinal only when it exposes neither): the ranks count as distinct when ONE of them tells them all apart (a driver that reports one placeholder uuid
    for every card must not end the run while the PCI addresses differ), and as shared when none does.  A rehearsal on one GPU (one-rank group, or gloo
    ranks sharing the card) only records the answer."""
    keys = [k for k in ("uuid", "pci") if all(i.get(k) for i in identities)] or ["device_index"]
    seen = {k: len(set(i[k] for i in identities)) for k in keys}
    key = next((k for k in keys if seen[k] == len(identities)), keys[0])
    distinct = seen[key] == len(identities)
    if not distinct and backend == "nccl" and not rehearsal:
        raise SystemExit("%d ranks but only %d distinct GPUs (%s = %s): not a one-rank-per-GPU run"
                         % (len(identities), seen[key], key, [i[key] for i in identities]))
    return {"distinct_devices": distinct, "identified_by": key, "devices_seen": seen[key]}


# =====================================================================================================================================================
# the run
# =====================================================================================================================================================
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=450)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--envs", type=int, default=65536, help="concurrent games per GPU")
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--obs-dtype", default="float32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the one-launch-per-turn and float64-observation legs (profiling runs)")
    ap.add_argument("--turns-per-launch", type=int, default=150,
                    help="consecutive turns each wavefront plays per launch of the step kernel (persistent rollout form; "
                         "1 = one launch per turn). Outputs are written every turn in both forms and the results are identical.")
    ap.add_argument("--workload", default="random", choices=["random", "scripted"],
                    help="random: BASELINE metric config (random vs random); scripted: BASELINE config 5 (cycle_rush_turn25 vs swarm)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--rehearse-distributed", action="store_true",
                    help="diagnostics only: run the N > 1 code path (process group, pack + gather, per-rank times) with a ONE-rank group on one GPU")
    ap.add_argument("--caller-actions", action="store_true",
                    help="profiling runs: the MAIN leg runs the caller-supplied-actions path (per turn evg_random_actions into a tensor + evg_step); needs "
                         "--turns-per-launch 1")
    ap.add_argument("--learner-seat", action="store_true",
                    help="profiling runs: the MAIN leg runs the learner-seat path (per turn evg_random_actions_seat into a tensor + evg_step_vs_policy); "
                         "needs --turns-per-launch 1")
    ap.add_argument("--opponent", default="random",
                    help="the on-device bot of the learner-seat leg (a name from everglades_amd._lib.POLICY_NAMES); random = the headline's game mix")
    ap.add_argument("--epsilon", type=float, default=0.1, help="exploring probability of the evg_smart_get_action leg (DQNAgent.get_action during training)")
    ap.add_argument("--timing", default="torch", choices=["native", "torch"],
                    help="single-rank timed region: launch duration from two pre-created torch events around an untimed call, one synchronisation in the "
                         "closing bracket (default; 1.05-1.1 us per step of host time in the 20-step shape, tools/driver_shape_timing_ab.sh) or from the native "
                         "driver's own events, read inside the call, which synchronises itself (1.2-4.7 us)")
    ap.add_argument("--collective", default="torch", choices=["torch", "evg"],
                    help="N > 1: the gather of episode results through torch.distributed (default; backend nccl = RCCL) or through the library's own RCCL entry "
                         "points (evg_comm_init / evg_gather_returns: pack kernel + grouped send / receive on the launches' stream, no framework stream hop)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how often the exact K-step region is timed (value = the median region); 0 = auto: 9 when K x a nominal 0.02 ms < 50 ms (K <= 2500), "
                         "else 1")
    ap.add_argument("--clock-warmup-ms", type=float, default=80.0,
                    help="untimed: a SCRATCH handle of the same size plays rollouts for this long right before the --warmup steps, so that the timed regions run at "
                         "the shader clock a long rollout sustains (the governor needs 30-40 ms of load to get there: "
                         "profiles/r05_d_default_run_kernel_stats.csv); 0 = off.  The line carries `value_cold` (one K-step region timed BEFORE it) beside `value`")
    ap.add_argument("--sustained-launches", type=int, default=3,
                    help="the `sustained` leg of the line: this many 150-turn launches (about 6.5 ms) between the same brackets as the timed region, so that a "
                         "short --steps line also holds one region long enough to check against a clock; 0 = off")
    ap.add_argument("--cache-mib", type=int, default=0,
                    help="profiling runs: evg_config.cache_mib of the handle (the memory-side cache budget a chunked rollout launch may cycle through; 0 = the "
                         "device's)")
    ap.add_argument("--pipeline", type=int, default=2, help="parts of the double-buffered leg (everglades_amd.PipelinedVecEnv)")
    ap.add_argument("--details", default="",
                    help="also write the FULL result object (every note and source string; the stdout line is its compact form) to this file")
    ap.add_argument("--library", default=None, help="diagnostics only (tools/ab.sh): path of another build of libevg.so")
    ap.add_argument("--diag-lanes", type=int, default=0, help="diagnostics only: kernel variant of libevg_diag.so (evg_diag_configure lanes)")
    args = ap.parse_args(argv)
    if (args.caller_actions or args.learner_seat) and args.turns_per_launch != 1:
        raise SystemExit("--caller-actions / --learner-seat need --turns-per-launch 1 (orders from a tensor exist in the single-turn form only)")
    return args


class Run(object):
    """Everything the phases of one benchmark process share: the process group, the device, the measured handle in its steady state and its rollout function."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        self.dev_index = self.local_rank % torch.cuda.device_count()      # one rank per GPU; wraps only in single-GPU rehearsals
        torch.cuda.set_device(self.dev_index)
        self.device = torch.device("cuda", self.dev_index)
        self.dist_on = self.world > 1 or args.rehearse_distributed          # the N > 1 code path (a one-rank group is a rehearsal only)
        if self.dist_on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(args.backend)
        import everglades_amd as evg
        self.evg = evg
        self.n_local = args.envs
        self.total = self.n_local * self.world
        self.first, cnt = evg.shard_range(self.total, self.world, self.rank)
        assert cnt == self.n_local
        # --caller-actions / --learner-seat (profiling runs): the main leg itself pays two launches per turn
        self.main_fused = "learner" if args.learner_seat else (not args.caller_actions)
        self.played = 0                       # turns the measured handle has played since its first reset (bounds the CPU replay)

    def barrier(self):
        """barrier + torch.cuda.synchronize(): the opening bracket drains the device first so that every rank enters the barrier idle.  (The CLOSING
        bracket of the N > 1 timed region is the path's one collective itself -- the gather to rank 0, which cannot complete before every rank has
        played its steps and packed its rows -- followed by one torch.cuda.synchronize(); the job's time is the max over ranks.)"""
        torch, dist = self.torch, self.dist
        torch.cuda.synchronize(self.device)
        if self.dist_on:
            dist.barrier(device_ids=[self.dev_index]) if self.args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize(self.device)

    def make_env(self, obs_dtype):
        """A handle in the desynchronised steady state + its rollout function (nsteps, timed, turns per launch) -> kernel ms sum."""
        args, evg = self.args, self.evg
        env = evg.EvergladesVecEnv(self.n_local, device=self.device, seed=args.seed, env_id_base=self.first, obs_dtype=obs_dtype, auto_reset=True,
                                   library=args.library, diag=dict(lanes=args.diag_lanes) if args.diag_lanes else None, cache_mib=args.cache_mib)
        env.reset()

        def rollout(nsteps, timed, tpl, fused=True, observe=True, prepare=False):
            """nsteps turns through the native rollout driver (evg_rollout_random / evg_rollout_policies, enqueued from C on torch's
            current stream).  fused: the step kernel draws / evaluates the orders of both seats itself; not fused (tpl must be 1):
            per turn the action kernel(s) write the orders into a tensor and evg_step reads them -- the caller-supplied-actions path.
            Returns the summed stream time in ms (HIP events recorded on that stream: around every persistent launch, or around the
            whole loop of single-turn launches)."""
            if fused == "learner":        # per turn: evg_random_actions_seat -> tensor [N,7,2] -> evg_step_vs_policy (bot inside the step kernel)
                out = env.rollout_vs(nsteps, args.opponent, seat=0, time_kernel=timed)
                return out[-1] * nsteps if timed else 0.0
            kw = dict(time_kernel=timed, fused=fused, turns_per_launch=tpl, observe=observe, record_actions=observe, prepare=prepare)
            out = (env.rollout_random(nsteps, **kw) if args.workload == "random" else env.rollout_policies(nsteps, "cycle_rush_turn25", "swarm", **kw))
            return out[-1] * nsteps if timed else 0.0

        desynchronise(env, self.first, args.workload, rollout)
        # settle: one more episode length in the launch form that is timed (also creates its timing events)
        rollout(PHASES, True, args.turns_per_launch, self.main_fused)
        return env, rollout

    # ------------------------------------------------------------------------------------------------------------------------------------------------
    def setup(self):
        """the measured handle in its steady state, the gather's buffers and communicator (first use outside the timed region), the launch graphs"""
        args, evg, torch, dist = self.args, self.evg, self.torch, self.dist
        self.env, self.rollout = self.make_env(args.obs_dtype)
        self.played = 2 * PHASES                                        # pre-roll + settle
        self.gather = evg.ResultGather(self.n_local, self.total, self.device, force=self.dist_on)   # preallocated buffers; rank 0 receives (one RCCL gather)
        self.win_counts_dev = torch.zeros(4, dtype=torch.int64, device=self.device)               # filled by the pack kernel: win bookkeeping of this rank's rows
        self.native = None
        if self.dist_on and args.collective == "evg":
            # RCCL through the C-ABI: the communicator's id goes from rank 0 to every rank over the process group that exists anyway
            box = [evg.NativeGather.unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, device=self.device if args.backend == "nccl" else None)
            self.native = evg.NativeGather(self.env, self.total, self.world, self.rank, box[0])
        if self.dist_on:      # first use opens the RCCL channels of the gather: not part of the timed region
            self.run_collective()
        if self.main_fused is True:
            # capture + instantiate the graphs of the K-step launch shape now (nothing is played) -- and of the sustained leg's
            self.rollout(args.steps, False, args.turns_per_launch, True, prepare=True)
            if self.sustained_turns():
                self.rollout(self.sustained_turns(), False, PHASES, True, prepare=True)

    def run_collective(self):
        """the path's ONE exchange, enqueued on the launches' stream: rank 0 gets [total, 4], the others None"""
        if self.native is not None:
            return self.native()
        return self.gather(self.env.packed_episode_results(out=self.gather.buffer))

    def time_one_region(self, nsteps, tpl):
        """exactly `nsteps` steps between barrier + synchronize brackets (single rank; no events): seconds"""
        self.barrier()
        t = time.perf_counter()
        self.rollout(nsteps, False, tpl, self.main_fused)
        self.barrier()
        self.played += nsteps
        return time.perf_counter() - t

    def warm_the_clock(self):
        """Untimed, not part of the measured games.  The shader clock of an MI355X that has just started working climbs from ~1.9-2.1 GHz to the ~2.35 GHz it
        then sustains over the first 30-40 ms of load, and this kernel's time is a constant number of CYCLES (33 k per turn at 65 536 envs: the dispatch list
        with GRBM_GUI_ACTIVE in profiles/r05_d_default_run_kernel_stats.csv) -- so a region timed 10 ms after the start of the process (the driver's --steps 20
        --warmup 5) measures the governor's ramp, 14 % below what every later millisecond of a rollout gets.  A scratch handle of the same size plays 150-turn
        rollouts for --clock-warmup-ms right before the W warm-up steps; the measured handle, its games and the W / K contract are untouched.  So that the
        line shows what this is worth, ONE K-step region is timed BEFORE it with the clock the process has at that moment (what rounds 1-4 reported as the
        value): same brackets, same launches, reported as `value_cold` next to `value`."""
        args, torch = self.args, self.torch
        cw = {"requested_ms": args.clock_warmup_ms, "ms": 0.0, "turns": 0}
        self.scratch = None
        if args.clock_warmup_ms > 0 and not self.dist_on and args.steps <= 2500:
            cold_s = self.time_one_region(args.steps, args.turns_per_launch)
            cw["cold_region"] = {"ms_per_step": cold_s / args.steps * 1e3, "value": self.total * args.steps / cold_s}
        if args.clock_warmup_ms > 0:
            self.scratch = self.evg.EvergladesVecEnv(self.n_local, device=self.device, seed=args.seed + 1, env_id_base=self.first, obs_dtype=args.obs_dtype,
                                                     auto_reset=True, library=args.library)
            self.scratch.reset()
            torch.cuda.synchronize(self.device)
            t_w = time.perf_counter()
            while (time.perf_counter() - t_w) * 1e3 < args.clock_warmup_ms:
                self.scratch.rollout_random(PHASES, turns_per_launch=PHASES)
                torch.cuda.synchronize(self.device)
                cw["turns"] += PHASES
            cw["ms"] = (time.perf_counter() - t_w) * 1e3
            # (the scratch handle is released AFTER the timed regions: freeing 200 MB of device memory takes the host milliseconds during which the GPU would idle)
        cw["note"] = ("untimed rollouts of a SCRATCH handle right before the W warm-up steps: the timed regions run at the shader clock a long rollout sustains "
                      "instead of inside the governor's ramp of the first 30-40 ms (--clock-warmup-ms 0 = off)")
        return cw

    def timed_regions(self):
        """W untimed warm-up steps, then the exact K-step region R times.  A region of a few hundred microseconds (the driver's --steps 20: 0.4 ms) is ONE draw
        from a distribution whose box-to-box and run-to-run spread is +-8 %: when K x a nominal 20 us per step is below 50 ms the region is repeated R = 9
        times -- each repeat bracketed exactly like the single region (barrier + synchronize on both sides, nothing else inside) -- and value / ms_per_step come
        from the MEDIAN region; min and max are reported next to it.  (The rule must give the SAME answer on every rank -- a rank that timed one region more
        would wait in a barrier nobody else enters -- so it uses a nominal 20 us per step, not this rank's own measurement.)
        Returns (per-region wall seconds [max over ranks], per-region kernel ms of THIS rank, per-region collective ms or None, all-ranks tensor or None)."""
        args, torch, dist = self.args, self.torch, self.dist
        if args.warmup > 0:
            self.rollout(args.warmup, True, args.turns_per_launch, self.main_fused)
            self.played += args.warmup
        repeats = repeats_for(args.steps, args.repeats)
        use_events = self.dist_on or args.timing == "torch"
        if use_events:      # torch creates an event at its first record(): not inside the timed region
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(repeats)]
            for tri in evs:
                for ev in tri:
                    ev.record()
        self.gathered = None
        regions = []          # per repeat: [wall seconds, summed stream ms of the step launches, collective ms or None]
        for rep in range(repeats):
            # ---- timed region: exactly K steps; with more than one rank the path's one collective (the gather of episode results) is inside it (a single
            # rank has nothing to exchange: its results are already where rank 0 reads them)
            self.barrier()
            t0 = time.perf_counter()
            k_ms = None
            if not self.dist_on and args.timing == "native":
                # HIP events around the step-kernel launches, recorded by the native driver on the stream it launches on and read after the last one
                k_ms = self.rollout(args.steps, True, args.turns_per_launch, self.main_fused)
                self.barrier()
            elif not self.dist_on:
                # the launches are only ENQUEUED (no event read-out, no synchronisation inside the call); their duration is taken from two stream events
                # (created and recorded once before the region) after the closing bracket, whose torch.cuda.synchronize() is the one host wait of the region
                evs[rep][0].record()
                self.rollout(args.steps, False, args.turns_per_launch, self.main_fused)
                evs[rep][1].record()
                self.barrier()
            else:
                # N > 1: step launches, ONE pack kernel and ONE collective -- the gather of 16 B per env to rank 0 -- enqueued back to back on the stream,
                # nothing in between waits for the host.  The gather IS the closing barrier where it matters: rank 0 cannot complete it before every rank has
                # finished its steps and packed its rows, and the job's time is the MAX over ranks (exchanged after the region), i.e. rank 0's.
                evs[rep][0].record()
                self.rollout(args.steps, False, args.turns_per_launch, self.main_fused)
                evs[rep][1].record()
                self.gathered = self.run_collective()
                evs[rep][2].record()
                torch.cuda.synchronize(self.device)
            regions.append([time.perf_counter() - t0, k_ms, None])
        if use_events:
            for rep in range(repeats):
                regions[rep][1] = evs[rep][0].elapsed_time(evs[rep][1])      # HIP events on the stream the step kernels run on (torch's current stream)
                if self.dist_on:
                    regions[rep][2] = evs[rep][1].elapsed_time(evs[rep][2])
        self.played += args.steps * repeats
        allr = None
        if self.dist_on:
            # per-rank times of every repeat, exchanged AFTER the timed regions; a region's time is the max over ranks, the line's the median region
            mine = torch.tensor([[r[0], r[1] / args.steps, r[2] * 1e3] for r in regions], device=self.device if args.backend == "nccl" else "cpu",
                                dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(allr, mine)
            allr = torch.stack(allr).cpu()                                   # [world, repeats, 3]
            region_s = allr[:, :, 0].max(dim=0).values.tolist()
        else:
            region_s = [r[0] for r in regions]
        return region_s, [r[1] for r in regions], [r[2] for r in regions], allr

    def sustained_turns(self):
        a = self.args
        return 0 if (a.sustained_launches <= 0 or self.dist_on or self.main_fused is not True or a.turns_per_launch < 2) else a.sustained_launches * PHASES

    def sustained_leg(self):
        """`sustained`: --sustained-launches (3) launches of 150 turns between the SAME brackets as a timed region (barrier + synchronize on both sides, the
        launches only enqueued in between) -- a region of ~6.5 ms that a wall clock outside this process can still resolve, printed next to a `value` that
        may come from 0.3 ms regions (the driver's --steps 20).  Single rank, persistent form; played on the measured handle RIGHT AFTER the K-step regions
        (nothing in between lets the device idle: same shader clock), and part of the games the CPU replays afterwards."""
        args = self.args
        turns = self.sustained_turns()
        if not turns:
            return None
        torch = self.torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        self.barrier()
        t = time.perf_counter()
        e0.record()
        self.rollout(turns, False, PHASES, True)
        e1.record()
        self.barrier()
        s = time.perf_counter() - t
        self.played += turns
        return {"turns": turns, "launches": args.sustained_launches, "turns_per_launch": PHASES, "region_ms": s * 1e3, "ms_per_step": s / turns * 1e3,
                "value": self.total * turns / s, "kernel_ms": e0.elapsed_time(e1) / turns,
                "what": "%d launches of %d turns between one pair of barrier + synchronize brackets, right after the K-step regions (same handle, same games going on)"
                        % (args.sustained_launches, PHASES)}

    def snapshot_games(self):
        """The games as they stand right after the timed regions (and the sustained leg): the episode results and win counters the line reports, the rows the
        single-rank "gather" returns, and -- bounded: at most 6 000 turns since the first reset -- the whole state the CPU oracle must reproduce afterwards."""
        args = self.args
        if self.env.check_fault():                           # (never expected: a chunk hand-over fault of a launch plan; raises)
            raise SystemExit("fault")
        self.stats = self.env.episode_stats()
        if not self.dist_on:
            self.gathered = self.gather(self.env.packed_episode_results())
        self.played_at_snapshot = self.played
        self.final_state = self.env.get_state() if (self.world == 1 and not args.no_cpu_baseline and args.workload == "random" and self.played <= 6000) else None

    # ------------------------------------------------------------------------------------------------------------------------------------------------
    def per_turn_leg(self, fused):
        """150 turns, one step launch per turn, warmed; wall clock between two barriers and the stream time between two events around the whole leg (so
        kernel_ms <= ms_per_step; the kernel alone is in profiles/*_kernel_stats.csv)"""
        args = self.args
        self.rollout(16, True, 1, fused)                        # warms the single-turn instantiation (and the action kernel) and creates the events
        self.barrier()
        t1 = time.perf_counter()
        k1 = self.rollout(150, True, 1, fused)
        self.barrier()
        d1 = time.perf_counter() - t1
        return {"env_steps_per_s": self.total * 150 / d1, "ms_per_step": d1 / 150 * 1e3, "kernel_ms": k1 / 150,
                "kernel_ms_is": "stream time per turn: two HIP events around the whole 150-turn leg / 150 (launches back to back, gaps included"
                                + ("" if fused is True else "; the action kernel of the turn included") + ")",
                "launches_per_turn": 2 if fused == "learner" else (1 if fused else (2 if args.workload == "random" else 3))}

    def timed_python_loop(self, turns_fn, launches_per_turn):
        """16 warm-up + 150 timed turns of a per-turn loop driven from Python over the C-ABI: wall clock between barriers, stream time between two events"""
        torch = self.torch
        turns_fn(16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        self.barrier()
        t1 = time.perf_counter()
        e0.record()
        turns_fn(150)
        e1.record()
        self.barrier()
        d1 = time.perf_counter() - t1
        return {"env_steps_per_s": self.total * 150 / d1, "ms_per_step": d1 / 150 * 1e3, "kernel_ms": e0.elapsed_time(e1) / 150, "launches_per_turn": launches_per_turn}

    def smart_legs(self):
        """The learner-seat turn with the Smart_State family's own decode on the device: per turn evg_smart_actions(Q [N,12,5] -> 7 order rows:
        DQNAgent.get_best_actions) + evg_step_vs_policy; Q = one of 8 prepared random tensors (the stand-in for the consumer's network output: QNetwork
        59-60-60-5 in the reference, not ours).  with_features adds evg_smart_state_compact (the network's input) in front: observation -> features ->
        [network] -> orders -> step, no host or framework glue.  with_epsilon is the TRAINING turn: evg_smart_get_action = DQNAgent.get_action with
        epsilon = --epsilon (0.1), i.e. the coin + get_random_actions for the exploring envs inside the same kernel."""
        args, env, torch = self.args, self.env, self.torch
        qs = [torch.randn((self.n_local, 12, 5), device=self.device) for _ in range(8)]
        sobs = env.observe_seat(0)
        sh = torch.empty((self.n_local, 34), dtype=torch.float32, device=self.device)
        sw = torch.empty((self.n_local, 12, 13), dtype=torch.float32, device=self.device)

        def turns(features, epsilon):
            def run(n):
                for t_ in range(n):
                    if features == "kernel":
                        env.smart_state_compact(-1, sobs, sh, sw)
                    rows = env.smart_actions(qs[t_ & 7], obs=sobs) if epsilon is None else env.smart_get_action(qs[t_ & 7], epsilon, seat=0, obs=sobs)
                    # (step_vs writes the next observation into sobs -- and, fused, the next features into sh / sw: evg_step_vs_policy_smart)
                    env.step_vs(args.opponent, rows, seat=0, features=(sh, sw) if features == "fused" else None)
            return run
        leg = self.timed_python_loop(turns(None, None), 2)
        leg["with_features"] = self.timed_python_loop(turns("kernel", None), 3)
        leg["with_features_fused"] = self.timed_python_loop(turns("fused", None), 2)
        leg["with_epsilon"] = dict(self.timed_python_loop(turns(None, args.epsilon), 2), epsilon=args.epsilon)
        leg["path"] = ("per turn, from a Python loop over the C-ABI: evg_smart_actions(Q [N,12,5] f32, one-seat obs) -> [N,7,2] orders (DQNAgent.get_best_actions "
                       "on the device) -> evg_step_vs_policy(opponent `%s` inside the step kernel); with_features: evg_smart_state_compact in front (the network's "
                       "input); with_features_fused: the features written by the step launch itself (evg_step_vs_policy_smart); with_epsilon: "
                       "evg_smart_get_action instead (DQNAgent.get_action, epsilon %g: coin + get_random_actions on the device)"
                       % (args.opponent, args.epsilon))
        return leg

    def pipelined_leg(self):
        """the double-buffered consumer (everglades_amd.PipelinedVecEnv): two half-batch handles on two streams, global env ids preserved, each playing one
        launch per turn FREE-RUNNING -- what the overlapped pattern (policy on half A while half B steps) converges to with a cheap policy"""
        args, evg, torch = self.args, self.evg, self.torch
        pipe = evg.PipelinedVecEnv(self.n_local, pipeline=args.pipeline, device=self.device, seed=args.seed, env_id_base=self.first, obs_dtype=args.obs_dtype,
                                   auto_reset=True)
        pipe.reset()
        phase = episode_phase(torch.arange(self.first, self.first + self.n_local, device=self.device, dtype=torch.int64))
        for j in range(PHASES):
            pipe.rollout_random_free(1)
            pipe.reset(mask=(phase == j).to(torch.uint8))
        pipe.rollout_random_free(PHASES, time_kernel=True)
        self.barrier()
        t1 = time.perf_counter()
        kp = pipe.rollout_random_free(600, time_kernel=True)      # (600 turns: the start of one host thread per part is inside the wall clock)
        self.barrier()
        dp = time.perf_counter() - t1
        leg = {"env_steps_per_s": self.total * 600 / dp, "ms_per_step": dp / 600 * 1e3, "kernel_ms": max(kp), "turns_timed": 600, "parts": args.pipeline,
               "envs_per_part": [c for _, c in pipe.ranges],
               "kernel_ms_is": "stream time per turn of the slowest part (two HIP events around its 600 single-turn launches); ms_per_step is the wall clock per "
                               "turn of the WHOLE batch",
               "stream_ms_per_turn_of_every_part": kp, "launches_per_turn": args.pipeline,
               "what": "PipelinedVecEnv.rollout_random_free: every part plays one launch per turn (orders drawn in the step kernel) on its own stream, nothing "
                       "joins them"}
        # ... and the learner-seat turn on the same two parts: per part and turn the learner's stand-in kernel + evg_step_vs_policy (bot inside), free-running
        pipe.rollout_vs_free(16, args.opponent, seat=0, time_kernel=True)
        self.barrier()
        t1 = time.perf_counter()
        kl = pipe.rollout_vs_free(600, args.opponent, seat=0, time_kernel=True)
        self.barrier()
        dl = time.perf_counter() - t1
        leg["learner_vs_bot"] = {"env_steps_per_s": self.total * 600 / dl, "ms_per_step": dl / 600 * 1e3, "kernel_ms": max(kl), "launches_per_turn": 2 * args.pipeline,
                                 "what": "PipelinedVecEnv.rollout_vs_free: per part and turn evg_random_actions_seat + evg_step_vs_policy(opponent `%s` inside the "
                                         "step kernel)" % args.opponent}
        pipe.close()
        return leg

    def extra_legs(self):
        """Reference legs, outside the timed region (single GPU): the forms that pay one step launch per turn (what env.step() costs per call), the learner's
        turn with the Smart_State decode, the persistent form without observations, two half-batch handles, the reference's own observation dtype."""
        args = self.args
        legs = dict.fromkeys(("one_launch_per_turn", "caller_actions_per_turn", "learner_vs_bot_per_turn", "learner_smart_actions_vs_bot_per_turn",
                              "pipelined_halves_per_turn", "obs_float64", "without_observations"))
        if self.world != 1 or args.no_extra_legs:
            return legs
        mf, tpl = self.main_fused, args.turns_per_launch
        if tpl > 1 or mf is not True:
            legs["one_launch_per_turn"] = self.per_turn_leg(True)
        if mf != "learner":
            leg = legs["learner_vs_bot_per_turn"] = self.per_turn_leg("learner")
            leg["path"] = ("per turn: evg_random_actions_seat -> caller tensor [N,7,2] (seat 0) -> evg_step_vs_policy(opponent = on-device `%s` on seat 1, "
                           "evaluated inside the step kernel; only seat 0's observation [N,105] written) -- evaluate.py:85-93,143-152 with a learner on one seat"
                           % args.opponent)
        if mf is True and args.workload == "random":
            legs["learner_smart_actions_vs_bot_per_turn"] = self.smart_legs()
        if mf is True:
            leg = legs["caller_actions_per_turn"] = self.per_turn_leg(False)
            leg["path"] = ("per turn: evg_random_actions -> caller tensor [N,2,7,2] -> evg_step(actions) (evaluate.py:143-152 with on-device agents)"
                           if args.workload == "random" else
                           "per turn: evg_scripted_actions x 2 (reading the previous observations) -> caller tensor -> evg_step(actions)")
        if tpl > 1 and mf is True:
            # the persistent form without observations and without recording the orders (evg_rollout_*(obs_out = NULL, actions_buf = NULL)): what the
            # evaluation harness runs (everglades_amd.evaluate: it reads only the episode results, evaluate.py:143-181)
            self.rollout(16, True, tpl, True, False)
            self.barrier()
            t1 = time.perf_counter()
            kq = self.rollout(150, True, tpl, True, False)
            self.barrier()
            dq = time.perf_counter() - t1
            legs["without_observations"] = {"env_steps_per_s": self.total * 150 / dq, "ms_per_step": dq / 150 * 1e3, "kernel_ms": kq / 150, "turns_per_launch": tpl,
                                            "what": "persistent rollout, no observation image / write-out, orders not recorded: rewards, done flags, scores and "
                                                    "episode results only"}
        if mf is True and args.workload == "random" and self.n_local >= 64:
            legs["pipelined_halves_per_turn"] = self.pipelined_leg()
        if args.obs_dtype != "float64":
            env64, rollout64 = self.make_env("float64")
            rollout64(8, True, tpl, mf)
            self.barrier()
            t1 = time.perf_counter()
            k64 = rollout64(150, True, tpl, mf)
            self.barrier()
            d64 = time.perf_counter() - t1
            legs["obs_float64"] = {"env_steps_per_s": self.total * 150 / d64, "ms_per_step": d64 / 150 * 1e3, "kernel_ms": k64 / 150, "turns_per_launch": tpl}
            env64.close()
        return legs

    # ------------------------------------------------------------------------------------------------------------------------------------------------
    def price_roofline(self, step_kernel_ms, ms_per_step, legs):
        """the roofline objects of the main leg and of the per-turn legs, from the committed counter passes of this build"""
        args, env = self.args, self.env
        n_local, tpl = self.n_local, args.turns_per_launch
        launches = (args.steps + tpl - 1) // tpl
        h = kernel_source_hash()
        pmc = committed_counters("pmc_traffic", n_local, args.workload, args.obs_dtype)
        sq = committed_counters("sq_counters", n_local, args.workload, args.obs_dtype)
        mand = MANDATORY_OUTPUT_BYTES[args.obs_dtype]
        main_form = "persistent" if tpl > 1 else ("caller_actions_per_turn" if args.caller_actions
                                                  else ("learner_vs_bot_per_turn" if args.learner_seat else "one_launch_per_turn"))
        roof = hbm_roofline(pmc, main_form, step_kernel_ms, args.steps / launches, n_local, args.obs_dtype, args.workload, h, region_ms_per_step=ms_per_step)
        n_launch, plan_text = env.launch_plan(tpl)
        survey_rate = SURVEY_ALGO_BYTES_PER_ENV_STEP * n_local / (step_kernel_ms * 1e-3) / 1e9
        roof.update({"kernel": plan_text, "kernel_launches_per_rollout_launch": n_launch, "launch_form": main_form,
                     "kernel_ms_is": ("HIP-event duration of every timed launch (two events on the stream it is launched on), summed / K" if tpl > 1
                                      else "stream time per turn (two HIP events around the timed launches)") if not self.dist_on else
                                     "stream time per turn: two HIP events on the launches' stream around all timed launches / K",
                     "launches_timed": launches, "turns_per_launch_timed": args.steps / launches, "env_steps_per_launch": n_local * args.steps / launches,
                     "launch_ms": step_kernel_ms * args.steps / launches,
                     "bytes_source_is": "rocprofv3 PMC passes of this build (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction, calibrated in the same passes), per "
                                        "env-step, times the env-steps of the timed launches",
                     "survey_8d_bytes_per_env_step": SURVEY_ALGO_BYTES_PER_ENV_STEP, "survey_8d_rate_GBps": survey_rate, "survey_8d_frac": survey_rate / HBM_PEAK_GBS,
                     "survey_8d_note": "SURVEY 8(d)'s byte model (the whole 1 780-byte state read + written every turn: 4 530 B per env-step) priced at this kernel "
                                       "time: ABOVE 1, i.e. the model is not applicable to this design -- the persistent form keeps group / node / stamp words on "
                                       "chip for the launch and touches only the float64 health rows combat hits; the game work itself is checked against the oracle "
                                       "(cpu_baseline.same_games_as_gpu).  Never the roofline numerator"})
        if roof.get("bytes_per_env_step"):
            roof["ratio_survey_8d_to_measured"] = SURVEY_ALGO_BYTES_PER_ENV_STEP / roof["bytes_per_env_step"]
        if pmc and roof["bytes_per_env_step"] < 0.5 * mand:
            roof["note"] = ("the counters see fewer bytes than the kernel writes: the working set of this batch (%.0f MB) stays in L2 / Infinity Cache, the launch "
                            "is latency- and issue-bound and `frac` says nothing about it" % (n_local * (1773 + 4 * 210 + 112) / 1e6))
        mark_cache_resident(roof, pmc, n_local, args.obs_dtype)
        if n_local == 65536:
            bm, fields = beyond_the_cache(pmc, committed_counters("pmc_traffic", 262144, args.workload, args.obs_dtype),
                                          committed_counters("pmc_traffic", 262144, args.workload, args.obs_dtype, variant="cycled"),
                                          committed_counters("pmc_traffic", 131071, args.workload, args.obs_dtype, variant="product_cycled"), args.obs_dtype)
            roof.update(fields)
            if bm:
                roof["beyond_mall"] = bm
        for key in ("one_launch_per_turn", "caller_actions_per_turn", "learner_vs_bot_per_turn"):
            leg = legs.get(key)
            if leg is not None:
                leg["roofline"] = hbm_roofline(pmc, key, leg["kernel_ms"], 1, n_local, args.obs_dtype, args.workload, h)
                leg["roofline"]["kernel"] = env.launch_plan(1)[1]
        return roof, valu_roofline(sq, "persistent" if tpl > 1 else "one_launch_per_turn", n_local, step_kernel_ms), main_form

    def distributed_block(self, summary, allr, identities):
        """N > 1: what the collective carried and where the ranks sat.  Every rank counts the winners of its own rows on its device (the pack kernel's counted
        form, on the rows the last timed gather sent: nothing has been played since); the sum over ranks -- one all-reduce of 4 integers, OUTSIDE the timed
        region -- must equal what rank 0 counts in the gathered rows."""
        args, dist, torch, evg, env = self.args, self.dist, self.torch, self.evg, self.env
        env.packed_episode_results(counts=self.win_counts_dev)
        summed = self.win_counts_dev if args.backend == "nccl" else self.win_counts_dev.cpu()
        dist.all_reduce(summed)
        summed = [int(x) for x in summed.tolist()]
        if self.rank != 0:
            return None
        med = summary["median_indices"]
        per_rank = [dict(identities[r], rank=r, seconds=mean_over(med, allr[r, :, 0].tolist()),
                         env_steps_per_s=self.n_local * args.steps / mean_over(med, allr[r, :, 0].tolist()),
                         kernel_ms_per_step=mean_over(med, allr[r, :, 1].tolist()), collective_us=mean_over(med, allr[r, :, 2].tolist())) for r in range(self.world)]
        gw = list(evg.ResultGather.win_counts(self.gathered))
        if gw != summed:
            raise SystemExit("the gathered rows (wins %s) are not what the ranks hold (sum of their own counts %s): the collective did not carry the results" % (gw, summed))
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception as ex:                       # reporting only
            ver = "unavailable (%s)" % type(ex).__name__
        g, native = self.gather, self.native
        rows_per_rank = g.rows_per_rank(self.gathered) if native is None else [
            int((self.gathered[a:a + c, 2] >= 0).sum()) for a, c in zip([sum(native.counts[:r]) for r in range(self.world)], native.counts)]
        steps_us = max(p["kernel_ms_per_step"] for p in per_rank) * args.steps * 1e3
        coll_us = max(p["collective_us"] for p in per_rank)
        rehearsal = self.world == 1 or args.backend != "nccl"
        d = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
             "rccl_ranks_seen": dist.get_world_size() if args.backend == "nccl" else None,
             "evg_comm_ranks": native.world if (native is not None and hasattr(native, "world")) else None,
             "collective": "one pack kernel + ONE torch.distributed.%s of [n,4] f32 episode results to rank 0 (everglades_amd.ResultGather), inside the timed region; "
                           "the win-count self-check (all_reduce of 4 integers) runs after it" % g.collective,
             "collective_us": coll_us,
             "collective_us_is": "stream time from the end of the last step launch to the end of the gather (pack kernel + gather), slowest rank, median region",
             "step_launches_us": steps_us,
             "closing_bracket": "completion of the gather (rank 0 receives every rank's rows: it cannot end before the slowest rank's steps) + "
                                "torch.cuda.synchronize(); per-rank times exchanged afterwards, max over ranks",
             "step_share_of_region": steps_us / (steps_us + coll_us), "expected": expected_if_wire_free(self.world, args.steps),
             "collective_calls": g.calls if native is None else summary["timing"]["repeats"] + 1,
             "collective_api": "torch.distributed" if native is None else "evg_gather_returns (RCCL through the C-ABI)",
             "gathered_rows": int(self.gathered.shape[0]), "gathered_rows_with_a_finished_episode_per_rank": rows_per_rank, "rows_expected_per_rank": g.counts,
             "gathered_wins_equal_sum_of_per_rank_counts": True, "wins_p0_p1_tie_unfinished_sum_over_ranks": summed, "per_rank": per_rank}
        d.update(check_distinct_devices(identities, args.backend, rehearsal))
        return d

    def report(self, summary, cw, sustained, legs, dist_block):
        """rank 0: the full result object"""
        args, env, evg = self.args, self.env, self.evg
        dt, steps = summary["seconds"], args.steps
        tpl = args.turns_per_launch
        step_kernel_ms = summary["kernel_ms_sum"] / steps
        ms_per_step = dt / steps * 1e3
        roof, valu, main_form = self.price_roofline(step_kernel_ms, ms_per_step, legs)
        st = self.stats
        timing = dict(summary["timing"], clock_warmup=cw)
        cold = (cw.get("cold_region") or {}).get("value")
        out = {
            "metric": "env-steps/sec at 65536 concurrent DemoMap games, 1/2/4/8 MI355X",
            "value": self.total * steps / dt, "unit": "env-steps/s", "n_gpus": self.world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            # what rounds 1-4 reported as `value`: ONE K-step region timed before the clock warm-up, same brackets (None: --clock-warmup-ms 0 or N > 1)
            "value_cold": cold,
            "value_protocol": ("median of %d regions of exactly K steps after %g ms of untimed clock warm-up on a scratch handle (round 5 on); value_cold = one "
                               "region before it (the protocol of rounds 1-4); sustained = %d x 150-turn launches in one region"
                               % (timing["repeats"], cw["ms"], args.sustained_launches)),
            "sustained": sustained,
            "config": dict({"workload": ("%d concurrent DemoMap games per GPU, random_actions vs random_actions drawn on device (fused into the step kernel, orders "
                                         "written to an [N,2,7,2] tensor; persistent rollout form, see turns_per_launch), auto-reset, obs %s [N,2,105]"
                                         if args.workload == "random" else
                                         "%d concurrent DemoMap games per GPU, on-device Cycle_BRush_Turn25 vs SwarmAgent (BASELINE config 5; both bots fused into the "
                                         "step kernel, orders written out; episodes end by BaseCapture after 84-94 turns), auto-reset, obs %s [N,2,105]")
                                        % (self.n_local, args.obs_dtype),
                            "window": "desynchronised steady state: 150-turn pre-roll restarts env e at pre-roll turn hash(e) mod 150 (episode phases uniform over "
                                      "0..149, unrelated between neighbouring envs), then 150 settle turns, --warmup turns and the K timed turns",
                            "envs_per_gpu": self.n_local, "total_envs": self.total, "turns_per_launch": tpl,
                            "turns_of_last_timed_launch": steps % tpl or min(tpl, steps), "launch_form": main_form},
                           **legs,
                           **{"parallelism": "env-sharded x%d" % self.world, "kernel_source_hash": kernel_source_hash(),
                              "episodes_finished_rank0": int(st["totals"][0]), "wins_p0_p1_tie_rank0": [int(x) for x in st["totals"][1:]],
                              "gathered_wins_all_ranks": list(evg.ResultGather.win_counts(self.gathered)) if self.gathered is not None else None}),
            "roofline": roof,
            "timing": timing,
        }
        if valu:
            out["roofline_valu_issue"] = valu
        if dist_block:
            out["distributed"] = dist_block
        if self.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.seed)
            if self.final_state is not None:
                out["cpu_baseline"]["same_games_as_gpu"] = cpu_parity(args.seed, self.n_local, self.played_at_snapshot - PHASES, st, self.final_state)
        return out


def main(argv=None):
    args = parse_args(argv)
    # multi-process GPU work on this image needs dmabuf IPC (the host driver supports nothing else): RCCL fails with
    # "hipIpcGetMemHandle: invalid argument" without it.  The boxes export it already; make sure the ranks see it either way.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started directly instead of through torch.distributed.run: launch the ranks as child processes (nothing has
        # touched the GPU yet) and leave with their exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    run = Run(args)
    run.setup()
    cw = run.warm_the_clock()
    region_s, kernel_ms, collective_ms, allr = run.timed_regions()
    summary = summarise_regions(region_s, kernel_ms, collective_ms, args.steps, run.total)
    sustained = run.sustained_leg()
    if run.scratch is not None:
        run.scratch.close()
        run.scratch = None
    run.snapshot_games()
    dist_block = None
    if run.dist_on:
        ident = device_identity(run.torch, run.dev_index, run.local_rank)
        identities = [None] * run.world
        run.dist.all_gather_object(identities, ident)
        dist_block = run.distributed_block(summary, allr, identities)
    legs = run.extra_legs()
    if run.rank == 0:
        out = run.report(summary, cw, sustained, legs, dist_block)
        if args.details:
            with open(args.details, "w") as f:
                f.write(json.dumps(out) + "\n")
        print(json.dumps(compact_line(out), separators=(",", ":")), flush=True)
    run.env.close()
    if run.dist_on:
        run.dist.barrier()
        run.dist.destroy_process_group()


if __name__ == "__main__":
    main()
