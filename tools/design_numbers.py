#!/usr/bin/env python3
"""Prints the "Numbers" list of DESIGN.md section 6 from the bench lines and counter summaries committed under profiles/ (one source per number).
usage: python tools/design_numbers.py [name]      (default r06_e)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = sys.argv[1] if len(sys.argv) > 1 else "r06_e"
F = lambda n: json.load(open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (NAME, n))))
fin, drv, c5, k4, i16, rc, gl = F("final"), F("driver_shape"), F("config5"), F("4096envs"), F("int16"), F("rehearse_rccl_1rank"), F("rehearse_gloo_2ranks_one_gpu")
pm = json.load(open(os.path.join(ROOT, "profiles", NAME + "_pmc_traffic.json")))["forms"]
G = lambda d: d["value"] / 1e9
def leg(d, k):
    l = d["config"][k]
    return l["env_steps_per_s"] / 1e9, l["ms_per_step"] * 1e3, l["kernel_ms"] * 1e3, (l.get("roofline") or {}).get("frac")
rows = []
rows.append("* env-steps/s, persistent rollout form, default 450 timed turns (median of %d regions) — **%.2f G** (%.2f µs/turn; regions %.2f–%.2f G) "
            "[`%s_bench_final.json`]"
            % (fin["timing"]["repeats"], G(fin), fin["ms_per_step"] * 1e3, fin["timing"]["min_value"] / 1e9, fin["timing"]["max_value"] / 1e9, NAME))
rows.append("* the driver's shape `--steps 20 --warmup 5` (median of %d regions) — **%.2f G** (%.2f µs/step, kernel %.2f µs/turn; regions %.2f–%.2f G) "
            "[`%s_bench_driver_shape.json`]"
            % (drv["timing"]["repeats"], G(drv), drv["ms_per_step"] * 1e3, drv["roofline"]["kernel_ms"] * 1e3, drv["timing"]["min_value"] / 1e9,
               drv["timing"]["max_value"] / 1e9, NAME))
# the default command under rocprofv3, with the shader clock of every dispatch
import re
dr = open(os.path.join(ROOT, "profiles", NAME + "_default_run_kernel_stats.csv")).read().splitlines()
tl = next(l for l in dr if "timed 150-turn launches" in l)
m = re.search(r"median (\d+) us = ([\d.]+) us per turn.*roofline.kernel_ms = ([\d.]+)", tl)
clk = [(float(a), float(b)) for a, b in re.findall(r"(\d+)@([\d.]+)", next(l for l in dr if "GRBM_GUI_ACTIVE" in l))]
slow, fast = max(clk), min(clk)
rows.append("* the default command under rocprofv3 — median timed region %s µs per turn (the line of that run: %.2f); a 150-turn dispatch lasts %.2f ms at "
            "%.2f GHz and %.2f ms at %.2f GHz "
            "(GRBM_GUI_ACTIVE): %.2f and %.2f M shader cycles — the time follows the clock [`%s_default_run_kernel_stats.csv`]"
            % (m.group(2), float(m.group(3)) * 1e3, slow[0] / 1e3, slow[1], fast[0] / 1e3, fast[1], slow[0] * slow[1] / 1e3, fast[0] * fast[1] / 1e3, NAME))
bx6 = os.path.join(ROOT, "profiles", NAME + "_boxes.txt")
if os.path.exists(bx6):
    mm = re.search(r"default value over the five boxes: ([\d.]+)-([\d.]+) G = ([\d.]+) %; of the driver-shape value: ([\d.]+)-([\d.]+) G = ([\d.]+) %", open(bx6).read())
    rows.append("* the same two commands on five boxes (separate gpurun calls) — default %s–%s G (%s %%: the boxes' sustained shader clocks differ, 2.30–2.38 GHz; "
                "the kernel's cycles do not), driver shape %s–%s G (%s %%) [`%s_boxes.txt`]" % (mm.group(1), mm.group(2), mm.group(3), mm.group(4), mm.group(5), mm.group(6), NAME))
bx = os.path.join(ROOT, "profiles", NAME + "_driver_shape_boxes.txt")
if os.path.exists(bx):
    mm = re.search(r"Spread of the value: ([\d.]+)-([\d.]+) G = ([\d.]+) %", open(bx).read())
    rows.append("* the driver's command in four separate gpurun calls (a fresh box each) — %.2f–%.2f G: %s %% [`%s_driver_shape_boxes.txt`]"
                % (float(mm.group(1)), float(mm.group(2)), mm.group(3), NAME))
cold = lambda d: (d.get("value_cold") or d["timing"]["clock_warmup"]["cold_region"]["value"]) / 1e9
rows.append("* `value_cold`, ONE K-step region timed before the clock warm-up and the W warm-up steps in the same run (what rounds 1–4 reported as "
            "the value) — driver shape %.2f G against "
            "%.2f G, default %.2f G against %.2f G [`%s_bench_driver_shape.json`, `%s_bench_final.json`]" % (cold(drv), G(drv), cold(fin), G(fin), NAME, NAME))
if drv.get("sustained"):
    sd, sf = drv["sustained"], fin["sustained"]
    rows.append("* `sustained`, %d launches of 150 turns in ONE region (%.2f ms) right after the K-step regions of the same run — driver shape **%.2f G** (%.2f "
                "µs/step, kernel %.2f µs/turn) beside its 20-step value of %.2f G; default run %.2f G (%.2f µs/step) [`%s_bench_driver_shape.json`, "
                "`%s_bench_final.json`]" % (sd["launches"], sd["region_ms"], sd["value"] / 1e9, sd["ms_per_step"] * 1e3, sd["kernel_ms"] * 1e3, G(drv),
                                            sf["value"] / 1e9, sf["ms_per_step"] * 1e3, NAME, NAME))
cwf = os.path.join(ROOT, "profiles", NAME + "_clock_warmup_ab.txt")
if os.path.exists(cwf):
    cw = [l for l in open(cwf).read().splitlines() if not l.startswith("#")]
    gv = lambda l: float(re.search(r"([\d.]+) G env-steps/s", l).group(1))
    rows.append("* without the clock warm-up (`--clock-warmup-ms 0`, what earlier rounds measured), same box, back to back — driver shape %.2f G against %.2f "
                "G, default %.2f G against %.2f G [`%s_clock_warmup_ab.txt`]" % (gv(cw[0]), gv(cw[1]), gv(cw[2]), gv(cw[3]), NAME))
for k, nm, both in (("one_launch_per_turn", "one launch per turn, orders drawn in the kernel", "kernel alone"),
                    ("caller_actions_per_turn", "one launch per turn, orders from a caller tensor (the Gym consumer)", "both kernels of the turn"),
                    ("learner_vs_bot_per_turn", "learner seat vs on-device bot (`evg_step_vs_policy`, stand-in policy)", "both kernels of the turn")):
    g, ms, km, fr = leg(fin, k)
    rows.append("* %s — %.2f G; %.1f µs wall, %.1f µs stream per turn; %s %.1f µs (rocprofv3); fabric frac %.2f [`%s_bench_final.json`, "
                "`%s_kernel_stats.csv`]" % (nm, g, ms, km, both, pm[k]["kernel_us_per_turn"], fr, NAME, NAME))
s = fin["config"]["learner_smart_actions_vs_bot_per_turn"]
rows.append("* the same with `evg_smart_actions` as the learner's decode; + `evg_smart_state_compact` in front — %.2f G, %.1f µs stream per turn; %.2f G, "
            "%.1f µs [`%s_bench_final.json`]"
            % (s["env_steps_per_s"] / 1e9, s["kernel_ms"] * 1e3, s["with_features"]["env_steps_per_s"] / 1e9, s["with_features"]["kernel_ms"] * 1e3, NAME))
if s.get("with_features_fused"):
    rows.append("* ... with the features written by the step launch itself (`evg_step_vs_policy_smart`) instead of the feature kernel — %.2f G, %.1f µs stream "
                "per turn [`%s_bench_final.json`]" % (s["with_features_fused"]["env_steps_per_s"] / 1e9, s["with_features_fused"]["kernel_ms"] * 1e3, NAME))
if s.get("with_epsilon"):
    rows.append("* the TRAINING turn: `evg_smart_get_action` (DQNAgent.get_action, epsilon %g: coin + get_random_actions on the device) instead of "
                "`evg_smart_actions` — %.2f G, %.1f µs stream per turn [`%s_bench_final.json`]"
                % (s["with_epsilon"]["epsilon"], s["with_epsilon"]["env_steps_per_s"] / 1e9, s["with_epsilon"]["kernel_ms"] * 1e3, NAME))
p = fin["config"]["pipelined_halves_per_turn"]
rows.append("* two half-batch handles free-running (`PipelinedVecEnv`) — %.2f G; %.1f µs wall, %.1f µs stream per turn of the whole batch (learner-seat "
            "turn: %.1f µs) [`%s_bench_final.json`]"
            % (p["env_steps_per_s"] / 1e9, p["ms_per_step"] * 1e3, p["kernel_ms"] * 1e3, p["learner_vs_bot"]["kernel_ms"] * 1e3, NAME))
n, o = fin["config"]["without_observations"], fin["config"]["obs_float64"]
rows.append("* persistent without observations / with float64 observations / int16 — %.2f G (%.1f µs) / %.2f G (%.1f µs) / %.2f G (%.1f µs) "
            "[`%s_bench_final.json`, `%s_bench_int16.json`]"
            % (n["env_steps_per_s"] / 1e9, n["ms_per_step"] * 1e3, o["env_steps_per_s"] / 1e9, o["ms_per_step"] * 1e3, G(i16), i16["ms_per_step"] * 1e3, NAME,
               NAME))
r = fin["roofline"]
rows.append("* roofline (`bound` hbm, `bound_detail` fabric): bytes per env-step, rate, fraction of 8 TB/s — persistent %.0f B → %.2f TB/s = **%.2f** at "
            "the `ms_per_step` of the region `value` comes from (`frac_kernel_events` %.2f at the launches' own HIP-event time; "
            "%.2f at rocprofv3's %.2f µs); one launch per turn %.0f B → **%.2f**; "
            "caller orders %.0f B → **%.2f**; learner seat %.0f B → **%.2f**; driver shape **%.2f** (`frac_kernel_events` %.2f) [`%s_pmc_traffic.json`, bench lines]"
            % (r["bytes_per_env_step"], r["achieved"] / 1e3, r["frac"], r.get("frac_kernel_events", r["frac"]), pm["persistent"]["frac_of_8TBps"],
               pm["persistent"]["kernel_us_per_turn"],
               pm["one_launch_per_turn"]["bytes_per_env_step"], pm["one_launch_per_turn"]["frac_of_8TBps"],
               pm["caller_actions_per_turn"]["bytes_per_env_step"], pm["caller_actions_per_turn"]["frac_of_8TBps"],
               pm["learner_vs_bot_per_turn"]["bytes_per_env_step"], pm["learner_vs_bot_per_turn"]["frac_of_8TBps"], drv["roofline"]["frac"],
               drv["roofline"].get("frac_kernel_events", drv["roofline"]["frac"]), NAME))
if r.get("algorithmic_bytes_per_env_step"):
    rows.append("* algorithmic bytes of this design / counter bytes — %.0f B (971 B of outputs + 2 × %.0f B of health rows) / %.0f B: "
                "`traffic_over_algorithmic` %.2f [`%s_bench_final.json`]"
                % (r["algorithmic_bytes_per_env_step"], (r["algorithmic_bytes_per_env_step"] - 971) / 2, r["bytes_per_env_step"],
                   r["traffic_over_algorithmic"], NAME))
b = r["beyond_mall"]
dg, pc, wr = b["diag_library_chunked_over_262144_envs"], b["product_library_chunked_over_131071_envs_cache_mib_1024"], b["whole_rounds_one_after_the_other"]["persistent"]
rows.append("* **`hbm_proper_frac`**: the persistent kernel cycled through %d MB (diagnostic library, 262 144 envs) — %.0f B, %.3f ns per env-step (%.3f at "
            "65 536), %.2f TB/s = **%.2f of the HBM peak** [`%s`]"
            % (dg["working_set_MB"], dg["bytes_per_env_step"], dg["ns_per_env_step"], dg["ns_per_env_step_at_65536"], dg["traffic_TBps"], dg["frac_of_8TBps"],
               os.path.basename(dg["source"])))
rows.append("* the product library cycled through %d MB (131 071 envs, `cache_mib` 1 024) / the product's whole rounds at 262 144 envs — %.2f (%.3f ns per "
            "env-step: 1.35 × the cache is still mostly cache-resident) / %.2f (%.3f ns) [`%s`, `%s`]"
            % (pc["working_set_MB"], pc["frac_of_8TBps"], pc["ns_per_env_step"], wr["frac_of_8TBps"], wr["ns_per_env_step"], os.path.basename(pc["source"]),
               os.path.basename(wr["source"])))
rows.append("* `survey_8d_frac` (4 530 B per env-step at the kernel time) — %.2f (default), %.2f (driver shape): above 1, model not applicable [bench lines]"
            % (r["survey_8d_frac"], drv["roofline"]["survey_8d_frac"]))
v = fin["roofline_valu_issue"]
rows.append("* `roofline_valu_issue` — %.0f VALU instructions per wave-turn → **%.2f** of one wave64 instruction per 4 cycles per SIMD at 2.4 GHz [`%s`]"
            % (v["valu_insts_per_wave_turn"], v["frac"], os.path.basename(v["source"])))
cb = fin["cpu_baseline"]
rows.append("* cpu_baseline (C oracle, OpenMP, %d cores of the GPU box, f64 observations) — %.1f M env-steps/s; the same 65 536 games × %d turns replayed: "
            "episode results, win counts %s and final state incl. float64 health equal: %s [`%s_bench_final.json`, `kind: \"port\"`]"
            % (cb["cores"], cb["value"] / 1e6, cb["same_games_as_gpu"]["turns"], cb["same_games_as_gpu"]["gpu_wins_p0_p1_tie"],
               cb["same_games_as_gpu"]["equal"], NAME))
rows.append("* Python reference, 1 core / 8 cores (build container) — 529–554 / 3 208 env-steps/s [BASELINE.md]")
rows.append("* BASELINE config 5 (Cycle_BRush_Turn25 vs SwarmAgent fused, 65 536 envs) — **%.2f G** (%.1f µs/turn), %.0f B → frac %.2f "
            "[`%s_bench_config5.json`, `%s_config5_pmc_traffic.json`]" % (G(c5), c5["ms_per_step"] * 1e3, c5["roofline"]["bytes_per_env_step"],
                                                                          c5["roofline"]["frac"], NAME, NAME))
rows.append("* BASELINE config 2 (4 096 envs; four-lane kernel; working set in L2 / Infinity Cache: latency-bound, `frac` meaningless) — **%.0f M** (%.1f "
            "µs/turn) [`%s_bench_4096envs.json`]" % (G(k4) * 1e3, k4["ms_per_step"] * 1e3, NAME))
d = rc["distributed"]
rows.append("* N > 1 code path, RCCL with a one-rank group, `--steps 20` — %.2f G; step launches %.0f µs + collective path (pack + gather) **%.1f µs** → "
            "`step_share_of_region` %.3f [`%s_bench_rehearse_rccl_1rank.json`]"
            % (G(rc), d["step_launches_us"], d["collective_us"], d["step_share_of_region"], NAME))
rows.append("* N > 1 code path, two ranks over gloo sharing the one GPU — %.2f G for 131 072 envs (host-copy collective %.0f µs): plumbing only "
            "[`%s_bench_rehearse_gloo_2ranks_one_gpu.json`]" % (G(gl), gl["distributed"]["collective_us"], NAME))
print("\n".join(rows))
