#!/usr/bin/env python3
"""Rewrites the bench table of README.md (between the markers <!-- bench-table --> and <!-- /bench-table -->) from the lines committed under profiles/.
usage: python tools/readme_table.py [name]      (default r06_e)"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = sys.argv[1] if len(sys.argv) > 1 else "r06_e"
F = lambda n: json.load(open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (NAME, n))))
fin, drv, c5, k4 = F("final"), F("driver_shape"), F("config5"), F("4096envs")
c, sm = fin["config"], fin["config"]["learner_smart_actions_vs_bot_per_turn"]
g = lambda l: "%.2f G" % (l["env_steps_per_s"] / 1e9)
us = lambda l, k="ms_per_step": "%.1f µs" % (l[k] * 1e3)
T = lambda d: d["timing"]
rows = ["  | form | env-steps/s on one MI355X | per turn |", "  |---|---|---|",
        "  | persistent rollout, default 450 timed turns (`value`) | **%.2f G** (regions %.2f–%.2f G; `value_cold` %.2f G; `sustained` %.2f G) | %.1f µs |"
        % (fin["value"] / 1e9, T(fin)["min_value"] / 1e9, T(fin)["max_value"] / 1e9, fin["value_cold"] / 1e9, fin["sustained"]["value"] / 1e9, fin["ms_per_step"] * 1e3),
        "  | the same, the driver's `--steps 20 --warmup 5` | **%.2f G** (regions %.2f–%.2f G; `value_cold` %.2f G; **`sustained` %.2f G**, %.1f ms region) | %.1f µs (%.1f µs) |"
        % (drv["value"] / 1e9, T(drv)["min_value"] / 1e9, T(drv)["max_value"] / 1e9, drv["value_cold"] / 1e9, drv["sustained"]["value"] / 1e9,
           drv["sustained"]["region_ms"], drv["ms_per_step"] * 1e3, drv["sustained"]["ms_per_step"] * 1e3),
        "  | one launch per turn, orders drawn in the kernel | %s | %s |" % (g(c["one_launch_per_turn"]), us(c["one_launch_per_turn"])),
        "  | one launch per turn, both seats' orders from a caller tensor (the Gym consumer) | %s | %s |" % (g(c["caller_actions_per_turn"]), us(c["caller_actions_per_turn"])),
        "  | learner seat vs on-device bot (`evg_step_vs_policy`; stand-in policy) | %s | %s |" % (g(c["learner_vs_bot_per_turn"]), us(c["learner_vs_bot_per_turn"])),
        "  | the same with `evg_smart_actions` as the learner's decode (+ compact features by their own kernel) | %s (%s) | %s (%s) |"
        % (g(sm), g(sm["with_features"]), us(sm, "kernel_ms"), us(sm["with_features"], "kernel_ms")),
        "  | ... with the features written by the step launch itself (`evg_step_vs_policy_smart`) | **%s** | %s |" % (g(sm["with_features_fused"]), us(sm["with_features_fused"], "kernel_ms")),
        "  | the TRAINING turn: `evg_smart_get_action`, epsilon %g | %s | %s |" % (sm["with_epsilon"]["epsilon"], g(sm["with_epsilon"]), us(sm["with_epsilon"], "kernel_ms")),
        "  | two half-batch handles free-running (`PipelinedVecEnv`) | %s | %s |" % (g(c["pipelined_halves_per_turn"]), us(c["pipelined_halves_per_turn"])),
        "  | persistent, no observations (the evaluation harness) | %s | %s |" % (g(c["without_observations"]), us(c["without_observations"]))]
r, rd = fin["roofline"], drv["roofline"]
para = ("  Fabric bytes (L2 ↔ Infinity Cache / HBM, PMC): %.0f B per env-step (this design's algorithmic bytes: %.0f) → **%.2f of 8 TB/s** at the `ms_per_step` of the "
        "region the value comes from (%.2f at the launches' own HIP-event time; %.2f with the algorithmic bytes; %.2f / %.2f in the driver's shape), with a round's "
        "working set cache-resident; the same kernel cycled through 723 MB: **%.2f of the HBM peak** (`roofline.hbm_proper_frac`); SURVEY §8(d)'s 4 530-byte model "
        "priced at the same time: %.2f (not applicable: the state stays on chip). VALU issue %.2f. 16-core C port %.1f M/s with the same 65 536 games replayed and "
        "compared (Python reference 529–554/s per core). BASELINE config 5 (scripted bots, fused) %.2f G; config 2 (4 096 envs) %.2f G."
        % (r["bytes_per_env_step"], r["algorithmic_bytes_per_env_step"], r["frac"], r["frac_kernel_events"], r.get("frac_algorithmic", 0.0), rd["frac"], rd["frac_kernel_events"],
           r["hbm_proper_frac"], r["survey_8d_frac"], fin["roofline_valu_issue"]["frac"], fin["cpu_baseline"]["value"] / 1e6, c5["value"] / 1e9, k4["value"] / 1e9))
block = "<!-- bench-table -->\n" + "\n".join(rows) + "\n\n" + para + "\n<!-- /bench-table -->"
p = os.path.join(ROOT, "README.md")
s = open(p).read()
assert "<!-- bench-table -->" in s and "<!-- /bench-table -->" in s
s = re.sub(r"<!-- bench-table -->.*?<!-- /bench-table -->", lambda m: block, s, flags=re.S)
open(p, "w").write(s)
print(block)
