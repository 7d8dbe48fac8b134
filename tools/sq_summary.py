#!/usr/bin/env python3
"""Summarises the SQ counter passes of tools/profile_sq.sh (gpurun_out/sq_<tag>/<form>_p*/) into
profiles/<name>_sq_counters.json: per launch form of the step kernel the mean counter values per dispatch of the timed
window, and per wave-turn.   usage: python tools/sq_summary.py <tag> <name> [envs] [workload] [obs_dtype]"""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _prof import ROOT, FORMS, FORM_KEY, kernel_source_hash, counter_rows, timed_window
tag, name = sys.argv[1], sys.argv[2]
ENVS = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else "random"
OBS = sys.argv[5] if len(sys.argv) > 5 else "float32"
import _prof
_prof.OBS_DTYPE = OBS
P = os.path.join(ROOT, "gpurun_out", "sq_" + tag)
out = {"kernel_source_hash": kernel_source_hash(), "envs": ENVS, "workload": WORKLOAD, "obs_dtype": OBS,
       "commands": {f: open(os.path.join(P, "cmd_%s.txt" % f)).read().strip() for f in FORMS if os.path.exists(os.path.join(P, "cmd_%s.txt" % f))},
       "note": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass>; dispatches of the timed window only.  SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_* and "
               "SQ_ACTIVE_INST_* tick in quad-cycles (MI355X_MICROARCH.md); instruction counters count wave-instructions.  per_wave_turn = per dispatch / "
               "waves / "
               "turns of the dispatch.", "kernels": {}}
for form, (tail, tpl, nwin) in FORMS.items():
    mean, meta = {}, None
    for d in sorted(x for x in glob.glob(os.path.join(P, form + "_p*")) if os.path.isdir(x)):
        rows, m = counter_rows(d, form)
        # the step kernel only (two-kernel forms list the action kernel too)
        keep = [i for i, x in enumerate(m) if not any(k in x["name"] for k in _prof.ACTION_KERNELS)]
        rows, m = [rows[i] for i in keep], [m[i] for i in keep]
        rows = rows[-FORMS[form][2]:]
        meta = m[-1]
        for c in rows[0]:
            mean[c] = sum(r[c] for r in rows) / len(rows)
        mean.setdefault("_ns", []).append(sum(x["ns"] for x in m[-len(rows):]) / len(rows))
    if not meta:
        continue
    ns = mean.pop("_ns")
    waves = meta["grid"] // 64
    pw = {c: v / waves / tpl for c, v in mean.items()}
    e = {"kernel": meta["name"], "waves": waves, "turns_per_launch": tpl, "vgpr": meta["vgpr"], "agpr": meta["agpr"], "sgpr": meta["sgpr"],
         "lds_bytes": meta["lds"],
         "scratch": meta["scratch"], "dispatch_ns_under_pmc": sum(ns) / len(ns), "per_wave_turn": pw}
    if "SQ_WAVE_CYCLES" in pw and "SQ_INSTS_VALU" in pw:
        e.update({"wave_cycles_per_wave_turn": 4 * pw["SQ_WAVE_CYCLES"], "valu_insts_per_wave_turn": pw["SQ_INSTS_VALU"],
                  "cycles_per_valu_inst": 4 * pw["SQ_WAVE_CYCLES"] / pw["SQ_INSTS_VALU"]})
    if "SQ_ACTIVE_INST_VALU" in pw and "SQ_WAIT_ANY" in pw:
        tot = pw["SQ_WAIT_ANY"] + pw["SQ_WAIT_INST_ANY"] + pw["SQ_ACTIVE_INST_ANY"]
        e["shares_of_wave_cycles"] = {"wait_any (s_waitcnt)": pw["SQ_WAIT_ANY"] / tot, "wait_inst_any (issue stall)": pw["SQ_WAIT_INST_ANY"] / tot,
                                      "active_inst_any": pw["SQ_ACTIVE_INST_ANY"] / tot, "active_valu": pw["SQ_ACTIVE_INST_VALU"] / tot,
                                      "active_lds": pw["SQ_ACTIVE_INST_LDS"] / tot, "active_scalar": pw["SQ_ACTIVE_INST_SCA"] / tot,
                                      "wait_inst_lds": pw["SQ_WAIT_INST_LDS"] / tot}
    out["kernels"][FORM_KEY[form]] = e
json.dump(out, open(os.path.join(ROOT, "profiles", name + "_sq_counters.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, e["kernel"], "waves", e["waves"], "vgpr", e["vgpr"], "+", e["agpr"], "lds", e["lds_bytes"], "ns", e["dispatch_ns_under_pmc"])
    for c, v in sorted(e["per_wave_turn"].items()):
        print("   %-28s %12.1f per wave-turn" % (c, v))
    print("   ", {k2: v for k2, v in e.items() if k2 not in ("per_wave_turn", "kernel")})
