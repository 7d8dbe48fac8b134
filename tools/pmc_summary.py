#!/usr/bin/env python3
"""Turns the output of tools/profile.sh (gpurun_out/prof_<tag>/) into the two files committed under profiles/:
  <name>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats table, evg kernels only
  <name>_pmc_traffic.json   HBM bytes per launch / per turn of the step kernel from the FETCH_SIZE and WRITE_SIZE passes,
                            corrected with the calibration copy of the same passes (tools/pmc_calib.py)
usage: python tools/pmc_summary.py <tag> <name>      e.g.  r01i r01_i"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
CMD = "python3 bench.py --steps 150 --warmup 150 --no-cpu-baseline"


def counters(sub):
    f = glob.glob(os.path.join(P, sub, "*", "*_counter_collection.csv"))[0]
    rows = {}
    for r in csv.DictReader(open(f)):
        rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return rows


# kernel stats
src = glob.glob(os.path.join(P, "stats", "*", "*_kernel_stats.csv"))[0]
lines = open(src).read().splitlines()
keep = [lines[0]] + [l for l in lines[1:] if "evg::" in l]
with open(os.path.join(ROOT, "profiles", name + "_kernel_stats.csv"), "w") as f:
    f.write('"# rocprofv3 --kernel-trace --stats --output-format csv -- %s (MI355X, 65536 envs). evg_step_kernel<float,64,true,false> = '
            'persistent form, one launch = 150 turns (warm-up + timed); <float,64,false,false> = the 150 one-launch-per-turn reference leg. '
            'evg kernels only."\n' % CMD)
    f.write("\n".join(keep) + "\n")

fe, wr = counters("fetch"), counters("write")
cf, cw = counters("calib_fetch"), counters("calib_write")
calib_f = max(max(v) for v in cf.values())          # the 256 MiB copy dominates
calib_w = max(max(v) for v in cw.values())
fetch_scale = 262144.0 / calib_f                     # expected KB / reported
write_scale = 262144.0 / calib_w
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- " + CMD,
       "envs": 65536,
       "calibration": {"copy_256MiB_FETCH_SIZE_KB": calib_f, "copy_256MiB_WRITE_SIZE_KB": calib_w, "expected_KB": 262144,
                       "fetch_scale": round(fetch_scale, 4), "write_scale": round(write_scale, 4),
                       "conclusion": "FETCH_SIZE reports 1/2 of the bytes fetched, WRITE_SIZE is exact (tools/pmc_calib.py in the same passes)"},
       "forms": {}}
for form, key in (("persistent", "true, false>"), ("one_launch_per_turn", "false, false>")):
    kf = [k for k in fe if "evg_step_kernel<float, 64, " + key in k][0]
    f_kb, w_kb = fe[kf], wr[kf]
    tpl = 150 if form == "persistent" else 1
    per_launch = (sum(f_kb) / len(f_kb) * round(fetch_scale) + sum(w_kb) / len(w_kb) * round(write_scale)) * 1024.0
    out["forms"][form] = {"kernel": kf, "launches": len(f_kb), "turns_per_launch": tpl, "FETCH_SIZE_KB_mean": sum(f_kb) / len(f_kb),
                          "WRITE_SIZE_KB_mean": sum(w_kb) / len(w_kb), "corrected_bytes_per_launch": per_launch,
                          "corrected_bytes_per_turn": per_launch / tpl, "bytes_per_env_step": per_launch / tpl / 65536}
out["corrected_bytes_per_launch"] = out["forms"]["persistent"]["corrected_bytes_per_launch"]
out["turns_per_launch"] = 150
out["corrected_bytes_per_turn"] = out["forms"]["persistent"]["corrected_bytes_per_turn"]
json.dump(out, open(os.path.join(ROOT, "profiles", name + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["forms"], indent=1))
print(open(os.path.join(ROOT, "profiles", name + "_kernel_stats.csv")).read())
