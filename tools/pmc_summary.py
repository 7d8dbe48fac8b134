#!/usr/bin/env python3
"""Turns the output of tools/profile.sh (gpurun_out/prof_<tag>/) into the two files committed under profiles/:
  <name>_kernel_stats.csv   the rocprofv3 --kernel-trace --stats tables of both launch forms (evg kernels only) plus, per form,
                            the mean duration of the step-kernel dispatches of the timed window (the last 150 turns)
  <name>_pmc_traffic.json   HBM-side bytes per launch / per env-step of the step kernel from the FETCH_SIZE and WRITE_SIZE
                            passes, corrected with the calibration copy of the same passes (tools/pmc_calib.py), keyed by the
                            hash of the kernel sources so that bench.py only uses figures of the build it runs
usage: python tools/pmc_summary.py <tag> <name> [envs] [workload] [obs_dtype] [variant]      e.g.  r02b r02_b"""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _prof import ROOT, FORMS, FORM_KEY, kernel_source_hash, counter_rows, timed_window, trace_durations
import csv
tag, name = sys.argv[1], sys.argv[2]
ENVS = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else "random"
OBS = sys.argv[5] if len(sys.argv) > 5 else "float32"
VARIANT = sys.argv[6] if len(sys.argv) > 6 else None      # e.g. "cycled": the chunked form forced over the whole batch (bench.py --diag-lanes 2)
import _prof
_prof.OBS_DTYPE = OBS
_prof.KERNELS_PER_TURN = 2 if WORKLOAD == "random" else 3
FORMS = {f: v for f, v in FORMS.items() if os.path.exists(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "cmd_%s.txt" % f))}     # the forms this run profiled
P = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
STATE_ROUND_TRIP = 2 * (24 * 4 + 6 * 4 + 6 * 4 + 4 + 4 + 2 * 4)      # words a launch reads at its start and writes at its end, per env (evg_device.h)


def calib(sub):
    f = glob.glob(os.path.join(P, sub, "*", "*_counter_collection.csv"))[0]
    return max(float(r["Counter_Value"]) for r in csv.DictReader(open(f)))     # the 256 MiB copy dominates


calib_f, calib_w = calib("calib_fetch"), calib("calib_write")
fetch_scale, write_scale = 262144.0 / calib_f, 262144.0 / calib_w            # expected KB / reported
out = {"kernel_source_hash": kernel_source_hash(), "envs": ENVS, "workload": WORKLOAD, "obs_dtype": OBS, "variant": VARIANT,
       "commands": {f: open(os.path.join(P, "cmd_%s.txt" % f)).read().strip().replace(ROOT + "/", "").replace(os.environ.get("GRAFT_REPO_ROOT", "\0") + "/",
                                                                                                              "") for f in FORMS},
       "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE; bytes = FETCH_SIZE KB x fetch_scale + WRITE_SIZE KB x "
                 "write_scale "
                 "(the gfx950 correction of MI355X_MICROARCH.md, re-derived by the calibration copy in the same passes); dispatches of the timed window only "
                 "(desynchronised steady state, bench.py docstring)",
       "calibration": {"copy_256MiB_FETCH_SIZE_KB": calib_f, "copy_256MiB_WRITE_SIZE_KB": calib_w, "expected_KB": 262144,
                       "fetch_scale": round(fetch_scale, 4), "write_scale": round(write_scale, 4)},
       "forms": {}}
stats_lines = []
for form, (tail, tpl, nwin) in FORMS.items():
    fr, fm = counter_rows(os.path.join(P, form + "_fetch"), form)
    wr, wm = counter_rows(os.path.join(P, form + "_write"), form)
    f_kb = [r["FETCH_SIZE"] for r in timed_window(fr, form)]
    w_kb = [r["WRITE_SIZE"] for r in timed_window(wr, form)]
    dur = timed_window(trace_durations(os.path.join(P, form + "_stats"), form), form)
    # dispatches per launch unit (caller / learner forms: action kernel(s) + step kernel = one turn)
    kpt = _prof.kernels_per_turn(form)
    per_launch = (sum(f_kb) / len(f_kb) * round(fetch_scale) + sum(w_kb) / len(w_kb) * round(write_scale)) * 1024.0 * kpt
    mean_ns = sum(dur) / len(dur) * kpt
    bpe = per_launch / tpl / ENVS
    rt = STATE_ROUND_TRIP if tpl > 1 else 0          # a single-turn launch's figure already contains its round trip
    out["forms"][FORM_KEY[form]] = {"kernel": fm[-1]["name"], "launches_in_window": len(f_kb) // kpt, "turns_per_launch": tpl,
                                    "FETCH_SIZE_KB_mean": sum(f_kb) / len(f_kb), "WRITE_SIZE_KB_mean": sum(w_kb) / len(w_kb),
                                    "corrected_bytes_per_launch": per_launch, "corrected_bytes_per_turn": per_launch / tpl, "bytes_per_env_step": bpe,
                                    "state_round_trip_bytes_per_env": rt, "bytes_per_env_step_steady": bpe - rt / tpl,
                                    "kernel_ns_mean_timed_window": mean_ns, "kernel_us_per_turn": mean_ns / tpl / 1e3,
                                    "traffic_TBps": per_launch / mean_ns / 1e3, "frac_of_8TBps": per_launch / mean_ns / 1e3 / 8.0,
                                    "kernels_per_turn": kpt,
                                    "vgpr_allocated": 2 * fm[-1]["vgpr"], "vgpr_rocprofv3_column": fm[-1]["vgpr"],
                                    "vgpr_note": "rocprofv3's VGPR_Count column counts in units of two registers on gfx950: the value printed is half of the "
                                                 "allocation, which is the "
                                                 "compiler's count (make resource-usage) rounded up to the allocation granule of 8",
                                    "agpr": fm[-1]["agpr"], "sgpr": fm[-1]["sgpr"], "lds_bytes": fm[-1]["lds"], "scratch": fm[-1]["scratch"]}
    src = glob.glob(os.path.join(P, form + "_stats", "*", "*_kernel_stats.csv"))[0]
    lines = open(src).read().splitlines()
    stats_lines.append('"# %s  --  rocprofv3 --kernel-trace --stats --output-format csv -- %s"' % (FORM_KEY[form], out["commands"][form]))
    stats_lines.append('"# step kernel, dispatches of the timed window only (last %d of %d): mean %.1f ns = %.2f us per turn; the --stats row below averages '
                       'ALL '
                       'dispatches of the run, incl. the 150 one-turn launches of the desynchronising pre-roll (synchronised early-episode positions)"'
                       % (len(dur), len(trace_durations(os.path.join(P, form + "_stats"), form)), mean_ns, mean_ns / tpl / 1e3))
    stats_lines += [lines[0]] + [l for l in lines[1:] if "evg::" in l]
open(os.path.join(ROOT, "profiles", name + "_kernel_stats.csv"), "w").write("\n".join(stats_lines) + "\n")
json.dump(out, open(os.path.join(ROOT, "profiles", name + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["forms"], indent=1))
print(open(os.path.join(ROOT, "profiles", name + "_kernel_stats.csv")).read())
