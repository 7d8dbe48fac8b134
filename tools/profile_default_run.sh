#!/bin/bash
# Run on the GPU box: the DEFAULT bench command (python3 bench.py: 450 timed turns = three 150-turn launches per region, nine regions, value = the median region)
# under rocprofv3 -- once with --kernel-trace --stats, once with --kernel-trace --pmc GRBM_GUI_ACTIVE (effective shader clock of a dispatch = GRBM_GUI_ACTIVE /
# 8 XCDs / duration, MI355X_MICROARCH.md) --, so that the kernel time of the headline line can be read against the profiler's durations of the SAME dispatches.
# tools/profile.sh's passes time ONE 150-turn launch 150 turns after the start of the process: earlier in the clock ramp of a run (see the output).
#   -> profiles/<name>_default_run_kernel_stats.csv
set -o pipefail
NAME=${1:-r05_d}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${NAME}_default
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clock -- $CMD > $OUT/bench_clock.json 2> $OUT/err_clock.txt || { tail -5 $OUT/err_clock.txt; exit 1; }
cd $R
python3 - <<PY
import csv, glob, json
KERNEL = "evg_step_kernel<float, 64, true, false, false, false"
f = glob.glob("$OUT/stats/*/*_kernel_trace.csv")[0]
rows = sorted((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"])
durs = [d for _, d in rows]
line = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
R, K = int(line["timing"]["repeats"]), int(line["steps"])
L = K // 150                                                        # launches per region
S = int((line.get("sustained") or {}).get("launches", 0))           # the sustained leg: S more 150-turn launches in one region, right after the timed ones
sust = durs[len(durs) - S:] if S else []
timed = durs[len(durs) - S - R * L:len(durs) - S]
regions = [sum(timed[i * L:(i + 1) * L]) for i in range(R)]
med = sorted(regions)[(R - 1) // 2]
clk = []
f = glob.glob("$OUT/clock/*/*_counter_collection.csv")[0]
for r in sorted((r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"])):
    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    clk.append((ns, float(r["Counter_Value"]) / 8 / ns))
stats = open(glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]).read().splitlines()
out = ['"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra-legs      (kernel_source_hash %s)"' % line["config"]["kernel_source_hash"],
       '"# persistent step-kernel dispatches of the run, in order (us): %s"' % " ".join("%.0f" % (d / 1e3) for d in durs),
       '"# the LAST %d dispatches are the sustained leg (one region): %s us; the line says sustained.kernel_ms = %s, sustained.ms_per_step = %s"'
       % (S, " ".join("%.0f" % (x / 1e3) for x in sust), (line.get("sustained") or {}).get("kernel_ms"), (line.get("sustained") or {}).get("ms_per_step")),
       '"# the %d before them are the timed 150-turn launches: %d regions of %d launches = %d turns each (bench.py reports the median region).  Region sums (us): %s; median %.0f us = '
       '%.2f us per turn.  The line of the same run: roofline.kernel_ms = %.5f, ms_per_step = %.5f (regions min %.5f, max %.5f)"'
       % (R * L, R, L, K, " ".join("%.0f" % (x / 1e3) for x in regions), med / 1e3, med / K / 1e3, line["roofline"]["kernel_ms"], line["ms_per_step"],
          line["timing"]["min_ms_per_step"], line["timing"]["max_ms_per_step"]),
       '"# (before them: the 150 one-turn launches and the 150-turn settle launch of the desynchronising window, the three launches of the COLD region timed before the clock warm-up -- value_cold --, the 150-turn launches of the clock warm-up on a scratch handle -- bench.py --clock-warmup-ms, reported in the line as timing.clock_warmup_ms = %s -- and the 150-turn warm-up launch)"' % line["timing"].get("clock_warmup_ms"),
       '"# second pass, --pmc GRBM_GUI_ACTIVE: duration (us) @ effective shader clock (GHz) of the same dispatches: %s"' % " ".join("%.0f@%.2f" % (ns / 1e3, g) for ns, g in clk),
       stats[0]] + [l for l in stats[1:] if "evg::" in l]
open("profiles/${NAME}_default_run_kernel_stats.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out[:6]))
PY
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_default_run_kernel_stats.csv gpurun_out/profiles_$NAME/
