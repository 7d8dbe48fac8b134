#!/bin/bash
# Run on the GPU box: rocprofv3 --kernel-trace --stats of EXACTLY the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5), so that the kernel time in
# its bench line (roofline.kernel_ms x 20) can be read against the profiler's duration of the timed 20-turn dispatch.  -> profiles/<name>_driver_shape_kernel_stats.csv
set -o pipefail
NAME=${1:-r04_f}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${NAME}_driver
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
cd $R
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/stats/*/*_kernel_trace.csv")[0]
rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort()
pers = [(i, n, d) for i, n, d in rows if "evg_step_kernel<float, 64, true, false, false, false" in n]
line = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
stats = open(glob.glob("$OUT/stats/*/*_kernel_stats.csv")[0]).read().splitlines()
R = int((line.get("timing") or {}).get("repeats", 1))
S = int((line.get("sustained") or {}).get("launches", 0))  # the sustained leg: S 150-turn launches in ONE region, right after the timed regions
sust = [d for _, _, d in pers[len(pers) - S:]] if S else []
timed = [d for _, _, d in pers[len(pers) - S - R:len(pers) - S]]    # the R timed regions are the R 20-turn dispatches before them (bench.py reports the median region)
med = sorted(timed)[(R - 1) // 2]
out = ['"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs"',
       '"# persistent step-kernel dispatches of the run, in order (ns): %s"' % " ".join(str(d) for _, _, d in pers),
       '"# the LAST %d are the launches of the sustained leg (one region of %d x 150 turns): %s ns = %.2f us per turn; the line of the same run says sustained.kernel_ms = %s, sustained.ms_per_step = %s (region %s ms)"'
       % (S, S, " ".join(str(d) for d in sust), (sum(sust) / (150.0 * S) / 1e3) if S else 0.0, (line.get("sustained") or {}).get("kernel_ms"), (line.get("sustained") or {}).get("ms_per_step"), (line.get("sustained") or {}).get("region_ms")),
       '"# the %d before them are the timed 20-turn launches (bench.py times the exact 20-step region %d times and reports the median region): %s ns; median %d ns = %.2f us per turn; '
       'the line of the same run says roofline.kernel_ms = %.5f (x 20 turns = %.1f us between the two stream events of its median region), ms_per_step = %.5f (min %.5f, max %.5f)"'
       % (R, R, " ".join(str(d) for d in timed), med, med / 20 / 1e3, line["roofline"]["kernel_ms"], line["roofline"]["kernel_ms"] * 20 * 1e3, line["ms_per_step"],
          (line.get("timing") or {}).get("min_ms_per_step", line["ms_per_step"]), (line.get("timing") or {}).get("max_ms_per_step", line["ms_per_step"])),
       '"# (the others: the 150-turn settle launch of the desynchronising window, the 20-turn COLD region timed before the clock warm-up -- value_cold --, the 150-turn launches of the clock warm-up on a scratch handle -- bench.py --clock-warmup-ms -- and the 5-turn warm-up launch; under the profiler every dispatch is serialised)"',
       stats[0]] + [l for l in stats[1:] if "evg::" in l]
open("profiles/${NAME}_driver_shape_kernel_stats.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out[:5]))
PY
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_driver_shape_kernel_stats.csv gpurun_out/profiles_$NAME/
