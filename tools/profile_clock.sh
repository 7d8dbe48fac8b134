#!/bin/bash
# Run on the GPU box: effective shader clock of the step kernel = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration
# (MI355X_MICROARCH.md, DVFS give-back); persistent form (3 ms dispatches).
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 150 --warmup 150 --repeats 1 --sustained-launches 0 --no-cpu-baseline --no-extra-legs > $OUT/bench.json 2> $OUT/err.txt || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/p/*/*_counter_collection.csv")[0]
for r in csv.DictReader(open(f)):
    if "evg_step_kernel<float, 64, true" in r["Kernel_Name"]:
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        print("dispatch %s: %.3f ms, GRBM_GUI_ACTIVE %.0f -> %.3f GHz" % (r["Dispatch_Id"], ns / 1e6, float(r["Counter_Value"]), float(r["Counter_Value"]) / 8 / ns))
PY
