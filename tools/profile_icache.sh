#!/bin/bash
# Run on the GPU box: instruction-cache behaviour of the step kernel in both launch forms (SQC counters + instruction fetch).
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for FORM in persistent perturn; do
  if [ $FORM = persistent ]; then TPL=150; else TPL=1; fi
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/$FORM -- python3 $R/bench.py --steps 150 --warmup 150 --repeats 1 --sustained-launches 0 --no-cpu-baseline --no-extra-legs --turns-per-launch $TPL > $OUT/bench_$FORM.json 2> $OUT/$FORM.err || exit 1
done
python3 - <<PY
import csv, glob
for form, tail, tpl, nwin in (("persistent", "true, false, false, false", 150, 1), ("perturn", "false, false, false, false", 1, 150)):
    f = glob.glob("$OUT/%s/*/*_counter_collection.csv" % form)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "evg_step_kernel<float, 64, " + tail in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-nwin:]
    mean = {c: sum(per[i][c] for i in ids) / len(ids) / 2048 / tpl for c in per[ids[0]]}
    print(form, "per wave-turn:", {c: round(v, 1) for c, v in mean.items()})
PY
