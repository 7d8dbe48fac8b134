#!/bin/bash
# Run on the GPU box: instruction counts of the step kernel per ablated phase (libevg_diag.so).  tools/ablate.py plays, per
# ablation mask, 60 + 3 x 50 turns in the persistent form (5 dispatches); the summary takes the last 3 dispatches of each mask.
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ablc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p -- python3 $R/tools/ablate.py 65536 50 > $OUT/ablate.txt 2> $OUT/err.txt || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/p/*/*_counter_collection.csv")[0]
per = {}
for r in csv.DictReader(open(f)):
    if "evg_step_kernel<float, 64, true" in r["Kernel_Name"]:
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)
masks = [0, 16, 2, 18, 4, 1, 17, 49, 0]
names = {0: "full", 16: "no obs write-out", 2: "no combat", 18: "no combat, no write-out", 4: "no movement", 1: "no orders", 17: "no orders, no write-out", 49: "no orders/write-out/state store"}
print("per wave-turn (2048 waves, 50 turns per dispatch; mean of the last 3 dispatches of each mask)")
for k, m in enumerate(masks):
    grp = ids[5 * k: 5 * k + 5][2:]
    if not grp: break
    mean = {c: sum(per[i][c] for i in grp) / len(grp) / 2048 / 50 for c in per[grp[0]]}
    print("ablate=%2d %-32s VALU %6.0f  SALU %5.0f  LDS %4.0f  VMEM %4.0f  BRANCH %4.0f  wave cycles %6.0f  wait_any %5.0f  wait_inst %5.0f" % (
        m, names[m], mean["SQ_INSTS_VALU"], mean["SQ_INSTS_SALU"], mean["SQ_INSTS_LDS"], mean["SQ_INSTS_VMEM"], mean["SQ_INSTS_BRANCH"],
        4 * mean["SQ_WAVE_CYCLES"], 4 * mean["SQ_WAIT_ANY"], 4 * mean["SQ_WAIT_INST_ANY"]))
PY
