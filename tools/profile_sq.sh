#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/profile_sq.sh <tag> [extra bench args]
# SQ counter passes of the bench command (8 SQ slots per pass on gfx950) for both launch forms of the step kernel; counters
# only (no trace domain besides --kernel-trace).  Summarise with tools/sq_summary.py <tag> <name>.
set -o pipefail
TAG=${1:-r02}
shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC"
for FORM in ${FORMS:-persistent perturn learner}; do
  if [ $FORM = persistent ]; then TPL=150; else TPL=1; fi
  EXTRA=""; if [ $FORM = learner ]; then EXTRA="--learner-seat"; fi
  CMD="python3 $R/bench.py --steps 150 --warmup 150 --repeats 1 --sustained-launches 0 --no-cpu-baseline --no-extra-legs --turns-per-launch $TPL $EXTRA $*"
  echo "$CMD" > $OUT/cmd_$FORM.txt
  rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/${FORM}_p1 -- $CMD > $OUT/bench_${FORM}_p1.json 2> $OUT/${FORM}_p1.err || exit 1
  rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT/${FORM}_p2 -- $CMD > $OUT/bench_${FORM}_p2.json 2> $OUT/${FORM}_p2.err || exit 1
  rocprofv3 --kernel-trace --pmc $P3 --output-format csv -d $OUT/${FORM}_p3 -- $CMD > $OUT/bench_${FORM}_p3.json 2> $OUT/${FORM}_p3.err || exit 1
done
find $OUT -name "*counter_collection.csv" | wc -l
