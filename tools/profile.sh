#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/profile.sh <tag> [extra bench args]
# For each launch form of the step kernel (persistent: 150 turns per launch; one launch per turn) the bench command is run
# three times under rocprofv3: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes (gfx950: TCC has
# 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2), plus the calibration workload under the same two counters.
# Summarise with tools/pmc_summary.py <tag> <name>.
set -o pipefail
TAG=${1:-r02}
shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# FORMS="persistent perturn caller learner" (default) selects the launch forms; caller = per turn evg_random_actions into a tensor + evg_step(actions);
# learner = per turn evg_random_actions_seat into a tensor + evg_step_vs_policy (bot inside the step kernel, one seat's observation)
for FORM in ${FORMS:-persistent perturn caller learner}; do
  if [ $FORM = persistent ]; then TPL=150; else TPL=1; fi
  EXTRA=""; if [ $FORM = caller ]; then EXTRA="--caller-actions"; fi; if [ $FORM = learner ]; then EXTRA="--learner-seat"; fi
  CMD="python3 $R/bench.py --steps 150 --warmup 150 --repeats 1 --sustained-launches 0 --no-cpu-baseline --no-extra-legs --turns-per-launch $TPL $EXTRA $*"
  echo "$CMD" > $OUT/cmd_$FORM.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${FORM}_stats -- $CMD > $OUT/bench_${FORM}_stats.json 2> $OUT/${FORM}_stats.err || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${FORM}_fetch -- $CMD > $OUT/bench_${FORM}_fetch.json 2> $OUT/${FORM}_fetch.err || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${FORM}_write -- $CMD > $OUT/bench_${FORM}_write.json 2> $OUT/${FORM}_write.err || exit 1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 $R/tools/pmc_calib.py > $OUT/calib_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 $R/tools/pmc_calib.py > $OUT/calib_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | wc -l
