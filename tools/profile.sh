#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/profile.sh <tag>
# rocprofv3 kernel stats of the bench command, then FETCH_SIZE and WRITE_SIZE in separate PMC passes (gfx950: TCC
# has 4 slots; FETCH_SIZE costs 3, WRITE_SIZE 2), each pass also over the calibration workload.
set -o pipefail
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 150 --warmup 150 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 $R/tools/pmc_calib.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 $R/tools/pmc_calib.py > $OUT/calib_write.log 2>&1
find $OUT -name "*.csv" | head -40
