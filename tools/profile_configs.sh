#!/bin/bash
# Run on the GPU box through gpurun:  [CONFIGS="c5 i16 4k"] bash tools/profile_configs.sh <name>     e.g. r04_f
# The counter passes of the other bench configurations (BASELINE config 5, int16 observations, 4 096 envs), so that their bench
# lines are priced with measured bytes too.  CONFIGS selects a subset (all three do not fit one 20-minute GPU call).
set -o pipefail
NAME=${1:-r04_x}
TAG=$(echo $NAME | tr -d _)
R=$GRAFT_REPO_ROOT
cd $R
fail() { tail -5 $1; exit 1; }
for C in ${CONFIGS:-c5 i16 4k}; do
  case $C in
    c5)  ARGS="--workload scripted"; OUT="${NAME}_config5 65536 scripted" ;;
    i16) ARGS="--obs-dtype int16";   OUT="${NAME}_int16 65536 random int16" ;;
    4k)  ARGS="--envs 4096";         OUT="${NAME}_4096envs 4096" ;;
  esac
  FORMS="${FORMS:-persistent perturn caller}" bash tools/profile.sh ${TAG}$C $ARGS > gpurun_out/${TAG}${C}_profile.log 2>&1 || fail gpurun_out/${TAG}${C}_profile.log
  python tools/pmc_summary.py ${TAG}$C $OUT > gpurun_out/${TAG}${C}_summary.txt 2>&1 || fail gpurun_out/${TAG}${C}_summary.txt
done
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_* gpurun_out/profiles_$NAME/
ls gpurun_out/profiles_$NAME
