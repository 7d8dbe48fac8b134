#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/profile_configs.sh <name>     e.g. r03_h
# The counter passes of the other bench configurations (BASELINE config 5, int16 observations, 4 096 envs), so that their bench
# lines are priced with measured bytes too.
set -o pipefail
NAME=${1:-r03_x}
TAG=$(echo $NAME | tr -d _)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh ${TAG}c5 --workload scripted > gpurun_out/${TAG}c5_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}c5_profile.log; exit 1; }
bash tools/profile.sh ${TAG}i16 --obs-dtype int16 > gpurun_out/${TAG}i16_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}i16_profile.log; exit 1; }
bash tools/profile.sh ${TAG}4k --envs 4096 > gpurun_out/${TAG}4k_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}4k_profile.log; exit 1; }
python tools/pmc_summary.py ${TAG}c5 ${NAME}_config5 65536 scripted > gpurun_out/${TAG}c5_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}c5_summary.txt; exit 1; }
python tools/pmc_summary.py ${TAG}i16 ${NAME}_int16 65536 random int16 > gpurun_out/${TAG}i16_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}i16_summary.txt; exit 1; }
python tools/pmc_summary.py ${TAG}4k ${NAME}_4096envs 4096 > gpurun_out/${TAG}4k_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}4k_summary.txt; exit 1; }
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_* gpurun_out/profiles_$NAME/
ls gpurun_out/profiles_$NAME
