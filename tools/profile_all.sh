#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/profile_all.sh <name>      e.g.  r03_f
# Everything bench.py's roofline objects are priced with, for the build in the tree: the three launch forms at 65 536 envs, the product's
# launch at 262 144 envs (whole rounds one after the other), the chunked form forced over all 262 144 envs (a working set the Infinity
# Cache cannot hold; diagnostic library), and the SQ counter passes.  Summaries land in profiles/<name>_*.
set -o pipefail
NAME=${1:-r03_x}
TAG=$(echo $NAME | tr -d _)
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}_profile.log; exit 1; }
FORMS="persistent perturn" bash tools/profile.sh ${TAG}262k --envs 262144 > gpurun_out/${TAG}262k_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}262k_profile.log; exit 1; }
FORMS="persistent" bash tools/profile.sh ${TAG}262kc --envs 262144 --library $R/everglades-ai-wargame_amd/libevg_diag.so --diag-lanes 2 > gpurun_out/${TAG}262kc_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}262kc_profile.log; exit 1; }
bash tools/profile_sq.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1 || { tail -5 gpurun_out/${TAG}_sq.log; exit 1; }
python tools/pmc_summary.py $TAG $NAME > gpurun_out/${TAG}_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}_summary.txt; exit 1; }
python tools/pmc_summary.py ${TAG}262k ${NAME}_262144envs 262144 > gpurun_out/${TAG}262k_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}262k_summary.txt; exit 1; }
python tools/pmc_summary.py ${TAG}262kc ${NAME}_262144envs_cycled 262144 random float32 cycled > gpurun_out/${TAG}262kc_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}262kc_summary.txt; exit 1; }
python tools/sq_summary.py $TAG $NAME > gpurun_out/${TAG}_sq_summary.txt 2>&1 || { tail -5 gpurun_out/${TAG}_sq_summary.txt; exit 1; }
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_* gpurun_out/profiles_$NAME/
ls gpurun_out/profiles_$NAME
