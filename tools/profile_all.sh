#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/profile_all.sh <name> [stage]      e.g.  r04_f main
# Everything bench.py's roofline objects are priced with, for the build in the tree, in three stages that each fit one gpurun call:
#   main  the four launch forms at 65 536 envs (persistent, one launch per turn, caller-supplied orders, learner seat): kernel trace + stats, FETCH_SIZE and
#         WRITE_SIZE passes, calibration copy
#   big   the product's launch at 262 144 envs (whole rounds one after the other); the PRODUCT library made to leave the Infinity Cache -- 131 071 envs with
#         evg_config.cache_mib = 1 024, whose plan is ONE chunked launch that cycles through all 359 MB every 25-turn chunk --; and the chunked form forced
#         over all 262 144 envs (723 MB; diagnostic library), the cross-check
#   sq    the SQ counter passes (persistent, one launch per turn, learner seat)
# Summaries land in profiles/<name>_* (and a copy under gpurun_out/profiles_<name>/, which is what comes back from the box).
set -o pipefail
NAME=${1:-r04_x}
STAGE=${2:-all}
TAG=$(echo $NAME | tr -d _)
R=$GRAFT_REPO_ROOT
cd $R
fail() { tail -5 $1; exit 1; }
if [ $STAGE = main ] || [ $STAGE = all ]; then
  bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || fail gpurun_out/${TAG}_profile.log
  python tools/pmc_summary.py $TAG $NAME > gpurun_out/${TAG}_summary.txt 2>&1 || fail gpurun_out/${TAG}_summary.txt
fi
if [ $STAGE = big ] || [ $STAGE = all ]; then
  FORMS="persistent perturn" bash tools/profile.sh ${TAG}262k --envs 262144 > gpurun_out/${TAG}262k_profile.log 2>&1 || fail gpurun_out/${TAG}262k_profile.log
  FORMS="persistent" bash tools/profile.sh ${TAG}262kc --envs 262144 --library $R/everglades-ai-wargame_amd/libevg_diag.so --diag-lanes 2 > gpurun_out/${TAG}262kc_profile.log 2>&1 || fail gpurun_out/${TAG}262kc_profile.log
  python tools/pmc_summary.py ${TAG}262k ${NAME}_262144envs 262144 > gpurun_out/${TAG}262k_summary.txt 2>&1 || fail gpurun_out/${TAG}262k_summary.txt
  FORMS="persistent" bash tools/profile.sh ${TAG}131kp --envs 131071 --cache-mib 1024 > gpurun_out/${TAG}131kp_profile.log 2>&1 || fail gpurun_out/${TAG}131kp_profile.log
  python tools/pmc_summary.py ${TAG}131kp ${NAME}_131071envs_product_cycled 131071 random float32 product_cycled > gpurun_out/${TAG}131kp_summary.txt 2>&1 || fail gpurun_out/${TAG}131kp_summary.txt
  python tools/pmc_summary.py ${TAG}262kc ${NAME}_262144envs_cycled 262144 random float32 cycled > gpurun_out/${TAG}262kc_summary.txt 2>&1 || fail gpurun_out/${TAG}262kc_summary.txt
fi
if [ $STAGE = sq ] || [ $STAGE = all ]; then
  bash tools/profile_sq.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1 || fail gpurun_out/${TAG}_sq.log
  python tools/sq_summary.py $TAG $NAME > gpurun_out/${TAG}_sq_summary.txt 2>&1 || fail gpurun_out/${TAG}_sq_summary.txt
fi
mkdir -p gpurun_out/profiles_$NAME && cp profiles/${NAME}_* gpurun_out/profiles_$NAME/
ls gpurun_out/profiles_$NAME
