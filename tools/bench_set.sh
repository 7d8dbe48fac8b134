#!/bin/bash
# Run on the GPU box through gpurun:  bash tools/bench_set.sh <tag>
# The bench lines kept under profiles/: default run, the driver's short shape, BASELINE config 5 (scripted bots, fused), the
# 4 096-env configuration and int16 observations.  Each is ONE JSON line in gpurun_out/bench_<tag>_<name>.json.
set -o pipefail
TAG=${1:-r02}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# stdout = the compact line the driver sees; --details = the full object (what is committed under profiles/)
run() { name=$1; shift; timeout -k 10 400 python bench.py --details gpurun_out/bench_${TAG}_$name.full.json "$@" > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err || { tail -5 gpurun_out/bench_${TAG}_$name.err; exit 1; }; }
run final
run driver_shape --steps 20 --warmup 5
run config5 --workload scripted --no-cpu-baseline
run 4096envs --envs 4096 --no-cpu-baseline
run int16 --obs-dtype int16 --no-cpu-baseline
# the N > 1 code path, rehearsed on the one GPU of the box: two ranks over gloo (two processes sharing the card), and RCCL with a one-rank group
run rehearse_gloo2 --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline
run rehearse_rccl1 --rehearse-distributed --steps 20 --warmup 5 --no-cpu-baseline
python - <<P
import json, glob
for f in sorted(x for x in glob.glob("gpurun_out/bench_${TAG}_*.json") if not x.endswith(".full.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])        # (gloo prints connection notes on stdout before the line)
    o = d["config"].get("one_launch_per_turn") or {}
    c = d["config"].get("caller_actions_per_turn") or {}
    print("%-28s %.3f G env-steps/s  %.2f us/step  roofline.frac %s  hash %s  per-turn %.3f G (frac %s)  caller actions %.3f G (frac %s)  %s" % (f.split("bench_")[1], d["value"] / 1e9, d["ms_per_step"] * 1e3,
          d["roofline"].get("frac"), d["config"].get("kernel_source_hash"), o.get("env_steps_per_s", 0) / 1e9, (o.get("roofline") or {}).get("frac"),
          c.get("env_steps_per_s", 0) / 1e9, (c.get("roofline") or {}).get("frac"), (d.get("distributed") or {}).get("backend", "")))
P
