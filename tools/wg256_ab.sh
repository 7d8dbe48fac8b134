#!/bin/bash
# Run on the GPU box: A/B of the single-turn launch form, one wavefront per workgroup (the product's: 2 048 workgroup dispatches at 65 536 envs) against four
# independent wavefronts per 256-thread workgroup (512 dispatches) -- both from the diagnostic library (lanes 64 / 256), alternating on one box.
# -> gpurun_out/wg256_ab.txt
cd "$(dirname "$0")/.."
D=$PWD/everglades-ai-wargame_amd/libevg_diag.so
: > gpurun_out/wg256_ab.txt
for rep in 1 2 3 4; do
  for lanes in 64 256; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-legs --turns-per-launch 1 --steps 150 --repeats 5 --library $D --diag-lanes $lanes "$@" 2>gpurun_out/wg256_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['timing']
print('waves per workgroup %d   %.3f G  step %.2f us (min %.2f max %.2f of %d regions)  stream time per turn %.2f us' % ($lanes // 64 if $lanes == 256 else 1, d['value']/1e9, d['ms_per_step']*1e3, t['min_ms_per_step']*1e3, t['max_ms_per_step']*1e3, t['repeats'], d['roofline']['kernel_ms']*1e3))" >> gpurun_out/wg256_ab.txt || { tail -5 gpurun_out/wg256_err.txt; exit 1; }
  done
done
cat gpurun_out/wg256_ab.txt
