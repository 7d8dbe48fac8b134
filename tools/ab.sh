#!/bin/bash
# A/B on ONE box: bench several builds of the library alternately through bench.py's diagnostic --library switch.
# usage (inside gpurun): LIBS="libevg_base.so libevg.so" bash tools/ab.sh [extra bench.py args]   -> gpurun_out/ab.txt
# (libevg_base.so = a copy of an earlier libevg.so kept beside the working build; it must be an ABI-3 build -- export evg_launch_plan --,
#  e.g. `git stash; make -C everglades-ai-wargame_amd/csrc; cp .../libevg.so .../libevg_base.so; git stash pop; make ...`)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LIBS=${LIBS:-"libevg_base.so libevg.so"}
: > gpurun_out/ab.txt
for rep in ${REPS:-1 2 3}; do
  for lib in $LIBS; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --library $PWD/everglades-ai-wargame_amd/$lib "$@" 2>gpurun_out/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['config'].get('one_launch_per_turn') or {}
print('%-20s' % '$lib', 'persistent %.3f G  step %.2f us  kernel %.2f us  (step - kernel %.2f us) | per-turn %.3f G  kernel %.2f us' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, (d['ms_per_step'] - d['roofline']['kernel_ms'])*1e3, o.get('env_steps_per_s',0)/1e9, o.get('kernel_ms',0)*1e3))" >> gpurun_out/ab.txt || { tail -5 gpurun_out/ab_err.txt; exit 1; }
  done
done
cat gpurun_out/ab.txt
