#!/bin/bash
# Run on the GPU box: the driver's bench shape (--steps 20 --warmup 5) six times, alternating the build that replays rollout launch plans as library-owned
# hipGraphs (make graphs -> libevg_graphs.so; captured before the timed region) and the product library (plain launches): is replay worth it, and how
# stable is the one short sample the driver takes?
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --library $PWD/everglades-ai-wargame_amd/libevg_graphs.so 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph replay   value %.3f G  step %.2f us  kernel %.2f us' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('plain launches value %.3f G  step %.2f us  kernel %.2f us' % (d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"; done
