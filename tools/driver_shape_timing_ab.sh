#!/bin/bash
# Run on the GPU box: the driver's bench shape (--steps 20 --warmup 5), alternating the two ways bench.py can take the launch duration of the timed region
# (--timing native: events recorded and read inside the rollout call, which synchronises; --timing torch: two pre-created torch events around an untimed
# call, one synchronisation in the closing bracket).
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do for t in native torch; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --timing $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-7s value %.3f G  step %.2f us  kernel %.2f us' % ('$t', d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"; done; done
