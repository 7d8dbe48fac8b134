#!/usr/bin/env python3
"""Diagnostic (libevg_diag.so): single-turn launches of the step kernel with the wave in hardware slot 1 of every SIMD delayed
by N x 256 cycles at its start.  Prints the mean step-kernel time per launch in a desynchronised steady state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
N = 65536
for stg in (1, 9, 17, 21, 25, 33, 1):        # knob value n = n - 1 sleeps of 256 cycles (0 = the product default)
    env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=stg << 8))
    env.reset()
    ids = torch.arange(N, device=env.device)
    for j in range(150):
        env.rollout_random(1)
        env.reset(mask=((ids % 150) == j).to(torch.uint8))
    env.rollout_random(150)
    res = [env.rollout_random(160, time_kernel=True)[-1] * 1e3 for _ in range(3)]
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); env.rollout_random(300); t1.record(); torch.cuda.synchronize()
    print("stagger %2d - 1 x 256 cycles: step kernel %s us per launch; 300 launches back to back: %.2f us per turn" % (stg, ["%.2f" % r for r in res], t0.elapsed_time(t1) / 300 * 1e3), flush=True)
    env.close()
