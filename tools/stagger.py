#!/usr/bin/env python3
"""Diagnostic (libevg_diag.so): single-turn launches of the step kernel with the wave in hardware slot 1 of every SIMD delayed
by N x 64 cycles at its start.  Prints the STREAM time per launch (two events around 160 back-to-back launches: kernel + launch gap; the kernel alone is in the rocprofv3 traces
under profiles/) in a desynchronised steady state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
N = 65536
SWEEP = [int(x) for x in sys.argv[1:]] or [85, 1, 81, 89, 85]
for a, b, h in [(a_, 0, 0) for a_ in SWEEP]:   # delay = slot x (a - 1) + simd x b sleeps of 64 cycles
    env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=(a << 8) | (b << 16)))
    env.reset()
    ids = torch.arange(N, device=env.device)
    for j in range(150):
        env.rollout_random(1)
        env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
    env.rollout_random(150)
    res = [env.rollout_random(160, time_kernel=True)[-1] * 1e3 for _ in range(3)]
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); env.rollout_random(300); t1.record(); torch.cuda.synchronize()
    print("stagger slot x (%2d - 1) + simd x %2d (x 64 cycles): stream time %s us per launch (160 launches each); 300 launches back to back: %.2f us per turn"
          % (a, b, ["%.2f" % r for r in res], t0.elapsed_time(t1) / 300 * 1e3), flush=True)
    env.close()
