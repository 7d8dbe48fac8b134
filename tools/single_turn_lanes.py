#!/usr/bin/env python3
"""Diagnostic (libevg_diag.so): one launch per turn at small batch sizes -- the two-lane kernel (what the product launches for every
single-turn step) against the four-lane kernel forced through evg_diag_configure(lanes = 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
for N in (1024, 4096, 8192, 16384, 24576, 32768, 49152):
    res = []
    for lanes in (64, 4):
        env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=lanes))
        env.reset()
        ids = torch.arange(N, device=env.device)
        for j in range(150):
            env.rollout_random(1)
            env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
        env.rollout_random(150)
        a = env.random_actions().clone()
        fused = min(env.rollout_random(300, time_kernel=True)[-1] * 1e3 for _ in range(3))
        caller = min(env.rollout_random(300, time_kernel=True, fused=False)[-1] * 1e3 for _ in range(3))
        res.append((fused, caller))
        env.close()
    print("%6d envs: one launch per turn, orders drawn in the kernel: two-lane %.2f us, four-lane %.2f us | orders from a tensor (+ action kernel): %.2f / %.2f us" % (N, res[0][0], res[1][0], res[0][1], res[1][1]), flush=True)
