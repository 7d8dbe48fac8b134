#!/usr/bin/env python3
"""Diagnostic: step-kernel time per turn against the batch size, both launch forms, in a desynchronised steady state
(32 envs per wavefront: 32 768 envs = one wave per SIMD, 65 536 = two)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
for N in (2048, 8192, 16384, 32768, 49152, 65536, 98304, 131072, 262144):
    env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True)
    env.reset()
    ids = torch.arange(N, device=env.device)
    for j in range(150):
        env.rollout_random(1)
        env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
    env.rollout_random(150, turns_per_launch=150)
    p = env.rollout_random(150, time_kernel=True, turns_per_launch=150)[-1] * 1e3
    env.rollout_random(16)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record(); env.rollout_random(300); t1.record(); torch.cuda.synchronize()
    s = t0.elapsed_time(t1) / 300 * 1e3
    print("%7d envs (%4.2f waves/SIMD): persistent %6.2f us/turn = %6.1f M env-steps/s | one launch per turn (300 back to back) %6.2f us/turn = %6.1f M env-steps/s" % (
        N, N / 32 / 1024, p, N / p, s, N / s), flush=True)
    env.close()
