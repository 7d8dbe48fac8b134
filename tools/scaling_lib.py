#!/usr/bin/env python3
"""Diagnostic: persistent step-kernel time per turn at a few batch sizes for several builds of the library (A/B on one box).
usage: python tools/scaling_lib.py "<N1,N2,...>" lib1.so lib2.so ...   (library names relative to everglades-ai-wargame_amd/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
sizes = [int(x) for x in sys.argv[1].split(",")]
libs = sys.argv[2:]
pkg = os.path.dirname(evg._lib.LIB_PATH)
for N in sizes:
    for lib in libs:
        env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=os.path.join(pkg, lib))
        env.reset()
        ids = torch.arange(N, device=env.device)
        for j in range(150):
            env.rollout_random(1)
            env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
        env.rollout_random(150, turns_per_launch=150)
        ts = [env.rollout_random(150, time_kernel=True, turns_per_launch=150)[-1] * 1e3 for _ in range(5)]
        print("%7d envs  %-18s persistent %6.2f us/turn (five 150-turn launches: %s)   plan: %s" % (N, lib, min(ts), " ".join("%.2f" % t for t in ts), env.launch_plan(150)[1].split(" | ")[0][:140]), flush=True)
        env.close()
