#!/usr/bin/env python3
"""Diagnostic: the persistent form per turn at small and medium batch sizes, the product's choice of step kernel (four lanes per env up to
what the device holds at three waves per SIMD, two lanes above: evg_launch_plan) against the two-lane kernel forced through the diagnostic
library -- the measurements quoted in the header of csrc/evg_step4.inc.  Desynchronised steady state, best of five 150-turn launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192, 16384, 24576, 32768, 40960, 49152, 65536]
for N in sizes:
    row = []
    for name, kw in (("product", dict(library=evg._lib.DIAG_LIB_PATH)), ("two-lane forced", dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=64))),
                     ("four-lane forced", dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=4)))):
        env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, **kw)
        env.reset()
        ids = torch.arange(N, device=env.device)
        for j in range(150):
            env.rollout_random(1)
            env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
        env.rollout_random(150, turns_per_launch=150)
        ts = [env.rollout_random(150, time_kernel=True, turns_per_launch=150)[-1] * 1e3 for _ in range(5)]
        row.append("%s %6.2f" % (name, min(ts)))
        plan = env.launch_plan(150)[1].split(" | ")[0][:90] if name == "product" else plan
        env.close()
    print("%7d envs  persistent us/turn:  %s   | product plan: %s" % (N, "   ".join(row), plan), flush=True)
