#!/usr/bin/env python3
"""Diagnostic: time the small kernels around the step kernel (torch events, N = 65 536) and price them against their
algorithmic bytes.  Not a benchmark of the product path (bench.py is)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = evg.EvergladesVecEnv(N, seed=3, auto_reset=True)
env.reset()
env.rollout_random(60, turns_per_launch=60)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


mask = torch.ones(N, dtype=torch.uint8, device=env.device)
cases = [
    ("evg_random_actions", lambda: env.random_actions(), 8 + 112),
    ("evg_scripted_actions (cycle_rush_turn25)", lambda: env.scripted_actions("cycle_rush_turn25", 0), 420 + 12 + 56),
    ("evg_scripted_actions (swarm)", lambda: env.scripted_actions("swarm", 1), 420 + 12 + 56),
    ("evg_fog_of_war (fog + knowledge)", lambda: (env.fog_of_war(), env.knowledge()), 2 * (96 + 22 + 8) + 44),
    ("evg_sightings", lambda: env.sightings(), 96 + 22 + 8 + 96),
    ("evg_smart_state", lambda: env.smart_state(0), 420 + 12 * 59 * 4),
    ("evg_observe", lambda: env.observe(), 96 + 24 + 22 + 8 + 840),
    ("evg_step (one turn, given actions)", lambda: env.step(env._actions), 4530),
]
for name, fn, nbytes in cases:
    us = timeit(fn)
    print("%-42s %8.1f us   %6d B/env algorithmic  -> %6.0f GB/s" % (name, us, nbytes, nbytes * N / us / 1e3))
st = env.get_state()
us = timeit(lambda: env.reset(mask), reps=20)
print("%-42s %8.1f us   %6d B/env algorithmic  -> %6.0f GB/s" % ("evg_reset (all envs)", us, 1771 + 840, (1771 + 840) * N / us / 1e3))
env.close()
