#!/usr/bin/env python3
"""Diagnostic: stream time of evg_smart_actions against evg_smart_get_action (DQNAgent.get_action: coin + get_random_actions) at 65 536 envs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True)
env.reset()
env.rollout_random(60, turns_per_launch=60)
q = torch.randn((N, 12, 5), device=env.device)
sobs = env.observe_seat(0)


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("evg_smart_actions                 %.2f us per call" % timed(lambda: env.smart_actions(q, obs=sobs)))
for eps in (0.0, 0.1, 1.0):
    print("evg_smart_get_action eps = %.1f    %.2f us per call" % (eps, timed(lambda: env.smart_get_action(q, eps, seat=0, obs=sobs))))
sh = torch.empty((N, 34), device=env.device)
sw = torch.empty((N, 12, 13), device=env.device)
rows = env.random_actions_seat(0).clone()
print("evg_step_vs_policy                %.2f us per call" % timed(lambda: env.step_vs("random", rows, seat=0)))
print("evg_step_vs_policy_smart          %.2f us per call (the same turn + the Smart_State features of the new observation)" % timed(lambda: env.step_vs("random", rows, seat=0, features=(sh, sw))))
print("evg_smart_state_compact           %.2f us per call (the separate feature kernel it replaces)" % timed(lambda: env.smart_state_compact(-1, sobs, sh, sw)))
env.close()
