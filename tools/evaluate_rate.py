#!/usr/bin/env python3
"""Diagnostic: games per second of the batched evaluation harness (everglades_amd.evaluate, the mirror of evaluate.py:127-181) between
on-device bots: 65 536 concurrent games per round, rollouts without observations (the harness reads only the episode results)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
for p0, p1 in (("random_actions", "random_actions"), ("cycle_rush_turn25", "swarm_agent"), ("bull_rush", "random_actions")):
    env = evg.EvergladesVecEnv(65536, seed=7, auto_reset=False)
    evg.evaluate(p0, p1, 65536, num_envs=65536, env=env)                    # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = evg.evaluate(p0, p1, 8 * 65536, num_envs=65536, env=env)
    dt = time.perf_counter() - t0
    print("%-20s vs %-16s %7d games in %6.1f ms = %5.1f M games/s; seat 0 wins %.4f (CI %.4f..%.4f), ties %d, mean length %.1f" % (
        p0, p1, r["games"], dt * 1e3, r["games"] / dt / 1e6, r["win_rate"], r["confint"][0], r["confint"][1], r["ties"], r["mean_length"]), flush=True)
    env.close()
