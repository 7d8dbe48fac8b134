#!/usr/bin/env python3
"""Diagnostic: host cost of the Python API per call, and hipGraph capture of a whole turn (torch.cuda.graph)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg

for N in (64, 65536):
    env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True)
    env.reset()
    a = env.random_actions()
    torch.cuda.synchronize()
    rows = env.random_actions_seat(0)
    for name, fn in (("random_actions+step", lambda: env.step(env.random_actions())), ("step only", lambda: env.step(a)),
                     ("random_actions_seat+step_vs", lambda: env.step_vs("swarm", env.random_actions_seat(0))), ("step_vs only", lambda: env.step_vs("swarm", rows))):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("N=%6d %-22s host issue %.1f us/turn, with GPU drain %.1f us/turn" % (N, name, t_issue / 2000 * 1e6, t_all / 2000 * 1e6), flush=True)
    # whole turn captured in a hipGraph: generator kernel + step kernel on the capture stream
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            env.step(env.random_actions())
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        env.step(env.random_actions())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        g.replay()
    torch.cuda.synchronize()
    print("N=%6d graph replay of one turn: %.1f us/turn" % (N, (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
    env.close()
