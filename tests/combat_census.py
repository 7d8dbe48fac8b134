#!/usr/bin/env python3
"""Measurement helper, not a test (it lives under tests/ because it drives the CPU checker, which only tests, smoke() and the
bench's cpu_baseline leg may load): how many combat work items -- fighting groups -- does a wavefront of 32 (or 64) envs have per
turn in the desynchronised steady state of the benchmark?  DESIGN.md section 6 "The tail of a short launch" quotes it:
    python tests/combat_census.py  ->  items per 32-env wave: mean 82, p10 65, p90 99, max 135; 64-item rounds per wave-turn 1.91;
                                        pooled over 64 envs on 128 lanes 1.98"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as om

N = 8192
o = om.Oracle(N, seed=5, auto_reset=True)
o.reset()
ids = np.arange(N, dtype=np.int64)
phase = (((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150          # bench.py: episode_phase()
for j in range(150):
    o.step_noobs(o.random_actions())
    o.reset(mask=(phase == j).astype(np.uint8))
i32, i64, r32, r64 = [], [], [], []
for t in range(60):
    g = o.get_state()["groups"]
    loc, mov, cnt = g[..., 0], g[..., 4], g[..., 6]
    fight = (cnt > 0) & (mov == 0)
    nf = np.zeros(N, int)
    for n in range(1, 12):
        at = (loc == n) & fight
        nf += at.any(axis=2).all(axis=1) * at.sum(axis=(1, 2))   # groups of both sides at a node held by both
    w32, w64 = nf.reshape(-1, 32).sum(1), nf.reshape(-1, 64).sum(1)
    i32.append(w32); i64.append(w64); r32.append(np.ceil(w32 / 64)); r64.append(np.ceil(w64 / 128))
    o.step_noobs(o.random_actions())
i32, i64, r32, r64 = (np.concatenate(x) for x in (i32, i64, r32, r64))
print("items per 32-env wave: mean %.1f  p10 %d  p50 %d  p90 %d  p99 %d  max %d" % (i32.mean(), *np.percentile(i32, [10, 50, 90, 99, 100])))
print("rounds of 64 items per wave-turn: mean %.2f  histogram %s; lanes busy %.0f %%" % (r32.mean(), np.bincount(r32.astype(int)).tolist(), 100 * i32.sum() / (r32.sum() * 64)))
print("pooled over 64 envs on 128 lanes: rounds per wave-turn mean %.2f  histogram %s; lanes busy %.0f %%" % (r64.mean(), np.bincount(r64.astype(int)).tolist(), 100 * i64.sum() / (r64.sum() * 128)))
