#!/usr/bin/env python3
"""Diagnostic (libevg_stamps.so): how long does each wavefront of ONE persistent launch of the step kernel run, and where?
Prints the distribution of per-wave launch durations (s_memrealtime, 100 MHz), grouped by XCD, by SIMD pairing (the two
waves that share a SIMD: older / younger by start time) and by CU.  usage: python tools/wave_times.py [envs] [turns]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import everglades_amd as evg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
TURNS = int(sys.argv[2]) if len(sys.argv) > 2 else 150
env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.STAMPS_LIB_PATH)
env.reset()
L = env.L
L.evg_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
ids = torch.arange(N, device=env.device)
for j in range(150):                      # desynchronise the episodes like bench.py does
    env.rollout_random(1)
    env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
env.rollout_random(150, turns_per_launch=150)
for rep in range(3):
    ms = env.rollout_random(TURNS, turns_per_launch=TURNS, time_kernel=True)[-1] * TURNS
    st = np.zeros(((N + 15) // 16, 16), np.uint64)
    assert L.evg_debug_read_stamps(env._h, st.ctypes.data_as(C.c_void_p)) == 0
    st = st[:(N + 31) // 32]
    t0 = (st[:, 14] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    t1 = (st[:, 15] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    hw = (st[:, 14] >> np.uint64(32)).astype(np.int64)
    xcc = (st[:, 15] >> np.uint64(32)).astype(np.int64) & 0xF
    dur = (t1 - t0) * 10.0 / 1e3          # us
    start = (t0 - t0.min()) * 10.0 / 1e3
    end = (t1 - t0.min()) * 10.0 / 1e3
    wave_id, simd, pipe, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    print("launch of %d turns: HIP-event time %.1f us; per-wave duration mean %.1f  min %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us; last start %.1f us, "
          "last end %.1f us"
          % (TURNS, ms * 1e3, dur.mean(), dur.min(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), start.max(), end.max()))
    print("   wave END times: p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % tuple(np.percentile(end, q) for q in (10, 50, 90, 99, 100)))
    print("   by XCD: " + "  ".join("%d: n=%d mean %.0f max %.0f" % (x, (xcc == x).sum(), dur[xcc == x].mean(), dur[xcc == x].max()) for x in sorted(set(xcc))))
    key = (xcc * 8 + se) * 64 + sh * 32 + cu * 2 + 0
    simd_key = key * 4 + simd
    pairs = {}
    for i, k in enumerate(simd_key):
        pairs.setdefault(int(k), []).append(i)
    sizes = np.bincount([len(v) for v in pairs.values()])
    print("   waves per (xcc, se, sh, cu, simd) slot histogram:", sizes.tolist(), " distinct slots:", len(pairs), " distinct CUs:", len(set(key.tolist())))
    older, younger = [], []
    for v in pairs.values():
        if len(v) == 2:
            a, b = sorted(v, key=lambda i: (t0[i], wave_id[i]))
            older.append(dur[a]); younger.append(dur[b])
    if older:
        older, younger = np.array(older), np.array(younger)
        print("   SIMD pairs (%d): first-started wave mean %.1f us, second-started mean %.1f us; mean |difference| %.1f us; slower-of-pair mean %.1f" %
              (len(older), older.mean(), younger.mean(), np.abs(older - younger).mean(), np.maximum(older, younger).mean()))
    byw = [dur[wave_id == w].mean() if (wave_id == w).any() else float("nan") for w in range(10)]
    print("   by HW wave slot:", ["%.0f" % x for x in byw], " counts", [int((wave_id == w).sum()) for w in range(10)])
    cu_mean = {}
    for i, k in enumerate(key):
        cu_mean.setdefault(int(k), []).append(dur[i])
    cm = np.array([np.mean(v) for v in cu_mean.values()])
    print("   per-CU mean duration: min %.1f  p50 %.1f  max %.1f us; waves per CU histogram %s" % (cm.min(), np.percentile(cm, 50), cm.max(),
                                                                                                   np.bincount([len(v) for v in cu_mean.values()]).tolist()))
env.close()
