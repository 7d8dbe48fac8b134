#!/usr/bin/env python3
"""Diagnostic (libevg_stamps.so): the TIMELINE of one single-turn launch of the step kernel (evg_step form, orders drawn in the kernel) in a desynchronised
steady state: when, relative to the first wavefront's start, do the wavefronts of hardware slot 0 and of slot 1 (the two waves of a SIMD) reach each phase
boundary?  Wave start / end are s_memrealtime (100 MHz, global); the phase stamps are shader-clock cycles (s_memtime) scaled per wave to its own duration.
usage: python tools/single_turn_timeline.py [envs] [launches]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import everglades_amd as evg

NAMES = ["start", "tables+state loaded (+stagger)", "orders", "combat0 snapshot", "combat1 worklist", "combatA draws", "combatB apply", "movement",
         "aggregates+capture", "rewards+stats+reset", "obs image", "state store", "obs write-out issued", "end"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.STAMPS_LIB_PATH)
env.reset()
L = env.L
L.evg_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
ids = torch.arange(N, device=env.device)
for j in range(150):                      # desynchronise the episodes like bench.py does
    env.rollout_random(1)
    env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
env.rollout_random(150)
nb = (N + 31) // 32
acc = {0: [], 1: []}
ends, evt = [], []
for rep in range(REPS):
    env.rollout_random(7)
    ms = env.rollout_random(1, time_kernel=True)[-1]
    st = np.zeros(((N + 15) // 16, 16), np.uint64)
    assert L.evg_debug_read_stamps(env._h, st.ctypes.data_as(C.c_void_p)) == 0
    st = st[:nb]
    t0 = (st[:, 14] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    t1 = (st[:, 15] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    slot = ((st[:, 14] >> np.uint64(32)).astype(np.int64)) & 15
    cyc = st[:, :14].astype(np.int64)
    for i in range(1, 14):                # a wave that skips the combat block keeps older stamps there: the phase counts as empty
        cyc[:, i] = np.maximum(cyc[:, i], cyc[:, i - 1])
    rel = (cyc - cyc[:, :1]).astype(np.float64)
    scale = ((t1 - t0) * 10.0 / 1e3) / np.maximum(rel[:, 13], 1.0)          # us per shader cycle of this wave
    start = (t0 - t0.min()) * 10.0 / 1e3
    absolute = start[:, None] + rel * scale[:, None]                        # [wave, boundary] in us since the first wave started
    for s_ in (0, 1):
        if (slot == s_).any():
            acc[s_].append(absolute[slot == s_].mean(axis=0))
    ends.append(((t1 - t0.min()) * 10.0 / 1e3).max())
    evt.append(ms * 1e3)
print("%d envs, %d single-turn launches: HIP-event time %.1f us (stamps build), last wave ends %.1f us after the first one started" % (N, REPS, np.mean(evt), np.mean(ends)))
print("mean time (us since the first wavefront of the launch started) at which the waves of a hardware slot pass each boundary, and the phase's mean length:")
for s_ in (0, 1):
    if not acc[s_]:
        continue
    a = np.mean(acc[s_], axis=0)
    print(" hardware slot %d (%s wave of its SIMD)" % (s_, "first" if s_ == 0 else "second, staggered"))
    prev = a[0]
    for i, nm in enumerate(NAMES):
        print("   %-34s at %6.2f us   (+%5.2f)" % (nm, a[i], a[i] - prev))
        prev = a[i]
env.close()
