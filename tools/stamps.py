#!/usr/bin/env python3
"""Diagnostic: where does a wave of the step kernel spend its cycles?  usage: stamps.py [envs] [turns per launch] [scripted]
  Needs `make -C .../csrc stamps`
(libevg_stamps.so, s_memtime at phase boundaries).  Not part of the product path, not a benchmark."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import everglades_amd as evg

NAMES = ["tables+state load", "orders", "combat0 snapshot", "combat1 worklist", "combatA draws", "combatB apply", "movement",
         "aggregates+capture", "rewards+stats+reset", "obs build", "state store", "obs write-out", "reset fill"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
TPL = int(sys.argv[2]) if len(sys.argv) > 2 else 1       # turns per launch (stamps are those of the launch's last turn)
SCRIPTED = len(sys.argv) > 3 and sys.argv[3] == "scripted"   # BASELINE config 5 (Cycle_BRush_Turn25 vs SwarmAgent, fused) instead of random orders
env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.STAMPS_LIB_PATH)
env.reset()
L = env.L
L.evg_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
nb_buf = (N + 15) // 16          # the library sizes the buffer for the 16-env-per-wave variant
nb = (N + 31) // 32              # workgroups of the default variant (32 envs per wave)
for upto in ((20, 50, 80) if SCRIPTED else (20, 80, 140)):
    cur = int(env.get_state()["env"][0, 0])
    if SCRIPTED:
        env.rollout_policies(upto - cur, "cycle_rush_turn25", "swarm", turns_per_launch=TPL)
    else:
        env.rollout_random(upto - cur, turns_per_launch=TPL)
    st = np.zeros((nb_buf, 16), np.uint64)
    assert L.evg_debug_read_stamps(env._h, st.ctypes.data_as(C.c_void_p)) == 0
    st = st[:nb]
    st = st.astype(np.int64)
    for i in range(1, 14):         # a wave that skips the combat block keeps that block's stamps of an earlier turn: older than its
        st[:, i] = np.maximum(st[:, i], st[:, i - 1])   # own stamp before -> the phase counts as empty
    d = np.diff(st[:, :14], axis=1)
    tot = st[:, 13] - st[:, 0]
    span = int(st[:, 13].max() - st[:, 0].min())
    print("turn %d: mean wave %.0f cycles (max %.0f), first-start to last-end %d cycles (memtime ticks, 100 MHz => x10 ns)" % (upto, tot.mean(), tot.max(),
                                                                                                                               span))
    for i, nm in enumerate(NAMES):
        print("   %-20s mean %8.0f  max %8.0f  (%4.1f%%)" % (nm, d[:, i].mean(), d[:, i].max(), 100.0 * d[:, i].mean() / tot.mean()))
env.close()
