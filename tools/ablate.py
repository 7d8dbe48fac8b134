#!/usr/bin/env python3
"""Diagnostic: price the phases of the step kernel by skipping them (libevg_diag.so, evg_diag_configure ablate bits).
bit0 orders, bit1 combat, bit2 movement, bit4 obs write-out, bit5 state store.  Not a benchmark."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
TPL = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # turns per launch (50 = persistent form)
MODE = sys.argv[3] if len(sys.argv) > 3 else "phases"
if MODE == "obs":
    # What the observation phase of the persistent form costs, on ONE handle state sequence per variant (same library, same games -- the ablations below do
    # not change the games): full | no write-out (the int16 image is still built in LDS, nothing is read back, converted or stored: ablate bit 4) | no
    # observations at all (obs_out = NULL: neither image nor write-out; the orders are not recorded either).  Alternated three times.
    for rep in range(3):
        for name, abl, observe in (("full", 0, True), ("image built, no write-out", 16, True), ("no image, no write-out", 0, False)):
            env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=abl))
            env.reset()
            env.rollout_random(300, turns_per_launch=TPL, observe=observe, record_actions=observe)
            ms = [env.rollout_random(150, time_kernel=True, turns_per_launch=TPL, observe=observe, record_actions=observe)[-1] * 1e3 for _ in range(3)]
            torch.cuda.synchronize()
            print("%-28s step kernel us per turn over three 150-turn launches: %s" % (name, ["%.2f" % r for r in ms]), flush=True)
            env.close()
    sys.exit(0)
for abl in (0, 16, 2, 18, 4, 1, 17, 49, 0):
    env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=abl))
    env.reset()
    env.rollout_random(60, turns_per_launch=TPL)
    res = []
    for seg in range(3):
        ms = env.rollout_random(50, time_kernel=True, turns_per_launch=TPL)[-1]
        res.append(ms * 1e3)
    torch.cuda.synchronize()
    print("ablate=%2d  step kernel us at turns 61-110 / 111-160(reset at 150) / 161-210: %s" % (abl, ["%.1f" % r for r in res]), flush=True)
    env.close()
