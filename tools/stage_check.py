#!/usr/bin/env python3
"""Round-6 experiment (make stage): the staged-order builds must play and RECORD the same games as the product library (full and ragged batches, persistent form)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import everglades_amd as evg

for lib in ("libevg_stage.so", "libevg_stage_nt.so"):
    for N in (65536, 65536 + 1000 + 13, 77):
        outs = []
        for path in (None, os.path.join(evg._lib.HERE, lib)):
            env = evg.EvergladesVecEnv(N, seed=9, auto_reset=True, library=path)
            env.reset()
            rec = []
            for tpl in (1, 37, 150):
                env.rollout_random(tpl, turns_per_launch=tpl)
                rec.append(env._actions.cpu().numpy().copy())
                rec.append(env.obs.cpu().numpy().copy())
            s = env.get_state()
            outs.append((rec, s))
            env.close()
        ok = all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and all(np.array_equal(outs[0][1][k], outs[1][1][k]) for k in outs[0][1])
        print(lib, N, "equal" if ok else "DIFFERENT", flush=True)
        assert ok
