"""Diagnostic: the Smart_State feature kernel (evg_smart_state / evg_smart_state_seat) at 65 536 envs for two builds of the library side by side
(libevg_base.so = a copy of an earlier libevg.so kept beside the working build): time per call and achieved bytes per second."""
import sys, torch
sys.path.insert(0, '.')
import everglades_amd as evg
import os
for lib in [l for l in ("libevg_base.so", "libevg.so") if os.path.exists("everglades-ai-wargame_amd/" + l)]:
    env = evg.EvergladesVecEnv(65536, seed=3, auto_reset=True, library="everglades-ai-wargame_amd/" + lib)
    env.reset(); env.rollout_random(60, turns_per_launch=60)
    so = env.observe_seat(0)
    for name, fn in (("full obs", lambda: env.smart_state(0)), ("seat obs", lambda: env.smart_state(0, so))):
        out = torch.empty((65536, 12, 59), dtype=torch.float32, device=env.device)
        f = (lambda: env.smart_state(0, out=out)) if name == "full obs" else (lambda: env.smart_state(0, so, out=out))
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        print("%-16s %-9s %6.1f us  %5.2f TB/s" % (lib, name, us, (420 + 2832) * 65536 / us / 1e6))
    if hasattr(env.L, "evg_smart_state_compact"):
        sh = torch.empty((65536, 34), dtype=torch.float32, device=env.device); sw = torch.empty((65536, 12, 13), dtype=torch.float32, device=env.device)
        for name, src in (("full obs", None), ("seat obs", so)):
            f = lambda: env.smart_state_compact(0, src, shared=sh, swarm=sw)
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 100 * 1e3
            print("%-16s %-9s %6.1f us  %5.2f TB/s   (compact: shared [34] + swarm [12][13])" % (lib, name, us, (420 + 760) * 65536 / us / 1e6))
    env.close()
