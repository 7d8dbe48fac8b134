import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg
N = 65536
for lib, lanes in (("libevg_x4.so", 32), ("libevg_x4.so", 64), ("libevg_diag.so", 64)):
    env = evg.EvergladesVecEnv(N, seed=1, auto_reset=True, library=os.path.join(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "everglades-ai-wargame_amd"), lib), diag=dict(lanes=lanes))
    env.reset()
    ids = torch.arange(N, device=env.device)
    for j in range(150):
        env.rollout_random(1)
        env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
    env.rollout_random(150, turns_per_launch=150)
    p = env.rollout_random(150, time_kernel=True, turns_per_launch=150)[-1] * 1e3
    env.rollout_random(16)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record(); env.rollout_random(300); t1.record(); torch.cuda.synchronize()
    print("%s lanes=%d: persistent %.2f us/turn; one launch per turn (300 back to back) %.2f us/turn" % (lib, lanes, p, t0.elapsed_time(t1) / 300 * 1e3), flush=True)
    env.close()
