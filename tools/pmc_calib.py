#!/usr/bin/env python3
"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters: copies of known size in three access shapes
(16 B/lane float4 copy, 4 B/lane strided-by-wave int32 copy, and the env-fastest SoA pattern of the step kernel
approximated by an index_select of 256-B segments)."""
import torch
n = 256 * 1024 * 1024 // 4          # 256 MiB of float32 (beyond L2, at the Infinity Cache size)
x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)                       # vectorised copy: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
z = x.view(torch.int32)[::2].contiguous()     # strided read of every other dword (touches all 256 MiB of lines), writes 128 MiB
torch.cuda.synchronize()
print("calib done", float(y[0]), int(z[0]))
