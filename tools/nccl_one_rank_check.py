#!/usr/bin/env python3
"""Diagnostic: the torch.distributed calls of the N > 1 bench path on the RCCL backend with a ONE-rank group on one GPU
(gather into unbind() views of a preallocated buffer, barrier with device_ids, all_gather of the per-rank timing tensor).
It cannot measure anything -- it only shows that the backend accepts the calls as bench.py / ResultGather issue them."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
send = torch.arange(8 * 4, dtype=torch.float32, device=dev).reshape(8, 4)
recv = torch.empty((1, 8, 4), dtype=torch.float32, device=dev)
dist.gather(send, list(recv.unbind(0)), dst=0)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
assert torch.equal(recv[0], send)
mine = torch.tensor([1.0, 2.0], device=dev, dtype=torch.float64)
allr = [torch.zeros_like(mine)]
dist.all_gather(allr, mine)
assert torch.equal(allr[0], mine)
print("rccl one-rank gather / barrier / all_gather ok:", dist.get_backend())
dist.destroy_process_group()
