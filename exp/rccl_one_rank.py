#!/usr/bin/env python3
"""RCCL smoke on ONE GPU (world_size 1): the exact calls bench.py makes for N > 1 -- gather of a device buffer into rows of a
pre-allocated [world, n] device tensor on a side stream, all_reduce MAX / SUM, barrier -- accepted and correct over the nccl backend."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
n = 3 * 1024 * 1024
st = torch.cuda.Stream()
local = torch.arange(n, dtype=torch.int64, device="cuda").to(torch.uint8)
out = torch.zeros((1, n), dtype=torch.uint8, device="cuda")
with torch.cuda.stream(st):
    dist.gather(local, gather_list=list(out.unbind(0)), dst=0)
torch.cuda.synchronize()
dist.barrier()
assert torch.equal(out[0], local)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
r = torch.tensor([1, 2, 3], dtype=torch.int64, device="cuda")
dist.all_reduce(r, op=dist.ReduceOp.SUM)
assert float(t.item()) == 1.5 and r.tolist() == [1, 2, 3]
dist.destroy_process_group()
print("rccl one-rank smoke OK")
