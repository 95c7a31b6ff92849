#!/usr/bin/env python3
"""One-rank RCCL sanity check of the asynchronous all-gather the bench uses at N > 1:
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_async_check.py"""
import os

import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
t = torch.randn(4096, 778, 3, device=dev)
outs, works = [], []
for k in range(3):
    out = torch.empty_like(t)
    w = dist.all_gather_into_tensor(out, t * (k + 1), async_op=True)
    x = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)   # work launched while it is in flight
    outs.append(out)
    works.append(w)
for w in works:
    w.wait()
torch.cuda.synchronize()
assert all(torch.equal(o, t * (k + 1)) for k, o in enumerate(outs))
dist.barrier(device_ids=[dev.index])
dist.destroy_process_group()
print("async all-gather ok")
