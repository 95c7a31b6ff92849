#!/usr/bin/env python3
"""Timing of the bf16 search at 512 / 1024 / 2048 codes, 1 Mi latents: time and fraction of the nominal 2.5 PFLOP/s per
codebook size, one line (for alternating runs of two builds on one box).    python tools/vq_rot_time.py [label]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
from dvq_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
n, D = 1 << 20, 256
out = []
for K in (512, 1024, 2048):
    g = torch.Generator(dev).manual_seed(K)
    E = torch.randn(K, D, device=dev, generator=g)
    z = (E[torch.randint(0, K, (n,), device=dev, generator=g)] + 0.3 * torch.randn(n, D, device=dev, generator=g)).to(torch.bfloat16)
    E = E.to(torch.bfloat16)
    ws = ops.vq_prepare(E)
    for _ in range(3):
        ops.vq_argmin(z, E, want_zq=False, prepared=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.vq_argmin(z, E, want_zq=False, prepared=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append(f"K={K}: {ms:.4f} ms {2.0 * n * K * D / ms / 1e9 / 2500:.3f}")
print(sys.argv[1] if len(sys.argv) > 1 else "-", " | ".join(out), flush=True)
