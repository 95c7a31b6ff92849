#!/usr/bin/env python3
"""What a search with the codebook split over co-scheduled workgroups could reach, measured with the kernels as they are:
a 512-code (1024-code) search of 1 Mi latents as ONE launch of two (four) 256-code searches over the SAME latents, every
part with its quarter-MiB share of the codebook resident in LDS (the batched entry point: workgroup (x, y) searches part y,
and the workgroups (x, 0..) walk the same latent groups in step, on CUs of one XCD).  The parts' winners are NOT merged, so
the figure is an upper bound for such a design.  Next to it the product's own search of the whole codebook.

    python tools/vq_pair_probe.py [label]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
from dvq_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
n, D = 1 << 20, 256


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for K in (512, 1024):
    g = torch.Generator(dev).manual_seed(K)
    E = torch.randn(K, D, device=dev, generator=g)
    z = (E[torch.randint(0, K, (n,), device=dev, generator=g)] + 0.3 * torch.randn(n, D, device=dev, generator=g)).to(torch.bfloat16)
    E = E.to(torch.bfloat16)
    ws = ops.vq_prepare(E)
    parts = [E[i:i + 256].contiguous() for i in range(0, K, 256)]
    wss = [ops.vq_prepare(p) for p in parts]
    zs = [z] * len(parts)
    for rnd in range(2):
        whole = timed(lambda: ops.vq_argmin(z, E, want_zq=False, prepared=ws))
        split = timed(lambda: ops.vq_argmin_batched(zs, parts, wss))
        frac = lambda ms: 2.0 * n * K * D / ms / 1e9 / 2500
        out.append(f"K={K}: whole {whole:.4f} ms {frac(whole):.3f}, {len(parts)} resident parts (unmerged) {split:.4f} ms {frac(split):.3f}")
print(sys.argv[1] if len(sys.argv) > 1 else "-", "\n  " + "\n  ".join(out), flush=True)
