#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc pass over the matrix-core kernels: three launches each of the bf16 codebook search at
1 Mi x 256-d latents against 128 / 512 / 1024 / 2048 / 8192 codes, then three of the object-cloud encoder trunk
(4096 x 3000 points).  tools/mfma_pmc_summary.py turns the counter CSV into one line per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
from dvq_hip import ops, synth  # noqa: E402
from dvq_hip._lib import DVQ_BF16  # noqa: E402
from network.pointnet_encoder import PointNetEncoder  # noqa: E402

dev = torch.device("cuda:0")
n, D = 1 << 20, 256
for K in (128, 512, 1024, 2048, 8192):
    g = torch.Generator(dev).manual_seed(K)
    E = torch.randn(K, D, device=dev, generator=g)
    z = (E[torch.randint(0, K, (n,), device=dev, generator=g)] + 0.3 * torch.randn(n, D, device=dev, generator=g)).to(torch.bfloat16)
    E = E.to(torch.bfloat16)
    ws = ops.vq_prepare(E)
    for _ in range(3):
        ops.vq_argmin(z, E, want_zq=False, prepared=ws)
    torch.cuda.synchronize()
    del z, E, ws
sd = synth.gennet_state_dict(0, with_prior=False)
enc = PointNetEncoder(channel=4)
enc.load_state_dict({k[len("obj_encoder_type."):]: v for k, v in sd.items() if k.startswith("obj_encoder_type.")})
enc = enc.to(dev).eval()
enc.compute_dtype = torch.bfloat16
w = enc._packed().as_dict()
x = synth.object_clouds_device(4096, 3000, 11, dev)
trans = torch.eye(3, device=dev).repeat(4096, 1, 1).contiguous()
for _ in range(3):
    ops.pointnet_mlp_max(x, trans, w, False, DVQ_BF16)
torch.cuda.synchronize()
