#!/usr/bin/env python3
"""A few launches of the bf16 object-cloud encoder kernel for a rocprofv3 --pmc pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
from dvq_hip import ops, synth  # noqa: E402
from dvq_hip._lib import DVQ_BF16  # noqa: E402
from network.pointnet_encoder import PointNetEncoder  # noqa: E402

dev = torch.device("cuda:0")
sd = synth.gennet_state_dict(0, with_prior=False)
enc = PointNetEncoder(channel=4)
enc.load_state_dict({k[len("obj_encoder_type."):]: v for k, v in sd.items() if k.startswith("obj_encoder_type.")})
enc = enc.to(dev).eval()
enc.compute_dtype = torch.bfloat16
w = enc._packed().as_dict()
x = synth.object_clouds_device(4096, 3000, 11, dev)
trans = torch.eye(3, device=dev).repeat(4096, 1, 1).contiguous()
for _ in range(12):
    ops.pointnet_mlp_max(x, trans, w, False, DVQ_BF16)
torch.cuda.synchronize()
