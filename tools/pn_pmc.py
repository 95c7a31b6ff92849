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

# the other HBM-facing kernels: three launches each of the hand layer (4096 hands) and of the bf16 codebook search
# at the reference's codebook shape (1 Mi x 256-d latents, 128 codes)
import mano as M  # noqa: E402
from dvq_hip.mano_io import Struct  # noqa: E402

rh = M.MANO('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45, batch_size=1,
            flat_hand_mean=True).to(dev).eval()
go, hp, tr = (t.to(dev) for t in synth.mano_inputs(4096, seed=1))
for _ in range(3):
    rh(global_orient=go, hand_pose=hp, transl=tr)
g = torch.Generator(dev).manual_seed(128)
E = torch.randn(128, 256, device=dev, generator=g)
z = (E[torch.randint(0, 128, (1 << 20,), device=dev, generator=g)] + 0.3 * torch.randn(1 << 20, 256, device=dev, generator=g)).to(torch.bfloat16)
E = E.to(torch.bfloat16)
ws = ops.vq_prepare(E)
for _ in range(3):
    ops.vq_argmin(z, E, want_zq=False, prepared=ws)
torch.cuda.synchronize()
