#!/usr/bin/env python3
"""A short refinement run (4096 grasps, 6 iterations, hands touching their objects) for a rocprofv3 --pmc pass over the
contact/penetration kernel and the MANO forward/backward kernels of the adaptation loop."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
import mano as M  # noqa: E402
from dvq_hip import synth  # noqa: E402
from dvq_hip.mano_io import Struct  # noqa: E402
from dvq_hip.tta import GraspRefiner  # noqa: E402

dev = torch.device("cuda:0")
B = 4096
rh = M.MANO('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45, batch_size=1,
            flat_hand_mean=True).to(dev).eval()
go, hp, tr = (t.to(dev) for t in synth.mano_inputs(B, seed=1))
param = torch.zeros(B, 61, device=dev)
param[:, 10:13], param[:, 13:58], param[:, 58:] = go, hp, tr
verts = rh(global_orient=go, hand_pose=hp, transl=tr).vertices
obj = verts.mean(1, keepdim=True) + 0.04 * torch.randn(B, 3000, 3, device=dev)
obj_pc = torch.cat([obj.permute(0, 2, 1), torch.full((B, 1, 3000), 0.2, device=dev)], 1).contiguous()
r = GraspRefiner(rh, use_graph=False)     # eager launches: every dispatch is visible to the counter pass
r.refine(obj_pc, param, iters=6)
torch.cuda.synchronize()
