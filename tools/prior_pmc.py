"""Workload for a rocprofv3 --pmc pass over the prior's kernels: one GatedPixelCNN.generate of 4096 grids (bf16) issued
launch by launch (no graph).  tools/pmc_sum.py averages the counters per kernel and grid size."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd"))
import torch
from dvq_hip import synth, rng
from network.gen_net import GenNet
dev = torch.device("cuda:0")
model = GenNet()
model.load_state_dict(synth.gennet_state_dict(0))
pr = model.GatedPixelCNN.to(dev).eval()
pr.compute_dtype = torch.bfloat16
pr.use_graph = False
B = 4096
lab = torch.randint(0, 128, (B,), device=dev)
u = rng.uniforms(torch.arange(B, device=dev), 9, 0)
pr.generate(None, lab, shape=(3, 3), batch_size=B, uniforms=u)
torch.cuda.synchronize()
