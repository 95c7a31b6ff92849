"""Workload for a kernel trace of the encode path (DVQVAE.forward, eval): five calls on 4096 hands and objects, bf16."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
import torch
import dvq_hip
from dvq_hip import synth
from network.DVQVAE import DVQVAE

dev = torch.device("cuda:0")
model = DVQVAE()
model.load_state_dict(synth.dvqvae_state_dict(0))
model = model.to(dev).eval()
dvq_hip.set_compute_dtype(model, torch.bfloat16)
B = 4096
obj = synth.object_clouds_device(B, 3000, 7, dev)
hand = torch.randn(B, 3, 778, device=dev) * 0.05
with torch.no_grad():
    for _ in range(5):
        model(obj, hand)
torch.cuda.synchronize()
