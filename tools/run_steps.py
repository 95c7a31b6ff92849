"""Workload for a rocprofv3 --kernel-trace of the generation step: six calls of GraspGenerator.generate on 4096 object
clouds (bf16), no synchronisation in between.  tools/step_timeline.py prints the launches of one step from the trace."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
import torch
from dvq_hip import synth
from dvq_hip.pipeline import GraspGenerator
dev = torch.device("cuda:0")
g = GraspGenerator.synthetic(dev, seed=0, dtype=torch.bfloat16)
obj = synth.object_clouds_device(4096, 3000, 11, dev)
for _ in range(6):
    g.generate(obj)
torch.cuda.synchronize()
