import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "d-vqvae_amd"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from dvq_hip import synth
from dvq_hip.pipeline import GraspGenerator
dev = torch.device("cuda:0")
g = GraspGenerator.synthetic(dev, seed=0, dtype=torch.bfloat16)
obj = synth.object_clouds_device(4096, 3000, 11, dev)
for _ in range(6):
    g.generate(obj)
torch.cuda.synchronize()
