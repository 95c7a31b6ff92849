"""Time of the prior's gate product launched alone (the bench's roofline_prior leg), for ablation builds."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from dvq_hip.pipeline import GraspGenerator
dev = torch.device("cuda:0")
g = GraspGenerator.synthetic(dev, seed=0, dtype=torch.bfloat16)
for k in range(3):
    r = bench.prior_roofline(g, 4096, dev, iters=50)
    print(os.environ.get("DVQ_HIP_LIB", "product"), "gate launch %.2f us" % (r["avg_launch_ms"] * 1e3), flush=True)
