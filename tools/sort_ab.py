"""A/B of the prior's class-ordered evaluation (4096 grids, bf16): GatedPixelCNN.generate with the rows in arrival order
and in class order, alternating, for one class in the batch and for 128 classes drawn at random; and that both orders
give the same grids."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
import torch
from dvq_hip import ops, synth
from dvq_hip.pipeline import GraspGenerator

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = GraspGenerator.synthetic(dev, seed=0, dtype=torch.bfloat16)
prior = g.model.GatedPixelCNN
u = ops.prior_uniforms(0, B, 9, 0, dev)
gen = torch.Generator(device=dev).manual_seed(5)
for name, label in (("one class", torch.full((B,), 7, dtype=torch.int64, device=dev)),
                    ("128 classes", torch.randint(0, 128, (B,), device=dev, generator=gen))):
    grids = {}
    for flag in (False, True):
        prior.sort_by_class = flag
        for _ in range(3):
            grids[flag] = prior.generate(None, label, shape=(3, 3), batch_size=B, uniforms=u)
    torch.cuda.synchronize()
    print(name, "- grids equal:", torch.equal(grids[False], grids[True]))
    for rep in range(3):
        for flag in (False, True):
            prior.sort_by_class = flag
            prior.generate(None, label, shape=(3, 3), batch_size=B, uniforms=u)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                prior.generate(None, label, shape=(3, 3), batch_size=B, uniforms=u)
            torch.cuda.synchronize()
            print(f"  {name} sort={flag}: {(time.perf_counter() - t0) * 50:.3f} ms per {B} grids", flush=True)
