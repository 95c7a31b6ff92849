"""Host time against device time of the generation step (4096 clouds, bf16): how long GraspGenerator.generate keeps the
host (no synchronisation) and how long the device needs per step; then the same step recorded once as ONE HIP graph
(the prior's launches issued into the recording instead of replaying its own graph) and replayed."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd"))
import torch
from dvq_hip import synth
from dvq_hip.pipeline import GraspGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
g = GraspGenerator.synthetic(dev, seed=0, dtype=torch.bfloat16)
obj = synth.object_clouds_device(B, 3000, 11, dev)
for _ in range(3):
    ref = g.generate(obj)
torch.cuda.synchronize()
K = 10
host = []
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter()
    g.generate(obj)
    host.append((time.perf_counter() - a) * 1e3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager+prior graph: host ms per call {sorted(host)[K // 2]:.2f} (min {min(host):.2f} max {max(host):.2f}), "
      f"issue loop {1e3 * (t1 - t0) / K:.2f} ms/step, device {1e3 * (t2 - t0) / K:.2f} ms/step", flush=True)

prior = g.model.GatedPixelCNN
prior.use_graph = False
for _ in range(2):
    g.generate(obj)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    g.generate(obj)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager, no graph: issue loop {1e3 * (t1 - t0) / K:.2f} ms/step, device {1e3 * (t2 - t0) / K:.2f} ms/step", flush=True)

graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    out = g.generate(obj)
torch.cuda.synchronize()
for _ in range(2):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    graph.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"whole-step graph: issue loop {1e3 * (t1 - t0) / K:.2f} ms/step, device {1e3 * (t2 - t0) / K:.2f} ms/step", flush=True)
print("vertices equal:", torch.equal(out["vertices"], ref["vertices"]), "params equal:", torch.equal(out["recon_param"], ref["recon_param"]))
