"""Time of the fused vertical-row kernel launched alone at the prior's shape (4096 rows, 512 -> 1024 features, two input rows)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd"))
import torch
from dvq_hip import ops
DEV = torch.device("cuda:0")
g = torch.Generator(DEV).manual_seed(6)
d, N, n_cls, M, n_rows = 512, 1024, 128, 4096, 2
label = torch.randint(0, n_cls, (M,), device=DEV, generator=g)
if os.environ.get("LABELS") == "sorted":
    label = label.sort().values
if os.environ.get("LABELS") == "same":
    label = torch.zeros_like(label)
bias = torch.randn(N, device=DEV, generator=g) * 0.1
cond = torch.randn(n_cls, N, device=DEV, generator=g) * 0.1
xs = [[ops.prior_packed_empty(M, d, DEV, True) for _ in range(3)] for _ in range(n_rows)]
tab = torch.randn(50, d, device=DEV, generator=g)
for r in range(n_rows):
    for c in range(3):
        ops.prior_embed(torch.randint(0, 50, (M,), device=DEV, generator=g), tab, xs[r][c])
ws = [[ops.prior_pack_weight(torch.randn(N, d, device=DEV, generator=g) * 0.05) for _ in range(3)] for _ in range(n_rows)]
outs = [ops.prior_packed_empty(M, N // 2, DEV, True) for _ in range(3)]
pres = [ops.prior_packed_empty(M, N, DEV, True) for _ in range(3)]
run = lambda: ops.prior_vrow([(xs[r], ws[r]) for r in range(n_rows)], d // 32, bias, cond, label, outs, pres, M, N)
for k in range(3):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("DVQ_HIP_LIB", "product")), "vrow launch %.2f us" % (e0.elapsed_time(e1) / 50 * 1e3), flush=True)
