"""The configuration the benchmark times -- 4096 object clouds x 3000 points through GraspGenerator.generate, fp32 and
bf16 -- under parity: the branches only large batches take (the fused vertical-row prior kernel from 3072 rows, one
launch per encoder stage from 64, bf16 heads from 512, HIP-graph replay of the prior at that batch size) run
together here and are checked against the CPU oracle and against small-batch runs of the same objects.
Reference path: DVQ-VAE/network/gen_net.py:78-125 + gen_diverse_grasp_obman.py:243-253."""
import pytest
import torch

import dvq_hip
from dvq_hip import rng, synth
from dvq_hip.pipeline import GraspGenerator
from oracle import dvq_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
B, N = 4096, 3000
BF16_VERTEX_TOL_M = 2e-3           # stated bf16 tolerance: 2 mm on a ~0.2 m hand
BF16_CELL_AGREEMENT = 0.92         # stated: share of sampled grid cells equal between the batch-4096 and the small-batch
                                   # kernels (another summation order moves bf16 activations by an ulp; same gate as
                                   # bf16 against fp32 in test_gpu_gen.py)


@pytest.fixture(scope="module")
def setup():
    gen = GraspGenerator.synthetic(DEV, seed=0, dtype=torch.float32, with_prior=True)
    obj = synth.object_clouds_device(B, N, 4321, DEV)
    sd = synth.gennet_state_dict(0, with_prior=True)
    sd.update(synth.mano_state_dict(0, batch_size=1, prefix="rh_mano."))
    return gen, obj, sd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_batch_teacher_forced_rows_vs_oracle(setup, dtype):
    gen, obj, sd = setup
    dvq_hip.set_compute_dtype(gen.model, dtype)
    xt = torch.randint(0, 128, (B, 3, 3), device=DEV, generator=torch.Generator(DEV).manual_seed(7))
    out = gen.generate(obj, x_tilde=xt)
    assert out["vertices"].shape == (B, 778, 3) and out["recon_param"].shape == (B, 61)
    assert torch.isfinite(out["vertices"]).all()
    rows = torch.cat([torch.arange(0, 8), torch.arange(B - 8, B)])
    ref = O.generate_grasps(sd, obj[rows.to(DEV)].cpu(), x_tilde=xt[rows.to(DEV)].cpu())
    v, vr = out["vertices"][rows.to(DEV)].cpu(), ref["vertices"]
    if dtype == torch.float32:
        err = float((v - vr).abs().max() / vr.abs().max())
        print(f"fp32, B={B}: vertex rel err vs oracle {err:.2e}")
        assert err <= 1e-4
        assert torch.equal(gen.model.last_idx6[rows.to(DEV)].cpu(), ref["idx6"])
    else:
        err = float((v - vr).abs().max())
        agree = float((gen.model.last_idx6[rows.to(DEV)].cpu() == ref["idx6"]).float().mean())
        print(f"bf16, B={B}: max vertex deviation vs oracle {err * 1e3:.3f} mm, object-code agreement {agree:.3f}")
        assert err <= BF16_VERTEX_TOL_M


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_batch_sampled_equals_small_batches(setup, dtype):
    """generate(obj)[s] against generate(obj[s]) for slices of fewer than 64 objects: the large batch takes the fused
    row kernel, ungrouped encoder launches, the bf16 heads and the graph recorded for 4096 rows; the small ones take per-cell
    launches, grouped encoders and fp32 heads.  Uniforms are keyed by global object id, so the draws are the same."""
    gen, obj, sd = setup
    dvq_hip.set_compute_dtype(gen.model, dtype)
    big = gen.generate(obj)
    grid_big, idx6_big = gen.model.last_x_tilde.clone(), gen.model.last_idx6.clone()
    big2 = gen.generate(obj)                                   # second call = graph replay; must repeat itself
    assert torch.equal(big2["vertices"], big["vertices"]) and torch.equal(gen.model.last_x_tilde, grid_big)
    assert int(grid_big.max()) < 128 and int(grid_big.min()) >= 0
    n_rows = n_same = n_cells_same = 0
    worst = 0.0
    for a, b in ((0, 40), (2000, 2063), (B - 40, B)):
        small = gen.generate(obj[a:b].contiguous(), first_id=a)
        grid_small = gen.model.last_x_tilde
        assert torch.equal(gen.model.last_idx6, idx6_big[a:b])        # the encoder arithmetic does not depend on the batch
        same = (grid_small == grid_big[a:b]).flatten(1).all(1)
        n_rows += b - a
        n_same += int(same.sum())
        n_cells_same += int((grid_small == grid_big[a:b]).sum())
        if same.any():
            d = (small["vertices"][same] - big["vertices"][a:b][same]).abs().max()
            worst = max(worst, float(d / big["vertices"][a:b][same].abs().max()) if dtype == torch.float32 else float(d))
    print(f"{dtype}: {n_same}/{n_rows} grids and {n_cells_same}/{9 * n_rows} cells equal; worst vertex deviation on equal grids "
          f"{worst:.2e}{' (relative)' if dtype == torch.float32 else ' m'}")
    if dtype == torch.float32:
        assert n_same == n_rows and worst <= 1e-4
    else:
        assert n_cells_same >= BF16_CELL_AGREEMENT * 9 * n_rows and n_same > 0 and worst <= BF16_VERTEX_TOL_M


def test_full_batch_sampled_fp32_first_rows_vs_oracle_prior(setup):
    """fp32, sampled: the grids the device drew for the first rows, fed to the oracle as forced samples, must
    reproduce the device's vertices, and each drawn index must be the inverse-CDF draw of the ORACLE's distribution
    at the same uniform (up to uniforms that fall within 1e-4 of a CDF step)."""
    gen, obj, sd = setup
    dvq_hip.set_compute_dtype(gen.model, torch.float32)
    out = gen.generate(obj)
    k = 6
    grid = gen.model.last_x_tilde[:k].cpu()
    ref = O.generate_grasps(sd, obj[:k].cpu(), x_tilde=grid)
    err = float((out["vertices"][:k].cpu() - ref["vertices"]).abs().max() / ref["vertices"].abs().max())
    assert err <= 1e-4
    _, probs = O.pixelcnn_generate(sd, ref["idx6"][:, 0], (3, 3), forced=grid)
    u = rng.uniforms(torch.arange(k), 9, gen.seed).view(k, 3, 3)
    cdf = probs.double().cumsum(-1)
    lo = torch.gather(cdf, -1, grid.unsqueeze(-1)).squeeze(-1) - torch.gather(probs.double(), -1, grid.unsqueeze(-1)).squeeze(-1)
    hi = torch.gather(cdf, -1, grid.unsqueeze(-1)).squeeze(-1)
    assert bool(((u >= lo - 1e-4) & (u < hi + 1e-4)).all())
