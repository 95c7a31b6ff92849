"""The CPU oracle replayed against vectors captured from the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then
compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from dvq_hip import synth
from oracle import dvq_oracle as O
from parity import check_vq_rows

VQ_CASES = [(128, 256), (128, 1024), (512, 256), (2048, 256), (8192, 256)]


@pytest.mark.parametrize("K,D", VQ_CASES)
@pytest.mark.parametrize("dist", ["A", "B"])
def test_vq_oracle_vs_reference(K, D, dist):
    g = load_golden("vq")
    z, E = synth.vq_case(K, D, 4096, dist, seed=0)
    idx, zq = O.vq_forward(z, E)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (4096, 1)
    assert torch.equal(zq, E[idx[:, 0]])
    # same ops, same library, same machine class: expected identical, near-tie rule as the bound
    check_vq_rows(z, E, idx[:, 0].numpy(), g[f"idx_{K}_{D}_{dist}"])


def test_vq_ragged_and_ties():
    g = load_golden("vq")
    z, E = synth.vq_case(128, 256, 37, "A", seed=3)
    E[5] = E[77]
    z[0] = E[77]
    idx, _ = O.vq_forward(z, E)
    assert idx[0, 0].item() == 5  # identical codes: lowest index wins
    assert np.array_equal(idx[:, 0].numpy(), g["idx_ragged37"])
    assert np.array_equal(O.vq_get_emb(torch.tensor([9]), E).numpy(), g["get_emb_9"])


VQ_TRAIN_CASES = [(128, 256, 96), (128, 1024, 24)]     # as in tests/golden/make_golden.py


def vq_train_case(K, D, n):
    z, E = synth.vq_case(K, D, n, "B", seed=5)
    w = torch.randn(n, D, generator=torch.Generator().manual_seed(11))
    return z, E, w


@pytest.mark.parametrize("K,D,n", VQ_TRAIN_CASES)
def test_vq_training_mode_vs_reference_golden(K, D, n):
    """quantizer.py:35-43, 56-67 through VQVAE.forward (VQVAE.py:29-42): loss, straight-through z_q, perplexity,
    indices, and the gradients of loss + <z_q, w> with respect to latents and codebook."""
    g = load_golden("vq_train")
    z, E, w = vq_train_case(K, D, n)
    zr, Er = z.clone().requires_grad_(True), E.clone().requires_grad_(True)
    loss, zq, perp, onehot, idx = O.vq_forward_train(zr, Er, 0.25, 1)
    (loss + (zq * w).sum()).backward()
    assert np.array_equal(idx[:, 0].numpy(), g[f"idx_{K}_{D}"])
    assert np.array_equal(onehot.argmax(1).numpy(), g[f"idx_{K}_{D}"]) and float(onehot.sum()) == n
    for name, got in (("loss", loss), ("zq", zq), ("perplexity", perp), ("dz", zr.grad), ("dE", Er.grad)):
        assert np.array_equal(got.detach().numpy(), g[f"{name}_{K}_{D}"]), name


@pytest.mark.parametrize("K,D", [(128, 256), (512, 256)])
@pytest.mark.parametrize("kind", ["mid", "bignorm"])
def test_vq_oracle_vs_reference_on_near_ties(K, D, kind):
    """Latents whose two best codes are nearly equidistant, so that the rounding order of quantizer.py:46-48 decides
    (the reference itself leaves the exact minimum on some of these rows).  The fixture must not be vacuous."""
    g = load_golden("vq_near")
    z, E = synth.vq_near_tie_case(K, D, 4096, kind, seed=0)
    assert int(g[f"near_{K}_{D}_{kind}"]) > 0 and int(g[f"ref_off_exact_{K}_{D}_{kind}"]) > 0
    idx, _ = O.vq_forward(z, E)
    n_diff = check_vq_rows(z, E, idx[:, 0].numpy(), g[f"idx_{K}_{D}_{kind}"])
    print(f"K={K} {kind}: {int(g[f'near_{K}_{D}_{kind}'])} near-tie rows, oracle differs from the reference on {n_diff}")


def test_pointnet_oracle_vs_reference():
    g = load_golden("pointnet")
    sd = synth.gennet_state_dict(0, with_prior=False)
    x4 = torch.cat([torch.from_numpy(g["real_clouds"]), synth.object_clouds(4, 3000, seed=7)], 0)
    feat, trans = O.pointnet_encoder(x4, sd, "obj_encoder_type.")
    np.testing.assert_allclose(trans.numpy(), g["trans4"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(feat.numpy(), g["feat4"], rtol=1e-5, atol=1e-5)
    for xk, fk, tk in (("x3", "feat3", "trans3"), ("xs", "feats", "transs")):
        feat, trans = O.pointnet_encoder(torch.from_numpy(g[xk]), sd, "recon_encoder.")
        np.testing.assert_allclose(trans.numpy(), g[tk], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(feat.numpy(), g[fk], rtol=1e-5, atol=1e-5)


def test_mlp_oracle_vs_reference():
    g = load_golden("mlp")
    sd = synth.gennet_state_dict(0, with_prior=False)
    dsd = synth.dvqvae_state_dict(0)
    np.testing.assert_allclose(O.decoder_mlp(torch.from_numpy(g["dec_in"]), sd, "decoder.").numpy(), g["dec_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.decoder_mlp(torch.from_numpy(g["pos_in"]), sd, "pos_decoder.").numpy(), g["pos_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.part_latent_mlp(torch.from_numpy(g["emb_in"]), dsd, "emb_0.").numpy(), g["emb_out"], rtol=1e-5, atol=1e-6)


def test_mano_oracle_vs_reference():
    g = load_golden("mano")
    sd = synth.mano_state_dict(0, batch_size=8)
    v, j, fp = O.mano_forward(sd, torch.from_numpy(g["global_orient"]), torch.from_numpy(g["hand_pose"]),
                              torch.from_numpy(g["transl"]))
    assert v.shape == (8, 778, 3) and j.shape == (8, 16, 3)
    np.testing.assert_allclose(fp.numpy(), g["full_pose"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(v.numpy(), g["vertices"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(j.numpy(), g["joints"], rtol=1e-5, atol=1e-6)


def test_mano_oracle_vs_reference_on_real_grasp_parameters():
    """120 parameter vectors the trained reference model produced (its stored HMDO outputs) + a rest pose and
    two large-angle rows, through the reference MANO class."""
    g = load_golden("mano_hmdo")
    p = torch.from_numpy(g["recon_params"])
    assert p.shape == (123, 61)
    sd = synth.mano_state_dict(0, batch_size=1)
    v, j, _ = O.mano_forward(sd, p[:, 10:13], p[:, 13:58], p[:, 58:61])
    np.testing.assert_allclose(j.numpy(), g["joints"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(v[::2].numpy(), g["vertices_even"], rtol=1e-5, atol=2e-6)


def test_dvqvae_eval_oracle_vs_reference():
    g = load_golden("dvqvae")
    sd = synth.dvqvae_state_dict(0)
    f0 = g["f0hand"].tolist()
    from dvq_hip.hand_parts import HAND_PARTS
    assert HAND_PARTS[0] == f0
    objs = synth.object_clouds(2, 3000, seed=99)
    emb_idx, obj_emb = O.dvqvae_eval(sd, objs, torch.from_numpy(g["hand_xyz"]), HAND_PARTS)
    assert emb_idx.shape == (14, 1) and emb_idx.dtype == torch.int64
    assert np.array_equal(emb_idx.numpy(), g["emb_idx"])
    np.testing.assert_allclose(obj_emb.numpy(), g["obj_emb"], rtol=0, atol=0)


def test_gen_oracle_vs_reference():
    g = load_golden("gen")
    sd = synth.gennet_state_dict(0, with_prior=True)
    sd.update(synth.mano_state_dict(0, batch_size=1, prefix="rh_mano."))
    n = g["recon"].shape[0]
    objs = synth.object_clouds(n, 3000, seed=1234)
    out = O.generate_grasps(sd, objs, x_tilde=torch.from_numpy(g["x_tilde"]))
    assert np.array_equal(out["idx6"][:, 0].numpy(), g["idx6"])
    np.testing.assert_allclose(out["recon"].numpy(), g["recon"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["recon_pos"].numpy(), g["recon_pos"], rtol=1e-4, atol=1e-5)
    ref_v = g["vertices"]
    err = np.abs(out["vertices"].numpy() - ref_v).max() / np.abs(ref_v).max()
    assert err <= 1e-5, err
    # prior logits (first 128 classes) on the teacher-forced grids
    lg = O.pixelcnn_logits(sd, torch.from_numpy(g["x_tilde"][:2]), torch.from_numpy(g["idx6"][:2]))
    np.testing.assert_allclose(lg[:, :128].numpy(), g["prior_logits"], rtol=1e-4, atol=1e-4)
