"""MANO, linear and PointNet kernels vs the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from dvq_hip import ops, synth
from dvq_hip.mano_io import Struct
from oracle import dvq_oracle as O
from parity import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_mano(batch_size=1):
    import mano
    return mano.MANO('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45,
                     batch_size=batch_size, flat_hand_mean=True).to(DEV).eval()


def test_mano_vs_reference_golden():
    g = load_golden("mano")
    rh = make_mano(8)
    go, hp, tr = (torch.from_numpy(g[k]).to(DEV) for k in ("global_orient", "hand_pose", "transl"))
    out = rh(betas=torch.full((8, 10), 5.0, device=DEV), global_orient=go, hand_pose=hp, transl=tr, return_full_pose=True)
    assert out.vertices.shape == (8, 778, 3) and out.joints.shape == (8, 16, 3)
    # tolerance of the task: 1e-4 relative on vertices in fp32
    assert rel_err(out.vertices, g["vertices"]) <= 1e-5
    assert rel_err(out.joints, g["joints"]) <= 1e-5
    assert rel_err(out.full_pose, g["full_pose"]) <= 1e-6


def test_mano_vs_reference_golden_on_real_grasp_parameters():
    """The reference MANO class on the 120 parameter vectors its trained model produced (DVQ-VAE-2's stored HMDO
    outputs; PCA coefficients up to +-7.6) + rest pose + large angles: joints of every row, vertices of every second."""
    g = load_golden("mano_hmdo")
    p = torch.from_numpy(g["recon_params"]).to(DEV)
    rh = make_mano(1)
    out = rh(betas=p[:, :10].contiguous(), global_orient=p[:, 10:13].contiguous(), hand_pose=p[:, 13:58].contiguous(),
             transl=p[:, 58:].contiguous())
    ev = rel_err(out.vertices[::2], g["vertices_even"])
    ej = rel_err(out.joints, g["joints"])
    print(f"MANO on real grasp parameters: vertices {ev:.2e}, joints {ej:.2e} (relative to max |.|)")
    assert ev <= 1e-5 and ej <= 1e-5


def test_mano_skinning_table_equals_dense_weights():
    """The kernel gathers a vertex's transform from a table of its non-zero skinning weights (joints ascending); with
    the table absent it sums all 16 dense weights as the reference does.  The skipped terms are exact zeros: same bits."""
    import ctypes
    from dvq_hip._lib import DvqManoConsts
    rh = make_mano(1)
    go, hp, tr = (t.to(DEV) for t in synth.mano_inputs(37, seed=4))
    c = rh._device_consts()
    assert c.skin_joints is not None and c.skin_width == 8
    dense = DvqManoConsts()
    ctypes.memmove(ctypes.byref(dense), ctypes.byref(c), ctypes.sizeof(c))
    dense.skin_joints, dense.skin_weights = None, None
    a = ops.mano_lbs(go, hp, tr, c, 778, want_v_posed=True)
    b = ops.mano_lbs(go, hp, tr, dense, 778, want_v_posed=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    # the lane-ordered copy of posedirs holds the same numbers: with it absent the kernel reads posedirs itself
    assert c.posedirs_lanes is not None
    plain = DvqManoConsts()
    ctypes.memmove(ctypes.byref(plain), ctypes.byref(c), ctypes.sizeof(c))
    plain.posedirs_lanes = None
    d = ops.mano_lbs(go, hp, tr, plain, 778, want_v_posed=True)
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1]) and torch.equal(a[3], d[3])



@pytest.mark.parametrize("V", [778, 37, 16, 5])
def test_mano_fewer_vertices_equal_the_first_vertices_of_the_full_model(V):
    """A vertex depends on its own row of every table only: with the tables cut to the first V vertices (groups of 16
    that end ragged, one group, less than one) the kernel must return the full model's first V vertices, bit for bit,
    in the table form with and without the lane-ordered posedirs and in the dense form."""
    import ctypes
    from dvq_hip._lib import DvqManoConsts
    from mano.model import lane_ordered_posedirs
    rh = make_mano(1)
    go, hp, tr = (t.to(DEV) for t in synth.mano_inputs(21, seed=9))
    c = rh._device_consts()
    keep = rh._consts[1]
    full = ops.mano_lbs(go, hp, tr, c, 778, want_v_posed=True)
    pd = keep["posedirs"].reshape(135, 778, 3)[:, :V].reshape(135, V * 3).contiguous()
    cut = dict(v_shaped=keep["v_shaped"][:V].contiguous(), posedirs=pd, weights=keep["weights"][:V].contiguous(),
               skin_joints=keep["skin_joints"][:V].contiguous(), skin_weights=keep["skin_weights"][:V].contiguous(),
               posedirs_lanes=lane_ordered_posedirs(pd, V))
    small = DvqManoConsts()
    ctypes.memmove(ctypes.byref(small), ctypes.byref(c), ctypes.sizeof(c))
    small.n_verts = V
    small.v_shaped, small.posedirs, small.lbs_weights = cut["v_shaped"].data_ptr(), pd.data_ptr(), cut["weights"].data_ptr()
    small.skin_joints, small.skin_weights = cut["skin_joints"].data_ptr(), cut["skin_weights"].data_ptr()
    small.posedirs_lanes = cut["posedirs_lanes"].data_ptr()
    for form in ("lanes", "table", "dense"):
        if form == "table":
            small.posedirs_lanes = None
        if form == "dense":
            small.skin_joints, small.skin_weights = None, None
        out = ops.mano_lbs(go, hp, tr, small, V, want_v_posed=True)
        assert torch.equal(out[0], full[0][:, :V]), (V, form)
        assert torch.equal(out[3], full[3][:, :V]), (V, form)
        assert torch.equal(out[1], full[1]), (V, form)


@pytest.mark.parametrize("B", [1, 3, 8, 9, 67])
def test_mano_batches_vs_oracle(B):
    rh = make_mano(1)
    go, hp, tr = synth.mano_inputs(B, seed=B)
    sd = synth.mano_state_dict(0, batch_size=1)
    v_ref, j_ref, _ = O.mano_forward(sd, go, hp, tr)
    out = rh(global_orient=go.to(DEV), hand_pose=hp.to(DEV), transl=tr.to(DEV))
    assert rel_err(out.vertices, v_ref) <= 1e-5 and rel_err(out.joints, j_ref) <= 1e-5
    out2 = rh(global_orient=go.to(DEV), hand_pose=hp.to(DEV), transl=None)   # module's own zero transl
    v0, _, _ = O.mano_forward(sd, go, hp, torch.zeros(B, 3))
    assert rel_err(out2.vertices, v0) <= 1e-5


def test_mano_full_size_linearity_in_translation():
    """B = 4096: vertices(transl=t) - vertices(transl=0) must be t for every vertex, and a
    slice of the batch must equal the same rows computed alone."""
    B = 4096
    rh = make_mano(1)
    go, hp, tr = (t.to(DEV) for t in synth.mano_inputs(B, seed=1))
    v = rh(global_orient=go, hand_pose=hp, transl=tr).vertices
    v0 = rh(global_orient=go, hand_pose=hp, transl=torch.zeros_like(tr)).vertices
    assert (v - v0 - tr[:, None, :]).abs().max().item() < 1e-6
    part = rh(global_orient=go[1000:1013].contiguous(), hand_pose=hp[1000:1013].contiguous(), transl=tr[1000:1013].contiguous()).vertices
    assert torch.equal(part, v[1000:1013])


@pytest.mark.parametrize("M,K,N,relu", [(1, 1024, 512, True), (5, 2560, 1024, True), (16, 256, 9, False),
                                        (17, 256, 55, False), (64, 2048, 1024, True), (300, 1024, 128, True),
                                        (4096, 128, 6, False)])
def test_linear_vs_torch(M, K, N, relu):
    r = np.random.default_rng(M + K)
    x = torch.from_numpy(r.normal(0, 1, (M, K)).astype(np.float32))
    W = torch.from_numpy(r.normal(0, K ** -0.5, (N, K)).astype(np.float32))
    b = torch.from_numpy(r.normal(0, 0.1, (N,)).astype(np.float32))
    ref = torch.nn.functional.linear(x.double(), W.double(), b.double())
    if relu:
        ref = ref.relu()
    y = ops.linear(x.to(DEV), W.to(DEV), b.to(DEV), relu=relu)
    assert (y.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    if N == 9:
        y2 = ops.linear(x.to(DEV), W.to(DEV), b.to(DEV), add_identity3=True)
        eye = torch.eye(3).reshape(1, 9)
        assert (y2.cpu().double() - (ref + eye)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("M,K,N,relu", [(4096, 2560, 1024, True), (600, 1024, 512, True), (513, 64, 64, False),
                                        (1000, 2048, 200, False)])
def test_linear_bf16_operands_vs_fp64_on_rounded_inputs(M, K, N, relu):
    """The bf16 head kernel: operands rounded to bf16 (x on the fly, W beforehand), products exact in fp32,
    fp32 accumulation.  The fp64 product of the ROUNDED operands is the reference; what is left is the order of the
    fp32 accumulation.  Ragged M and N (partial tiles), a strided input and a strided output are covered; batches
    below ops.LINEAR_BF16_MIN_ROWS must take the fp32 kernel even when the bf16 weight is offered."""
    r = np.random.default_rng(M + K)
    x = torch.from_numpy(r.normal(0, 1, (M, K + 8)).astype(np.float32)).to(DEV)[:, 4:4 + K]
    W = torch.from_numpy(r.normal(0, K ** -0.5, (N, K)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(r.normal(0, 0.1, (N,)).astype(np.float32)).to(DEV)
    Wb = ops.to_bf16(W)
    assert torch.equal(Wb, W.to(torch.bfloat16))
    wide = torch.zeros(M, N + 16, device=DEV)
    y = ops.linear(x, W, b, relu=relu, out=wide[:, 8:8 + N], W_bf16=Wb)
    ref = torch.nn.functional.linear(x.to(torch.bfloat16).double(), Wb.double(), b.double())
    if relu:
        ref = ref.relu()
    assert (y.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert wide[:, :8].abs().sum() == 0 and wide[:, 8 + N:].abs().sum() == 0
    small = ops.linear(x[:100], W, b, relu=relu, W_bf16=Wb)
    assert torch.equal(small, ops.linear(x[:100], W, b, relu=relu))


def test_mlp_heads_vs_reference_golden():
    from network.DVQVAE import Decoder, Encoder
    g = load_golden("mlp")
    sd = synth.gennet_state_dict(0, with_prior=False)
    dsd = synth.dvqvae_state_dict(0)
    dec = Decoder([1024, 256, 55], 2560)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    pdec = Decoder([1024, 128, 6], 2048)
    pdec.load_state_dict({k[len("pos_decoder."):]: v for k, v in sd.items() if k.startswith("pos_decoder.")})
    emb = Encoder([1024, 512], 256)
    emb.load_state_dict({k[len("emb_0."):]: v for k, v in dsd.items() if k.startswith("emb_0.")})
    for mod, i, o in ((dec, "dec_in", "dec_out"), (pdec, "pos_in", "pos_out"), (emb, "emb_in", "emb_out")):
        y = mod.to(DEV)(torch.from_numpy(g[i]).to(DEV))
        np.testing.assert_allclose(y.cpu().numpy(), g[o], rtol=1e-4, atol=1e-5)


def _encoder(prefix, channel, dtype=torch.float32):
    from network.pointnet_encoder import PointNetEncoder
    sd = synth.gennet_state_dict(0, with_prior=False)
    enc = PointNetEncoder(global_feat=True, feature_transform=False, channel=channel)
    enc.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    enc.compute_dtype = dtype
    return enc.to(DEV).eval()


def test_pointnet_f32_vs_reference_golden():
    g = load_golden("pointnet")
    enc = _encoder("obj_encoder_type.", 4)
    x4 = torch.cat([torch.from_numpy(g["real_clouds"]), synth.object_clouds(4, 3000, seed=7)], 0)
    feat, trans, tf = enc(x4.to(DEV))
    assert tf is None and feat.shape == (8, 1024) and trans.shape == (8, 3, 3)
    assert rel_err(trans, g["trans4"]) <= 1e-5
    assert rel_err(feat, g["feat4"]) <= 1e-5
    enc3 = _encoder("recon_encoder.", 3)
    for xk, fk, tk in (("x3", "feat3", "trans3"), ("xs", "feats", "transs")):   # N = 778 and N = 83 (ragged chunk)
        feat, trans, _ = enc3(torch.from_numpy(g[xk]).to(DEV))
        assert rel_err(trans, g[tk]) <= 1e-5 and rel_err(feat, g[fk]) <= 1e-5


def test_pointnet_bf16_tolerance():
    """bf16 operands for the 64->128->1024 layers, fp32 accumulation.  Stated tolerance:
    3e-2 of the feature range on the global feature, 2e-2 on the 3x3 transform."""
    g = load_golden("pointnet")
    enc = _encoder("obj_encoder_type.", 4, torch.bfloat16)
    x4 = torch.cat([torch.from_numpy(g["real_clouds"]), synth.object_clouds(4, 3000, seed=7)], 0)
    feat, trans, _ = enc(x4.to(DEV))
    e_t, e_f = rel_err(trans, g["trans4"]), rel_err(feat, g["feat4"])
    print(f"bf16 encoder: trans rel err {e_t:.3e}, feat rel err {e_f:.3e}")
    assert e_t <= 2e-2 and e_f <= 3e-2


@pytest.mark.parametrize("N,B", [(1, 3), (31, 3), (83, 5), (123, 7), (128, 4), (129, 6), (212, 5), (256, 2), (257, 3),
                                 (1024, 3), (1025, 3), (3000, 3)])
def test_pointnet_ragged_point_counts(N, B):
    """Point counts around every chunk and tile boundary.  Clouds of up to 256 points share a workgroup (two or
    four clouds per block), so the batch sizes include ones that leave a block partly empty."""
    sd = synth.gennet_state_dict(0, with_prior=False)
    enc = _encoder("obj_encoder_pos.", 4)
    x = synth.object_clouds(B, 3000, seed=3)[:, :, :N].contiguous()
    ref_f, ref_t = O.pointnet_encoder(x, sd, "obj_encoder_pos.")
    feat, trans, _ = enc(x.to(DEV))
    assert rel_err(trans, ref_t) <= 1e-5 and rel_err(feat, ref_f) <= 1e-5


def test_pointnet_permutation_and_duplication_invariance():
    """max-pool properties at full size: permuting or duplicating points of a cloud leaves
    the global feature bit-identical (the STN sees the same set too)."""
    enc = _encoder("obj_encoder_type.", 4)
    x = synth.object_clouds(64, 3000, seed=21).to(DEV)
    f0, t0, _ = enc(x)
    perm = torch.randperm(3000, device=DEV, generator=torch.Generator(DEV).manual_seed(0))
    f1, t1, _ = enc(x[:, :, perm].contiguous())
    assert torch.equal(f0, f1) and torch.equal(t0, t1)
    f2, _, _ = enc(torch.cat([x, x[:, :, :500]], dim=2).contiguous())
    assert torch.equal(f0, f2)


def test_head_mlp_bf16_pieces_equal_the_concatenated_input():
    """The two wide layers of an MLP head on packed bf16 activations (ops.head_mlp_bf16): input pieces given as matrices
    or as codebook rows picked by index, against the same layers on the concatenated input -- an fp64 evaluation of the
    bf16-rounded operands (accumulation order is all that may differ) and the register-operand bf16 kernel."""
    g = torch.Generator(DEV).manual_seed(5)
    for M in (512, 1000):                      # 1000: not a multiple of the 32-row tile
        table = torch.randn(128, 256, device=DEV, generator=g)
        idx = torch.randint(0, 128, (M,), device=DEV, generator=g)
        a = torch.randn(M, 256, device=DEV, generator=g)
        b = torch.randn(M, 1024, device=DEV, generator=g)
        W0 = torch.randn(512, 1536, device=DEV, generator=g) * 0.03
        b0 = torch.randn(512, device=DEV, generator=g) * 0.1
        W1 = torch.randn(128, 512, device=DEV, generator=g) * 0.05
        b1 = torch.randn(128, device=DEV, generator=g) * 0.1
        assert ops.head_shapes_ok(M, [256, 256, 1024], 512, 128) and not ops.head_shapes_ok(64, [256], 512, 128)
        packed = ops.head_pack(W0, [256, 256, 1024], W1)
        got = ops.head_mlp_bf16([(idx, table), a, b], packed, b0, b1)
        x = torch.cat([table[idx], a, b], 1)
        r = lambda t: t.to(torch.bfloat16).double()
        h = torch.relu(r(x) @ r(W0).t() + b0.double()).float()
        want = torch.relu(r(h) @ r(W1).t() + b1.double())
        scale = want.abs().max().item()
        # a hidden value on a bf16 rounding boundary may round the other way under another summation order
        assert (got.double() - want).abs().max().item() <= 4e-3 * scale
        old = ops.linear(ops.linear(x, W0, b0, relu=True, W_bf16=ops.to_bf16(W0)), W1, b1, relu=True, W_bf16=ops.to_bf16(W1))
        assert (got - old).abs().max().item() <= 4e-3 * scale


# ------------------------------------------------------------------ kernels around the step (no framework launches)
@pytest.mark.parametrize("B", [1, 5, 4096])
def test_mano_reads_column_ranges_of_the_parameter_vector_in_place(B):
    """The call sites slice global_orient / hand_pose / transl out of one [B,61] vector
    (gen_diverse_grasp_obman.py:243-253): the layer reads those slices in place and gives the same bits as on copies;
    hip_vertices (zero orientation and translation without zero tensors) equals forward with explicit zeros."""
    rh = make_mano()
    go, hp, tr = synth.mano_inputs(B, seed=31)
    param = torch.zeros(B, 61)
    param[:, 10:13], param[:, 13:58], param[:, 58:] = go, hp, tr
    param = param.to(DEV)
    a = rh(betas=param[:, :10], global_orient=param[:, 10:13], hand_pose=param[:, 13:58], transl=param[:, 58:])
    b = rh(global_orient=go.to(DEV), hand_pose=hp.to(DEV), transl=tr.to(DEV))
    assert torch.equal(a.vertices, b.vertices) and torch.equal(a.joints, b.joints)
    z = torch.zeros(B, 3, device=DEV)
    assert torch.equal(rh.hip_vertices(param[:, 13:58]), rh(global_orient=z, hand_pose=hp.to(DEV), transl=z).vertices)
    assert torch.equal(rh.hip_vertices(param[:, 13:58], param[:, 10:13], param[:, 58:]), b.vertices)
    # a broadcast row (stride 0) is not a column range: it is copied, not read past its end
    one = torch.zeros(1, 3, device=DEV)
    assert torch.equal(rh(global_orient=one.expand(B, 3), hand_pose=hp.to(DEV), transl=one.expand(B, 3)).vertices,
                       rh(global_orient=z, hand_pose=hp.to(DEV), transl=z).vertices)
    if B > 1:
        with pytest.raises(RuntimeError):
            ops.mano_lbs(one.expand(B, 3), hp.to(DEV), None, rh._device_consts(), 778)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,N", [(3, 778), (3, 83), (4, 3000)])
def test_pointnet_reads_point_major_clouds_in_place(dtype, C, N):
    """A [B,N,C] buffer seen as [B,C,N] (vertices.permute(0, 2, 1), gen_net.py:120) gives the bits of its
    channel-major copy, transform net included."""
    from network.pointnet_encoder import PointNetEncoder
    sd = synth.gennet_state_dict(0, with_prior=False)
    pre = "recon_encoder." if C == 3 else "obj_encoder_pos."
    enc = PointNetEncoder(channel=C)
    enc.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
    enc = enc.to(DEV).eval()
    enc.compute_dtype = dtype
    pts = (0.05 * torch.randn(37, N, C, generator=torch.Generator().manual_seed(N))).to(DEV)      # [B,N,C]
    view = pts.permute(0, 2, 1)
    assert ops.is_point_major(view) and not ops.is_point_major(view.contiguous())
    f0, t0, _ = enc(view.contiguous())
    f1, t1, _ = enc(view)
    assert torch.equal(f0, f1) and torch.equal(t0, t1)


def test_assemble_param_and_class_order_kernels():
    g = torch.Generator().manual_seed(5)
    recon, pos = torch.randn(1000, 55, generator=g).to(DEV), torch.randn(1000, 6, generator=g).to(DEV)
    want = torch.cat([recon[:, 0:10], pos[:, 0:3], recon[:, 10:55], pos[:, 3:6]], 1)
    assert torch.equal(ops.assemble_param(recon, pos), want)
    wide = torch.randn(1000, 80, generator=g).to(DEV)
    assert torch.equal(ops.assemble_param(wide[:, 7:62], pos), torch.cat([wide[:, 7:17], pos[:, 0:3], wide[:, 17:62], pos[:, 3:6]], 1))
    for B, n_classes in ((1, 128), (63, 128), (64, 5), (4096, 128), (5000, 1), (70001, 1000)):
        label = torch.randint(0, n_classes, (B,), generator=g).to(DEV)
        u = torch.rand(B, 9, dtype=torch.float64, generator=g).to(DEV)
        ls, us, perm = torch.empty_like(label), torch.empty_like(u), torch.empty(B, dtype=torch.int32, device=DEV)
        ops.prior_class_order(label, u, n_classes, ls, us, perm)
        want_l, want_p = torch.sort(label, stable=True)
        assert torch.equal(perm.long(), want_p), (B, n_classes)          # the stable order: independent of timing
        assert torch.equal(ls, want_l) and torch.equal(us, u[want_p])
        x = torch.randint(0, 128, (B, 9), generator=g).to(DEV)
        back = ops.unpermute_rows(x, perm)
        want_x = torch.empty_like(x)
        want_x[want_p] = x
        assert torch.equal(back, want_x)
