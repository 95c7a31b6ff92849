"""Test-time adaptation kernels (MANO backward, contact/penetration gradient, the SGD loop)
against the CPU oracle's autograd, through the C ABI.

Tolerances (float32 arithmetic in a different summation order than autograd's):
  MANO backward      max|g - g_ref| <= 2e-5 * max|g_ref| per output tensor
  loss terms         relative 1e-5
  nearest neighbour  equal indices except where the two best squared distances agree to 4 ulps
  d loss / d verts   max abs error <= 1e-5 * max|g_ref|
  refinement loop    every parameter update within 2e-4 relative of the oracle's (see the test)
"""
import numpy as np
import pytest
import torch

from dvq_hip import synth
from dvq_hip.hand_parts import CONTACT_VERTS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def hand(dev):
    import mano
    from dvq_hip.mano_io import Struct
    rh = mano.MANO('', data_struct=Struct(**synth.mano_model_data(0)), use_pca=True, num_pca_comps=45, batch_size=1,
                   flat_hand_mean=True).to(dev).eval()
    return rh, synth.mano_state_dict(0)


def _scene(sd, B, seed, spread=0.04):
    """Parameters of B hands and, per hand, 3000 object points scattered around it."""
    from oracle import dvq_oracle as O
    go, hp, tr = synth.mano_inputs(B, seed=seed)
    param = torch.zeros(B, 61)
    param[:, :10] = torch.randn(B, 10, generator=torch.Generator().manual_seed(seed))   # ignored by the layer
    param[:, 10:13], param[:, 13:58], param[:, 58:] = go, hp, tr
    verts, _, _ = O.mano_forward(sd, go, hp, tr)
    g = torch.Generator().manual_seed(seed + 100)
    obj = verts.mean(1, keepdim=True) + spread * torch.randn(B, 3000, 3, generator=g)
    obj_pc = torch.cat([obj.permute(0, 2, 1), torch.full((B, 1, 3000), 0.2)], 1).contiguous()    # [B,4,N] as gen() takes it
    return param, obj, obj_pc


def _relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def test_mano_backward_matches_autograd(dev, hand):
    from dvq_hip import ops
    from oracle import dvq_oracle as O
    rh, sd = hand
    B = 7
    go, hp, tr = synth.mano_inputs(B, seed=3)
    gv = torch.randn(B, 778, 3, generator=torch.Generator().manual_seed(9))
    leaf = [t.clone().requires_grad_(True) for t in (go, hp, tr)]
    verts, _, _ = O.mano_forward(sd, *leaf)
    ref = torch.autograd.grad((verts * gv).sum(), leaf)
    consts = rh._device_consts()
    v, _, _, vp = ops.mano_lbs(go.to(dev), hp.to(dev), tr.to(dev), consts, 778, False, False, want_v_posed=True)
    assert _relmax(v.cpu(), verts.detach()) < 1e-5
    got = ops.mano_lbs_backward(go.to(dev), hp.to(dev), consts, vp, gv.to(dev))
    for name, g, r in zip(("global_orient", "hand_pose", "transl"), got, ref):
        assert _relmax(g.cpu(), r) < 2e-5, name


def test_mano_layer_is_differentiable_like_the_reference_layer(dev, hand):
    """The reference optimises recon_param through rh_mano with autograd (gen_HDMO_TTA.py:119-151)."""
    from oracle import dvq_oracle as O
    rh, sd = hand
    B = 4
    go, hp, tr = synth.mano_inputs(B, seed=5)
    w = torch.randn(B, 778, 3, generator=torch.Generator().manual_seed(1))
    p_ref = torch.cat([torch.zeros(B, 10), go, hp, tr], 1).requires_grad_(True)
    v_ref, _, _ = O.mano_forward(sd, p_ref[:, 10:13], p_ref[:, 13:58], p_ref[:, 58:])
    (v_ref * w).sum().backward()
    p = p_ref.detach().clone().to(dev).requires_grad_(True)
    out = rh(betas=p[:, :10], global_orient=p[:, 10:13], hand_pose=p[:, 13:58], transl=p[:, 58:])
    assert out.vertices.requires_grad and not out.joints.requires_grad
    (out.vertices * w.to(dev)).sum().backward()
    assert _relmax(p.grad.cpu(), p_ref.grad) < 2e-5
    assert float(p.grad[:, :10].abs().max()) == 0.0
    with torch.no_grad():                                   # the inference path is untouched
        assert not rh(global_orient=p[:, 10:13], hand_pose=p[:, 13:58], transl=p[:, 58:]).vertices.requires_grad


def test_contact_gradient_matches_oracle(dev, hand):
    from dvq_hip.tta import GraspRefiner
    from oracle import tta_oracle as T
    rh, sd = hand
    param, obj, obj_pc = _scene(sd, 3, seed=11)
    ref = T.loss_and_grad(sd, sd["faces_tensor"], obj, param, CONTACT_VERTS)
    got = GraspRefiner(rh).loss_and_grad(obj_pc.to(dev), param.to(dev), want_nn=True)
    assert int(ref["interior"].sum()) > 100 and int(ref["cmap"].sum()) > 100       # both terms are exercised
    assert torch.allclose(got["losses"][:, 0].cpu(), ref["penetr"].detach(), rtol=1e-5)
    assert torch.allclose(got["losses"][:, 1].cpu(), ref["contact"].detach(), rtol=1e-5)
    nn = got["nn_idx"].cpu().long()
    diff = nn != ref["nn_idx"]
    if diff.any():                                          # only near-ties may resolve differently
        d = ((obj.unsqueeze(2) - ref["verts"].unsqueeze(1)) ** 2).sum(-1)
        a = d.gather(2, nn.unsqueeze(-1)).squeeze(-1)[diff]
        b = ref["nn_dist"].detach()[diff]
        assert ((a - b).abs() <= 4 * torch.finfo(torch.float32).eps * b).all()
    assert int(diff.sum()) <= 2
    assert _relmax(got["grad_verts"].cpu(), ref["grad_verts"]) < 1e-5
    assert _relmax(got["grad_param"].cpu(), ref["grad_param"]) < 5e-5
    # [B,N,3] layout and a strided view give the same numbers
    from dvq_hip import ops
    r = GraspRefiner(rh)
    gv2, l2 = ops.tta_contact_grad(obj.to(dev).contiguous(), got["verts"], r.mesh, T.CONTACT_DIST_SQ, 120.0, 7500.0, 5.0, 1.0,
                                   points_last=False)
    assert torch.equal(gv2, got["grad_verts"]) and torch.equal(l2, got["losses"])
    gv3, _ = ops.tta_contact_grad(obj_pc.to(dev)[:, :3, :], got["verts"], r.mesh, T.CONTACT_DIST_SQ, 120.0, 7500.0, 5.0, 1.0)
    assert torch.equal(gv3, got["grad_verts"])


def test_contact_gradient_with_the_closed_hand_mesh(dev, hand):
    """The reference takes its vertex normals from a closed hand mesh of 1552 faces, not from the model's own 1538
    (gen_HDMO_TTA.py:61-62: closed_mano_faces.pkl = the model's faces + 14 over the wrist).  A face set handed to
    GraspRefiner(rh, faces=...) must be the one the kernel uses: same scene, a synthetic face set of that shape,
    against the oracle with the same faces.  (Normals / nearest neighbour themselves: parity unpinned, as stated in
    oracle/tta_oracle.py.)"""
    from dvq_hip.tta import GraspRefiner
    from oracle import tta_oracle as T
    rh, sd = hand
    closed = torch.from_numpy(synth.closed_hand_faces(0))
    assert closed.shape == (1552, 3) and torch.equal(closed[:1538], sd["faces_tensor"])
    param, obj, obj_pc = _scene(sd, 3, seed=23)
    ref = T.loss_and_grad(sd, closed, obj, param, CONTACT_VERTS)
    ref_open = T.loss_and_grad(sd, sd["faces_tensor"], obj, param, CONTACT_VERTS)
    got = GraspRefiner(rh, faces=closed).loss_and_grad(obj_pc.to(dev), param.to(dev))
    assert int(ref["interior"].sum()) > 100
    assert torch.allclose(got["losses"][:, 0].cpu(), ref["penetr"].detach(), rtol=1e-5)
    assert torch.allclose(got["losses"][:, 1].cpu(), ref["contact"].detach(), rtol=1e-5)
    assert _relmax(got["grad_verts"].cpu(), ref["grad_verts"]) < 1e-5
    assert _relmax(got["grad_param"].cpu(), ref["grad_param"]) < 5e-5
    # the extra faces matter: the 16 loop vertices get other normals, so the interior mask and the gradient move
    assert not torch.equal(ref["interior"], ref_open["interior"]) or _relmax(ref["grad_verts"], ref_open["grad_verts"]) > 1e-4
    assert _relmax(got["grad_verts"].cpu(), ref_open["grad_verts"]) > 1e-6
    # the whole loop on the closed mesh (small steps, masks fixed): GPU trajectory vs the oracle's SGD
    iters, lr = 8, 1e-8
    p_ref, _ = T.refine(sd, closed, obj, param, CONTACT_VERTS, iters=iters, lr=lr)
    out = GraspRefiner(rh, faces=closed, lr=lr).refine(obj_pc.to(dev), param.to(dev), iters=iters)
    assert _relmax(out["recon_param"].cpu() - param, p_ref - param) < 2e-4


def test_exact_distance_ties_go_to_the_lowest_vertex_id(dev, hand):
    """Two vertices at one position, one inside the contact set and one outside: the nearest-vertex index of a point
    next to them is the LOWER id whichever part holds it (the kernel searches the contact set first; the rule is that
    of a plain first-minimum scan over all vertices, which the oracle states too)."""
    from dvq_hip import ops
    from dvq_hip.tta import GraspRefiner
    from oracle import tta_oracle as T
    rh, sd = hand
    param, obj, obj_pc = _scene(sd, 1, seed=5)
    r = GraspRefiner(rh)
    verts = r.loss_and_grad(obj_pc.to(dev), param.to(dev))["verts"].clone()
    contact = sorted(CONTACT_VERTS)
    rest = sorted(set(range(778)) - set(contact))
    c_hi, n_lo = contact[-1], rest[0]             # contact id above a non-contact id
    c_lo, n_hi = contact[0], rest[-1]             # contact id below a non-contact id
    assert n_lo < c_hi and c_lo < n_hi
    verts[0, c_hi] = verts[0, n_lo]
    verts[0, n_hi] = verts[0, c_lo]
    pts = obj.clone().to(dev)
    pts[0, 0] = verts[0, n_lo] + 1e-4
    pts[0, 1] = verts[0, c_lo] - 1e-4
    nn = torch.zeros(1, pts.shape[1], dtype=torch.int32, device=dev)
    ops.tta_contact_grad(pts.contiguous(), verts, r.mesh, T.CONTACT_DIST_SQ, 120.0, 7500.0, 5.0, 1.0, nn_idx=nn, points_last=False)
    assert int(nn[0, 0]) == n_lo and int(nn[0, 1]) == c_lo


def test_no_contact_and_repeatability(dev, hand):
    from dvq_hip.tta import GraspRefiner
    rh, sd = hand
    param, obj, obj_pc = _scene(sd, 2, seed=13)
    far = obj_pc.clone()
    far[:, :3] += 5.0                                       # five metres away: nothing within 2 cm
    r = GraspRefiner(rh)
    out = r.loss_and_grad(far.to(dev), param.to(dev))
    assert float(out["losses"][:, 1].abs().max()) == 0.0 and torch.isfinite(out["grad_param"]).all()
    a = r.loss_and_grad(obj_pc.to(dev), param.to(dev))
    b = r.loss_and_grad(obj_pc.to(dev), param.to(dev))
    assert torch.equal(a["grad_verts"], b["grad_verts"]) and torch.equal(a["grad_param"], b["grad_param"])


def test_refinement_loop_matches_oracle(dev, hand):
    """The loss is discontinuous in the parameters (nearest-neighbour and interior masks flip as the hand
    moves), and at the reference's step size the synthetic scene moves about a centimetre per step, so two
    float32 implementations drift apart chaotically after a few steps.  The loop is therefore checked
    (a) free-running against the oracle's torch.optim.SGD at a step size small enough to keep the masks
    fixed, and (b) at the reference's step size one step at a time: every update of the GPU trajectory must
    be the momentum update built from the oracle's gradients at the GPU's own iterates."""
    from dvq_hip.tta import GraspRefiner
    from oracle import tta_oracle as T
    rh, sd = hand
    param, obj, obj_pc = _scene(sd, 3, seed=17)
    faces = sd["faces_tensor"]
    # (a) free-running, small steps
    iters, lr = 12, 1e-8
    p_ref, hist = T.refine(sd, faces, obj, param, CONTACT_VERTS, iters=iters, lr=lr)
    r = GraspRefiner(rh, lr=lr, use_graph=True)
    out = r.refine(obj_pc.to(dev), param.to(dev), iters=iters)
    d_ref = p_ref - param
    d_got = out["recon_param"].cpu() - param
    assert float(d_ref.abs().max()) > 1e-5                                   # the loop moved the hand
    assert _relmax(d_got, d_ref) < 2e-4
    assert torch.equal(out["recon_param"][:, :10].cpu(), param[:, :10])      # shape coefficients untouched
    assert torch.allclose(out["first"].cpu(), hist[0], rtol=1e-5)
    assert torch.allclose(out["last"].cpu(), hist[-1], rtol=1e-4)
    eager = GraspRefiner(rh, lr=lr, use_graph=False).refine(obj_pc.to(dev), param.to(dev), iters=iters)
    assert torch.equal(eager["recon_param"], out["recon_param"])             # graph replay == eager launches
    again = r.refine(obj_pc.to(dev), param.to(dev), iters=iters)             # the captured graph is reusable
    assert torch.equal(again["recon_param"], out["recon_param"])
    # (b) the reference's step size and momentum, step by step
    r = GraspRefiner(rh, use_graph=False)
    traj = [param] + [r.refine(obj_pc.to(dev), param.to(dev), iters=k)["recon_param"].cpu() for k in range(1, 6)]
    buf = torch.zeros_like(param)
    for k in range(5):
        g = T.loss_and_grad(sd, faces, obj, traj[k], CONTACT_VERTS)["grad_param"]
        buf = T.MOMENTUM * buf + g
        step_ref = -T.LR * buf
        assert _relmax(traj[k + 1] - traj[k], step_ref) < 2e-4, k
    assert float((traj[5] - traj[0]).abs().max()) > 1e-3


def test_refiner_refuses_cpu_tensors(dev, hand):
    from dvq_hip.tta import GraspRefiner
    rh, sd = hand
    param, obj, obj_pc = _scene(sd, 1, seed=1)
    with pytest.raises(RuntimeError):
        GraspRefiner(rh).refine(obj_pc, param)


def test_refiner_with_an_empty_batch_and_with_no_steps(dev, hand):
    """A rank without objects hands the loop an empty batch: nothing is launched or recorded, shapes are kept.
    iters = 0 returns the parameters as they came and no loss values."""
    from dvq_hip.tta import GraspRefiner
    rh, sd = hand
    param, obj, obj_pc = _scene(sd, 2, seed=2)
    r = GraspRefiner(rh, iters=4)
    for _ in range(2):
        out = r.refine(obj_pc[:0].to(dev), param[:0].to(dev))
        assert tuple(out["recon_param"].shape) == (0, 61) and tuple(out["first"].shape) == (0, 2) and tuple(out["last"].shape) == (0, 2)
    out = r.refine(obj_pc.to(dev), param.to(dev), iters=0)
    assert torch.equal(out["recon_param"].cpu(), param) and out["first"] is None and out["last"] is None
    moved = r.refine(obj_pc.to(dev), param.to(dev))             # the same refiner still works on a real batch afterwards
    assert tuple(moved["first"].shape) == (2, 2) and not torch.equal(moved["recon_param"].cpu(), param)
