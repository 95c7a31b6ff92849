#!/usr/bin/env python3
"""Capture golden vectors from the reference implementation (CPU, fp32).

Run in the build container only (it needs /root/reference):
    python tests/golden/make_golden.py
It imports the reference's own modules unmodified, loads the seeded synthetic weights
of dvq_hip.synth into them with a strict load_state_dict (which also proves the
checkpoint key layout), runs them, and writes inputs that cannot be regenerated plus
all expected outputs to tests/golden/*.npz.  Nothing from the reference's source is
stored: only arrays.

Shims applied to make the reference run on a CPU-only box (none edits a reference file):
  * network.vqvae.quantizer.device (module global 'cuda', quantizer.py:7) -> cpu
  * Tensor.to('cuda') inside GenNet.gen (gen_net.py:79) is redirected to the cpu
  * `trimesh` (imported by mano/utils.py:23 for visualisation only) is stubbed
  * MANO is constructed through its own `data_struct=` argument: from a synthetic
    MANO-shaped model for the committed fixtures; for the uncommitted `mano_real` step from
    the arrays of the licensed MANO_RIGHT.pkl, read as data (never unpickled)
  * DVQVAE.forward reads an undefined global `f0hand` (DVQVAE.py:93); the complement of
    the five listed vertex sets in range(778) is injected (a documented reconstruction)
  * GenNet.gen only works at batch 1 (quantizer.py:70-74), so batches are B=1 loops
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "d-vqvae_amd"))
from dvq_hip import synth  # noqa: E402

torch.set_num_threads(8)


def import_reference():
    tm = types.ModuleType("trimesh")
    tm.Trimesh = type("Trimesh", (), {})
    sys.modules.setdefault("trimesh", tm)
    # The repo's own drop-in packages use the same top-level names; make sure the
    # reference's win for this process.
    for name in [m for m in sys.modules if m == "network" or m.startswith("network.") or m == "mano"
                 or m.startswith("mano.") or m == "utils" or m.startswith("utils.")]:
        del sys.modules[name]
    # (the reference's `network` directory has no __init__.py, so the repo's regular package of the same name would
    # shadow it from anywhere on sys.path: take the repo's package root off the path; dvq_hip.synth stays loaded)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p) != os.path.join(ROOT, "d-vqvae_amd")]
    sys.path.insert(0, os.path.join(REF, "DVQ-VAE-2", "MANO-master"))
    sys.path.insert(0, os.path.join(REF, "DVQ-VAE"))
    import network.vqvae.quantizer as quantizer
    quantizer.device = torch.device("cpu")
    import network.pointnet_encoder as pe
    import network.DVQVAE as dv
    import network.gen_net as gn
    import network.pixelcnn.models as pm
    import mano
    import mano.utils as mutils
    assert quantizer.__file__.startswith(REF) and mano.__file__.startswith(REF), (quantizer.__file__, mano.__file__)
    return types.SimpleNamespace(quantizer=quantizer, pe=pe, dv=dv, gn=gn, pm=pm, mano=mano, mutils=mutils)


class cuda_to_cpu:
    """Redirect .to('cuda') to the cpu while the reference's gen() runs."""

    def __enter__(self):
        self.orig = torch.Tensor.to

        def to(t, *a, **k):
            a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
            if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                k["device"] = "cpu"
            return self.orig(t, *a, **k)
        torch.Tensor.to = to

    def __exit__(self, *exc):
        torch.Tensor.to = self.orig


def ref_mano(ref, seed, batch_size):
    d = synth.mano_model_data(seed)
    return ref.mano.MANO(model_path="", data_struct=ref.mutils.Struct(**d), use_pca=True, num_pca_comps=45,
                         batch_size=batch_size, flat_hand_mean=True)


OUT_DIR = HERE


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


VQ_CASES = [(128, 256), (128, 1024), (512, 256), (2048, 256), (8192, 256)]
VQ_N = 4096


def golden_vq(ref):
    out = {}
    for K, D in VQ_CASES:
        for dist in "AB":
            z, E = synth.vq_case(K, D, VQ_N, dist, seed=0)
            vq = ref.quantizer.VectorQuantizer(K, D, 0.25, 1)
            vq.embedding.weight.data.copy_(E)
            with torch.no_grad():
                idx, zq = vq(z, False)
            assert idx.dtype == torch.int64 and tuple(idx.shape) == (VQ_N, 1)
            assert torch.equal(zq, E[idx[:, 0]])  # one-hot matmul == row gather, exactly
            out[f"idx_{K}_{D}_{dist}"] = idx[:, 0].numpy().astype(np.int32)
    # ragged / tiny / tie cases
    z, E = synth.vq_case(128, 256, 37, "A", seed=3)
    E[5] = E[77]                      # two identical codes: lowest index must win
    z[0] = E[77]
    vq = ref.quantizer.VectorQuantizer(128, 256, 0.25, 1)
    vq.embedding.weight.data.copy_(E)
    with torch.no_grad():
        idx, _ = vq(z, False)
    out["idx_ragged37"] = idx[:, 0].numpy().astype(np.int32)
    # get_emb at its only supported size
    with torch.no_grad():
        e = vq.get_emb(torch.tensor([9]), 256)
    out["get_emb_9"] = e.numpy()
    save("vq", **out)


VQ_NEAR_CASES = [(128, 256), (512, 256)]


def golden_vq_near_ties(ref):
    """The reference quantizer on latents whose two best codes are nearly equidistant (synth.vq_near_tie_case):
    rows whose exact gap is below a few fp32 spacings of |z|^2 + |E|^2 are the ones on which the evaluation order
    of quantizer.py:46-48 decides.  Stored: the reference's indices; printed: how many rows are near-ties under the
    rule of tests/parity.py and on how many the reference itself leaves the exact (fp64) minimum."""
    sys.path.insert(0, ROOT)
    from oracle import dvq_oracle as O
    out = {}
    for K, D in VQ_NEAR_CASES:
        for kind in ("mid", "bignorm"):
            z, E = synth.vq_near_tie_case(K, D, VQ_N, kind, seed=0)
            vq = ref.quantizer.VectorQuantizer(K, D, 0.25, 1)
            vq.embedding.weight.data.copy_(E)
            with torch.no_grad():
                idx, zq = vq(z, False)
            assert torch.equal(zq, E[idx[:, 0]])
            best, _, gap = O.vq_gap_fp64(z.numpy(), E.numpy())
            tau = O.vq_tau(z.numpy(), E.numpy())
            near = int((gap <= tau).sum())
            off = int((idx[:, 0].numpy() != best).sum())
            print(f"near-tie case K={K} D={D} {kind}: {near} of {VQ_N} rows within tau, reference != exact argmin on {off}")
            assert near > 0
            out[f"idx_{K}_{D}_{kind}"] = idx[:, 0].numpy().astype(np.int32)
            out[f"near_{K}_{D}_{kind}"] = np.int32(near)
            out[f"ref_off_exact_{K}_{D}_{kind}"] = np.int32(off)
    save("vq_near", **out)


VQ_TRAIN_CASES = [(128, 256, 96), (128, 1024, 24)]


def vq_train_weights(n, D):
    """The fixed cotangent of z_q in the training-mode fixture (regenerated by the tests from the same seed)."""
    return torch.randn(n, D, generator=torch.Generator().manual_seed(11))


def golden_vq_train(ref):
    """Training-mode outputs of the reference: VQVAE.forward (VQVAE.py:29-42) -> (loss, straight-through z_q,
    perplexity), the quantizer's own five outputs (quantizer.py:35-43, 56-67: one-hot rows stored as column sums, and
    indices), and the gradients of  loss + <z_q, w>  with respect to the latents and the codebook.  Latents and codebook
    are regenerated from the seed (synth.vq_case 'B', seed 5), w by vq_train_weights."""
    import network.VQVAE as vw
    out = {}
    for K, D, n in VQ_TRAIN_CASES:
        z, E = synth.vq_case(K, D, n, "B", seed=5)
        w = vq_train_weights(n, D)
        model = vw.VQVAE(0, 0, 0, K, D, 0.25, a=1)
        model.vector_quantization.embedding.weight.data.copy_(E)
        zr = z.clone().requires_grad_(True)
        loss, zq, perp = model(zr)
        (loss + (zq * w).sum()).backward()
        with torch.no_grad():
            loss5, zq5, perp5, onehot, idx = model.vector_quantization(z, True)
        assert torch.equal(loss5, loss) and torch.equal(zq5, zq) and torch.equal(perp5, perp)
        assert tuple(onehot.shape) == (n, K) and tuple(idx.shape) == (n, 1) and torch.equal(onehot.argmax(1), idx[:, 0])
        out[f"loss_{K}_{D}"] = loss.detach().numpy()
        out[f"zq_{K}_{D}"] = zq.detach().numpy()
        out[f"perplexity_{K}_{D}"] = perp.detach().numpy()
        out[f"idx_{K}_{D}"] = idx[:, 0].numpy().astype(np.int32)
        out[f"dz_{K}_{D}"] = zr.grad.numpy()
        out[f"dE_{K}_{D}"] = model.vector_quantization.embedding.weight.grad.numpy()
    save("vq_train", **out)


def real_clouds():
    """The four object clouds bundled with the reference ((3000,3) float64 .npy, plain
    arrays): used as realistic encoder inputs with the dataset's 4-channel layout
    (DVQ-VAE/dataset/utils_HO3D_FPHA.py:75-84: 4th channel = bbox diagonal)."""
    base = os.path.join(REF, "DVQ-VAE", "models", "Object_models")
    clouds = []
    for root, _, files in sorted(os.walk(base)):
        for f in sorted(files):
            if f.endswith("resampled.npy"):
                p = np.load(os.path.join(root, f), allow_pickle=False)
                if p.shape == (3000, 3):
                    clouds.append(p)
    assert len(clouds) == 4, len(clouds)
    out = []
    for p in clouds:
        diag = np.linalg.norm(p.max(0) - p.min(0))
        q = p + np.array(synth.OBJ_TRANSLATION)
        out.append(np.concatenate([q.T, np.full((1, 3000), diag)], 0).astype(np.float32))
    return torch.from_numpy(np.stack(out))


def golden_pointnet(ref):
    sd = synth.gennet_state_dict(0, with_prior=False)
    real = real_clouds()
    x4 = torch.cat([real, synth.object_clouds(4, 3000, seed=7)], 0)
    enc = ref.pe.PointNetEncoder(global_feat=True, feature_transform=False, channel=4).eval()
    enc.load_state_dict({k[len("obj_encoder_type."):]: v for k, v in sd.items() if k.startswith("obj_encoder_type.")})
    with torch.no_grad():
        feat4, trans4, _ = enc(x4)
    enc3 = ref.pe.PointNetEncoder(global_feat=True, feature_transform=False, channel=3).eval()
    enc3.load_state_dict({k[len("recon_encoder."):]: v for k, v in sd.items() if k.startswith("recon_encoder.")})
    r = np.random.default_rng(11)
    x3 = torch.from_numpy(r.normal(0, 0.06, size=(3, 3, 778)).astype(np.float32))
    xs = torch.from_numpy(r.normal(0, 0.03, size=(2, 3, 83)).astype(np.float32))   # smallest hand part
    with torch.no_grad():
        feat3, trans3, _ = enc3(x3)
        feats, transs, _ = enc3(xs)
    save("pointnet", real_clouds=real, feat4=feat4, trans4=trans4, x3=x3, feat3=feat3, trans3=trans3,
         xs=xs, feats=feats, transs=transs)


def golden_mlp(ref):
    sd = synth.gennet_state_dict(0, with_prior=False)
    dsd = synth.dvqvae_state_dict(0)
    r = np.random.default_rng(5)
    dec = ref.dv.Decoder(layer_sizes=[1024, 256, 55], latent_size=2560).eval()
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    pdec = ref.dv.Decoder(layer_sizes=[1024, 128, 6], latent_size=2048).eval()
    pdec.load_state_dict({k[len("pos_decoder."):]: v for k, v in sd.items() if k.startswith("pos_decoder.")})
    emb = ref.dv.Encoder([1024, 512], 256).eval()
    emb.load_state_dict({k[len("emb_0."):]: v for k, v in dsd.items() if k.startswith("emb_0.")})
    a = torch.from_numpy(r.normal(0, 0.5, size=(5, 2560)).astype(np.float32))
    b = torch.from_numpy(r.normal(0, 0.5, size=(5, 2048)).astype(np.float32))
    c = torch.from_numpy(r.normal(0, 0.5, size=(5, 1024)).astype(np.float32))
    with torch.no_grad():
        save("mlp", dec_in=a, dec_out=dec(a), pos_in=b, pos_out=pdec(b), emb_in=c, emb_out=emb(c))


def mano_special_inputs():
    go, hp, tr = synth.mano_inputs(8, seed=2)
    hp[0] = 0; go[0] = 0; tr[0] = 0            # rest pose: hits the +1e-8 norm path
    hp[1] *= 3.0                               # large angles
    go[2] = torch.tensor([3.1, -3.1, 3.1])
    return go, hp, tr


def golden_mano(ref):
    go, hp, tr = mano_special_inputs()
    m = ref_mano(ref, 0, batch_size=8).eval()
    betas = torch.full((8, 10), 5.0)           # ignored by the reference (model.py:335)
    with torch.no_grad():
        o = m(betas=betas, global_orient=go.clone(), hand_pose=hp.clone(), transl=tr.clone(), return_full_pose=True)
        o0 = m(betas=None, global_orient=go.clone(), hand_pose=hp.clone(), transl=tr.clone())
    assert torch.equal(o.vertices, o0.vertices)
    sd = m.state_dict()
    mine = synth.mano_state_dict(0, batch_size=8)
    assert list(sd.keys()) == list(mine.keys()), (list(sd.keys()), list(mine.keys()))
    for k in sd:
        assert sd[k].dtype == mine[k].dtype and torch.equal(sd[k], mine[k]), k
    assert np.array_equal(np.asarray(m.faces), synth.mano_model_data(0)["f"])
    save("mano", global_orient=go, hand_pose=hp, transl=tr, vertices=o.vertices, joints=o.joints,
         full_pose=o.full_pose)


def hmdo_records():
    """The grasps the trained reference model generated for 12 HMDO objects, as its script stored them
    (DVQ-VAE-2/diverse_grasp/HMDO/obj_id_*.json, plain JSON; gen_diverse_grasp_ho3d.py:315-331 writes this
    layout): 120 real 61-d MANO parameter vectors and the placements they were generated under."""
    import json
    base = os.path.join(REF, "DVQ-VAE-2", "diverse_grasp", "HMDO")
    params, R, r, trans = [], [], [], []
    for k in range(12):
        with open(os.path.join(base, f"obj_id_{k}.json")) as f:
            d = json.load(f)
        params.append(np.asarray(d["recon_params"], dtype=np.float64).reshape(-1, 61))
        R.append(np.asarray(d["R_list"], dtype=np.float64))
        r.append(np.asarray(d["r_list"], dtype=np.float64))
        trans.append(np.asarray(d["trans_list"], dtype=np.float64))
    return np.concatenate(params), np.concatenate(R), np.concatenate(r), np.concatenate(trans)


def golden_mano_hmdo(ref):
    """The reference MANO class on the 120 real parameter vectors (call-site slicing of
    gen_diverse_grasp_ho3d.py:244-258: betas | global_orient | hand_pose | transl), plus a rest pose and two
    large-angle rows scaled from real ones.  Joints for every row, vertices for every second row (file size)."""
    params, R, r, trans = hmdo_records()
    assert params.shape == (120, 61) and R.shape == (120, 3, 4) and r.shape == (120, 3)
    p = torch.from_numpy(params.astype(np.float32))
    extra = p[:3].clone()
    extra[0, 10:] = 0                          # rest pose at the origin
    extra[1, 13:58] *= 3.0                     # large finger angles on real statistics
    extra[2, 10:13] = torch.tensor([3.1, -3.1, 3.1])
    p = torch.cat([p, extra])
    m = ref_mano(ref, 0, batch_size=p.shape[0]).eval()
    with torch.no_grad():
        o = m(betas=p[:, :10].clone(), global_orient=p[:, 10:13].clone(), hand_pose=p[:, 13:58].clone(),
              transl=p[:, 58:].clone())
    save("mano_hmdo", recon_params=p, joints=o.joints, vertices_even=o.vertices[::2], R=R, r=r, trans=trans)
    # one record verbatim (data, 17 KB): pins the reader of the reference's output format
    import shutil
    shutil.copyfile(os.path.join(REF, "DVQ-VAE-2", "diverse_grasp", "HMDO", "obj_id_0.json"),
                    os.path.join(HERE, "hmdo_obj_id_0.json"))


MANO_DIR = os.path.join(REF, "DVQ-VAE", "models", "mano")


def golden_mano_real(ref):
    """The reference MANO class on the constants of the official right-hand model file that ships in the reference
    tree, for the 123 rows of golden_mano_hmdo.  The file is under the MANO licence: nothing derived from it is
    committed -- this step is run on demand by tests/test_real_mano_constants.py, which points OUT_DIR at a temporary
    directory.  The file is a pickle; it is read as data by dvq_hip.mano_io (which neither imports nor calls anything
    the file names) and handed to the reference class through its own `data_struct=` argument, the path
    mano/model.py:140-158 takes after its pickle.load."""
    from dvq_hip import mano_io
    data = mano_io.read_model_file(os.path.join(MANO_DIR, "MANO_RIGHT.pkl"))
    params, _, _, _ = hmdo_records()
    p = torch.from_numpy(params.astype(np.float32))
    extra = p[:3].clone()
    extra[0, 10:] = 0
    extra[1, 13:58] *= 3.0
    extra[2, 10:13] = torch.tensor([3.1, -3.1, 3.1])
    p = torch.cat([p, extra])
    m = ref.mano.MANO(model_path="", data_struct=ref.mutils.Struct(**data), use_pca=True, num_pca_comps=45,
                      batch_size=p.shape[0], flat_hand_mean=True).eval()
    sd = m.state_dict()
    mine = synth.mano_state_dict(batch_size=p.shape[0], data=data)
    assert list(sd.keys()) == list(mine.keys())
    for k in sd:
        assert sd[k].dtype == mine[k].dtype and torch.equal(sd[k], mine[k]), k
    with torch.no_grad():
        o = m(betas=p[:, :10].clone(), global_orient=p[:, 10:13].clone(), hand_pose=p[:, 13:58].clone(),
              transl=p[:, 58:].clone())
    save("mano_real", recon_params=p, joints=o.joints, vertices=o.vertices)


def golden_gen(ref, n_obj=16):
    sd = synth.gennet_state_dict(0, with_prior=True)
    model = ref.gn.GenNet()
    keys = list(model.state_dict().keys())
    assert keys == list(sd.keys()), "state_dict key order/layout differs from the reference"
    model.load_state_dict(sd)
    model.eval()
    rh = ref_mano(ref, 0, batch_size=1).eval()
    model.set_rh_mano(rh)
    assert len(model.state_dict()) == len(sd) + 14
    objs = synth.object_clouds(n_obj, 3000, seed=1234)
    captured = {}
    orig_generate = model.GatedPixelCNN.generate

    def generate(*a, **k):
        x = orig_generate(*a, **k)
        captured["x"] = x.clone()
        return x
    model.GatedPixelCNN.generate = generate
    rec, pos, xt, verts, logits = [], [], [], [], []
    with cuda_to_cpu(), torch.no_grad():
        for b in range(n_obj):
            torch.manual_seed(100 + b)
            r, p = model.gen(objs[b:b + 1])
            rec.append(r); pos.append(p); xt.append(captured["x"])
            prm = torch.zeros(1, 61)
            prm[:, 0:10] = r[:, 0:10]; prm[:, 10:13] = p[:, 0:3]; prm[:, 13:58] = r[:, 10:55]; prm[:, 58:61] = p[:, 3:6]
            out = rh(betas=prm[:, :10], global_orient=prm[:, 10:13], hand_pose=prm[:, 13:58], transl=prm[:, 58:])
            verts.append(out.vertices)
    xt = torch.cat(xt)
    # prior logits on the final grids (for parity of any re-implementation of the prior)
    with torch.no_grad():
        idx6 = []
        for b in range(n_obj):
            f, _, _ = model.obj_encoder_type(objs[b:b + 1])
            i6, _ = model.vqvae6.inference(f)
            idx6.append(i6[:, 0])
        idx6 = torch.cat(idx6)
        lg = model.GatedPixelCNN(xt[:2], idx6[:2])
    save("gen", recon=torch.cat(rec), recon_pos=torch.cat(pos), x_tilde=xt, vertices=torch.cat(verts),
         idx6=idx6, prior_logits=lg[:, :128].contiguous())


def f0hand_reconstruction(ref):
    import inspect
    import re
    src = inspect.getsource(ref.dv.DVQVAE.forward)
    used = set()
    for name in ("handc", "f1hand", "f2hand", "f3hand", "f4hand"):
        m = re.search(name + r"\s*=\s*\[([^\]]*)\]", src)
        used |= {int(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()}
    return sorted(set(range(778)) - used)


def golden_dvqvae(ref):
    f0 = f0hand_reconstruction(ref)
    assert len(f0) == 83, len(f0)
    ref.dv.f0hand = f0
    sd = synth.dvqvae_state_dict(0)
    model = ref.dv.DVQVAE(obj_inchannel=4)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.eval()
    objs = synth.object_clouds(2, 3000, seed=99)
    r = np.random.default_rng(17)
    hand = torch.from_numpy((r.normal(0, 0.05, size=(2, 3, 778)) + np.array([[0.1], [0.0], [-0.6]])).astype(np.float32))
    with torch.no_grad():
        emb_idx, obj_emb = model(objs, hand)
    save("dvqvae", f0hand=np.array(f0, np.int32), hand_xyz=hand, emb_idx=emb_idx, obj_emb=obj_emb)


STEPS = ("vq", "vq_near_ties", "vq_train", "pointnet", "mlp", "mano", "mano_hmdo", "dvqvae", "gen")


def main():
    """All committed fixtures, or only the named steps: `make_golden.py vq_near_ties mano`; `--out DIR` writes
    somewhere else (used for `mano_real`, whose output is not committed)."""
    global OUT_DIR
    args = sys.argv[1:]
    if "--out" in args:
        i = args.index("--out")
        OUT_DIR = args[i + 1]
        del args[i:i + 2]
    from dvq_hip import mano_io  # noqa: F401  (before the reference's packages take over the import path)
    ref = import_reference()
    for step in (args or STEPS):
        globals()["golden_" + step](ref)


if __name__ == "__main__":
    main()
