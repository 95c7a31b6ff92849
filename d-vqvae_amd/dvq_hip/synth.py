"""Seeded synthetic weights, MANO-shaped constants and inputs.

No pretrained checkpoint ships with the reference and the MANO model file cannot be
redistributed, so every test and benchmark runs on values generated here.  Values are
a pure function of (seed, tensor name): each tensor draws from its own numpy PCG64
stream keyed by crc32(name), so adding or reordering keys never changes another
tensor.  The same call therefore reproduces, on the GPU box, exactly the weights the
golden vectors were captured with.

Key layouts follow the reference's state_dicts:
  GenNet   DVQ-VAE/network/gen_net.py:14-34   (333 keys + 14 rh_mano.* after set_rh_mano)
  DVQVAE   DVQ-VAE/network/DVQVAE.py:12-35
  MANO     DVQ-VAE-2/MANO-master/mano/model.py:169-276 (registered buffers/parameters)
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

MANO_PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
N_VERTS = 778
N_FACES = 1538
# Object-cloud translation used by every generation script
# (DVQ-VAE/gen_diverse_grasp_ho3d.py:221).
OBJ_TRANSLATION = (-0.0793, 0.0208, -0.6924)


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def _uniform(seed, name, shape, bound):
    return torch.from_numpy(_rng(seed, name).uniform(-bound, bound, size=shape).astype(np.float32))


def _normal(seed, name, shape, std, mean=0.0):
    return torch.from_numpy((_rng(seed, name).standard_normal(size=shape) * std + mean).astype(np.float32))


def _linear(sd, seed, name, out_f, in_f, conv=False):
    bound = 1.0 / math.sqrt(in_f)
    shape = (out_f, in_f, 1) if conv else (out_f, in_f)
    sd[name + ".weight"] = _uniform(seed, name + ".weight", shape, bound)
    sd[name + ".bias"] = _uniform(seed, name + ".bias", (out_f,), bound)


def _bn(sd, seed, name, c):
    """Non-trivial eval-mode statistics, otherwise BatchNorm is the identity."""
    sd[name + ".weight"] = torch.from_numpy(_rng(seed, name + ".weight").uniform(0.5, 1.5, c).astype(np.float32))
    sd[name + ".bias"] = _normal(seed, name + ".bias", (c,), 0.1)
    sd[name + ".running_mean"] = _normal(seed, name + ".running_mean", (c,), 0.1)
    sd[name + ".running_var"] = torch.from_numpy(_rng(seed, name + ".running_var").uniform(0.5, 1.5, c).astype(np.float32))
    sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def pointnet_state(sd: Dict[str, torch.Tensor], seed: int, prefix: str, channel: int) -> None:
    """58 keys of one PointNetEncoder(channel) under `prefix` (with trailing dot)."""
    p = prefix + "stn."
    _linear(sd, seed, p + "conv1", 64, channel, conv=True)
    _linear(sd, seed, p + "conv2", 128, 64, conv=True)
    _linear(sd, seed, p + "conv3", 1024, 128, conv=True)
    _linear(sd, seed, p + "fc1", 512, 1024)
    _linear(sd, seed, p + "fc2", 256, 512)
    _linear(sd, seed, p + "fc3", 9, 256)
    for i, c in enumerate((64, 128, 1024, 512, 256), start=1):
        _bn(sd, seed, f"{p}bn{i}", c)
    _linear(sd, seed, prefix + "conv1", 64, channel, conv=True)
    _linear(sd, seed, prefix + "conv2", 128, 64, conv=True)
    _linear(sd, seed, prefix + "conv3", 1024, 128, conv=True)
    for i, c in enumerate((64, 128, 1024), start=1):
        _bn(sd, seed, f"{prefix}bn{i}", c)


def _codebooks(sd, seed, scale_obj=1.0):
    for k in range(6):
        name = f"vqvae{k}.vector_quantization.embedding.weight"
        sd[name] = _uniform(seed, name, (128, 256), 1.0 / 128)
    name = "vqvae6.vector_quantization.embedding.weight"
    # Object codebook: spread like a PointNet global feature so that nearest-code
    # decisions are not all near-ties.
    sd[name] = _normal(seed, name, (128, 1024), 0.5 * scale_obj)


def _decoder(sd, seed, prefix, sizes, latent):
    ins = [latent] + list(sizes[:-1])
    for i, (a, b) in enumerate(zip(ins, sizes)):
        _linear(sd, seed, f"{prefix}MLP.L{i}", b, a)


def pixelcnn_state(sd, seed, prefix="GatedPixelCNN.", input_dim=512, dim=512, n_layers=15, n_classes=128,
                   valid_classes=128):
    """140 keys of GatedPixelCNN(input_dim, dim, n_layers) (pixelcnn/models.py:130-159):
    xavier-uniform conv weights, zero conv biases, N(0,1) embeddings (the reference's
    weights_init only touches Conv modules).  Output classes >= valid_classes get a
    -1e4 bias so that samples stay inside the 128-row part codebooks."""
    def xavier(name, shape):
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        b = math.sqrt(6.0 / (fan_in + fan_out))
        sd[name] = _uniform(seed, name, shape, b)

    sd[prefix + "embedding.weight"] = _normal(seed, prefix + "embedding.weight", (input_dim, dim), 1.0)
    for i in range(n_layers):
        lp = f"{prefix}layers.{i}."
        k = 5 if i == 0 else 3
        sd[lp + "class_cond_embedding.weight"] = _normal(seed, lp + "class_cond_embedding.weight",
                                                         (n_classes, 2 * dim), 1.0)
        xavier(lp + "vert_stack.weight", (2 * dim, dim, k // 2 + 1, k))
        sd[lp + "vert_stack.bias"] = torch.zeros(2 * dim)
        xavier(lp + "vert_to_horiz.weight", (2 * dim, 2 * dim, 1, 1))
        sd[lp + "vert_to_horiz.bias"] = torch.zeros(2 * dim)
        xavier(lp + "horiz_stack.weight", (2 * dim, dim, 1, k // 2 + 1))
        sd[lp + "horiz_stack.bias"] = torch.zeros(2 * dim)
        xavier(lp + "horiz_resid.weight", (dim, dim, 1, 1))
        sd[lp + "horiz_resid.bias"] = torch.zeros(dim)
    xavier(prefix + "output_conv.0.weight", (2048, dim, 1, 1))
    sd[prefix + "output_conv.0.bias"] = torch.zeros(2048)
    xavier(prefix + "output_conv.2.weight", (input_dim, 2048, 1, 1))
    b = torch.zeros(input_dim)
    b[valid_classes:] = -1e4
    sd[prefix + "output_conv.2.bias"] = b


def gennet_state_dict(seed: int = 0, with_prior: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """State dict of GenNet() before set_rh_mano (333 keys with the prior)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    pointnet_state(sd, seed, "obj_encoder_type.", 4)
    pointnet_state(sd, seed, "obj_encoder_pos.", 4)
    _codebooks(sd, seed)
    _decoder(sd, seed, "decoder.", (1024, 256, 55), 2560)
    pointnet_state(sd, seed, "recon_encoder.", 3)
    _decoder(sd, seed, "pos_decoder.", (1024, 128, 6), 2048)
    if with_prior:
        pixelcnn_state(sd, seed)
    return sd


def dvqvae_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """State dict of DVQVAE(obj_inchannel=4)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k in range(6):
        _linear(sd, seed, f"emb_{k}.MLP.L0", 512, 1024)
        _linear(sd, seed, f"emb_{k}.linear_means", 256, 512)
        _linear(sd, seed, f"emb_{k}.linear_log_var", 256, 512)
    pointnet_state(sd, seed, "obj_encoder_type.", 4)
    pointnet_state(sd, seed, "obj_encoder_pos.", 4)
    for k in range(6):
        pointnet_state(sd, seed, f"fing_{k}.", 3)
    _codebooks(sd, seed)
    _decoder(sd, seed, "decoder.", (1024, 256, 55), 2560)
    pointnet_state(sd, seed, "recon_encoder.", 3)
    _decoder(sd, seed, "pos_decoder.", (1024, 128, 6), 2048)
    return sd


# ------------------------------------------------------------------------- MANO
def mano_model_data(seed: int = 0) -> Dict[str, np.ndarray]:
    """A MANO-shaped model in the layout of the fields the reference reads from
    MANO_RIGHT.pkl (mano/model.py:169-257): same array shapes, the true kinematic
    tree, a skinning-weight matrix with 1-6 non-zeros per vertex (mean about 2.6) whose
    rows sum to 1, and a row-stochastic sparse joint regressor.  Values are random:
    it stands in for the licensed file, it is not a hand."""
    r = _rng(seed, "mano")
    parents = np.array(MANO_PARENTS, dtype=np.int64)
    # joints: a root plus five 3-joint chains, roughly hand sized (metres)
    J = np.zeros((16, 3), np.float64)
    for f in range(5):
        ang = (f - 2) * 0.35
        d = np.array([math.sin(ang), 0.15 * (f - 2) / 2, math.cos(ang)])
        for s in range(3):
            J[1 + 3 * f + s] = d * (0.05 + 0.03 * s) * 1.2
    # vertices scattered around their dominant joint
    owner = r.integers(0, 16, size=N_VERTS)
    v_template = J[owner] + r.normal(0, 0.012, size=(N_VERTS, 3))
    nnz = r.choice([1, 2, 3, 4, 5, 6], size=N_VERTS, p=[0.18, 0.36, 0.26, 0.12, 0.05, 0.03])
    weights = np.zeros((N_VERTS, 16), np.float64)
    for v in range(N_VERTS):
        js = [owner[v]]
        cand = [j for j in range(16) if j != owner[v]]
        js += list(r.choice(cand, size=nnz[v] - 1, replace=False))
        w = r.uniform(0.05, 1.0, size=len(js))
        w[0] += 1.0
        weights[v, js] = w / w.sum()
    J_reg = np.zeros((16, N_VERTS), np.float64)
    for j in range(16):
        ids = r.choice(N_VERTS, size=24, replace=False)
        w = r.uniform(0.1, 1.0, size=24)
        J_reg[j, ids] = w / w.sum()
    shapedirs = r.normal(0, 2e-3, size=(N_VERTS, 3, 10))
    posedirs = r.normal(0, 2e-3, size=(N_VERTS, 3, 135))
    q, _ = np.linalg.qr(r.normal(size=(45, 45)))
    hands_components = q * r.uniform(0.2, 1.0, size=(45, 1))
    hands_mean = r.normal(0, 0.2, size=45)
    faces = r.integers(0, N_VERTS, size=(N_FACES, 3)).astype(np.uint32)
    kintree = np.stack([np.array([4294967295] + list(MANO_PARENTS[1:]), dtype=np.int64),
                        np.arange(16, dtype=np.int64)])
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_reg,
                weights=weights, hands_components=hands_components, hands_mean=hands_mean,
                kintree_table=kintree, f=faces)


def closed_hand_faces(seed: int = 0) -> np.ndarray:
    """A face set shaped like the reference's closed hand mesh (models/mano/closed_mano_faces.pkl, read by
    gen_HDMO_TTA.py:61-62): the model's own 1538 triangles followed by 14 that fan over a loop of 16 vertices (in the
    official file: the open wrist), (1552, 3) int64.  The loop here is 16 arbitrary distinct vertices."""
    f = mano_model_data(seed)["f"].astype(np.int64)
    loop = _rng(seed, "wrist_loop").choice(N_VERTS, size=16, replace=False)
    fan = [[loop[0], loop[i], loop[i + 1]] for i in range(1, 15)]          # a 16-gon takes 14 triangles
    return np.concatenate([f, np.asarray(fan, dtype=np.int64)])


def mano_state_dict(seed: int = 0, batch_size: int = 1, num_pca_comps: int = 45, flat_hand_mean: bool = True,
                    prefix: str = "", data=None) -> "OrderedDict[str, torch.Tensor]":
    """The 14 tensors MANO(...) registers (mano/model.py:170-276), built from
    mano_model_data(seed) -- or from `data`, a dict of the same fields read from a model file --
    exactly as the reference constructor derives them."""
    d = mano_model_data(seed) if data is None else data

    def f32(a):
        if hasattr(a, "todense"):                  # the joint regressor of the official files is a sparse matrix
            a = a.todense()
        return torch.from_numpy(np.array(a, dtype=np.float32))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd["betas"] = torch.zeros(batch_size, 10)
    sd["global_orient"] = torch.zeros(batch_size, 3)
    sd["transl"] = torch.zeros(batch_size, 3)
    sd["hand_pose"] = torch.zeros(batch_size, num_pca_comps)
    sd["faces_tensor"] = torch.from_numpy(np.asarray(d["f"]).astype(np.int64))
    sd["v_template"] = f32(d["v_template"])
    sd["shapedirs"] = f32(d["shapedirs"])
    sd["J_regressor"] = f32(d["J_regressor"])
    sd["posedirs"] = f32(np.reshape(d["posedirs"], [-1, 135]).T)
    par = torch.tensor(MANO_PARENTS, dtype=torch.int64)
    sd["parents"] = par
    sd["lbs_weights"] = f32(d["weights"])
    sd["hand_components"] = f32(np.asarray(d["hands_components"])[:num_pca_comps])
    hm = np.zeros_like(d["hands_mean"]) if flat_hand_mean else d["hands_mean"]
    sd["hand_mean"] = f32(hm)
    sd["pose_mean"] = torch.cat([torch.zeros(3), f32(hm)])
    return OrderedDict((prefix + k, v) for k, v in sd.items())


# ----------------------------------------------------------------------- inputs
def _euler_xyz(a):
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def object_clouds(B: int, N: int = 3000, seed: int = 1234, first: int = 0) -> torch.Tensor:
    """[B,4,N] fp32 object clouds in the layout the reference datasets produce
    (DVQ-VAE/dataset/HO3D_diversity_generation.py:50-54): xyz in metres + a 4th channel
    holding the object's bounding-box diagonal, repeated over the points.

    Object i (global index first+i) is a box with half-extents U(0.02,0.13) sampled on
    its surface, rotated by random Euler angles (Rx*Ry*Rz) and moved to the fixed
    offset the generation scripts use.  Each object has its own stream, so a shard
    [first, first+B) equals the same rows of the full batch."""
    out = np.empty((B, 4, N), np.float32)
    for i in range(B):
        r = _rng(seed, f"obj{first + i}")
        h = r.uniform(0.02, 0.13, size=3)
        p = r.uniform(-1.0, 1.0, size=(N, 3)) * h
        face = r.integers(0, 3, size=N)
        sign = r.integers(0, 2, size=N) * 2 - 1
        p[np.arange(N), face] = sign * h[face]
        diag = np.linalg.norm(p.max(0) - p.min(0))
        R = _euler_xyz(r.uniform(0, 2 * math.pi, size=3))
        p = p @ R.T + np.array(OBJ_TRANSLATION)
        out[i, :3] = p.T.astype(np.float32)
        out[i, 3] = np.float32(diag)
    return torch.from_numpy(out)


def vq_case(K: int, D: int, n: int, dist: str = "A", seed: int = 0):
    """Latents and codebook for the VQ-argmin cases.
    dist 'A': reference init, E ~ U(-1/K, 1/K) (quantizer.py:27), z ~ N(0, 0.01^2).
    dist 'B': trained-like, E ~ N(0,1), z = E[j] + N(0, 0.3^2) with j uniform."""
    r = _rng(seed, f"vq{K}x{D}{dist}")
    if dist == "A":
        E = r.uniform(-1.0 / K, 1.0 / K, size=(K, D)).astype(np.float32)
        z = (r.standard_normal(size=(n, D)) * 0.01).astype(np.float32)
    elif dist == "B":
        E = r.standard_normal(size=(K, D)).astype(np.float32)
        j = r.integers(0, K, size=n)
        z = (E[j] + 0.3 * r.standard_normal(size=(n, D))).astype(np.float32)
    else:
        raise ValueError(dist)
    return torch.from_numpy(z), torch.from_numpy(E)


def vq_near_tie_case(K: int, D: int, n: int, kind: str, seed: int = 0):
    """Latents built so that the two best codes are nearly equidistant, i.e. where the order in which the
    reference rounds its distance (quantizer.py:46-48: fl(fl(|z|^2 + |E|^2) - 2 z.E)) decides the index.
    kind 'mid'    : E ~ N(0,1); z = (E[a] + E[b]) / 2 + eps N(0,1), a != b.  The exact gap between codes a and b is
                    2 eps n.(E[b] - E[a]); eps = 1e-5 puts it around the fp32 spacing of |z|^2 + |E|^2 (~3e-5 near 430).
    kind 'bignorm': reference initialisation E ~ U(-1/K, 1/K) with |z|^2 ~ 4096 >> |E|^2 ~ D / (3 K^2): the code
                    norms and most of the products vanish below the spacing of |z|^2 (4.9e-4)."""
    r = _rng(seed, f"vqtie{K}x{D}{kind}")
    if kind == "mid":
        E = r.standard_normal(size=(K, D)).astype(np.float32)
        a = r.integers(0, K, size=n)
        b = (a + 1 + r.integers(0, K - 1, size=n)) % K
        mid = 0.5 * (E[a].astype(np.float64) + E[b].astype(np.float64))
        z = (mid + 1e-5 * r.standard_normal(size=(n, D))).astype(np.float32)
    elif kind == "bignorm":
        E = r.uniform(-1.0 / K, 1.0 / K, size=(K, D)).astype(np.float32)
        z = (4.0 * r.standard_normal(size=(n, D))).astype(np.float32)
    else:
        raise ValueError(kind)
    return torch.from_numpy(z), torch.from_numpy(E)


def mano_inputs(B: int, seed: int = 0):
    """(global_orient [B,3] ~U(-pi,pi), hand_pose PCA coeffs [B,45] ~N(0,2.3^2),
    transl [B,3] near the object offset) -- the spread of the parameter vectors the
    reference stores in DVQ-VAE-2/diverse_grasp/HMDO/obj_id_*.json."""
    r = _rng(seed, "mano_in")
    go = r.uniform(-math.pi, math.pi, size=(B, 3)).astype(np.float32)
    hp = (r.standard_normal(size=(B, 45)) * 2.3).astype(np.float32)
    tr = (np.array(OBJ_TRANSLATION) + r.normal(0, 0.03, size=(B, 3))).astype(np.float32)
    return torch.from_numpy(go), torch.from_numpy(hp), torch.from_numpy(tr)


def object_clouds_device(B: int, N: int, seed: int, device) -> torch.Tensor:
    """Same family of clouds as object_clouds(), generated on the device with torch's
    generator (fast enough for B = 4096; not bit-identical to the numpy recipe)."""
    g = torch.Generator(device).manual_seed(int(seed))
    h = torch.rand(B, 1, 3, device=device, generator=g) * 0.11 + 0.02
    p = (torch.rand(B, N, 3, device=device, generator=g) * 2 - 1) * h
    face = torch.randint(0, 3, (B, N, 1), device=device, generator=g)
    sign = torch.randint(0, 2, (B, N, 1), device=device, generator=g).float() * 2 - 1
    p.scatter_(2, face, sign * torch.gather(h.expand(B, N, 3), 2, face))
    diag = (p.amax(1) - p.amin(1)).norm(dim=1)
    a = torch.rand(B, 3, device=device, generator=g) * (2 * math.pi)
    cx, sx, cy, sy, cz, sz = a[:, 0].cos(), a[:, 0].sin(), a[:, 1].cos(), a[:, 1].sin(), a[:, 2].cos(), a[:, 2].sin()
    o, z = torch.ones_like(cx), torch.zeros_like(cx)
    Rx = torch.stack([o, z, z, z, cx, -sx, z, sx, cx], 1).view(B, 3, 3)
    Ry = torch.stack([cy, z, sy, z, o, z, -sy, z, cy], 1).view(B, 3, 3)
    Rz = torch.stack([cz, -sz, z, sz, cz, z, z, z, o], 1).view(B, 3, 3)
    R = Rx @ Ry @ Rz
    p = p @ R.transpose(1, 2) + torch.tensor(OBJ_TRANSLATION, device=device)
    out = torch.empty(B, 4, N, device=device)
    out[:, :3] = p.transpose(1, 2)
    out[:, 3] = diag[:, None]
    return out.contiguous()
