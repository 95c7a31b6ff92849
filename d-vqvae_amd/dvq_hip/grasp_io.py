"""Host-side data formats either side of the generation path.

Inputs: the [4, N] object cloud the reference's datasets hand to model.gen (three
coordinates + the object's bounding-box diagonal repeated over the points,
DVQ-VAE/dataset/HO3D_diversity_generation.py:47-56, dataset/utils_HO3D_FPHA.py:75-84),
and the random rigid placement the generation scripts apply to it before every call
(DVQ-VAE/gen_diverse_grasp_ho3d.py:212-228).

Outputs: the per-object JSON those scripts write and the reference's evaluation reads
(gen_diverse_grasp_ho3d.py:322-331; diverse_grasp/diversity.py:33-41):
    {"recon_params": [[61 floats]] per grasp, "R_list": [3x4] per grasp,
     "trans_list": [3x1] per grasp, "r_list": [3 Euler angles] per grasp}

The reference generates the 100 placements of an object one call at a time; here they are one
batch through GraspGenerator.  Nothing in this file touches a kernel.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# the constant translation of gen_diverse_grasp_ho3d.py:221 (its random part is multiplied by 0)
DEFAULT_TRANSLATION = np.array([-0.0793, 0.0208, -0.6924])


# ------------------------------------------------------------------ inputs
def get_diameter(vp: np.ndarray) -> float:
    """Bounding-box diagonal of a vertex array [V,3] (utils_HO3D_FPHA.py:75-84)."""
    vp = np.asarray(vp)
    ext = vp.max(axis=0) - vp.min(axis=0)
    return float(np.sqrt((np.abs(ext) ** 2).sum()))


def make_obj_pc(points: np.ndarray, scale: float, n_point: int = 3000) -> torch.Tensor:
    """[4, n_point] float32: the first n_point resampled surface points and the object's
    scale as a constant fourth channel (HO3D_diversity_generation.py:50-54)."""
    pts = np.asarray(points)
    if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < n_point:
        raise ValueError(f"need at least {n_point} points of shape [*,3], got {pts.shape}")
    verts = torch.tensor(pts[:n_point], dtype=torch.float32)
    ch = torch.tensor(scale).type_as(verts).repeat(n_point, 1)
    return torch.cat((verts, ch), dim=-1).permute(1, 0).contiguous()


def load_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Vertices [V,3] and triangle faces [F,3] (0-based) of a Wavefront .obj file; polygons
    are fanned into triangles.  Stands in for the reference's utils.fast_load_obj
    (dataset/utils_HO3D_FPHA.py:13), whose other outputs the path does not use."""
    verts: List[List[float]] = []
    faces: List[List[int]] = []
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v" and len(tok) >= 4:
                verts.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif tok[0] == "f" and len(tok) >= 4:
                ids = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(ids) - 1):
                    faces.append([ids[0], ids[k], ids[k + 1]])
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def sample_surface(verts: np.ndarray, faces: np.ndarray, count: int, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """`count` points uniformly on the surface of a triangle mesh (triangle chosen in proportion
    to its area, then uniform barycentric coordinates) and the face each came from.  The reference
    uses trimesh.sample.sample_surface for this (utils_HO3D_FPHA.py:61-66); trimesh is not
    available here, so the points are a same-distribution replacement, not the same stream."""
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    if not np.isfinite(area).all() or area.sum() <= 0:
        raise ValueError("mesh has no surface area")
    g = np.random.Generator(np.random.PCG64(seed))
    face_id = np.searchsorted(np.cumsum(area) / area.sum(), g.random(count), side="right").clip(0, len(f) - 1)
    r1, r2 = g.random(count), g.random(count)
    flip = r1 + r2 > 1.0
    r1[flip], r2[flip] = 1.0 - r1[flip], 1.0 - r2[flip]
    pts = a[face_id] + r1[:, None] * (b[face_id] - a[face_id]) + r2[:, None] * (c[face_id] - a[face_id])
    return pts, face_id


def obj_pc_from_mesh(path: str, n_point: int = 3000, seed: int = 0) -> Tuple[torch.Tensor, np.ndarray, np.ndarray]:
    """obj_pc [4,n_point] for a mesh file the way load_objects_HO3D builds it
    (utils_HO3D_FPHA.py:12-19): scale from the mesh vertices, points from the surface.  A
    `resampled.npy` next to the mesh, as the reference's preprocessing leaves it, is used if present."""
    verts, faces = load_obj(path)
    cached = os.path.join(os.path.dirname(path), "resampled.npy")
    if os.path.exists(cached):
        pts = np.load(cached, allow_pickle=False)
    else:
        pts, _ = sample_surface(verts, faces, n_point, seed)
    return make_obj_pc(pts, get_diameter(verts), n_point), verts, faces


# ------------------------------------------------------------------ placements
def euler_rotation(theta: Sequence[float]) -> np.ndarray:
    """Rx(theta_x) @ Ry(theta_y) @ Rz(theta_z), float64 (gen_diverse_grasp_ho3d.py:215-219)."""
    tx, ty, tz = (float(t) for t in theta)
    rx = np.array([[1, 0, 0], [0, np.cos(tx), -np.sin(tx)], [0, np.sin(tx), np.cos(tx)]])
    ry = np.array([[np.cos(ty), 0, np.sin(ty)], [0, 1, 0], [-np.sin(ty), 0, np.cos(ty)]])
    rz = np.array([[np.cos(tz), -np.sin(tz), 0], [np.sin(tz), np.cos(tz), 0], [0, 0, 1]])
    return rx @ ry @ rz


def random_placements(n: int, rng: Optional[np.random.RandomState] = None) -> Dict[str, np.ndarray]:
    """n random placements drawn exactly as the reference's loop draws them: per grasp three
    uniforms for the Euler angles, then three more for the (zero-weighted) translation jitter
    (gen_diverse_grasp_ho3d.py:214,221), from numpy's legacy global stream or `rng`.
    Returns r [n,3] angles, trans [n,3,1], R [n,3,4] = [rot | trans]."""
    rs = rng if rng is not None else np.random
    r = np.empty((n, 3))
    trans = np.empty((n, 3, 1))
    R = np.empty((n, 3, 4))
    for i in range(n):
        r[i] = rs.random_sample(3) * np.pi * 2
        t = DEFAULT_TRANSLATION + rs.random_sample(3) * 0.2 * 0
        trans[i] = t.reshape(3, 1)
        R[i] = np.hstack((euler_rotation(r[i]), trans[i]))
    return dict(r=r, trans=trans, R=R)


def place_clouds(obj_pc: torch.Tensor, R: np.ndarray) -> torch.Tensor:
    """[n,4,N]: the cloud obj_pc [4,N] under each rigid transform R[i] = [rot | trans]; the
    scale channel is copied (gen_diverse_grasp_ho3d.py:224-228).  The product is taken in
    float64 on the host and rounded once to float32, as the reference does through numpy."""
    xyz = obj_pc[:3].detach().cpu().numpy().astype(np.float64)             # [3,N] (float32 values)
    R = np.asarray(R, dtype=np.float64)
    moved = np.matmul(R[:, :, :3], xyz[None]) + R[:, :, 3:4]               # [n,3,N]
    out = obj_pc.detach().cpu().unsqueeze(0).repeat(R.shape[0], 1, 1).clone()
    out[:, :3, :] = torch.tensor(moved, dtype=torch.float32)
    return out


# ------------------------------------------------------------------ outputs
def grasp_record(recon_param: torch.Tensor, placements: Dict[str, np.ndarray]) -> Dict[str, list]:
    """The dictionary the reference dumps per object (gen_diverse_grasp_ho3d.py:315-329).  Each
    grasp's parameters are a [1,61] nested list, as produced there by a batch of one."""
    p = recon_param.detach().cpu().numpy()
    n = p.shape[0]
    if any(len(placements[k]) != n for k in ("R", "trans", "r")):
        raise ValueError("placements and parameters disagree on the number of grasps")
    return {
        "recon_params": [p[i:i + 1].tolist() for i in range(n)],
        "R_list": [placements["R"][i].tolist() for i in range(n)],
        "trans_list": [placements["trans"][i].tolist() for i in range(n)],
        "r_list": [np.asarray(placements["r"][i]).tolist() for i in range(n)],
    }


def write_grasp_json(path: str, record: Dict[str, list]) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(record, f)


def read_grasp_json(path: str) -> Dict[str, np.ndarray]:
    """recon_params [n,61], R [n,3,4], trans [n,3,1], r [n,3] from a file written by the
    reference's scripts or by write_grasp_json."""
    with open(path, "r") as f:
        d = json.load(f)
    return dict(recon_params=np.asarray([g[0] for g in d["recon_params"]], dtype=np.float32).reshape(-1, 61),
                R=np.asarray(d["R_list"], dtype=np.float64).reshape(-1, 3, 4),
                trans=np.asarray(d["trans_list"], dtype=np.float64).reshape(-1, 3, 1),
                r=np.asarray(d["r_list"], dtype=np.float64).reshape(-1, 3))


def diverse_grasps(generator, obj_pc: torch.Tensor, n_grasps: int = 100, rng: Optional[np.random.RandomState] = None,
                   first_id: int = 0, device=None, refiner=None) -> Tuple[Dict[str, list], torch.Tensor]:
    """The body of the reference's per-object loop (gen_diverse_grasp_ho3d.py:212-257,315-318) as
    one batch: n_grasps random placements of obj_pc [4,N] -> gen -> parameter vector -> MANO.
    Returns (JSON-ready record, vertices [n,778,3] on the device).  `generator` is a
    dvq_hip.pipeline.GraspGenerator; grasp i uses the sampling stream of object id first_id+i.
    With `refiner` (a dvq_hip.tta.GraspRefiner) every grasp is adapted to its cloud before it is recorded, as the
    reference's adaptation script does between generation and the final hand (gen_HDMO_TTA.py:113-163)."""
    pl = random_placements(n_grasps, rng)
    clouds = place_clouds(obj_pc, pl["R"])
    dev = device if device is not None else next(generator.model.parameters()).device
    clouds = clouds.to(dev)
    out = generator.generate(clouds, first_id=first_id)
    param, verts = out["recon_param"], out["vertices"]
    if refiner is not None:
        param = refiner.refine(clouds, param)["recon_param"]
        verts = generator.rh_mano(betas=param[:, :10], global_orient=param[:, 10:13], hand_pose=param[:, 13:58],
                                  transl=param[:, 58:]).vertices
    return grasp_record(param, pl), verts


def diversity(params: np.ndarray, cls_num: int = 20, seed: Optional[int] = None) -> Tuple[float, float]:
    """Entropy of the k-means cluster occupancy and mean distance to the assigned centre
    (diverse_grasp/diversity.py:7-15).  scipy's k-means draws its initial centres from numpy's
    global stream unless `seed` is given."""
    import scipy.cluster.vq
    from scipy.stats import entropy
    x = np.asarray(params, dtype=np.float64)
    codes, _ = scipy.cluster.vq.kmeans(x, cls_num, seed=seed)
    vecs, dist = scipy.cluster.vq.vq(x, codes)
    counts, _ = np.histogram(vecs, len(codes))
    return float(entropy(counts)), float(np.mean(dist))
