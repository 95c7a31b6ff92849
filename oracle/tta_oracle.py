"""CPU oracle for the test-time adaptation loop (refinement of generated grasps).

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE; only tests/ and the benchmark's CPU
baseline may import it.

It restates DVQ-VAE/gen_HDMO_TTA.py:119-152 -- SGD with momentum on the 61-d MANO parameter
vector against  loss = 1 * contact_loss + 5 * penetr_loss  (gen_HDMO_TTA.py:150), with
  penetr_loss, contact_loss = TTT_loss(...)           DVQ-VAE/utils/loss.py:104-127
  Contact_loss                                        DVQ-VAE/utils/loss.py:72-101
  get_NN / get_interior                               DVQ-VAE/utils/utils_loss.py:7-45
on top of the MANO forward of oracle/dvq_oracle.py, differentiated by torch autograd (which is
what the reference does).  The consistency term of TTT_loss needs the ContactNet module that
the reference does not ship, and gen_HDMO_TTA.py:150 leaves it out of the optimised loss; it is
not restated.

PARITY UNPINNED for two building blocks: the reference takes nearest neighbours and vertex
normals from pytorch3d (knn_points, Meshes.verts_normals_packed), which is absent from the
reference checkout and from this environment, and the reference holds no fixture for them.
They are restated here from their documented behaviour: squared Euclidean distance to the
nearest target point with the lowest index winning ties; vertex normal = normalised sum of the
cross products (area-weighted normals) of the incident faces.  The MANO gradient is pinned
through autograd of the pinned forward.

The reference runs with a batch of one.  For a batch this oracle treats every grasp as its own
problem (loss, contact-point count and optimiser state per row), which is the reference's
behaviour applied row by row.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

from . import dvq_oracle as O

LR = 0.00000625          # gen_HDMO_TTA.py:120
MOMENTUM = 0.8           # gen_HDMO_TTA.py:120
ITERS = 300              # gen_HDMO_TTA.py:133
CONTACT_DIST_SQ = 0.02 ** 2   # loss.py:126
W_PENETR, W_CONTACT = 5.0, 1.0   # gen_HDMO_TTA.py:150


def knn1(src: torch.Tensor, trg: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils_loss.py:7-24 with k=1: squared distance [B,N1] from every src point to its nearest
    trg point and that point's index [B,N1] (first minimum)."""
    d = ((src.unsqueeze(2) - trg.unsqueeze(1)) ** 2).sum(-1)        # [B,N1,N2]
    idx = d.argmin(dim=2)
    return d.gather(2, idx.unsqueeze(-1)).squeeze(-1), idx


def verts_normals(verts: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """Per-vertex normals [B,V,3] of a batch of meshes sharing `faces` [F,3] (loss.py:116-117)."""
    f = faces.long()
    v0, v1, v2 = verts[:, f[:, 0]], verts[:, f[:, 1]], verts[:, f[:, 2]]
    n = torch.zeros_like(verts)
    n = n.index_add(1, f[:, 1], torch.cross(v2 - v1, v0 - v1, dim=2))
    n = n.index_add(1, f[:, 2], torch.cross(v0 - v2, v1 - v2, dim=2))
    n = n.index_add(1, f[:, 0], torch.cross(v1 - v0, v2 - v0, dim=2))
    return torch.nn.functional.normalize(n, eps=1e-6, dim=2)


def ttt_loss(hand_xyz: torch.Tensor, faces: torch.Tensor, obj_xyz: torch.Tensor, contact_verts: Sequence[int],
             guard_empty: bool = True) -> Dict[str, torch.Tensor]:
    """Per-grasp penetration and contact terms [B] (loss.py:104-127 with B = 1 per row), and the
    nearest-neighbour data they were built from.  With guard_empty a grasp without any contact
    point gets contact = 0 where the reference would divide 0 by 0."""
    normal = verts_normals(hand_xyz, faces)
    nn_dist, nn_idx = knn1(obj_xyz, hand_xyz)
    gi = nn_idx.unsqueeze(-1).expand(-1, -1, 3)
    vec = hand_xyz.gather(1, gi) - obj_xyz                           # utils_loss.py:38-39
    interior = (vec * normal.gather(1, gi)).sum(-1) > 0              # utils_loss.py:42-44
    penetr = 120 * (nn_dist * interior).sum(1)                       # loss.py:120, B = 1
    cmap = nn_dist < CONTACT_DIST_SQ                                 # loss.py:126
    prior = hand_xyz[:, list(contact_verts)]                         # loss.py:93-94
    cd, cd_idx = knn1(obj_xyz, prior)                                # loss.py:98
    n_points = cmap.sum(1)
    s = (cd * cmap).sum(1)
    if guard_empty:
        contact = torch.where(n_points > 0, s / n_points.clamp(min=1), torch.zeros_like(s))
    else:
        contact = s / n_points
    contact = 2.5 * 3000.0 * contact                                 # loss.py:100-101, :126
    return dict(penetr=penetr, contact=contact, nn_dist=nn_dist, nn_idx=nn_idx, interior=interior, cmap=cmap,
                contact_idx=cd_idx)


def loss_and_grad(sd: O.SD, faces, obj_xyz, param, contact_verts, mano_prefix: str = ""):
    """loss [B], d loss / d param [B,61], and the loss pieces, for param = [betas10|orient3|pose45|transl3]."""
    p = param.detach().clone().requires_grad_(True)
    verts, _, _ = O.mano_forward(sd, p[:, 10:13], p[:, 13:58], p[:, 58:61], p=mano_prefix)
    parts = ttt_loss(verts, faces, obj_xyz, contact_verts)
    loss = W_CONTACT * parts["contact"] + W_PENETR * parts["penetr"]
    grad_v, = torch.autograd.grad(loss.sum(), verts, retain_graph=True)
    grad_p, = torch.autograd.grad(loss.sum(), p)
    parts.update(loss=loss.detach(), grad_param=grad_p, grad_verts=grad_v, verts=verts.detach())
    return parts


def refine(sd: O.SD, faces, obj_xyz, param, contact_verts, iters: int = ITERS, lr: float = LR,
           momentum: float = MOMENTUM, mano_prefix: str = ""):
    """gen_HDMO_TTA.py:119-152: returns the adapted parameters [B,61] and the per-iteration
    (penetr, contact) values [iters,B,2]."""
    p = param.detach().clone().requires_grad_(True)
    opt = torch.optim.SGD([p], lr=lr, momentum=momentum)
    hist = []
    for _ in range(iters):
        opt.zero_grad()
        verts, _, _ = O.mano_forward(sd, p[:, 10:13], p[:, 13:58], p[:, 58:61], p=mano_prefix)
        parts = ttt_loss(verts, faces, obj_xyz, contact_verts)
        (W_CONTACT * parts["contact"] + W_PENETR * parts["penetr"]).sum().backward()
        opt.step()
        hist.append(torch.stack([parts["penetr"].detach(), parts["contact"].detach()], 1))
    return p.detach(), torch.stack(hist)
