"""Test-time adaptation of generated grasps on the GPU.

The reference refines every generated grasp with 300 steps of SGD (momentum 0.8, lr 6.25e-6)
on its 61-d MANO parameter vector against  1 * contact + 5 * penetration
(DVQ-VAE/gen_HDMO_TTA.py:119-152; loss terms DVQ-VAE/utils/loss.py:72-127,
utils/utils_loss.py:7-45), one grasp at a time through autograd.  GraspRefiner runs the same
iteration for a whole batch of grasps with three launches per step -- MANO forward, the fused
nearest-neighbour / contact / penetration kernel, MANO backward -- plus the momentum update, every
grasp being its own problem (own loss, own contact-point count, own optimiser state), i.e. the
reference's batch of one applied row by row.

Not reproduced: the ContactNet consistency term.  The reference computes it but leaves it out of
the optimised loss (gen_HDMO_TTA.py:150), and the module that produces its target is not shipped.
A grasp with no object point within 2 cm of the hand gets contact = 0 here; the reference divides
0 by 0 there and the parameters become NaN.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import DvqHandMesh
from .hand_parts import CONTACT_VERTS

LR = 0.00000625               # gen_HDMO_TTA.py:120
MOMENTUM = 0.8                # gen_HDMO_TTA.py:120
ITERS = 300                   # gen_HDMO_TTA.py:133
CONTACT_DIST_SQ = 0.02 ** 2   # loss.py:126
PENETR_SCALE = 120.0          # loss.py:120
CONTACT_SCALE = 2.5 * 3000.0  # loss.py:101,126
W_PENETR, W_CONTACT = 5.0, 1.0  # gen_HDMO_TTA.py:150


def hand_mesh(faces, n_verts: int, contact_verts: Sequence[int], device):
    """Device tables of a hand mesh for ops.tta_contact_grad: faces [F,3], the faces incident to
    each vertex (CSR) and the search order (contact set first).  Returns (DvqHandMesh, tensors to keep alive)."""
    f = np.asarray(faces.detach().cpu().numpy() if torch.is_tensor(faces) else faces, dtype=np.int64).reshape(-1, 3)
    if f.size and (f.min() < 0 or f.max() >= n_verts):
        raise ValueError("face index outside the vertex range")
    vert = f.reshape(-1)
    face = np.repeat(np.arange(f.shape[0]), 3)
    order = np.argsort(vert, kind="stable")
    ptr = np.zeros(n_verts + 1, dtype=np.int64)
    np.add.at(ptr, vert + 1, 1)
    ptr = np.cumsum(ptr)
    cv = np.unique(np.asarray(list(contact_verts), dtype=np.int64))
    if cv.size == 0 or cv.min() < 0 or cv.max() >= n_verts:
        raise ValueError("the contact set must name at least one vertex of the mesh")
    rest = np.setdiff1d(np.arange(n_verts), cv)
    search_order = np.concatenate([cv, rest])          # contact set first, each group ascending
    keep = dict(faces=torch.tensor(f, dtype=torch.int32, device=device).contiguous(),
                ptr=torch.tensor(ptr, dtype=torch.int32, device=device),
                vf=torch.tensor(face[order], dtype=torch.int32, device=device),
                order=torch.tensor(search_order, dtype=torch.int32, device=device))
    if keep["vf"].numel() == 0:
        keep["vf"] = torch.zeros(1, dtype=torch.int32, device=device)
        keep["faces"] = torch.zeros(1, 3, dtype=torch.int32, device=device)
    mesh = DvqHandMesh(int(n_verts), int(f.shape[0]), int(cv.size), keep["faces"].data_ptr(), keep["ptr"].data_ptr(),
                       keep["vf"].data_ptr(), keep["order"].data_ptr())
    return mesh, keep


def read_closed_faces(path: str) -> np.ndarray:
    """The closed hand mesh the reference's adaptation script reads (models/mano/closed_mano_faces.pkl,
    gen_HDMO_TTA.py:61-62): an integer [F,3] array (1552 faces: the model's 1538 plus 14 over the wrist), from a pickle
    file read as data (dvq_hip.mano_io: nothing the file names is imported or called) or from a .npy file."""
    from . import mano_io
    f = np.load(path, allow_pickle=False) if path.endswith(".npy") else mano_io.read_pickled_arrays(path)
    f = np.asarray(f)
    if f.ndim != 2 or f.shape[1] != 3 or not np.issubdtype(f.dtype, np.integer):
        raise ValueError(f"{path}: expected an integer [F,3] face array, got {f.dtype} {f.shape}")
    return f.astype(np.int64)


class GraspRefiner:
    """refine(obj_pc [B,4,N], recon_param [B,61]) -> adapted parameters, as gen_HDMO_TTA.py:119-152."""

    def __init__(self, rh_mano, faces=None, contact_verts: Sequence[int] = CONTACT_VERTS, lr: float = LR,
                 momentum: float = MOMENTUM, iters: int = ITERS, w_penetr: float = W_PENETR, w_contact: float = W_CONTACT,
                 use_graph: bool = True):
        """`faces`: the closed hand mesh the reference takes vertex normals from
        (models/mano/closed_mano_faces.pkl, gen_HDMO_TTA.py:61-62), [F,3]; default: the MANO model's own faces."""
        self.rh_mano = rh_mano
        self.lr, self.momentum, self.iters = float(lr), float(momentum), int(iters)
        self.w_penetr, self.w_contact = float(w_penetr), float(w_contact)
        self.use_graph = use_graph
        dev = rh_mano.v_template.device
        self.n_verts = int(rh_mano.v_template.shape[0])
        self.mesh, self._keep = hand_mesh(faces if faces is not None else rh_mano.faces_tensor, self.n_verts, contact_verts, dev)
        self._state = {}

    # ------------------------------------------------------------------ one evaluation
    @torch.no_grad()
    def loss_and_grad(self, obj: torch.Tensor, param: torch.Tensor, want_nn: bool = False):
        """(penetr, contact) [B,2], d(1*contact + 5*penetr)/d param [B,61] (zero for the ten shape
        coefficients, which the MANO layer ignores), and the intermediate tensors."""
        consts = self.rh_mano._device_consts()
        go, hp, tr = (param[:, 10:13].contiguous(), param[:, 13:58].contiguous(), param[:, 58:61].contiguous())
        verts, _, _, vp = ops.mano_lbs(go, hp, tr, consts, self.n_verts, False, False, want_v_posed=True)
        nn = torch.empty(obj.shape[0], obj.shape[2], dtype=torch.int32, device=obj.device) if want_nn else None
        gv, losses = ops.tta_contact_grad(obj, verts, self.mesh, CONTACT_DIST_SQ, PENETR_SCALE, CONTACT_SCALE,
                                          self.w_penetr, self.w_contact, nn_idx=nn)
        g_go, g_hp, g_tr = ops.mano_lbs_backward(go, hp, consts, vp, gv)
        grad = torch.zeros_like(param)
        grad[:, 10:13], grad[:, 13:58], grad[:, 58:61] = g_go, g_hp, g_tr
        return dict(losses=losses, grad_param=grad, grad_verts=gv, verts=verts, nn_idx=nn)

    # ------------------------------------------------------------------ the loop
    def _buffers(self, B, N, dev):
        key = (B, N, str(dev))
        if key not in self._state:
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            ncomp = int(self.rh_mano.hand_components.shape[0]) if self.rh_mano.use_pca else 45
            self._state = {key: dict(p=[z(B, 3), z(B, ncomp), z(B, 3)], buf=[z(B, 3), z(B, ncomp), z(B, 3)],
                                     g=[z(B, 3), z(B, ncomp), z(B, 3)], verts=z(B, self.n_verts, 3), vp=z(B, self.n_verts, 3),
                                     gv=z(B, self.n_verts, 3), losses=z(B, 2), obj=None, graph=None)}
        return self._state[key]

    def _step(self, st):
        consts = self.rh_mano._device_consts()
        go, hp, tr = st["p"]
        ops.mano_lbs(go, hp, tr, consts, self.n_verts, False, False, want_v_posed=True, verts=st["verts"], v_posed=st["vp"])
        ops.tta_contact_grad(st["obj"], st["verts"], self.mesh, CONTACT_DIST_SQ, PENETR_SCALE, CONTACT_SCALE, self.w_penetr,
                             self.w_contact, grad_verts=st["gv"], losses=st["losses"])
        ops.mano_lbs_backward(go, hp, consts, st["vp"], st["gv"], out=tuple(st["g"]))
        # torch.optim.SGD with momentum: buf = momentum * buf + grad (buf = grad on the first step); p -= lr * buf
        torch._foreach_mul_(st["buf"], self.momentum)
        torch._foreach_add_(st["buf"], st["g"])
        torch._foreach_add_(st["p"], st["buf"], alpha=-self.lr)

    @torch.no_grad()
    def refine(self, obj: torch.Tensor, param: torch.Tensor, iters: Optional[int] = None):
        """obj [B,C>=3,N] float32 object clouds as passed to gen(); param [B,61].  Returns
        dict(recon_param [B,61], first [B,2], last [B,2]): the adapted parameters and the
        (penetration, contact) values of the first and the last iteration (the two the reference
        prints, gen_HDMO_TTA.py:153-158)."""
        iters = self.iters if iters is None else int(iters)
        if not obj.is_cuda or not param.is_cuda:
            raise RuntimeError("GraspRefiner needs tensors on a HIP device; there is no CPU fallback")
        B, N = obj.shape[0], obj.shape[2]
        st = self._buffers(B, N, obj.device)
        if st["obj"] is None or st["obj"].shape != obj.shape:
            st["obj"] = torch.empty_like(obj, memory_format=torch.contiguous_format)
            st["graph"] = None
        # a recorded graph holds the hand layer's constant tables by address and the step sizes and loss weights by
        # value: drop it when any of them has changed since the recording
        self.rh_mano._device_consts()
        baked = (self.rh_mano._consts_key, self.lr, self.momentum, self.w_penetr, self.w_contact)
        if st.get("baked") != baked:
            st["graph"], st["baked"] = None, baked
        st["obj"].copy_(obj)
        for dst, src in zip(st["p"], (param[:, 10:13], param[:, 13:58], param[:, 58:61])):
            dst.copy_(src)
        for b in st["buf"]:
            b.zero_()
        first = last = None
        if iters > 0 and B > 0:        # (an empty batch -- a rank without objects -- has nothing to adapt or record)
            if self.use_graph and st["graph"] is None and iters > 2:
                saved = [t.clone() for t in st["p"] + st["buf"]]
                self._step(st)                                   # warm-up outside capture
                torch.cuda.synchronize(obj.device)
                for t, s in zip(st["p"] + st["buf"], saved):
                    t.copy_(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._step(st)
                st["graph"] = g
                for t, s in zip(st["p"] + st["buf"], saved):     # capture does not execute, but keep the state exact
                    t.copy_(s)
            for it in range(iters):
                if self.use_graph and st["graph"] is not None:
                    st["graph"].replay()
                else:
                    self._step(st)
                if it == 0:
                    first = st["losses"].clone()
            last = st["losses"].clone()
        elif iters > 0:
            first, last = st["losses"].clone(), st["losses"].clone()      # [0,2]
        out = param.clone()
        out[:, 10:13], out[:, 13:58], out[:, 58:61] = st["p"]
        return dict(recon_param=out, first=first, last=last)
