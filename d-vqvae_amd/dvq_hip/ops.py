"""Tensor-level wrappers over the C ABI: device pointers + torch's current HIP stream in,
status checked, outputs returned as fresh torch tensors on the input's device.

torch is used here for memory and streams only.  Every function requires its tensors
to live on a HIP device and raises otherwise -- there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import DVQ_BF16, DVQ_F32, DvqHandMesh, DvqManoConsts, DvqPointMlp, DvqPriorGather, DvqPriorJob, DvqPriorVRow


# Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
# roofline figure): TIMERS[name] collects (start_event, end_event, work, on_default_stream) per launch.
TIMERS = None


def enable_timers(on: bool = True):
    global TIMERS
    TIMERS = {} if on else None


class _timed:
    def __init__(self, name, work, ref):
        self.name, self.work, self.ref = name, work, ref

    def __enter__(self):
        # (no event pairs while a HIP graph is being recorded: they would become nodes of the graph)
        self.on = TIMERS is not None and not torch.cuda.is_current_stream_capturing()
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.ref.device))

    def __exit__(self, *exc):
        if self.on and TIMERS is not None:
            self.e1.record(torch.cuda.current_stream(self.ref.device))
            dev = self.ref.device
            TIMERS.setdefault(self.name, []).append((self.e0, self.e1, self.work,
                                                     torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev)))


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("dvq_hip kernels need tensors on a HIP device (got a CPU tensor); "
                               "there is no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError("dvq_hip kernels need contiguous tensors")


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DVQ_F32
    if t.dtype == torch.bfloat16:
        return DVQ_BF16
    raise RuntimeError(f"unsupported dtype {t.dtype}: float32 or bfloat16 expected")


# --------------------------------------------------------------------------- VQ
def vq_workspace(K: int, D: int, dtype: torch.dtype, device) -> torch.Tensor:
    code = DVQ_F32 if dtype == torch.float32 else DVQ_BF16
    nbytes = _lib.load().dvq_vq_workspace_bytes(K, D, code)
    if nbytes == 0:
        _lib.check(-1, "dvq_vq_workspace_bytes")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def vq_prepare(E: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Re-lay the codebook E [K,D] for the search kernel; returns the filled workspace."""
    _need_gpu(E)
    K, D = E.shape
    ws = workspace if workspace is not None else vq_workspace(K, D, E.dtype, E.device)
    with torch.cuda.device(E.device):
        _lib.check(_lib.load().dvq_vq_prepare(_p(E), K, D, _dtype_code(E), _p(ws), _stream(E)), "dvq_vq_prepare")
    return ws


def vq_argmin(z: torch.Tensor, E: torch.Tensor, want_zq: bool = True,
              prepared: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """idx int64 [n], z_q [n,D] (or None) for latents z [n,D] against codebook E [K,D]."""
    _need_gpu(z, E, prepared)
    if z.dtype != E.dtype:
        raise RuntimeError("vq_argmin: z and E must have the same dtype")
    n, D = z.shape
    K = E.shape[0]
    if E.shape[1] != D:
        raise RuntimeError(f"vq_argmin: z has {D} dims, codebook {E.shape[1]}")
    idx = torch.empty(n, dtype=torch.int64, device=z.device)
    zq = torch.empty_like(z) if want_zq else None
    lib = _lib.load()
    with torch.cuda.device(z.device):
        if prepared is None:
            ws = vq_workspace(K, D, E.dtype, E.device)
            rc = lib.dvq_vq_argmin(_p(z), _p(E), n, K, D, _dtype_code(z), _p(idx), _p(zq), _p(ws), _stream(z))
        else:
            rc = lib.dvq_vq_argmin_prepared(_p(z), _p(E), n, K, D, _dtype_code(z), _p(idx), _p(zq), _p(prepared), _stream(z))
    _lib.check(rc, "dvq_vq_argmin")
    return idx, zq


def vq_argmin_static(z: torch.Tensor, E: torch.Tensor, idx: torch.Tensor, zq: Optional[torch.Tensor] = None) -> None:
    """vq_argmin into caller-provided outputs (idx int64 [n], zq [n,D] or None) with the library's built-in
    workspace: no device memory is allocated by this call or below it.  Raises for codebooks beyond the built-in
    workspace.  Calls that overlap on one GPU share that workspace and race."""
    _need_gpu(z, E, idx, zq)
    n, D = z.shape
    K = E.shape[0]
    if z.dtype != E.dtype or E.shape[1] != D or idx.dtype != torch.int64 or idx.numel() != n:
        raise RuntimeError("vq_argmin_static: z [n,D] and E [K,D] of one dtype, idx int64 [n] expected")
    if zq is not None and (zq.dtype != z.dtype or tuple(zq.shape) != (n, D)):
        raise RuntimeError("vq_argmin_static: zq must match z")
    with torch.cuda.device(z.device):
        rc = _lib.load().dvq_vq_argmin_static(_p(z), _p(E), n, K, D, _dtype_code(z), _p(idx), _p(zq), _stream(z))
    _lib.check(rc, "dvq_vq_argmin_static")


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def vq_argmin_batched(zs, Es, prepared) -> list:
    """Indices (int64 [n] each) of several searches of one shape -- own latents, codebook and prepared workspace
    each -- in one launch (up to 8)."""
    _need_gpu(*zs, *Es, *prepared)
    n, D = zs[0].shape
    K = Es[0].shape[0]
    for z, E in zip(zs, Es):
        if z.shape != (n, D) or E.shape != (K, D) or z.dtype != Es[0].dtype or E.dtype != Es[0].dtype or not z.is_contiguous():
            raise RuntimeError("vq_argmin_batched: all searches must have one shape and dtype, contiguous latents")
    idx = [torch.empty(n, dtype=torch.int64, device=zs[0].device) for _ in zs]
    with torch.cuda.device(zs[0].device):
        rc = _lib.load().dvq_vq_argmin_prepared_batched(len(zs), _ptr_array(zs), _ptr_array(Es), n, K, D, _dtype_code(zs[0]),
                                                        _ptr_array(idx), _ptr_array(prepared), _stream(zs[0]))
    _lib.check(rc, "dvq_vq_argmin_prepared_batched")
    return idx


def vq_gather(idx: torch.Tensor, E: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = E[idx[b]] (fp32).  `out` may be a column slice of a wider row-major buffer."""
    _need_gpu(idx, E)
    if idx.dtype != torch.int64 or E.dtype != torch.float32:
        raise RuntimeError("vq_gather: idx must be int64 and E float32")
    n = idx.numel()
    K, D = E.shape
    if out is None:
        out = torch.empty(n, D, dtype=torch.float32, device=E.device)
    if not out.is_cuda or out.stride(1) != 1 or out.shape != (n, D):
        raise RuntimeError("vq_gather: bad output view")
    with torch.cuda.device(E.device):
        rc = _lib.load().dvq_vq_gather(_p(idx), _p(E), n, K, D, _p(out), out.stride(0) if n > 1 else max(D, out.stride(0)), _stream(E))
    _lib.check(rc, "dvq_vq_gather")
    return out


# --------------------------------------------------------------------- PointNet
def fold_bn(W, b, gamma, beta, mean, var, eps: float = 1e-5):
    """(A, c) with the eval-mode BatchNorm folded into the affine layer (W [out,in])."""
    _need_gpu(W, b, gamma, beta, mean, var)
    out_f, in_f = W.shape
    A = torch.empty_like(W)
    c = torch.empty_like(b)
    with torch.cuda.device(W.device):
        rc = _lib.load().dvq_fold_bn(_p(W), _p(b), _p(gamma), _p(beta), _p(mean), _p(var), eps, out_f, in_f,
                                     _p(A), _p(c), _stream(W))
    _lib.check(rc, "dvq_fold_bn")
    return A, c


def pointnet_pack_w3(a3: torch.Tensor, dtype_code: int) -> torch.Tensor:
    _need_gpu(a3)
    nbytes = _lib.load().dvq_pointnet_w3_bytes(dtype_code)
    out = torch.empty(nbytes, dtype=torch.uint8, device=a3.device)
    with torch.cuda.device(a3.device):
        rc = _lib.load().dvq_pointnet_pack_w3(_p(a3), dtype_code, _p(out), _stream(a3))
    _lib.check(rc, "dvq_pointnet_pack_w3")
    return out


def is_point_major(x: torch.Tensor) -> bool:
    """x [B,C,N] is a view of a contiguous [B,N,C] buffer (and not also contiguous as [B,C,N])."""
    B, C, N = x.shape
    return x.dim() == 3 and not x.is_contiguous() and x.stride(1) == 1 and x.stride(2) == C and (B <= 1 or x.stride(0) == C * N)


def pointnet_mlp_max(x: torch.Tensor, trans: Optional[torch.Tensor], w: dict, relu3: bool, dtype_code: int) -> torch.Tensor:
    """Shared per-point MLP C->64->128->1024 + max over points.  x [B,C,N] fp32 ->
    [B,1024] fp32; x is either contiguous (channel-major clouds, the reference's layout) or a permuted view of a
    contiguous [B,N,C] buffer (point-major, e.g. a vertex array), read in place.  `w` holds the folded tensors
    a1,c1,a2,c2,c3 and the packed w3p."""
    pm = is_point_major(x)
    _need_gpu(None if pm else x, trans, w["a1"], w["c1"], w["a2"], w["c2"], w["w3p"], w["c3"])
    if pm and not x.is_cuda:
        raise RuntimeError("dvq_hip kernels need tensors on a HIP device (got a CPU tensor); there is no CPU fallback")
    B, C, N = x.shape
    feat = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
    s = DvqPointMlp(w["a1"].data_ptr(), w["c1"].data_ptr(), w["a2"].data_ptr(), w["c2"].data_ptr(),
                    w["w3p"].data_ptr(), w["c3"].data_ptr())
    with torch.cuda.device(x.device), _timed(f"pointnet_mlp_max[C={C},N={N}]", (B, C, N), x):
        rc = _lib.load().dvq_pointnet_mlp_max(_p(x), _p(trans), B, C, N, int(pm), ctypes.byref(s), int(relu3), dtype_code,
                                              _p(feat), _stream(x))
    _lib.check(rc, "dvq_pointnet_mlp_max")
    return feat


def pointnet_mlp_max_grouped(groups, dtype_code: int) -> list:
    """Several pointnet_mlp_max calls of one channel count as one launch (up to 8).
    groups: list of (x [B,C,N], trans or None, w dict, relu3); returns the [B,1024] features."""
    structs = (_lib.DvqPointGroup * len(groups))()
    feats = []
    C = groups[0][0].shape[1]
    for i, (x, trans, w, relu3) in enumerate(groups):
        _need_gpu(x, trans, w["a1"], w["c1"], w["a2"], w["c2"], w["w3p"], w["c3"])
        if x.shape[1] != C or not x.is_contiguous() or (trans is not None and not trans.is_contiguous()):
            raise RuntimeError("pointnet_mlp_max_grouped: one channel count for all groups, contiguous tensors")
        B, _, N = x.shape
        feat = torch.empty(B, 1024, dtype=torch.float32, device=x.device)
        feats.append(feat)
        structs[i] = _lib.DvqPointGroup(x.data_ptr(), None if trans is None else trans.data_ptr(), B, N,
                                        DvqPointMlp(w["a1"].data_ptr(), w["c1"].data_ptr(), w["a2"].data_ptr(), w["c2"].data_ptr(),
                                                    w["w3p"].data_ptr(), w["c3"].data_ptr()), int(relu3), feat.data_ptr())
    x0 = groups[0][0]
    with torch.cuda.device(x0.device):
        rc = _lib.load().dvq_pointnet_mlp_max_grouped(structs, len(groups), C, dtype_code, _stream(x0))
    _lib.check(rc, "dvq_pointnet_mlp_max_grouped")
    return feats


def gather_parts(pts: torch.Tensor, idx: torch.Tensor, part_ptr: torch.Tensor, sizes, center: bool) -> list:
    """Vertex subsets of pts [B,3,V] (optionally centred on the mean over all V first), all parts in one launch.
    idx int32: the index lists back to back; part_ptr int32 [P+1]: their offsets (both on the device); sizes: the P
    list lengths (host).  Returns P contiguous tensors [B,3,n_k] (views of one buffer)."""
    _need_gpu(pts, idx, part_ptr)
    if pts.dim() != 3 or pts.shape[1] != 3 or not pts.is_contiguous() or idx.dtype != torch.int32 or part_ptr.dtype != torch.int32:
        raise RuntimeError("gather_parts: pts must be contiguous [B,3,V]; idx and part_ptr int32")
    B, _, V = pts.shape
    total = int(sum(sizes))
    out = torch.empty(3 * B * total, dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        rc = _lib.load().dvq_gather_parts(_p(pts), B, V, _p(idx), _p(part_ptr), len(sizes), int(center), _p(out), _stream(pts))
    _lib.check(rc, "dvq_gather_parts")
    parts, off = [], 0
    for n_k in sizes:
        parts.append(out[3 * B * off: 3 * B * (off + n_k)].view(B, 3, n_k))
        off += n_k
    return parts


# ----------------------------------------------------------------------- linear
# below this many rows the fp32 kernels are bound by latency, not by the matrix cores, and bf16 operands buy nothing
LINEAR_BF16_MIN_ROWS = 512


def to_bf16(w: torch.Tensor) -> torch.Tensor:
    """bf16 copy (round to nearest even) of a contiguous fp32 tensor, as dvq_linear_bf16 wants its weight."""
    _need_gpu(w)
    w = w.contiguous()
    out = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().dvq_to_bf16(_p(w), _p(out), w.numel(), _stream(w))
    _lib.check(rc, "dvq_to_bf16")
    return out


def linear(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], relu: bool = False,
           add_identity3: bool = False, out: Optional[torch.Tensor] = None,
           W_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x W^T + b).  x [M,K] and `out` [M,N] may be column slices of wider row-major buffers
    (unit inner stride).  fp32 arithmetic, unless a bf16 copy of the weight is passed as well and the
    shape is one the bf16 kernel takes (M >= LINEAR_BF16_MIN_ROWS, K % 32 == 0, N >= 64): then x is
    rounded to bf16 on the fly and the products accumulate in fp32."""
    _need_gpu(W, b)
    if not x.is_cuda or x.stride(-1) != 1 or x.dim() != 2:
        raise RuntimeError("linear: x must be a 2-D HIP tensor with unit inner stride")
    M, K = x.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise RuntimeError(f"linear: x has {K} features, weight expects {W.shape[1]}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    ldx = x.stride(0) if M > 1 else max(K, x.stride(0))
    ldy = out.stride(0) if M > 1 else max(N, out.stride(0))
    if W_bf16 is not None and not add_identity3 and M >= LINEAR_BF16_MIN_ROWS and K % 32 == 0 and N >= 64:
        if W_bf16.dtype != torch.bfloat16 or tuple(W_bf16.shape) != (N, K) or not W_bf16.is_contiguous():
            raise RuntimeError("linear: W_bf16 must be a contiguous bf16 copy of W")
        with torch.cuda.device(x.device):
            rc = _lib.load().dvq_linear_bf16(_p(x), ldx, _p(W_bf16), _p(b), M, K, N, int(relu), _p(out), ldy, _stream(x))
        _lib.check(rc, "dvq_linear_bf16")
        return out
    with torch.cuda.device(x.device):
        rc = _lib.load().dvq_linear(_p(x), ldx, _p(W), _p(b), M, K, N, int(relu), int(add_identity3), _p(out), ldy, _stream(x))
    _lib.check(rc, "dvq_linear")
    return out


def linear_batched(xs, Ws, bs, relu: bool = False, add_identity3: bool = False, Ws_bf16=None) -> list:
    """Several linear() calls of one shape -- own x [M,K], W [N,K], b each -- as one launch (up to 8).  With bf16 copies of
    the weights and a shape the bf16 kernel takes, the products run on bf16 operands as in linear()."""
    _need_gpu(*xs, *Ws, *[b for b in bs if b is not None])
    M, K = xs[0].shape
    N = Ws[0].shape[0]
    for x, W in zip(xs, Ws):
        if x.shape != (M, K) or W.shape != (N, K) or x.stride(-1) != 1 or x.stride(0) != xs[0].stride(0):
            raise RuntimeError("linear_batched: all products must have one shape and row stride")
    use_bf16 = (Ws_bf16 is not None and all(w is not None for w in Ws_bf16) and not add_identity3
                and M >= LINEAR_BF16_MIN_ROWS and K % 32 == 0 and N >= 64)
    outs = [torch.empty(M, N, dtype=torch.float32, device=xs[0].device) for _ in xs]
    ldx = xs[0].stride(0) if M > 1 else max(K, xs[0].stride(0))
    with torch.cuda.device(xs[0].device):
        rc = _lib.load().dvq_linear_batched(len(xs), _ptr_array(xs), ldx, _ptr_array(Ws_bf16 if use_bf16 else Ws), _ptr_array(bs),
                                            M, K, N, int(relu), int(add_identity3), _ptr_array(outs), N, int(use_bf16),
                                            _stream(xs[0]))
    _lib.check(rc, "dvq_linear_batched")
    return outs


# ------------------------------------------------------------------- call-site glue
PRIOR_ORDER_MAX_CLASSES = 1024


def assemble_param(recon: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """[B,61] = [betas 10 | global_orient 3 | pose 45 | transl 3] from the decoder's [B,55] and the position decoder's
    [B,6] outputs (gen_diverse_grasp_obman.py:243-247), one launch."""
    B = recon.shape[0]
    for t, w in ((recon, 55), (pos, 6)):
        if not t.is_cuda or t.dtype != torch.float32 or t.shape != (B, w) or t.stride(1) != 1:
            raise RuntimeError("assemble_param: float32 [B,55] and [B,6] tensors on a HIP device expected")
    out = torch.empty(B, 61, dtype=torch.float32, device=recon.device)
    with torch.cuda.device(recon.device):
        rc = _lib.load().dvq_assemble_param(_p(recon), recon.stride(0) if B > 1 else 55, _p(pos), pos.stride(0) if B > 1 else 6, B,
                                            _p(out), _stream(recon))
    _lib.check(rc, "dvq_assemble_param")
    return out


def prior_class_order(label: torch.Tensor, u: Optional[torch.Tensor], n_classes: int, label_sorted: torch.Tensor,
                      u_sorted: Optional[torch.Tensor], perm: torch.Tensor) -> None:
    """Stable counting sort of a batch by class: label int64 [B] -> label_sorted, the rows of u (float64 [B,n]) in that
    order -> u_sorted, perm int32 [B] (sorted position -> original row).  One launch."""
    _need_gpu(label, u, label_sorted, u_sorted, perm)
    B = label.numel()
    if label.dtype != torch.int64 or label_sorted.dtype != torch.int64 or perm.dtype != torch.int32 or label_sorted.numel() != B \
            or perm.numel() != B:
        raise RuntimeError("prior_class_order: int64 labels, int32 permutation, one entry per row")
    n_cols = 0
    if u is not None:
        if u.dtype != torch.float64 or u_sorted is None or u_sorted.dtype != torch.float64 or u.shape != u_sorted.shape or u.shape[0] != B:
            raise RuntimeError("prior_class_order: u and u_sorted must be float64 [B,n]")
        n_cols = int(u.shape[1])
    with torch.cuda.device(label.device):
        rc = _lib.load().dvq_prior_class_order(_p(label), _p(u), B, n_cols, int(n_classes), _p(label_sorted), _p(u_sorted), _p(perm),
                                               _stream(label))
    _lib.check(rc, "dvq_prior_class_order")


def unpermute_rows(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """out[perm[r]] = x[r] for int64 rows [B,n]."""
    _need_gpu(x, perm)
    if x.dtype != torch.int64 or x.dim() != 2 or perm.dtype != torch.int32 or perm.numel() != x.shape[0]:
        raise RuntimeError("unpermute_rows: int64 [B,n] rows and an int32 [B] permutation expected")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().dvq_unpermute_rows(_p(x), _p(perm), x.shape[0], x.shape[1], _p(out), _stream(x))
    _lib.check(rc, "dvq_unpermute_rows")
    return out


# ------------------------------------------------------------------------- MANO
def mano_lbs(global_orient, hand_pose, transl, consts: DvqManoConsts, n_verts: int,
             want_joints: bool = True, want_full_pose: bool = False, want_v_posed: bool = False,
             verts: Optional[torch.Tensor] = None, v_posed: Optional[torch.Tensor] = None):
    """MANO forward.  Returns (verts, joints, full_pose) and, with want_v_posed, the blend-shaped rest
    vertices the backward pass needs as a fourth element.  `verts` / `v_posed` may be preallocated.  The three inputs
    may be column ranges of wider row-major buffers (their rows must be contiguous); global_orient=None stands for
    zeros, transl=None for no translation."""
    B = hand_pose.shape[0]
    dev = hand_pose.device

    def rows(t, width, what):
        """Row stride of a [B, width] input whose rows are contiguous (a column range of a wider buffer is fine)."""
        if t is None:
            return 0
        if not t.is_cuda:
            raise RuntimeError("dvq_hip kernels need tensors on a HIP device (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32 or t.dim() != 2 or t.shape != (B, width) or (width > 1 and t.stride(1) != 1) \
                or (B > 1 and t.stride(0) < width):
            raise RuntimeError(f"mano_lbs: {what} must be float32 [B,{width}] with contiguous, non-overlapping rows")
        return int(t.stride(0)) if B > 1 else width
    ld_go = rows(global_orient, 3, "global_orient")
    ld_hp = rows(hand_pose, consts.n_comps, "hand_pose")
    ld_tr = rows(transl, 3, "transl")
    if verts is None:
        verts = torch.empty(B, n_verts, 3, dtype=torch.float32, device=dev)
    joints = torch.empty(B, 16, 3, dtype=torch.float32, device=dev) if want_joints else None
    full = torch.empty(B, 48, dtype=torch.float32, device=dev) if want_full_pose else None
    if want_v_posed and v_posed is None:
        v_posed = torch.empty(B, n_verts, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("mano_lbs", (B,), hand_pose):
        rc = _lib.load().dvq_mano_lbs_rows(_p(global_orient), ld_go, _p(hand_pose), ld_hp, _p(transl), ld_tr, B,
                                           ctypes.byref(consts), _p(verts), _p(joints), _p(full),
                                           _p(v_posed) if want_v_posed else None, _stream(hand_pose))
    _lib.check(rc, "dvq_mano_lbs_rows")
    if want_v_posed:
        return verts, joints, full, v_posed
    return verts, joints, full


def mano_lbs_backward(global_orient, hand_pose, consts: DvqManoConsts, v_posed, grad_verts, out=None):
    """(dL/dglobal_orient [B,3], dL/dhand_pose [B,ncomp], dL/dtransl [B,3]) from dL/dvertices.  `out` may be
    a preallocated triple."""
    _need_gpu(global_orient, hand_pose, v_posed, grad_verts)
    B = hand_pose.shape[0]
    dev = hand_pose.device
    if out is None:
        out = (torch.empty(B, 3, dtype=torch.float32, device=dev), torch.empty(B, hand_pose.shape[1], dtype=torch.float32, device=dev),
               torch.empty(B, 3, dtype=torch.float32, device=dev))
    g_go, g_hp, g_tr = out
    with torch.cuda.device(dev), _timed("mano_lbs_backward", (B,), hand_pose):
        rc = _lib.load().dvq_mano_lbs_backward(_p(global_orient), _p(hand_pose), B, ctypes.byref(consts), _p(v_posed),
                                               _p(grad_verts), _p(g_go), _p(g_hp), _p(g_tr), _stream(hand_pose))
    _lib.check(rc, "dvq_mano_lbs_backward")
    return g_go, g_hp, g_tr


# ------------------------------------------------------- test-time adaptation
def tta_contact_grad(obj: torch.Tensor, verts: torch.Tensor, mesh: DvqHandMesh, contact_dist_sq: float,
                     penetr_scale: float, contact_scale: float, w_penetr: float, w_contact: float,
                     grad_verts: Optional[torch.Tensor] = None, losses: Optional[torch.Tensor] = None,
                     nn_idx: Optional[torch.Tensor] = None, points_last: bool = True):
    """obj: [B,C>=3,N] channel-major cloud as the datasets produce it (points_last; only its first three
    channels are read) or [B,N,3]; any strides.  verts [B,V,3].  nn_idx: optional int32 [B,N] output.
    Returns (grad_verts [B,V,3], losses [B,2] = (penetr, contact))."""
    _need_gpu(verts, grad_verts, losses, nn_idx)
    if not obj.is_cuda:
        raise RuntimeError("dvq_hip kernels need tensors on a HIP device (got a CPU tensor); there is no CPU fallback")
    if obj.dtype != torch.float32 or verts.dtype != torch.float32:
        raise RuntimeError("tta_contact_grad: float32 expected")
    B, V = verts.shape[0], verts.shape[1]
    if obj.dim() != 3 or obj.shape[0] != B or obj.shape[1 if points_last else 2] < 3:
        raise RuntimeError("tta_contact_grad: obj must be [B,C>=3,N] (points_last) or [B,N,3]")
    if points_last:
        n_points, sb, sc, sp = obj.shape[2], obj.stride(0), obj.stride(1), obj.stride(2)
    else:
        n_points, sb, sc, sp = obj.shape[1], obj.stride(0), obj.stride(2), obj.stride(1)
    if nn_idx is not None and (nn_idx.dtype != torch.int32 or nn_idx.numel() != B * n_points):
        raise RuntimeError("tta_contact_grad: nn_idx must be int32 [B,N]")
    dev = verts.device
    if grad_verts is None:
        grad_verts = torch.empty(B, V, 3, dtype=torch.float32, device=dev)
    if losses is None:
        losses = torch.empty(B, 2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed("tta_contact_grad", (B, n_points), verts):
        rc = _lib.load().dvq_tta_contact_grad(_p(obj), sb, sc, sp, n_points, _p(verts), B, ctypes.byref(mesh), contact_dist_sq,
                                              penetr_scale, contact_scale, w_penetr, w_contact, _p(grad_verts), _p(losses),
                                              _p(nn_idx), _stream(verts))
    _lib.check(rc, "dvq_tta_contact_grad")
    return grad_verts, losses


# ------------------------------------------------------------------------ prior
PRIOR_PACK, PRIOR_GATE, PRIOR_ROWS = 0, 1, 2


def prior_packed_empty(rows: int, features: int, device, zero: bool = False, dtype_code: int = DVQ_BF16) -> torch.Tensor:
    """Buffer for `rows` x `features` activations in the packed layout of the prior kernels (bf16 or fp32)."""
    n = _lib.load().dvq_prior_packed_bytes(rows, features, dtype_code)
    if n == 0:
        raise RuntimeError("prior_packed_empty: features must be a positive multiple of 32")
    return (torch.zeros if zero else torch.empty)(n, dtype=torch.uint8, device=device)


def prior_pack_weight(W: torch.Tensor, dtype_code: int = DVQ_BF16) -> torch.Tensor:
    """W [N, K] fp32 (any strides, e.g. one kernel tap of a conv weight) -> packed bf16 or fp32."""
    if not W.is_cuda or W.dtype != torch.float32 or W.dim() != 2:
        raise RuntimeError("prior_pack_weight: a 2-D float32 HIP tensor is required")
    N, K = W.shape
    out = torch.empty(N * K * (4 if dtype_code == DVQ_F32 else 2), dtype=torch.uint8, device=W.device)
    with torch.cuda.device(W.device):
        rc = _lib.load().dvq_prior_pack_weight(_p(W), W.stride(0), W.stride(1), N, K, dtype_code, _p(out), _stream(W))
    _lib.check(rc, "dvq_prior_pack_weight")
    return out


def prior_embed(idx: torch.Tensor, table: torch.Tensor, out: torch.Tensor, dtype_code: int = DVQ_BF16) -> torch.Tensor:
    """out (packed) <- table[idx]; idx is a 1-D int64 view (may be strided)."""
    _need_gpu(table, out)
    M = idx.shape[0]
    V, d = table.shape
    with torch.cuda.device(table.device):
        rc = _lib.load().dvq_prior_embed(_p(idx), idx.stride(0), _p(table), V, M, d, dtype_code, _p(out), _stream(table))
    _lib.check(rc, "dvq_prior_embed")
    return out


def prior_job(taps, bias, cond=None, label=None, resid=None, out_packed=None, out_pre=None, out_rows=None,
              pre_add=None) -> DvqPriorJob:
    """taps: list of (packed activations, packed weights, k_blocks)."""
    j = DvqPriorJob()
    if not 1 <= len(taps) <= 10:
        raise RuntimeError("prior_job: 1..10 taps")
    for t, (x, w, kb) in enumerate(taps):
        j.taps[t].x, j.taps[t].w, j.taps[t].k_blocks = x.data_ptr(), w.data_ptr(), kb
    j.n_taps = len(taps)
    j.bias = bias.data_ptr()
    for name, t in (("cond", cond), ("label", label), ("resid", resid), ("out_packed", out_packed), ("out_pre", out_pre),
                    ("out_rows", out_rows), ("pre_add", pre_add)):
        setattr(j, name, t.data_ptr() if t is not None else None)
    return j


def prior_gemm(jobs, M: int, N: int, epilogue: int, relu: bool, ref: torch.Tensor, dtype_code: int = DVQ_BF16) -> None:
    arr = (DvqPriorJob * len(jobs))(*jobs)
    with torch.cuda.device(ref.device), _timed("prior_gemm", (M, N, len(jobs)), ref):
        rc = _lib.load().dvq_prior_gemm(arr, len(jobs), M, N, epilogue, int(relu), dtype_code, _stream(ref))
    _lib.check(rc, "dvq_prior_gemm")


def prior_sample(logits: torch.Tensor, u: torch.Tensor, x_out: torch.Tensor, table: Optional[torch.Tensor] = None,
                 emb_out: Optional[torch.Tensor] = None, dtype_code: int = DVQ_BF16) -> None:
    """x_out[m] <- inverse-CDF draw from softmax(logits[m]) at uniform u[m]; u and x_out may be strided columns of
    [M, cells] tensors.  With `table`/`emb_out` the chosen embedding rows are written in packed form."""
    _need_gpu(logits)
    if logits.dtype != torch.float32 or u.dtype != torch.float64 or x_out.dtype != torch.int64:
        raise RuntimeError("prior_sample: logits float32, u float64, x_out int64 expected")
    M, n = logits.shape
    if u.dim() != 1 or x_out.dim() != 1 or u.shape[0] != M or x_out.shape[0] != M:
        raise RuntimeError("prior_sample: u and x_out must be 1-d views with one entry per row")
    with torch.cuda.device(logits.device):
        rc = _lib.load().dvq_prior_sample(_p(logits), n, M, _p(u), u.stride(0), _p(x_out), x_out.stride(0), _p(table),
                                          0 if table is None else table.shape[1], dtype_code, _p(emb_out), _stream(logits))
    _lib.check(rc, "dvq_prior_sample")


def prior_uniforms(first_id: int, n: int, n_cells: int, seed: int, device) -> torch.Tensor:
    """float64 [n, n_cells] uniforms of objects first_id .. first_id + n - 1: the values of dvq_hip.rng.uniforms, one launch."""
    out = torch.empty(n, n_cells, dtype=torch.float64, device=device)
    if n > 0:
        with torch.cuda.device(device):
            rc = _lib.load().dvq_prior_uniforms(int(first_id), n, n_cells, int(seed) & 0xFFFFFFFFFFFFFFFF, _p(out),
                                                torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "dvq_prior_uniforms")
    return out


def prior_gather_rows(pairs, idx: torch.Tensor, M: int, V: int, dtype_code: int = DVQ_BF16) -> None:
    """out[m] <- table[idx[m]] for every (table, out, features) of `pairs`: packed activations of V rows handed out to
    the M rows of a batch.  One launch per 96 buffers."""
    _need_gpu(idx)
    if idx.dtype != torch.int64 or idx.dim() != 1 or idx.shape[0] != M or not idx.is_contiguous():
        raise RuntimeError("prior_gather_rows: idx must be a contiguous int64 vector with one entry per row")
    lib = _lib.load()
    for b in range(0, len(pairs), 96):
        part = pairs[b:b + 96]
        arr = (DvqPriorGather * len(part))()
        for k, (table, out, features) in enumerate(part):
            _need_gpu(table, out)
            arr[k].table, arr[k].out, arr[k].features = table.data_ptr(), out.data_ptr(), features
        with torch.cuda.device(idx.device), _timed("prior_gather_rows", (M, V, len(part)), idx):
            rc = lib.dvq_prior_gather_rows(arr, len(part), _p(idx), M, V, dtype_code, _stream(idx))
        _lib.check(rc, "dvq_prior_gather_rows")


def prior_vrow(rows, k_blocks: int, bias, cond, label, out_packed, out_pre, M: int, N: int, pre_add=None) -> None:
    """Vertical stack of one grid row.  rows: 1 or 2 entries (xs, ws) with xs = the three packed input columns of
    that input row and ws = the packed tap weights for column offsets -1, 0, +1; out_packed / out_pre: three
    packed outputs each (out_pre entries may be None).  pre_add: optionally three [classes, N] tables, one per cell,
    added by label before the gate (and to out_pre)."""
    v = DvqPriorVRow()
    v.n_rows, v.k_blocks = len(rows), k_blocks
    keep = []
    for r, (xs, ws) in enumerate(rows):
        for c in range(3):
            _need_gpu(xs[c], ws[c])
            v.x[r][c], v.w[r][c] = xs[c].data_ptr(), ws[c].data_ptr()
            keep += [xs[c], ws[c]]
    v.bias, v.cond, v.label = bias.data_ptr(), cond.data_ptr(), label.data_ptr()
    for j in range(3):
        v.pre_add[j] = pre_add[j].data_ptr() if pre_add is not None else None
    for j in range(3):
        v.out_packed[j] = out_packed[j].data_ptr()
        v.out_pre[j] = out_pre[j].data_ptr() if out_pre[j] is not None else None
    with torch.cuda.device(bias.device):
        rc = _lib.load().dvq_prior_vrow(ctypes.byref(v), M, N, _stream(bias))
    _lib.check(rc, "dvq_prior_vrow")


# ------------------------------------------------------------------ MLP heads
# The two wide layers of an MLP head (STN 1024-512-256, decoder 2560-1024-256, position decoder 2048-1024-128) at
# large batches, bf16 operands: the input pieces are packed once (prior_embed: rows of an fp32 matrix, or codebook
# rows picked by index -- the concatenation the reference builds is never materialised, each piece is a tap of the
# first product), the first layer's output stays packed in bf16 (exactly the rounding the second layer's operand
# load would apply) and the second layer writes fp32 rows.  Same LDS-staged kernel as the prior's products: 2-3x
# the rate of the register-operand dvq_linear_bf16 at these shapes.
_ROW_IDS = {}


def _row_ids(M: int, device) -> torch.Tensor:
    """0 .. M-1 on `device`, made once per size."""
    key = (M, str(device))
    if key not in _ROW_IDS:
        if len(_ROW_IDS) > 16:
            _ROW_IDS.clear()
        _ROW_IDS[key] = torch.arange(M, dtype=torch.int64, device=device)
    return _ROW_IDS[key]


def head_pack(W0: torch.Tensor, widths, W1: torch.Tensor):
    """Packed bf16 weights for head_mlp_bf16: W0 [N0, sum(widths)] cut into one block per input piece, W1 [N1, N0]."""
    offs = [0]
    for w in widths:
        offs.append(offs[-1] + w)
    if offs[-1] != W0.shape[1] or W1.shape[1] != W0.shape[0]:
        raise RuntimeError("head_pack: piece widths / layer shapes do not chain")
    W0 = W0.detach().float()
    return ([prior_pack_weight(W0[:, a:b], DVQ_BF16) for a, b in zip(offs[:-1], offs[1:])], prior_pack_weight(W1.detach().float(), DVQ_BF16),
            tuple(widths), W0.shape[0], W1.shape[0])


def head_shapes_ok(M: int, widths, N0: int, N1: int) -> bool:
    return (M >= LINEAR_BF16_MIN_ROWS and all(w % 64 == 0 for w in widths) and len(widths) <= 10 and N0 % 128 == 0 and N1 % 128 == 0)


def head_mlp_bf16(pieces, packed, b0: torch.Tensor, b1: torch.Tensor, relu1: bool = True) -> torch.Tensor:
    """relu(relu(cat(pieces) W0^T + b0) W1^T + b1) -> fp32 [M, N1].  A piece is an fp32 [M, d] matrix (unit inner stride)
    or a pair (idx int64 [M], table fp32 [V, d]) standing for table[idx]."""
    w0p, w1p, widths, N0, N1 = packed
    first = pieces[0][0] if isinstance(pieces[0], tuple) else pieces[0]
    M, dev = first.shape[0], first.device
    ar = None
    taps = []
    for pc, wp, d in zip(pieces, w0p, widths):
        buf = prior_packed_empty(M, d, dev, False, DVQ_BF16)
        if isinstance(pc, tuple):
            idx, table = pc
            if table.shape[1] != d:
                raise RuntimeError("head_mlp_bf16: piece width mismatch")
            prior_embed(idx, table, buf, DVQ_BF16)
        else:
            if pc.shape != (M, d) or pc.dtype != torch.float32 or not pc.is_contiguous():
                raise RuntimeError("head_mlp_bf16: pieces must be contiguous float32 [M, d]")
            if ar is None:
                ar = _row_ids(M, dev)
            prior_embed(ar, pc, buf, DVQ_BF16)
        taps.append((buf, wp, d // 32))
    h = prior_packed_empty(M, N0, dev, False, DVQ_BF16)
    prior_gemm([prior_job(taps, b0, out_packed=h)], M, N0, PRIOR_PACK, True, b0, DVQ_BF16)
    out = torch.empty(M, N1, dtype=torch.float32, device=dev)
    prior_gemm([prior_job([(h, w1p, N0 // 32)], b1, out_rows=out)], M, N1, PRIOR_ROWS, relu1, b0, DVQ_BF16)
    return out
