"""ctypes binding of libdvq_hip.so (declared in include/dvq_hip.h).

There is no CPU or eager-PyTorch fallback: if the library is missing the import of
any module that needs a kernel raises, and every call with a non-zero status raises
RuntimeError carrying dvq_last_error().
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVQ_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libdvq_hip.so")

DVQ_F32 = 0
DVQ_BF16 = 1


class DvqPointMlp(ctypes.Structure):
    _fields_ = [("a1", c_void_p), ("c1", c_void_p), ("a2", c_void_p), ("c2", c_void_p),
                ("w3p", c_void_p), ("c3", c_void_p)]


class DvqPointGroup(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("trans", c_void_p), ("B", c_int), ("N", c_int), ("w", DvqPointMlp), ("relu3", c_int),
                ("feat_out", c_void_p)]


class DvqManoConsts(ctypes.Structure):
    _fields_ = [("n_verts", c_int), ("n_comps", c_int), ("v_shaped", c_void_p), ("joints", c_void_p),
                ("posedirs", c_void_p), ("lbs_weights", c_void_p), ("components", c_void_p),
                ("pose_mean", c_void_p), ("parents", c_void_p), ("skin_joints", c_void_p), ("skin_weights", c_void_p),
                ("skin_width", c_int), ("posedirs_lanes", c_void_p)]


class DvqHandMesh(ctypes.Structure):
    _fields_ = [("n_verts", c_int), ("n_faces", c_int), ("n_contact", c_int), ("faces", c_void_p), ("vert_face_ptr", c_void_p),
                ("vert_face", c_void_p), ("search_order", c_void_p)]


class DvqPriorTap(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("k_blocks", c_int)]


class DvqPriorJob(ctypes.Structure):
    _fields_ = [("taps", DvqPriorTap * 10), ("n_taps", c_int), ("bias", c_void_p), ("cond", c_void_p),
                ("label", c_void_p), ("resid", c_void_p), ("out_packed", c_void_p), ("out_pre", c_void_p),
                ("out_rows", c_void_p), ("pre_add", c_void_p)]


class DvqPriorVRow(ctypes.Structure):
    _fields_ = [("x", (c_void_p * 3) * 2), ("w", (c_void_p * 3) * 2), ("n_rows", c_int), ("k_blocks", c_int),
                ("bias", c_void_p), ("cond", c_void_p), ("label", c_void_p), ("out_packed", c_void_p * 3),
                ("out_pre", c_void_p * 3), ("pre_add", c_void_p * 3)]


class DvqPriorGather(ctypes.Structure):
    _fields_ = [("table", c_void_p), ("out", c_void_p), ("features", c_int)]


# name -> (restype, argtypes); mirrors include/dvq_hip.h one to one
SIGNATURES = {
    "dvq_version": (c_int, []),
    "dvq_last_error": (c_char_p, []),
    "dvq_vq_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvq_vq_prepare": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dvq_vq_argmin": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvq_vq_argmin_prepared": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvq_vq_argmin_static": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dvq_vq_argmin_prepared_batched": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), c_int64, c_int, c_int, c_int,
                                               POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "dvq_vq_gather": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "dvq_pointnet_w3_bytes": (c_size_t, [c_int]),
    "dvq_pointnet_pack_w3": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "dvq_pointnet_mlp_max": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(DvqPointMlp), c_int, c_int, c_void_p, c_void_p]),
    "dvq_pointnet_mlp_max_grouped": (c_int, [POINTER(DvqPointGroup), c_int, c_int, c_int, c_void_p]),
    "dvq_gather_parts": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dvq_fold_bn": (c_int, [c_void_p] * 6 + [c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dvq_linear": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "dvq_linear_batched": (c_int, [c_int, POINTER(c_void_p), c_int64, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int,
                                   c_int, c_int, POINTER(c_void_p), c_int64, c_int, c_void_p]),
    "dvq_to_bf16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "dvq_linear_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "dvq_prior_packed_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dvq_prior_pack_weight": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dvq_prior_embed": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dvq_prior_gemm": (c_int, [POINTER(DvqPriorJob), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dvq_prior_gather_rows": (c_int, [POINTER(DvqPriorGather), c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dvq_prior_uniforms": (c_int, [c_int64, c_int64, c_int, ctypes.c_uint64, c_void_p, c_void_p]),
    "dvq_prior_vrow": (c_int, [POINTER(DvqPriorVRow), c_int, c_int, c_void_p]),
    "dvq_prior_sample": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dvq_mano_lbs_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, POINTER(DvqManoConsts), c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "dvq_assemble_param": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "dvq_prior_class_order": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dvq_unpermute_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "dvq_mano_lbs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, POINTER(DvqManoConsts), c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    "dvq_mano_lbs_backward": (c_int, [c_void_p, c_void_p, c_int, POINTER(DvqManoConsts), c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "dvq_tta_contact_grad": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, POINTER(DvqHandMesh), c_float,
                                     c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def load():
    """Load the shared library once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libdvq_hip.so not found at {LIB_PATH}: build it with `make -C d-vqvae_amd/csrc` "
                "(there is no fallback path)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().dvq_last_error()
        raise RuntimeError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")
