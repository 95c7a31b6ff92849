"""Reading MANO model files without executing anything stored in them.

The official MANO_RIGHT.pkl / MANO_LEFT.pkl are Python pickles that reference chumpy and
scipy classes; the reference loads them with pickle.load (mano/model.py:151-153), which
runs whatever the file names.  Here the pickle is decoded by a restricted unpickler that
only ever constructs numpy arrays/dtypes/scalars, scipy sparse matrices and inert
stand-ins for the two chumpy classes (their array payload is all that is needed), and
refuses every other global.  `.npz` files are read with allow_pickle=False.
"""
from __future__ import annotations

import io
import pickle

import numpy as np

FIELDS = ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "hands_components", "hands_mean",
          "kintree_table", "f")


class Struct(object):
    def __init__(self, **kwargs):
        for key, val in kwargs.items():
            setattr(self, key, val)


def to_np(array, dtype=np.float32):
    if 'scipy.sparse' in str(type(array)):
        array = np.array(array.todense(), dtype=dtype)
    elif hasattr(array, 'detach'):
        array = array.detach().cpu().numpy()
    elif isinstance(array, _ChArray):
        array = array.value()
    return np.array(array, dtype=dtype)


class _ChArray:
    """Stand-in for chumpy.ch.Ch / chumpy.reordering.Select: keeps the pickled state."""

    def __init__(self, *a, **k):
        self.state = {}

    def __setstate__(self, state):
        self.state = state if isinstance(state, dict) else {"x": state}

    def value(self):
        s = self.state
        if "x" in s:
            x = s["x"]
            return x.value() if isinstance(x, _ChArray) else np.asarray(x)
        if "a" in s and "idxs" in s:   # Select: a.ravel()[idxs].reshape(preferred_shape)
            a = s["a"].value() if isinstance(s["a"], _ChArray) else np.asarray(s["a"])
            out = a.ravel()[np.asarray(s["idxs"])]
            shape = s.get("preferred_shape")
            return out.reshape(shape) if shape is not None else out
        raise ValueError("unrecognised chumpy object in MANO file")

    def __array__(self, dtype=None, copy=None):
        v = self.value()
        return v.astype(dtype) if dtype is not None else v


_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("scipy.sparse.csc", "csc_matrix"), ("scipy.sparse._csc", "csc_matrix"),
    ("scipy.sparse.csr", "csr_matrix"), ("scipy.sparse._csr", "csr_matrix"),
    ("_codecs", "encode"),     # how Python 3 writes an array's bytes at protocol 2
    ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor"), ("__builtin__", "object"), ("builtins", "object"),
}
_CHUMPY = {("chumpy.ch", "Ch"), ("chumpy.reordering", "Select"), ("chumpy.ch", "MatVecMult"), ("chumpy.reordering", "Reorder")}


class _SafeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _CHUMPY:
            return _ChArray
        if (module, name) in _ALLOWED:
            if module in ("copy_reg", "__builtin__"):
                module = {"copy_reg": "copyreg", "__builtin__": "builtins"}[module]
            if module.startswith("numpy.core"):
                module = module.replace("numpy.core", "numpy._core") if hasattr(np, "_core") else module
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"MANO file names {module}.{name}, which is not a plain array type; refusing to load it")


def read_model_file(path: str) -> dict:
    """dict with the FIELDS the MANO layer needs, from a .pkl or .npz model file."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as f:
            return {k: f[k] for k in f.files}
    with open(path, "rb") as fh:
        data = _SafeUnpickler(io.BytesIO(fh.read()), encoding="latin1").load()
    if not isinstance(data, dict):
        raise ValueError("MANO model file does not hold a dict")
    missing = [k for k in FIELDS if k not in data]
    if missing:
        raise ValueError(f"MANO model file lacks {missing}")
    return {k: data[k] for k in data}
