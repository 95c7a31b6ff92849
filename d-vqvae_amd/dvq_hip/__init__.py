"""dvq_hip: MI355X kernels (libdvq_hip.so) behind the D-VQVAE generation path.

    ops      tensor-level wrappers over the C ABI (include/dvq_hip.h)
    synth    seeded synthetic weights / inputs
"""
import torch

_weights_epoch = 0


def weights_epoch() -> int:
    """Counter that every module mixes into the key of its packed (kernel-layout) copy of its weights."""
    return _weights_epoch


def weights_changed() -> None:
    """Tell the modules that weights were edited behind autograd's back.

    Every module keeps a packed copy of its weights (folded BatchNorm, MFMA fragment order, bf16 copies, the prior's
    per-class tables, the hand model's tables) and rebuilds it when a parameter or buffer has another address or another
    `_version`.  In-place edits through `tensor.data` -- `w.data.copy_(...)`, `w.data.uniform_()`, the idiom of the
    reference's own initialisers (quantizer.py:27, pixelcnn/models.py:11-18) -- do not move `_version`, so after such an
    edit of a model that has already run, call this once: every packed copy in the process is rebuilt on its next use.
    `load_state_dict`, optimizer steps, `.to(...)` and in-place operations on the parameter itself are seen without it."""
    global _weights_epoch
    _weights_epoch += 1


class DropsPackedCopies:
    """Mixin for modules that keep packed copies of their weights, launch buffers or recorded graphs: `_packed_defaults`
    maps each such attribute to a factory of its empty value, and pickles and deep copies of the module carry the empty
    values (the copies hold device addresses, ctypes structures and HIP graphs; all are rebuilt on the next use)."""
    _packed_defaults = {}

    def __getstate__(self):
        state = dict(self.__dict__)
        for name, empty in self._packed_defaults.items():
            if name in state:
                state[name] = empty()
        return state


def set_compute_dtype(model, dtype):
    """Select exact-fp32 (torch.float32) or bf16-operand (torch.bfloat16) matrix-core
    arithmetic for every PointNet encoder and MLP head inside `model` (the heads switch to bf16 operands only for
    batches of at least ops.LINEAR_BF16_MIN_ROWS rows; smaller batches stay in fp32)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("dtype must be torch.float32 or torch.bfloat16")
    for m in model.modules():
        if hasattr(m, "compute_dtype"):
            m.compute_dtype = dtype
    return model
