"""dvq_hip: MI355X kernels (libdvq_hip.so) behind the D-VQVAE generation path.

    ops      tensor-level wrappers over the C ABI (include/dvq_hip.h)
    synth    seeded synthetic weights / inputs
"""
import torch


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
