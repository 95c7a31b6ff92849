"""dvq_vq_argmin_static -- the entry point that allocates nothing -- against dvq_vq_argmin, through the C ABI."""
import ctypes

import pytest
import torch

from dvq_hip import _lib, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LIMIT = 9 << 20


def _both(K, D, n, dtype, seed):
    z, E = synth.vq_case(K, D, n, "B", seed=seed)
    z, E = z.to(dtype).to(DEV), E.to(dtype).to(DEV)
    ref_idx, ref_zq = ops.vq_argmin(z, E)                      # dvq_vq_argmin, caller-owned workspace
    idx = torch.full((n,), -7, dtype=torch.int64, device=DEV)
    zq = torch.zeros_like(z)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    ops.vq_argmin_static(z, E, idx, zq)                        # dvq_vq_argmin_static
    assert torch.cuda.memory_allocated() == before
    assert torch.equal(idx, ref_idx), f"indices differ K={K} D={D} n={n} {dtype}"
    assert torch.equal(zq, ref_zq)


def test_static_workspace_two_calls_in_a_row_second_smaller():
    """Bit-equal to dvq_vq_argmin; the second call is smaller in every dimension (rows, codes, width), so a result
    that depended on what the first call left in the shared workspace would show."""
    for dtype in (torch.float32, torch.bfloat16):
        _both(2048, 256, 70001, dtype, seed=1)
        _both(100, 64, 1000, dtype, seed=2)
        _both(128, 1024, 37, dtype, seed=3)                    # few long rows: the split kernel + 64-bit atomics
        _both(128, 256, 4096, dtype, seed=4)
    _both(8192, 256, 5000, torch.float32, seed=5)              # the largest codebook the built-in workspace takes
    _both(128, 256, 0, torch.float32, seed=6)


def test_static_workspace_rejects_oversize_and_bad_shapes():
    lib = _lib.load()
    assert lib.dvq_vq_workspace_bytes(8192, 256, _lib.DVQ_F32) <= LIMIT < lib.dvq_vq_workspace_bytes(16384, 256, _lib.DVQ_F32)
    z = torch.zeros(64, 256, device=DEV)
    E = torch.zeros(16384, 256, device=DEV)
    idx = torch.zeros(64, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError, match="built-in"):
        ops.vq_argmin_static(z, E, idx)
    rc = lib.dvq_vq_argmin_static(ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(E.data_ptr()), 64, 128, 250, 0,
                                  ctypes.c_void_p(idx.data_ptr()), None, None)
    assert rc == -1
    torch.cuda.synchronize()
