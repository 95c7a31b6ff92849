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
    ops.vq_argmin_static(z, E, idx, zq)                        # dvq_vq_argmin_static
    assert torch.equal(idx, ref_idx), f"indices differ K={K} D={D} n={n} {dtype}"
    assert torch.equal(zq, ref_zq)
    # Allocation check on a repeat of the same call (the first launch of a kernel may make the RUNTIME load code or
    # grow its own spill area, which is not the library allocating): neither torch's allocator nor the device's free
    # memory (hipMemGetInfo: sees a hipMalloc made below torch) may move.
    idx.fill_(-7)
    torch.cuda.synchronize()
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    ops.vq_argmin_static(z, E, idx, zq)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "device free memory moved across dvq_vq_argmin_static"
    assert torch.equal(idx, ref_idx)


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


def test_static_workspace_call_recorded_in_a_hip_graph():
    """The callers this entry point is for include code under stream capture.  One call per shape is made outside the
    capture first (the first call on a device looks up the workspace's address and raises a kernel's dynamic-LDS
    limit -- runtime calls that do not belong in a capture); the recorded call is then replayed on new latents."""
    for dtype, K in ((torch.float32, 128), (torch.bfloat16, 512)):
        z, E = synth.vq_case(K, 256, 5000, "B", seed=31)
        z2, _ = synth.vq_case(K, 256, 5000, "B", seed=32)
        zbuf, E = z.to(dtype).to(DEV), E.to(dtype).to(DEV)
        idx = torch.empty(5000, dtype=torch.int64, device=DEV)
        zq = torch.empty_like(zbuf)
        ops.vq_argmin_static(zbuf, E, idx, zq)                  # outside capture, same shape
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ops.vq_argmin_static(zbuf, E, idx, zq)
        for src in (z2, z):
            zbuf.copy_(src.to(dtype))
            idx.fill_(-1)
            g.replay()
            torch.cuda.synchronize()
            ref_idx, ref_zq = ops.vq_argmin(zbuf, E)
            assert torch.equal(idx, ref_idx) and torch.equal(zq, ref_zq)
