"""Data-parallel sharding of grasp generation: one process per GPU.

Objects are independent (eval-mode BatchNorm, no cross-sample operation), so rank r of G
takes the contiguous slice [r*ceil(n/G), ...) of the object list, weights are replicated
and nothing is exchanged while computing.  The only collective is the all-gather of the
finished outputs (vertices, 9336 B per grasp): RCCL over xGMI when the process group is
'nccl', gloo in the CPU tests and in the several-ranks-on-one-GPU rehearsal.
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[begin, end) of the objects rank `rank` owns; slices are contiguous, ordered by rank
    and differ in size by at most one block of ceil(n/world)."""
    per = (n + world_size - 1) // world_size
    b = min(n, rank * per)
    return b, min(n, b + per)


def pad_rows(t: torch.Tensor, m: int) -> torch.Tensor:
    """Pad a short last shard to m rows (repeating the last row) so all ranks gather equal blocks."""
    if t.shape[0] == m:
        return t
    if t.shape[0] == 0:
        return torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, t[-1:].expand(m - t.shape[0], *t.shape[1:])], 0)


class _HostStagedGather:
    """All-gather of a device tensor over a host-memory backend (gloo), staged explicitly: device -> pinned copy
    queued on the CURRENT stream right behind the producer, one event after it; wait() blocks the host on that event
    alone (not on the device), runs the collective on the pinned buffers and queues the copy back.

    torch's own gloo path for device tensors stages through a fresh HIGH-PRIORITY pool stream per collective.  On
    this ROCm stack such a stream, made to wait for an event of a stream that is still busy with the generation
    step's work (when first seen: the prior's replayed graph beside an encoder on a second stream), sometimes never drains -- reproducible
    in one process without any process group (tools/gloo_hang_probe.py emu1); the same wait on a normal-priority
    pool stream (what the RCCL backend uses), or on a high-priority stream that has been used before, completes.
    Staging on the caller's stream takes that stream out of the picture."""

    def __init__(self, t: torch.Tensor, ws: int):
        self.host_in = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.host_in.copy_(t, non_blocking=True)
        self.ready = torch.cuda.Event()
        self.ready.record(torch.cuda.current_stream(t.device))
        self.host_out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, pin_memory=True)
        self.out = torch.empty(self.host_out.shape, dtype=t.dtype, device=t.device)
        self.done = False

    def wait(self):
        if not self.done:
            self.ready.synchronize()
            dist.all_gather_into_tensor(self.host_out, self.host_in)
            self.out.copy_(self.host_out, non_blocking=True)     # ordered before later work on the current stream
            self.done = True
        return True


def gather_rows(t: torch.Tensor, n_total: Optional[int] = None, async_op: bool = False):
    """All-gather per-rank row blocks [m, ...] into [world*m, ...] (out of place), in rank order.
    With `n_total` the blocks are the shards of shard_range(n_total, rank, world): a short (or empty) last shard
    is padded to ceil(n_total/world) rows for the collective and the padding is trimmed from the result; a block
    of any other size is refused (every rank must bring the same number of rows or the collective mismatches).
    Without it all ranks must pass equally sized blocks.
    With async_op=True returns (rows, work), and `rows` is complete once work.wait() has been called (work is None
    when there is nothing to exchange); the caller MUST call it.  Over RCCL the collective is issued here and runs
    beside whatever is launched next.  Over gloo with device tensors (the one-GPU rehearsal) only the device-to-host
    copy is issued here: the host collective itself runs inside work.wait(), so nothing overlaps, and a rank that
    drops its handle without waiting never enters the collective and hangs its peers."""
    rank, ws = world()
    # (a world of one has nothing to exchange; with DVQ_FORCE_COLLECTIVE=1 and an initialised process group the
    # collective is issued all the same: the one-rank rehearsal of the RCCL path on a box with a single GPU)
    rehearse = ws == 1 and os.environ.get("DVQ_FORCE_COLLECTIVE") == "1" and dist.is_available() and dist.is_initialized()
    if ws == 1 and not rehearse:
        out = t if n_total is None else t[:n_total]
        return (out, None) if async_op else out
    if n_total is not None:
        b, e = shard_range(n_total, rank, ws)
        per = (n_total + ws - 1) // ws
        if t.shape[0] not in (e - b, per):      # (a block the caller has padded to the common size already is fine)
            raise RuntimeError(f"gather_rows: rank {rank} of {ws} holds {t.shape[0]} rows, its shard of {n_total} has {e - b}")
        t = pad_rows(t, per)
    t = t.contiguous()
    if t.is_cuda and dist.get_backend() == "gloo":
        h = _HostStagedGather(t, ws)
        out = h.out if n_total is None else h.out[:n_total]
        if async_op:
            return out, h
        h.wait()
        return out
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, t, async_op=async_op)
    out = out if n_total is None else out[:n_total]
    return (out, work) if async_op else out
