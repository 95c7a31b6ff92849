"""Data-parallel sharding of grasp generation: one process per GPU.

Objects are independent (eval-mode BatchNorm, no cross-sample operation), so rank r of G
takes the contiguous slice [r*ceil(n/G), ...) of the object list, weights are replicated
and nothing is exchanged while computing.  The only collective is the all-gather of the
finished outputs (vertices, 9336 B per grasp): RCCL over xGMI when the process group is
'nccl', gloo in the CPU tests.
"""
from typing import Tuple

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


def gather_rows(t: torch.Tensor, n_total: int = None, async_op: bool = False):
    """All-gather equally sized per-rank row blocks [m, ...] into [world*m, ...] (out of place);
    `n_total` trims padding rows when the last shard was short.
    With async_op=True returns (rows, work): the collective runs on the backend's own stream beside
    whatever is launched next, and `rows` is complete once work.wait() has been called (work is None
    when there is nothing to exchange)."""
    rank, ws = world()
    if ws == 1:
        out = t if n_total is None else t[:n_total]
        return (out, None) if async_op else out
    t = t.contiguous()
    if t.is_cuda and dist.get_backend() == "gloo":
        # rehearsal configuration (several ranks on one GPU): gloo's staging of device tensors was seen to hang when the
        # collective is issued behind work still running on the stream, and not when the device is idle
        torch.cuda.synchronize(t.device)
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(out, t, async_op=async_op)
    out = out if n_total is None else out[:n_total]
    return (out, work) if async_op else out


def pad_rows(t: torch.Tensor, m: int) -> torch.Tensor:
    """Pad a short last shard to m rows (repeating the last row) so all ranks gather equal blocks."""
    if t.shape[0] == m:
        return t
    if t.shape[0] == 0:
        return torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, t[-1:].expand(m - t.shape[0], *t.shape[1:])], 0)
