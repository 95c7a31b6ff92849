"""Parity rules shared by the CPU and GPU tests."""
import numpy as np
import torch

from oracle import dvq_oracle as O


def check_vq_rows(z, E, idx, ref_idx, ulps=None):
    """Index parity rule: equal to the reference wherever the exact (fp64) best-vs-second
    gap exceeds tau = `ulps` fp32 spacings of |z|^2+max|E|^2 (default max(4, sqrt(D)), see
    oracle.dvq_oracle.vq_tau for why it grows with the row width); on the remaining near-tie
    rows the chosen code must itself be within tau of the exact minimum.
    Returns the number of near-tie rows that differ."""
    z = z.numpy() if torch.is_tensor(z) else z
    E = E.numpy() if torch.is_tensor(E) else E
    idx = np.asarray(idx).reshape(-1)
    ref_idx = np.asarray(ref_idx).reshape(-1)
    best, dmin, gap = O.vq_gap_fp64(z, E)
    tau = O.vq_tau(z, E, ulps)
    differ = idx != ref_idx
    assert not np.any(differ & (gap > tau)), f"{int((differ & (gap > tau)).sum())} rows differ outside near-ties"
    d_at = O.vq_dist_fp64_at(z, E, idx)
    slack = d_at - dmin
    assert np.all(slack <= tau + 1e-30), f"chosen code beyond tau of the minimum on {(slack > tau).sum()} rows"
    return int(differ.sum())


def rel_err(a, b):
    """max|a-b| / max|b| -- the vertex tolerance of the task is stated in this form."""
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(np.abs(b).max(), 1e-30))
