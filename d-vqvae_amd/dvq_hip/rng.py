"""Counter-based uniforms for sampling the prior.

Every draw is a pure function of (seed, global sample id, cell), so a batch sharded over
any number of GPUs samples exactly what the unsharded batch would.  The mixer is the
splitmix64 finaliser evaluated with wrapping int64 tensor arithmetic on the device.
"""
import torch

_MASK53 = (1 << 53) - 1


def _mix(z: torch.Tensor) -> torch.Tensor:
    # splitmix64; int64 multiplication wraps, logical shifts emulated by masking
    def shr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)
    z = (z ^ shr(z, 30)) * -4658895280553007687      # 0xBF58476D1CE4E5B9
    z = (z ^ shr(z, 27)) * -7723592293110705685      # 0x94D049BB133111EB
    return z ^ shr(z, 31)


def uniforms(sample_ids: torch.Tensor, n_cells: int, seed: int = 0) -> torch.Tensor:
    """float64 uniforms in [0,1) of shape [len(sample_ids), n_cells]."""
    ids = sample_ids.to(torch.int64).reshape(-1, 1)
    cells = torch.arange(n_cells, dtype=torch.int64, device=ids.device).reshape(1, -1)
    off = (int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF      # golden-ratio stride per seed, wrapped
    off = off - (1 << 64) if off >= (1 << 63) else off
    ctr = ids * 1024 + cells + off
    bits = _mix(_mix(ctr) + 0x632BE59BD9B4E019)
    return ((bits >> 11) & _MASK53).to(torch.float64) / float(1 << 53)


def sample_rows(probs: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """Inverse-CDF sample of one index per row of `probs` [B,K] with uniforms u [B]."""
    cdf = torch.cumsum(probs.to(torch.float64), dim=1)
    cdf = cdf / cdf[:, -1:]
    idx = torch.searchsorted(cdf, u.reshape(-1, 1).to(torch.float64), right=True).squeeze(1)
    return idx.clamp_(max=probs.shape[1] - 1)
