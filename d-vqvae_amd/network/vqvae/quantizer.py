"""VectorQuantizer backed by the fused HIP nearest-codebook kernel.

Mirrors DVQ-VAE/network/vqvae/quantizer.py: VectorQuantizer(n_e, e_dim, beta, al) with one
parameter `embedding.weight [n_e, e_dim]`; forward(z, istrain) returns
(indices int64 [n,1], z_q) in eval mode and (loss, z_q, perplexity, min_encodings, indices)
in training mode; get_emb(indices, dim) looks rows up.  Unlike the reference, tensors stay
on the device of the input (the reference hard-codes 'cuda', quantizer.py:7) and get_emb
accepts any number of indices (the reference only one, quantizer.py:72-74).
"""
import torch
import torch.nn as nn

import dvq_hip
from dvq_hip import ops


class VectorQuantizer(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_prep": lambda: None, "_prep_key": lambda: None}

    def __init__(self, n_e, e_dim, beta, al):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.al = al
        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)
        self._prep = None      # (key, workspace): codebook re-laid for the kernel
        self._prep_key = None

    def _prepared(self):
        w = self.embedding.weight
        key = (w.data_ptr(), w._version, w.device, w.dtype, dvq_hip.weights_epoch())
        if self._prep_key != key:
            self._prep = ops.vq_prepare(w.detach())
            self._prep_key = key
        return self._prep

    def _search(self, z_flat):
        w = self.embedding.weight.detach()
        idx, z_q = ops.vq_argmin(z_flat.detach().contiguous(), w, want_zq=True, prepared=self._prepared())
        return idx.unsqueeze(1), z_q

    def search(self, z):
        """Indices only, int64 [n,1] (forward(z, False)[0] without gathering the rows)."""
        z_flat = z.reshape(-1, self.e_dim)
        idx, _ = ops.vq_argmin(z_flat.detach().contiguous(), self.embedding.weight.detach(), want_zq=False, prepared=self._prepared())
        return idx.unsqueeze(1)

    def forward(self, z, istrain):
        z_flat = z.reshape(-1, self.e_dim)
        idx, z_q = self._search(z_flat)
        z_q = z_q.view(z.shape)
        if not istrain:
            return idx, z_q
        # training-mode outputs (quantizer.py:56-67); the row lookup is differentiable
        # w.r.t. the codebook through a plain index, everything else is elementwise
        z_q = self.embedding.weight[idx[:, 0]].view(z.shape)
        loss = self.al * torch.mean((z_q.detach() - z) ** 2) + self.beta * torch.mean((z_q - z.detach()) ** 2)
        z_q = z + (z_q - z).detach()
        min_encodings = torch.zeros(idx.shape[0], self.n_e, device=z.device)
        min_encodings.scatter_(1, idx, 1)
        e_mean = torch.mean(min_encodings, dim=0)
        perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
        return loss, z_q, perplexity, min_encodings, idx

    def get_emb(self, min_encoding_indices, dim):
        """Rows of the codebook, [n, dim].  An index outside the codebook raises (the reference's one-hot scatter_
        fails there, quantizer.py:70-72); the check reads the indices back, which this call -- not used inside
        gen() -- can afford."""
        idx = min_encoding_indices.reshape(-1).to(torch.int64).contiguous()
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.n_e):
            raise RuntimeError(f"get_emb: index out of range for a codebook of {self.n_e} rows")
        return ops.vq_gather(idx, self.embedding.weight.detach()).view(-1, dim)
