"""GraspGenerator: the generation scripts' inner loop as one object.

What DVQ-VAE/gen_diverse_grasp_obman.py:236-253 does per object -- model.gen(obj_pc),
assemble the 61-d MANO parameter vector, rh_mano(...) -> vertices -- for a whole batch of
object clouds on one GPU, plus the all-gather of the results when several ranks each hold
a shard of the objects.
"""
from typing import Optional

import torch

from . import parallel, rng, set_compute_dtype, synth
from .mano_io import Struct


def assemble_param(recon: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """[betas10 | global_orient3 | pose45 | transl3]  (gen_diverse_grasp_obman.py:243-247)"""
    B = recon.shape[0]
    p = torch.empty((B, 61), dtype=recon.dtype, device=recon.device)
    p[:, 0:10] = recon[:, 0:10]
    p[:, 10:13] = pos[:, 0:3]
    p[:, 13:58] = recon[:, 10:55]
    p[:, 58:61] = pos[:, 3:6]
    return p


class GraspGenerator:
    def __init__(self, model, rh_mano, seed: int = 0):
        self.model = model
        self.rh_mano = rh_mano
        self.seed = seed
        if model.rh_mano is None:
            model.set_rh_mano(rh_mano)

    @classmethod
    def synthetic(cls, device, seed: int = 0, dtype=torch.float32, with_prior: bool = True):
        """Random-init weights of the reference architecture + a MANO-shaped hand model."""
        import mano
        from network.gen_net import GenNet
        model = GenNet()
        model.load_state_dict(synth.gennet_state_dict(seed, with_prior=with_prior), strict=with_prior)
        rh = mano.MANO('', data_struct=Struct(**synth.mano_model_data(seed)), use_pca=True, num_pca_comps=45,
                       batch_size=1, flat_hand_mean=True)
        model = model.to(device).eval()
        rh = rh.to(device).eval()
        set_compute_dtype(model, dtype)
        return cls(model, rh, seed)

    @torch.no_grad()
    def generate(self, obj: torch.Tensor, first_id: int = 0, x_tilde: Optional[torch.Tensor] = None,
                 gather: bool = False, n_total: Optional[int] = None, async_gather: bool = False):
        """obj [B,4,N] (this rank's shard; object b has global id first_id+b).
        Returns dict(recon_param [B,61], vertices [B,778,3]); with gather=True the vertices
        (and parameters) of all ranks, in global object order.  With async_gather=True the two
        all-gathers are left running beside whatever the caller launches next: the dict then carries
        "pending", the work handles to wait() on before the gathered tensors are read."""
        B = obj.shape[0]
        if x_tilde is None:
            ids = torch.arange(first_id, first_id + B, device=obj.device)
            u = rng.uniforms(ids, 9, self.seed)
            recon, pos = self.model.gen(obj, uniforms=u)
        else:
            recon, pos = self.model.gen(obj, x_tilde=x_tilde)
        param = assemble_param(recon, pos)
        out = self.rh_mano(betas=param[:, :10], global_orient=param[:, 10:13].contiguous(),
                           hand_pose=param[:, 13:58].contiguous(), transl=param[:, 58:].contiguous())
        verts = out.vertices
        if gather and async_gather:
            verts, wv = parallel.gather_rows(verts, n_total, async_op=True)
            param, wp = parallel.gather_rows(param, n_total, async_op=True)
            return dict(recon_param=param, vertices=verts, pending=[w for w in (wv, wp) if w is not None])
        if gather:
            verts = parallel.gather_rows(verts, n_total)
            param = parallel.gather_rows(param, n_total)
        return dict(recon_param=param, vertices=verts)
