"""GraspGenerator: the generation scripts' inner loop as one object.

What DVQ-VAE/gen_diverse_grasp_obman.py:236-253 does per object -- model.gen(obj_pc),
assemble the 61-d MANO parameter vector, rh_mano(...) -> vertices -- for a whole batch of
object clouds on one GPU, plus the all-gather of the results when several ranks each hold
a shard of the objects.
"""
from typing import Optional

import torch

from . import ops, parallel, rng, set_compute_dtype, synth
from .mano_io import Struct


def assemble_param(recon: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """[betas10 | global_orient3 | pose45 | transl3]  (gen_diverse_grasp_obman.py:243-247)"""
    if recon.is_cuda:
        return ops.assemble_param(recon, pos)                                            # one launch
    return torch.cat([recon[:, 0:10], pos[:, 0:3], recon[:, 10:55], pos[:, 3:6]], 1)


class GraspGenerator:
    def __init__(self, model, rh_mano, seed: int = 0):
        self.model = model
        self.rh_mano = rh_mano
        self.seed = seed
        self._in_flight = []          # work handles of asynchronous gathers, one list per generate() call
        if model.rh_mano is None:
            model.set_rh_mano(rh_mano)

    def finish(self, keep_newest: bool = False):
        """Complete the asynchronous gathers issued so far (all but the newest call's with keep_newest)."""
        while len(self._in_flight) > (1 if keep_newest else 0):
            for w in self._in_flight.pop(0):
                w.wait()

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
        all-gathers are left running beside whatever the caller launches next -- one call's gathers are in flight
        at a time, the previous call's are completed before this one returns -- and finish() completes the last;
        the dict also carries "pending", this call's work handles.
        Objects are dealt to ranks by parallel.shard_range(n_total, rank, world); a rank whose shard is empty
        (B = 0) launches nothing and still takes part in the collective."""
        B = obj.shape[0]
        prior = getattr(self.model, "GatedPixelCNN", None)
        if self._in_flight and x_tilde is None and prior is not None and not prior.graph_ready(B):
            self.finish()            # the prior is about to record a HIP graph: no collective may be pending then
        if B == 0:
            # an empty shard still takes part in the gather: its blocks must have the other ranks' row shape
            n_verts = int(self.rh_mano.v_template.shape[0])
            param = torch.zeros((0, 61), dtype=torch.float32, device=obj.device)
            verts = torch.zeros((0, n_verts, 3), dtype=torch.float32, device=obj.device)
        elif x_tilde is None:
            if obj.is_cuda:
                u = ops.prior_uniforms(first_id, B, 9, self.seed, obj.device)      # = rng.uniforms of these ids
            else:
                u = rng.uniforms(torch.arange(first_id, first_id + B, device=obj.device), 9, self.seed)
            recon, pos = self.model.gen(obj, uniforms=u)
        else:
            recon, pos = self.model.gen(obj, x_tilde=x_tilde)
        if B > 0:
            param = assemble_param(recon, pos)
            # the three slices of the parameter vector are read in place (row stride 61)
            out = self.rh_mano(betas=param[:, :10], global_orient=param[:, 10:13], hand_pose=param[:, 13:58],
                               transl=param[:, 58:])
            verts = out.vertices
        if gather and async_gather:
            verts, wv = parallel.gather_rows(verts, n_total, async_op=True)
            param, wp = parallel.gather_rows(param, n_total, async_op=True)
            pending = [w for w in (wv, wp) if w is not None]
            self._in_flight.append(pending)
            self.finish(keep_newest=True)
            return dict(recon_param=param, vertices=verts, pending=pending)
        if gather:
            verts = parallel.gather_rows(verts, n_total)
            param = parallel.gather_rows(param, n_total)
        return dict(recon_param=param, vertices=verts)
