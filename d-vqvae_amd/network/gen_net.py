"""GenNet: object cloud -> MANO parameters, on the HIP kernels.

Mirrors DVQ-VAE/network/gen_net.py:13-125: same submodule names (333 state_dict keys,
+14 `rh_mano.*` after set_rh_mano), gen(obj [B,4,N]) -> (recon [B,55], recon_pos [B,6]).
The reference's gen() is only valid at batch 1; here any B works and row b equals the
reference applied to obj[b:b+1].  Tensors stay on the input's device.

gen(obj, x_tilde=None, generator=None): `x_tilde` (int64 [B,3,3]) bypasses the
autoregressive prior with given part indices (teacher forcing); otherwise the prior is
sampled, optionally with a torch.Generator, or by inverse CDF from given `uniforms`
[B,9] (dvq_hip.rng: reproducible for any sharding of the batch).
"""
import torch
import torch.nn as nn

from dvq_hip import ops
from network.DVQVAE import Decoder, Encoder  # noqa: F401  (re-exported like the reference)
from network.pixelcnn.models import GatedPixelCNN
from network.pointnet_encoder import PointNetEncoder, encode_together
from network.VQVAE import VQVAE

# grid cells of the prior sample that hold part indices 0..5 (gen_net.py:95-100)
PART_CELLS = ((0, 1), (0, 2), (1, 1), (1, 2), (2, 1), (2, 2))


class GenNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.obj_encoder_type = PointNetEncoder(global_feat=True, feature_transform=False, channel=4)
        self.obj_encoder_pos = PointNetEncoder(global_feat=True, feature_transform=False, channel=4)
        for k in range(6):
            setattr(self, f"vqvae{k}", VQVAE(h_dim=128, res_h_dim=32, n_res_layers=2, n_embeddings=128,
                                             embedding_dim=256, beta=0.25, a=1))
        self.vqvae6 = VQVAE(h_dim=128, res_h_dim=32, n_res_layers=2, n_embeddings=128, embedding_dim=1024, beta=2, a=0)
        self.decoder = Decoder(layer_sizes=[1024, 256, 55], latent_size=2560)
        self.num = 0
        self.rh_mano = None
        self.recon_encoder = PointNetEncoder(global_feat=True, feature_transform=False, channel=3)
        self.pos_decoder = Decoder(layer_sizes=[1024, 128, 6], latent_size=2048)
        self.GatedPixelCNN = GatedPixelCNN(512, 512, 15)
        self.together_below = 64               # both object encoders stage by stage in grouped launches below this batch size

    def set_rh_mano(self, Rh_mano):
        Rh_mano.eval()
        self.rh_mano = Rh_mano

    def _decode(self, part_idx, obj_feat_type, obj_feat_pos):
        """codebook rows + object feature -> decoder -> MANO -> recon encoder -> pos decoder
        (gen_net.py:101-123).  part_idx: six int64 [B] index vectors (may be strided views)."""
        B = obj_feat_type.shape[0]
        # six codebook rows and the un-quantised object feature (gen_net.py:109), handed to the decoder piece by piece
        pieces = [(part_idx[k], getattr(self, f"vqvae{k}").vector_quantization.embedding.weight.detach()) for k in range(6)]
        recon = self.decoder.forward_pieces(pieces + [obj_feat_type]).view(B, 55)
        if hasattr(self.rh_mano, "hip_vertices") and recon.is_cuda:
            # zero orientation and translation (gen_net.py:114-118) without the zero tensors, the pose coefficients read
            # in place from the decoder's output
            verts = self.rh_mano.hip_vertices(recon[:, 10:55])
        else:
            zero3 = torch.zeros(B, 3, dtype=torch.float32, device=recon.device)
            verts = self.rh_mano(betas=recon[:, :10], global_orient=zero3, hand_pose=recon[:, 10:55].contiguous(),
                                 transl=zero3).vertices
        # [B,778,3] seen as [B,3,778] (gen_net.py:120): the encoder reads the point-major buffer in place
        hand_feat, _, _ = self.recon_encoder(verts.detach().permute(0, 2, 1))
        pos = self.pos_decoder.forward_pieces([hand_feat, obj_feat_pos]).view(B, 6)
        return recon, pos

    def gen(self, obj, x_tilde=None, generator=None, uniforms=None):
        if self.rh_mano is None:
            raise RuntimeError("GenNet.gen: call set_rh_mano(mano_layer) first")
        B = obj.size()[0]
        if B == 0:                                  # an empty batch decodes to nothing
            self.last_x_tilde = torch.zeros((0, 3, 3), dtype=torch.int64, device=obj.device)
            self.last_idx6 = torch.zeros((0, 1), dtype=torch.int64, device=obj.device)
            return torch.zeros((0, 55), device=obj.device), torch.zeros((0, 6), device=obj.device)
        together = B < self.together_below and obj.is_cuda
        if together:
            # small batches: both object encoders stage by stage in grouped launches (each launch alone would leave
            # most of the GPU idle)
            (obj_feat_type, _, _), (obj_feat_pos, _, _) = encode_together([self.obj_encoder_type, self.obj_encoder_pos], obj)
        else:
            obj_feat_type, _, _ = self.obj_encoder_type(obj)
        # (the reference also fetches the quantised object feature here and never uses it, gen_net.py:83: only the index)
        idx6 = self.vqvae6.vector_quantization.search(obj_feat_type)
        # The position encoder does not depend on the prior, but running it beside the prior on a stream of its own
        # gains nothing: an encoder workgroup pair fills a SIMD's registers, the prior's launches then wait for
        # slots and both slow down (17.1 ms against 17.0 ms per 4096 objects one after the other).
        if not together:
            obj_feat_pos, _, _ = self.obj_encoder_pos(obj)
        if x_tilde is None:
            label = idx6[:, 0]
            x_tilde = self.GatedPixelCNN.generate(None, label, shape=(3, 3), batch_size=B, generator=generator,
                                                  uniforms=uniforms)
        xt = x_tilde.to(torch.int64)
        part_idx = [xt[:, r, c] for r, c in PART_CELLS]      # strided views of the grid: no gather, no copies
        self.last_x_tilde = x_tilde
        self.last_idx6 = idx6
        return self._decode(part_idx, obj_feat_type, obj_feat_pos)

    def gen_byid(self, idx6):
        """gen_net.py:41-75: samples the prior for a given object code and decodes an
        all-zero latent (the reference ignores the sampled indices here)."""
        B = idx6.reshape(-1).shape[0]
        label = idx6.reshape(-1).to(torch.int64)
        self.GatedPixelCNN.generate(None, label, shape=(3, 3), batch_size=B)
        dev = label.device
        recon = self.decoder(torch.zeros(B, 2560, device=dev)).view(B, 55)
        return recon, torch.zeros(B, 6, device=dev)
