"""DVQVAE and its MLP heads on the HIP kernels.

Mirrors DVQ-VAE/network/DVQVAE.py: DVQVAE(obj_inchannel=4), Encoder(layer_sizes, latent_size),
Decoder(layer_sizes, latent_size) with the reference's submodule names (emb_{i}, fing_{i},
obj_encoder_type/pos, vqvae0..6, decoder, recon_encoder, pos_decoder, MLP.L{i},
linear_means, linear_log_var).  Only the eval branch of DVQVAE.forward is implemented:
(obj_pc [B,4,N], hand_xyz [B,3,778]) -> (emb_idx int64 [7B,1], obj_emb [B,1024]).

The reference reads an undefined name for the first hand part (DVQVAE.py:93); the vertex
set used here is the documented reconstruction in dvq_hip.hand_parts.
"""
import torch
import torch.nn as nn

import dvq_hip
from dvq_hip import ops
from dvq_hip._lib import DVQ_BF16, DVQ_F32
from dvq_hip.hand_parts import HAND_PARTS
from network.pointnet_encoder import PointNetEncoder
from network.VQVAE import VQVAE


def _bf16_weight(layer):
    """bf16 copy of an nn.Linear's weight, kept on the layer and refreshed when the weight changes."""
    W = layer.weight.detach()
    key = (W.data_ptr(), W._version, dvq_hip.weights_epoch())
    if getattr(layer, "_dvq_bf16_key", None) != key:
        layer._dvq_bf16 = ops.to_bf16(W)
        layer._dvq_bf16_key = key
    return layer._dvq_bf16


def _linear(layer, x, relu=False, out=None, bf16=False):
    """nn.Linear (+ReLU) on the HIP kernels; with bf16=True large batches run on the bf16 matrix-core kernel."""
    W = layer.weight.detach()
    Wb = _bf16_weight(layer) if (bf16 and W.is_cuda and x.shape[0] >= ops.LINEAR_BF16_MIN_ROWS) else None
    return ops.linear(x, W, layer.bias.detach(), relu=relu, out=out, W_bf16=Wb)


class Encoder(nn.Module):
    def __init__(self, layer_sizes, latent_size):
        super().__init__()
        self.MLP = nn.Sequential()
        for i, (in_size, out_size) in enumerate(zip(layer_sizes[:-1], layer_sizes[1:])):
            self.MLP.add_module(name="L{:d}".format(i), module=nn.Linear(in_size, out_size))
            self.MLP.add_module(name="A{:d}".format(i), module=nn.ReLU())
        self.linear_means = nn.Linear(layer_sizes[-1], latent_size)
        self.linear_log_var = nn.Linear(layer_sizes[-1], latent_size)  # in the checkpoint, never used
        self.compute_dtype = torch.float32

    def forward(self, x, out=None):
        x = x.detach()
        bf16 = self.compute_dtype == torch.bfloat16
        for m in self.MLP:
            if isinstance(m, nn.Linear):
                x = _linear(m, x, relu=True, bf16=bf16)
        return _linear(self.linear_means, x, out=out, bf16=bf16)


class Decoder(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_heads": dict, "_head_wkey": lambda: None}

    def __init__(self, layer_sizes, latent_size):
        super().__init__()
        self.MLP = nn.Sequential()
        for i, (in_size, out_size) in enumerate(zip([latent_size] + layer_sizes[:-1], layer_sizes)):
            self.MLP.add_module(name="L{:d}".format(i), module=nn.Linear(in_size, out_size))
            if i + 1 < len(layer_sizes):
                self.MLP.add_module(name="A{:d}".format(i), module=nn.ReLU())
        self.compute_dtype = torch.float32

    def forward(self, z):
        return self.forward_pieces([z.detach()])

    def forward_pieces(self, pieces):
        """The decoder applied to the concatenation of `pieces` along the feature axis, without building it: a piece is
        an fp32 [B, d] matrix or a pair (idx int64 [B], table [V, d]) standing for table[idx] (codebook rows).  With
        bf16 operands, three layers and a large batch the two wide layers run as dvq_hip.ops.head_mlp_bf16."""
        layers = [m for m in self.MLP if isinstance(m, nn.Linear)]
        bf16 = self.compute_dtype == torch.bfloat16
        first = pieces[0][0] if isinstance(pieces[0], tuple) else pieces[0]
        widths = [(pc[1] if isinstance(pc, tuple) else pc).shape[1] for pc in pieces]
        if (bf16 and first.is_cuda and len(layers) == 3
                and ops.head_shapes_ok(first.shape[0], widths, layers[0].out_features, layers[1].out_features)):
            W0, W1 = layers[0].weight.detach(), layers[1].weight.detach()
            # packed weights per split of the input into pieces (forward(z) brings one piece, gen() seven): callers that
            # alternate between splits do not re-pack; a new weight version drops them all
            wkey = (W0.data_ptr(), W0._version, W1.data_ptr(), W1._version, dvq_hip.weights_epoch())
            if getattr(self, "_head_wkey", None) != wkey:
                self._heads, self._head_wkey = {}, wkey
            head = self._heads.get(tuple(widths))
            if head is None:
                head = self._heads[tuple(widths)] = ops.head_pack(W0, widths, W1)
            x = ops.head_mlp_bf16([pc if isinstance(pc, tuple) else pc.detach().contiguous() for pc in pieces], head,
                                  layers[0].bias.detach(), layers[1].bias.detach())
            return _linear(layers[2], x)
        if len(pieces) == 1 and not isinstance(pieces[0], tuple):
            x = pieces[0].detach()
        else:
            # the concatenation (gen_net.py:101-109), codebook rows gathered straight into their column range
            x = torch.empty(first.shape[0], sum(widths), dtype=torch.float32, device=first.device)
            col = 0
            for pc, w in zip(pieces, widths):
                if isinstance(pc, tuple):
                    ops.vq_gather(pc[0].contiguous(), pc[1], out=x[:, col:col + w])
                else:
                    x[:, col:col + w] = pc.detach()
                col += w
        for i, m in enumerate(layers):
            x = _linear(m, x, relu=(i + 1 < len(layers)), bf16=bf16)
        return x


class DVQVAE(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_part_tab": lambda: None}

    def __init__(self, obj_inchannel=4):
        super().__init__()
        self.obj_inchannel = obj_inchannel
        self.handembnns = [Encoder([1024, 512], 256) for _ in range(6)]
        for i, emb in enumerate(self.handembnns):
            self.add_module('emb_{}'.format(i), emb)
        self.obj_encoder_type = PointNetEncoder(global_feat=True, feature_transform=False, channel=obj_inchannel)
        self.obj_encoder_pos = PointNetEncoder(global_feat=True, feature_transform=False, channel=obj_inchannel)
        self.hand_encoders = [PointNetEncoder(global_feat=True, feature_transform=False, channel=3) for _ in range(6)]
        for i, enc in enumerate(self.hand_encoders):
            self.add_module('fing_{}'.format(i), enc)
        for k in range(6):
            setattr(self, f"vqvae{k}", VQVAE(h_dim=128, res_h_dim=32, n_res_layers=2, n_embeddings=128,
                                             embedding_dim=256, beta=0.25, a=1))
        self.vqvae6 = VQVAE(h_dim=128, res_h_dim=32, n_res_layers=2, n_embeddings=128, embedding_dim=1024, beta=2, a=0)
        self.decoder = Decoder(layer_sizes=[1024, 256, 55], latent_size=2560)
        self.rh_mano = None
        self.recon_encoder = PointNetEncoder(global_feat=True, feature_transform=False, channel=3)
        self.pos_decoder = Decoder(layer_sizes=[1024, 128, 6], latent_size=2048)

    def set_rh_mano(self, Rh_mano):
        self.rh_mano = Rh_mano
        Rh_mano.eval()

    def _part_tables(self, device):
        """The six vertex lists back to back (int32) and their offsets, on `device`."""
        cached = getattr(self, "_part_tab", None)
        if cached is None or cached[0] != device:
            sizes = [len(p) for p in HAND_PARTS]
            flat = torch.tensor([v for p in HAND_PARTS for v in p], dtype=torch.int32, device=device)
            ptr = torch.tensor([0] + [sum(sizes[:k + 1]) for k in range(len(sizes))], dtype=torch.int32, device=device)
            self._part_tab = cached = (device, flat, ptr, sizes)
        return cached[1:]

    def forward(self, obj_pc, hand_xyz):
        """The six hand parts go through every stage together: one launch for the (centred) vertex subsets, one per
        PointNet stage for the six part encoders and their six transform nets, one per fully connected layer for the
        six heads, one for the six codebook searches (DVQVAE.py:42-98, 130-142)."""
        if self.training:
            raise RuntimeError("DVQVAE: only the inference branch is implemented; call model.eval()")
        encs, heads = self.hand_encoders, self.handembnns
        dt = encs[0].compute_dtype
        if hand_xyz.dim() != 3 or hand_xyz.shape[1] != 3:
            raise RuntimeError(f"DVQVAE: hand_xyz must be [B, 3, V], got {tuple(hand_xyz.shape)}")
        if any(e.compute_dtype != dt for e in encs) or any(h.compute_dtype != dt for h in heads) or not hand_xyz.is_cuda:
            return self.forward_per_part(obj_pc, hand_xyz)
        code = DVQ_BF16 if dt == torch.bfloat16 else DVQ_F32
        bf16 = dt == torch.bfloat16
        obj_feat_type, _, _ = self.obj_encoder_type(obj_pc)
        hand = hand_xyz.detach().to(torch.float32).contiguous()
        flat, ptr, sizes = self._part_tables(hand.device)
        parts = ops.gather_parts(hand, flat, ptr, sizes, center=True)            # centre, then hand_xyz[:, :, idx]
        for e in encs:
            e.stn.compute_dtype = dt
        stn = [e.stn._packed() for e in encs]                                     # (trunk, (w4, b4, w4h), (w5, b5, w5h))
        g = ops.pointnet_mlp_max_grouped([(parts[k], None, stn[k][0].as_dict(), True) for k in range(6)], code)
        g = ops.linear_batched(g, [p[1][0] for p in stn], [p[1][1] for p in stn], relu=True, Ws_bf16=[p[1][2] for p in stn])
        g = ops.linear_batched(g, [p[2][0] for p in stn], [p[2][1] for p in stn], relu=True, Ws_bf16=[p[2][2] for p in stn])
        trans = ops.linear_batched(g, [e.stn.fc3.weight.detach() for e in encs], [e.stn.fc3.bias.detach() for e in encs],
                                   add_identity3=True)
        feats = ops.pointnet_mlp_max_grouped([(parts[k], trans[k], encs[k]._packed().as_dict(), False) for k in range(6)], code)
        x = feats
        n_mlp = len([m for m in heads[0].MLP if isinstance(m, nn.Linear)])
        for i in range(n_mlp):
            layers = [[m for m in h.MLP if isinstance(m, nn.Linear)][i] for h in heads]
            x = ops.linear_batched(x, [l.weight.detach() for l in layers], [l.bias.detach() for l in layers], relu=True,
                                   Ws_bf16=[_bf16_weight(l) for l in layers] if bf16 else None)
        layers = [h.linear_means for h in heads]
        lat = ops.linear_batched(x, [l.weight.detach() for l in layers], [l.bias.detach() for l in layers],
                                 Ws_bf16=[_bf16_weight(l) for l in layers] if bf16 else None)
        vqs = [getattr(self, f"vqvae{k}").vector_quantization for k in range(6)]
        idxs = ops.vq_argmin_batched(lat, [v.embedding.weight.detach() for v in vqs], [v._prepared() for v in vqs])
        idx6, obj_emb = self.vqvae6.inference(obj_feat_type)
        return torch.cat([idx6] + [i.unsqueeze(1) for i in idxs], 0), obj_emb

    def forward_per_part(self, obj_pc, hand_xyz):
        """The same computation with one launch per part and stage (the six parts one after the other)."""
        if self.training:
            raise RuntimeError("DVQVAE: only the inference branch is implemented; call model.eval()")
        hand_xyz = hand_xyz.detach()
        hand_xyz = hand_xyz - hand_xyz.mean(dim=2, keepdim=True)      # centre the hand (DVQVAE.py:49-51)
        obj_feat_type, _, _ = self.obj_encoder_type(obj_pc)
        idxs = []
        for k in range(6):
            part = hand_xyz[:, :, HAND_PARTS[k]].contiguous()
            feat, _, _ = self.hand_encoders[k](part)
            latent = self.handembnns[k](feat)
            idx_k, _ = getattr(self, f"vqvae{k}").inference(latent)
            idxs.append(idx_k)
        idx6, obj_emb = self.vqvae6.inference(obj_feat_type)
        return torch.cat([idx6] + idxs, 0), obj_emb
