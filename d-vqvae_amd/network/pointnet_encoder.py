"""PointNet encoder on the fused HIP per-point-MLP + max-pool kernel.

Mirrors DVQ-VAE/network/pointnet_encoder.py: STN3d(channel), STNkd(k),
PointNetEncoder(global_feat, feature_transform, channel) with the same submodule names
(conv1..3, fc1..3, bn1..5, stn.*) and therefore the same 58 state_dict keys per encoder.
forward(x [B,C,N]) -> (feat [B,1024], trans [B,3,3], None).

Inference only: BatchNorm uses its running statistics (the generation scripts call
model.eval()); the eval-mode BatchNorm of every layer is folded into the preceding
affine map once per weight version and cached.
`compute_dtype` (torch.float32 or torch.bfloat16) selects exact-fp32 or bf16-operand
matrix-core arithmetic for the 64->128->1024 layers; set it per module or with
dvq_hip.set_compute_dtype(model, dtype).
"""
import torch
import torch.nn as nn

import dvq_hip
from dvq_hip import ops
from dvq_hip._lib import DVQ_BF16, DVQ_F32


def _version_key(module):
    return (dvq_hip.weights_epoch(),) + tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def _fold(conv_or_fc, bn):
    W = conv_or_fc.weight.detach()
    W = W.reshape(W.shape[0], -1).contiguous()
    return ops.fold_bn(W, conv_or_fc.bias.detach().contiguous(), bn.weight.detach(), bn.bias.detach(),
                       bn.running_mean, bn.running_var, bn.eps)


class _Trunk:
    """conv1..3 + bn1..3 of one module, folded and packed for dvq_pointnet_mlp_max."""

    def __init__(self, m, dtype_code):
        self.a1, self.c1 = _fold(m.conv1, m.bn1)
        self.a2, self.c2 = _fold(m.conv2, m.bn2)
        a3, self.c3 = _fold(m.conv3, m.bn3)
        self.w3p = ops.pointnet_pack_w3(a3, dtype_code)

    def as_dict(self):
        return dict(a1=self.a1, c1=self.c1, a2=self.a2, c2=self.c2, w3p=self.w3p, c3=self.c3)


def _dtype_code(dtype):
    if dtype == torch.float32:
        return DVQ_F32
    if dtype == torch.bfloat16:
        return DVQ_BF16
    raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")


def _check_input(module, x, channel):
    if module.training:
        raise RuntimeError("dvq_hip PointNet modules are inference-only: call model.eval() first "
                           "(BatchNorm batch statistics are not implemented)")
    if x.dim() != 3 or x.shape[1] != channel:
        raise RuntimeError(f"expected x [B,{channel},N], got {tuple(x.shape)}")
    x = x.detach().to(torch.float32)
    if ops.is_point_major(x):
        return x         # a [B,N,C] buffer seen as [B,C,N] (e.g. vertices.permute(0, 2, 1)): the kernels read it in place
    return x.contiguous()


class STN3d(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_cache_key": lambda: None, "_cache": lambda: None, "_head": lambda: None}

    def __init__(self, channel):
        super().__init__()
        self.channel = channel
        self.conv1 = nn.Conv1d(channel, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 9)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self.compute_dtype = torch.float32
        self._cache_key = None
        self._cache = None
        self._head = None

    def _packed(self):
        key = (_version_key(self), self.compute_dtype)
        if key != self._cache_key:
            trunk = _Trunk(self, _dtype_code(self.compute_dtype))
            f1 = _fold(self.fc1, self.bn4)
            f2 = _fold(self.fc2, self.bn5)
            self._head = None
            if self.compute_dtype == torch.bfloat16:   # bf16 copies of the folded weights for the large-batch kernels
                self._head = ops.head_pack(f1[0], [f1[0].shape[1]], f2[0])
                f1 = f1 + (ops.to_bf16(f1[0]),)
                f2 = f2 + (ops.to_bf16(f2[0]),)
            else:
                f1, f2 = f1 + (None,), f2 + (None,)
            self._cache = (trunk, f1, f2)
            self._cache_key = key
        return self._cache

    def forward(self, x):
        x = _check_input(self, x, self.channel)
        trunk, (w4, b4, w4h), (w5, b5, w5h) = self._packed()
        # conv/bn/relu x3 + max over points (pointnet_encoder.py:29-33)
        g = ops.pointnet_mlp_max(x, None, trunk.as_dict(), True, _dtype_code(self.compute_dtype))
        if self._head is not None and ops.head_shapes_ok(g.shape[0], [g.shape[1]], w4.shape[0], w5.shape[0]):
            g = ops.head_mlp_bf16([g], self._head, b4, b5)        # fc1+bn4+relu, fc2+bn5+relu (:35-36), large batches
        else:
            g = ops.linear(g, w4, b4, relu=True, W_bf16=w4h)      # fc1+bn4+relu (:35)
            g = ops.linear(g, w5, b5, relu=True, W_bf16=w5h)      # fc2+bn5+relu (:36)
        g = ops.linear(g, self.fc3.weight.detach(), self.fc3.bias.detach(), add_identity3=True)  # fc3 + I (:37-43)
        return g.view(-1, 3, 3)


class STNkd(nn.Module):
    """Feature-space transform net (pointnet_encoder.py:86-123).  Declared so that
    checkpoints and constructor calls keep working; no encoder on the generation path
    enables it (feature_transform=False everywhere) and its forward has no kernel."""

    def __init__(self, k=64):
        super().__init__()
        self.conv1 = nn.Conv1d(k, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.bn4 = nn.BatchNorm1d(512)
        self.bn5 = nn.BatchNorm1d(256)
        self.k = k

    def forward(self, x):
        raise NotImplementedError("STNkd (feature_transform=True) is not on the generation path and has no HIP kernel")


class PointNetEncoder(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_cache_key": lambda: None, "_cache": lambda: None}

    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        self.channel = channel
        self.stn = STN3d(channel)
        self.conv1 = nn.Conv1d(channel, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(128)
        self.bn3 = nn.BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        if self.feature_transform:
            self.fstn = STNkd(k=64)
        self.compute_dtype = torch.float32
        self._cache_key = None
        self._cache = None

    def _packed(self):
        own = [self.conv1, self.conv2, self.conv3, self.bn1, self.bn2, self.bn3]
        key = (tuple(_version_key(m) for m in own), self.compute_dtype)
        if key != self._cache_key:
            self._cache = _Trunk(self, _dtype_code(self.compute_dtype))
            self._cache_key = key
        return self._cache

    def forward(self, x):
        if self.feature_transform or not self.global_feat:
            raise NotImplementedError("only global_feat=True, feature_transform=False (the generation path) has a HIP kernel")
        x = _check_input(self, x, self.channel)
        self.stn.compute_dtype = self.compute_dtype
        trans = self.stn(x)
        # xyz @ trans, conv/bn/relu x2, conv/bn, max over points (pointnet_encoder.py:146-164)
        feat = ops.pointnet_mlp_max(x, trans.contiguous(), self._packed().as_dict(), False, _dtype_code(self.compute_dtype))
        return feat, trans, None


def encode_together(encoders, x):
    """Several PointNetEncoders of one configuration applied to the SAME input, every stage as one grouped launch
    (transform-net trunks, their three fully connected layers, encoder trunks).  Same results as calling them one after
    the other; worth it for small batches, where each launch on its own leaves most of the GPU idle.
    Returns [(feat, trans, None), ...]."""
    e0 = encoders[0]
    if any(e.feature_transform or not e.global_feat or e.channel != e0.channel or e.compute_dtype != e0.compute_dtype
           for e in encoders):
        raise RuntimeError("encode_together: encoders must share channel count, compute dtype and the global-feature path")
    for e in encoders:
        x = _check_input(e, x, e.channel).contiguous()       # (the grouped launches take channel-major clouds)
        e.stn.compute_dtype = e.compute_dtype
    code = _dtype_code(e0.compute_dtype)
    stn = [e.stn._packed() for e in encoders]                      # (trunk, (w4, b4, w4h), (w5, b5, w5h))
    g = ops.pointnet_mlp_max_grouped([(x, None, p[0].as_dict(), True) for p in stn], code)
    g = ops.linear_batched(g, [p[1][0] for p in stn], [p[1][1] for p in stn], relu=True, Ws_bf16=[p[1][2] for p in stn])
    g = ops.linear_batched(g, [p[2][0] for p in stn], [p[2][1] for p in stn], relu=True, Ws_bf16=[p[2][2] for p in stn])
    trans = ops.linear_batched(g, [e.stn.fc3.weight.detach() for e in encoders], [e.stn.fc3.bias.detach() for e in encoders],
                               add_identity3=True)
    feats = ops.pointnet_mlp_max_grouped([(x, t, e._packed().as_dict(), False) for e, t in zip(encoders, trans)], code)
    return [(f, t.view(-1, 3, 3), None) for f, t in zip(feats, trans)]
