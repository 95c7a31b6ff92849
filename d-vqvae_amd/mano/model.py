"""MANO hand layer on the fused HIP pose->skinning kernel.

Mirrors DVQ-VAE-2/MANO-master/mano/model.py: load(model_path, is_rhand=True, **kwargs),
MANO(model_path, is_rhand, data_struct, create_*/value arguments, use_pca, num_pca_comps,
flat_hand_mean, batch_size, ...), the same 14 registered tensors (betas, global_orient,
transl, hand_pose, faces_tensor, v_template, shapedirs, J_regressor, posedirs, parents,
lbs_weights, hand_components, hand_mean, pose_mean), `.faces`, and
forward(betas, global_orient, hand_pose, transl, return_verts, return_tips,
return_full_pose, pose2rot) -> ModelOutput(vertices, joints, full_pose, betas,
global_orient, hand_pose).

Behaviour kept from the reference on purpose: the `betas` passed to forward do not
change the mesh -- the reference hands lbs the module's own `betas` parameter
(model.py:335), which every call site leaves at zero.  Unlike the reference, the call
batch does not have to equal the constructor's batch_size.
"""
import os.path as osp
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

import dvq_hip
from dvq_hip import ops
from dvq_hip._lib import DvqManoConsts
from dvq_hip.mano_io import Struct, read_model_file, to_np

ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'hand_pose'])
ModelOutput.__new__.__defaults__ = (None,) * len(ModelOutput._fields)

# fingertip vertices appended to the joints with return_tips=True (mano/joints_info.py:17-25)
TIP_IDS = {'mano': {'thumb': 744, 'index': 320, 'middle': 443, 'ring': 554, 'pinky': 671}}


def load(model_path, is_rhand=True, **kwargs):
    return MANO(model_path, is_rhand, **kwargs)


def _f32(a):
    return torch.as_tensor(to_np(a), dtype=torch.float32)


class _LbsFunction(torch.autograd.Function):
    """Differentiable vertices for callers that optimise the pose through the layer, as the
    reference's test-time adaptation does with autograd (DVQ-VAE/gen_HDMO_TTA.py:136-151).
    Joints and full_pose are returned without a gradient path."""

    @staticmethod
    def forward(ctx, go, hp, tr, module):
        consts = module._device_consts()
        verts, joints, full, vp = ops.mano_lbs(go, hp, tr, consts, module.v_template.shape[0], True, True,
                                               want_v_posed=True)
        ctx.save_for_backward(go, hp, vp)
        ctx.module = module
        ctx.has_transl = tr is not None
        ctx.mark_non_differentiable(joints, full)
        return verts, joints, full

    @staticmethod
    def backward(ctx, g_verts, g_joints, g_full):
        go, hp, vp = ctx.saved_tensors
        g_go, g_hp, g_tr = ops.mano_lbs_backward(go, hp, ctx.module._device_consts(), vp,
                                                 g_verts.to(torch.float32).contiguous())
        return g_go, g_hp, (g_tr if ctx.has_transl else None), None


def model_file_for(model_path, is_rhand, ext):
    """(file to read, handedness): a directory stands for the MANO_RIGHT / MANO_LEFT file inside it; a file's own name
    says which hand it is (mano/model.py:141-148)."""
    if osp.isdir(model_path):
        side = 'RIGHT' if is_rhand else 'LEFT'
        return osp.join(model_path, f'MANO_{side}.{ext}'), is_rhand
    return model_path, 'RIGHT' in osp.basename(model_path)


def skinning_table(lbs_weights, width=8):
    """Per vertex, the joints with a non-zero skinning weight in ascending order and their weights, zero-padded to
    `width` slots: (fits, joints uint8 [V, width], weights [V, width]).  `fits` is False when some vertex has more
    non-zero weights than slots (the kernel then takes the dense weight matrix).  The terms the table leaves out of the
    dense sum over 16 joints are exact zeros, so both forms give the same bits."""
    nz = lbs_weights != 0
    fits = int(nz.sum(1).max()) <= width
    order = torch.argsort((~nz).to(torch.int8), dim=1, stable=True)[:, :width]
    return fits, order.to(torch.uint8).contiguous(), torch.gather(lbs_weights, 1, order).contiguous()


def lane_ordered_posedirs(posedirs, V):
    """posedirs [135, V*3] in the order the forward kernel's lanes read it, [ceil(V/16), 36, 64, 3]: entry (g, s, lane)
    holds vertex min(16g + lane%16, V-1), pose feature 4s + lane//16 (zeros for features >= 135)."""
    dev = posedirs.device
    lane = torch.arange(64, device=dev)
    vert = (16 * torch.arange((V + 15) // 16, device=dev)[:, None, None] + lane % 16).clamp(max=V - 1)
    feat = 4 * torch.arange(36, device=dev)[None, :, None] + lane // 16
    padded = torch.cat([posedirs.reshape(135, V, 3), posedirs.new_zeros(9, V, 3)])
    return padded[feat, vert].contiguous()


class MANO(dvq_hip.DropsPackedCopies, nn.Module):
    _packed_defaults = {"_consts_key": lambda: None, "_consts": lambda: None}
    NUM_BODY_JOINTS = 1
    NUM_HAND_JOINTS = 15
    NUM_JOINTS = NUM_BODY_JOINTS + NUM_HAND_JOINTS
    NUM_BETAS = 10

    def __init__(self, model_path, is_rhand=True, data_struct=None, create_betas=True, betas=None,
                 create_global_orient=True, global_orient=None, create_transl=True, transl=None,
                 create_hand_pose=True, hand_pose=None, use_pca=True, num_pca_comps=6, flat_hand_mean=False,
                 batch_size=1, joint_mapper=None, v_template=None, dtype=torch.float32, vertex_ids=None,
                 use_compressed=True, ext='pkl', **kwargs):
        side = None
        if data_struct is None:
            mano_path, side = model_file_for(model_path, is_rhand, ext)
            if not osp.exists(mano_path):
                raise AssertionError(f'Path {mano_path} does not exist!')      # the reference asserts here
            data_struct = Struct(**read_model_file(mano_path))
        super().__init__()
        if side is not None and not osp.isdir(model_path):
            self.is_rhand = side
        if dtype != torch.float32:
            raise NotImplementedError("the HIP MANO kernel computes in float32")
        self.tip_ids = TIP_IDS['mano']
        self.batch_size = batch_size
        self.dtype = dtype
        self.joint_mapper = joint_mapper
        self.use_pca = use_pca
        self.num_pca_comps = num_pca_comps
        self.flat_hand_mean = flat_hand_mean

        self.faces = data_struct.f
        self.register_buffer('faces_tensor', torch.as_tensor(to_np(self.faces, dtype=np.int64), dtype=torch.long))

        def param(name, value, shape):
            if value is None:
                t = torch.zeros(shape, dtype=dtype)
            elif torch.is_tensor(value):
                t = value.clone().detach().to(dtype)
            else:
                t = torch.tensor(value, dtype=dtype)
            self.register_parameter(name, nn.Parameter(t, requires_grad=True))

        if create_betas:
            param('betas', betas, [batch_size, self.NUM_BETAS])
        if create_global_orient:
            param('global_orient', global_orient, [batch_size, 3])
        if create_transl:
            param('transl', transl, [batch_size, 3])

        self.register_buffer('v_template', _f32(data_struct.v_template))
        self.register_buffer('shapedirs', _f32(data_struct.shapedirs))
        self.register_buffer('J_regressor', _f32(data_struct.J_regressor))
        pd = to_np(data_struct.posedirs)
        self.register_buffer('posedirs', _f32(np.reshape(pd, [-1, pd.shape[-1]]).T))
        parents = torch.as_tensor(to_np(data_struct.kintree_table, dtype=np.int64)[0]).long()
        parents[0] = -1
        self.register_buffer('parents', parents)
        self.register_buffer('lbs_weights', _f32(data_struct.weights))

        comps = to_np(data_struct.hands_components)[:num_pca_comps]
        self.np_hand_components = comps
        if self.use_pca:
            self.register_buffer('hand_components', torch.tensor(comps, dtype=dtype))
        mean = to_np(data_struct.hands_mean)
        hand_mean = np.zeros_like(mean) if flat_hand_mean else mean
        self.register_buffer('hand_mean', _f32(hand_mean))

        hand_pose_dim = num_pca_comps if use_pca else 3 * self.NUM_HAND_JOINTS
        if create_hand_pose:
            param('hand_pose', hand_pose, [batch_size, hand_pose_dim])
        self.register_buffer('pose_mean', torch.cat([torch.zeros([3], dtype=dtype), self.hand_mean], dim=0))
        self._consts_key = None
        self._consts = None

    def get_num_verts(self):
        return int(self.v_template.size(0))

    def get_num_faces(self):
        return len(self.faces)

    def extra_repr(self):
        lines = [f'Number of betas: {self.NUM_BETAS}']
        if self.use_pca:
            lines.append(f'Number of PCA components: {self.num_pca_comps}')
        lines.append(f'Flat hand mean: {self.flat_hand_mean}')
        return '\n'.join(lines)

    def add_joints(self, vertices, joints, joint_ids=None):
        """joints followed by the vertices `joint_ids` (default: the five fingertips), [B, 16 + len(ids), 3]."""
        ids = list(self.tip_ids.values()) if joint_ids is None else joint_ids
        ids = torch.as_tensor(ids, dtype=torch.long, device=vertices.device)
        return torch.cat((joints, vertices[:, ids]), dim=1)

    def _device_consts(self):
        """Constant tables for the kernel.  The shape term uses the module's own betas
        (all-zero unless a caller edited the parameter), exactly as lbs receives them in
        the reference (model.py:335, lbs.py:77-81): v_shaped = v_template + betas[0]*shapedirs
        and the rest joints J = J_regressor * v_shaped are computed here once."""
        tensors = [self.v_template, self.shapedirs, self.J_regressor, self.posedirs, self.lbs_weights,
                   self.pose_mean, self.parents] + ([self.hand_components] if self.use_pca else [])
        if hasattr(self, 'betas'):
            tensors.append(self.betas)
        key = (dvq_hip.weights_epoch(),) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self._consts_key:
            dev = self.v_template.device
            with torch.no_grad():
                if hasattr(self, 'betas'):
                    v_shaped = self.v_template + torch.einsum('l,mkl->mk', self.betas[0], self.shapedirs)
                else:
                    v_shaped = self.v_template.clone()
                joints = torch.einsum('ik,ji->jk', v_shaped, self.J_regressor)
                comps = self.hand_components if self.use_pca else torch.eye(45, device=dev)
                width = 8                                 # slots of the kernel's table; denser models take the dense path
                has_table, skin_joints, skin_weights = skinning_table(self.lbs_weights, width)
                posedirs_lanes = lane_ordered_posedirs(self.posedirs, int(v_shaped.shape[0]))
                keep = dict(v_shaped=v_shaped.contiguous(), joints=joints.contiguous(), posedirs_lanes=posedirs_lanes,
                            posedirs=self.posedirs.contiguous(), weights=self.lbs_weights.contiguous(),
                            comps=comps.contiguous().float(), pose_mean=self.pose_mean.contiguous(),
                            parents=self.parents.to(torch.int32).contiguous(),
                            skin_joints=skin_joints, skin_weights=skin_weights)
            c = DvqManoConsts(int(v_shaped.shape[0]), int(comps.shape[0]), keep['v_shaped'].data_ptr(),
                              keep['joints'].data_ptr(), keep['posedirs'].data_ptr(), keep['weights'].data_ptr(),
                              keep['comps'].data_ptr(), keep['pose_mean'].data_ptr(), keep['parents'].data_ptr(),
                              keep['skin_joints'].data_ptr() if has_table else None,
                              keep['skin_weights'].data_ptr() if has_table else None, width,
                              keep['posedirs_lanes'].data_ptr() if has_table else None)
            self._consts = (c, keep)
            self._consts_key = key
            self._pose_mean_is_zero = bool((self.pose_mean == 0).all())     # flat_hand_mean: x - 0 == x, no launch needed
        return self._consts[0]

    @torch.no_grad()
    def hip_vertices(self, hand_pose, global_orient=None, transl=None):
        """Vertices [B,V,3] only, for callers inside this package: global_orient=None / transl=None stand for zeros (what
        gen() passes, gen_net.py:114-118) without a zero tensor being made, and the inputs may be column ranges of wider
        buffers.  forward() keeps the reference's meaning of None (the module's own parameters)."""
        return ops.mano_lbs(global_orient, hand_pose, transl, self._device_consts(), self.v_template.shape[0], False, False)[0]

    def forward(self, betas=None, global_orient=None, hand_pose=None, transl=None, return_verts=True,
                return_tips=False, return_full_pose=False, pose2rot=True, **kwargs):
        if not pose2rot:
            raise NotImplementedError("pose2rot=False (rotation-matrix input) has no HIP kernel")
        global_orient = global_orient if global_orient is not None else self.global_orient
        betas = betas if betas is not None else getattr(self, 'betas', None)
        hand_pose = hand_pose if hand_pose is not None else self.hand_pose
        if transl is None and hasattr(self, 'transl'):
            transl = self.transl
        if hasattr(self, 'betas') and self.betas.shape[0] > 1 and bool((self.betas != self.betas[0:1]).any()):
            raise NotImplementedError("per-row values in the module's own betas parameter are not supported")
        vertices = joints = full_pose = None
        want_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (global_orient, hand_pose, transl))
        keep = (lambda t: t) if want_grad else (lambda t: t.detach())

        def as_rows(t):
            """float32 with contiguous rows: a column range of a wider buffer (the call sites' slices of the 61-d
            parameter vector) is passed to the kernel as it is, anything else is copied."""
            t = keep(t).to(torch.float32)
            in_place = (t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1])
                        and not want_grad)          # (an expanded row has stride 0: copied)
            return t if in_place else t.contiguous()
        hp = as_rows(hand_pose)
        go = as_rows(global_orient)
        B = hp.shape[0]

        def rows(t, what):   # the module's own batch-1 defaults broadcast, as in the reference
            if t.shape[0] == B:
                return t
            if t.shape[0] == 1:
                return t.expand(B, -1).contiguous()
            raise RuntimeError(f"{what} has batch {t.shape[0]}, hand_pose has {B}")
        go = rows(go, "global_orient")
        tr = None
        if transl is not None:
            tr = rows(as_rows(transl), "transl")
        if return_verts:
            if want_grad:
                vertices, joints, full_pose = _LbsFunction.apply(go, hp, tr, self)
            else:
                vertices, joints, full_pose = ops.mano_lbs(go, hp, tr, self._device_consts(), self.v_template.shape[0], True, True)
            if return_tips:
                joints = self.add_joints(vertices, joints)
            if self.joint_mapper is not None:
                joints = self.joint_mapper(joints)
        elif return_full_pose:
            pose = hp @ self.hand_components if self.use_pca else hp
            full_pose = torch.cat([go, pose], dim=1) + self.pose_mean
        # the reference returns the PCA-expanded pose in `hand_pose` (model.py:328,356)
        if full_pose is None:
            pose45 = hand_pose
        elif getattr(self, '_pose_mean_is_zero', False):
            pose45 = full_pose[:, 3:]
        else:
            pose45 = full_pose[:, 3:] - self.pose_mean[3:]
        return ModelOutput(vertices=vertices, joints=joints, betas=betas, global_orient=global_orient,
                           hand_pose=pose45, full_pose=full_pose if return_full_pose else None)
