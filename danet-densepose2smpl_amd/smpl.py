"""SMPL layer with the reference's call signature (/root/reference/models/smpl.py:15-46).

``SMPL(model_dir_or_dict, batch_size=, gender=, create_transl=False)`` and
``smpl(betas=, body_pose=, global_orient=, pose2rot=True)`` -> namedtuple with the fields
``vertices, joints, smpl_joints, joints_J19, betas, body_pose, global_orient, full_pose``.
The arithmetic (smplx LBS in the reference) is the fused HIP kernel of csrc/smpl_lbs.hip.
"""
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

from . import assets, constants, ops

ModelOutput_ = namedtuple('ModelOutput_', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose',
                                           'smpl_joints', 'joints_J19'])
ModelOutput_.__new__.__defaults__ = (None,) * len(ModelOutput_._fields)


FUSED_JOINTS = True        # joint selections of models/smpl.py:34-37 as one HIP launch (False: the index ops)


class SMPL(nn.Module):
    def __init__(self, model=None, batch_size=1, gender='neutral', create_transl=False, **kwargs):
        """model: dict of numpy arrays (see assets.make_synthetic_smpl), a path -- the official SMPL_*.pkl or a directory holding
        SMPL_<GENDER>.pkl like smplx's model_path (assets.load_smpl_pkl; `joint_regressor_extra` = the reference's
        J_regressor_extra.npy), or a converted .npz (assets.load_smpl_npz) --, or None for the seeded synthetic model."""
        super().__init__()
        if model is None:
            model = assets.make_synthetic_smpl(0)
        elif isinstance(model, str):
            import os
            if os.path.isdir(model):
                model = os.path.join(model, 'SMPL_%s.pkl' % gender.upper())
            model = (assets.load_smpl_npz(model) if model.endswith('.npz')
                     else assets.load_smpl_pkl(model, kwargs.get('joint_regressor_extra')))
        self.batch_size = batch_size
        self.gender = gender
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
        V = model['v_template'].shape[0]
        shapedirs = np.asarray(model['shapedirs'], np.float32).reshape(V * 3, -1)
        NB = shapedirs.shape[1]
        J_regressor = np.asarray(model['J_regressor'], np.float64)
        # state-dict names follow smplx; checkpoints strip them anyway
        self.register_buffer('faces_tensor', torch.as_tensor(np.asarray(model['faces'], np.int64)))
        self.register_buffer('v_template', f32(model['v_template']))
        self.register_buffer('shapedirs', f32(shapedirs))                       # [V*3, NB]
        self.register_buffer('J_regressor', f32(model['J_regressor']))
        self.register_buffer('posedirs', f32(model['posedirs']))                # [207, V*3]
        self.register_buffer('parents', torch.as_tensor(np.asarray(model['parents'], np.int32)))
        self.register_buffer('lbs_weights', f32(model['lbs_weights']))
        self.register_buffer('J_regressor_extra', f32(model['J_regressor_extra']))
        self.register_buffer('landmark_verts', torch.as_tensor(np.asarray(model['landmark_verts'], np.int32)))
        # joint regression folded into constants once (fp64): J = J_template + J_shapedirs . beta
        self.register_buffer('J_template', f32(J_regressor @ np.asarray(model['v_template'], np.float64)), persistent=False)
        self.register_buffer('J_shapedirs', f32((J_regressor @ shapedirs.astype(np.float64).reshape(V, 3 * NB)).reshape(24 * 3, NB)),
                             persistent=False)
        self.register_buffer('joint_map', torch.tensor(constants.JOINT_MAP_49, dtype=torch.long), persistent=False)
        self.register_buffer('j24_to_j19', torch.tensor(constants.J24_TO_J19, dtype=torch.long), persistent=False)
        self.faces = np.asarray(model['faces'])
        self.num_betas = NB

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, **kwargs):
        B = betas.shape[0]
        if pose2rot:
            full_pose = torch.cat([global_orient.reshape(B, -1, 3), body_pose.reshape(B, -1, 3)], dim=1)   # [B,24,3]
            rotmats = ops.rodrigues_smplx(full_pose.reshape(-1, 3)).view(B, 24, 3, 3)
        else:
            # callers that slice ONE [B,24,3,3] tensor into the two arguments (smpl_regressor.py:138-139) may hand that tensor over as
            # `rotmats=` as well: the concatenation and its backward (two copies and an add) are then not needed
            rotmats = kwargs.get('rotmats')
            if rotmats is not None:
                if tuple(rotmats.shape) != (B, 24, 3, 3) or rotmats.data_ptr() != global_orient.data_ptr():
                    raise ValueError('SMPL.forward: rotmats must be the [B,24,3,3] tensor global_orient / body_pose are slices of')
            else:
                rotmats = torch.cat([global_orient.reshape(B, -1, 3, 3), body_pose.reshape(B, -1, 3, 3)], dim=1)
            full_pose = rotmats
        vertices, joints54 = ops.smpl_lbs(betas, rotmats, self)
        if joints54.is_cuda and FUSED_JOINTS:
            # (models/smpl.py:34-37) the three selections in one launch, their gradients summed in one
            joints, joints_J19, smpl_joints = ops.smpl_joints(joints54, self.joint_map, self.j24_to_j19)
        else:
            joints = joints54[:, self.joint_map, :]                  # [B,49,3]  (models/smpl.py:35)
            smpl_joints = joints54[:, :24]                           # (models/smpl.py:34)
            joints_J19 = joints[:, -24:, :][:, self.j24_to_j19, :]   # (models/smpl.py:36-37)
        return ModelOutput_(vertices=vertices, global_orient=global_orient, body_pose=body_pose, joints=joints,
                            joints_J19=joints_J19, smpl_joints=smpl_joints, betas=betas, full_pose=full_pose)
