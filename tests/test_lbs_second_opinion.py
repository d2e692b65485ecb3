"""A SECOND, independent restatement of the SMPL layer, used only here: fp64 torch tensors, written op for op after the
published smplx `lbs()` sequence (blend_shapes -> vertices2joints -> pose feature (R - I) -> pose blend shapes ->
batch_rigid_transform with homogeneous 4x4 matrices and `transforms - pad(transforms @ joints)` -> skinning weights times
the flattened 4x4s -> homogeneous vertices) plus what /root/reference/models/smpl.py:27-46 adds (nine regressed extra
joints on the POSED vertices, landmark vertices).  Gradients come from torch.autograd, not from hand-derived formulas.

Provenance: `rodrigues_smplx`, `transform_mat` and `batch_rigid_transform` below are written AFTER the published source of the
third-party package smplx (smplx/lbs.py, Max-Planck-Gesellschaft; not vendored under /root/reference, not installed in this
image), whose function structure they follow statement by statement -- that is the point of this file: to restate what the
reference's dependency computes, independently of oracle/'s own formulation.  smplx's source is distributed under the
Max-Planck non-commercial scientific research licence; this file is test infrastructure only (imported by nothing outside
tests/) and "independent" above means independent of the C oracle, not of smplx.

The C oracle (oracle/lbs_ref_impl.inc) is a different formulation by construction: 3x4 affine chains without homogeneous
rows, rest-pose subtraction folded into the translation, a hand-written backward pass.  smplx itself and the licensed
model file are absent from this image, so neither restatement can be run against the original; two independently
written ones agreeing in value AND gradient (fp64, 1e-9) is the strongest pin available.
"""
import numpy as np
import torch
import torch.nn.functional as F

import oracle
from conftest import rand_pose_shape

DT = torch.float64


def _t(a):
    return torch.as_tensor(np.asarray(a), dtype=DT)


def rodrigues_smplx(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues: angle = |r + 1e-8|, R = I + sin K + (1 - cos) K K."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=DT)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((n, 3, 3))
    ident = torch.eye(3, dtype=DT).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def transform_mat(R, t):
    return torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(t, [0, 0, 0, 1], value=1)], dim=2)


def batch_rigid_transform(rot_mats, joints, parents):
    joints = torch.unsqueeze(joints, dim=-1)
    rel_joints = joints.clone()
    rel_joints[:, 1:] = rel_joints[:, 1:] - joints[:, parents[1:]]
    transforms_mat = transform_mat(rot_mats.reshape(-1, 3, 3), rel_joints.reshape(-1, 3, 1)).reshape(-1, joints.shape[1], 4, 4)
    transform_chain = [transforms_mat[:, 0]]
    for i in range(1, parents.shape[0]):
        transform_chain.append(torch.matmul(transform_chain[parents[i]], transforms_mat[:, i]))
    transforms = torch.stack(transform_chain, dim=1)
    posed_joints = transforms[:, :, :3, 3]
    joints_homogen = F.pad(joints, [0, 0, 0, 1])
    rel_transforms = transforms - F.pad(torch.matmul(transforms, joints_homogen), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, rel_transforms


def lbs_torch(model, betas, rot_mats):
    """-> verts [B,V,3], joints54 [B,54,3] = 24 posed joints | 21 landmark vertices | 9 extra regressed joints."""
    B = betas.shape[0]
    v_template = _t(model['v_template'])
    V = v_template.shape[0]
    shapedirs = _t(model['shapedirs']).reshape(V, 3, -1)
    posedirs = _t(model['posedirs'])                                  # [207, V * 3]
    J_regressor = _t(model['J_regressor'])
    parents = torch.as_tensor(np.asarray(model['parents']), dtype=torch.long).clone()
    lbs_weights = _t(model['lbs_weights'])
    v_shaped = v_template + torch.einsum('bl,mkl->bmk', [betas, shapedirs])
    J = torch.einsum('bik,ji->bjk', [v_shaped, J_regressor])
    ident = torch.eye(3, dtype=DT)
    pose_feature = (rot_mats[:, 1:, :, :] - ident).view([B, -1])
    pose_offsets = torch.matmul(pose_feature, posedirs).view(B, -1, 3)
    v_posed = pose_offsets + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    W = lbs_weights.unsqueeze(dim=0).expand([B, -1, -1])
    T = torch.matmul(W, A.view(B, 24, 16)).view(B, -1, 4, 4)
    v_posed_homo = torch.cat([v_posed, torch.ones([B, V, 1], dtype=DT)], dim=2)
    verts = torch.matmul(T, torch.unsqueeze(v_posed_homo, dim=-1))[:, :, :3, 0]
    landmarks = verts[:, torch.as_tensor(np.asarray(model['landmark_verts']), dtype=torch.long)]
    extra = torch.einsum('bik,ji->bjk', [verts, _t(model['J_regressor_extra'])])
    return verts, torch.cat([J_transformed, landmarks, extra], dim=1)


def test_forward_two_restatements_agree(smpl_model):
    B = 5
    betas, pose = rand_pose_shape(B, 21, pose_sigma=0.5)
    rot = rodrigues_smplx(_t(pose).reshape(-1, 3)).reshape(B, 24, 3, 3)
    v_t, j_t = lbs_torch(smpl_model, _t(betas), rot)
    v_c, j_c = oracle.lbs_forward(smpl_model, betas, rot.numpy(), True)
    assert np.abs(v_t.numpy() - v_c).max() < 1e-9
    assert np.abs(j_t.numpy() - j_c).max() < 1e-9
    # axis-angle entry point of the C oracle: its own Rodrigues against smplx's formula
    v_a, j_a = oracle.lbs_forward(smpl_model, betas, pose, False)
    assert np.abs(v_t.numpy() - v_a).max() < 1e-7
    assert np.abs(j_t.numpy() - j_a).max() < 1e-7


def test_backward_autograd_vs_hand_written(smpl_model):
    B = 3
    betas, pose = rand_pose_shape(B, 8, pose_sigma=0.4)
    rng = np.random.default_rng(5)
    gv = rng.normal(0, 1, (B, 6890, 3))
    gj = rng.normal(0, 1, (B, 54, 3))
    rot0 = rodrigues_smplx(_t(pose).reshape(-1, 3)).reshape(B, 24, 3, 3)
    bt = _t(betas).requires_grad_(True)
    rt = rot0.clone().requires_grad_(True)
    v, j = lbs_torch(smpl_model, bt, rt)
    ((v * _t(gv)).sum() + (j * _t(gj)).sum()).backward()
    gb, gr = oracle.lbs_backward(smpl_model, betas, rot0.numpy(), gv, gj)
    sb, sr = np.abs(gb).max(), np.abs(gr).max()
    assert np.abs(bt.grad.numpy() - gb).max() < 1e-9 * max(1.0, sb)
    assert np.abs(rt.grad.numpy() - gr).max() < 1e-9 * max(1.0, sr)
    # the two seeds separately (vertices only / joints only): the joint seed exercises landmarks + regressed joints
    for seed_v, seed_j in ((gv, None), (None, gj)):
        bt = _t(betas).requires_grad_(True)
        rt = rot0.clone().requires_grad_(True)
        v, j = lbs_torch(smpl_model, bt, rt)
        loss = (v * _t(seed_v)).sum() if seed_v is not None else (j * _t(seed_j)).sum()
        loss.backward()
        gb, gr = oracle.lbs_backward(smpl_model, betas, rot0.numpy(), seed_v, seed_j)
        assert np.abs(bt.grad.numpy() - gb).max() < 1e-9 * max(1.0, np.abs(gb).max())
        assert np.abs(rt.grad.numpy() - gr).max() < 1e-9 * max(1.0, np.abs(gr).max())


def test_f32_oracle_within_north_star_tolerance_of_second_opinion(smpl_model):
    """The fp32 C oracle (what the HIP kernels are compared with at 1e-4) against the fp64 torch restatement."""
    B = 4
    betas, pose = rand_pose_shape(B)
    rot = rodrigues_smplx(_t(pose).reshape(-1, 3)).reshape(B, 24, 3, 3)
    v_t, j_t = lbs_torch(smpl_model, _t(betas), rot)
    v32, j32 = oracle.lbs_forward(smpl_model, betas, rot.numpy(), True, np.float32)
    assert np.abs(v_t.numpy() - v32).max() < 2e-5 and np.abs(j_t.numpy() - j32).max() < 2e-5
