"""GPU parity of the fused SMPL layer (csrc/smpl_lbs.hip) against the CPU oracle.
Tolerance: 1e-4 abs fp32 on vertices / joints (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import numpy_ref as R
from conftest import rand_pose_shape

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope='module')
def smpl(smpl_model):
    from danet_densepose2smpl_amd.smpl import SMPL
    return SMPL(smpl_model).cuda()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


@pytest.mark.parametrize('B', [1, 4, 7, 32, 33])
def test_forward_rotmat_matches_oracle(smpl, smpl_model, B):
    betas, pose = rand_pose_shape(B, 100 + B)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    out = smpl(betas=_t(betas), body_pose=_t(rot[:, 1:]), global_orient=_t(rot[:, :1]), pose2rot=False)
    v_ref, j54 = oracle.lbs_forward(smpl_model, betas.astype(np.float32), rot.astype(np.float32), True)
    assert out.vertices.shape == (B, 6890, 3) and out.joints.shape == (B, 49, 3)
    assert out.smpl_joints.shape == (B, 24, 3) and out.joints_J19.shape == (B, 19, 3)
    np.testing.assert_allclose(out.vertices.cpu().numpy(), v_ref, atol=TOL)
    from danet_densepose2smpl_amd import constants as C
    np.testing.assert_allclose(out.joints.cpu().numpy(), j54[:, C.JOINT_MAP_49], atol=TOL)
    np.testing.assert_allclose(out.smpl_joints.cpu().numpy(), j54[:, :24], atol=TOL)
    np.testing.assert_allclose(out.joints_J19.cpu().numpy(), j54[:, C.JOINT_MAP_49][:, -24:][:, C.J24_TO_J19], atol=TOL)


def test_forward_axis_angle_config0(smpl, smpl_model):
    """BASELINE.json configs[0]: batch=4 random pose/shape -> 6890 verts."""
    betas, pose = rand_pose_shape(4)
    out = smpl(betas=_t(betas), body_pose=_t(pose[:, 3:]), global_orient=_t(pose[:, :3]))
    v_ref, j54 = oracle.lbs_forward(smpl_model, betas.astype(np.float32), pose.astype(np.float32), False)
    np.testing.assert_allclose(out.vertices.cpu().numpy(), v_ref, atol=TOL)
    np.testing.assert_allclose(out.smpl_joints.cpu().numpy(), j54[:, :24], atol=TOL)


def test_known_answers_on_gpu(smpl, smpl_model):
    B = 2
    eye = np.tile(np.eye(3), (B, 24, 1, 1))
    out = smpl(betas=_t(np.zeros((B, 10))), body_pose=_t(eye[:, 1:]), global_orient=_t(eye[:, :1]), pose2rot=False)
    np.testing.assert_allclose(out.vertices[0].cpu().numpy(), smpl_model['v_template'], atol=1e-6)
    J = smpl_model['J_regressor'].astype(np.float64) @ smpl_model['v_template'].astype(np.float64)
    np.testing.assert_allclose(out.smpl_joints[0].cpu().numpy(), J, atol=1e-5)


def test_deterministic(smpl):
    betas, pose = rand_pose_shape(8, 5)
    a = smpl(betas=_t(betas), body_pose=_t(pose[:, 3:]), global_orient=_t(pose[:, :3]))
    b = smpl(betas=_t(betas), body_pose=_t(pose[:, 3:]), global_orient=_t(pose[:, :3]))
    assert torch.equal(a.vertices, b.vertices) and torch.equal(a.joints, b.joints)


@pytest.mark.parametrize('B', [2, 9])
def test_backward_matches_oracle(smpl, smpl_model, B):
    betas, pose = rand_pose_shape(B, 7 + B)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    rng = np.random.default_rng(B)
    gv = rng.normal(0, 1, (B, 6890, 3)) * 1e-2
    gj = rng.normal(0, 1, (B, 54, 3))
    tb, tr = _t(betas).requires_grad_(True), _t(rot).requires_grad_(True)
    from danet_densepose2smpl_amd import ops
    verts, j54 = ops.smpl_lbs(tb, tr, smpl)
    ((verts * _t(gv)).sum() + (j54 * _t(gj)).sum()).backward()
    gb_ref, gr_ref = oracle.lbs_backward(smpl_model, betas, rot, gv, gj)
    sb, sr = np.abs(gb_ref).max(), np.abs(gr_ref).max()
    np.testing.assert_allclose(tb.grad.cpu().numpy(), gb_ref, atol=2e-4 * sb)
    np.testing.assert_allclose(tr.grad.cpu().numpy(), gr_ref, atol=2e-4 * sr)


def test_backward_through_module_outputs(smpl, smpl_model):
    """Gradient through the 49-joint re-index used by the losses (smpl_regressor.py:176-207)."""
    B = 3
    betas, pose = rand_pose_shape(B, 21)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    tb, tr = _t(betas).requires_grad_(True), _t(rot).requires_grad_(True)
    out = smpl(betas=tb, body_pose=tr[:, 1:], global_orient=tr[:, :1], pose2rot=False)
    rng = np.random.default_rng(0)
    wj = rng.normal(0, 1, (B, 49, 3))
    (out.joints * _t(wj)).sum().backward()
    from danet_densepose2smpl_amd import constants as C
    gj54 = np.zeros((B, 54, 3))
    for i, j in enumerate(C.JOINT_MAP_49):
        gj54[:, j] += wj[:, i]
    gb_ref, gr_ref = oracle.lbs_backward(smpl_model, betas, rot, None, gj54)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), gb_ref, atol=2e-4 * np.abs(gb_ref).max())
    np.testing.assert_allclose(tr.grad.cpu().numpy(), gr_ref, atol=2e-4 * np.abs(gr_ref).max())


def test_geometry_helpers_vs_reference_goldens():
    from conftest import golden
    from danet_densepose2smpl_amd import ops
    g = golden('g1_geometry')
    np.testing.assert_allclose(ops.batch_rodrigues(_t(g['theta'])).cpu().numpy(), g['R'], atol=1e-5)
    x6 = _t(g['x6']).requires_grad_(True)
    R6 = ops.rot6d_to_rotmat(x6)
    np.testing.assert_allclose(R6.detach().cpu().numpy(), g['R6'], atol=1e-5)
    (R6 * _t(g['w6'])).sum().backward()
    np.testing.assert_allclose(x6.grad.cpu().numpy(), g['x6_grad'], atol=1e-4, rtol=1e-3)
    th = rand_pose_shape(4)[1].reshape(-1, 3)
    np.testing.assert_allclose(ops.rodrigues_smplx(_t(th)).cpu().numpy(), R.batch_rodrigues(th), atol=1e-5)
