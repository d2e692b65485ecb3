"""Known-answer tests that pin the SMPL-layer oracle (oracle/lbs_ref_impl.inc).
smplx and the SMPL model file are absent and the reference has no tests, so the oracle is
pinned analytically: identity pose, rigid global rotation, one-hot skinning,
linearity in beta, pose2rot equivalence, fp64 finite-difference gradients."""
import numpy as np

import oracle
from oracle import numpy_ref as R
from conftest import rand_pose_shape


def _eye_rot(B):
    return np.tile(np.eye(3), (B, 24, 1, 1))


def test_identity_gives_template(smpl_model):
    B = 2
    verts, j54 = oracle.lbs_forward(smpl_model, np.zeros((B, 10)), _eye_rot(B), True)
    vt = smpl_model['v_template'].astype(np.float64)
    np.testing.assert_allclose(verts[0], vt, atol=1e-12)
    np.testing.assert_allclose(j54[0, :24], smpl_model['J_regressor'].astype(np.float64) @ vt, atol=1e-12)
    np.testing.assert_allclose(j54[0, 24:45], vt[smpl_model['landmark_verts']], atol=1e-12)
    np.testing.assert_allclose(j54[0, 45:], smpl_model['J_regressor_extra'].astype(np.float64) @ vt, atol=1e-12)
    assert verts.shape == (B, 6890, 3) and j54.shape == (B, 54, 3)


def test_global_rotation_is_rigid_about_root(smpl_model):
    betas, _ = rand_pose_shape(1, 3)
    rot = _eye_rot(1)
    v0, j0 = oracle.lbs_forward(smpl_model, betas, rot, True)
    Rg = R.batch_rodrigues(np.array([[0.3, -0.8, 0.5]]))[0]
    rot[0, 0] = Rg
    v1, j1 = oracle.lbs_forward(smpl_model, betas, rot, True)
    root = j0[0, 0]
    # fp32 skin-weight rows sum to 1 +- 6e-8, hence not 1e-12
    np.testing.assert_allclose(v1[0], (v0[0] - root) @ Rg.T + root, atol=1e-7)
    np.testing.assert_allclose(j1[0], (j0[0] - root) @ Rg.T + root, atol=1e-7)


def test_one_hot_skinning_follows_joint_transform(smpl_model):
    m = dict(smpl_model)
    rng = np.random.default_rng(0)
    owner = rng.integers(0, 24, 6890)
    w = np.zeros((6890, 24), np.float32)
    w[np.arange(6890), owner] = 1
    m['lbs_weights'] = w
    m['posedirs'] = np.zeros_like(m['posedirs'])
    betas, pose = rand_pose_shape(1, 5)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(1, 24, 3, 3)
    verts, j54 = oracle.lbs_forward(m, betas, rot, True)
    v_shaped = m['v_template'].astype(np.float64) + m['shapedirs'].astype(np.float64) @ betas[0]
    J = m['J_regressor'].astype(np.float64) @ v_shaped
    # independent chain in homogeneous 4x4 form
    G = [None] * 24
    for i in range(24):
        T = np.eye(4)
        T[:3, :3] = rot[0, i]
        p = m['parents'][i]
        T[:3, 3] = J[i] - (J[p] if p >= 0 else 0)
        G[i] = T if p < 0 else G[p] @ T
    exp = np.empty_like(v_shaped)
    for v in range(6890):
        g = G[owner[v]]
        exp[v] = g[:3, :3] @ (v_shaped[v] - J[owner[v]]) + g[:3, 3]
    np.testing.assert_allclose(verts[0], exp, atol=1e-10)
    np.testing.assert_allclose(j54[0, :24], np.stack([g[:3, 3] for g in G]), atol=1e-10)


def test_linear_in_beta_at_identity_pose(smpl_model):
    b1, _ = rand_pose_shape(1, 1)
    b2, _ = rand_pose_shape(1, 2)
    f = lambda b: oracle.lbs_forward(smpl_model, b, _eye_rot(1), True)[0]
    np.testing.assert_allclose(f(0.3 * b1 + 0.7 * b2), 0.3 * f(b1) + 0.7 * f(b2), atol=1e-10)


def test_pose2rot_matches_rotmat_path(smpl_model):
    betas, pose = rand_pose_shape(4)
    va, ja = oracle.lbs_forward(smpl_model, betas, pose, False)
    # smplx's own Rodrigues (I + sin K + (1-cos) K^2) vs the quaternion form of utils/geometry.py
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(4, 24, 3, 3)
    vb, jb = oracle.lbs_forward(smpl_model, betas, rot, True)
    np.testing.assert_allclose(va, vb, atol=1e-6)
    np.testing.assert_allclose(ja, jb, atol=1e-6)


def test_f32_matches_f64(smpl_model):
    betas, pose = rand_pose_shape(4)
    v64, j64 = oracle.lbs_forward(smpl_model, betas, pose, False, np.float64)
    v32, j32 = oracle.lbs_forward(smpl_model, betas, pose, False, np.float32)
    assert np.abs(v64 - v32).max() < 2e-5 and np.abs(j64 - j32).max() < 2e-5


def test_backward_finite_difference(smpl_model):
    B = 2
    betas, pose = rand_pose_shape(B, 11)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    rng = np.random.default_rng(3)
    gv = rng.normal(0, 1, (B, 6890, 3))
    gj = rng.normal(0, 1, (B, 54, 3))

    def loss(b, r):
        v, j = oracle.lbs_forward(smpl_model, b, r, True)
        return (v * gv).sum() + (j * gj).sum()
    gb, gr = oracle.lbs_backward(smpl_model, betas, rot, gv, gj)
    eps = 1e-6
    for (bi, li) in [(0, 0), (1, 3), (1, 9)]:
        d = np.zeros_like(betas)
        d[bi, li] = eps
        fd = (loss(betas + d, rot) - loss(betas - d, rot)) / (2 * eps)
        assert abs(fd - gb[bi, li]) <= 1e-5 * max(1.0, abs(fd)), (bi, li, fd, gb[bi, li])
    for (bi, j, r, c) in [(0, 0, 0, 0), (0, 0, 2, 1), (1, 3, 1, 2), (0, 9, 0, 1), (1, 16, 2, 2), (1, 23, 1, 0), (0, 12, 1, 1)]:
        d = np.zeros_like(rot)
        d[bi, j, r, c] = eps
        fd = (loss(betas, rot + d) - loss(betas, rot - d)) / (2 * eps)
        assert abs(fd - gr[bi, j, r, c]) <= 1e-5 * max(1.0, abs(fd)), (bi, j, r, c, fd, gr[bi, j, r, c])
    # None seeds are zeros
    gb2, gr2 = oracle.lbs_backward(smpl_model, betas, rot, gv, None)
    gb3, gr3 = oracle.lbs_backward(smpl_model, betas, rot, None, gj)
    np.testing.assert_allclose(gb2 + gb3, gb, atol=1e-9)
    np.testing.assert_allclose(gr2 + gr3, gr, atol=1e-9)
