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


@pytest.mark.parametrize('B', [1, 4, 7, 32, 33])
def test_one_launch_forward_equals_three_launch_forward(smpl, smpl_model, B):
    """csrc/smpl_lbs.hip: the forward as ONE kernel (smpl_fused_fwd_kernel, the default) against prep -> main -> finalize
    (ticket = NULL) on the same inputs: vertices, all 54 joints, and everything the backward pass reads (context rows,
    v_posed) -- so the gradients of the one-launch path are the three-launch path's.  Twice in a row: the arrival tickets
    are reset by the kernel itself."""
    from danet_densepose2smpl_amd import ops
    betas, pose = rand_pose_shape(B, 300 + B, pose_sigma=0.4)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    rng = np.random.default_rng(B)
    gv, gj = _t(rng.normal(0, 1, (B, 6890, 3)) * 1e-2), _t(rng.normal(0, 1, (B, 54, 3)))
    res = {}
    for one in (True, False, True):
        ops.LBS_ONE_LAUNCH = one
        try:
            tb, tr = _t(betas).requires_grad_(True), _t(rot).requires_grad_(True)
            verts, j54 = ops.smpl_lbs(tb, tr, smpl)
            ((verts * gv).sum() + (j54 * gj).sum()).backward()
            torch.cuda.synchronize()
            res.setdefault(one, []).append((verts.detach().clone(), j54.detach().clone(), tb.grad.clone(), tr.grad.clone()))
        finally:
            ops.LBS_ONE_LAUNCH = True
    a, b, a2 = res[True][0], res[False][0], res[True][1]
    for x, y in zip(a, a2):
        assert torch.equal(x, y)                                   # (deterministic, ticket reset included)
    assert (a[0] - b[0]).abs().max().item() <= 2e-6 and (a[1] - b[1]).abs().max().item() <= 2e-6
    for x, y in zip(a[2:], b[2:]):
        assert (x - y).abs().max().item() <= 1e-5 * (y.abs().max().item() + 1e-6)
    v_ref, j_ref = oracle.lbs_forward(smpl_model, betas.astype(np.float32), rot.astype(np.float32), True)
    np.testing.assert_allclose(a[0].cpu().numpy(), v_ref, atol=TOL)
    np.testing.assert_allclose(a[1].cpu().numpy(), j_ref, atol=TOL)


@pytest.mark.parametrize('B', [1, 8, 32, 33])
def test_one_launch_backward_equals_three_launch_backward(smpl, smpl_model, B):
    """csrc/smpl_lbs.hip smpl_fused_bwd_kernel (VERDICT r4 missing 4): the backward pass as ONE launch -- per-tile partials,
    pose-feature contraction and the fixed-order reduction + chain back-propagation separated by grid-wide barriers -- against the
    three launches it replaces (DANET-side switch: no barrier state handed in).  Same arithmetic, same summation orders: the
    gradients are bit-identical; both match the oracle; the barrier's error word stays clear; three calls in a row (the barrier
    state is reused).  (Inside a hipGraph: tests/test_gpu_zz_paths.py -- the captured train step holds this launch, and its replays
    must reproduce the eager gradients bit for bit.)  B = 33 is 540 workgroups, more than fit the device two per compute unit: the host must take the
    three-launch path by itself."""
    from danet_densepose2smpl_amd import ops, nn as dnn, conv as dconv, _lib
    betas, pose = rand_pose_shape(B, 500 + B, pose_sigma=0.4)
    rot = R.batch_rodrigues(pose.reshape(-1, 3)).reshape(B, 24, 3, 3)
    rng = np.random.default_rng(B)
    gv, gj = _t(rng.normal(0, 1, (B, 6890, 3)) * 1e-2), _t(rng.normal(0, 1, (B, 54, 3)))
    fits = bool(_lib.lib().danet_smpl_lbs_backward_fused_ok(B, 6890, 0))
    assert fits == (B <= 32)
    assert dnn.ONEPASS and dnn._onepass_bar(torch.device('cuda')) is not None           # (default stream = the one-pass stream outside a trainer)

    def run(fused):
        prev, prev_f = dnn.ONEPASS, ops.SMPL_BWD_FUSED
        dnn.ONEPASS = fused                        # no barrier state -> three launches
        ops.SMPL_BWD_FUSED = fused                 # (the one-launch form is opt-in since round 6: DANET_LBS_BWD_FUSED)
        try:
            dconv.FUSION.clear()
            tb, tr = _t(betas).requires_grad_(True), _t(rot).requires_grad_(True)
            verts, j54 = ops.smpl_lbs(tb, tr, smpl)
            ((verts * gv).sum() + (j54 * gj).sum()).backward()
            torch.cuda.synchronize()
            return tb.grad.clone(), tr.grad.clone(), dconv.FUSION.get('smpl_bwd_fused', 0)
        finally:
            dnn.ONEPASS, ops.SMPL_BWD_FUSED = prev, prev_f
    one = [run(True) for _ in range(3)]
    three = run(False)
    assert three[2] == 0 and all(o[2] == (1 if fits else 0) for o in one)
    assert not dnn.onepass_error()
    for o in one:
        assert torch.equal(o[0], three[0]) and torch.equal(o[1], three[1])
    gb_ref, gr_ref = oracle.lbs_backward(smpl_model, betas, rot, gv.cpu().numpy(), gj.cpu().numpy())
    np.testing.assert_allclose(one[0][0].cpu().numpy(), gb_ref, atol=2e-4 * np.abs(gb_ref).max())
    np.testing.assert_allclose(one[0][1].cpu().numpy(), gr_ref, atol=2e-4 * np.abs(gr_ref).max())


def test_one_launch_forward_replays_from_a_graph(smpl, smpl_model):
    """The SMPL forward inside a hipGraph (as in the captured train step): replays give the eager result, again and again
    (the tickets end every launch at zero).  (That it is ONE kernel is visible in the kernel statistics, profiles/r04_*.)"""
    B = 32
    betas, pose = rand_pose_shape(B, 77)
    tb, tp = _t(betas), _t(pose)
    eager = smpl(betas=tb, body_pose=tp[:, 3:], global_orient=tp[:, :3])
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        from danet_densepose2smpl_amd import ops
        rotm = ops.rodrigues_smplx(tp.reshape(-1, 3)).reshape(B, 24, 3, 3)
        ops.smpl_lbs(tb, rotm, smpl)                               # (this stream's ticket buffer exists before the capture)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            verts, j54 = ops.smpl_lbs(tb, rotm, smpl)
        for _ in range(3):
            verts.zero_(); j54.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(verts, eager.vertices)
            assert (j54[:, :24] - eager.smpl_joints).abs().max().item() == 0


@pytest.mark.gpu
def test_fused_joint_selection_and_sliced_rotations_equal_the_index_ops():
    """models/smpl.py:31-37 through ops.smpl_joints (one launch forward, one backward) and with the [B,24,3,3] tensor that global_orient / body_pose
    are slices of handed over as `rotmats=` (no concatenation) against the index-op formulation: outputs identical, gradients w.r.t. betas and the
    rotations equal to 1e-6 when every output (vertices, 49 joints, J19, the 24 SMPL joints) carries a gradient, and when only some do."""
    from danet_densepose2smpl_amd import smpl as dsmpl, assets
    dev = torch.device('cuda')
    model = dsmpl.SMPL(assets.make_synthetic_smpl(0)).to(dev)
    torch.manual_seed(3)
    B = 9
    betas0 = torch.randn(B, 10, device=dev)
    rot0 = torch.linalg.qr(torch.randn(B, 24, 3, 3, device=dev))[0].contiguous()
    ws = [torch.randn(B, model.v_template.shape[0], 3, device=dev), torch.randn(B, 49, 3, device=dev), torch.randn(B, 19, 3, device=dev),
          torch.randn(B, 24, 3, device=dev)]
    res = {}
    for fused in (True, False):
        dsmpl.FUSED_JOINTS = fused
        try:
            for which in ((0, 1, 2, 3), (1,), (0, 3)):
                betas, rot = betas0.clone().requires_grad_(True), rot0.clone().requires_grad_(True)
                if fused:
                    out = model(betas=betas, body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False, rotmats=rot)   # slices of one tensor, handed over
                else:
                    out = model(betas=betas, body_pose=rot[:, 1:].clone(), global_orient=rot[:, :1].clone(), pose2rot=False)
                outs = [out.vertices, out.joints, out.joints_J19, out.smpl_joints]
                sum((outs[k] * ws[k]).sum() for k in which).backward()
                res[(fused, which)] = ([o.detach().clone() for o in outs], betas.grad.clone(), rot.grad.clone())
        finally:
            dsmpl.FUSED_JOINTS = True
    for which in ((0, 1, 2, 3), (1,), (0, 3)):
        a, b = res[(True, which)], res[(False, which)]
        for x, y in zip(a[0], b[0]):
            assert torch.equal(x, y)
        assert (a[1] - b[1]).abs().max().item() <= 1e-6 * max(1.0, b[1].abs().max().item())
        assert (a[2] - b[2]).abs().max().item() <= 1e-6 * max(1.0, b[2].abs().max().item())
    with pytest.raises(ValueError):
        model(betas=betas0, body_pose=rot0[:, 1:], global_orient=rot0[:, :1].clone(), pose2rot=False, rotmats=rot0)
