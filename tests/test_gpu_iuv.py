"""GPU parity of the global IUV glue kernels (csrc/iuv_ops.hip) -- SURVEY.md 8 rows H4 (soft-argmax), H5 (body_uv_losses),
H6 (iuvmap_clean / iuv_img2map) and H7 (SMPL-side loss helpers on the device) -- against the reference's own golden
vectors (tests/golden/g2, g3, g10: produced by importing /root/reference, see make_golden.py) and against the tensor-op
statement of the same arithmetic (iuvmap.py / iuv_estimator.py, themselves pinned to those goldens on the CPU).
Integer-valued planes (one-hot index, argmax) must be bit-exact; fp32 sums 1e-5 relative; bf16 outputs exact after the
same rounding."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _cfg():
    from danet_densepose2smpl_amd.config import reset_cfg
    reset_cfg()


def test_clean_planes_bit_exact_vs_reference_golden():
    """iuvmap_clean (utils/iuvmap.py:6-38) inside the fused op: the one-hot index plane and the argmax are integer-exact,
    U / V equal the golden planes after the single bf16 rounding of the regressor operand."""
    from danet_densepose2smpl_amd import iuv_ops
    g = golden('g2_iuvmap')
    t = lambda k: torch.from_numpy(g[k]).cuda()
    sums, mp, am = iuv_ops.iuv_global(t('U'), t('V'), t('I'), t('A'))
    assert mp.dtype == torch.bfloat16 and mp.shape == (2, 80, 16, 16) and float(sums.abs().sum()) == 0.0
    mp = mp.float().cpu()
    np.testing.assert_array_equal(mp[:, 50:75].numpy(), g['cI'])
    np.testing.assert_array_equal(am.cpu().numpy(), g['I'].argmax(1).astype(np.uint8))
    for sl, k in ((slice(0, 25), 'cU'), (slice(25, 50), 'cV')):
        np.testing.assert_array_equal(mp[:, sl].numpy(), torch.from_numpy(g[k]).bfloat16().float().numpy())
    assert float(mp[:, 75:].abs().sum()) == 0.0
    # the tensor-op statement on the device gives the same planes exactly (H6 on the GPU)
    from danet_densepose2smpl_amd.iuvmap import iuvmap_clean, iuv_img2map
    for a, k in zip(iuvmap_clean(t('U'), t('V'), t('I'), t('A')), ('cU', 'cV', 'cI', 'cA')):
        np.testing.assert_array_equal(a.cpu().numpy(), g[k])
    for a, k in zip(iuv_img2map(t('img')), ('mU', 'mV', 'mI', 'mA')):
        np.testing.assert_array_equal(a.cpu().numpy(), g[k])


def test_global_losses_vs_reference_golden():
    """iuv_img2map + body_uv_losses (iuv_estimator.py:304-341) of the reference, incl. its has_iuv batch filter."""
    from danet_densepose2smpl_amd import iuv_ops
    from danet_densepose2smpl_amd.config import cfg
    _cfg()
    g = golden('g10_losses')
    t = lambda k: torch.from_numpy(g[k]).cuda()
    w = t('has_iuv').float()
    B, HW = 6, 64
    sums, _, _ = iuv_ops.iuv_global(t('u'), t('v'), t('idx'), t('ann'), t('iuv_gt'), w)
    got = {'loss_U': sums[0] / B * cfg.DANET.POINT_REGRESSION_WEIGHTS, 'loss_V': sums[1] / B * cfg.DANET.POINT_REGRESSION_WEIGHTS,
           'loss_I': sums[2] / (w.sum() * HW), 'loss_A': sums[3] / (w.sum() * HW)}
    for k, v in got.items():
        np.testing.assert_allclose(v.item(), g[k], rtol=2e-5, err_msg=k)
    # H7 on the device: the SMPL-side loss helpers, same goldens as tests/test_host_logic.py
    from danet_densepose2smpl_amd.smpl_regressor import SMPL_Regressor as R
    lp, lb = R.smpl_losses(t('pred_rot'), t('pb'), t('gt_rot'), t('gb'), t('has_smpl'))
    np.testing.assert_allclose(lp.item(), g['loss_pose'], rtol=1e-5)
    np.testing.assert_allclose(lb.item(), g['loss_betas'], rtol=1e-5)
    np.testing.assert_allclose(R.keypoint_loss(t('kp2'), t('gk2'), 0.25, 1.0).item(), g['loss_kp2d'], rtol=1e-5)
    np.testing.assert_allclose(R.keypoint_3d_loss(t('pj'), t('g3'), t('has_kp3d')).item(), g['loss_kp3d'], rtol=1e-5)
    np.testing.assert_allclose(R.shape_loss(t('pv'), t('gv'), t('has_smpl')).item(), g['loss_verts'], rtol=1e-5)
    np.testing.assert_allclose(R.l1_losses(t('a'), t('b'), t('has_smpl')).item(), g['loss_l1'], rtol=1e-5)


@pytest.mark.parametrize('with_keep', [False, True])
def test_fused_op_forward_backward_vs_tensor_ops(with_keep):
    """Forward and gradient of the fused op == autograd through iuv_img2map + body_uv_losses + part drop + iuvmap_clean +
    cat, on the conv epilogue's padded layout, at the working size (B=4, 64x64)."""
    from danet_densepose2smpl_amd import iuv_ops
    from danet_densepose2smpl_amd.iuvmap import iuvmap_clean, iuv_img2map
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator as E
    _cfg()
    B, S = 4, 64
    gen = torch.Generator().manual_seed(5)
    pad = lambda c, ld: torch.randn(B, S, S, ld, generator=gen).cuda().permute(0, 3, 1, 2)[:, :c]       # views of padded NHWC buffers
    u, v, ix, an = pad(25, 32), pad(25, 32), pad(25, 32), pad(15, 16)
    part = torch.randint(0, 25, (B, 1, S, S), generator=gen).float() / 24.
    gt = torch.cat([part, torch.rand(B, 2, S, S, generator=gen)], 1).cuda()
    w = torch.tensor([1., 0., 1., 1.]).cuda()
    keep = None
    if with_keep:
        keep = (torch.rand(B, 25, generator=gen) > 0.3).float().cuda()
        keep[:, 0] = 1
    gmap = torch.randn(B, 80, S, S, generator=gen).cuda()
    coef = torch.tensor([0.7, -0.3, 1.1, 0.4]).cuda()

    leaves = [t.detach().clone().requires_grad_(True) for t in (u, v, ix, an)]
    uvia = iuv_img2map(gt)
    B_ = float(B)
    fg = (uvia[2] > 0).float() * w.view(B, 1, 1, 1)
    import torch.nn.functional as F
    sU = (F.smooth_l1_loss(leaves[0], uvia[0], reduction='none') * fg).sum()
    sV = (F.smooth_l1_loss(leaves[1], uvia[1], reduction='none') * fg).sum()
    ce = lambda p, m: (F.cross_entropy(p, m.argmax(1), reduction='none') * w.view(B, 1, 1)).sum()
    sI, sA = ce(leaves[2], uvia[2]), ce(leaves[3], uvia[3])
    k4 = 1.0 if keep is None else keep.view(B, 25, 1, 1)
    cu, cv, ci, _ = iuvmap_clean(leaves[0] * k4, leaves[1] * k4, leaves[2] * k4, leaves[3])
    ref_map = torch.cat([cu, cv, ci], 1)
    ref_sums = torch.stack([sU, sV, sI, sA])
    ((ref_sums * coef).sum() + (ref_map * gmap[:, :75]).sum()).backward()

    mine = [t.detach().clone().requires_grad_(True) for t in (u, v, ix, an)]
    # re-create the padded views so that the op reads them without a copy
    mine = [torch.nn.functional.pad(t.detach().permute(0, 2, 3, 1), (0, ld - t.shape[1])).permute(0, 3, 1, 2).requires_grad_(True)
            for t, ld in zip((u, v, ix, an), (32, 32, 32, 16))]
    views = [m[:, :c] for m, c in zip(mine, (25, 25, 25, 15))]
    sums, mp, am = iuv_ops.iuv_global(*views, gt, w, keep)
    ((sums * coef).sum() + (mp.float() * gmap).sum()).backward()
    np.testing.assert_allclose(sums.detach().cpu().numpy(), ref_sums.detach().cpu().numpy(), rtol=2e-5)
    assert torch.equal(mp[:, 50:75].float(), ref_map[:, 50:].detach())                                   # one-hot plane: exact
    assert torch.equal(mp[:, :50].float(), ref_map[:, :50].detach().bfloat16().float())
    assert torch.equal(am.long(), ix.argmax(1))
    for m, r, c, name in zip(mine, leaves, (25, 25, 25, 15), 'u v index ann'.split()):
        gm, gr = m.grad[:, :c], r.grad
        if name in ('u', 'v'):          # the clean path multiplies a bf16-rounded upstream gradient in the fused op
            tol = 1e-2 * gr.abs().max().item()
        else:
            tol = 2e-5 * gr.abs().max().item() + 1e-7
        assert (gm - gr).abs().max().item() <= tol, name
        assert float(m.grad[:, c:].abs().sum()) == 0.0, name                                              # padding channels get zeros


def test_softargmax_vs_reference_golden_and_autograd():
    from danet_densepose2smpl_amd import iuv_ops
    from danet_densepose2smpl_amd.geometry import softmax_integral_tensor
    g = golden('g3_graph')
    hm = torch.from_numpy(g['hm']).cuda()
    np.testing.assert_allclose(iuv_ops.softargmax(hm, 10.0).cpu().numpy(), g['softint'], atol=2e-4)
    gen = torch.Generator().manual_seed(1)
    hm = torch.randn(3, 24, 14, 18, generator=gen).cuda()                   # non-square map: x runs over the last axis
    gout = torch.randn(3, 24, 2, generator=gen).cuda()
    a = hm.clone().requires_grad_(True)
    ra = softmax_integral_tensor(3.0 * a, 24, 18, 14)
    (ra * gout).sum().backward()
    b = hm.clone().requires_grad_(True)
    rb = iuv_ops.softargmax(b, 3.0)
    (rb * gout).sum().backward()
    assert (ra - rb).abs().max().item() <= 1e-4
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * a.grad.abs().max().item() + 1e-8


def test_fused_smpl_losses_vs_reference_golden_and_tensor_ops():
    """csrc/loss_ops.hip: (a) the reference's own loss values (g10: smpl_losses, keypoint_loss, keypoint_3d_loss, shape_loss,
    l1_losses of smpl_regressor.py:233-298) for the golden inputs; (b) forward AND gradients of all ten terms against the
    tensor-op formulation of smpl_regressor.SMPL_Regressor on random inputs with partially selected rows."""
    from danet_densepose2smpl_amd import loss_ops
    from danet_densepose2smpl_amd.smpl_regressor import SMPL_Regressor as R, _masked_mean
    from danet_densepose2smpl_amd.geometry import perspective_projection
    _cfg()
    gen = torch.Generator().manual_seed(11)
    B, V = 6, 50
    rnd = lambda *s: torch.randn(*s, generator=gen).cuda()
    para = torch.cat([torch.rand(B, 1, generator=gen).cuda() * 0.5 + 0.6, rnd(B, 2) * 0.1, rnd(B, 10), rnd(B, 216) * 0.5], 1)
    target = torch.cat([rnd(B, 3), rnd(B, 10), rnd(B, 216) * 0.5], 1)
    jr0, jp0, jp1, gt_pts = rnd(B, 216) * 0.5, rnd(B, 24, 3), rnd(B, 24, 3), rnd(B, 24, 3)
    joints, verts, tverts = rnd(B, 49, 3) * 0.3, rnd(B, V, 3), rnd(B, V, 3)
    kps2d = torch.cat([rnd(B, 49, 2) * 0.5, torch.rand(B, 49, 1, generator=gen).cuda()], 2)
    kps3d = torch.cat([rnd(B, 24, 3) * 0.3, torch.rand(B, 24, 1, generator=gen).cuda()], 2)
    has_smpl = torch.tensor([1., 0., 1., 1., 0., 1.]).cuda()
    has_kp3d = torch.tensor([0., 1., 1., 0., 1., 1.]).cuda()
    W = {'SMPL_POSE': 60., 'JOINT_POSITION': 1., 'PROJ_KPS': 300., 'KPS3D': 300., 'SMPL_BETAS': 0.06, 'VERTS': 0.7}
    f, S, opw, gtw = 5000., 224., 0.25, 1.0

    def ref(para, jr0, jp0, jp1, joints, verts):
        cam, betas, rot = para[:, :3], para[:, 3:13], para[:, 13:].reshape(B, 24, 3, 3)
        gt_rot = target[:, 13:].reshape(B, 24, 3, 3)
        out = {}
        out['joint_rotation0'] = _masked_mean(((jr0 - target[:, 13:]) ** 2).sum(1), has_smpl, 216) * W['SMPL_POSE']
        out['joint_position0'] = R.l1_losses(jp0, gt_pts, has_smpl) * W['JOINT_POSITION']
        out['joint_position1'] = R.l1_losses(jp1, gt_pts, has_smpl) * W['JOINT_POSITION']
        cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * f / (S * cam[:, 0] + 1e-9)], -1)
        kp = perspective_projection(joints, None, cam_t, f, torch.zeros(B, 2, device='cuda')) / (S / 2.)
        lp, lb = R.smpl_losses(rot, betas, gt_rot, target[:, 3:13], has_smpl)
        out['keypoints_2d'] = R.keypoint_loss(kp, kps2d, opw, gtw) * W['PROJ_KPS']
        out['keypoints_3d'] = R.keypoint_3d_loss(joints, kps3d, has_kp3d) * W['KPS3D']
        out['smpl_pose'], out['smpl_betas'] = lp * W['SMPL_POSE'], lb * W['SMPL_BETAS']
        out['smpl_verts'] = R.shape_loss(verts, tverts, has_smpl) * W['VERTS']
        out['cam'] = (torch.exp(-cam[:, 0] * 10) ** 2).mean()
        return out
    leaves_r = [t.clone().requires_grad_(True) for t in (para, jr0, jp0, jp1, joints, verts)]
    leaves_m = [t.clone().requires_grad_(True) for t in (para, jr0, jp0, jp1, joints, verts)]
    r = ref(*leaves_r)
    m = loss_ops.smpl_losses(leaves_m[0], [leaves_m[1]], [leaves_m[2], leaves_m[3]], leaves_m[4], leaves_m[5], target, gt_pts, tverts,
                             kps2d, kps3d, has_smpl, has_kp3d, f, S, opw, gtw, W)
    assert set(m) == set(r)
    coef = {k: 0.3 + 0.1 * i for i, k in enumerate(sorted(r))}
    sum(r[k] * coef[k] for k in r).backward()
    sum(m[k] * coef[k] for k in m).backward()
    for k in r:
        assert abs(float(m[k]) - float(r[k])) <= 2e-5 * abs(float(r[k])) + 1e-7, (k, float(m[k]), float(r[k]))
    for a, b_, name in zip(leaves_m, leaves_r, 'para jrot0 jpos0 jpos1 joints verts'.split()):
        assert (a.grad - b_.grad).abs().max().item() <= 2e-5 * b_.grad.abs().max().item() + 1e-7, name

    # the reference's own values for its golden inputs (rows selected by has_smpl / has_kp3d)
    g = golden('g10_losses')
    t = lambda k: torch.from_numpy(g[k]).cuda()
    Bg = 6
    para_g = torch.cat([torch.ones(Bg, 3).cuda(), t('pb'), t('pred_rot').reshape(Bg, 216)], 1)
    target_g = torch.cat([torch.zeros(Bg, 3).cuda(), t('gb'), t('gt_rot')], 1)
    z = torch.zeros
    out = loss_ops.smpl_losses(para_g, [para_g[:, 13:]], [t('a'), t('a')], t('pj'), t('pv'), target_g, t('b'), t('gv'),
                               torch.zeros(Bg, 49, 3).cuda(), t('g3'), t('has_smpl').float(), t('has_kp3d').float(), f, S, 0.25, 1.0,
                               {'SMPL_POSE': 1., 'JOINT_POSITION': 1., 'PROJ_KPS': 1., 'KPS3D': 1., 'SMPL_BETAS': 1., 'VERTS': 1.})
    for k, gk in (('smpl_pose', 'loss_pose'), ('joint_rotation0', 'loss_pose'), ('smpl_betas', 'loss_betas'), ('keypoints_3d', 'loss_kp3d'),
                  ('smpl_verts', 'loss_verts'), ('joint_position0', 'loss_l1')):
        np.testing.assert_allclose(float(out[k]), float(g[gk]), rtol=1e-5, err_msg=k)
