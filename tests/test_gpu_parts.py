"""GPU parity of the fused partial-IUV ops (csrc/part_ops.hip) against the tensor-op formulation
(danet.py / iuv_estimator.py of this package, which mirror danet.py:264-283 and iuv_estimator.py:206-246
of the reference) on the same seeded inputs."""
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _inputs(B, S, seed, with_keep=True):
    g = torch.Generator().manual_seed(seed)
    pred = (torch.randn(B, 504, S, S, generator=g) * 1.5).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    keep = (torch.rand(B, 24, 7, generator=g) > 0.3).float().cuda() if with_keep else None
    if keep is not None:
        keep[:, :, 0] = 1.0
    return pred, keep


@pytest.mark.parametrize('B,S,with_keep', [(3, 16, True), (2, 32, False), (1, 7, True)])
def test_part_clean_matches_tensor_ops(B, S, with_keep):
    from danet_densepose2smpl_amd import part_ops
    from danet_densepose2smpl_amd.iuvmap import iuvmap_clean
    pred, keep = _inputs(B, S, 3, with_keep)
    p1 = pred.clone().requires_grad_(True)
    view, x24 = part_ops.part_clean(p1, keep)
    assert x24.shape == (B * 24, 24, S, S) and view.shape == (B, 24, 3, 7, S, S)
    assert float(x24[:, 21:].abs().max()) == 0.0
    # tensor-op reference (fp32 math on the same bf16 prediction)
    p2 = pred.clone().requires_grad_(True)
    pp = p2.reshape(B, 24, 3, 7, S, S)
    if keep is not None:
        pp = pp * keep.view(B, 24, 1, 7, 1, 1)
    flat = pp.reshape(B * 24, 3, 7, S, S)
    u, v, i, _ = iuvmap_clean(flat[:, 0], flat[:, 1], flat[:, 2])
    ref = torch.stack([u, v, i], dim=1).reshape(B, 24, 3, 7, S, S)
    assert torch.equal(view.float(), ref.bfloat16().float())
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5)).bfloat16().cuda()
    (view.float() * gy.float()).sum().backward()
    (ref * gy.float()).sum().backward()
    assert torch.equal(p1.grad.float(), p2.grad.float().bfloat16().float())


@pytest.mark.parametrize('align', [True, False])
def test_part_losses_match_tensor_ops(align):
    from danet_densepose2smpl_amd import part_ops
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator, DP2SMPL_MAPPING
    from danet_densepose2smpl_amd.iuvmap import iuv_img2map
    B, S = 3, 32
    pred, _ = _inputs(B, S, 11, False)
    g = torch.Generator().manual_seed(7)
    part = torch.randint(0, 25, (B, S // 4, S // 4), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2)   # blobs
    img = torch.stack([part.float() / 24., torch.rand(B, S, S, generator=g), torch.rand(B, S, S, generator=g)], 1)
    img[:, 1:] *= (part > 0).float().unsqueeze(1)
    img = img.cuda()
    theta = torch.zeros(B, 24, 2, 3)
    sc = 0.3 + 0.5 * torch.rand(B, 24, generator=g)
    theta[:, :, 0, 0] = sc
    theta[:, :, 1, 1] = sc
    theta[:, :, :, 2] = torch.rand(B, 24, 2, generator=g) * 1.4 - 0.7
    theta = theta.cuda()
    w = torch.tensor([1.0, 0.0, 1.0]).cuda()
    sel = torch.tensor(DP2SMPL_MAPPING, dtype=torch.long).cuda()

    p1 = pred.clone().requires_grad_(True)
    sums = part_ops.part_losses(p1, img, theta, w, sel, align)
    coef = torch.tensor([0.7, 1.3, 2.1], device='cuda')
    (sums * coef).sum().backward()

    # tensor-op formulation
    p2 = pred.clone().requires_grad_(True)
    pp = p2.reshape(B, 24, 3, 7, S, S)
    U, V, I, _ = iuv_img2map(img)
    simp = IUV_Estimator.part_iuv_simp(types.SimpleNamespace(_dp_sel=sel), U, V, I)
    flat = simp.reshape(B * 24, 21, S, S)
    grid = F.affine_grid(theta.reshape(B * 24, 2, 3), list(flat.shape), align_corners=align)
    gt = F.grid_sample(flat, grid, mode='bilinear', padding_mode='zeros', align_corners=align).reshape(B, 24, 3, 7, S, S)
    fg = (gt[:, :, 2] > 0).float() * w.view(B, 1, 1, 1, 1)
    rU = (F.smooth_l1_loss(pp[:, :, 0].float(), gt[:, :, 0], reduction='none') * fg).sum()
    rV = (F.smooth_l1_loss(pp[:, :, 1].float(), gt[:, :, 1], reduction='none') * fg).sum()
    logp = F.log_softmax(pp[:, :, 2].float(), dim=2)
    tgt = torch.argmax(gt[:, :, 2], dim=2, keepdim=True)
    rI = (-logp.gather(2, tgt) * w.view(B, 1, 1, 1, 1)).sum()
    ref = torch.stack([rU, rV, rI])
    (ref * coef).sum().backward()
    assert torch.allclose(sums, ref, rtol=2e-4, atol=1e-2), (sums, ref)
    d = (p1.grad.float() - p2.grad.float()).abs().max().item()
    assert d <= 2e-2 * p2.grad.float().abs().max().item(), d        # bf16 gradients; a resampled weight exactly on the fg>0 edge may flip


def test_group_padded_prediction_layout_matches_unpadded():
    """The part ops on the grouped conv's zero-padded output (24 channels per joint) == on the 21-channel layout."""
    from danet_densepose2smpl_amd import part_ops
    from danet_densepose2smpl_amd.iuv_estimator import DP2SMPL_MAPPING
    B, S = 2, 16
    pred, keep = _inputs(B, S, 21, True)
    pad = torch.zeros(B, S, S, 24, 24, dtype=torch.bfloat16, device='cuda')
    pad[..., :21] = pred.permute(0, 2, 3, 1).reshape(B, S, S, 24, 21)
    pad = pad.reshape(B, S, S, 576).permute(0, 3, 1, 2)
    g = torch.Generator().manual_seed(3)
    img = torch.stack([torch.randint(0, 25, (B, S, S), generator=g).float() / 24., torch.rand(B, S, S, generator=g),
                       torch.rand(B, S, S, generator=g)], 1).cuda()
    theta = torch.zeros(B, 24, 2, 3)
    theta[:, :, 0, 0] = theta[:, :, 1, 1] = 0.6
    theta[:, :, :, 2] = torch.rand(B, 24, 2, generator=g) - 0.5
    theta = theta.cuda()
    sel = torch.tensor(DP2SMPL_MAPPING, dtype=torch.long).cuda()
    outs = []
    for t in (pred, pad):
        p = t.clone().requires_grad_(True)
        v6 = part_ops.padded_view6(p) if p.shape[1] == 576 else p.reshape(B, 24, 3, 7, S, S)
        view, x24 = part_ops.part_clean(v6, keep)
        sums = part_ops.part_losses(v6, img, theta, None, sel, True)
        (x24.float().square().sum() * 0.01 + sums.sum()).backward()
        gr = p.grad.permute(0, 2, 3, 1).reshape(B, S, S, 24, -1)
        if gr.shape[-1] == 24:
            assert float(gr[..., 21:].abs().max()) == 0.0
        outs.append((x24.float(), sums, gr[..., :21].float()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5)
    assert (outs[0][2] - outs[1][2]).abs().max().item() <= 1e-2 * outs[0][2].abs().max().item()


@pytest.mark.parametrize('align', [0, 1])
def test_part_losses_vs_reference_golden(align):
    """csrc/part_ops.hip against the reference itself: the golden vectors of IUV_Estimator.forward (tests/golden/g7_*: the
    reference's part_iuv_simp + affine_grid / grid_sample of the ground truth + per-joint body_uv_losses,
    iuv_estimator.py:206-256,422-445) hold the reference's own partial prediction, key-point centres and the three
    partial losses.  Feeding that prediction (rounded once to bf16, the kernel's input type) and the thetas built from
    those centres must reproduce the reference's losses to the rounding of the input: 2e-3 relative."""
    import numpy as np
    import sys
    from conftest import golden, GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden import formula_params
    from danet_densepose2smpl_amd import part_ops
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict, cfg
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16, 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.,
                   'DANET.PARTDROP_RATE': 0., 'DANET.ALIGN_CORNERS': bool(align)})
    g = golden('g7_estimator_align%d' % align)
    est = IUV_Estimator(pretrained=False)
    with torch.no_grad():
        est.learned_ratio.copy_(torch.from_numpy(g['learned_ratio']))
        est.learned_offset.copy_(torch.from_numpy(g['learned_offset']))
    est = est.cuda().eval()                                         # (eval: no jitter; affine_para only)
    centers = torch.from_numpy(g['stn_kps_pred']).cuda()
    # visibility-driven hidden parts depend on the index head; the golden run has none hidden when scores >= threshold:
    # reproduce the reference's thetas from its own centres and check them through the resampled ground truth below
    from danet_densepose2smpl_amd.iuv_estimator import _sample_points
    from danet_densepose2smpl_amd.iuvmap import iuvmap_clean
    idx = torch.from_numpy(g['index']).cuda()
    index_cl = iuvmap_clean(idx, idx, idx)[2]
    score = _sample_points(torch.einsum('jc,bchw->bjhw', est._vis_membership, index_cl), centers, bool(align))
    hidden = (score < cfg.DANET.STN_PART_VIS_SCORE) if cfg.DANET.STN_PART_VIS_SCORE > 0 else None
    thetas, _ = est.affine_para(centers, hidden)
    pred = torch.from_numpy(g['part_iuv_pred']).cuda()              # [B,24,3,7,S,S] fp32 from the reference
    B, S = pred.shape[0], pred.shape[-1]
    pred_b = pred.reshape(B, 504, S, S).bfloat16().contiguous(memory_format=torch.channels_last)
    sums = part_ops.part_losses(pred_b, torch.from_numpy(g['iuv_gt']).cuda(), thetas, torch.ones(B, device='cuda'), est._dp_sel, bool(align))
    lU = float(sums[0]) / B * cfg.DANET.POINT_REGRESSION_WEIGHTS / 24.
    lV = float(sums[1]) / B * cfg.DANET.POINT_REGRESSION_WEIGHTS / 24.
    lI = float(sums[2]) / (B * 24 * S * S)
    ref = {k: float(g['loss__' + k].sum()) for k in ('loss_pU', 'loss_pV', 'loss_pIndexUV')}
    # only meaningful if the golden run hid no part (then affine_para(centers, None) IS the reference's theta): the
    # resampled ground truth built from these thetas must equal the reference's part_iuv_gt
    from danet_densepose2smpl_amd.iuvmap import iuv_img2map
    uvia = iuv_img2map(torch.from_numpy(g['iuv_gt']).cuda())
    simp = est.part_iuv_simp(*uvia[:3]).reshape(B * 24, 21, S, S)
    grid = F.affine_grid(thetas.reshape(B * 24, 2, 3), list(simp.shape), align_corners=bool(align))
    gt = F.grid_sample(simp, grid, mode='bilinear', padding_mode='zeros', align_corners=bool(align)).reshape(B, 24, 3, 7, S, S)
    if np.abs(gt.cpu().numpy() - g['part_iuv_gt']).max() > 1e-4:
        pytest.fail('thetas rebuilt from the golden centres / index head do not reproduce the reference ground-truth maps')
    for ours, k in ((lU, 'loss_pU'), (lV, 'loss_pV'), (lI, 'loss_pIndexUV')):
        assert abs(ours - ref[k]) <= 2e-3 * abs(ref[k]) + 1e-6, (k, ours, ref[k])


@pytest.mark.parametrize('with_keep,with_w,S', [(True, True, 16), (False, False, 64), (True, False, 12)])
def test_part_joint_equals_the_two_separate_ops(with_keep, with_w, S):
    """part_ops.part_joint (ONE autograd node: its backward is one launch, danet_part_backward_fused) == part_clean + part_losses with
    autograd's add of their two gradients: x24 and the three finished losses bit for bit, d pred within one bf16 rounding of the sum
    (the separate path rounds each term and their sum to bf16, the fused one the sum once); also with only one of the two consumers
    carrying a gradient.  S = 64: the ground-truth image in LDS; S = 12: H * W = 144 is no multiple of 256 (ragged last tile)."""
    from danet_densepose2smpl_amd import part_ops
    from danet_densepose2smpl_amd.iuv_estimator import DP2SMPL_MAPPING
    B = 2
    pred, keep = _inputs(B, S, 21, with_keep)
    pad = torch.zeros(B, S, S, 24, 24, dtype=torch.bfloat16, device='cuda')
    pad[..., :21] = pred.permute(0, 2, 3, 1).reshape(B, S, S, 24, 21)
    pad = pad.reshape(B, S, S, 576).permute(0, 3, 1, 2)
    g = torch.Generator().manual_seed(5)
    img = torch.stack([torch.randint(0, 25, (B, S, S), generator=g).float() / 24., torch.rand(B, S, S, generator=g),
                       torch.rand(B, S, S, generator=g)], 1).cuda()
    theta = torch.zeros(B, 24, 2, 3)
    theta[:, :, 0, 0] = theta[:, :, 1, 1] = 0.6
    theta[:, :, :, 2] = torch.rand(B, 24, 2, generator=g) - 0.5
    theta = theta.cuda()
    w = torch.tensor([1., 0.5]).cuda() if with_w else None
    sel = torch.tensor(DP2SMPL_MAPPING, dtype=torch.long).cuda()
    scales = ((0.3, 0.), (0.7, 0.), (1., 24. * S * S))
    gx = torch.randn(B * 24, 24, S, S, generator=g).cuda().bfloat16()
    cw = torch.tensor([1.3, 0.4, 2.0]).cuda()
    for use_x24, use_loss in ((True, True), (True, False), (False, True)):
        res = []
        for joint in (False, True):
            p = pad.clone().requires_grad_(True)
            v6 = part_ops.padded_view6(p)
            if joint:
                x24, lU, lV, lI = part_ops.part_joint(v6, keep, img, theta, w, sel, True, scales)
            else:
                _, x24 = part_ops.part_clean(v6, keep)
                lU, lV, lI = part_ops.part_losses(v6, img, theta, w, sel, True, scales=scales)
            loss = 0
            if use_x24:
                loss = loss + (x24.float() * gx.float()).sum()
            if use_loss:
                loss = loss + lU.sum() * cw[0] + lV.sum() * cw[1] + lI.sum() * cw[2]
            loss.backward()
            res.append((x24.detach(), torch.cat([lU, lV, lI]).detach(), p.grad.float()))
        assert torch.equal(res[0][0], res[1][0])
        assert torch.equal(res[0][1], res[1][1])
        d, ref = res[1][2] - res[0][2], res[0][2]
        assert float(d.abs().max()) <= 2 ** -7 * float(ref.abs().max()) + 1e-12, (use_x24, use_loss, float(d.abs().max()), float(ref.abs().max()))
        if not (use_x24 and use_loss):
            assert torch.equal(res[0][2], res[1][2])                       # one consumer: the same kernel as the separate op
        gr = res[1][2].permute(0, 2, 3, 1).reshape(B, S, S, 24, 24)
        assert float(gr[..., 21:].abs().max()) == 0.0
