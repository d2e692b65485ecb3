"""csrc/glue.hip against the tensor-op formulations it replaces: several zero-pad copies in one launch (forward + gradients, exact)
and the STN parameters of the part crops (affine_para + visibility score, /root/reference/models/danet/iuv_estimator.py:176-186,
262-301) against the vectorised tensor form kept in iuv_estimator.py (1e-6)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_pad_multi_equals_f_pad_forward_and_backward():
    from danet_densepose2smpl_amd.glue import pad_multi
    dev = torch.device('cuda')
    torch.manual_seed(0)
    shapes = [((12, 48, 1, 1), (16, 48, 1, 1)), ((12,), (16,)), ((12, 12, 3, 3), (16, 16, 3, 3)), ((48, 12, 1, 1), (48, 16, 1, 1)),
              ((25, 75, 7, 7), (32, 80, 7, 7))] + [((5 + k,), (8 + k,)) for k in range(14)]          # 19 jobs: two launches
    ts = [torch.randn(*s, device=dev, requires_grad=True) for s, _ in shapes]
    outs = pad_multi([(t, d) for t, (_, d) in zip(ts, shapes)])
    gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(outs, gs)
    for t, (s, d), o, g in zip(ts, shapes, outs, gs):
        pads = []
        for a, b in zip(reversed(s), reversed(d)):
            pads += [0, b - a]
        assert torch.equal(o, F.pad(t.detach(), pads))
        assert torch.equal(t.grad, g[tuple(slice(0, a) for a in s)])
    # grouped view: [groups * Cout_g, Cin_g, 1, 1] -> [groups * (Cout_g + pad), ...] (the part head's 24 x 21 -> 24 x 24 channels)
    w = torch.randn(24 * 21, 48, 1, 1, device=dev, requires_grad=True)
    b = torch.randn(24 * 21, device=dev, requires_grad=True)
    wp, bp = pad_multi([(w, (24, 21, 48, 1), (24, 24, 48, 1), (24 * 24, 48, 1, 1)), (b, (24, 21), (24, 24), (24 * 24,))])
    ref = F.pad(w.detach().view(24, 21, 48, 1, 1), (0, 0, 0, 0, 0, 0, 0, 3)).reshape(24 * 24, 48, 1, 1)
    assert torch.equal(wp, ref) and torch.equal(bp, F.pad(b.detach().view(24, 21), (0, 3)).reshape(-1))
    gw = torch.randn_like(wp)
    wp.backward(gw)
    assert torch.equal(w.grad, gw.view(24, 24, 48, 1, 1)[:, :21].reshape(24 * 21, 48, 1, 1)) and b.grad is None


@pytest.mark.parametrize('align', [True, False])
@pytest.mark.parametrize('vis', [0.0, 0.4])
def test_stn_theta_kernel_equals_affine_para_with_the_visibility_test(align, vis):
    from danet_densepose2smpl_amd.config import cfg
    from danet_densepose2smpl_amd import iuv_estimator as ie
    dev = torch.device('cuda')
    torch.manual_seed(1)
    old = (cfg.DANET.STN_PART_VIS_SCORE, cfg.DANET.STN_SCALE_JITTER)
    cfg.DANET.STN_PART_VIS_SCORE, cfg.DANET.STN_SCALE_JITTER = vis, 0.
    try:
        est = ie.IUV_Estimator(pretrained=False).to(dev)
        est.train()
        with torch.no_grad():
            est.learned_ratio.copy_(torch.rand(24, device=dev) * 2 - 0.3)         # some negative: relu() cuts them
            est.learned_offset.copy_(torch.rand(24, device=dev) * 0.3 - 0.05)
        B, H, W = 7, 64, 64
        centers = (torch.rand(B, 24, 2, device=dev) * 2.4 - 1.2)                  # some outside the image
        am = torch.randint(0, 25, (B, H, W), device=dev, dtype=torch.uint8)
        got = est.stn_theta(centers, am, align)
        hidden = None
        if vis > 0:
            maps = est._vis_membership.t()[am.long()].permute(0, 3, 1, 2)
            hidden = ie._sample_points(maps, centers, align) < vis
            assert 0 < int(hidden[:, 1:].sum()) < hidden[:, 1:].numel()
        want, _ = est.affine_para(centers, hidden)
        assert got.shape == want.shape == (B, 24, 2, 3)
        assert (got - want).abs().max().item() < 1e-6
    finally:
        cfg.DANET.STN_PART_VIS_SCORE, cfg.DANET.STN_SCALE_JITTER = old


def test_stn_theta_jitter_stays_inside_its_band():
    from danet_densepose2smpl_amd.config import cfg
    from danet_densepose2smpl_amd import iuv_estimator as ie
    dev = torch.device('cuda')
    torch.manual_seed(2)
    old = (cfg.DANET.STN_PART_VIS_SCORE, cfg.DANET.STN_SCALE_JITTER)
    cfg.DANET.STN_PART_VIS_SCORE, cfg.DANET.STN_SCALE_JITTER = 0., 0.2
    try:
        est = ie.IUV_Estimator(pretrained=False).to(dev)
        est.train()
        centers = torch.rand(16, 24, 2, device=dev) * 2 - 1
        am = torch.zeros(16, 64, 64, device=dev, dtype=torch.uint8)
        cfg.DANET.STN_SCALE_JITTER = 0.
        base = est.stn_theta(centers, am, True)
        cfg.DANET.STN_SCALE_JITTER = 0.2
        jit = est.stn_theta(centers, am, True)
        r = jit[:, :, 0, 0] / base[:, :, 0, 0]
        assert r.min() >= 0.9 * 0.9 - 1e-5 and r.max() <= 1.1 * 1.1 + 1e-5 and r.std() > 0.02      # two factors in 1 +- 0.1
        assert torch.equal(jit[:, :, :, 2], base[:, :, :, 2]) and torch.equal(jit[:, :, 0, 0], jit[:, :, 1, 1])
    finally:
        cfg.DANET.STN_PART_VIS_SCORE, cfg.DANET.STN_SCALE_JITTER = old
