"""GPU parity of the MFMA convolution kernels against a plain PyTorch fp32 reference of the same
op (F.conv2d on the bf16-rounded operands).  Tolerances: the kernels accumulate in fp32 and round
the output once to bf16 (rel 2^-9), so fwd/dgrad must agree to 1e-2 of the output scale;
wgrad is fp32 out: 2e-3 of scale (atomics reorder the fp32 sum)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, stride, pad, groups, H, W, bias) -- the HRNet-W48 shapes of SURVEY.md A.2 plus the
# regressor / partial-IUV shapes; B kept small
SHAPES = [
    (48, 48, 3, 1, 1, 1, 64, 64, False), (96, 96, 3, 1, 1, 1, 32, 32, False),
    (192, 192, 3, 1, 1, 1, 16, 16, False), (384, 384, 3, 1, 1, 1, 8, 8, False),
    (64, 64, 3, 1, 1, 1, 64, 64, False), (256, 48, 3, 1, 1, 1, 64, 64, False),
    (64, 256, 1, 1, 0, 1, 64, 64, False), (48, 96, 3, 2, 1, 1, 64, 64, False),
    (256, 96, 3, 2, 1, 1, 64, 64, False), (256, 64, 1, 1, 0, 1, 64, 64, False),
    (64, 64, 3, 2, 1, 1, 128, 128, False), (48, 25, 3, 1, 1, 1, 64, 64, True),
    (48, 24, 3, 1, 1, 1, 64, 64, True), (96, 48, 1, 1, 0, 1, 32, 32, False),
    (3, 64, 3, 2, 1, 1, 256, 256, False), (48, 15, 3, 1, 1, 1, 64, 64, True),
    (12, 12, 3, 1, 1, 1, 64, 64, False), (48, 12, 1, 1, 0, 1, 64, 64, False),
    (12, 48, 1, 1, 0, 1, 64, 64, False), (384, 48, 1, 1, 0, 1, 8, 8, False),
    (48 * 24, 21 * 24, 3, 1, 1, 24, 32, 32, True),          # predict_partial_iuv (grouped)
    (64, 64, 7, 2, 3, 1, 64, 64, False),                    # SmplResNet stem
    (75, 64, 1, 1, 0, 1, 64, 64, False), (21, 64, 1, 1, 0, 1, 64, 64, False),
    (256 * 24, 128 * 24, 3, 2, 1, 24, 4, 4, False),         # limb_reslayer grouped layer4
    (256 * 24, 128 * 24, 1, 2, 0, 24, 4, 4, False),
    (128 * 24, 6 * 24, 1, 1, 0, 24, 1, 1, True),            # grouped 1x1 pose regressors
    (128, 256, 3, 2, 1, 1, 16, 16, False), (512, 512, 3, 1, 1, 1, 2, 2, False),
    (96, 64, 3, 2, 1, 1, 17, 17, False), (64, 128, 1, 2, 0, 1, 15, 15, False),      # odd sizes: ragged parity classes of the strided dgrad
]


def _ref_inputs(shape, B, seed):
    Cin, Cout, k, stride, pad, groups, H, W, bias = shape
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin // groups, k, k, generator=g) / np.sqrt(k * k * Cin / groups)).bfloat16().float()
    b = torch.randn(Cout, generator=g) if bias else None
    return x.cuda(), w.cuda(), None if b is None else b.cuda()


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s[:8])))
def test_conv_forward_backward_vs_torch_fp32(shape):
    from danet_densepose2smpl_amd import conv
    Cin, Cout, k, stride, pad, groups, H, W, bias = shape
    B = 2 if H * W >= 1024 else 4
    x, w, b = _ref_inputs(shape, B, 1234)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = None if b is None else b.clone().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=True, allow_tf32=False):
        yr = F.conv2d(xr, wr, br, stride, pad, 1, groups)
    gy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(7)).bfloat16().float().cuda()
    yr.backward(gy)

    xt = x.clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    bt = None if b is None else b.clone().requires_grad_(True)
    y = conv.conv2d(xt, wt, bt, stride, pad, 1, groups)
    assert y.dtype == torch.bfloat16 and y.shape == yr.shape
    y.backward(gy.bfloat16())

    def close(a, r, rel, what):
        scale = r.abs().max().item() + 1e-6
        err = (a.float() - r).abs().max().item()
        assert err <= rel * scale, '%s: max err %g vs scale %g (%s)' % (what, err, scale, shape)
    close(y, yr, 1e-2, 'forward')
    close(xt.grad, xr.grad, 1e-2, 'dgrad')
    close(wt.grad, wr.grad, 3e-3, 'wgrad')
    if b is not None:
        close(bt.grad, br.grad, 2e-3, 'dbias')


def test_conv_fp32_output_and_full_batch():
    """B=32 at the dominant HRNet shape, fp32 epilogue (used by the heads)."""
    from danet_densepose2smpl_amd import conv
    shape = (48, 48, 3, 1, 1, 1, 64, 64, True)
    x, w, b = _ref_inputs(shape, 32, 5)
    y = conv.conv2d(x, w, b, 1, 1, 1, 1, out_fp32=True)
    yr = F.conv2d(x, w, b, 1, 1)
    assert y.dtype == torch.float32
    assert (y - yr).abs().max().item() <= 2e-3 * yr.abs().max().item()


def test_conv_rejects_bad_shapes():
    from danet_densepose2smpl_amd import conv
    with pytest.raises(ValueError):
        conv.conv2d(torch.zeros(1, 5, 8, 8, device='cuda'), torch.zeros(4, 3, 3, 3, device='cuda'))


@pytest.mark.parametrize('cin,cout,k,pad,outpad,H', [(2048, 256, 4, 1, 0, 4), (256, 256, 4, 1, 0, 16), (64, 32, 3, 1, 1, 8), (32, 48, 2, 0, 0, 8)])
def test_conv_transpose_vs_torch_fp32(cin, cout, k, pad, outpad, H):
    """PoseResNet deconv head (res_module.py:169-194): k4 s2 p1, plus the k3/k2 variants of _get_deconv_cfg."""
    from danet_densepose2smpl_amd.deconv import ConvTranspose2d
    g = torch.Generator().manual_seed(cin + k)
    x = torch.randn(2, cin, H, H, generator=g).bfloat16().float().cuda()
    m = ConvTranspose2d(cin, cout, k, 2, pad, outpad, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_((torch.randn(m.weight.shape, generator=g) / np.sqrt(cin * k * k / 4)).bfloat16().float())
    xr = x.clone().requires_grad_(True)
    wr = m.weight.detach().clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, 2, pad, outpad)
    gy = torch.randn(yr.shape, generator=g).bfloat16().float().cuda()
    yr.backward(gy)
    xt = x.clone().requires_grad_(True)
    y = m(xt)
    assert y.shape == yr.shape
    y.backward(gy.bfloat16())

    def close(a, r, rel, what):
        scale = r.abs().max().item() + 1e-6
        err = (a.float() - r).abs().max().item()
        assert err <= rel * scale, '%s: max err %g vs scale %g' % (what, err, scale)
    close(y, yr, 1e-2, 'deconv forward')
    close(xt.grad, xr.grad, 1e-2, 'deconv dgrad')
    close(m.weight.grad, wr.grad, 3e-3, 'deconv wgrad')


def test_weight_bank_matches_individual_packing():
    """One batched launch packs every recorded weight exactly like danet_conv_pack_weights does."""
    from danet_densepose2smpl_amd import conv as dconv
    torch.manual_seed(0)
    convs = [dconv.Conv2d(48, 48, 3, padding=1, bias=False), dconv.Conv2d(48, 96, 3, stride=2, padding=1, bias=False),
             dconv.Conv2d(96, 40, 1, bias=True), dconv.Conv2d(40, 40, 3, padding=1, groups=5, bias=False),
             dconv.Conv2d(40, 64, 7, stride=2, padding=3, bias=False), dconv.Conv2d(64, 25, 3, padding=1, bias=True),
             dconv.Conv2d(25, 512, 1, bias=False), dconv.Conv2d(512, 72, 3, padding=1, groups=2, bias=False)]
    convs = [c.cuda() for c in convs]
    bank = dconv.WeightBank()
    bank.start_recording()
    dconv.RECORDER = bank
    x = torch.randn(2, 48, 16, 16, device='cuda', requires_grad=True)
    y = x
    for c in convs:
        y = c(y)
    y.float().sum().backward()
    odd = [torch.nn.Parameter(torch.randn(25, 12, 3, 3, device='cuda')), torch.nn.Parameter(torch.randn(16, 3, 7, 7, device='cuda'))]
    for w in odd:                               # channel counts that are not multiples of 8: the per-element launch
        bank.note(w, 1, 0, 0)
        bank.note(w, 1, 1, 0)
    bank.build()
    assert dconv.RECORDER is None and len(bank.entries) >= 8          # forward + dgrad operand of each weight
    assert any(k[0][4] is not None for k in bank.entries)             # channel-padded layers (96 -> 40, 64 -> 25, 25 -> 512): packed at the padded
    #                                                                   widths straight from the parameter (pad_to), recorded like the others
    assert 0 < bank.total_bricks and 0 < bank.total                   # both the brick and the per-element launch are exercised
    with torch.no_grad():
        for w in [c.weight for c in convs] + odd:
            w.mul_(1.5)                                               # an "optimizer step"
    bank.refresh()
    torch.cuda.synchronize()
    for (key, wref, view, _) in bank.entries:
        w = wref()
        hit = dconv._PACK_CACHE[key]
        assert hit[1].data_ptr() == view.data_ptr() and hit[0] == w._version
        dconv._PACK_CACHE.pop(key)
        ref = dconv.pack_weight(w, key[2], key[1], key[3], key[4])
        assert torch.equal(view[:ref.numel()].view(torch.int16), ref.view(torch.int16)), key
    dconv._PACK_CACHE.clear()


def test_deferred_multi_problem_wgrad_matches_immediate():
    """Weight gradients queued during backward and computed by the multi-problem launches (conv.flush_wgrads)
    equal the per-layer launches (fp32; only the partial-sum order differs)."""
    from danet_densepose2smpl_amd import conv as dconv
    torch.manual_seed(0)
    cfgs = [(48, 48, 3, 1, 1, 32), (96, 96, 3, 1, 1, 16), (192, 192, 3, 1, 1, 8), (48, 96, 3, 2, 1, 32), (64, 256, 1, 1, 0, 16),
            (256, 64, 1, 1, 0, 16), (64, 64, 7, 2, 3, 32), (48, 24, 3, 1, 1, 32),
            # channel-padded layers (widths no multiple of 8: the IUV heads, the heat-map head's Bottleneck(48, 12)): computed at the padded
            # widths by the same multi-problem launches and cropped into .grad afterwards (round 5; they used to run inside the backward chain)
            (48, 25, 3, 1, 1, 32), (48, 12, 1, 1, 0, 32), (12, 12, 3, 1, 1, 32), (12, 48, 1, 1, 0, 32), (75, 64, 1, 1, 0, 16),
            # 4 x 4 maps (the regressor tails): the transpose-read kernel's pair mode when the batch is even, the generic kernel otherwise
            (256, 256, 3, 1, 1, 4), (64, 128, 3, 1, 1, 4)]
    convs = [dconv.Conv2d(ci, co, k, s, p, bias=False).cuda() for ci, co, k, s, p, _ in cfgs]
    xs = [torch.randn(4 if h == 4 else 3, ci, h, h, device='cuda') for ci, _, _, _, _, h in cfgs]

    def run(defer):
        for c in convs:
            c.weight.grad = None
        dconv.DEFER_WGRAD = defer
        try:
            loss = sum((c(x).float() * torch.cos(torch.arange(c(x).numel(), device='cuda').view_as(c(x)) * 0.01)).sum() for c, x in zip(convs, xs))
            loss.backward()
        finally:
            dconv.DEFER_WGRAD = False
        if defer:
            assert len(dconv._WQ) + len(dconv._WQG) == len(convs)
        dconv.flush_wgrads()
        torch.cuda.synchronize()
        return [c.weight.grad.clone() for c in convs]
    ref = run(False)
    got = run(True)
    for r, g, cfg in zip(ref, got, cfgs):
        assert (g - r).abs().max().item() <= 1e-3 * r.abs().max().item() + 1e-6, cfg


def test_multi_problem_wgrad3x3_direct_and_reduced_jobs():
    """danet_conv_wgrad3x3_multi on the four HRNet branch shapes at B = 32 (the production launch): the deep layers' jobs take no
    pixel split and write their one block straight into dW (Wg3P.direct: no partial copy, no reduction pass), the shallow ones go
    through partial blocks + the fixed-order reduction -- both equal torch's fp32 weight gradient; with beta = 1 (accumulation)
    every job takes the reduced form and adds to what dW held; two runs are bit-identical."""
    import ctypes
    from danet_densepose2smpl_amd import conv as dconv, _lib
    from danet_densepose2smpl_amd._lib import ptr, stream, check
    L = _lib.lib()
    torch.manual_seed(5)
    B = 32
    shapes = [(48, 64), (96, 32), (192, 16), (384, 8)] * 3           # 12 jobs of one kernel instance in one launch
    xs = [dconv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda')) for c, s in shapes]
    gs = [dconv.nhwc_bf16(torch.randn(B, c, s, s, device='cuda') * 0.1) for c, s in shapes]
    refs = [torch.nn.grad.conv2d_weight(x.float(), (c, c, 3, 3), g.float(), stride=1, padding=1) for x, g, (c, s) in zip(xs, gs, shapes)]

    def run(beta, init):
        outs = [torch.full((c, c, 3, 3), init, device='cuda') for c, s in shapes]
        jobs = (_lib.Wg3Job * len(shapes))()
        for j, x, g, o, (c, s) in zip(jobs, xs, gs, outs, shapes):
            j.x, j.dy, j.dw = x.data_ptr(), g.data_ptr(), o.data_ptr()
            j.B, j.H, j.W, j.Cin, j.Cout, j.groups, j.stride = B, s, s, c, c, 1, 1
        need = L.danet_conv_wgrad3x3_multi_ws_floats(ctypes.addressof(jobs), len(shapes))
        ws = torch.full((need,), float('nan'), device='cuda')
        check(L.danet_conv_wgrad3x3_multi(ctypes.addressof(jobs), len(shapes), ptr(ws), need, beta, stream()), 'wgrad3x3_multi')
        torch.cuda.synchronize()
        return outs
    a, b = run(0.0, float('nan')), run(0.0, 7.0)
    for o, o2, r, sh in zip(a, b, refs, shapes):
        assert torch.isfinite(o).all() and torch.equal(o, o2), sh
        assert (o - r).abs().max().item() <= 2e-4 * r.abs().max().item(), sh
    for o, r, sh in zip(run(1.0, 0.5), refs, shapes):
        assert (o - (r + 0.5)).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-5, sh


@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('B,Cin,Cout,groups,osz', [(768, 256, 256, 1, 4), (32, 256, 256, 1, 4), (6, 64, 32, 1, 4), (10, 48, 96, 1, 4), (4, 96, 48, 2, 4),
                                                   (2, 16, 16, 1, 4), (768, 128, 256, 1, 4),
                                                   # 2 x 2 output maps: eight images per chunk (body_net layer4, the grouped limb layer4)
                                                   (32, 512, 512, 1, 2), (32, 3072, 3072, 24, 2), (8, 64, 32, 1, 2), (16, 256, 512, 1, 2), (24, 48, 48, 1, 2)],
                         ids=lambda v: str(v))
def test_wgrad3x3_pair_mode_two_4x4_images_per_chunk(B, Cin, Cout, groups, osz, stride):
    """csrc/conv_wgrad3x3.hip small-map mode (round 6): 3x3 weight gradients on 4 x 4 (two images per chunk) and 2 x 2 (eight) output maps -- limb_net layer3 over the 768 part
    crops (res_module.py:393-464; 0.56 ms per step on the generic gather kernel before) -- with two images sharing one 4 x 8 chunk, each
    behind a zero halo of its own.  Against torch's fp32 weight gradient on the bf16-rounded operands; alone, next to ordinary jobs in
    the same call, accumulated (beta = 1), and twice (bit-identical)."""
    import ctypes
    from danet_densepose2smpl_amd import conv as dconv, _lib
    from danet_densepose2smpl_amd._lib import ptr, stream, check
    L = _lib.lib()
    S = osz * stride                                # input size: osz x osz OUTPUT maps (stride 2: limb_net layer3's first block, 8 x 8 -> 4 x 4)
    assert L.danet_conv_wgrad3x3_pair_ok(B, S, S, Cin, Cout, 3, 3, stride, 1, 1, groups) == 1
    assert L.danet_conv_wgrad3x3_ok(S, S, Cin, Cout, 3, 3, stride, 1, 1, groups) == 0
    assert L.danet_conv_wgrad3x3_pair_ok(B + 1, S, S, Cin, Cout, 3, 3, stride, 1, 1, groups) == 0          # an odd batch has no partner image
    torch.manual_seed(B + Cin)
    x = dconv.nhwc_bf16(torch.randn(B, Cin, S, S, device='cuda'))
    g = dconv.nhwc_bf16(torch.randn(B, Cout, osz, osz, device='cuda') * 0.1)
    ref = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin // groups, 3, 3), g.float(), stride=stride, padding=1, groups=groups)
    # an ordinary 8-wide job of the same instance family rides in the same call
    x2 = dconv.nhwc_bf16(torch.randn(4, Cin, 8 * stride, 8 * stride, device='cuda'))
    g2 = dconv.nhwc_bf16(torch.randn(4, Cout, 8, 8, device='cuda') * 0.1)
    ref2 = torch.nn.grad.conv2d_weight(x2.float(), (Cout, Cin // groups, 3, 3), g2.float(), stride=stride, padding=1, groups=groups)

    def run(beta, init):
        o = torch.full((Cout, Cin // groups, 3, 3), init, device='cuda')
        o2 = torch.full((Cout, Cin // groups, 3, 3), init, device='cuda')
        jobs = (_lib.Wg3Job * 2)()
        for j, (xx, gg, oo, bb, hh) in zip(jobs, ((x, g, o, B, S), (x2, g2, o2, 4, 8 * stride))):
            j.x, j.dy, j.dw = xx.data_ptr(), gg.data_ptr(), oo.data_ptr()
            j.B, j.H, j.W, j.Cin, j.Cout, j.groups, j.stride = bb, hh, hh, Cin, Cout, groups, stride
        need = L.danet_conv_wgrad3x3_multi_ws_floats(ctypes.addressof(jobs), 2)
        ws = torch.full((max(int(need), 1),), float('nan'), device='cuda')
        check(L.danet_conv_wgrad3x3_multi(ctypes.addressof(jobs), 2, ptr(ws), need, beta, stream()), 'wgrad3x3_multi')
        torch.cuda.synchronize()
        return o, o2
    a, b = run(0.0, float('nan')), run(0.0, 3.0)
    assert torch.isfinite(a[0]).all() and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert (a[0] - ref).abs().max().item() <= 2e-4 * ref.abs().max().item(), float((a[0] - ref).abs().max() / ref.abs().max())
    assert (a[1] - ref2).abs().max().item() <= 2e-4 * ref2.abs().max().item()
    c = run(1.0, 0.5)
    assert (c[0] - (ref + 0.5)).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-5


# (Cin, Cout, H, W, B): HRNet branch shapes at 256^2 and 224^2 inputs (row widths 64..8 and 56..7), the regressor
# trunks (small images: several per tile), a non-square image, channel counts with 1..4 output tiles per block
C3_SHAPES = [(48, 48, 64, 64, 4), (96, 96, 32, 32, 4), (192, 192, 16, 16, 8), (384, 384, 8, 8, 8),
             (48, 48, 56, 56, 2), (96, 96, 28, 28, 3), (192, 192, 14, 14, 4), (384, 384, 7, 7, 6),
             (64, 64, 16, 16, 24), (128, 128, 8, 8, 24), (256, 256, 4, 4, 48), (512, 512, 2, 2, 32),
             (64, 64, 64, 64, 2), (48, 32, 20, 12, 3), (16, 16, 8, 8, 2), (48, 24, 64, 64, 2), (80, 48, 16, 16, 5)]


@pytest.mark.parametrize('shape', C3_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('tiling', [(0, 0), (8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (4, 4)], ids=lambda t: 'mt%d_kw%d' % t)
def test_lds_tile_3x3_kernel_vs_torch_fp32(shape, tiling):
    """conv3x3.hip (persistent LDS-tile kernel) against F.conv2d in fp32 on the bf16-rounded operands: forward with the
    fused BatchNorm statistics, data gradient, for the planner's tiling and every forced register tiling / K split
    the shape admits (a tiling that cannot run falls back to the gather kernel, which must agree as well)."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    Cin, Cout, H, W, B = shape
    g = torch.Generator().manual_seed(Cin * 7 + H)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float().cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)).bfloat16().float().cuda()
    gy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float().cuda()
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w, None, 1, 1)
    yr.backward(gy)
    L.danet_conv3x3_set(1, tiling[0], tiling[1], 0, -1)
    try:
        xt = x.clone().requires_grad_(True)
        wt = w.clone().requires_grad_(True)
        y = dconv.conv2d(xt, wt, None, 1, 1, want_stats=True)
        sums = getattr(y, '_bn_sums', None)
        y.backward(gy.bfloat16())
        torch.cuda.synchronize()
    finally:
        L.danet_conv3x3_set(1, 0, 0, 0, -1)

    def close(a, r, rel, what):
        scale = r.abs().max().item() + 1e-6
        err = (a.float() - r).abs().max().item()
        assert err <= rel * scale, '%s: max err %g vs scale %g (%s, tiling %s)' % (what, err, scale, shape, tiling)
    close(y, yr, 1e-2, 'forward')
    close(xt.grad, xr.grad, 1e-2, 'dgrad')
    close(wt.grad, _wgrad_ref(x, w, gy), 3e-3, 'wgrad')
    if Cout % 8 == 0:
        assert sums is not None
        # the epilogue accumulates the statistics from its fp32 accumulators (conv3x3.hip) or from the bf16-rounded values
        # (conv_fast.hip): both sit within the rounding noise of the fp32 result's sums
        s = dconv.bn_sums_total(sums, Cout)
        yf = yr.detach()
        close(s[0], yf.sum(dim=(0, 2, 3)), 5e-3, 'statistics: sum')
        close(s[1], (yf * yf).sum(dim=(0, 2, 3)), 2e-3, 'statistics: sum of squares')


def _wgrad_ref(x, w, gy):
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, 1, 1).backward(gy)
    return wr.grad


def test_lds_tile_3x3_bias_fp32_relu_and_lockstep_launch():
    """Head-style epilogue (bias, fp32 output) on the LDS-tile kernel, and the four HRNet branches in ONE launch
    (forward and data gradient) against their separate launches."""
    from danet_densepose2smpl_amd import conv as dconv
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 48, 64, 64, generator=g).bfloat16().float().cuda()
    w = (torch.randn(25, 48, 3, 3, generator=g) / 20).bfloat16().float().cuda()
    b = torch.randn(25, generator=g).cuda()
    y = dconv.conv2d(x, w, b, 1, 1, out_fp32=True)
    yr = F.conv2d(x, w, b, 1, 1)
    assert y.dtype == torch.float32 and (y - yr).abs().max().item() <= 2e-3 * yr.abs().max().item()

    chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
    convs = [dconv.Conv2d(c, c, 3, 1, 1, bias=False).cuda() for c in chans]
    xs = [torch.randn(4, c, s, s, device='cuda').bfloat16().float() for c, s in zip(chans, sizes)]
    gys = [torch.randn(4, c, s, s, device='cuda').bfloat16() for c, s in zip(chans, sizes)]
    xa = [t.clone().requires_grad_(True) for t in xs]
    ys = dconv.multi_conv(convs, xa)
    torch.autograd.backward(ys, gys)
    for c, x0, xi, yi, gyi in zip(convs, xs, xa, ys, gys):
        xr = x0.clone().requires_grad_(True)
        yr = F.conv2d(xr, c.weight.detach().bfloat16().float(), None, 1, 1)
        yr.backward(gyi.float())
        assert (yi.float() - yr).abs().max().item() <= 1e-2 * yr.abs().max().item()
        assert (xi.grad.float() - xr.grad).abs().max().item() <= 1e-2 * xr.grad.abs().max().item()


@pytest.mark.parametrize('B,H,Cin,Cout', [(64, 64, 64, 64), (160, 64, 64, 64), (36, 96, 64, 128)])
def test_filter_row_wgrad_7x7_stride2(B, H, Cin, Cout):
    """conv_wgrad_rows.hip (7x7 stride-2 stems of the regressor ResNets at part-crop batch sizes) == the generic weight
    gradient kernel (same bf16 operands, fp32 accumulation; only the summation order differs) and == torch fp32."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    from danet_densepose2smpl_amd._lib import ptr, stream, check
    L = _lib.lib()
    torch.manual_seed(B)
    OH = H // 2
    assert L.danet_conv_wgrad_rows_ok(B, H, H, Cin, OH, OH, Cout, 7, 7, 2, 3, 1, 1)
    x = dconv.nhwc_bf16(torch.randn(B, Cin, H, H, device='cuda'))
    gy = dconv.nhwc_bf16(torch.randn(B, Cout, OH, OH, device='cuda'))
    xp, gyp = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
    gw = torch.full((Cout, Cin, 7, 7), float('nan'), device='cuda')
    nws = L.danet_conv_wgrad_rows_ws_floats(B, OH, OH, Cin, Cout, 7, 7, 1)
    ws = torch.full((nws,), float('nan'), device='cuda')
    check(L.danet_conv_wgrad_rows(ptr(xp), ptr(gyp), ptr(gw), ptr(ws), nws, B, H, H, Cin, OH, OH, Cout, 7, 7, 2, 3, 1, 0.0, stream()), 'rows')
    ref = torch.empty_like(gw)
    n2 = L.danet_conv_wgrad_ws_floats(Cout, Cin, 7, 7)
    ws2 = torch.empty(n2, device='cuda')
    check(L.danet_conv_wgrad(ptr(xp), ptr(gyp), ptr(ref), ptr(ws2), n2, B, H, H, Cin, OH, OH, Cout, 7, 7, 2, 3, 1, 1, 0.0, 0, stream()), 'generic')
    scale = ref.abs().max().item()
    assert torch.isfinite(gw).all()
    assert (gw - ref).abs().max().item() <= 2e-4 * scale
    if B <= 64:
        t = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin, 7, 7), gy.float(), stride=2, padding=3)
        assert (gw - t).abs().max().item() <= 2e-4 * scale
    # accumulate form
    check(L.danet_conv_wgrad_rows(ptr(xp), ptr(gyp), ptr(gw), ptr(ws), nws, B, H, H, Cin, OH, OH, Cout, 7, 7, 2, 3, 1, 1.0, stream()), 'rows')
    assert (gw - 2 * ref).abs().max().item() <= 4e-4 * scale


@pytest.mark.parametrize('B,H,W,Cin,Cout,groups', [(4, 64, 64, 48, 96, 1), (3, 32, 32, 96, 192, 1), (2, 16, 16, 192, 384, 1),
                                                   (5, 64, 64, 48, 48, 1), (2, 128, 128, 64, 64, 1), (6, 16, 16, 64, 128, 1),
                                                   (2, 32, 48, 80, 48, 2), (2, 64, 64, 256, 96, 1)])
def test_transpose_read_wgrad_3x3_stride2(B, H, W, Cin, Cout, groups):
    """conv_wgrad3x3.hip at stride 2 (HRNet transitions / fuse downsamples, ResNet stage entries) == the generic weight
    gradient kernel == torch fp32."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    from danet_densepose2smpl_amd._lib import ptr, stream, check
    L = _lib.lib()
    torch.manual_seed(B + Cin)
    OH, OW = H // 2, W // 2
    assert L.danet_conv_wgrad3x3_ok(H, W, Cin, Cout, 3, 3, 2, 1, 1, groups)
    x = dconv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))
    gy = dconv.nhwc_bf16(torch.randn(B, Cout, OH, OW, device='cuda'))
    xp, gyp = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
    gw = torch.full((Cout, Cin // groups, 3, 3), float('nan'), device='cuda')
    nws = L.danet_conv_wgrad3x3_ws_floats(B, H, W, Cin, Cout, groups, 2)
    ws = torch.full((nws,), float('nan'), device='cuda')
    check(L.danet_conv_wgrad3x3(ptr(xp), ptr(gyp), ptr(gw), ptr(ws), nws, B, H, W, Cin, Cout, groups, 2, 0.0, 0, stream()), 'wgrad3x3 s2')
    t = torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin // groups, 3, 3), gy.float(), stride=2, padding=1, groups=groups)
    scale = t.abs().max().item()
    assert torch.isfinite(gw).all()
    assert (gw - t).abs().max().item() <= 2e-4 * scale
    # multi-problem form, mixed with a stride-1 problem of the same tile shape
    x1 = dconv.nhwc_bf16(torch.randn(B, Cin, OH, OW, device='cuda'))
    gw_m, gw_1 = torch.empty_like(gw), torch.empty_like(gw)
    jobs = (_lib.Wg3Job * 2)()
    for j, (xx, hh, ww, st, out) in zip(jobs, ((x, H, W, 2, gw_m), (x1, OH, OW, 1, gw_1))):
        j.x, j.dy, j.dw = xx.data_ptr(), gy.data_ptr(), out.data_ptr()
        j.B, j.H, j.W, j.Cin, j.Cout, j.groups, j.stride = B, hh, ww, Cin, Cout, groups, st
    import ctypes
    need = L.danet_conv_wgrad3x3_multi_ws_floats(ctypes.addressof(jobs), 2)
    wsm = torch.empty(need, device='cuda')
    check(L.danet_conv_wgrad3x3_multi(ctypes.addressof(jobs), 2, ptr(wsm), need, 0.0, stream()), 'multi')
    assert (gw_m - t).abs().max().item() <= 2e-4 * scale
    t1 = torch.nn.grad.conv2d_weight(x1.float(), (Cout, Cin // groups, 3, 3), gy.float(), stride=1, padding=1, groups=groups)
    assert (gw_1 - t1).abs().max().item() <= 2e-4 * t1.abs().max().item()


def test_fused_addend_on_the_gather_kernel():
    """The residual-branch gradient added in the data-gradient epilogue of a 1x1 convolution (conv_fast.hip; the Bottleneck
    blocks' first convolution): y = conv(x) + addend, rounded once."""
    from danet_densepose2smpl_amd import conv, _lib
    L = _lib.lib()
    torch.manual_seed(0)
    B, H, W, Cin, Cout = 4, 24, 20, 64, 256
    gy = conv.nhwc_bf16(torch.randn(B, Cin, H, W, device='cuda'))                 # dgrad of a 256 -> 64 1x1 conv: gathers 64, produces 256
    w = torch.nn.Parameter(torch.randn(Cin, Cout, 1, 1, device='cuda') * 0.1)
    add = conv.nhwc_bf16(torch.randn(B, Cout, H, W, device='cuda'))
    wp1 = conv.pack_weight(w, 1, 1)
    assert L.danet_conv_forward_kernel(B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, 1, 0) % 10 == 1
    y = conv._conv_fwd_raw(gy, wp1, None, B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, True, False, False, None, None, add)
    ref = torch.nn.functional.conv_transpose2d(gy.float(), w.detach().bfloat16().float(), None, 1, 0) + add.float()
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 8e-3



def test_production_launches_run_on_the_streamed_kernel_and_match_torch_fp32():
    """What the benched step launches, at ITS size: the four HRNet-W48 branch shapes at B = 32 -- each
    alone and the four-branch lockstep launch -- must be taken by conv3x3_stream_kernel (plan != 0, kernel id 3: no silent
    fall-back to the tile or gather kernels), and that exact launch -- forward with the fused BatchNorm statistics, and the
    data gradient -- must match F.conv2d in fp32 on the same bf16-rounded operands at 1e-2 of scale."""
    import ctypes
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    B = 32
    chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
    g = torch.Generator().manual_seed(77)
    xs = [torch.randn(B, c, s, s, generator=g).bfloat16().cuda() for c, s in zip(chans, sizes)]
    ws = [(torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)).bfloat16().float().cuda() for c in chans]
    gys = [torch.randn(B, c, s, s, generator=g).bfloat16().cuda() for c, s in zip(chans, sizes)]
    refs = []
    for x, w, gy in zip(xs, ws, gys):
        xr = x.float().requires_grad_(True)
        yr = F.conv2d(xr, w, None, 1, 1)
        yr.backward(gy.float())
        refs.append((yr.detach(), xr.grad))

    def close(a, r, rel, what):
        scale = r.abs().max().item() + 1e-6
        err = (a.float() - r).abs().max().item()
        assert err <= rel * scale, '%s: max err %g vs scale %g' % (what, err, scale)

    xn = [dconv.nhwc_bf16(x) for x in xs]
    gn = [dconv.nhwc_bf16(gy) for gy in gys]
    wp0 = [dconv.pack_weight(torch.nn.Parameter(w), 1, 0) for w in ws]
    wp1 = [dconv.pack_weight(torch.nn.Parameter(w), 1, 1) for w in ws]
    for nprob in (1, 4):
        for c, s in zip(chans, sizes):
            plan = L.danet_conv3x3_stream_plan(B, s, s, c, c, nprob)
            assert plan != 0 and plan % 10 == 3, (c, s, nprob, plan)          # taken, NT = 3 (48-channel N-blocks)
    # single launches through the entry point every Conv2d uses
    for i, (c, s) in enumerate(zip(chans, sizes)):
        dims = (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1)
        for tr_, xin, wpk in ((False, xn[i], wp0[i]), (True, gn[i], wp1[i])):
            one = (_lib.ConvJob * 1)()
            dconv._conv_job(one[0], xin, wpk, xin, dims, tr_)
            assert L.danet_conv_forward_multi_kernel(ctypes.addressof(one), 1) == 3, (c, s, tr_)       # 3 = conv3x3_stream_kernel
        sums = torch.zeros(L.danet_bn_ws_floats(c), device='cuda')
        y = dconv._conv_fwd_raw(xn[i], wp0[i], None, B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1, False, False, False, sums)
        gx = dconv._conv_fwd_raw(gn[i], wp1[i], None, B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1, True, False, False)
        close(y, refs[i][0], 1e-2, 'forward %d' % c)
        close(gx, refs[i][1], 1e-2, 'dgrad %d' % c)
        st = dconv.bn_sums_total(sums, c)
        close(st[0], refs[i][0].sum(dim=(0, 2, 3)), 5e-3, 'statistics: sum %d' % c)
        close(st[1], (refs[i][0] ** 2).sum(dim=(0, 2, 3)), 2e-3, 'statistics: squares %d' % c)
    # the four-branch lockstep launch (forward + statistics, then the data gradients)
    for transposed in (False, True):
        jobs = (_lib.ConvJob * 4)()
        ys = [torch.empty_like(x) for x in xn]
        sums = [torch.zeros(L.danet_bn_ws_floats(c), device='cuda') for c in chans]
        for j, x, gy, a, b, y, c, s, sm in zip(jobs, xn, gn, wp0, wp1, ys, chans, sizes, sums):
            dconv._conv_job(j, gy if transposed else x, b if transposed else a, y, (B, s, s, c, s, s, c, 3, 3, 1, 1, 1, 1), transposed,
                            None if transposed else sm)
        assert L.danet_conv_forward_multi_kernel(ctypes.addressof(jobs), 4) == 3          # conv3x3_stream_kernel
        dconv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), 4, _lib.stream()), 'multi')
        torch.cuda.synchronize()
        for i, c in enumerate(chans):
            close(ys[i], refs[i][1 if transposed else 0], 1e-2, 'four-branch %s %d' % ('dgrad' if transposed else 'forward', c))
            if not transposed:
                st = dconv.bn_sums_total(sums[i], c)
                close(st[0], refs[i][0].sum(dim=(0, 2, 3)), 5e-3, 'four-branch statistics: sum %d' % c)
                close(st[1], (refs[i][0] ** 2).sum(dim=(0, 2, 3)), 2e-3, 'four-branch statistics: squares %d' % c)


PW_SHAPES = [  # (Cin, Cout, H, W, B): Bottleneck projections, the limb regressor's input projection, the narrow head bottlenecks, ragged sizes
    (64, 256, 64, 64, 32), (256, 64, 64, 64, 32), (24, 64, 64, 64, 96), (64, 24, 64, 64, 96), (48, 16, 64, 64, 32), (16, 48, 64, 64, 32),
    (96, 192, 24, 20, 19), (128, 128, 33, 31, 9), (40, 72, 50, 50, 4), (256, 48, 32, 32, 10)]


@pytest.mark.parametrize('shape', PW_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_pointwise_kernel_vs_torch_fp32(shape):
    """csrc/conv_pw.hip (1x1 / stride-1 layers: the layer's weights in LDS, persistent workgroups) is what danet_conv_forward
    launches for these problems (kernel id ...3), forward and data gradient; against F.conv2d in fp32 on the bf16-rounded
    operands at 1e-2 of scale, fused BatchNorm statistics included; and against the gather kernel (danet_conv_pw_set(0)) on the
    same operands, with bias + ReLU + fp32 output and with the fused bf16 addend."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    Cin, Cout, H, W, B = shape
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)).bfloat16().float().cuda()
    gy = torch.randn(B, Cout, H, W, generator=g).bfloat16().float().cuda()
    assert L.danet_conv_forward_kernel(B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, 0, 0) % 10 == 3
    assert L.danet_conv_forward_kernel(B, H, W, Cout, H, W, Cin, 1, 1, 1, 0, 1, 1, 1, 0) % 10 == 3      # the data gradient
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, w)
    yr.backward(gy)

    def close(a, r, rel, what):
        scale = r.abs().max().item() + 1e-6
        err = (a.float() - r.float()).abs().max().item()
        assert err <= rel * scale, '%s: max err %g vs scale %g (%s)' % (what, err, scale, shape)
    res = {}
    b = torch.randn(Cout, generator=g).cuda()
    add = dconv.nhwc_bf16(torch.randn(B, Cout, H, W, generator=g).cuda())
    for on in (1, 0):
        prev = L.danet_conv_pw_set(on)
        try:
            xt = x.clone().requires_grad_(True)
            wt = w.clone().requires_grad_(True)
            y = dconv.conv2d(xt, wt, None, 1, 0, want_stats=True)
            sums = getattr(y, '_bn_sums', None)
            y.backward(gy.bfloat16())
            xn, wp = dconv.nhwc_bf16(x), dconv.pack_weight(torch.nn.Parameter(w), 1, 0)
            y32 = dconv._conv_fwd_raw(xn, wp, b, B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, False, True, True)
            yadd = dconv._conv_fwd_raw(xn, wp, None, B, H, W, Cin, H, W, Cout, 1, 1, 1, 0, 1, 1, False, False, False, None, None, add)
            torch.cuda.synchronize()
            res[on] = (y.detach(), xt.grad, None if sums is None else dconv.bn_sums_total(sums, Cout), y32, yadd, add, wt.grad.clone())
        finally:
            L.danet_conv_pw_set(prev)
    y, gx, st, y32, yadd, add, gw = res[1]
    close(y, yr, 1e-2, 'forward')
    close(gx, xr.grad, 1e-2, 'dgrad')
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr).backward(gy)
    close(gw, wr.grad, 3e-3, 'wgrad (csrc/conv_pw_wgrad.hip)')
    if Cout % 8 == 0:
        assert st is not None
        yb = y.float()
        close(st[0], yb.sum(dim=(0, 2, 3)), 2e-3, 'statistics: sum')
        close(st[1], (yb * yb).sum(dim=(0, 2, 3)), 2e-3, 'statistics: sum of squares')
    for k, tol, what in ((0, 1e-2, 'forward vs gather kernel'), (1, 1e-2, 'dgrad vs gather kernel'), (3, 2e-3, 'bias + ReLU + fp32'), (4, 1e-2, 'addend')):
        close(res[1][k], res[0][k], tol, what)
    close(yadd, yr.detach() + add.float(), 1.5e-2, 'addend vs fp32')


def test_pointwise_wgrad_kernel_multi_problem_and_switch():
    """csrc/conv_pw_wgrad.hip through danet_conv_wgrad_multi (the trainer's deferred, multi-problem path): several 1x1 layers of
    different widths in one call, with a 3x3 problem mixed in (it stays on the generic kernel), against torch's fp32 weight
    gradient; deterministic (two calls agree bit for bit); and with the kernel switched off the generic path gives the same
    numbers up to summation order.  beta = 1 accumulates."""
    import ctypes
    from danet_densepose2smpl_amd import conv as dconv, _lib
    from danet_densepose2smpl_amd._lib import ptr, stream, check
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    specs = [(64, 256, 1, 8, 64), (256, 64, 1, 8, 64), (24, 64, 1, 40, 64), (64, 24, 1, 40, 64), (48, 16, 1, 4, 64), (16, 48, 1, 4, 64), (32, 32, 3, 4, 32), (96, 192, 1, 6, 30)]
    xs, gys, refs = [], [], []
    for cin, cout, k, B, H in specs:
        x = torch.randn(B, cin, H, H, generator=g).bfloat16().float().cuda()
        gy = torch.randn(B, cout, H, H, generator=g).bfloat16().float().cuda()
        refs.append(torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), gy, padding=k // 2))
        xs.append(dconv.nhwc_bf16(x)); gys.append(dconv.nhwc_bf16(gy))

    def run(beta=0.0, init=None):
        dws = [torch.full((cout, cin, k, k), float('nan'), device='cuda') if init is None else init[i].clone() for i, (cin, cout, k, B, H) in enumerate(specs)]
        jobs = (_lib.WgJob * len(specs))()
        for j, x, gy, dw, (cin, cout, k, B, H) in zip(jobs, xs, gys, dws, specs):
            j.x, j.dy, j.dw = x.data_ptr(), gy.data_ptr(), dw.data_ptr()
            (j.B, j.H, j.W, j.Cin, j.OH, j.OW, j.Cout, j.R, j.S, j.stride, j.pad, j.dil, j.groups) = (B, H, H, cin, H, H, cout, k, k, 1, k // 2, 1, 1)
        need = L.danet_conv_wgrad_multi_ws_floats(ctypes.addressof(jobs), len(specs))
        ws = torch.zeros(need, device='cuda')
        check(L.danet_conv_wgrad_multi(ctypes.addressof(jobs), len(specs), ptr(ws), need, beta, stream()), 'wgrad_multi')
        torch.cuda.synchronize()
        return dws
    a, b = run(), run()
    prev = L.danet_conv_pw_wgrad_set(0)
    try:
        c = run()
    finally:
        L.danet_conv_pw_wgrad_set(prev)
    acc = run(beta=1.0, init=[r.clone() for r in refs])
    for (cin, cout, k, B, H), x1, x2, x3, x4, r in zip(specs, a, b, c, acc, refs):
        scale = r.abs().max().item()
        assert torch.isfinite(x1).all(), (cin, cout)
        if k == 1 and (cin, cout) != (96, 192):                                  # (96 -> 192: 12 x 6 blocks exceed a wave's 16 accumulator tiles: generic kernel)
            assert torch.equal(x1, x2), (cin, cout)                             # deterministic (the generic kernel's atomics are not)
        assert (x1 - r).abs().max().item() <= 3e-3 * scale, (cin, cout, k)
        assert (x1 - x3).abs().max().item() <= 1e-3 * scale, (cin, cout, k)     # == the generic kernel up to order
        assert (x4 - 2 * r).abs().max().item() <= 6e-3 * scale, (cin, cout, k)


@pytest.mark.parametrize('B,H', [(32, 64), (3, 32)])
def test_grouped_partial_iuv_head_runs_on_the_streamed_kernel(B, H):
    """The 24-group partial-IUV head (/root/reference/models/danet/iuv_estimator.py:193-206: 24 x (48 -> 21, padded 24) channels, 3x3) is
    a problem of conv3x3_stream_kernel (kernel id ...4: one group's channels of a pixel tile per tile, the group's 96-byte slice of every
    2304-byte pixel row staged by LDS-DMA) instead of the gather kernel; forward against F.conv2d(groups = 24) in fp32 on the bf16-rounded
    operands, with the fused BatchNorm statistics of all 576 channels, and against the gather kernel itself (stream switched off)."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    G, Cg, Ng = 24, 48, 24
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, G * Cg, H, H, generator=g).bfloat16().cuda()
    w = (torch.randn(G * Ng, Cg, 3, 3, generator=g) / np.sqrt(9 * Cg)).bfloat16().float().cuda()
    dconv.stream_tables(x.device)
    with _lib.knobs(g3=0):          # (round 6: without fused statistics the narrow-group kernel, csrc/conv_g3.hip, takes the head first)
        assert L.danet_conv_forward_kernel(B, H, H, G * Cg, H, H, G * Ng, 3, 3, 1, 1, 1, G, 0, 0) % 10 == 4
    yr = F.conv2d(x.float(), w, None, 1, 1, 1, G)
    xn, wp = dconv.nhwc_bf16(x), dconv.pack_weight(torch.nn.Parameter(w), G, 0)
    bias = torch.randn(G * Ng, generator=g).cuda()
    res = {}
    for on in (1, 0):
        prev = L.danet_conv3x3_stream_set(on, -1, -1, -1)
        prev_g3 = L.knob('g3', 0)
        try:
            sums = torch.zeros(L.danet_bn_ws_floats(G * Ng), device='cuda')
            y = dconv._conv_fwd_raw(xn, wp, None, B, H, H, G * Cg, H, H, G * Ng, 3, 3, 1, 1, 1, G, False, False, False, sums)
            yb = dconv._conv_fwd_raw(xn, wp, bias, B, H, H, G * Cg, H, H, G * Ng, 3, 3, 1, 1, 1, G, False, False, False)      # the head's own form: with a bias
            torch.cuda.synchronize()
            res[on] = (y.float(), dconv.bn_sums_total(sums, G * Ng), yb.float())
        finally:
            L.danet_conv3x3_stream_set(prev, -1, -1, -1)
            L.knob('g3', prev_g3)
    scale = yr.abs().max().item()
    assert (res[1][0] - yr).abs().max().item() <= 1e-2 * scale
    assert (res[1][0] - res[0][0]).abs().max().item() <= 1e-2 * scale
    assert (res[1][2] - (yr + bias.view(1, -1, 1, 1))).abs().max().item() <= 1e-2 * (scale + bias.abs().max().item())
    st = res[1][1]
    assert (st[0] - yr.sum(dim=(0, 2, 3))).abs().max().item() <= 5e-3 * yr.sum(dim=(0, 2, 3)).abs().max().item() + 1e-3 * scale * B * H
    assert (st[1] - (yr * yr).sum(dim=(0, 2, 3))).abs().max().item() <= 3e-3 * (yr * yr).sum(dim=(0, 2, 3)).abs().max().item()


@pytest.mark.parametrize('B,H,W', [(32, 64, 64), (3, 32, 32), (2, 20, 48), (2, 7, 16)])
def test_narrow_group_3x3_kernel_forward_and_data_gradient(B, H, W):
    """csrc/conv_g3.hip (round 6): the 24-group partial-IUV head (iuv_estimator.py:193-206, 24 x (48 -> 21, padded to 24) channels, 3x3)
    forward WITH ITS BIAS and its data gradient (24 -> 48 per group, taps mirrored) -- one (image, group, row band) per workgroup, the
    band in LDS, the group's weights in registers -- selected (kernel id ...5) and compared with F.conv2d / its input gradient in fp32 on
    the bf16-rounded operands and with the kernels that ran these passes before (streamed kernel / gather kernel: knob g3 = 0); row
    counts that do not fill a band, widths of 1 .. 4 fragments; two runs bit-identical."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    G, Cg, Ng = 24, 48, 24
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, G * Cg, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(G * Ng, Cg, 3, 3, generator=g) / np.sqrt(9 * Cg)).bfloat16().float().cuda()
    w.view(G, Ng, Cg, 3, 3)[:, 21:] = 0                                          # the head's three padding channels per group
    bias = torch.randn(G * Ng, generator=g).cuda()
    gy = (torch.randn(B, G * Ng, H, W, generator=g) * 0.1).bfloat16().cuda()
    dconv.stream_tables(x.device)
    assert L.danet_conv_forward_kernel(B, H, W, G * Cg, H, W, G * Ng, 3, 3, 1, 1, 1, G, 0, 0) % 10 == 5
    assert L.danet_conv_forward_kernel(B, H, W, G * Ng, H, W, G * Cg, 3, 3, 1, 1, 1, G, 1, 0) % 10 == 5
    xr = x.float().requires_grad_(True)
    yr = F.conv2d(xr, w, bias, 1, 1, 1, G)
    gxr, = torch.autograd.grad(yr, xr, gy.float())
    wpar = torch.nn.Parameter(w)
    xn, gn = dconv.nhwc_bf16(x), dconv.nhwc_bf16(gy)
    wp0, wp1 = dconv.pack_weight(wpar, G, 0), dconv.pack_weight(wpar, G, 1)
    out = {}
    for on in (1, 1, 0):
        with _lib.knobs(g3=on):
            y = dconv._conv_fwd_raw(xn, wp0, bias, B, H, W, G * Cg, H, W, G * Ng, 3, 3, 1, 1, 1, G, False, False, False)
            gx = dconv._conv_fwd_raw(gn, wp1, None, B, H, W, G * Ng, H, W, G * Cg, 3, 3, 1, 1, 1, G, True, False, False)
            torch.cuda.synchronize()
        out.setdefault(on, []).append((y.float(), gx.float()))
    (y1, gx1), (y2, gx2) = out[1]
    y0, gx0 = out[0][0]
    assert torch.equal(y1, y2) and torch.equal(gx1, gx2)
    sy, sg = yr.abs().max().item(), gxr.abs().max().item()
    assert (y1 - yr).abs().max().item() <= 1e-2 * sy and (gx1 - gxr).abs().max().item() <= 1e-2 * sg
    assert (y1 - y0).abs().max().item() <= 1e-2 * sy and (gx1 - gx0).abs().max().item() <= 1e-2 * sg


@pytest.mark.parametrize('B,Cin', [(768, 64), (64, 64), (96, 32), (70, 16)])
def test_stem_7x7_stride2_lds_tile_kernel_vs_torch_fp32(B, Cin):
    """csrc/conv_stem.hip (7x7 / stride 2 / pad 3 on LDS tiles: 16-channel slabs by LDS-DMA, even / odd column planes, two taps per
    k-step, K split over two waves) against F.conv2d in fp32 on the bf16-rounded operands: every image border (top / bottom strips,
    left / right halo columns), the fused BatchNorm statistics, odd tile counts per workgroup; through Conv2d's autograd function, so
    the data and weight gradients (on their own kernels) are checked in the same call."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    H, Cout = 64, 64
    assert L.danet_conv_stem_ok(B, H, H, Cin, 32, 32, Cout, 7, 7, 2, 3, 1, 1) == 1
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, Cin, H, H, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 7, 7, generator=g) / np.sqrt(49 * Cin)).bfloat16().float().cuda()
    nref = min(B, 40)
    idx = torch.linspace(0, B - 1, nref).long().cuda()
    xr = x[idx].float().requires_grad_(True)
    yr = F.conv2d(xr, w, None, 2, 3)
    xt = x.clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    dconv.FUSION.clear()
    y = dconv.conv2d(xt, wt, None, 2, 3, want_stats=True)
    sums = getattr(y, '_bn_sums', None)
    torch.cuda.synchronize()
    scale = yr.abs().max().item()
    assert (y[idx].float() - yr).abs().max().item() <= 1e-2 * scale
    prev = L.danet_conv_stem_set(0)
    try:
        y2 = dconv.conv2d(x, w, None, 2, 3)                       # the gather kernel on the same operands
    finally:
        L.danet_conv_stem_set(prev)
    assert (y.float() - y2.float()).abs().max().item() <= 1e-2 * scale
    st = dconv.bn_sums_total(sums, Cout)
    yb = y.float()
    assert (st[0] - yb.sum(dim=(0, 2, 3))).abs().max().item() <= 2e-3 * yb.sum(dim=(0, 2, 3)).abs().max().item() + 1e-4 * scale * B
    assert (st[1] - (yb * yb).sum(dim=(0, 2, 3))).abs().max().item() <= 2e-3 * (yb * yb).sum(dim=(0, 2, 3)).abs().max().item()
    if B <= 96:
        gy = torch.randn(B, Cout, 32, 32, generator=g).bfloat16().cuda()
        y.backward(gy)
        xf = x.float().requires_grad_(True)
        wf = w.clone().requires_grad_(True)
        F.conv2d(xf, wf, None, 2, 3).backward(gy.float())
        assert (xt.grad.float() - xf.grad).abs().max().item() <= 1e-2 * xf.grad.abs().max().item()
        assert (wt.grad - wf.grad).abs().max().item() <= 3e-3 * wf.grad.abs().max().item()


@pytest.mark.parametrize('B', [768, 64, 70])
def test_stem_7x7_stride2_data_gradient_lds_tile_kernel(B):
    """csrc/conv_stem_dgrad.hip (the four parity classes of the 7x7 / stride-2 data gradient on one staged dy tile, table-driven k-steps,
    fused BatchNorm-backward sums) against (a) conv_transpose2d in fp32 on the bf16-rounded operands (sampled images, every border strip),
    (b) the gather kernel on the same operands incl. the same fused sums, (c) the sums recomputed from the kernel's own rounded output."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    H, C = 64, 64
    assert L.danet_conv_stem_dgrad_ok(B, H, H, C, 32, 32, C, 7, 7, 2, 3, 1, 1) == 1
    g = torch.Generator().manual_seed(100 + B)
    gy = dconv.nhwc_bf16(torch.randn(B, C, 32, 32, generator=g).cuda())
    w = (torch.randn(C, C, 7, 7, generator=g) / np.sqrt(49 * C / 4)).bfloat16().float().cuda()
    bn_x = dconv.nhwc_bf16(torch.randn(B, C, H, H, generator=g).cuda())
    bn_y = dconv.nhwc_bf16(torch.randn(B, C, H, H, generator=g).cuda())
    saved = torch.cat([torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5]).cuda()
    nred = L.danet_bn_ws_floats(C)
    for gate in (bn_y, None):
        red = torch.zeros(nred, device='cuda')
        gx = dconv._conv_stem_dgrad_raw(gy, dconv.pack_weight(w, 1, 1, 16), B, H, H, C, 32, 32, C, (bn_x, gate, saved, red, 0))
        red2 = torch.zeros(nred, device='cuda')
        gx2 = dconv._conv_fwd_raw(gy, dconv.pack_weight(w, 1, 1), None, B, 32, 32, C, H, H, C, 7, 7, 2, 3, 1, 1, True, False, False, None,
                                  (bn_x, gate, saved, red2, 0))
        torch.cuda.synchronize()
        idx = torch.linspace(0, B - 1, min(B, 24)).long().cuda()
        ref = F.conv_transpose2d(gy[idx].float(), w, None, 2, 3, 1)
        scale = ref.abs().max().item()
        assert (gx[idx].float() - ref).abs().max().item() <= 1e-2 * scale
        assert (gx.float() - gx2.float()).abs().max().item() <= 1e-2 * scale
        gf = gx.float() * ((gate.float() > 0).float() if gate is not None else 1.0)
        xhat = (bn_x.float() - saved[:C].view(1, -1, 1, 1)) * saved[C:].view(1, -1, 1, 1)
        s1, s2 = gf.sum(dim=(0, 2, 3)), (gf * xhat).sum(dim=(0, 2, 3))
        st, st2 = dconv.bn_sums_total(red, C), dconv.bn_sums_total(red2, C)
        tol1, tol2 = 2e-3 * gf.abs().sum(dim=(0, 2, 3)).max().item(), 2e-3 * (gf * xhat).abs().sum(dim=(0, 2, 3)).max().item()
        assert (st[0] - s1).abs().max().item() <= tol1 and (st[1] - s2).abs().max().item() <= tol2
        assert (st[0] - st2[0]).abs().max().item() <= 4 * tol1 and (st[1] - st2[1]).abs().max().item() <= 4 * tol2
    gx3 = dconv._conv_stem_dgrad_raw(gy, dconv.pack_weight(w, 1, 1, 16), B, H, H, C, 32, 32, C, None)
    assert torch.equal(gx3, gx)                                  # no sums requested: the same data gradient, bit for bit


def test_stem_kernels_repeat_bit_identically():
    """The stem kernels count their waits by hand (weight ring in inline asm, LDS-DMA copies, inline-asm MFMAs the hazard recogniser does
    not see): a wait that is one count short shows up as a rare run-to-run difference long before it shows up as a NaN.  No output of
    these kernels goes through atomics, so 400 launches on fixed inputs must be bit-identical (tools/soak.py runs the long version
    over all round-4 kernels)."""
    from danet_densepose2smpl_amd import conv as dconv
    B, C, H = 326, 64, 64                                   # 1 304 tiles over 256 workgroups: ragged tile counts
    g = torch.Generator().manual_seed(5)
    x = dconv.nhwc_bf16(torch.randn(B, C, H, H, generator=g).cuda())
    gy = dconv.nhwc_bf16(torch.randn(B, C, 32, 32, generator=g).cuda())
    w = (torch.randn(C, C, 7, 7, generator=g) * 0.02).cuda()
    wp, wpt = dconv.pack_weight(w, 1, 0, 16), dconv.pack_weight(w, 1, 1, 16)
    for fn in (lambda: dconv._conv_stem_raw(x, wp, B, H, H, C, 32, 32, C), lambda: dconv._conv_stem_dgrad_raw(gy, wpt, B, H, H, C, 32, 32, C, None)):
        ref = fn().clone()
        assert torch.isfinite(ref.float()).all()
        for i in range(400):
            out = fn()
            if i % 8 == 7:
                assert torch.equal(out, ref), i


@pytest.mark.parametrize('B,H,W', [(768, 16, 16), (32, 64, 64), (300, 16, 16), (20, 64, 64), (140, 32, 16)])
def test_conv3x3a_row_tile_kernel_forward_and_data_gradient(B, H, W):
    """csrc/conv3x3a.hip (3x3 / stride 1, 64 -> 64 channels on the stem kernels' machinery: whole-tile LDS-DMA staging, table-driven
    k-steps of one tap x 32 channels, K cut 12 | 6 over two waves, AGPR accumulators): forward with fused output statistics and data
    gradient with (a) the residual addend, (b) the fused BatchNorm-backward sums gated by the BatchNorm's output, (c) by its byte mask
    -- against F.conv2d / conv_transpose2d in fp32 on the bf16-rounded operands and against the kernels that ran these layers before."""
    from danet_densepose2smpl_amd import conv as dconv, _lib
    L = _lib.lib()
    C = 64
    assert L.danet_conv3x3a_ok(B, H, W, C, C, 3, 3, 1, 1, 1, 1) == 1
    g = torch.Generator().manual_seed(B + H)
    x = dconv.nhwc_bf16(torch.randn(B, C, H, W, generator=g).cuda())
    w = (torch.randn(C, C, 3, 3, generator=g) / np.sqrt(9 * C / 4)).bfloat16().float().cuda()
    idx = torch.linspace(0, B - 1, min(B, 24)).long().cuda()
    nred = L.danet_bn_ws_floats(C)
    # ---- forward + statistics
    sums = torch.zeros(nred, device='cuda')
    y = dconv._conv3x3a_raw(x, dconv.pack_weight(w, 1, 0, 16), B, H, W, False, sums)
    y0 = dconv._conv_fwd_raw(x, dconv.pack_weight(w, 1, 0), None, B, H, W, C, H, W, C, 3, 3, 1, 1, 1, 1, False, False, False)
    torch.cuda.synchronize()
    ref = F.conv2d(x[idx].float(), w, None, 1, 1)
    scale = ref.abs().max().item()
    assert (y[idx].float() - ref).abs().max().item() <= 1e-2 * scale
    assert (y.float() - y0.float()).abs().max().item() <= 1e-2 * scale
    yb, st = y.float(), dconv.bn_sums_total(sums, C)
    assert (st[0] - yb.sum(dim=(0, 2, 3))).abs().max().item() <= 2e-3 * yb.abs().sum(dim=(0, 2, 3)).max().item()
    assert (st[1] - (yb * yb).sum(dim=(0, 2, 3))).abs().max().item() <= 2e-3 * (yb * yb).sum(dim=(0, 2, 3)).max().item()
    # ---- data gradient
    gy = dconv.nhwc_bf16(torch.randn(B, C, H, W, generator=g).cuda())
    wpt = dconv.pack_weight(w, 1, 1, 16)
    refg = F.conv_transpose2d(gy[idx].float(), w, None, 1, 1)
    gscale = refg.abs().max().item()
    gx = dconv._conv3x3a_raw(gy, wpt, B, H, W, True)
    assert (gx[idx].float() - refg).abs().max().item() <= 1e-2 * gscale
    add = dconv.nhwc_bf16(torch.randn(B, C, H, W, generator=g).cuda())
    gxa = dconv._conv3x3a_raw(gy, wpt, B, H, W, True, None, None, add)
    assert (gxa[idx].float() - (refg + add[idx].float())).abs().max().item() <= 1e-2 * (gscale + add.float().abs().max().item())
    bn_x = dconv.nhwc_bf16(torch.randn(B, C, H, W, generator=g).cuda())
    bn_y = dconv.nhwc_bf16(torch.randn(B, C, H, W, generator=g).cuda())
    saved = torch.cat([torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5]).cuda()
    mask = (bn_y.permute(0, 2, 3, 1) > 0).to(torch.uint8).contiguous()
    xhat = (bn_x.float() - saved[:C].view(1, -1, 1, 1)) * saved[C:].view(1, -1, 1, 1)
    for gate_t, mode in ((bn_y, 0), (mask, 2), (None, 0)):
        red = torch.zeros(nred, device='cuda')
        gxb = dconv._conv3x3a_raw(gy, wpt, B, H, W, True, None, (bn_x, gate_t, saved, red, mode))
        torch.cuda.synchronize()
        assert torch.equal(gxb, gx)
        gf = gx.float() * ((bn_y.float() > 0).float() if gate_t is not None else 1.0)
        s1, s2 = gf.sum(dim=(0, 2, 3)), (gf * xhat).sum(dim=(0, 2, 3))
        st = dconv.bn_sums_total(red, C)
        assert (st[0] - s1).abs().max().item() <= 2e-3 * gf.abs().sum(dim=(0, 2, 3)).max().item(), mode
        assert (st[1] - s2).abs().max().item() <= 2e-3 * (gf * xhat).abs().sum(dim=(0, 2, 3)).max().item(), mode
    # ---- repeatability of the hand-counted waits
    for _ in range(100):
        out = dconv._conv3x3a_raw(gy, wpt, B, H, W, True)
    assert torch.equal(out, gx)


@pytest.mark.parametrize('shape,groups,pad_to', [((25, 48, 3, 3), 1, (32, 48)), ((64, 75, 7, 7), 1, (64, 80)), ((12, 48, 1, 1), 1, (16, 48)),
                                                 ((12, 12, 3, 3), 1, (16, 16)), ((24 * 21, 48, 3, 3), 24, (24 * 24, 48)), ((15, 3, 3, 3), 1, (16, 8))])
def test_packing_at_padded_widths_equals_packing_a_zero_padded_copy(shape, groups, pad_to):
    """danet_conv_pack_weights_padded (pack_weight pad_to): a weight packed at zero-padded widths straight from the unpadded tensor is
    bit-identical, in every mode and K order, to packing an explicitly zero-padded copy -- and a convolution run through it (wpad) has the
    same output, data gradient and (cropped) weight gradient as one run on the padded copy."""
    from danet_densepose2smpl_amd import conv as dconv
    g = torch.Generator().manual_seed(sum(shape))
    w = torch.randn(*shape, generator=g).cuda()
    Cout, Cin_g = shape[0], shape[1]
    Cp, Cip = pad_to
    wv = w.view(groups, Cout // groups, Cin_g, shape[2], shape[3])
    wpad = F.pad(wv, (0, 0, 0, 0, 0, Cip - Cin_g, 0, Cp // groups - Cout // groups)).reshape(Cp, Cip, shape[2], shape[3]).contiguous()
    for mode in (0, 1):
        for chunk in ((0, 16) if (Cip if mode == 0 else Cp // groups) % 16 == 0 else (0,)):
            a = dconv.pack_weight(w, groups, mode, chunk, pad_to)
            b = dconv.pack_weight(wpad, groups, mode, chunk)
            assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16)), (mode, chunk)
    if groups == 1 and shape[2] == 3:
        B, H = 4, 16
        x = dconv.nhwc_bf16(torch.randn(B, Cip, H, H, generator=g).cuda()).requires_grad_(True)
        gy = dconv.nhwc_bf16(torch.randn(B, Cp, H, H, generator=g).cuda())
        wa = w.clone().requires_grad_(True)
        ya = dconv.Conv2dFunction.apply(x, wa, None, 1, 1, 1, 1, False, None, None, None, pad_to)
        ya.backward(gy)
        gxa, x.grad = x.grad.clone(), None
        wb = wpad.clone().requires_grad_(True)
        yb = dconv.Conv2dFunction.apply(x, wb, None, 1, 1, 1, 1, False)
        yb.backward(gy)
        assert torch.equal(ya, yb) and torch.equal(gxa, x.grad)
        assert wa.grad.shape == w.shape and torch.equal(wa.grad, wb.grad[:Cout, :Cin_g])


@pytest.mark.parametrize('B,chans,sizes', [(32, (48, 96, 192, 384), (64, 32, 16, 8)), (8, (48, 96, 192, 384), (64, 32, 16, 8)),
                                           (32, (48, 96), (64, 32)), (32, (48, 96, 192), (64, 32, 16)), (16, (32, 64), (32, 16)), (8, (16,), (32,))])
@pytest.mark.parametrize('with_res', [False, True])
def test_conv_bn_one_launch_equals_two_launches(B, chans, sizes, with_res):
    """conv -> train-mode BatchNorm (+ identity) -> ReLU of the lockstep HRNet branch layers (/root/reference/models/module/
    res_module.py:39-56, hr_module.py:155-177) as ONE launch (csrc/conv3x3s.hip s3_bn_tail: grid barrier after the last tile,
    every workgroup normalises the tiles it wrote) against the two launches (streamed convolution, then bn_apply_multi): outputs,
    saved mean / invstd, running statistics, ReLU gate bytes and every gradient BIT-IDENTICAL; the HRNet shape sets must actually
    take the one-launch path (no silent fall-back), and the result is checked against a torch fp32 restatement."""
    import copy
    from danet_densepose2smpl_amd import conv as dconv, nn as dnn
    g = torch.Generator().manual_seed(5 + B + len(chans))
    n = len(chans)
    convs = [dconv.Conv2d(c, c, 3, 1, 1, bias=False).cuda() for c in chans]
    bns = [dnn.BatchNorm2d(c).cuda() for c in chans]
    for cv, bn, c in zip(convs, bns, chans):
        with torch.no_grad():
            cv.weight.copy_((torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)).bfloat16().float())
            bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
            bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    xs0 = [torch.randn(B, c, s, s, generator=g).bfloat16().cuda() for c, s in zip(chans, sizes)]
    rs0 = [torch.randn(B, c, s, s, generator=g).bfloat16().cuda() for c, s in zip(chans, sizes)] if with_res else None
    gys = [torch.randn(B, c, s, s, generator=g).bfloat16().cuda() for c, s in zip(chans, sizes)]

    def run(fuse):
        cv2, bn2 = copy.deepcopy(convs), copy.deepcopy(bns)
        for m in cv2 + bn2:
            m.train()
        xs = [x.clone().requires_grad_(True) for x in xs0]
        rs = None if rs0 is None else [r.clone().requires_grad_(True) for r in rs0]
        prev, dnn.CONV_BN = dnn.CONV_BN, fuse
        dconv.FUSION.clear()
        try:
            ys = dnn.multi_conv_bn(cv2, xs, bn2, rs, relu=True)
        finally:
            dnn.CONV_BN = prev
        counts = dict(dconv.FUSION)
        ctxs = [y._bn_ctx for y in ys]
        torch.autograd.backward(ys, gys)
        torch.cuda.synchronize()
        return {'y': [y.detach() for y in ys], 'saved': [c_[2] for c_ in ctxs], 'mask': [c_[3] for c_ in ctxs],
                'rm': [b.running_mean.clone() for b in bn2], 'rv': [b.running_var.clone() for b in bn2],
                'gx': [x.grad for x in xs], 'gr': None if rs is None else [r.grad for r in rs],
                'gw': [c_.weight.grad for c_ in cv2], 'gg': [b.weight.grad for b in bn2], 'gb': [b.bias.grad for b in bn2], 'counts': counts}

    two, one = run(False), run(True)
    assert two['counts'].get('conv_bn_one_launch', 0) == 0
    if chans[0] == 48:
        assert one['counts'].get('conv_bn_one_launch', 0) == n and one['counts'].get('bn_forward_in_conv_launch', 0) == n, one['counts']
    for key in ('y', 'saved', 'rm', 'rv', 'gx', 'gw', 'gg', 'gb') + (('gr',) if with_res else ()):
        for i, (a, b) in enumerate(zip(one[key], two[key])):
            assert torch.equal(a, b), (key, i, float((a.float() - b.float()).abs().max()))
    for i, (a, b) in enumerate(zip(one['mask'], two['mask'])):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), ('mask', i)
    # ... and the values themselves: conv (bf16-rounded output) -> batch statistics -> affine -> + residual -> relu in fp32
    for i, (cv, bn) in enumerate(zip(convs, bns)):
        raw = F.conv2d(xs0[i].float(), cv.weight.detach(), None, 1, 1).bfloat16().float()
        mean, var = raw.mean(dim=(0, 2, 3)), raw.var(dim=(0, 2, 3), unbiased=False)
        ref = (raw - mean[None, :, None, None]) * torch.rsqrt(var + bn.eps)[None, :, None, None] * bn.weight.detach()[None, :, None, None] + bn.bias.detach()[None, :, None, None]
        if with_res:
            ref = ref + rs0[i].float()
        ref = ref.clamp(min=0)
        err = float((one['y'][i].float() - ref).abs().max())
        assert err <= 3e-2 * max(1.0, float(ref.abs().max())), (i, err)
        assert float((one['saved'][i][0] - mean).abs().max()) < 2e-3 and float((one['rm'][i] - (0.9 * bn.running_mean + 0.1 * mean)).abs().max()) < 1e-3
