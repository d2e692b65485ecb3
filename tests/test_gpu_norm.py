"""GPU parity of the BatchNorm(+ReLU+residual) and fuse kernels vs plain PyTorch fp32 ops."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, r, rel, what):
    scale = r.abs().max().item() + 1e-6
    err = (a.float() - r.float()).abs().max().item()
    assert err <= rel * scale, '%s: err %g scale %g' % (what, err, scale)


@pytest.mark.parametrize('C,H,B,relu,use_res', [(48, 64, 4, True, False), (96, 32, 4, True, True), (64, 16, 2, False, False),
                                                 (12, 64, 2, True, False), (384, 8, 4, True, True), (3072, 2, 4, True, False), (2048, 4, 8, True, True), (3072, 2, 32, False, False),
                                                 (256, 4, 96, False, True)])
def test_bn_train_forward_backward(C, H, B, relu, use_res):
    from danet_densepose2smpl_amd import nn as dnn
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3).bfloat16().float().cuda()
    res = torch.randn(B, C, H, H, generator=g).bfloat16().float().cuda() if use_res else None
    gy = torch.randn(B, C, H, H, generator=g).bfloat16().float().cuda()
    bn = dnn.BatchNorm2d(C, momentum=0.1).cuda()
    ref = torch.nn.BatchNorm2d(C, momentum=0.1).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
        ref.weight.copy_(bn.weight); ref.bias.copy_(bn.bias)
    xr = x.clone().requires_grad_(True)
    rr = None if res is None else res.clone().requires_grad_(True)
    yr = ref(xr)
    if rr is not None:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(gy)
    xt = x.clone().requires_grad_(True)
    rt = None if res is None else res.clone().requires_grad_(True)
    y = bn(xt, res=rt, relu=relu)
    y.backward(gy.bfloat16())
    _close(y, yr, 1e-2, 'y')
    # masks may differ where |y| < bf16 eps; compare gradients in an L2 sense as well as max
    _close(xt.grad, xr.grad, 3e-2, 'dx')
    if rt is not None:
        _close(rt.grad, rr.grad, 1e-2, 'dres')
    _close(bn.weight.grad, ref.weight.grad, 2e-2, 'dgamma')
    _close(bn.bias.grad, ref.bias.grad, 2e-2, 'dbeta')
    _close(bn.running_mean, ref.running_mean, 1e-3, 'running_mean')
    _close(bn.running_var, ref.running_var, 1e-3, 'running_var')
    # eval mode uses the running statistics
    bn.eval(); ref.eval()
    with torch.no_grad():
        ye = bn(x, relu=relu)
        yre = ref(x)
        yre = F.relu(yre) if relu else yre
    _close(ye, yre, 1e-2, 'eval')


def test_sum_relu_multi_equals_the_per_output_launches():
    """nn.sum_relu_multi (round 5): the four fuse sums of a 4-branch HighResolutionModule -- shifts (0,1,2,3), (0,0,1,2), (0,0,0,1),
    (0,0,0,0), several terms of an output sharing a shift -- in ONE launch forward and ONE backward: outputs and every term's gradient
    equal the per-output launches bit for bit (same kernels' bodies, same fp32 sums)."""
    from danet_densepose2smpl_amd import nn as dnn, conv as dconv
    g = torch.Generator().manual_seed(3)
    B, C = 2, 48
    sizes = [32, 16, 8, 4]

    def make():
        groups, leaves = [], []
        for i in range(4):
            terms, shifts = [], []
            for j in range(4):
                s = j - i if j > i else 0
                t = dconv.nhwc_bf16(torch.randn(B, C, sizes[i] >> s, sizes[i] >> s, generator=torch.Generator().manual_seed(10 * i + j)).cuda()).requires_grad_(True)
                terms.append(t); shifts.append(s); leaves.append(t)
            groups.append((terms, shifts))
        return groups, leaves
    gys = [dconv.nhwc_bf16(torch.randn(B, C, s, s, generator=g).cuda()) for s in sizes]
    res = []
    for multi in (True, False):
        groups, leaves = make()
        ys = dnn.sum_relu_multi(groups, relu=True) if multi else [dnn.sum_relu(t, sh, relu=True) for t, sh in groups]
        torch.autograd.backward(ys, gys)
        torch.cuda.synchronize()
        res.append(([y.detach().clone() for y in ys], [t.grad.clone() for t in leaves]))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    # only two of the outputs receive a gradient: the others' terms get None, the launch holds two jobs
    groups, leaves = make()
    ys = dnn.sum_relu_multi(groups, relu=True)
    torch.autograd.backward([ys[0], ys[2]], [gys[0], gys[2]])
    for k, t in enumerate(leaves):
        assert (t.grad is not None) == (k // 4 in (0, 2))


def test_sum_relu_fuse():
    from danet_densepose2smpl_amd import nn as dnn
    g = torch.Generator().manual_seed(0)
    B, C = 2, 48
    a = torch.randn(B, C, 32, 32, generator=g).bfloat16().float().cuda()
    b = torch.randn(B, C, 16, 16, generator=g).bfloat16().float().cuda()
    c = torch.randn(B, C, 8, 8, generator=g).bfloat16().float().cuda()
    gy = torch.randn(B, C, 32, 32, generator=g).bfloat16().float().cuda()
    ar, br, cr = (t.clone().requires_grad_(True) for t in (a, b, c))
    yr = F.relu(ar + F.interpolate(br, scale_factor=2, mode='nearest') + F.interpolate(cr, scale_factor=4, mode='nearest'))
    yr.backward(gy)
    at, bt, ct = (t.clone().requires_grad_(True) for t in (a, b, c))
    y = dnn.sum_relu([at, bt, ct], [0, 1, 2], relu=True)
    y.backward(gy.bfloat16())
    _close(y, yr, 1e-2, 'y')
    _close(at.grad, ar.grad, 1e-2, 'da')
    _close(bt.grad, br.grad, 1e-2, 'db')
    _close(ct.grad, cr.grad, 1e-2, 'dc')
    # four terms (shifts 0..3, as for the highest-resolution output of a 4-branch module): the one-launch backward equals
    # the per-shift launches bit for bit (same fp32 window sums, rounded once)
    d = torch.randn(B, C, 4, 4, generator=g).bfloat16().float().cuda()
    res = {}
    for mode in (True, False):
        dnn.SUM_BWD_ALL = mode
        ts = [t.clone().requires_grad_(True) for t in (a, b, c, d)]
        y4 = dnn.sum_relu(ts, [0, 1, 2, 3], relu=True)
        y4.backward(gy.bfloat16())
        res[mode] = [t.grad.clone() for t in ts]
    dnn.SUM_BWD_ALL = True
    for u, v in zip(res[True], res[False]):
        assert torch.equal(u, v)
    ar4 = [t.clone().requires_grad_(True) for t in (a, b, c, d)]
    yr4 = F.relu(ar4[0] + sum(F.interpolate(t, scale_factor=2 ** k, mode='nearest') for k, t in enumerate(ar4[1:], 1)))
    yr4.backward(gy)
    for u, v in zip(res[True], ar4):
        _close(u, v.grad, 1e-2, 'd term')
    # plain relu
    xt = a.clone().requires_grad_(True)
    r = dnn.relu(xt)
    r.backward(gy.bfloat16())
    _close(r, F.relu(a), 1e-2, 'relu')
    _close(xt.grad, gy * (a > 0), 1e-2, 'drelu')


@pytest.mark.parametrize('align', [True, False])
def test_stn_gather_vs_torch_grid_sample(align):
    from danet_densepose2smpl_amd import nn as dnn
    g = torch.Generator().manual_seed(3)
    B, C, H, P = 3, 48, 32, 24
    x = torch.randn(B, C, H, H, generator=g).bfloat16().float().cuda()
    s = torch.rand(B, P, generator=g) * 0.9 + 0.05
    s[0, 0] = 0.0                       # degenerate scale
    s[1, 3] = 1.6                       # samples outside the map (zero padding)
    c = (torch.rand(B, P, 2, generator=g) - 0.5) * 1.6
    theta = torch.zeros(B, P, 2, 3)
    theta[:, :, 0, 0] = s; theta[:, :, 1, 1] = s; theta[:, :, :, 2] = c
    theta = theta.cuda()
    gy = torch.randn(B, P * C, H, H, generator=g).bfloat16().float().cuda()
    xr = x.clone().requires_grad_(True)
    outs = []
    for p in range(P):
        grid = F.affine_grid(theta[:, p], x.size(), align_corners=align)
        outs.append(F.grid_sample(xr, grid, align_corners=align))
    yr = torch.cat(outs, dim=1)
    yr.backward(gy)
    xt = x.clone().requires_grad_(True)
    y = dnn.stn_gather(xt, theta, align_corners=align)
    y.backward(gy.bfloat16())
    _close(y, yr, 1e-2, 'stn y')
    _close(xt.grad, xr.grad, 1e-2, 'stn dx')


@pytest.mark.parametrize('Cin,Cout,H,B,k,stride,groups', [(48, 48, 64, 4, 3, 1, 1), (64, 256, 32, 2, 1, 1, 1), (96, 192, 32, 3, 3, 2, 1),
                                                          (384, 384, 8, 2, 3, 1, 1), (40, 24, 17, 3, 3, 1, 1), (64, 64, 16, 2, 3, 1, 4)])
def test_conv_epilogue_bn_statistics(Cin, Cout, H, B, k, stride, groups):
    """BatchNorm statistics accumulated by the conv epilogue == the separate statistics pass on the
    same bf16 conv output.  The LDS-tile 3x3 kernel sums its fp32 accumulators, the separate pass the rounded bf16
    values: the means differ by the mean of the rounding errors (~2^-9 |y| / sqrt(count)), so the bound on the
    running mean is relative to the size of the activations, not to the (near zero) mean itself."""
    from danet_densepose2smpl_amd import conv as dconv, nn as dnn
    torch.manual_seed(0)
    dev = 'cuda'
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin // groups, k, k, device=dev) * 0.1
    bn = dnn.BatchNorm2d(Cout).to(dev).train()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-0.5, 0.5)
    y_f = dconv.conv2d(x, w, None, stride, k // 2, 1, groups, want_stats=True)
    assert getattr(y_f, '_bn_sums', None) is not None
    out_f = bn(y_f, relu=True).float()
    rm_f, rv_f = bn.running_mean.clone(), bn.running_var.clone()
    bn.reset_running_stats()
    y_u = dconv.conv2d(x, w, None, stride, k // 2, 1, groups)
    assert getattr(y_u, '_bn_sums', None) is None
    assert torch.equal(y_f, y_u)
    out_u = bn(y_u, relu=True).float()
    err = (rm_f - bn.running_mean).abs().max().item()
    assert err <= 2e-5 * y_u.float().abs().max().item(), ('running_mean', err)
    _close(rv_f, bn.running_var, 2e-3, 'running_var')
    _close(out_f, out_u, 1e-2, 'bn output')          # bf16 outputs: one ulp where the mean differs in the last fp32 bits


@pytest.mark.parametrize('C,Cout,H,B,relu,use_res', [(48, 48, 32, 4, True, False), (96, 48, 16, 3, True, True), (64, 128, 16, 2, False, False)])
def test_dgrad_epilogue_bn_backward_reduction(C, Cout, H, B, relu, use_res):
    """conv(bn(x)): the conv's data-gradient epilogue reduces the BatchNorm-backward sums; the result must equal
    the separate reduction pass (same bf16 gradient, fp32 atomics in another order)."""
    from danet_densepose2smpl_amd import conv as dconv, nn as dnn
    torch.manual_seed(1)
    dev = 'cuda'
    x0 = torch.randn(B, C, H, H, device=dev)
    r0 = torch.randn(B, C, H, H, device=dev) if use_res else None
    w0 = torch.randn(Cout, C, 3, 3, device=dev) * 0.1
    gy = torch.randn(B, Cout, H, H, device=dev).bfloat16()
    outs = []
    for fuse in (True, False):
        dconv.FUSE_BN_BWD_REDUCE = fuse
        try:
            bn = dnn.BatchNorm2d(C).to(dev).train()
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5, generator=torch.Generator(device=dev).manual_seed(3))
                bn.bias.uniform_(-0.5, 0.5, generator=torch.Generator(device=dev).manual_seed(4))
            x = x0.clone().requires_grad_(True)
            r = None if r0 is None else r0.clone().requires_grad_(True)
            w = w0.clone().requires_grad_(True)
            h = bn(x, r, relu=relu)
            assert (getattr(h, '_bn_ctx', None) is not None)
            y = dconv.conv2d(h, w, None, 1, 1)
            y.backward(gy)
            outs.append((x.grad.float(), None if r is None else r.grad.float(), bn.weight.grad.clone(), bn.bias.grad.clone()))
        finally:
            dconv.FUSE_BN_BWD_REDUCE = False
    (xf, rf, gwf, gbf), (xu, ru, gwu, gbu) = outs
    _close(gwf, gwu, 1e-4, 'd gamma')
    _close(gbf, gbu, 1e-4, 'd beta')
    _close(xf, xu, 1e-2, 'dx')
    if rf is not None:
        assert torch.equal(rf, ru)


def test_multi_batch_norm_matches_per_module():
    """One launch over four differently shaped BatchNorm(+res)(+ReLU) problems == the four separate launches."""
    from danet_densepose2smpl_amd import nn as dnn
    torch.manual_seed(2)
    shapes = [(3, 48, 32, 32), (3, 96, 16, 16), (3, 192, 8, 8), (3, 384, 4, 4)]
    for use_res in (False, True):
        bns_a = [dnn.BatchNorm2d(s[1]).cuda().train() for s in shapes]
        bns_b = [dnn.BatchNorm2d(s[1]).cuda().train() for s in shapes]
        for a, b in zip(bns_a, bns_b):
            with torch.no_grad():
                a.weight.uniform_(0.5, 1.5); a.bias.uniform_(-0.5, 0.5)
            b.load_state_dict(a.state_dict())
        xs = [torch.randn(s, device='cuda') for s in shapes]
        rs = [torch.randn(s, device='cuda') for s in shapes] if use_res else [None] * 4
        gs = [torch.randn(s, device='cuda').bfloat16() for s in shapes]
        xa = [x.clone().requires_grad_(True) for x in xs]
        ra = [None if r is None else r.clone().requires_grad_(True) for r in rs]
        ya = dnn.multi_batch_norm(bns_a, xa, ra, relu=True)
        torch.autograd.backward(ya, gs)
        xb = [x.clone().requires_grad_(True) for x in xs]
        rb = [None if r is None else r.clone().requires_grad_(True) for r in rs]
        yb = [b(x, r, relu=True) for b, x, r in zip(bns_b, xb, rb)]
        torch.autograd.backward(yb, gs)
        for i in range(4):
            _close(ya[i], yb[i], 1e-2, 'y %d' % i)           # (statistics via float atomics: last-bit differences round differently in bf16)
            _close(xa[i].grad, xb[i].grad, 1e-2, 'dx %d' % i)
            _close(bns_a[i].weight.grad, bns_b[i].weight.grad, 1e-4, 'dgamma %d' % i)
            _close(bns_a[i].bias.grad, bns_b[i].bias.grad, 1e-4, 'dbeta %d' % i)
            _close(bns_a[i].running_var, bns_b[i].running_var, 1e-5, 'running_var %d' % i)
            if use_res:
                _close(ra[i].grad, rb[i].grad, 1e-2, 'dres %d' % i)
            assert int(bns_a[i].num_batches_tracked) == 1


@pytest.mark.parametrize('C,H,B,use_res', [(48, 64, 4, False), (96, 32, 4, True), (12, 16, 2, True), (2052, 2, 4, False), (2052, 2, 4, True)])
def test_relu_gate_modes_agree(C, H, B, use_res):
    """The backward's ReLU gate from the forward's byte mask (residual) or recomputed from x (no residual) equals the
    gate read from the saved output y (csrc/norm_act.hip ldmask), for the single and the multi launch."""
    from danet_densepose2smpl_amd import nn as dnn
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3).cuda()
    res = torch.randn(B, C, H, H, generator=g).cuda() if use_res else None
    gy = torch.randn(B, C, H, H, generator=g).bfloat16().cuda()
    wgt, bia = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    out = {}
    for mode in (True, False):
        dnn.RELU_MASK = mode
        try:
            for multi in (False, True):
                if multi and C > 1024:
                    continue
                bn = dnn.BatchNorm2d(C).cuda().train()
                with torch.no_grad():
                    bn.weight.copy_(wgt); bn.bias.copy_(bia)
                xt = x.clone().requires_grad_(True)
                rt = None if res is None else res.clone().requires_grad_(True)
                y = dnn.multi_batch_norm([bn], [xt], [rt], relu=True)[0] if multi else bn(xt, res=rt, relu=True)
                y.backward(gy)
                out[(mode, multi)] = (y.detach(), xt.grad, None if rt is None else rt.grad, bn.weight.grad.clone(), bn.bias.grad.clone())
        finally:
            dnn.RELU_MASK = True
    for multi in (False, True):
        if (True, multi) not in out:
            continue
        a, b = out[(True, multi)], out[(False, multi)]
        if use_res:                      # the gate itself, exactly: d_res = gy where this run's own y is positive
            assert torch.equal(a[2], (gy.float() * (a[0] > 0).float()).to(a[2].dtype))
        if use_res and torch.equal(a[0], b[0]):
            assert torch.equal(a[2], b[2])
        # dx: the per-channel sums are float atomics (order differs from run to run: last-bit differences), while one
        # wrong gate bit is an error of a whole gradient value
        _close(a[1], b[1], 4e-3 if torch.equal(a[0], b[0]) else 2e-2, 'dx')
        _close(a[3], b[3], 1e-5, 'dgamma'); _close(a[4], b[4], 1e-5, 'dbeta')      # (float atomics)


@pytest.mark.parametrize('C,H,B,use_res', [(48, 64, 8, False), (96, 32, 8, True), (192, 16, 4, True), (64, 16, 96, False), (12, 16, 2, True),
                                          (3072, 2, 32, False), (2048, 4, 16, True)])
def test_onepass_backward_matches_two_kernel_backward(C, H, B, use_res):
    """bn_bwd_onepass_kernel (registers across a grid barrier) == reduce + apply kernels: same gate bits, same sums up to
    the atomics' order; the barrier never times out."""
    from danet_densepose2smpl_amd import nn as dnn, conv as dconv
    g = torch.Generator().manual_seed(C + B)
    x = (torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3).cuda()
    res = torch.randn(B, C, H, H, generator=g).cuda() if use_res else None
    gy = torch.randn(B, C, H, H, generator=g).bfloat16().cuda()
    wgt, bia = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    out = {}
    dnn.ONEPASS_STREAM = None          # (a Trainer of an earlier test confines the one-pass launches to its own stream)
    for one in (True, False):
        dnn.ONEPASS = one
        try:
            for multi in ((False, True) if C <= 1024 else (False,)):      # wider than one slab: the single entry point's slab launch
                bn = dnn.BatchNorm2d(C).cuda().train()
                with torch.no_grad():
                    bn.weight.copy_(wgt); bn.bias.copy_(bia)
                xt = x.clone().requires_grad_(True)
                rt = None if res is None else res.clone().requires_grad_(True)
                dconv.FUSION.clear()
                y = dnn.multi_batch_norm([bn], [xt], [rt], relu=True)[0] if multi else bn(xt, res=rt, relu=True)
                y.backward(gy)
                assert (dconv.FUSION.get('bn_bwd_onepass', 0) > 0) == one
                out[(one, multi)] = (xt.grad, None if rt is None else rt.grad, bn.weight.grad.clone(), bn.bias.grad.clone())
        finally:
            dnn.ONEPASS = True
    torch.cuda.synchronize()
    assert not dnn.onepass_error()
    for multi in ((False, True) if C <= 1024 else (False,)):
        a, b = out[(True, multi)], out[(False, multi)]
        _close(a[0], b[0], 4e-3, 'dx')
        if use_res:
            assert torch.equal(a[1], b[1])
        _close(a[2], b[2], 1e-5, 'dgamma'); _close(a[3], b[3], 1e-5, 'dbeta')


@pytest.mark.parametrize('shape', [(3, 64, 32, 32), (2, 16, 9, 7), (5, 8, 1, 6), (2, 128, 16, 16)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('mode', ['bf16', 'fp32'])
def test_maxpool3x3s2_vs_torch(shape, mode):
    """csrc/pool.hip against F.max_pool2d(3, 2, 1): outputs exact; gradients exact where the window maxima are unique, and with
    ties (bf16 inputs drawn from few values) the FIRST maximum in scan order gets the gradient, as torch's kernel does on the
    same tensor."""
    import torch.nn.functional as F
    from danet_densepose2smpl_amd import conv, nn as dnn
    torch.manual_seed(4)
    B, C, H, W = shape
    x = (torch.randn(B, C, H, W, device='cuda') * 4).round() / 4            # few distinct values: plenty of ties
    if mode == 'bf16':
        x = x.bfloat16()
    gy = torch.randn(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device='cuda').to(x.dtype)
    xr = x.float().clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gr, = torch.autograd.grad(yr, xr, gy.float())
    xo = x.clone().requires_grad_(True)
    with conv.precision(mode):
        y = dnn.maxpool3x3s2(xo)
        g, = torch.autograd.grad(y, xo, gy)
    assert y.dtype == x.dtype and torch.equal(y.float(), yr)
    tol = 0.0 if mode == 'fp32' else 2e-2                                    # bf16: the <= 4 gradient terms of a pixel are summed in fp32 and rounded once
    assert float((g.float() - gr).abs().max()) <= tol * float(gr.abs().max()) + 1e-12


@pytest.mark.parametrize('shape', [(32, 576, 16, 16), (3, 24, 9, 7), (2, 2048, 4, 4), (5, 32, 33, 17), (32, 32, 64, 64)], ids=lambda s: 'x'.join(map(str, s)))
def test_channel_sum_vs_torch(shape):
    """danet_channel_sum (bias gradients) == gy.sum(dim=(0, 2, 3)) in fp32, for bf16 and fp32 NHWC tensors."""
    from danet_densepose2smpl_amd import conv
    torch.manual_seed(5)
    for dt in (torch.bfloat16, torch.float32):
        gy = conv.nhwc_as(torch.randn(*shape, device='cuda'), dt)
        ref = gy.float().sum(dim=(0, 2, 3))
        got = conv.channel_sum(gy)
        assert got.dtype == torch.float32 and float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-3



def test_onepass_backward_with_a_co_residency_budget():
    """nn.ONEPASS_MAX_BLOCKS (danet_bn_backward_onepass max_blocks): a data-parallel trainer keeps compute units free for the
    communication library's kernels.  A four-tensor job set that fills the device in one launch is split into launches that
    fit the budget -- same results --, and a single tensor that needs more workgroups than the budget takes the two-kernel
    path instead of a barrier that could not be met."""
    import ctypes
    from danet_densepose2smpl_amd import nn as dnn, conv as dconv, _lib
    g = torch.Generator().manual_seed(11)
    shapes = [(32, 48, 64, 64), (32, 96, 32, 32), (32, 192, 16, 16), (32, 384, 8, 8)]
    xs = [(torch.randn(*s, generator=g) * 1.5 + 0.3).cuda() for s in shapes]
    gys = [torch.randn(*s, generator=g).bfloat16().cuda() for s in shapes]
    dnn.ONEPASS_STREAM = None
    out = {}
    prev = dnn.ONEPASS_MAX_BLOCKS
    try:
        for cap in (0, 464, 300):
            dnn.ONEPASS_MAX_BLOCKS = cap
            bns = [dnn.BatchNorm2d(s[1]).cuda().train() for s in shapes]
            xt = [x.clone().requires_grad_(True) for x in xs]
            dconv.FUSION.clear()
            ys = dnn.multi_batch_norm(bns, xt, None, relu=True)
            torch.autograd.backward(ys, gys)
            assert dconv.FUSION.get('bn_bwd_onepass', 0) == 4, (cap, dict(dconv.FUSION))
            out[cap] = [t.grad.float() for t in xt] + [b.weight.grad.clone() for b in bns]
        torch.cuda.synchronize()
        assert not dnn.onepass_error()
        for cap in (464, 300):
            for a, b in zip(out[cap][:4], out[0][:4]):
                _close(a, b, 4e-3, 'dx at budget %d' % cap)
            for a, b in zip(out[cap][4:], out[0][4:]):
                _close(a, b, 1e-4, 'dgamma at budget %d' % cap)
        # the planner itself: the 48-channel tensor alone needs > 100 workgroups
        L = _lib.lib()
        job = (_lib.BnBwdJob * 1)()
        j = job[0]
        x, gy = xs[0].bfloat16(), gys[0]
        scratch = torch.zeros(4 * L.danet_bn_ws_floats(48), device='cuda')
        j.dy, j.x, j.y, j.gamma, j.saved = gy.data_ptr(), x.data_ptr(), None, None, scratch.data_ptr()
        j.dx, j.dres, j.dparam, j.red = x.data_ptr(), None, scratch.data_ptr(), scratch.data_ptr()
        j.beta, j.mask, j.mask_mode, j.M, j.C, j.red_state, j.relu = None, None, 0, 32 * 64 * 64, 48, 1, 0
        assert L.danet_bn_backward_onepass_ok(ctypes.addressof(job), 1, 0) == 1
        assert L.danet_bn_backward_onepass_ok(ctypes.addressof(job), 1, 100) == 0
    finally:
        dnn.ONEPASS_MAX_BLOCKS = prev
