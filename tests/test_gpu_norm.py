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
                                                 (12, 64, 2, True, False), (384, 8, 4, True, True), (3072, 2, 4, True, False),
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
