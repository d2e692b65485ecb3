"""fp32 instantiation of the BatchNorm / fuse-sum / STN kernels against the tensor-op formulations (forward + gradients)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, nn as dnn        # noqa: E402


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def bn_case(B, C, H, W, relu, res):
    torch.manual_seed(1)
    x = torch.randn(B, C, H, W, device='cuda') * 2 + 0.5
    r = torch.randn(B, C, H, W, device='cuda') if res else None
    bn = dnn.BatchNorm2d(C).cuda().train()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.5, 0.5)
    gy = torch.randn(B, C, H, W, device='cuda')
    xs = [x.clone().requires_grad_(True) for _ in range(2)]
    rs = [None if r is None else r.clone().requires_grad_(True) for _ in range(2)]
    yr = F.batch_norm(xs[0], None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
    if r is not None:
        yr = yr + rs[0]
    if relu:
        yr = F.relu(yr)
    gr = torch.autograd.grad(yr, [xs[0], bn.weight, bn.bias] + ([rs[0]] if res else []), gy)
    with conv.precision('fp32'):
        y = bn(xs[1], rs[1], relu)
        g = torch.autograd.grad(y, [xs[1], bn.weight, bn.bias] + ([rs[1]] if res else []), gy)
    print('bn', (B, C, H, W), 'relu', relu, 'res', res, 'y', '%.1e' % rel(y, yr), 'grads', ['%.1e' % rel(a, b) for a, b in zip(g, gr)], y.dtype, flush=True)


def sum_case():
    torch.manual_seed(2)
    ts = [torch.randn(2, 48, 16 >> s, 16 >> s, device='cuda', requires_grad=True) for s in (0, 1, 2)]
    gy = torch.randn(2, 48, 16, 16, device='cuda')
    yr = F.relu(sum(t if s == 0 else F.interpolate(t, scale_factor=2 ** s, mode='nearest') for s, t in enumerate(ts)))
    gr = torch.autograd.grad(yr, ts, gy)
    with conv.precision('fp32'):
        y = dnn.sum_relu(ts, [0, 1, 2], True)
        g = torch.autograd.grad(y, ts, gy)
    print('sum_relu y', '%.1e' % rel(y, yr), 'grads', ['%.1e' % rel(a, b) for a, b in zip(g, gr)], y.dtype, flush=True)


def fan_case():
    x = torch.randn(2, 48, 8, 8, device='cuda', requires_grad=True)
    with conv.precision('fp32'):
        vs = dnn.fan_out(x, 4)
        y = sum((i + 1) * v for i, v in enumerate(vs))
        g, = torch.autograd.grad(y, x, torch.ones_like(y))
    print('fan_out', '%.1e' % rel(g, torch.full_like(g, 10.0)), flush=True)


def stn_case():
    torch.manual_seed(3)
    x = torch.randn(2, 16, 12, 12, device='cuda', requires_grad=True)
    th = torch.zeros(2, 3, 2, 3, device='cuda')
    th[:, :, 0, 0] = torch.rand(2, 3, device='cuda') * 0.5 + 0.3
    th[:, :, 1, 1] = torch.rand(2, 3, device='cuda') * 0.5 + 0.3
    th[:, :, :, 2] = torch.rand(2, 3, 2, device='cuda') - 0.5
    outs = []
    for i in range(3):
        grid = F.affine_grid(th[:, i], [2, 16, 12, 12], align_corners=True)
        outs.append(F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=True))
    yr = torch.cat(outs, 1)
    gy = torch.randn_like(yr)
    gr, = torch.autograd.grad(yr, x, gy)
    with conv.precision('fp32'):
        y = dnn.stn_gather(x, th, align_corners=True)
        g, = torch.autograd.grad(y, x, gy)
    print('stn y', '%.1e' % rel(y, yr), 'dx', '%.1e' % rel(g, gr), y.dtype, flush=True)


if __name__ == '__main__':
    for relu in (False, True):
        for res in (False, True):
            bn_case(2, 48, 16, 16, relu, res)
    bn_case(4, 64, 32, 32, True, False)
    bn_case(3, 256, 7, 5, True, True)
    sum_case()
    fan_case()
    stn_case()
