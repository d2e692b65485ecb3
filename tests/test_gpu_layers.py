"""Single layers and blocks of the product against golden vectors of the REFERENCE'S OWN modules (SURVEY.md Appendix G
rows G4 and G5; tests/golden/g4_gcn.npz, g5_layers.npz produced by make_golden.py from /root/reference/models/module/
GCN.py, res_module.py:27-97, hr_module.py:15-179): BasicBlock, Bottleneck (+downsample), the grouped (g=24) BasicBlock
of the limb layer4, a two-branch HighResolutionModule with its fuse layers, strided / 1x1 / stem / 7x7 convolutions,
the three-layer GCN.  Train-mode BatchNorm, B = 2: outputs, input gradients, weight / BatchNorm-parameter gradients and
the updated running statistics.  Each case runs twice: in the fp32 verification mode (1e-4 of scale; gradients 1e-3)
and on the bf16 MFMA path (outputs 2e-2).  bf16 gradients through ReLU blocks are compared by a trimmed relative RMS of
0.12: ~0.25 % of the inner activations lie within bf16 rounding of zero and switch their ReLU, and every input-gradient
element gathers 9 x C of them (measured 0.05-0.09; the fp32 mode is the exactness check, the plain convolutions -- no
ReLU -- agree to 5e-2 max)."""
import re
import sys

import numpy as np
import pytest
import torch

from conftest import golden, GOLDEN
sys.path.insert(0, GOLDEN)
from make_golden import formula_params, formula_input    # noqa: E402

pytestmark = pytest.mark.gpu


def _cfg():
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})


def _expect(g, key):
    """(array, stride): a golden entry `key` or its strided-sample form `key__s<stride>`."""
    if key in g.files:
        return g[key], 1
    for k in g.files:
        m = re.match(re.escape(key) + r'__s(\d+)$', k)
        if m:
            return g[k], int(m.group(1))
    raise KeyError(key)


def _close(a, g, key, tol, what, robust=False):
    """max |a - ref| <= tol * max |ref|.  robust (bf16 gradients): a ReLU whose pre-activation lies within bf16 rounding
    of zero may switch, which changes the gradient AT THAT ELEMENT by its full size -- a handful of such elements is
    arithmetic, not a defect (0.3 % switched elements alone are a 5 % relative RMS) -- so the bound is on the relative RMS
    error of the best 97 % of the elements and on the share of elements beyond the tolerance (<= 3 %)."""
    ref, stride = _expect(g, key)
    a = a.detach().float().cpu().flatten()[::stride].numpy() if stride > 1 else a.detach().float().cpu().numpy()
    ref = ref.reshape(a.shape)
    scale = np.abs(ref).max() + 1e-6
    err = np.abs(a - ref)
    if robust:
        e = np.sort(err.flatten())
        keep = e[:max(1, int(0.97 * e.size))]                       # the switched-ReLU elements sit in the top few per cent
        rms = np.sqrt((keep ** 2).mean()) / (np.sqrt((ref ** 2).mean()) + 1e-12)
        share = (err > tol * scale).mean()
        assert rms <= tol and share <= 0.03 + (0.03 if tol > 0.1 else 0.0), '%s %s: trimmed relative RMS %g, %.2f %% of elements beyond %g of scale' % (what, key, rms, 100 * share, tol)
    else:
        assert err.max() <= tol * scale, '%s %s: max err %g vs scale %g' % (what, key, err.max(), scale)


def _build(tag):
    from danet_densepose2smpl_amd import hrnet, resnet, conv
    import torch.nn as nn
    if tag == 'basic48':
        return resnet.BasicBlock(48, 48), [('g5.basic48', (2, 48, 16, 16))]
    if tag == 'bottle64':
        ds = nn.Sequential(conv.Conv2d(64, 256, 1, bias=False), resnet.BatchNorm2d(256, momentum=0.1))
        return resnet.Bottleneck(64, 64, 1, ds), [('g5.bottle64', (2, 64, 16, 16))]
    if tag == 'basic_g24':
        ds = nn.Sequential(conv.Conv2d(256 * 24, 128 * 24, 1, 2, bias=False, groups=24), resnet.BatchNorm2d(128 * 24, momentum=0.1))
        return resnet.BasicBlock(256, 128, 2, ds, groups=24), [('g5.basic_g24', (2, 256 * 24, 4, 4))]
    if tag == 'hrm2':
        return hrnet.HighResolutionModule(2, resnet.BasicBlock, [4, 4], [48, 96], [48, 96], 'SUM', True), \
            [('g5.hrm2a', (2, 48, 16, 16)), ('g5.hrm2b', (2, 96, 8, 8))]
    m = re.match(r'conv_(\d+)_(\d+)_k(\d+)_s(\d+)$', tag)
    ci, co, k, st = map(int, m.groups())
    hw = {(48, 96): 16, (384, 48): 4, (3, 64): 32, (64, 64): 16}[(ci, co)]
    return conv.Conv2d(ci, co, k, st, (k - 1) // 2, bias=False), [('g5.conv%d%d%d' % (ci, co, k), (2, ci, hw, hw))]


TAGS = ['basic48', 'bottle64', 'basic_g24', 'hrm2', 'conv_48_96_k3_s2', 'conv_384_48_k1_s1', 'conv_3_64_k3_s2', 'conv_64_64_k7_s2']


@pytest.mark.parametrize('mode', ['fp32', 'bf16', 'bf16-gather'])
@pytest.mark.parametrize('tag', TAGS)
def test_block_vs_reference_module_golden(tag, mode):
    """'bf16-gather': the 3x3 layers pinned to the gather kernel (conv_fast.hip) -- also the path on which the residual
    gradient (conv.ResLink) is added by a tensor op instead of the LDS-tile kernel's epilogue."""
    from danet_densepose2smpl_amd import conv, _lib
    _cfg()
    if mode == 'bf16-gather':
        prev = _lib.lib().danet_conv3x3_set(0, -1, -1, 0, -1)
        try:
            _run_block_case(tag, 'bf16')
        finally:
            _lib.lib().danet_conv3x3_set(prev, -1, -1, 0, -1)
    else:
        _run_block_case(tag, mode)


def _run_block_case(tag, mode):
    from danet_densepose2smpl_amd import conv
    g = golden('g5_layers')
    mod, inputs = _build(tag)
    formula_params(mod)
    mod = mod.cuda().train()
    xs = [formula_input(n, s, -1.0, 1.0).cuda().requires_grad_(True) for n, s in inputs]
    t_out, t_grad = (1e-4, 1e-3) if mode == 'fp32' else (2e-2, 0.2 if tag == 'hrm2' else 0.12)      # (hrm2: four blocks deep per branch + fuse)
    with conv.precision(mode):
        ys = mod(list(xs) if len(xs) > 1 else xs[0])
        ys = list(ys) if isinstance(ys, (list, tuple)) else [ys]
        loss = 0
        for i, y in enumerate(ys):
            loss = loss + (y.float() * formula_input('%s.w%d' % (tag, i), tuple(y.shape), -1.0, 1.0).cuda()).sum()
        loss.backward()
        conv.flush_wgrads()
    for i, y in enumerate(ys):
        _close(y, g, '%s__y%d' % (tag, i), t_out, mode)
    for i, x in enumerate(xs):
        _close(x.grad, g, '%s__dx%d' % (tag, i), t_grad if not tag.startswith('conv_') or mode == 'fp32' else 5e-2, mode, robust=(mode == 'bf16' and not tag.startswith('conv_')))
    n = 0
    for k, p in mod.named_parameters():
        if p.grad is not None and p.dim() in (1, 4):
            _close(p.grad, g, '%s__grad__%s' % (tag, k.replace('.', '__')), t_grad if not tag.startswith('conv_') or mode == 'fp32' else 5e-2, mode, robust=(mode == 'bf16' and not tag.startswith('conv_')))
            n += 1
    for k, b in mod.named_buffers():
        if k.endswith('running_mean') or k.endswith('running_var'):
            _close(b, g, '%s__buf__%s' % (tag, k.replace('.', '__')), 1e-4 if mode == 'fp32' else 1e-2, mode)
    assert n >= 1


def test_gcn_vs_reference_module_golden():
    """G4: the regressor's three-layer GCN (GCN.py:12-92) with BatchNorm1d(24) per node, refinement adjacency from g3."""
    from danet_densepose2smpl_amd import gcn
    g = golden('g4_gcn')
    net = gcn.GCN(128, 256, 128, 3, 24)
    formula_params(net)
    net = net.cuda().train()
    x = formula_input('g4.x', (4, 24, 128), -1.0, 1.0).cuda().requires_grad_(True)
    y = net(x, torch.from_numpy(g['A']).cuda())
    (y * formula_input('g4.w', tuple(y.shape), -1.0, 1.0).cuda()).sum().backward()
    _close(y, g, 'y', 1e-4, 'gcn')
    _close(x.grad, g, 'x_grad', 1e-3, 'gcn')
    for k, p in net.named_parameters():
        _close(p.grad, g, 'grad__' + k.replace('.', '__'), 1e-3, 'gcn')


def test_narrow_bottleneck_padded_path_equals_plain_path():
    """resnet.Bottleneck(48, 12) (the heat-map head, res_module.py:364): the block run at width 16 throughout (zero-padded weights,
    BatchNorm with gamma = beta = 0 on the extra channels) == the same block with the activations padded and sliced around every
    convolution: outputs, input / weight / BatchNorm gradients and running statistics; state-dict shapes unchanged."""
    from danet_densepose2smpl_amd import resnet
    torch.manual_seed(7)
    blk = resnet.Bottleneck(48, 12).cuda().train()
    ref = resnet.Bottleneck(48, 12).cuda().train()
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(4, 48, 24, 20, device='cuda')
    gy = torch.randn(4, 48, 24, 20, device='cuda')
    outs = []
    for m, flag in ((blk, True), (ref, False)):
        resnet.PAD_NARROW_BLOCKS = flag
        try:
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            y.backward(gy.to(y.dtype))
        finally:
            resnet.PAD_NARROW_BLOCKS = True
        outs.append((y.float(), xi.grad.float(), {k: p.grad.float() for k, p in m.named_parameters()}, {k: b.clone() for k, b in m.named_buffers()}))
    (y0, g0, p0, b0), (y1, g1, p1, b1) = outs

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))
    assert rel(y0, y1) < 2e-2 and rel(g0, g1) < 4e-2
    for k in p1:
        assert p0[k].shape == p1[k].shape and rel(p0[k], p1[k]) < 5e-2, k
    for k in b1:
        assert b0[k].shape == b1[k].shape
        if b1[k].dtype.is_floating_point:
            assert rel(b0[k].float(), b1[k].float()) < 1e-2, k
    assert blk.bn1.running_mean.shape == (12,) and blk.state_dict()['bn1.running_var'].shape == (12,)

