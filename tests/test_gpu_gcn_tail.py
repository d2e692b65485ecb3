"""csrc/gcn_tail.hip (the regressor's graph tail in one launch per direction) against the torch operations it replaces
(smpl_regressor.DecomposedPredictor.forward below `rot_feats`; /root/reference/models/danet/smpl_regressor.py:846-900): the four outputs,
the gradient of the limb features and of EVERY parameter (graph-convolution weights / biases, BatchNorm1d affine parameters, the edge
importance behind the normalised adjacency, the grouped heads), and the running statistics.  Both sides compute in fp32 with different
summation orders: values within 2e-5 of scale (2e-4 behind rot6d), gradients within 2e-4 in L2 (see _close_grad).  The reference-golden tests of the whole predictor
(tests/test_gpu_fp32.py g9 / g20, tests/test_gpu_models.py) run THROUGH this kernel."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    reset_cfg()
    cfg_from_dict(kw)


def _net(seed):
    from danet_densepose2smpl_amd.smpl_regressor import DecomposedPredictor
    torch.manual_seed(seed)
    pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
    net = DecomposedPredictor(None, (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6), pretrained=False)
    with torch.no_grad():                    # away from the initial values that hide errors (unit BatchNorm scales, all-ones edges, 0.01-gain heads)
        for n, p in net.named_parameters():
            if n.startswith(('r2p_gcn', 'p2r_gcn', 'refine_gcn', 'pose_regressors', 'coord_regressors')):
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.2)
                elif 'regressors' in n:
                    p.copy_(torch.randn_like(p) * 0.05)
        net.edge_importance.copy_(torch.rand(1, 24, 24) * 1.5 - 0.3)          # some entries negative: ReLU gate of the edges
    return net.cuda().train()


def _tail_torch(net, rot_feats):
    """The unfused operations (the code path `fused_tail` returns None for)."""
    from danet_densepose2smpl_amd import gcn_tail
    keep = gcn_tail.GCN_TAIL
    gcn_tail.GCN_TAIL = False
    try:
        return _tail(net, rot_feats)
    finally:
        gcn_tail.GCN_TAIL = keep


def _tail(net, rot_feats):
    """DecomposedPredictor.forward from `rot_feats` on (its body / limb nets are not under test here)."""
    import torch.nn.functional as F
    from danet_densepose2smpl_amd import gcn_tail
    from danet_densepose2smpl_amd.gcn import normalize_undigraph
    from danet_densepose2smpl_amd.geometry import rot6d_to_rotmat
    fused = gcn_tail.fused_tail(net, rot_feats)
    if fused is not None:
        return fused
    nbs = rot_feats.shape[0]
    p0 = net._grouped_head(net.pose_regressors[0], rot_feats).reshape(nbs, -1) + net.mean_pose
    jr0 = rot6d_to_rotmat(p0).reshape(nbs, -1)
    pos_init = net.r2p_gcn(rot_feats, net.r2p_A[0])
    jp0 = net._grouped_head(net.coord_regressors[0], pos_init)
    graph_A = net.A_mask * F.relu(net.edge_importance)
    norm_A = normalize_undigraph(net.I_n[0] + graph_A)[0]
    pos_ref = pos_init + net.refine_gcn(pos_init, norm_A)
    jp1 = net._grouped_head(net.coord_regressors[1], pos_ref)
    rot_ref = net.p2r_gcn(pos_ref, net.p2r_A[0])
    pose6 = net._grouped_head(net.pose_regressors[-1], rot_ref).reshape(nbs, -1) + net.mean_pose
    return jr0, jp0, jp1, rot6d_to_rotmat(pose6).reshape(nbs, -1)


def _close(a, b, tol, what):
    a, b = a.detach(), b.detach()
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert err <= tol * scale, '%s: err %.3g of scale %.3g' % (what, err, scale)
    return err / scale


def _close_grad(a, b, what):
    """Gradients: a ReLU gate whose pre-activation is within rounding of zero may open on one side and not on the other (a handful of
    the 0.5 M gates per pass) -- one element's worth of difference: the maximum within 5e-3 of scale, the L2 difference within 2e-4."""
    a, b = a.detach().double(), b.detach().double()
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    assert l2 <= (2e-4 if a.numel() >= 1000 else 2e-3), '%s: relative L2 difference %.3g' % (what, l2)       # (24-element vectors: one gate is 1/24 of the norm)
    _close(a, b, 5e-3, what)
    return l2


@pytest.mark.parametrize('B', [32, 4, 1, 19])
def test_fused_tail_matches_the_torch_operations(B):
    _cfg()
    from danet_densepose2smpl_amd import conv as dconv, nn as dnn
    from conftest import record
    dnn.ONEPASS_STREAM = None
    ref = _net(11 + B)
    net = copy.deepcopy(ref)
    g = torch.Generator().manual_seed(B)
    x = (torch.randn(B, 24, 128, generator=g).abs() * 0.7).cuda()          # (a mean over ReLU outputs in the model: non-negative)
    ws = [torch.randn(B, 216, generator=g).cuda(), torch.randn(B, 24, 3, generator=g).cuda(), torch.randn(B, 24, 3, generator=g).cuda(),
          torch.randn(B, 216, generator=g).cuda()]
    outs = {}
    for name, m, fn in (('torch', ref, _tail_torch), ('fused', net, _tail)):
        xt = x.clone().requires_grad_(True)
        dconv.FUSION.clear()
        o = fn(m, xt)
        assert (dconv.FUSION.get('gcn_tail', 0) == 1) == (name == 'fused')
        sum((a * w).sum() for a, w in zip(o, ws)).backward()
        outs[name] = (o, xt.grad)
    torch.cuda.synchronize()
    assert not dnn.onepass_error()
    meas = {}
    for k, a, b in zip(('jr0', 'jp0', 'jp1', 'pose'), outs['fused'][0], outs['torch'][0]):
        assert a.shape == b.shape
        meas[k] = _close(a, b, 2e-4 if k in ('jr0', 'pose') else 2e-5, k)       # (rot6d: two normalisations amplify the heads' rounding)
    meas['d_rot_feats'] = _close_grad(outs['fused'][1], outs['torch'][1], 'd rot_feats')
    pr, pn = dict(ref.named_parameters()), dict(net.named_parameters())
    checked = 0
    for k, p in pr.items():
        if p.grad is None:
            assert pn[k].grad is None, k
            continue
        meas['d_' + k] = _close_grad(pn[k].grad, p.grad, 'd ' + k)
        checked += 1
    assert checked >= 29
    for k, b in ref.named_buffers():
        if k.endswith(('running_mean', 'running_var')) and ('gcn' in k):
            _close(dict(net.named_buffers())[k], b, 1e-5, k)
        if k.endswith('num_batches_tracked') and ('gcn' in k):
            assert int(dict(net.named_buffers())[k]) == int(b) == 1
    record('gcn_tail_fused_vs_torch_B%d' % B, {k: float(v) for k, v in meas.items() if not k.startswith('d_') or k in
                                                ('d_rot_feats', 'd_edge_importance', 'd_refine_gcn.gc.1.weight', 'd_r2p_gcn.act.0.0.weight')})


def test_fused_tail_with_missing_output_gradients_and_twice_in_a_row():
    """A loss that touches only smpl_pose (the other three outputs get no gradient: NULL pointers), run twice: the barrier state is reused."""
    _cfg()
    from danet_densepose2smpl_amd import nn as dnn
    dnn.ONEPASS_STREAM = None
    ref = _net(5)
    net = copy.deepcopy(ref)
    x = (torch.randn(8, 24, 128).abs() * 0.5).cuda()
    wgt = torch.randn(8, 216).cuda()
    for _ in range(2):
        res = []
        for m, fn in ((ref, _tail_torch), (net, _tail)):
            m.zero_grad(set_to_none=True)
            xt = x.clone().requires_grad_(True)
            (fn(m, xt)[3] * wgt).sum().backward()       # (not the square: a rotation's squared norm is the constant 3)
            res.append((xt.grad, m.edge_importance.grad.clone(), m.refine_gcn.gc[0].weight.grad.clone()))
        for a, b, w in zip(res[1], res[0], ('d rot_feats', 'd edge', 'd W1')):
            _close_grad(a, b, w)
    assert net.pose_regressors[0][1].weight.grad.abs().max() == 0          # (its output had no gradient: a zero gradient, fully written)
    torch.cuda.synchronize()
    assert not dnn.onepass_error()


def test_fused_tail_is_not_taken_outside_its_configuration():
    _cfg()
    from danet_densepose2smpl_amd import gcn_tail
    net = _net(3)
    assert gcn_tail.applicable(net, torch.zeros(4, 24, 128, device='cuda'))
    assert not gcn_tail.applicable(net, torch.zeros(40, 24, 128, device='cuda'))            # more rows than a workgroup's LDS tile
    net.eval()
    assert not gcn_tail.applicable(net, torch.zeros(4, 24, 128, device='cuda'))             # running statistics: the torch operations
    net.train()
    with torch.no_grad():
        assert not gcn_tail.applicable(net, torch.zeros(4, 24, 128, device='cuda'))


def test_a_barrier_that_gave_up_is_reported_outside_a_trainer():
    """nn.onepass_watch: with no Trainer to read the barrier's error word, the next barrier launch after a time-out raises."""
    _cfg()
    from danet_densepose2smpl_amd import nn as dnn
    dnn.ONEPASS_STREAM = None
    net = _net(2)
    x = (torch.randn(4, 24, 128).abs() * 0.5).cuda()
    _tail(net, x.clone().requires_grad_(True))
    torch.cuda.synchronize()
    bar = dnn._onepass_state(x.device)
    bar[2] = 0x301ec                                    # as a timed-out one-pass launch leaves it
    try:
        with pytest.raises(RuntimeError, match='0x301ec'):
            for _ in range(3):                            # (the look is one launch behind)
                _tail(net, x.clone().requires_grad_(True))
                torch.cuda.synchronize()
    finally:
        bar.zero_()
        dnn._WATCH.clear()
    _tail(net, x.clone().requires_grad_(True))
    torch.cuda.synchronize()
    assert not dnn.onepass_error()
