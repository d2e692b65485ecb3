"""Whole-model PATH-EQUIVALENCE tests (eager vs hipGraph replay, lockstep vs per-branch execution, one autograd call vs the
segmented backward, one rank vs the data-parallel path).  They live in a file that sorts LAST so that `pytest -x` reaches
the per-op parity tests (test_gpu_smpl / _raster / _norm / _optim / _parts ...) before any of them.

Since round 5 every per-channel statistic and loss sum of the step is accumulated order-independently (double-precision
atomics of fp32 partial sums: csrc/conv_common.h), so two executions of the same step agree bit for bit in the PRODUCTION
BatchNorm configuration -- no test in this file has a noise precondition any more."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, GOLDEN
sys.path.insert(0, GOLDEN)
from make_golden import formula_params, formula_input, damp_residual_branches    # noqa: E402

pytestmark = pytest.mark.gpu

KEYS = ['predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm', 'xd']


def _rel(a, ref):
    ref = np.asarray(ref, np.float32)
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-6))


def _rms_cos(a, ref):
    ref = torch.as_tensor(np.asarray(ref, np.float32)).flatten().double()
    a = (a.detach().float().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).flatten().double()
    rms = float((a - ref).pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-12))
    cos = float((a * ref).sum() / (a.norm() * ref.norm() + 1e-12))
    return rms, cos


def _cfg(**kw):
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    reset_cfg()
    cfg_from_dict(kw)



# measured on the bench configuration (round 2); the bounds leave a little room for deliberate changes
# (measured: bn_stats 337 fused / 6 own, one-pass BatchNorm backward 3xx of 343, residual_grad 110 fused / 4 added)
FUSION_MIN = {'bn_stats_fused': 330, 'bn_bwd_onepass': 250, 'residual_grad_fused': 105}
FUSION_MAX = {'residual_grad_added': 8, 'bn_stats_own': 12}


def test_full_size_graphed_step_properties():
    """The bench configuration itself (B = 32, 256x256, hipGraph replay): the losses of a replay are finite and equal to
    the eager step's on the same batch and weights (learning rate ~0) within the bf16 noise of the atomics' summation
    order; every parameter that takes part in the step has a finite gradient living in the flat gradient store; the
    rendered ground-truth part plane is integer-valued; a second replay on the same batch reproduces the first."""
    _cfg(**{'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64, 'DANET.PARTDROP_RATE': 0.,
            'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(32), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 32, dev, seed=3)
    runs = []
    for _ in range(3):
        _, le = tr.train_step(batch)
        runs.append({k: float(v.sum()) for k, v in le.items()})
    e = runs[0]
    # run-to-run spread of EAGER steps on this random-weight net (atomics' summation order amplified through the soft-argmax / STN
    # crop chain: tools/noise_probe.py measures 2-3.5 % on loss_roi / cam over 6 runs, with or without the newer kernels): the 5 %
    # bound below is widened by twice the spread seen here, so a tail draw of that noise does not fail the suite
    spread = {k: max(r[k] for r in runs) - min(r[k] for r in runs) for k in e}
    tr.capture(batch, warmup=1)
    _, l1 = tr.train_step_graphed()
    g1 = {k: float(v.sum()) for k, v in l1.items()}
    _, l2 = tr.train_step_graphed()
    g2 = {k: float(v.sum()) for k, v in l2.items()}
    torch.cuda.synchronize()
    assert set(g1) == set(e) and len(g1) == 17
    for k in e:
        assert np.isfinite(g1[k]) and min(abs(g1[k] - r[k]) for r in runs) <= 5e-2 * abs(e[k]) + 2 * spread[k] + 1e-4, (k, e[k], g1[k], spread[k])
        assert abs(g2[k] - g1[k]) <= 5e-2 * abs(g1[k]) + 2 * spread[k] + 1e-4, (k, g1[k], g2[k], spread[k])
    flat = tr.store.flat
    assert torch.isfinite(flat).all() and float(flat.abs().max()) > 0
    for n, p in tr.model.named_parameters():
        if p.grad is not None:
            assert p.grad.data_ptr() == tr.store.grad_ptr(p), n
    uv = tr.model.iuv_renderer.verts2uvimg(batch['target_verts'], batch['target_cam'])
    assert uv.shape == (32, 3, 64, 64) and (torch.round(uv[:, 0] * 24) == uv[:, 0] * 24).all()
    # the attribute-carried fusions really are in the captured graph (conv.FUSION; they vanish silently if a view or
    # a copy gets between producer and consumer)
    fc = tr.fusion_counts
    print('fusion counts of the captured step:', fc)
    assert fc.get('bn_stats_fused', 0) >= FUSION_MIN['bn_stats_fused'], fc
    assert fc.get('bn_bwd_onepass', 0) >= FUSION_MIN['bn_bwd_onepass'], fc
    from danet_densepose2smpl_amd import nn as dnn
    assert not dnn.onepass_error()
    assert fc.get('residual_grad_fused', 0) >= FUSION_MIN['residual_grad_fused'], fc
    assert fc.get('residual_grad_added', 0) <= FUSION_MAX['residual_grad_added'], fc
    assert fc.get('bn_stats_own', 0) <= FUSION_MAX['bn_stats_own'], fc


def test_graphed_step_matches_eager_step():
    """hipGraph replay (accumulator arena, weight bank, weight packing inside the graph) computes what plain eager launches
    compute, in the production BatchNorm configuration: the learning rate is ~0, so every step sees the same weights, and
    since the step's sums are order-independent (csrc/conv_common.h) the replayed losses and weight gradients must EQUAL the
    eager step's bit for bit.  The first eager step computes every weight gradient inside its backward node (single-problem
    launches), the later ones through the deferred multi-problem launches: a different decomposition into partial sums,
    so those two agree to fp32 rounding (1e-4 relative L2 per tensor), not bit for bit."""
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0.,
            'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    dev = torch.device('cuda')
    torch.manual_seed(0)
    from danet_densepose2smpl_amd import trainer as trainer_mod
    NB = 2
    tr = Trainer(default_options(NB), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, NB, dev, seed=1)
    batch['pretrain_mode'] = True            # (danet.py: IUV estimator only, no SMPL regressor)
    trainer_mod.DEFER_WGRAD = False          # first step: every weight gradient computed inside its backward node ...
    try:
        _, losses = tr.train_step(batch)
    finally:
        trainer_mod.DEFER_WGRAD = True       # ... afterwards: queued and computed by the multi-problem launches
    eager = {k: v.detach().float().sum().clone() for k, v in losses.items()}
    named = [(n, p) for n, p in tr.model.named_parameters() if p.grad is not None and p.dim() == 4]
    g_eager = {n: p.grad.detach().clone() for n, p in named}
    _, losses = tr.train_step(batch)
    eager2 = {k: v.detach().float().sum().clone() for k, v in losses.items()}
    g_eager2 = {n: p.grad.detach().clone() for n, p in named}
    tr.capture(batch, warmup=2)
    tr.train_step_graphed()
    _, losses = tr.train_step_graphed()
    torch.cuda.synchronize()
    graphed = {k: v.detach().float().sum().clone() for k, v in losses.items()}
    for k in eager:
        assert torch.equal(eager[k], eager2[k]) and torch.equal(eager2[k], graphed[k]), (k, float(eager[k]), float(eager2[k]), float(graphed[k]))
    assert tr.bank is not None and tr.bank.jobs is not None and len(tr.bank.entries) > 200

    def rel(a, ref):
        return ((a - ref).norm() / (ref.norm() + 1e-12)).item()
    r_defer = {n: rel(g_eager2[n], g_eager[n]) for n, _ in named}
    assert max(r_defer.values()) < 1e-4, ('deferred vs immediate', sorted(r_defer.items(), key=lambda kv: -kv[1])[:3])
    for n, p in named:
        assert torch.equal(p.grad, g_eager2[n]), ('graph vs eager', n, rel(p.grad, g_eager2[n]))


@pytest.mark.parametrize('lds_tile', [False, True], ids=['gather_kernel', 'lds_tile_kernel'])
def test_lockstep_branches_match_per_branch_execution(lds_tile):
    """HRNet with the branches advanced in lockstep (multi-problem conv / multi-tensor BatchNorm launches) == branch-by-
    branch execution.  With the 3x3 layers pinned to the gather kernel both orders run the same arithmetic (outputs
    agree to bf16 rounding of a few reduction orders); with the LDS-tile kernel enabled a lockstep set may take a
    different kernel / K split than its members alone (the 2x2-pixel branch of this 64x64 test does not tile), which
    this ill-conditioned tiny net amplifies -- looser bound."""
    _cfg(**{'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    from danet_densepose2smpl_amd import hrnet, _lib
    torch.manual_seed(0)
    net = hrnet.PoseHighResolutionNet(part_out_dim=7)
    formula_params(net)
    net = net.cuda().train()
    img = torch.randn(4, 3, 64, 64, device='cuda')
    res = []
    prev = _lib.lib().danet_conv3x3_set(int(lds_tile), -1, -1, 0, -1)
    try:
        for lock in (False, True):
            hrnet.LOCKSTEP_BRANCHES = lock
            try:
                net.zero_grad(set_to_none=True)
                out = net(img)
                loss = sum((out[k].float() * torch.cos(torch.arange(out[k].numel(), device='cuda').view_as(out[k]) * 0.37)).sum() for k in KEYS[:5])
                loss.backward()
            finally:
                hrnet.LOCKSTEP_BRANCHES = True
            res.append(({k: out[k].detach().float().clone() for k in KEYS}, {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    finally:
        _lib.lib().danet_conv3x3_set(prev, -1, -1, 0, -1)
    (o0, g0), (o1, g1) = res
    for k in KEYS:
        assert _rms_cos(o1[k], o0[k].cpu().numpy())[0] < (6e-2 if lds_tile else 2e-2), k
    worst = max(((g1[n] - g0[n]).norm() / (g0[n].norm() + 1e-12)).item() for n in g0 if g0[n].dim() == 4)
    # chaotic at this size (2x2-pixel maps in the deepest branch, 16 samples per BatchNorm channel): any change of rounding
    # order moves single layers' gradients by tens of per cent (measured 0.40-0.51); exactness of the blocks is pinned by
    # tests/test_gpu_layers.py against the reference's own modules, this test guards the wiring of the lockstep path
    assert worst < 0.8, worst


def test_graphed_full_step_losses_match_eager():
    """The full model (regressor included) at batch 2 through hipGraph replay, production BatchNorm configuration: every loss
    term equals the eager step's bit for bit (learning rate ~0; the gradients of the full model are compared in the next test)."""
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0.,
            'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(2), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 2, dev, seed=1)
    _, l1 = tr.train_step(batch)
    e1 = {k: v.detach().float().sum().clone() for k, v in l1.items()}
    _, l2 = tr.train_step(batch)
    e2 = {k: v.detach().float().sum().clone() for k, v in l2.items()}
    tr.capture(batch, warmup=1)
    tr.train_step_graphed()
    _, lg = tr.train_step_graphed()
    torch.cuda.synchronize()
    g = {k: v.detach().float().sum().clone() for k, v in lg.items()}
    assert set(g) == set(e1) and len(g) == 17
    for k in e1:
        assert torch.isfinite(e1[k]) and torch.equal(e1[k], e2[k]) and torch.equal(e1[k], g[k]), (k, float(e1[k]), float(e2[k]), float(g[k]))


def test_graph_equals_eager_in_the_production_batchnorm_configuration():
    """Path equivalence in the configuration bench.py runs (replica accumulators + conv-epilogue statistics + the one-pass
    BatchNorm backward with its grid barrier), with NO noise precondition: every per-channel statistic,
    loss sum, bias gradient and atomically accumulated weight gradient of the step is an order-independent sum (doubles:
    csrc/conv_common.h), so three eager executions and four hipGraph replays of one step (same batch, same weights, learning
    rate ~0) must produce BIT-IDENTICAL losses and -- parameter by parameter -- bit-identical gradients.  A wrong-but-finite
    interaction of the three fusions, a race, or a replay that reads stale memory (round 5 found one: a memset NODE racing
    with the bias-gradient kernel turned one head bias gradient into inf after a few replays) cannot pass this.
    The only tolerated differences: gradient elements below 1e-6 of their tensor's largest (doubles are exact only while the
    partial sums of an element lie within 2^25 of each other: elements that are zero up to rounding may differ in that
    rounding), and the handful of GCN parameters whose gradient is a sum of huge cancelling terms computed by torch's own
    library kernels (BatchNorm1d / BLAS), which choose their algorithm differently under capture."""
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0.,
            'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    from danet_densepose2smpl_amd import nn as dnn, conv as dconv, _lib
    if _lib.lib().danet_bn_acc_bytes() != 8:
        pytest.skip('library built with DANET_BN_ACC32 (fp32 accumulators, the A-B timing form): sums depend on the arrival order')
    dev = torch.device('cuda')
    torch.manual_seed(0)
    assert dnn.ONEPASS and dconv.FUSE_BN_STATS
    tr = Trainer(default_options(8), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 8, dev, seed=1)
    tr.train_step(batch)

    def snap(losses):
        torch.cuda.synchronize()
        return ({k: v.detach().float().sum().clone() for k, v in losses.items()},
                {n: p.grad.detach().float().clone() for n, p in tr.model.named_parameters() if p.grad is not None})
    dconv.FUSION.clear()
    runs = [snap(tr.train_step(batch)[1])]
    assert dconv.FUSION.get('bn_bwd_onepass', 0) > 100 and dconv.FUSION.get('bn_stats_fused', 0) > 100, dict(dconv.FUSION)
    runs += [snap(tr.train_step(batch)[1]) for _ in range(2)]
    tr.capture(batch, warmup=1)
    assert tr.fusion_counts.get('bn_bwd_onepass', 0) > 100 and tr.fusion_counts.get('bn_stats_fused', 0) > 100
    from danet_densepose2smpl_amd import ops as _ops
    assert tr.fusion_counts.get('smpl_bwd_fused', 0) == (1 if _ops.SMPL_BWD_FUSED else 0)   # one launch only when opted in (DANET_LBS_BWD_FUSED)
    runs += [snap(tr.train_step_graphed()[1]) for _ in range(4)]
    assert not dnn.onepass_error()
    ref = runs[0]
    assert len(ref[0]) == 17 and len(ref[1]) > 1000
    for i, r in enumerate(runs[1:], 1):
        kind = 'eager' if i < 3 else 'replay'
        for k in ref[0]:
            assert torch.equal(r[0][k], ref[0][k]), ('loss differs', kind, i, k, float(r[0][k]), float(ref[0][k]))
        assert set(r[1]) == set(ref[1])
        for n in ref[1]:
            assert torch.isfinite(r[1][n]).all(), ('non-finite gradient', kind, i, n)
            if torch.equal(r[1][n], ref[1][n]):
                continue
            worst = float((r[1][n] - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30))
            library_side = 'gcn' in n and (n.endswith('.bias') or '.act.' in n)
            assert worst <= (4.0 if library_side else 1e-6), ('gradient differs', kind, i, n, worst)
    # every replay equals every other replay, the library-side parameters included (same tolerance for elements that are zero up
    # to rounding: measured 2e-27 against a largest element of 0.14 in one grouped 1x1 weight gradient)
    for r in runs[4:]:
        for n in ref[1]:
            d = float((r[1][n] - runs[3][1][n]).abs().max() / (runs[3][1][n].abs().max() + 1e-30))
            assert d <= 1e-6, ('replays differ', n, d)


def test_data_parallel_graph_path_single_rank():
    """The N > 1 execution path of bench.py on a 1-rank RCCL group: eager steps with the backward pass in segments
    (segments.py) and the bucketed all-reduces released between them, then the hipGraph capture with the all-reduces and
    Adam INSIDE the graph.  It must run and agree with the single-process trainer, whose backward pass is ONE autograd call (a
    sum over one rank is the identity, and cutting the graph changes no arithmetic)."""
    import torch.distributed as dist
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0.,
            'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    from danet_densepose2smpl_amd.trainer import reserve_comm_channels
    dev = torch.device('cuda', 0)
    port = 29500 + (os.getpid() % 2000)
    had = os.environ.get('NCCL_MAX_NCHANNELS')
    reserve_comm_channels()              # what a launch script does before the communicator exists (bench.py): the trainer only READS the limit
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
    try:
        res = {}
        for mode in ('single', 'ddp'):
            torch.manual_seed(0)
            tr = Trainer(default_options(2), device=dev, distributed=(mode == 'ddp'), lr=1e-30, bucket_mb=8.0)
            assert (tr.reducer is not None) == (mode == 'ddp') and len(tr.store.buckets) > 8
            assert tr.onepass_blocks == (2 * (256 - int(os.environ['NCCL_MAX_NCHANNELS'])) if mode == 'ddp' else 0)
            batch = synthetic_in_dict(tr.model, 2, dev, seed=1)
            batch['pretrain_mode'] = True
            tr.train_step(batch)
            _, l_eager = tr.train_step(batch)
            if mode == 'ddp':       # every bucket once, in bucket order; from the second step on (the first one learns which parameters
                nb = len(tr.store.buckets)      # are in use) the backward pass runs in segments and releases complete buckets between them
                assert tr.segmented and tr.store.issued == list(range(nb)), tr.store.issued
                assert 1 <= tr.store.issued_early < nb, (tr.store.issued_early, nb)
                eager_early = tr.store.issued_early
            tr.capture(batch, warmup=1)
            if mode == 'ddp':
                assert tr._reduce_in_graph, 'the RCCL all-reduces were not captured into the hipGraph'
                issued, early = tr.captured_collectives      # what the capture recorded: every bucket once, in order, the same ones
                assert issued == list(range(nb)) and early == eager_early, (issued, early)      # between the backward segments as in the eager step
            tr.train_step_graphed()
            _, l_graph = tr.train_step_graphed()
            torch.cuda.synchronize()
            named = [(n, p) for n, p in tr.model.named_parameters() if p.grad is not None and p.dim() == 4]
            for n, p in named:
                assert p.grad.data_ptr() == tr.store.grad_ptr(p), n      # zero-copy: .grad IS the bucket slot
            res[mode] = ({k: float(v.sum()) for k, v in l_eager.items()}, {k: float(v.sum()) for k, v in l_graph.items()},
                         {n: p.grad.detach().clone() for n, p in named})
        # production BatchNorm configuration (order-independent sums, csrc/conv_common.h): the forward pass of the two trainers is
        # the same arithmetic, so every loss -- eager and replayed -- is identical; the data-parallel trainer packs its one-pass
        # BatchNorm launches under a smaller co-residency budget and flushes its weight gradients bucket by bucket, i.e. other
        # partial sums of the same totals: gradients agree to fp32 rounding
        for k in res['single'][0]:
            for a, b in ((res['ddp'][0][k], res['single'][0][k]), (res['ddp'][1][k], res['single'][1][k]), (res['single'][1][k], res['single'][0][k])):
                assert a == b, (k, a, b)
        rel = {n: ((res['ddp'][2][n] - res['single'][2][n]).norm() / (res['single'][2][n].norm() + 1e-12)).item() for n in res['single'][2]}
        assert max(rel.values()) < 1e-3, sorted(rel.items(), key=lambda kv: -kv[1])[:3]
    finally:
        dist.destroy_process_group()
        if had is None:
            os.environ.pop('NCCL_MAX_NCHANNELS', None)


def test_segmented_backward_equals_one_autograd_call():
    """segments.py: cutting the autograd graph at the HRNet module boundaries and at the estimator -> regressor interface
    changes no arithmetic -- the same trainer, same batch, same weights, backward as ONE call and in segments: every loss is
    identical and so is every gradient (production BatchNorm configuration; the step's sums are order-independent)."""
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0.,
            'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    from danet_densepose2smpl_amd import segments
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(2), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 2, dev, seed=1)
    tr.train_step(batch)                                   # (BatchNorm running statistics, weight bank)
    runs = []
    for seg in (False, False, True):
        tr.segmented = seg
        levels = []
        orig = segments.backward

        def spy(losses, between=None, _o=orig, _l=levels):
            _l.append(segments.level())
            return _o(losses, between)
        segments.backward = spy
        try:
            _, l = tr.train_step(batch)
        finally:
            segments.backward = orig
        torch.cuda.synchronize()
        assert (levels == [9]) if seg else (levels == []), levels        # 8 HRNet modules + the regressor
        runs.append(({k: float(v.sum()) for k, v in l.items()},
                     {n: p.grad.detach().clone() for n, p in tr.model.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb), (ls, gs) = runs
    assert set(gs) == set(ga)
    for k in la:
        assert la[k] == lb[k] == ls[k], (k, la[k], lb[k], ls[k])
    # two one-call runs are bit-identical (order-independent sums); the segmented run launches the same kernels on the same
    # operands, gradient by gradient
    # (elements that are zero up to rounding -- 1e-27 beside 0.2 in one grouped 1x1 weight gradient of the regressor -- may differ in
    # that rounding: doubles add exactly only partial sums within 2^25 of each other)
    dmax = lambda x, y: float((x - y).abs().max() / (y.abs().max() + 1e-30))       # noqa: E731
    for n in ga:
        assert dmax(gb[n], ga[n]) <= 1e-6, ('two identical runs differ', n, dmax(gb[n], ga[n]))
        assert dmax(gs[n], ga[n]) <= 1e-6, ('segmented vs one call', n, dmax(gs[n], ga[n]))
    assert sum(1 for n in ga if not torch.equal(gs[n], ga[n])) <= 8


def test_bench_dry_launch_line_on_a_one_rank_group():
    """`bench.py --dry --force-ddp`: the launch line of a multi-GPU run (process group, trainer, capture with the in-graph
    all-reduces, one step, the JSON line with its `allreduce` record) end to end on one GPU."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + (os.getpid() + 777) % 2000), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-ddp', '--dry', '--batch', '4'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    ar = line['allreduce']
    assert line['dry'] and line['exec'] == 'hipgraph' and ar['mode'] == 'in-graph', line
    assert ar['buckets'] >= 10 and 1 <= ar['released_during_backward'] < ar['buckets'], ar
    assert ar['comm_channels_reserved'] >= 1 and ar['onepass_max_blocks'] == 2 * (256 - ar['comm_channels_reserved']), ar


def _two_proc_worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)          # (both processes share GPU 0: RCCL needs one GPU per rank)
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16, 'DANET.PARTDROP_RATE': 0.,
                   'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    dev = torch.device('cuda', 0)
    torch.manual_seed(10 + rank)                                         # different initial weights: the broadcast must fix them
    tr = Trainer(default_options(2), device=dev, distributed=True, lr=1e-4, bucket_mb=4.0)
    assert tr.store.world == world and len(tr.store.buckets) > 10
    batch = synthetic_in_dict(tr.model, 2, dev, seed=100 + rank)         # every rank its own shard
    p0 = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu()
    # one backward pass, local gradients kept aside, then the store's own reduction: reduced == sum of the local ones
    with tr._on_stream():
        tr._core(batch, reduce=False, with_optimizer=False)
        local = tr.store.flat.clone()
        for bi in range(len(tr.store.buckets)):
            tr.store.reduce_bucket(bi)
        tr.store.wait()
        summed = tr.store.flat.clone()
    torch.cuda.synchronize()
    # the full pipelined step (bucket-wise all-reduce between the weight-gradient launches, Adam with grad_scale = 1/2)
    tr.train_step(batch)
    torch.cuda.synchronize()
    p1 = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu()
    flat1 = tr.store.flat.cpu()
    # a one-pass BatchNorm barrier "times out" on rank 1 only (its sticky error word is set): rank 1's gradients are garbage
    # and, after the all-reduce, so are rank 0's sums -- the word travels with the last bucket and BOTH ranks' Adam kernels skip the step
    from danet_densepose2smpl_amd import nn as dnn
    if rank == 1:
        dnn.onepass_poison(dev).fill_(1)
    m0 = tr.optimizer.exp_avg.clone()
    tr.train_step(batch)
    torch.cuda.synchronize()
    p2 = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu()
    psum, moved = float(tr.store.poison), not torch.equal(tr.optimizer.exp_avg, m0)
    raised = False
    try:
        tr.check_onepass()           # decided from the all-reduced word: both ranks recover and raise together
    except RuntimeError as e:
        raised = 'on some rank' in str(e) or rank == 1
    onepass_after = dnn.ONEPASS
    tr.train_step(batch)            # the two-kernel BatchNorm backward from here on, on both ranks: a normal step again
    torch.cuda.synchronize()
    p3 = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu()
    torch.save({'p0': p0, 'local': local.cpu(), 'summed': summed.cpu(), 'flat': flat1, 'p1': p1, 'p2': p2, 'p3': p3, 'psum': psum,
                'moved': moved, 'raised': raised, 'onepass_after': onepass_after, 'psum_after': float(tr.store.poison)}, os.path.join(tmp, 'r%d.pt' % rank))
    dist.destroy_process_group()


def test_two_process_trainer_gradient_allreduce(tmp_path):
    """SURVEY 8e on the REAL trainer path: two processes (gloo backend, both on GPU 0), each with its own shard.  The
    bucketed reduction of the flat gradient store returns, on both ranks, exactly the sum of the two ranks' local
    gradients of the same backward pass (fp32; the average is that sum times grad_scale = 1/2, folded into Adam), for
    every parameter incl. the never-used ones (zeros); after the full pipelined step both ranks hold the same gradients
    and -- starting from the broadcast weights -- the same parameters."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_proc_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    assert torch.equal(r0['p0'], r1['p0'])                                       # broadcast from rank 0
    want = r0['local'] + r1['local']
    scale = want.abs().max().item()
    assert scale > 0
    for r in (r0, r1):
        assert (r['summed'] - want).abs().max().item() <= 1e-5 * scale          # fp32 sum of two terms
    assert torch.equal(r0['summed'], r1['summed'])
    assert (r0['local'] - r1['local']).abs().max().item() > 1e-3 * scale          # the shards really differ
    assert torch.equal(r0['flat'], r1['flat']) and r0['flat'].abs().max().item() > 0
    assert torch.equal(r0['p1'], r1['p1']) and not torch.equal(r0['p1'], r0['p0'])
    # the poisoned step (error word set on rank 1 only): skipped on BOTH ranks -- parameters and moments untouched, replicas identical --,
    # both ranks raise from check_onepass and switch the one-pass path off, and the next step is a normal one on both
    for r in (r0, r1):
        assert r['psum'] == 1.0 and not r['moved'] and torch.equal(r['p2'], r['p1']), (r['psum'], r['moved'])
        assert r['raised'] and r['onepass_after'] is False and r['psum_after'] == 0.0
    assert torch.equal(r0['p3'], r1['p3']) and not torch.equal(r0['p3'], r0['p2'])



def _two_proc_capture_worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)          # (both processes share GPU 0: RCCL needs one GPU per rank)
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16, 'DANET.PARTDROP_RATE': 0.,
                   'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
    dev = torch.device('cuda', 0)
    torch.manual_seed(20 + rank)
    tr = Trainer(default_options(2), device=dev, distributed=True, lr=1e-4, bucket_mb=4.0)
    batch = synthetic_in_dict(tr.model, 2, dev, seed=200 + rank)
    tr.train_step(batch)
    tr.train_step(batch)                       # (the second step releases buckets between the backward segments)
    early_eager = tr.store.issued_early
    torch.cuda.synchronize()
    tr.capture(batch, warmup=1)                # gloo collectives are no stream work: the trainer must capture forward + backward only ...
    in_graph = tr._reduce_in_graph             # ... and reduce + update after every replay
    p_before = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu()
    for _ in range(3):
        _, losses = tr.train_step_graphed()    # replay, then reduce_all() + optimizer.step() on the step's stream
    torch.cuda.synchronize()
    p_after = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu()
    finite = bool(all(torch.isfinite(v.float()).all() for v in losses.values()))
    torch.save({'in_graph': in_graph, 'early_eager': early_eager, 'p_before': p_before, 'p_after': p_after, 'finite': finite,
                'flat': tr.store.flat.cpu()}, os.path.join(tmp, 'c%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_process_capture_falls_back_to_reduce_after_replay(tmp_path):
    """Trainer.capture() in a two-process group whose collectives cannot be captured (gloo; both ranks on
    GPU 0).  The trainer must not try to put them into the graph (a capture that fails half-way left the step's stream in
    capture mode when round 5 tried: the decision is made from the backend, up front), capture forward + backward only, and run
    `reduce_all()` + the optimizer after every replay ("after the graph replay" in bench.py's line) -- the execution path a
    failed RCCL capture falls back to as well.  Three such steps leave both ranks with identical, finite, CHANGED parameters and
    identical gradient sums.  (With RCCL the in-graph capture succeeds: test_data_parallel_graph_path_single_rank.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_proc_capture_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = torch.load(tmp_path / 'c0.pt'), torch.load(tmp_path / 'c1.pt')
    for c in (c0, c1):
        assert c['in_graph'] is False and c['finite'] and c['early_eager'] >= 1
        assert torch.isfinite(c['p_after']).all() and not torch.equal(c['p_after'], c['p_before'])
    assert torch.equal(c0['p_before'], c1['p_before']) and torch.equal(c0['p_after'], c1['p_after'])
    assert torch.equal(c0['flat'], c1['flat'])


def test_regroup_parts_equals_the_reshape_and_its_gradient():
    """glue.regroup_parts (one launch each way) == x.reshape(NB, -1, H, W) on the logical NCHW tensor, bf16 and fp32."""
    from danet_densepose2smpl_amd.glue import regroup_parts
    for dt, (NB, J, C, H) in ((torch.bfloat16, (3, 24, 256, 4)), (torch.float32, (2, 24, 64, 2)), (torch.bfloat16, (2, 5, 8, 3))):
        x = torch.randn(NB * J, C, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = regroup_parts(x, NB)
        ref = x.detach().reshape(NB, J * C, H, H)
        assert y.shape == ref.shape and torch.equal(y, ref)
        assert y.permute(0, 2, 3, 1).is_contiguous()
        g = torch.randn_like(ref)
        y.backward(g)
        assert torch.equal(x.grad, g.reshape(NB * J, C, H, H))


def test_pack_image_equals_cast_and_pad():
    from danet_densepose2smpl_amd.glue import pack_image
    x = torch.randn(3, 3, 20, 12, device='cuda') * 3
    x[0, 0, 0, 0] = float('inf'); x[0, 1, 0, 1] = 1e-40
    y = pack_image(x)
    assert y.shape == (3, 8, 20, 12) and y.dtype == torch.bfloat16 and y.permute(0, 2, 3, 1).is_contiguous()
    assert torch.equal(y[:, :3], x.bfloat16()) and float(y[:, 3:].abs().max()) == 0


def test_side_stream_window_caps_the_barrier_budget_and_closes():
    """smpl_regressor: while body_net runs on its side stream (forward: fork .. join; backward: the mirror image, bracketed by two
    identity nodes) nn.onepass_budget() is one workgroup per compute unit; afterwards the window is closed again."""
    _cfg(**{'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    from danet_densepose2smpl_amd import nn as dnn, smpl_regressor as sr
    from danet_densepose2smpl_amd.smpl_regressor import DecomposedPredictor
    dnn.ONEPASS_STREAM = None
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert dnn.SIDE_LIVE == 0 and dnn.onepass_budget() == dnn.ONEPASS_MAX_BLOCKS
    dnn.SIDE_LIVE = 1
    try:
        assert dnn.onepass_budget() == cus
        keep, dnn.ONEPASS_MAX_BLOCKS = dnn.ONEPASS_MAX_BLOCKS, 64
        assert dnn.onepass_budget() == 64
        dnn.ONEPASS_MAX_BLOCKS = -1
        assert dnn.onepass_budget() == -1
        dnn.ONEPASS_MAX_BLOCKS = keep
    finally:
        dnn.SIDE_LIVE = 0
    pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
    net = DecomposedPredictor(None, (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6), pretrained=False).cuda().train()
    seen = []
    hook = net.limb_net[3].layer1[0].bn1.register_full_backward_hook(lambda m, gi, go: seen.append(dnn.SIDE_LIVE))
    iuv = torch.randn(2, 75, 64, 64, device='cuda', requires_grad=True)
    part = torch.randn(2, 24, 3, 7, 64, 64, device='cuda', requires_grad=True)
    rd = net(iuv, part)
    assert dnn.SIDE_LIVE == 0                                   # the forward window is closed at the join
    (rd['para'].float().sum() + sum(t.float().sum() for t in rd['joint_position'])).backward()
    hook.remove()
    assert seen == [1] if sr.BODY_STREAM else seen == [0]       # limb_net's backward ran inside the window
    assert dnn.SIDE_LIVE == 0 and iuv.grad is not None and part.grad is not None
    from danet_densepose2smpl_amd import conv
    conv.flush_wgrads()
    torch.cuda.synchronize()
    assert not dnn.onepass_error()
