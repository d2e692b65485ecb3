"""GPU parity of the HIP model path against golden vectors produced by the reference itself
(tests/golden/make_golden.py, formula parameters) and against the plain-torch fp32 oracle.

Tolerances.  Convolutions run in bf16 on MFMA with fp32 accumulation and activations are stored in
bf16 (the reference is fp32).  One conv+BN layer agrees with fp32 to ~3e-3 (tests/test_gpu_conv.py,
test_gpu_norm.py).  Through the ~90 sequential conv/BN layers of HRNet-W48 with RANDOM weights the
bf16 rounding noise is amplified (random deep ReLU nets are chaotic: measured 0.3% after the stem,
2% after layer1, 5% after stage2, 8% after stage3, 18% at the heads), so whole-network outputs are
compared with a loose bound (relative RMS error < 0.35, cosine > 0.93) and STRUCTURAL exactness is
pinned separately: every stage of the product is run on the fp32 oracle's own intermediate
activations (teacher forcing) and must agree to 4e-2.  The geometric paths (SMPL, raster) have
their own 1e-4 / bit-exact tests."""
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


@pytest.mark.parametrize('name,cls', [('g6_hrnet', 'hrnet'), ('g6_poseresnet', 'resnet')])
def test_backbone_vs_reference_golden(name, cls):
    _cfg(**{'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    from danet_densepose2smpl_amd import hrnet, resnet
    g = golden(name)
    net = (hrnet.PoseHighResolutionNet if cls == 'hrnet' else resnet.PoseResNet)(part_out_dim=7)
    formula_params(net)
    if cls == 'resnet':
        damp_residual_branches(net)               # as in make_golden.g6_backbones
    net = net.cuda().train()
    img = torch.from_numpy(g['img']).cuda().requires_grad_(True)
    out = net(img)
    # bf16 convs vs the fp32 reference through a whole random-weight network: the bounds are what the
    # fp32-vs-bf16-autocast comparison of the SAME torch network gives on CPU (x ~2 margin)
    rms_max, cos_min = (0.35, 0.93) if cls == 'hrnet' else (0.2, 0.975)
    for k in KEYS:
        o = out[k] if (k != 'xd' or cls == 'hrnet') else out[k][:, ::4]
        rms, cos = _rms_cos(o, g[k])
        assert rms < rms_max and cos > cos_min, (k, rms, cos)
    loss = sum((out[k].float() * torch.cos(torch.arange(out[k].numel(), dtype=torch.float32, device='cuda').view_as(out[k]) * 0.37)).sum() for k in KEYS[:5])
    loss.backward()
    assert torch.isfinite(img.grad).all()        # deep-net input gradients are chaotic: per-op backward tests pin them
    gw = {k: p.grad for k, p in net.named_parameters()}
    for k in g.files:
        if k.startswith('grad__final_pred'):                     # head gradients see only the head
            nm = k[len('grad__'):].replace('__', '.')
            assert _rms_cos(gw[nm], g[k])[1] > 0.9, nm
    assert _rel(net.bn1.running_mean, g['bn1_running_mean']) < 2e-2


def test_hrnet_vs_fp32_oracle_at_working_size():
    """Same weights through the plain-torch fp32 oracle (CPU) and the HIP path, 128x128, B=4."""
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32})
    from danet_densepose2smpl_amd import hrnet
    from oracle import torch_ref
    ref = torch_ref.HRNet(part_out_dim=7)
    formula_params(ref)
    net = hrnet.PoseHighResolutionNet(part_out_dim=7)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    ref.train()
    img = torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        o_ref = ref(img)
        o = net(img.cuda())
    for k in KEYS:
        rms, cos = _rms_cos(o[k], o_ref[k].numpy())
        assert rms < 0.35 and cos > 0.93, (k, rms, cos)


def test_poseresnet_stages_teacher_forced_vs_fp32_oracle():
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32})
    from danet_densepose2smpl_amd import resnet
    from oracle import torch_ref
    ref = torch_ref.PoseResNet(part_out_dim=7)
    formula_params(ref)
    net = resnet.PoseResNet(part_out_dim=7)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    ref.train()
    names = ['conv1', 'layer1', 'layer2', 'layer3', 'layer4', 'deconv_layers.0', 'deconv_layers.3', 'deconv_layers.6',
             'final_pred.predict_v', 'final_pred.predict_hm']
    cap = {}
    rmods, pmods = dict(ref.named_modules()), dict(net.named_modules())
    hs = [rmods[n].register_forward_hook((lambda n: lambda m, i, o: cap.__setitem__(n, (i[0], o)))(n)) for n in names]
    with torch.no_grad():
        ref(torch.randn(4, 3, 128, 128, generator=torch.Generator().manual_seed(3)))
    for h in hs:
        h.remove()
    worst = {}
    with torch.no_grad():
        for n in names:
            xin, yref = cap[n]
            worst[n] = _rms_cos(pmods[n](xin.cuda()), yref.numpy())[0]
    assert max(worst.values()) < 4e-2, worst


@pytest.mark.parametrize('size,B', [(128, 4), (256, 2)])
def test_hrnet_stages_teacher_forced_vs_fp32_oracle(size, B):
    """Every stage of the HIP HRNet, fed with the fp32 oracle's own input to that stage, must
    reproduce the oracle's output of that stage (structure / wiring check without chaos).  (256, 2) is the benched resolution:
    the oracle is pinned to the reference there by golden g16 (tests/test_oracle_nets.py)."""
    _cfg(**{'DANET.INIMG_SIZE': size, 'DANET.HEATMAP_SIZE': size // 4})
    from danet_densepose2smpl_amd import hrnet
    from oracle import torch_ref
    ref = torch_ref.HRNet(part_out_dim=7)
    formula_params(ref)
    net = hrnet.PoseHighResolutionNet(part_out_dim=7)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    ref.train()
    names = ['layer1', 'transition1.0', 'transition1.1', 'stage2.0', 'transition2.2', 'stage3.0', 'stage3.1', 'stage3.2',
             'stage3.3', 'transition3.3', 'stage4.0', 'stage4.1', 'stage4.2', 'final_pred.predict_hm', 'final_pred.predict_u',
             'final_pred.predict_ann_index', 'stage2.0.fuse_layers.1.0', 'stage4.0.fuse_layers.3.0']
    cap = {}
    rmods, pmods = dict(ref.named_modules()), dict(net.named_modules())

    def mk(n):
        def f(m, i, o):
            cap[n] = (i[0], o)
        return f
    hs = [rmods[n].register_forward_hook(mk(n)) for n in names]
    img = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref(img)
    for h in hs:
        h.remove()
    to_dev = lambda t: [x.cuda() for x in t] if isinstance(t, (list, tuple)) else t.cuda()
    worst = {}
    with torch.no_grad():
        for n in names:
            xin, yref = cap[n]
            y = pmods[n](to_dev(xin))
            ys, yr = (y, yref) if isinstance(y, (list, tuple)) else ([y], [yref])
            assert len(ys) == len(yr)
            worst[n] = max(_rms_cos(a, b.numpy())[0] for a, b in zip(ys, yr))
    assert max(worst.values()) < 4e-2, worst


@pytest.mark.parametrize('align', [0, 1])
def test_iuv_estimator_vs_reference_golden(align):
    _cfg(**{'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16, 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.,
            'DANET.PARTDROP_RATE': 0., 'DANET.ALIGN_CORNERS': bool(align)})
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator
    g = golden('g7_estimator_align%d' % align)
    est = IUV_Estimator(pretrained=False)
    formula_params(est, skip=('learned_ratio', 'learned_offset', '_'))
    with torch.no_grad():
        est.learned_ratio.copy_(torch.from_numpy(g['learned_ratio']))
        est.learned_offset.copy_(torch.from_numpy(g['learned_offset']))
    est = est.cuda().train()
    t = lambda k: torch.from_numpy(g[k]).cuda()
    from danet_densepose2smpl_amd import iuv_estimator
    for fused in (True, False):              # the fused HIP losses and the tensor-op formulation, both against the reference
        iuv_estimator.FUSED_PART_LOSSES = fused
        try:
            rd = est(t('img'), t('iuv_gt'), t('kps'), has_iuv=torch.ones(2, device='cuda'))
        finally:
            iuv_estimator.FUSED_PART_LOSSES = True
        for a, k in zip(rd['uvia_pred'], ('u', 'v', 'index', 'ann')):
            assert _rms_cos(a, g[k])[0] < 0.35, k
        assert np.abs(rd['stn_kps_pred'].cpu().numpy() - g['stn_kps_pred']).max() < 0.1
        assert _rms_cos(rd['part_iuv_pred'], g['part_iuv_pred'])[0] < 0.45
        if not fused:
            # the GT partial maps depend on the predicted centres only through theta: compare loosely
            # (the fused path never materialises them)
            assert np.abs(rd['part_iuv_gt'].cpu().numpy() - g['part_iuv_gt']).mean() < 3e-2
        for k in g.files:
            if k.startswith('loss__'):
                ours = float(rd['losses'][k[6:]].detach().sum())
                ref = float(g[k].sum())
                assert abs(ours - ref) <= 0.15 * abs(ref) + 1e-2, (fused, k, ours, ref)


def test_decomposed_predictor_vs_reference_golden():
    _cfg(**{'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    from danet_densepose2smpl_amd.smpl_regressor import DecomposedPredictor
    g = golden('g9_predictor')
    pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
    net = DecomposedPredictor(None, (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6), pretrained=False)
    formula_params(net, skip=('mean_', 'I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A'))
    net = net.cuda()
    iuv = formula_input('g9.iuv', (4, 75, 64, 64)).cuda()
    part = formula_input('g9.part', (4, 24, 3, 7, 64, 64)).cuda()
    net.train()
    rd = net(iuv, part)
    assert _rel(rd['para'][:, :13], g['para_train'][:, :13]) < 5e-2
    assert np.abs(rd['para'][:, 13:].detach().cpu().numpy() - g['para_train'][:, 13:]).max() < 0.1      # rotation matrices
    assert _rel(rd['joint_position'][0], g['jp0']) < 8e-2 and _rel(rd['joint_position'][1], g['jp1']) < 8e-2
    assert np.abs(rd['joint_rotation'][0].detach().cpu().numpy() - g['jr0']).max() < 0.1
    net.eval()
    with torch.no_grad():
        pe = net(iuv, part)['para']
    assert np.abs(pe.cpu().numpy() - g['para_eval']).max() < 5e-2 * max(1.0, np.abs(g['para_eval']).max())


def test_danet_train_step_and_inference():
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    torch.manual_seed(0)
    tr = Trainer(default_options(2), device=torch.device('cuda'), distributed=False)
    batch = synthetic_in_dict(tr.model, 2, torch.device('cuda'), seed=1)
    out, losses = tr.train_step(batch)
    expected = {'loss_U', 'loss_V', 'loss_IndexUV', 'loss_segAnn', 'loss_roi', 'loss_pU', 'loss_pV', 'loss_pIndexUV',
                'joint_rotation0', 'joint_position0', 'joint_position1', 'keypoints_2d', 'keypoints_3d', 'smpl_pose',
                'smpl_betas', 'smpl_verts', 'cam'}
    assert set(losses) == expected                       # SURVEY.md A.1 loss keys (has_dp absent -> no *dp keys)
    assert all(torch.isfinite(v).all() for v in losses.values())
    unused = [n for n, p in tr.model.named_parameters() if p.grad is None]
    assert all(('rot2pos' in n or 'pos2rot' in n) for n in unused), unused[:5]     # SURVEY.md 7 (never used by 'gcn')
    assert out['prediction']['vertices'].shape == (2, 6890, 3)
    tr.model.eval()
    pred = tr.model.infer_net(batch['img'])
    assert pred['para'].shape == (2, 229)
    with pytest.raises(ValueError):
        tr.model.train()
        tr.model.infer_net(batch['img'])


@pytest.mark.parametrize('B', [2, 16])
def test_danet_resnet50_inference_config2(B):
    """BASELINE.json configs[1]: DaNet inference, ResNet-50 backbone, 256x256 (B = 16 is the configuration's batch)."""
    _cfg(**{'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64, 'DANET.IUV_REGRESSOR': 'resnet'})
    from danet_densepose2smpl_amd.danet import DaNet
    from danet_densepose2smpl_amd.trainer import default_options
    torch.manual_seed(0)
    model = DaNet(default_options(B), None, pretrained=False).cuda().eval()
    img = torch.randn(B, 3, 256, 256, device='cuda')
    pred = model.infer_net(img)
    para = pred['para']
    assert para.shape == (B, 229) and torch.isfinite(para).all()
    rot = para[:, 13:].reshape(-1, 3, 3)
    eye = torch.eye(3, device='cuda').expand_as(rot)
    assert (torch.bmm(rot, rot.transpose(1, 2)) - eye).abs().max() < 1e-4      # rot6d -> valid rotations
    out = model.iuv2smpl.smpl(betas=para[:, 3:13], body_pose=para[:, 13:].reshape(B, 24, 3, 3)[:, 1:],
                              global_orient=para[:, 13:].reshape(B, 24, 3, 3)[:, :1], pose2rot=False)
    iuv = model.iuv_renderer.verts2uvimg(out.vertices, para[:, :3])
    assert iuv.shape == (B, 3, 64, 64)
    part = torch.round(iuv[:, 0] * 24)
    assert (part == iuv[:, 0] * 24).all() and part.min() >= 0 and part.max() <= 24        # integer-valued part-index plane


def test_dp_point_losses_inside_the_estimator_and_train_step():
    """DensePose point supervision wired into IUV_Estimator.forward (GPU tensors through the same
    torch code that tests/test_host_logic.py pins against the reference) and into the train step."""
    _cfg(**{'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16, 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.,
            'DANET.PARTDROP_RATE': 0.})
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    g = golden('g7_estimator_align1')
    gd = golden('g11_dp_losses_align1')
    est = IUV_Estimator(pretrained=False)
    formula_params(est, skip=('learned_ratio', 'learned_offset', '_'))
    est = est.cuda().train()
    t = lambda k: torch.from_numpy(g[k]).cuda()
    dp = {k[4:]: torch.from_numpy(gd[k])[:2].cuda() for k in gd.files if k.startswith('dp__')}
    has_dp = torch.tensor([1., 0.], device='cuda')
    rd = est(t('img'), t('iuv_gt'), t('kps'), uvia_dp_gt=dp, has_iuv=torch.ones(2, device='cuda'), has_dp=has_dp)
    u, v, i, a = rd['uvia_pred']
    ref = IUV_Estimator.dp_uvia_losses(u, v, i, a, dp, has_dp, True)
    for k, r in zip(('loss_Udp', 'loss_Vdp', 'loss_IndexUVdp', 'loss_segAnndp'), ref):
        assert torch.isfinite(rd['losses'][k]).all() and float(rd['losses'][k]) > 0
        assert abs(float(rd['losses'][k]) - float(r)) <= 1e-5 * abs(float(r))
    sum(v_.sum() for v_ in rd['losses'].values()).backward()
    assert est.iuv_est.final_pred.predict_ann_index.weight.grad.abs().sum() > 0
    # train step with the zero blobs of a non-COCO batch: four extra zero losses, like iuv_estimator.py:118-121
    tr = Trainer(default_options(2), device=torch.device('cuda'), distributed=False)
    batch = synthetic_in_dict(tr.model, 2, torch.device('cuda'), seed=1, with_dp=True)
    _, losses = tr.train_step(batch)
    assert len(losses) == 21 and all(float(losses[k].sum()) == 0.0 for k in ('loss_Udp', 'loss_Vdp', 'loss_IndexUVdp', 'loss_segAnndp'))


def test_prepare_batch_builds_a_trainable_in_dict():
    """the step prologue on the device.  Labels rendered from a known camera must give that camera
    back (least squares on exact projections), and the produced in_dict must train."""
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32})
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(2), device=dev, distributed=False)
    ref = synthetic_in_dict(tr.model, 2, dev, seed=3)
    raw = {'img': ref['img'], 'keypoints': ref['keypoints'], 'pose': ref['opt_pose'], 'betas': ref['opt_betas'],
           'pose_3d': ref['pose_3d'], 'has_smpl': torch.ones(2), 'has_pose_3d': torch.ones(2), 'has_dp': torch.zeros(2)}
    batch = tr.prepare_batch(raw)
    assert set(ref) <= set(batch)
    # synthetic_in_dict projected the label joints with target_cam: the least-squares camera must reproduce it
    assert (batch['target_cam'] - ref['target_cam']).abs().max() < 2e-3
    assert (batch['target_smpl_kps'] - ref['target_smpl_kps']).abs().max() < 2e-3
    assert torch.equal(batch['target_verts'], ref['target_verts'])
    _, losses = tr.train_step(batch)
    assert all(torch.isfinite(v).all() for v in losses.values()) and len(losses) == 17
