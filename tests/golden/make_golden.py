"""Generates tests/golden/*.npz by IMPORTING the reference (only possible in the build
container, where /root/reference exists).  The outputs are data (inputs + expected outputs);
no reference source is stored.  Re-run:  python tests/golden/make_golden.py [names...]

Parameters of every reference nn.Module are overwritten with `formula_params` (a closed-form
function of the state-dict key and the element index), so the fixtures hold only inputs and
outputs: the test re-creates the same parameters on the build's module, whose state-dict
keys match the reference's.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def formula_tensor(key, shape, dtype=torch.float32):
    """Deterministic pseudo-random values in a range suited to the tensor's role."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = zlib.crc32(key.encode()) & 0x7fffffff
    idx = np.arange(n, dtype=np.float64)
    base = np.sin(idx * 12.9898 + (seed % 1000) * 0.37) * 43758.5453
    u = base - np.floor(base)                         # ~U[0,1)
    leaf = key.split('.')[-1]
    if leaf == 'running_var':
        v = 0.5 + u
    elif leaf == 'running_mean':
        v = (u - 0.5) * 0.2
    elif leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    elif leaf == 'weight' and len(shape) == 1:        # BN / norm gamma: small residual branches keep the
        v = 0.25 + 0.2 * u                             # random deep net well conditioned (bf16-vs-fp32 comparable)
    elif leaf == 'bias':
        v = (u - 0.5) * 0.1
    else:                                             # conv / linear / gcn weights
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if leaf == 'weight' and len(shape) == 2 and 'gc.' in key:
            fan_in = shape[0]
        v = (u - 0.5) * 2.0 * np.sqrt(3.0 / max(fan_in, 1))
        if key.endswith('predict_hm.1.weight'):
            v = v * 0.02       # keep the x10-temperature soft-argmax of the joint heat-maps smooth (the
                               # reference initialises these with std 0.001, hr_module.py:383-396)
    return torch.from_numpy(v.reshape(shape).astype(np.float32)).to(dtype)


def formula_input(name, shape, lo=0.0, hi=1.0):
    """Deterministic input tensor (no RNG, nothing to store): values in [lo, hi)."""
    n = int(np.prod(shape))
    seed = zlib.crc32(name.encode()) & 0x7fffffff
    idx = np.arange(n, dtype=np.float64)
    base = np.sin(idx * 78.233 + (seed % 997) * 0.11) * 12543.7453
    u = base - np.floor(base)
    return torch.from_numpy((lo + (hi - lo) * u).reshape(shape).astype(np.float32))


def formula_params(module, skip=()):
    sd = module.state_dict()
    new = {}
    for k, t in sd.items():
        if any(k.startswith(s) for s in skip):
            new[k] = t
        else:
            new[k] = formula_tensor(k, tuple(t.shape), t.dtype) if t.dtype.is_floating_point else t
    module.load_state_dict(new)


def damp_residual_branches(module, factor=0.3):
    """Scale the last BatchNorm gamma of every bottleneck (the usual zero-init-residual idea).  With the
    raw formula gammas a ResNet-50 amplifies bf16 rounding noise ~2x per stage (0.5 rel-RMS at the
    head); damped, fp32 vs bf16 agree to ~0.08, which makes the fixture a meaningful parity check."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k.endswith('bn3.weight'):
                p.mul_(factor)


def ref_env(overrides=None):
    """Appendix-B import shims: make the reference's torch-only modules importable."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)
    for pkg in ('models', 'models.danet'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, pkg.replace('.', '/'))]
            sys.modules[pkg] = m
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    torch.Tensor.cuda = lambda self, *a, **k: self
    comm = types.ModuleType('torch.cuda.comm')
    comm.broadcast = lambda t, devices=None: [t]
    torch.cuda.comm = comm
    sys.modules['torch.cuda.comm'] = comm
    if 'smplx' not in sys.modules:
        smplx = types.ModuleType('smplx')

        class _S(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()
        smplx.SMPL = _S
        bm = types.ModuleType('smplx.body_models')
        from collections import namedtuple
        bm.ModelOutput = namedtuple('ModelOutput', ['vertices', 'joints', 'full_pose', 'betas',
                                                    'global_orient', 'body_pose'])
        lbs = types.ModuleType('smplx.lbs')
        lbs.vertices2joints = lambda J, v: torch.einsum('bik,ji->bjk', [v, J])
        smplx.body_models, smplx.lbs = bm, lbs
        sys.modules.update({'smplx': smplx, 'smplx.body_models': bm, 'smplx.lbs': lbs})
    from models.core.config import cfg, _merge_a_into_b
    from utils.collections import AttrDict
    y = yaml.safe_load(open(os.path.join(REF, 'configs/danet_default.yaml')))
    _merge_a_into_b(AttrDict(y), cfg)
    cfg.DANET.REFINEMENT = AttrDict(cfg.DANET.REFINEMENT)
    cfg.MSRES_MODEL.EXTRA = AttrDict(cfg.MSRES_MODEL.EXTRA)
    for k, v in (overrides or {}).items():
        node = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------
def g1_geometry():
    ref_env()
    from utils import geometry as G
    g = torch.Generator().manual_seed(1234)
    theta = torch.randn(64, 3, generator=g) * 0.7
    theta[:4] *= 1e-6
    theta[4] = 0.0
    R = G.batch_rodrigues(theta)
    quat = torch.randn(16, 4, generator=g)
    Rq = G.quat_to_rotmat(quat)
    x6 = torch.randn(8, 144, generator=g, requires_grad=True)
    R6 = G.rot6d_to_rotmat(x6)
    w6 = torch.randn(R6.shape, generator=g)
    (R6 * w6).sum().backward()
    pts = torch.randn(4, 49, 3, generator=g)
    t = torch.randn(4, 3, generator=g) * 0.1 + torch.tensor([0., 0., 40.])
    rot = G.batch_rodrigues(torch.randn(4, 3, generator=g) * 0.2)
    cc = torch.randn(4, 2, generator=g)
    proj = G.perspective_projection(pts, rot, t, 5000., cc)
    save('g1_geometry', theta=theta, R=R, quat=quat, Rq=Rq, x6=x6.detach(), R6=R6, w6=w6, x6_grad=x6.grad,
         pts=pts, t=t, rot=rot, cc=cc, proj=proj)


def g2_iuvmap():
    ref_env()
    from utils.iuvmap import iuvmap_clean, iuv_img2map
    g = torch.Generator().manual_seed(1234)
    U, V, I = (torch.randn(2, 25, 16, 16, generator=g) for _ in range(3))
    A = torch.randn(2, 15, 16, 16, generator=g)
    cu, cv_, ci, ca = iuvmap_clean(U, V, I, A)
    part = torch.randint(0, 25, (2, 16, 16), generator=g)
    img = torch.stack([part.float() / 24., torch.rand(2, 16, 16, generator=g), torch.rand(2, 16, 16, generator=g)], 1)
    img[:, 1:] *= (part > 0).float().unsqueeze(1)
    # get_device() on CPU returns -1: harmless
    mu, mv, mi, ma = iuv_img2map(img)
    save('g2_iuvmap', U=U, V=V, I=I, A=A, cU=cu, cV=cv_, cI=ci, cA=ca, img=img, part=part, mU=mu, mV=mv, mI=mi, mA=ma)


def g3_graph():
    ref_env()
    from utils.graph import Graph, normalize_digraph, normalize_undigraph
    from utils.keypoints import softmax_integral_tensor
    g1 = Graph(layout='smpl', norm_type='none').A
    g2 = Graph(layout='smpl_2neigh', strategy='uniform', norm_type='none').A
    gen = torch.Generator().manual_seed(7)
    Ar = torch.rand(1, 24, 24, generator=gen)
    und = normalize_undigraph(Ar)
    dig0 = normalize_digraph(Ar[0].numpy().astype(np.float64), AD_mode=False)
    dig1 = normalize_digraph(Ar[0].numpy().astype(np.float64), AD_mode=True)
    hm = torch.randn(2, 24, 16, 16, generator=gen)
    si = softmax_integral_tensor(10 * hm, 24, 16, 16)
    save('g3_graph', A_smpl=g1, A_smpl2=g2, Ar=Ar, und=und, dig_da=dig0, dig_ad=dig1, hm=hm, softint=si)



def g4_gcn():
    """models/module/GCN.py GCN(128, 256, 128, 3, 24) -- output, input gradient and weight
    gradients, with the refinement adjacency of DecomposedPredictor (utils/graph.py:232-261)."""
    ref_env()
    from models.module.GCN import GCN
    net = GCN(128, 256, 128, 3, 24)
    formula_params(net)
    net.train()
    x = formula_input('g4.x', (4, 24, 128), -1.0, 1.0).requires_grad_(True)
    A = torch.from_numpy(np.load(os.path.join(HERE, 'g3_graph.npz'))['und'])
    y = net(x, A)
    w = formula_input('g4.w', tuple(y.shape), -1.0, 1.0)
    (y * w).sum().backward()
    grads = {('grad__' + k.replace('.', '__')): p.grad for k, p in net.named_parameters()}
    save('g4_gcn', A=A, y=y, x_grad=x.grad, **grads)


def g5_layers():
    """single layers / blocks of the reference's own modules (res_module.py:27-97,
    hr_module.py:15-179), train-mode BatchNorm, B = 2: output, input gradient, sentinel weight gradients, BatchNorm
    running statistics after the step."""
    ref_env({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    from models.module.res_module import BasicBlock, Bottleneck
    from models.module.hr_module import HighResolutionModule
    import torch.nn as nn
    out = {}

    def run(tag, mod, xs):
        formula_params(mod)
        mod.train()
        xs = [x.clone().requires_grad_(True) for x in xs]
        ys = mod(list(xs) if len(xs) > 1 else xs[0])
        ys = ys if isinstance(ys, (list, tuple)) else [ys]
        loss = 0
        for i, y in enumerate(ys):
            loss = loss + (y * formula_input('%s.w%d' % (tag, i), tuple(y.shape), -1.0, 1.0)).sum()
        loss.backward()
        for i, y in enumerate(ys):
            out['%s__y%d' % (tag, i)] = y.detach()
        for i, x in enumerate(xs):
            out['%s__dx%d' % (tag, i)] = x.grad
        for k, p in mod.named_parameters():
            if p.grad is not None and p.dim() in (1, 4):
                out['%s__grad__%s' % (tag, k.replace('.', '__'))] = p.grad
        for k, b in mod.named_buffers():
            if k.endswith('running_mean') or k.endswith('running_var'):
                out['%s__buf__%s' % (tag, k.replace('.', '__'))] = b.detach().clone()

    run('basic48', BasicBlock(48, 48), [formula_input('g5.basic48', (2, 48, 16, 16), -1.0, 1.0)])
    ds = nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256, momentum=0.1))
    run('bottle64', Bottleneck(64, 64, 1, ds), [formula_input('g5.bottle64', (2, 64, 16, 16), -1.0, 1.0)])
    ds24 = nn.Sequential(nn.Conv2d(256 * 24, 128 * 24, 1, 2, bias=False, groups=24), nn.BatchNorm2d(128 * 24, momentum=0.1))
    run('basic_g24', BasicBlock(256, 128, 2, ds24, groups=24), [formula_input('g5.basic_g24', (2, 256 * 24, 4, 4), -1.0, 1.0)])
    hrm = HighResolutionModule(2, BasicBlock, [4, 4], [48, 96], [48, 96], 'SUM', True)
    run('hrm2', hrm, [formula_input('g5.hrm2a', (2, 48, 16, 16), -1.0, 1.0), formula_input('g5.hrm2b', (2, 96, 8, 8), -1.0, 1.0)])
    for (ci, co, k, st, pd, gr, hw) in [(48, 96, 3, 2, 1, 1, 16), (384, 48, 1, 1, 0, 1, 4), (3, 64, 3, 2, 1, 1, 32), (64, 64, 7, 2, 3, 1, 16)]:
        run('conv_%d_%d_k%d_s%d' % (ci, co, k, st), nn.Conv2d(ci, co, k, st, pd, groups=gr, bias=False), [formula_input('g5.conv%d%d%d' % (ci, co, k), (2, ci, hw, hw), -1.0, 1.0)])
    # keep the fixture small: arrays beyond 16k elements are stored as a strided sample of the flattened tensor
    # (key suffix __s<stride>; the test takes the same sample)
    small = {}
    for k, v in out.items():
        v = v.detach()
        if v.numel() > 16384:
            stride = (v.numel() + 8191) // 8192
            small['%s__s%d' % (k, stride)] = v.flatten()[::stride].clone()
        else:
            small[k] = v
    save('g5_layers', **small)


ALL = {'g1': g1_geometry, 'g2': g2_iuvmap, 'g3': g3_graph, 'g4': g4_gcn, 'g5': g5_layers}

# ----------------------------------------------------------------------------------------------
# network-level fixtures (parameters = formula_params, so only inputs/outputs are stored)
def _img(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, S, S, generator=g)


def g6_backbones():
    ref_env({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    from models.module.hr_module import PoseHighResolutionNet
    from models.module.res_module import PoseResNet
    for name, cls in (('g6_hrnet', PoseHighResolutionNet), ('g6_poseresnet', PoseResNet)):
        torch.manual_seed(0)
        net = cls(part_out_dim=7)
        formula_params(net)
        if name == 'g6_poseresnet':
            damp_residual_branches(net)
        net.train()
        img = _img(2 if name == 'g6_hrnet' else 4, 64, 11).requires_grad_(True)      # ResNet-50 layer4 is 2x2: B=4 keeps BN sane
        out = net(img)
        keys = ['predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm', 'xd']
        loss = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=torch.float32).view_as(out[k]) * 0.37)).sum() for k in keys[:5])
        loss.backward()
        gw = {k.replace('.', '__'): p.grad for k, p in net.named_parameters()
              if k in ('conv1.weight', 'final_pred.predict_u.weight', 'final_pred.predict_hm.1.bias', 'bn1.weight')}
        save(name, img=img.detach(), img_grad=img.grad, bn1_running_mean=net.bn1.running_mean,
             **{k: (out[k].detach() if k != 'xd' or name == 'g6_hrnet' else out[k].detach()[:, ::4]) for k in keys},
             **{'grad__' + k: v for k, v in gw.items()})


def g7_estimator():
    for align in (False, True):
        cfg = ref_env({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16, 'DANET.STN_CENTER_JITTER': 0.,
                       'DANET.STN_SCALE_JITTER': 0., 'DANET.PARTDROP_RATE': 0.})
        import torch.nn.functional as F
        ag, gs = F.affine_grid, F.grid_sample
        if align:
            F.affine_grid = lambda theta, size, align_corners=None: ag(theta, size, align_corners=True)
            F.grid_sample = lambda x, grid, mode='bilinear', padding_mode='zeros', align_corners=None: gs(x, grid, mode, padding_mode, align_corners=True)
        try:
            from models.danet.iuv_estimator import IUV_Estimator
            torch.manual_seed(0)
            est = IUV_Estimator(pretrained=False)
            formula_params(est, skip=('learned_ratio', 'learned_offset'))
            est.train()
            g = torch.Generator().manual_seed(5)
            img = _img(2, 64, 12)
            part = torch.randint(0, 25, (2, 16, 16), generator=g)
            part[:, :3] = 0
            gt = torch.stack([part.float() / 24., torch.rand(2, 16, 16, generator=g), torch.rand(2, 16, 16, generator=g)], 1)
            gt[:, 1:] *= (part > 0).float().unsqueeze(1)
            kps = torch.cat([torch.rand(2, 24, 2, generator=g) * 1.6 - 0.8, torch.ones(2, 24, 1)], -1)
            kps[0, 3, 2] = 0.0
            has_iuv = torch.tensor([1, 1], dtype=torch.uint8).bool()
            rd = est(img, gt, kps, has_iuv=has_iuv)
            save('g7_estimator_align%d' % int(align), img=img, iuv_gt=gt, kps=kps,
                 learned_ratio=est.learned_ratio, learned_offset=est.learned_offset,
                 u=rd['uvia_pred'][0], v=rd['uvia_pred'][1], index=rd['uvia_pred'][2], ann=rd['uvia_pred'][3],
                 stn_kps_pred=rd['stn_kps_pred'], part_iuv_pred=rd['part_iuv_pred'], part_iuv_gt=rd['part_iuv_gt'],
                 **{'loss__' + k: v.detach().reshape(-1) for k, v in rd['losses'].items()})
        finally:
            F.affine_grid, F.grid_sample = ag, gs


def g9_predictor():
    ref_env({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    from models.danet.smpl_regressor import DecomposedPredictor
    torch.manual_seed(0)
    pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
    mean = (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6)
    net = DecomposedPredictor(None, mean, pretrained=False)
    formula_params(net, skip=('mean_', 'I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A'))
    iuv = formula_input('g9.iuv', (4, 75, 64, 64))
    part = formula_input('g9.part', (4, 24, 3, 7, 64, 64))
    net.train()
    rd = net(iuv, part)
    buf = {k: getattr(net, k) for k in ('I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A')}
    out = dict(para_train=rd['para'], jp0=rd['joint_position'][0], jp1=rd['joint_position'][1],
               jr0=rd['joint_rotation'][0], **buf)
    net.eval()
    with torch.no_grad():
        out['para_eval'] = net(iuv, part)['para']
    save('g9_predictor', **out)


def g10_losses():
    ref_env()
    from models.danet.smpl_regressor import SMPL_Regressor
    from models.danet.iuv_estimator import IUV_Estimator
    import torch.nn as nn
    g = torch.Generator().manual_seed(10)
    B = 6
    dummy = types.SimpleNamespace(criterion_shape=nn.L1Loss(), criterion_keypoints=nn.MSELoss(reduction='none'),
                                  criterion_regr=nn.MSELoss(), device=torch.device('cpu'))
    pred_rot, gt_rot = torch.randn(B, 24, 3, 3, generator=g), torch.randn(B, 216, generator=g)
    pb, gb = torch.randn(B, 10, generator=g), torch.randn(B, 10, generator=g)
    has_smpl = torch.tensor([1, 0, 1, 1, 0, 1])
    has_kp3d = torch.tensor([0, 1, 1, 0, 1, 1])
    lp, lb = SMPL_Regressor.smpl_losses(dummy, pred_rot, pb, gt_rot, gb, has_smpl)
    kp2 = torch.randn(B, 49, 2, generator=g)
    gk2 = torch.cat([torch.randn(B, 49, 2, generator=g), torch.rand(B, 49, 1, generator=g)], -1)
    l2d = SMPL_Regressor.keypoint_loss(dummy, kp2, gk2, 0.25, 1.0)
    pj = torch.randn(B, 49, 3, generator=g)
    g3 = torch.cat([torch.randn(B, 24, 3, generator=g), torch.rand(B, 24, 1, generator=g)], -1)
    l3d = SMPL_Regressor.keypoint_3d_loss(dummy, pj, g3, has_kp3d)
    pv, gv = torch.randn(B, 50, 3, generator=g), torch.randn(B, 50, 3, generator=g)
    lv = SMPL_Regressor.shape_loss(dummy, pv, gv, has_smpl)
    a, b = torch.randn(B, 24, 3, generator=g), torch.randn(B, 24, 3, generator=g)
    l1 = SMPL_Regressor.l1_losses(dummy, a, b, has_smpl)
    # body_uv_losses (global, 25 classes) with a partial has_iuv mask
    u, v, idx = (torch.randn(B, 25, 8, 8, generator=g) for _ in range(3))
    ann = torch.randn(B, 15, 8, 8, generator=g)
    part = torch.randint(0, 25, (B, 8, 8), generator=g)
    from utils.iuvmap import iuv_img2map
    gt = torch.stack([part.float() / 24., torch.rand(B, 8, 8, generator=g), torch.rand(B, 8, 8, generator=g)], 1)
    gt[:, 1:] *= (part > 0).float().unsqueeze(1)
    uvia = iuv_img2map(gt)
    has_iuv = torch.tensor([1, 1, 0, 1, 0, 1]).bool()
    est = types.SimpleNamespace()
    lU, lV, lI, lA = IUV_Estimator.body_uv_losses(est, u, v, idx, ann, uvia, has_iuv)
    save('g10_losses', pred_rot=pred_rot, gt_rot=gt_rot, pb=pb, gb=gb, has_smpl=has_smpl, has_kp3d=has_kp3d,
         loss_pose=lp, loss_betas=lb, kp2=kp2, gk2=gk2, loss_kp2d=l2d, pj=pj, g3=g3, loss_kp3d=l3d,
         pv=pv, gv=gv, loss_verts=lv, a=a, b=b, loss_l1=l1,
         u=u, v=v, idx=idx, ann=ann, iuv_gt=gt, has_iuv=has_iuv, loss_U=lU, loss_V=lV, loss_I=lI, loss_A=lA)


def g11_dp_losses():
    """DensePose point supervision (iuv_estimator.py:343-419) on the has_dp subset, as the reference's forward calls it
    (:106-117); gradients w.r.t. the FULL prediction batch (zero rows for samples without DensePose labels)."""
    import torch.nn.functional as F
    ag, gs = F.affine_grid, F.grid_sample
    for align in (False, True):
        ref_env({'DANET.HEATMAP_SIZE': 16})
        if align:
            F.grid_sample = lambda x, grid, mode='bilinear', padding_mode='zeros', align_corners=None: gs(x, grid, mode, padding_mode, align_corners=True)
        else:
            F.grid_sample = lambda x, grid, mode='bilinear', padding_mode='zeros', align_corners=None: gs(x, grid, mode, padding_mode, align_corners=False)
        try:
            from models.danet.iuv_estimator import IUV_Estimator
            g = torch.Generator().manual_seed(11)
            B, S = 4, 16
            u, v, idx = (torch.randn(B, 25, S, S, generator=g).requires_grad_(True) for _ in range(3))
            ann = torch.randn(B, 15, S, S, generator=g).requires_grad_(True)
            has_dp = torch.tensor([1, 0, 1, 1])
            I = torch.randint(0, 25, (B, 196), generator=g)
            I[:, 150:] = 0                                          # unused point slots are zero-filled in the dataset
            onehot = torch.nn.functional.one_hot(I, 25).permute(0, 2, 1).float()          # [B,25,196]
            wts = onehot * (I > 0).float().unsqueeze(1)
            dp = {'body_uv_X_points': torch.rand(B, 196, generator=g) * 15.0 + 0.3,
                  'body_uv_Y_points': torch.rand(B, 196, generator=g) * 15.0 + 0.3,
                  'body_uv_Ind_points': torch.zeros(B, 196), 'body_uv_I_points': I.float(),
                  'body_uv_U_points': (torch.rand(B, 25, 196, generator=g) * wts).reshape(B, 4900),
                  'body_uv_V_points': (torch.rand(B, 25, 196, generator=g) * wts).reshape(B, 4900),
                  'body_uv_point_weights': wts.reshape(B, 4900),
                  'body_uv_ann_labels': torch.randint(0, 15, (B, S * S), generator=g).to(torch.int32),
                  'body_uv_ann_weights': torch.ones(B, S * S)}
            on = has_dp == 1
            lU, lV, lI, lA = IUV_Estimator.dp_uvia_losses(None, u[on], v[on], idx[on], ann[on], **{k: t[on] for k, t in dp.items()})
            (lU * 1.0 + lV * 2.0 + lI * 3.0 + lA * 4.0).backward()
            save('g11_dp_losses_align%d' % int(align), u=u.detach(), v=v.detach(), idx=idx.detach(), ann=ann.detach(), has_dp=has_dp,
                 loss_Udp=lU.detach(), loss_Vdp=lV.detach(), loss_IndexUVdp=lI.detach(), loss_segAnndp=lA.detach(),
                 gu=u.grad, gv=v.grad, gidx=idx.grad, gann=ann.grad, **{'dp__' + k: t for k, t in dp.items()})
        finally:
            F.affine_grid, F.grid_sample = ag, gs


def g12_label_prologue():
    """The geometry of the reference's step prologue (train/trainer.py:170-210): camera translation by weighted least
    squares (utils/geometry.py:94-157), projected SMPL key-points, weak-perspective camera for the renderer."""
    ref_env()
    from utils.geometry import estimate_translation, perspective_projection
    g = torch.Generator().manual_seed(12)
    B, res, focal = 5, 224, 5000.
    joints = torch.randn(B, 49, 3, generator=g) * 0.3
    smpl_joints = torch.randn(B, 24, 3, generator=g) * 0.3
    t_true = torch.stack([torch.randn(B, generator=g) * 0.1, torch.randn(B, generator=g) * 0.1, 8 + 4 * torch.rand(B, generator=g)], -1)
    proj = perspective_projection(joints, rotation=torch.eye(3).unsqueeze(0).expand(B, -1, -1), translation=t_true, focal_length=focal,
                                  camera_center=torch.zeros(B, 2) + 0.5 * res)
    kp = torch.cat([proj / (0.5 * res) - 1 + 0.01 * torch.randn(B, 49, 2, generator=g), torch.rand(B, 49, 1, generator=g)], -1)
    kp[0, 30:35, 2] = 0.0
    kp_orig = kp.clone()
    kp_orig[:, :, :-1] = 0.5 * res * (kp_orig[:, :, :-1] + 1)
    cam_t = estimate_translation(joints, kp_orig, focal_length=focal, img_size=res)
    has_iuv = torch.tensor([1, 0, 1, 1, 0], dtype=torch.uint8)
    has_dp = torch.tensor([0, 1, 0, 0, 0])
    smpl_2dkps = torch.rand(B, 24, 3, generator=g)
    tk = torch.zeros(B, 24, 3)
    tk[:, :, :2] = perspective_projection(smpl_joints, rotation=torch.eye(3).unsqueeze(0).expand(B, -1, -1), translation=cam_t,
                                          focal_length=focal, camera_center=torch.zeros(B, 2) + 0.5 * res)
    tk[:, :, :2] = tk[:, :, :2] / (0.5 * res) - 1
    tk[has_iuv == 1, :, 2] = 1
    tk[has_dp == 1] = smpl_2dkps[has_dp == 1]
    cam = torch.zeros(B, 3)
    cam[:, 1:] = cam_t[:, :2]
    cam[:, 0] = (2. * focal / res) / cam_t[:, 2]
    save('g12_label_prologue', joints=joints, smpl_joints=smpl_joints, keypoints=kp, cam_t=cam_t, has_iuv=has_iuv, has_dp=has_dp,
         smpl_2dkps=smpl_2dkps, target_smpl_kps=tk, target_cam=cam)


def g13_eval_metrics():
    """MPJPE and Procrustes-aligned reconstruction error (eval.py:183-216, utils/pose_utils.py:10-75)."""
    ref_env()
    from utils.pose_utils import reconstruction_error, compute_similarity_transform_batch
    rng = np.random.default_rng(13)
    B, J = 6, 14
    gt = rng.normal(0, 0.3, (B, J, 3)).astype(np.float32)
    pred = np.zeros_like(gt)
    for i in range(B):                        # similarity-transformed + noisy copies (one reflected: det(R) < 0 branch)
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if i == 3:
            Q[:, 0] *= -np.sign(np.linalg.det(Q))
        pred[i] = (1.0 + 0.3 * rng.normal()) * gt[i] @ Q.T + rng.normal(0, 0.2, (1, 3)) + rng.normal(0, 0.02, (J, 3))
    pred = pred.astype(np.float32)
    mpjpe = np.sqrt(((pred - gt) ** 2).sum(-1)).mean(-1)
    save('g13_eval_metrics', pred=pred, gt=gt, mpjpe=mpjpe.astype(np.float32),
         aligned=compute_similarity_transform_batch(pred, gt).astype(np.float32),
         recon=reconstruction_error(pred, gt, reduction=None).astype(np.float32))


def g14_augment():
    """Label-side augmentation arithmetic (utils/imutils.py:11-153, datasets/base_dataset.py:158-187): the crop transform
    with its truncation to integers, keypoint / pose flips, 2D / 3D keypoint processing, Gaussian heat-map targets."""
    ref_env()
    import utils.imutils as im
    rng = np.random.default_rng(14)
    B = 12
    center = rng.uniform(80, 400, (B, 2))
    scale = rng.uniform(0.6, 2.5, B)
    rot = rng.uniform(-60, 60, B)
    rot[::3] = 0
    flip = (rng.uniform(size=B) < 0.5).astype(np.int64)
    T = np.stack([im.get_transform(center[b], scale[b], [224, 224], rot=rot[b]) for b in range(B)])
    pts = rng.uniform(1, 500, (B, 9, 2))
    fwd = np.stack([np.stack([im.transform(pts[b, n], center[b], scale[b], [224, 224], rot=rot[b]) for n in range(9)]) for b in range(B)])
    opts = rng.uniform(1, 224, (B, 9, 2))
    inv = np.stack([np.stack([im.transform(opts[b, n], center[b], scale[b], [224, 224], invert=1, rot=rot[b]) for n in range(9)]) for b in range(B)])
    out = dict(center=center, scale=scale, rot=rot, flip=flip, T=T, pts=pts, fwd=fwd.astype(np.float64), opts=opts, inv=inv.astype(np.float64))
    for N in (24, 49):
        kp = np.concatenate([rng.uniform(0, 500, (B, N, 2)), rng.uniform(0, 1, (B, N, 1))], -1)
        res = []
        for b in range(B):                               # base_dataset.py:158-171 with the reference's transform / flip_kp
            k = kp[b].copy()
            for i in range(N):
                k[i, 0:2] = im.transform(k[i, 0:2] + 1, center[b], scale[b], [224, 224], rot=rot[b])
            k[:, :-1] = 2. * k[:, :-1] / 224 - 1.
            if flip[b]:
                k = im.flip_kp(k)
            res.append(k.astype('float32'))
        out['kp%d' % N], out['j2d%d' % N] = kp, np.stack(res)
    S = np.concatenate([rng.normal(0, 0.5, (B, 24, 3)), rng.uniform(0, 1, (B, 24, 1))], -1)
    res = []
    for b in range(B):                                   # base_dataset.py:173-187
        s_ = S[b].copy()
        rm = np.eye(3)
        if not rot[b] == 0:
            rr = -rot[b] * np.pi / 180
            sn, cs = np.sin(rr), np.cos(rr)
            rm[0, :2] = [cs, -sn]
            rm[1, :2] = [sn, cs]
        s_[:, :-1] = np.einsum('ij,kj->ki', rm, s_[:, :-1])
        if flip[b]:
            s_ = im.flip_kp(s_)
        res.append(s_.astype('float32'))
    out['S'], out['j3d'] = S, np.stack(res)
    pose = rng.normal(0, 0.4, (B, 72))
    out['pose'], out['pose_flipped'] = pose, np.stack([im.flip_pose(pose[b].copy()) for b in range(B)])
    joints = np.concatenate([rng.uniform(-0.15, 1.15, (B, 17, 2)), np.ones((B, 17, 1))], -1).astype(np.float32)
    vis = (rng.uniform(size=(B, 17, 1)) < 0.8).astype(np.float32)
    hm, hw = [], []
    for b in range(B):
        t_, w_ = im.generate_heatmap(torch.from_numpy(joints[b]), 56, sigma=1 + b % 2, joints_vis=np.repeat(vis[b], 3, 1))
        hm.append(t_.numpy()); hw.append(np.asarray(w_))
    out['hm_joints'], out['hm_vis'], out['hm'], out['hm_w'] = joints, vis, np.stack(hm).astype(np.float16), np.stack(hw).astype(np.float32)
    save('g14_augment', **out)


ALL.update({'g14': g14_augment, 'g13': g13_eval_metrics, 'g12': g12_label_prologue, 'g6': g6_backbones, 'g7': g7_estimator, 'g9': g9_predictor, 'g10': g10_losses, 'g11': g11_dp_losses})


def g15_dp_producer():
    """DensePose-COCO dp_dict producer (utils/dp_utils.py:12-140) and the mirror-symmetry labels
    (utils/densepose_methods.py:31-59), run on SYNTHETIC symmetry tables and annotations.  The reference module needs three
    things this image lacks: the licensed UV_*.mat tables (written here as synthetic .mat files into a scratch directory the
    reference loads them from), pycocotools' mask decode and cv2.remap (both third-party; this repo's restatements
    dp_utils.rle_decode / remap_nearest are plugged in for them) -- so the fixture pins the reference's own arithmetic around
    those two primitives, not the primitives."""
    import tempfile
    from scipy.io import savemat
    ref_env()
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from danet_densepose2smpl_amd import dp_utils as ours
    rng = np.random.default_rng(15)
    u_tab, v_tab = ours.synthetic_symmetry_tables()      # closed-form lookup images: the test re-creates them
    obj = lambda t: np.array([[t[i] for i in range(24)]], dtype=object)
    scratch = tempfile.mkdtemp()
    os.makedirs(os.path.join(scratch, 'data/UV_data'))
    savemat(os.path.join(scratch, 'data/UV_data/UV_symmetry_transforms.mat'), {'U_transforms': obj(u_tab), 'V_transforms': obj(v_tab)})
    savemat(os.path.join(scratch, 'data/UV_data/UV_Processed.mat'),
            {'All_FaceIndices': np.zeros((4, 1)), 'All_Faces': np.ones((4, 3)), 'All_U_norm': np.zeros((6, 1)), 'All_V_norm': np.zeros((6, 1)),
             'All_vertices': np.zeros((1, 6))})
    cv2 = sys.modules['cv2']
    cv2.INTER_NEAREST, cv2.BORDER_CONSTANT = 0, 0
    cv2.remap = lambda img, mx, my, interpolation=None, borderMode=None, borderValue=None: ours.remap_nearest(img, mx, my)
    pm = types.ModuleType('pycocotools')
    pmm = types.ModuleType('pycocotools.mask')
    pmm.decode = ours.rle_decode
    pm.mask = pmm
    sys.modules.update({'pycocotools': pm, 'pycocotools.mask': pmm})
    os.chdir(scratch)
    import utils.dp_utils as ref_dp                     # (instantiates DensePoseMethods from the scratch directory's tables)
    out = {}
    yy, xx = np.mgrid[0:256, 0:256]
    for case in range(4):
        # 14 overlapping part blobs (some parts absent), up to 120 annotated points
        masks, polys = [], []
        for i in range(14):
            if (case + i) % 5 == 0:
                polys.append([])
                masks.append(np.zeros((256, 256), np.uint8))
                continue
            cx, cy, r = rng.uniform(40, 216), rng.uniform(40, 216), rng.uniform(12, 60)
            m = (((xx - cx) ** 2 + (yy - cy) ** 2) < r * r).astype(np.uint8)
            masks.append(m)
            polys.append(ours.rle_encode(m))
        n = int(rng.integers(20, 120))
        ann = {'bbox': [float(rng.uniform(20, 120)), float(rng.uniform(20, 120)), float(rng.uniform(90, 260)), float(rng.uniform(120, 300))],
               'dp_masks': polys, 'dp_I': rng.integers(1, 25, n).astype(np.float64).tolist(), 'dp_U': rng.uniform(0, 1, n).tolist(),
               'dp_V': rng.uniform(0, 1, n).tolist(), 'dp_x': rng.uniform(0, 255, n).tolist(), 'dp_y': rng.uniform(0, 255, n).tolist()}
        center = [ann['bbox'][0] + ann['bbox'][2] / 2 + float(rng.uniform(-15, 15)), ann['bbox'][1] + ann['bbox'][3] / 2 + float(rng.uniform(-15, 15))]
        scale = float(max(ann['bbox'][2], ann['bbox'][3]) / 200. * rng.uniform(0.9, 1.3))
        flipped = case % 2
        d = ref_dp.dp_annot_process(ann, 56, 224, center, scale, flipped)
        out['c%d_counts' % case] = np.array([p['counts'] if p else b'' for p in polys], dtype=object).astype('S')
        for k in ('bbox', 'dp_I', 'dp_U', 'dp_V', 'dp_x', 'dp_y'):
            out['c%d_%s' % (case, k)] = np.asarray(ann[k], dtype=np.float64)
        out['c%d_center' % case], out['c%d_scale' % case], out['c%d_flipped' % case] = np.asarray(center), np.asarray(scale), np.asarray(flipped)
        for k, v in d.items():
            out['c%d_out_%s' % (case, k)] = v
        if case == 0:                                      # the symmetry function on its own
            I, U, V = np.asarray(ann['dp_I']), np.asarray(ann['dp_U']), np.asarray(ann['dp_V'])
            lab = ours.get_densepose_mask(polys)
            res = ref_dp.DP.get_symmetric_densepose(I, U, V, np.asarray(ann['dp_x']), np.asarray(ann['dp_y']), lab)
            for name, r_ in zip(('I', 'U', 'V', 'x', 'y', 'mask'), res):
                out['sym_' + name] = np.asarray(r_)
            out['sym_in_mask'] = lab
    os.chdir(REF)
    save('g15_dp_producer', **out)


ALL['g15'] = g15_dp_producer


def _sub(t, step=4):
    """Every `step`-th pixel of a map plus its per-channel means: a 256^2 fixture stays small and still sees every pixel."""
    return t.detach()[..., ::step, ::step].contiguous(), t.detach().double().mean(dim=(-2, -1)).float()


def g16_hrnet256():
    """Round-5 review item 5: the reference's own PoseHighResolutionNet at the BENCHED resolution (256 x 256 -> 64 x 64 maps, B = 2,
    train-mode BatchNorm) -- every earlier reference fixture of the backbone is 64 x 64.  The expected values are the reference run
    in DOUBLE precision (module.double()); `floor__<key>` is how far the reference's own fp32 run is from that (max abs) -- the
    error floor of ANY fp32 implementation of this network with these parameters, which the fp32-mode tolerance is derived from."""
    ref_env({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    from models.module.hr_module import PoseHighResolutionNet
    torch.manual_seed(0)
    net = PoseHighResolutionNet(part_out_dim=7)
    formula_params(net)
    net.train()
    img = formula_input('g16.img', (2, 3, 256, 256), -2.0, 2.0)
    with torch.no_grad():
        out32 = net(img)
    rm32 = net.bn1.running_mean.clone()
    formula_params(net)                                                   # (the fp32 run moved the running statistics)
    net = net.double().train()
    with torch.no_grad():
        out = net(img.double())
    arrs = {}
    for k in ('predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm', 'xd'):
        arrs[k], arrs[k + '__mean'] = _sub(out[k].float())
        arrs['floor__' + k] = (out32[k].double() - out[k]).abs().max().float()
        arrs['scale__' + k] = out[k].abs().max().float()
        print(k, 'fp32 reference vs fp64 reference: max abs %.3g at scale %.3g' % (float(arrs['floor__' + k]), float(arrs['scale__' + k])))
    save('g16_hrnet256', bn1_running_mean=net.bn1.running_mean.float(), bn2_running_var=net.bn2.running_var.float(),
         floor__bn1_running_mean=(rm32.double() - net.bn1.running_mean).abs().max().float(), **arrs)


def g18_hrnet256_b32():
    """The BENCHED workload's backbone (32 images, 256 x 256, train-mode BatchNorm) through the reference in double precision, with the
    reference's own fp32 floor (round-5 review, missing 5: "every reference golden is 64^2 / B = 2 or one 256^2 predictor input").  Every
    16th pixel of the six maps + per-image-and-channel means over all pixels."""
    ref_env({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    from models.module.hr_module import PoseHighResolutionNet
    torch.manual_seed(0)
    net = PoseHighResolutionNet(part_out_dim=7)
    formula_params(net)
    net.train()
    img = formula_input('g18.img', (32, 3, 256, 256), -2.0, 2.0)
    with torch.no_grad():
        out32 = net(img)
    formula_params(net)
    net = net.double().train()
    with torch.no_grad():
        out = net(img.double())
    arrs = {}
    for k in ('predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm', 'xd'):
        arrs[k], arrs[k + '__mean'] = _sub(out[k].float(), 16)
        arrs['floor__' + k] = (out32[k].double() - out[k]).abs().max().float()
        arrs['scale__' + k] = out[k].abs().max().float()
        print(k, 'fp32 reference vs fp64 reference: max abs %.3g at scale %.3g' % (float(arrs['floor__' + k]), float(arrs['scale__' + k])))
    save('g18_hrnet256_b32', bn1_running_mean=net.bn1.running_mean.float(), **arrs)


def g19_inputs(B=32, S=64):
    """Closed-form inputs of g19 (nothing stored): image, ground-truth IUV image (4 x 4 blobs of one part, top rows background) and
    joint centres with visibility."""
    img = formula_input('g19.img', (B, 3, 256, 256), -2.0, 2.0)
    part = (formula_input('g19.part', (B, S // 4, S // 4)) * 25).floor().clamp(0, 24).repeat_interleave(4, 1).repeat_interleave(4, 2)
    part[:, :6] = 0
    gt = torch.cat([(part / 24.).unsqueeze(1), formula_input('g19.uv', (B, 2, S, S))], 1)
    gt[:, 1:] *= (part > 0).float().unsqueeze(1)
    kps = torch.cat([formula_input('g19.kps', (B, 24, 2), -0.8, 0.8), torch.ones(B, 24, 1)], -1)
    kps[0, 3, 2] = 0.0
    return img, gt, kps


def g19_grad_sample(gw):
    """What g19 / g20 keep of a weight gradient: all of a small one, every k-th element (~40 k of them) of a large one."""
    flat = gw.reshape(-1)
    return flat if flat.numel() <= 50000 else flat[::max(1, flat.numel() // 40000)].contiguous()


def g19_estimator256_b32():
    """The estimator half of the BENCHED train step on the reference itself: IUV_Estimator.forward in train mode at 32 x 256 x 256
    (HRNet-W48 -> heads -> soft-argmax -> 24 STN crops -> grouped partial head -> all eight IUV losses, jitters 0,
    align_corners = True as torch 1.1) and its backward pass: the loss scalars, the STN centres, sub-sampled predictions and four
    sentinel weight gradients (stem, a stage-3 branch conv, a global head, the grouped partial head)."""
    cfg = ref_env({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64, 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.,
                   'DANET.PARTDROP_RATE': 0.})
    import torch.nn.functional as F
    ag, gs = F.affine_grid, F.grid_sample
    F.affine_grid = lambda theta, size, align_corners=None: ag(theta, size, align_corners=True)
    F.grid_sample = lambda x, grid, mode='bilinear', padding_mode='zeros', align_corners=None: gs(x.to(grid.dtype), grid, mode, padding_mode, align_corners=True)
    try:
        from models.danet.iuv_estimator import IUV_Estimator
        torch.manual_seed(0)
        est = IUV_Estimator(pretrained=False)
        formula_params(est, skip=('learned_ratio', 'learned_offset'))
        est.train()
        B, S = 32, 64
        img, gt, kps = g19_inputs(B, S)
        has_iuv = torch.ones(B, dtype=torch.uint8).bool()
        names = ('iuv_est.conv1.weight', 'iuv_est.stage3.1.branches.2.1.conv2.weight', 'iuv_est.final_pred.predict_u.weight',
                 'iuv_est.final_pred.predict_partial_iuv.weight')
        # the same pass in DOUBLE first: how far the reference's own fp32 gradients are from it is the floor of any fp32 implementation
        # (the backward pass of a random-weight 90-layer ReLU net amplifies rounding: ~2 % on the stem's weight gradient)
        est.double()
        torch.set_default_dtype(torch.float64)
        try:
            rd64 = est(img.double(), gt.double(), kps.double(), has_iuv=has_iuv)
            sum(v.sum() for v in rd64['losses'].values()).backward()
        finally:
            torch.set_default_dtype(torch.float32)
        g64 = {n: p.grad.clone() for n, p in est.named_parameters() if n in names}
        l64 = {k: v.detach().reshape(-1).float() for k, v in rd64['losses'].items()}
        del rd64
        est.zero_grad(set_to_none=True)
        est.float()
        formula_params(est, skip=('learned_ratio', 'learned_offset'))         # (the double pass moved the running statistics)
        rd = est(img, gt, kps, has_iuv=has_iuv)
        total = sum(v.sum() for v in rd['losses'].values())
        total.backward()
        pd = dict(est.named_parameters())
        floors = {}
        for n in names:
            d = (pd[n].grad.double() - g64[n]).abs().max() / g64[n].abs().max()
            floors['gfloor__' + n.replace('.', '__')] = d.float()
            print(n, 'fp32 reference gradient vs fp64: %.3g of scale' % float(d))
        for k in l64:
            print(k, 'fp32 loss vs fp64: rel %.3g' % float((rd['losses'][k].detach().reshape(-1)[0] - l64[k][0]).abs() / l64[k][0].abs()))
        save('g19_estimator256_b32', learned_ratio=est.learned_ratio, learned_offset=est.learned_offset, **floors,
             **{'grad64__' + n.replace('.', '__'): g19_grad_sample(g64[n].float()) for n in names},
             index=_sub(rd['uvia_pred'][2], 16)[0], u=_sub(rd['uvia_pred'][0], 16)[0],
             stn_kps_pred=rd['stn_kps_pred'], part_iuv_pred=rd['part_iuv_pred'].detach()[:, ::6, :, :, ::16, ::16].contiguous(),
             **{'loss__' + k: v.detach().reshape(-1) for k, v in rd['losses'].items()},
             **{'grad__' + n.replace('.', '__'): g19_grad_sample(pd[n].grad) for n in names})
    finally:
        F.affine_grid, F.grid_sample = ag, gs


def g20_inputs(B=32, S=64):
    """Closed-form regressor inputs shaped like danet.py:247,276-283 produces them: cleaned global maps [B,75,S,S] (U | V | one-hot index)
    and cleaned partial maps [B,24,3,7,S,S]; one-hot planes from an arg-max of formula logits, U / V masked by them."""
    import torch.nn.functional as F
    idx = F.one_hot(formula_input('g20.idx', (B, 25, S, S)).argmax(1), 25).permute(0, 3, 1, 2).float()
    uv = formula_input('g20.uv', (B, 2, 25, S, S))
    iuv = torch.cat([uv[:, 0] * idx, uv[:, 1] * idx, idx], 1)
    pidx = F.one_hot(formula_input('g20.pidx', (B, 24, 7, S, S)).argmax(2), 7).permute(0, 1, 4, 2, 3).float()
    puv = formula_input('g20.puv', (B, 24, 2, 7, S, S))
    part = torch.stack([puv[:, :, 0] * pidx, puv[:, :, 1] * pidx, pidx], 2)
    return iuv, part


def g20_predictor_b32():
    """The regressor half of the BENCHED train step on the reference: DecomposedPredictor.forward (train mode) + backward at B = 32 --
    body_net on [32,75,64,64], limb_net on the 768 part maps, the grouped limb layer4, the three GCNs and the grouped regressors
    (smpl_regressor.py:397-928) -- para / joint positions / rotation, and sentinel weight gradients with the reference's own fp32 floor."""
    ref_env({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    from models.danet.smpl_regressor import DecomposedPredictor
    torch.manual_seed(0)
    pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
    mean = (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6)
    names = ('limb_net.0.weight', 'limb_net.3.conv1.weight', 'limb_net.3.layer3.1.conv2.weight', 'body_net.3.layer4.1.conv2.weight',
             'limb_reslayer.layer4.0.conv1.weight', 'refine_gcn.gc.1.weight')
    skip = ('mean_', 'I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A')
    iuv, part = g20_inputs()
    res = {}
    for dt in (torch.float64, torch.float32):
        net = DecomposedPredictor(None, mean, pretrained=False)
        formula_params(net, skip=skip)
        net = net.to(dt).train()
        torch.set_default_dtype(dt)
        try:
            rd = net(iuv.to(dt), part.to(dt))
            w = torch.cos(torch.arange(rd['para'].numel(), dtype=dt).view_as(rd['para']) * 0.37)
            loss = (rd['para'] * w).sum() + sum(t.sum() for t in rd['joint_position']) + rd['joint_rotation'][0].sum()
            loss.backward()
        finally:
            torch.set_default_dtype(torch.float32)
        pd = dict(net.named_parameters())
        res[dt] = ({'para': rd['para'].detach(), 'jp0': rd['joint_position'][0].detach(), 'jp1': rd['joint_position'][1].detach(),
                    'jr0': rd['joint_rotation'][0].detach()}, {n: pd[n].grad.clone() for n in names})
    out64, g64 = res[torch.float64]
    out32, g32 = res[torch.float32]
    arrs = {k: v.float() for k, v in out64.items()}
    for k in out64:
        arrs['floor__' + k] = (out32[k].double() - out64[k]).abs().max().float()
        print(k, 'fp32 vs fp64 reference: max abs %.3g' % float(arrs['floor__' + k]))
    for n in names:
        key = n.replace('.', '__')
        arrs['grad64__' + key] = g19_grad_sample(g64[n].float())
        arrs['gfloor__' + key] = ((g32[n].double() - g64[n]).abs().max() / g64[n].abs().max()).float()
        print(n, 'fp32 reference gradient vs fp64: %.3g of scale' % float(arrs['gfloor__' + key]))
    save('g20_predictor_b32', **arrs)


def g17_infer():
    """SURVEY 8 row f2 on the device (round-5 review item 4): the reference's inference path danet.py:61-131 -- IUV_Estimator (eval)
    -> iuvmap_clean -> per-part iuvmap_clean -> DecomposedPredictor (eval) -> para -- run with formula parameters; the GPU test
    writes the same parameters as a reference-layout checkpoint FILE, loads it into the HIP DaNet and calls infer_net."""
    ref_env({'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.,
             'DANET.PARTDROP_RATE': 0.})
    import torch.nn.functional as F
    ag, gs = F.affine_grid, F.grid_sample
    # the reference was written for torch 1.1 (align_corners=True semantics) -- the build's default
    F.affine_grid = lambda theta, size, align_corners=None: ag(theta, size, align_corners=True)
    F.grid_sample = lambda x, grid, mode='bilinear', padding_mode='zeros', align_corners=None: gs(x, grid, mode, padding_mode, align_corners=True)
    try:
        from models.danet.iuv_estimator import IUV_Estimator
        from models.danet.smpl_regressor import DecomposedPredictor
        from utils.iuvmap import iuvmap_clean
        torch.manual_seed(0)
        est = IUV_Estimator(pretrained=False)
        formula_params(est, skip=('learned_ratio', 'learned_offset'))
        pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
        mean = (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6)
        pred = DecomposedPredictor(None, mean, pretrained=False)
        formula_params(pred, skip=('mean_', 'I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A'))
        est.eval(); pred.eval()
        img = formula_input('g17.img', (2, 3, 128, 128), -2.0, 2.0)
        with torch.no_grad():
            uv = est(img)
            u, v, idx, ann = iuvmap_clean(*uv['uvia_pred'])
            iuv_map = torch.cat([u, v, idx], dim=1)
            pp = uv['part_iuv_pred']
            parts = []
            for p in range(pp.size(1)):                                       # danet.py:93-100
                pu, pv, pi, _ = iuvmap_clean(pp[:, p, 0], pp[:, p, 1], pp[:, p, 2])
                parts.append(torch.stack([pu, pv, pi], dim=1))
            part_iuv_map = torch.stack(parts, dim=1)
            para = pred(iuv_map, part_iuv_map)['para']
        # how decisive the arg-max planes are: the smallest top-1 / top-2 logit gap (a flip would change the regressor's input)
        top = uv['uvia_pred'][2].topk(2, dim=1).values
        save('g17_infer', learned_ratio=est.learned_ratio, learned_offset=est.learned_offset, para=para,
             u_raw=uv['uvia_pred'][0], index_raw=uv['uvia_pred'][2], ann_raw=uv['uvia_pred'][3], stn_kps_pred=uv['stn_kps_pred'],
             part_iuv_pred=pp[:, ::6].contiguous(), index_clean=idx.argmax(1).to(torch.uint8),
             min_index_gap=(top[:, 0] - top[:, 1]).min())
    finally:
        F.affine_grid, F.grid_sample = ag, gs


ALL['g16'] = g16_hrnet256
ALL['g17'] = g17_infer
ALL['g18'] = g18_hrnet256_b32
ALL['g19'] = g19_estimator256_b32
ALL['g20'] = g20_predictor_b32


def _main():
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()


if __name__ == '__main__':
    _main()
