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
    elif leaf == 'weight' and len(shape) == 1:        # BN / norm gamma
        v = 0.8 + 0.4 * u
    elif leaf == 'bias':
        v = (u - 0.5) * 0.1
    else:                                             # conv / linear / gcn weights
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if leaf == 'weight' and len(shape) == 2 and 'gc.' in key:
            fan_in = shape[0]
        v = (u - 0.5) * 2.0 * np.sqrt(3.0 / max(fan_in, 1))
    return torch.from_numpy(v.reshape(shape).astype(np.float32)).to(dtype)


def formula_params(module, skip=()):
    sd = module.state_dict()
    new = {}
    for k, t in sd.items():
        if any(k.startswith(s) for s in skip):
            new[k] = t
        else:
            new[k] = formula_tensor(k, tuple(t.shape), t.dtype) if t.dtype.is_floating_point else t
    module.load_state_dict(new)


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


ALL = {'g1': g1_geometry, 'g2': g2_iuvmap, 'g3': g3_graph}

if __name__ == '__main__':
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()
