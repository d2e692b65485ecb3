"""Times csrc/stn.hip at the bench shape (B = 32, 48 channels, 64 x 64, 24 parts: 12.6 MB in, 302 MB out), replayed from a hipGraph,
on (a) the thetas of the benched train step itself (captured from one eager step of the synthetic batch) and (b) spread-out
joint-centric thetas (scales 0.15 .. 0.7).  DANET_STN_V1=1 selects the round-1 kernels.  usage: python tools/stn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import nn as dnn, iuv_estimator      # noqa: E402

dev = torch.device('cuda')
B, C, H, P = 32, 48, 64, 24


def step_thetas():
    from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    torch.manual_seed(1234)
    tr = Trainer(default_options(B), device=dev, distributed=False)
    batch = synthetic_in_dict(tr.model, B, dev, seed=1234)
    got = []
    orig = iuv_estimator.stn_gather if hasattr(iuv_estimator, 'stn_gather') else None
    name = 'stn_gather'
    mod = iuv_estimator if orig is not None else dnn
    orig = getattr(mod, name)

    def spy(x, theta, *a, **k):
        got.append(theta.detach().clone())
        return orig(x, theta, *a, **k)
    setattr(mod, name, spy)
    try:
        tr.train_step(batch)
    finally:
        setattr(mod, name, orig)
    torch.cuda.synchronize()
    del tr
    return got[0]


def spread_thetas():
    g = torch.Generator().manual_seed(0)
    s = torch.rand(B, P, generator=g) * 0.55 + 0.15
    c = (torch.rand(B, P, 2, generator=g) - 0.5) * 1.2
    theta = torch.zeros(B, P, 2, 3)
    theta[:, :, 0, 0] = s; theta[:, :, 1, 1] = s; theta[:, :, :, 2] = c
    return theta.to(dev)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            fn()
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            gr.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


from danet_densepose2smpl_amd import _lib                    # noqa: E402
from danet_densepose2smpl_amd._lib import ptr, check, stream    # noqa: E402
L = _lib.lib()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, H, C, generator=g).to(dev).bfloat16().contiguous()           # NHWC
y = torch.empty(B, H, H, P * C, dtype=torch.bfloat16, device=dev)
gy = torch.randn(B, H, H, P * C, generator=g).to(dev).bfloat16().contiguous()
dx = torch.empty_like(x)
tag = 'v1' if os.environ.get('DANET_STN_V1', '0') not in ('', '0') else 'round 6'
for nm, th in (('thetas of the benched step', step_thetas()), ('spread thetas', spread_thetas())):
    th = th.reshape(B, P, 2, 3).float().contiguous()
    sc = th[:, :, 0, 0]
    fwd = lambda: check(L.danet_stn_gather_forward(ptr(x), ptr(th), B, H, H, C, P, H, H, 1, ptr(y), stream()), 'fwd')     # noqa: E731
    bwd = lambda: check(L.danet_stn_gather_backward(ptr(gy), ptr(th), B, H, H, C, P, H, H, 1, ptr(dx), stream()), 'bwd')   # noqa: E731
    t_f, t_b = timed(fwd), timed(bwd)
    mb = y.numel() * 2 / 1e6
    print('[%s] %s (scale min %.3f median %.3f max %.3f): forward %.1f us (%.2f TB/s written), backward %.1f us (%.2f TB/s read)  dx checksum %.6g' %
          (tag, nm, float(sc.min()), float(sc.median()), float(sc.max()), t_f, mb / t_f, t_b, mb / t_b, float(dx.float().abs().sum())))
