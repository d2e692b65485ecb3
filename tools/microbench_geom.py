"""Micro-benchmark of the SMPL layer and the IUV rasteriser on one GPU (HIP events).
Prints one JSON line per kernel group with the algorithmic-bytes roofline."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import assets, ops          # noqa: E402
from danet_densepose2smpl_amd.smpl import SMPL            # noqa: E402
from danet_densepose2smpl_amd.renderer import IUV_Renderer  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device('cuda:0')
    model = assets.make_synthetic_smpl(0)
    smpl = SMPL(model).to(dev)
    rend = IUV_Renderer(256, 64, smpl_model=model)
    g = torch.Generator(device='cpu').manual_seed(0)
    betas = torch.randn(B, 10, generator=g).clamp(-3, 3).to(dev)
    pose = (torch.randn(B, 72, generator=g) * 0.2).to(dev)
    rot = ops.rodrigues_smplx(pose.view(-1, 3)).view(B, 24, 3, 3)
    cam = torch.tensor([[0.9, 0.0, 0.0]] * B, device=dev)
    consts = 4 * (207 * 20670 + 20670 * 10 + 6890 * 24 + 20670 + 9 * 6890)
    per_item = 4 * (226 + (6890 + 54) * 3)
    fwd_bytes = consts + B * per_item
    t = timeit(lambda: ops.smpl_lbs(betas, rot, smpl))
    print(json.dumps({'kernel': 'smpl_lbs_forward', 'B': B, 'us': t * 1e6, 'items_per_s': B / t,
                      'alg_bytes': fwd_bytes, 'GBps': fwd_bytes / t / 1e9, 'frac_hbm': fwd_bytes / t / HBM_PEAK}))
    bt, rt = betas.clone().requires_grad_(True), rot.clone().requires_grad_(True)
    gv = torch.randn(B, 6890, 3, device=dev)
    gj = torch.randn(B, 54, 3, device=dev)

    def fb():
        v, j = ops.smpl_lbs(bt, rt, smpl)
        torch.autograd.backward([v, j], [gv, gj])
        bt.grad = None
        rt.grad = None
    t2 = timeit(fb, iters=100)
    bwd_bytes = consts + B * 4 * (2 * 6890 * 3 + 54 * 3 + 226)
    print(json.dumps({'kernel': 'smpl_lbs_fwd+bwd', 'B': B, 'us': t2 * 1e6, 'bwd_only_us': (t2 - t) * 1e6,
                      'alg_bytes': fwd_bytes + bwd_bytes, 'GBps': (fwd_bytes + bwd_bytes) / t2 / 1e9}))
    verts, _ = ops.smpl_lbs(betas, rot, smpl)
    r_bytes = B * (6890 * 12 + 3 * 64 * 64 * 4) + 7829 * 4 + 13774 * 24
    t3 = timeit(lambda: rend.verts2uvimg(verts, cam))
    print(json.dumps({'kernel': 'iuv_raster_forward', 'B': B, 'us': t3 * 1e6, 'images_per_s': B / t3,
                      'alg_bytes': r_bytes, 'GBps': r_bytes / t3 / 1e9, 'frac_hbm': r_bytes / t3 / HBM_PEAK}))


if __name__ == '__main__':
    main()
