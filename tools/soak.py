"""Repeat the round-4 kernels with hand-counted waits / tickets many times on fixed inputs and demand bit-identical outputs every time
(no output of these kernels goes through atomics): a wait that is one count short shows up as a rare mismatch long before it shows up as
a NaN in a bench.  usage: python tools/soak.py [iterations]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv, _lib          # noqa: E402

L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
torch.manual_seed(0)
rec = {}


def soak(name, fn, n=N):
    ref = fn().clone()
    bad = 0
    for i in range(n):
        out = fn()
        if i % 16 == 15 or i == n - 1:                 # compare every 16th launch (and the last): the launches in between overlap freely
            bad += int(not torch.equal(out, ref))
    torch.cuda.synchronize()
    rec[name] = {'launches': n, 'mismatches': bad, 'finite': bool(torch.isfinite(ref.float()).all())}


B, C, H = 768, 64, 64
x = conv.nhwc_bf16(torch.randn(B, C, H, H, device='cuda'))
gy = conv.nhwc_bf16(torch.randn(B, C, H // 2, H // 2, device='cuda'))
w = torch.nn.Parameter(torch.randn(C, C, 7, 7, device='cuda') * 0.02)
wp16, wp16t = conv.pack_weight(w, 1, 0, 16), conv.pack_weight(w, 1, 1, 16)
soak('conv_stem_kernel', lambda: conv._conv_stem_raw(x, wp16, B, H, H, C, H // 2, H // 2, C))
soak('conv_stem_dgrad_kernel', lambda: conv._conv_stem_dgrad_raw(gy, wp16t, B, H, H, C, H // 2, H // 2, C, None))
# odd tile counts per workgroup (B = 70: 280 tiles over 256 workgroups)
x2, gy2 = x[:70].contiguous(memory_format=torch.channels_last), gy[:70].contiguous(memory_format=torch.channels_last)
soak('conv_stem_kernel B=70', lambda: conv._conv_stem_raw(x2, wp16, 70, H, H, C, H // 2, H // 2, C))
soak('conv_stem_dgrad_kernel B=70', lambda: conv._conv_stem_dgrad_raw(gy2, wp16t, 70, H, H, C, H // 2, H // 2, C, None))
# pointwise forward (weights in LDS, prefetch ring) and the grouped head on the streamed kernel
xp = conv.nhwc_bf16(torch.randn(32, 64, 64, 64, device='cuda'))
wpw = torch.nn.Parameter(torch.randn(256, 64, 1, 1, device='cuda') * 0.1)
soak('conv_pw_kernel 64->256', lambda: conv._conv_fwd_raw(xp, conv.pack_weight(wpw, 1, 0), None, 32, 64, 64, 64, 64, 64, 256, 1, 1, 1, 0, 1, 1, False, False, False))
G = 24
xg = conv.nhwc_bf16(torch.randn(32, G * 48, 64, 64, device='cuda'))
wg = torch.nn.Parameter(torch.randn(G * 24, 48, 3, 3, device='cuda') * 0.05)
soak('grouped head on conv3x3_stream_kernel', lambda: conv._conv_fwd_raw(xg, conv.pack_weight(wg, G, 0), None, 32, 64, 64, G * 48, 64, 64, G * 24, 3, 3, 1, 1, 1, G, False, False, False), n=N // 3)
# SMPL forward as one launch (ticket, last-arriver reduction)
from danet_densepose2smpl_amd import assets, smpl as dsmpl    # noqa: E402
model = dsmpl.SMPL(assets.make_synthetic_smpl(0)).cuda()
betas = torch.randn(32, 10, device='cuda')
rot = torch.linalg.qr(torch.randn(32, 24, 3, 3, device='cuda'))[0]
with torch.no_grad():
    soak('smpl_fused_fwd_kernel', lambda: model(betas=betas, body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False).vertices, n=2 * N)
# SMPL backward as ONE launch (round 5: three phases, two fence-free grid barriers, partials written and read at agent scope): a stale
# read across a barrier would show as a mismatch against the first call's gradients
from danet_densepose2smpl_amd import ops, nn as dnn    # noqa: E402
ops.SMPL_BWD_FUSED = True       # (opt-in since round 6; this soak is about the one-launch kernel)
gvs, gjs = torch.randn(32, 6890, 3, device='cuda') * 1e-2, torch.randn(32, 54, 3, device='cuda')


def lbs_bwd():
    tb, tr = betas.clone().requires_grad_(True), rot.clone().requires_grad_(True)
    v_, j_ = ops.smpl_lbs(tb, tr, model)
    gb, gr = torch.autograd.grad((v_ * gvs).sum() + (j_ * gjs).sum(), [tb, tr])
    return torch.cat([gb.reshape(-1), gr.reshape(-1)])


conv.FUSION.clear()
soak('smpl_fused_bwd_kernel', lbs_bwd, n=N)
rec['smpl_fused_bwd_kernel']['one_launch_calls'] = int(conv.FUSION.get('smpl_bwd_fused', 0))
rec['smpl_fused_bwd_kernel']['barrier_error'] = bool(dnn.onepass_error())
# the four-branch lockstep launch of conv3x3_stream_kernel (the step's workhorse) -- alone, and with a memory-hungry copy running on a second
# stream: its stage copies are published on the strength of a vmcnt count that assumes they complete before younger register loads
# ("a device fact found on the way"); a late copy would show here as a mismatch
import ctypes    # noqa: E402
chans, sizes = (48, 96, 192, 384), (64, 32, 16, 8)
xs = [conv.nhwc_bf16(torch.randn(32, c, s_, s_, device='cuda')) for c, s_ in zip(chans, sizes)]
wps = [conv.pack_weight(torch.nn.Parameter(torch.randn(c, c, 3, 3, device='cuda') * 0.05), 1, 0) for c in chans]
ys = [torch.empty_like(x_) for x_ in xs]
jobs = (_lib.ConvJob * 4)()
for j, x_, wp_, y_, c, s_ in zip(jobs, xs, wps, ys, chans, sizes):
    conv._conv_job(j, x_, wp_, y_, (32, s_, s_, c, s_, s_, c, 3, 3, 1, 1, 1, 1), False, None)
assert 'conv3x3_stream' in conv._multi_kernel_name(jobs, 4, 48)


def four_branch():
    conv.check(L.danet_conv_forward_multi(ctypes.addressof(jobs), 4, _lib.stream()), 'multi')
    return torch.cat([y_.reshape(-1) for y_ in ys])


soak('four-branch conv3x3_stream_kernel', four_branch, n=N)
big_a, big_b = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'), torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
side = torch.cuda.Stream()
stop = {'n': 0}


def four_branch_contended():
    with torch.cuda.stream(side):
        big_b.copy_(big_a, non_blocking=True)          # 0.5 GB of HBM traffic per call on the other stream
    return four_branch()


soak('four-branch conv3x3_stream_kernel under a concurrent 256 MB copy', four_branch_contended, n=N)
torch.cuda.synchronize()
print(json.dumps(rec))
