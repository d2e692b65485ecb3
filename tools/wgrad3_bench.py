"""The 3x3 / stride-1 weight gradients of one HRNet stage-4 module slice (12 problems: three block levels x four branches, B = 32) through
danet_conv_wgrad3x3_multi, replayed from a hipGraph; checks the result against F.conv2d's weight gradient on the bf16-rounded operands.
(Round 6 used it to measure a 96 x 48 block variant: 229 us against 107 -- dropped.)  usage: python tools/wgrad3_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import _lib                    # noqa: E402
from danet_densepose2smpl_amd._lib import ptr, check, stream    # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
B = 32
shapes = [(48, 64), (96, 32), (192, 16), (384, 8)] * 3
g = torch.Generator().manual_seed(0)
xs = [torch.randn(B, H, H, C, generator=g).to(dev).bfloat16() for C, H in shapes]
dys = [torch.randn(B, H, H, C, generator=g).to(dev).bfloat16() for C, H in shapes]
dws = [torch.empty(C, C, 3, 3, device=dev) for C, H in shapes]
n = len(shapes)
jobs = (_lib.Wg3Job * n)()
for j, x, dy, dw, (C, H) in zip(jobs, xs, dys, dws, shapes):
    j.x, j.dy, j.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    j.B, j.H, j.W, j.Cin, j.Cout, j.groups, j.stride = B, H, H, C, C, 1, 1
need = L.danet_conv_wgrad3x3_multi_ws_floats(ctypes.addressof(jobs), n)
ws = torch.empty(need, dtype=torch.float32, device=dev)
run = lambda: check(L.danet_conv_wgrad3x3_multi(ctypes.addressof(jobs), n, ptr(ws), need, 0.0, stream()), 'wgrad3x3_multi')     # noqa: E731
run()
torch.cuda.synchronize()
worst = 0.0
for x, dy, dw, (C, H) in list(zip(xs, dys, dws, shapes))[:4]:
    xr = x.float().permute(0, 3, 1, 2).contiguous()
    w = torch.zeros(C, C, 3, 3, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(xr, w, None, 1, 1)
    gw, = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
    worst = max(worst, float((dw - gw).abs().max() / gw.abs().max()))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        run()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e3 / 20
flops = sum(2.0 * B * H * H * C * C * 9 for C, H in shapes)
print('[%s] 12 problems: %.1f us per flush (kernels + reductions) = %.0f TFLOP/s = %.3f of the bf16 peak; max rel err vs F.conv2d %.2e' %
      ('default', t, flops / t / 1e6, flops / t / 1e6 / 2500.0, worst))
