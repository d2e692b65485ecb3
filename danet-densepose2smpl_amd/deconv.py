"""ConvTranspose2d on the MFMA conv kernels (PoseResNet deconv head,
/root/reference/models/module/res_module.py:169-194).

A transposed convolution with weight W_t[Cin][Cout][R][S] is the data gradient of the ordinary
strided convolution C' whose weight tensor is that same array read as [Cout'=Cin][Cin'=Cout][R][S]:
  forward   = dgrad(C')   -> conv kernel with the transposed gather (mode-1 packing)
  d input   = forward(C') -> ordinary strided conv (mode-0 packing)
  d weight  = wgrad(C') with C' input := grad_output, C' output-grad := input.
"""
import torch
import torch.nn as nn

from . import _lib
from ._lib import ptr, check, stream
from .conv import nhwc_bf16, pack_weight, _conv_fwd_raw, new_wgrad


class ConvTranspose2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, outpad):
        x = nhwc_bf16(x)
        B, Cin, H, W = x.shape
        Cin_w, Cout, R, S = weight.shape
        if Cin_w != Cin:
            raise ValueError('conv_transpose2d: input has %d channels, weight expects %d' % (Cin, Cin_w))
        OH, OW = (H - 1) * stride - 2 * pad + R + outpad, (W - 1) * stride - 2 * pad + S + outpad
        wp = pack_weight(weight, 1, 1)
        b = None if bias is None else bias.detach().float().contiguous()
        y = _conv_fwd_raw(x, wp, b, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, 1, 1, True, False, False)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        B, Cin, H, W = x.shape
        _, Cout, R, S = weight.shape
        gy = nhwc_bf16(gy)
        OH, OW = gy.shape[2], gy.shape[3]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wp0 = pack_weight(weight, 1, 0)
            gx = _conv_fwd_raw(gy, wp0, None, B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, 1, 1, False, False, False)
        if ctx.needs_input_grad[1]:
            gw = new_wgrad(weight, (Cin, Cout, R, S), x.device)
            nws = L.danet_conv_wgrad_ws_floats(Cin, Cout, R, S)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            check(L.danet_conv_wgrad(ptr(gy.permute(0, 2, 3, 1)), ptr(x.permute(0, 2, 3, 1)), ptr(gw), ptr(ws), nws,
                                     B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, 1, 1, 0.0, 0, stream()), 'danet_conv_wgrad')
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.float().sum(dim=(0, 2, 3))
        return gx, gw, gb, None, None, None


class ConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, cin, cout, kernel, stride, padding, output_padding, bias=False):
        super().__init__(cin, cout, kernel, stride, padding, output_padding, bias=bias)

    def forward(self, x):
        from . import conv as _conv
        if _conv.PRECISION == 'fp32':                       # verification mode: the transposed conv as an fp32 tensor op
            import torch.nn.functional as F
            return F.conv_transpose2d(x.float(), self.weight, self.bias, self.stride, self.padding, self.output_padding)
        return ConvTranspose2dFunction.apply(x, self.weight, self.bias, self.stride[0], self.padding[0], self.output_padding[0])
