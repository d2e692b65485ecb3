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


class ConvTranspose2dF32Function(torch.autograd.Function):
    """The same three passes in fp32 on the fp32 MFMA kernels (csrc/conv_f32m.hip); channel counts multiples of 4 (16 for
    the gathered side of the strided pass) -- ConvTranspose2d.forward checks danet_conv_f32m_ok first."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, outpad):
        from . import conv as _conv
        L = _lib.lib()
        x = _conv.nhwc_as(x, torch.float32)
        B, Cin, H, W = x.shape
        _, Cout, R, S = weight.shape
        OH, OW = (H - 1) * stride - 2 * pad + R + outpad, (W - 1) * stride - 2 * pad + S + outpad
        w = weight.detach().float().contiguous()              # C' weight [Cout' = Cin][Cin' = Cout][R][S]
        wp = _conv._pack_weight_f32(weight, w, 1, 1, Cin, Cout)
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty(B, OH, OW, Cout, dtype=torch.float32, device=x.device)
        check(L.danet_conv_f32m_forward(ptr(x.permute(0, 2, 3, 1)), ptr(wp), ptr(b), ptr(y), B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, 1, 1, 1, 0,
                                        stream()), 'danet_conv_f32m_forward')
        ctx.save_for_backward(x, w)
        ctx.weight = weight
        ctx.cfg = (stride, pad, bias is not None)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from . import conv as _conv
        L = _lib.lib()
        x, w = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        B, Cin, H, W = x.shape
        _, Cout, R, S = w.shape
        gy = _conv.nhwc_as(gy, torch.float32)
        OH, OW = gy.shape[2], gy.shape[3]
        g = gy.permute(0, 2, 3, 1)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:           # forward of C': gy [Cout ch, OH x OW] -> [Cin ch, H x W]
            wp0 = _conv._pack_weight_f32(ctx.weight, w, 1, 0, Cin, Cout)
            gx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=gy.device)
            check(L.danet_conv_f32m_forward(ptr(g), ptr(wp0), None, ptr(gx), B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, 1, 1, 0, 0, stream()),
                  'danet_conv_f32m_forward')
            gx = gx.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:           # wgrad of C': input := gy, output gradient := x
            gw = torch.empty_like(w)
            wdims = (B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, 1, 1, Cin, Cout)
            ws = torch.empty(L.danet_conv_f32m_wgrad_ws_floats(*wdims), dtype=torch.float32, device=gy.device)
            check(L.danet_conv_f32m_wgrad(ptr(g), ptr(x.permute(0, 2, 3, 1)), ptr(gw), ptr(ws), *wdims, stream()), 'danet_conv_f32m_wgrad')
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(dim=(0, 2, 3))
        return gx, gw, gb, None, None, None


class ConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, cin, cout, kernel, stride, padding, output_padding, bias=False):
        super().__init__(cin, cout, kernel, stride, padding, output_padding, bias=bias)

    def forward(self, x):
        from . import conv as _conv
        if _conv.PRECISION == 'fp32':
            B, Cin, H, W = x.shape
            Cout, (R, S), st, pd, op = self.weight.shape[1], self.kernel_size, self.stride[0], self.padding[0], self.output_padding[0]
            OH, OW = (H - 1) * st - 2 * pd + R + op, (W - 1) * st - 2 * pd + S + op
            L = _lib.lib()
            if x.is_cuda and _conv.F32_MFMA and L.danet_conv_f32m_ok(B, H, W, Cin, OH, OW, Cout, R, S, st, pd, 1, 1, 1) and \
                    L.danet_conv_f32m_ok(B, OH, OW, Cout, H, W, Cin, R, S, st, pd, 1, 1, 0):
                return ConvTranspose2dF32Function.apply(x, self.weight, self.bias, st, pd, op)
            # no tensor-op fall-back: the path's transposed convolutions are 2048 -> 256 and 256 -> 256 (res_module.py:169-194)
            raise RuntimeError('ConvTranspose2d (fp32 mode): %d -> %d channels, kernel %d, stride %d on %s is outside the fp32 MFMA kernels '
                               '(csrc/conv_f32m.hip: channel counts in multiples of 4, device tensors)' % (Cin, Cout, R, st, x.device))
        return ConvTranspose2dFunction.apply(x, self.weight, self.bias, self.stride[0], self.padding[0], self.output_padding[0])
