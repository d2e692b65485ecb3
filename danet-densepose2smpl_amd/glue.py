"""Launch-count glue on csrc/glue.hip: several small zero-pad copies in one launch (pad_multi) -- the parameters of layers whose
widths are no multiple of 8 are padded every step (F.pad = fill + copy forward, copy backward, per tensor).  GPU only."""
import ctypes

import torch

from . import _lib
from ._lib import check, stream, ptr

PAD_MAX = 16


def _dims4(shape):
    shape = tuple(int(s) for s in shape)
    if len(shape) > 4:
        raise ValueError('pad_multi: at most 4 dimensions')
    return (1,) * (4 - len(shape)) + shape


def _launch(srcs, src_views, dst_views, out_shapes, outs=None):
    """srcs[k] (dense fp32, numel == prod(src_views[k])) -> new tensor of out_shapes[k] (numel == prod(dst_views[k])), or into the given
    dense fp32 tensors `outs`."""
    L = _lib.lib()
    if outs is None:
        outs = [torch.empty(tuple(s), dtype=torch.float32, device=srcs[0].device) for s in out_shapes]
    for k0 in range(0, len(srcs), PAD_MAX):
        n = min(PAD_MAX, len(srcs) - k0)
        sp = (ctypes.c_void_p * n)(*[srcs[k0 + k].data_ptr() for k in range(n)])
        dp = (ctypes.c_void_p * n)(*[outs[k0 + k].data_ptr() for k in range(n)])
        sd = (ctypes.c_int * (4 * n))(*[v for k in range(n) for v in _dims4(src_views[k0 + k])])
        dd = (ctypes.c_int * (4 * n))(*[v for k in range(n) for v in _dims4(dst_views[k0 + k])])
        check(L.danet_pad_multi(ctypes.addressof(sp), ctypes.addressof(dp), ctypes.addressof(sd), ctypes.addressof(dd), n, stream()),
              'danet_pad_multi')
    return outs


class PadMultiFunction(torch.autograd.Function):
    """spec[k] = (source view shape, padded view shape, output shape); tensors[k] dense fp32.  One launch forward, one backward."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        for t in tensors:
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError('pad_multi runs on fp32 GPU tensors only')
        ctx.spec = spec
        ctx.set_materialize_grads(False)          # an unused output's gradient stays None (no zero tensor, no crop job)
        ctx.shapes = [tuple(t.shape) for t in tensors]
        outs = _launch([t.detach().contiguous() for t in tensors], [s[0] for s in spec], [s[1] for s in spec], [s[2] for s in spec])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        idx = [k for k, g in enumerate(gs) if g is not None and ctx.needs_input_grad[k + 1]]
        res = [None] * len(gs)
        if idx:
            outs = _launch([gs[k].float().contiguous() for k in idx], [ctx.spec[k][1] for k in idx], [ctx.spec[k][0] for k in idx],
                           [ctx.shapes[k] for k in idx])
            for k, o in zip(idx, outs):
                res[k] = o
        return (None,) + tuple(res)


def pad_multi(items):
    """items: [(tensor, padded shape)] or [(tensor, source view, padded view, output shape)] -> padded tensors (zeros behind every
    dimension's end), autograd-tracked; one launch for all of them."""
    spec, tensors = [], []
    for it in items:
        if len(it) == 2:
            t, shape = it
            spec.append((tuple(t.shape), tuple(shape), tuple(shape)))
        else:
            t, sv, dv, shape = it
            spec.append((tuple(sv), tuple(dv), tuple(shape)))
        tensors.append(t)
    return PadMultiFunction.apply(tuple(spec), *tensors)


def crop_into(srcs, src_views, dst_views, outs):
    """crop for several tensors in one launch per 16, written into EXISTING dense fp32 tensors (the deferred weight gradients of
    channel-padded layers: computed at the padded widths by the multi-problem launches, cropped into the parameters' gradient slots)."""
    _launch(list(srcs), list(src_views), list(dst_views), None, outs=list(outs))


def crop(t, src_view, dst_view, out_shape):
    """The leading block of a dense fp32 tensor: t viewed as src_view -> new tensor of out_shape (viewed as dst_view <= src_view in every
    dimension); one launch, no autograd (the backward of a pad, used on weight gradients computed at padded widths)."""
    return _launch([t.contiguous()], [src_view], [dst_view], [out_shape])[0]


def loss_finalize(n, scales, w, nw, sums=None, rows=1, grads=None):
    """csrc/glue.hip danet_loss_finalize: the n finished losses from raw double sums (forward: `sums`, rows x n doubles) or the n
    backward coefficients from the incoming gradients (`grads`: list of 1-element tensors / None).  scales = ((a_i, b_i), ...):
    loss_i = sum_i * a_i / (max(sum(w), 1) * b_i) where b_i > 0, sum_i * a_i otherwise; w = per-sample weights [nw] or None (= ones)."""
    import ctypes
    dev = (sums if sums is not None else next(g for g in grads if g is not None)).device
    a = (ctypes.c_float * 8)(*([float(x[0]) for x in scales] + [0.0] * (8 - n)))
    b = (ctypes.c_float * 8)(*([float(x[1]) for x in scales] + [0.0] * (8 - n)))
    out = torch.empty(n, dtype=torch.float32, device=dev)
    gp = None
    keep = []
    if grads is not None:
        arr = (ctypes.c_void_p * 8)()
        for i, g in enumerate(grads):
            if g is not None:
                g = g.detach().to(torch.float32).contiguous()
                keep.append(g)
                arr[i] = g.data_ptr()
        gp = ctypes.addressof(arr)
    check(_lib.lib().danet_loss_finalize(ptr(sums), int(rows), int(n), ctypes.addressof(a), ctypes.addressof(b), ptr(w), int(nw), gp, ptr(out), stream()),
          'danet_loss_finalize')
    return out


class RegroupPartsFunction(torch.autograd.Function):
    """[NB * J, C, H, W] channels-last (the J part crops of each image) -> [NB, J * C, H, W] channels-last (J channel groups of one map):
    /root/reference/models/danet/smpl_regressor.py:826 `limb_feat.view(nbs, -1, h, w)`, one launch each way (csrc/glue.hip)."""

    @staticmethod
    def forward(ctx, x, NB):
        BJ, C, H, W = x.shape
        J = BJ // NB
        xc = x.detach()
        if not xc.permute(0, 2, 3, 1).is_contiguous():
            xc = xc.contiguous(memory_format=torch.channels_last)
            if not xc.permute(0, 2, 3, 1).is_contiguous():
                xc = xc.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        y = torch.empty(NB, H, W, J * C, dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
        check(_lib.lib().danet_regroup_parts(ptr(xc.permute(0, 2, 3, 1)), ptr(y.permute(0, 2, 3, 1)), NB, J, H * W, C * x.element_size(), 0, stream()),
              'danet_regroup_parts')
        ctx.dims = (NB, J, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        NB, J, C, H, W = ctx.dims
        g = gy
        if not g.permute(0, 2, 3, 1).is_contiguous():
            g = g.contiguous(memory_format=torch.channels_last)
            if not g.permute(0, 2, 3, 1).is_contiguous():
                g = g.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        gx = torch.empty(NB * J, H, W, C, dtype=g.dtype, device=g.device).permute(0, 3, 1, 2)
        check(_lib.lib().danet_regroup_parts(ptr(g.permute(0, 2, 3, 1)), ptr(gx.permute(0, 2, 3, 1)), NB, J, H * W, C * g.element_size(), 1, stream()),
              'danet_regroup_parts')
        return gx, None


def regroup_parts(x, NB):
    """x [NB * J, C, H, W] -> [NB, J * C, H, W] (== x.reshape(NB, -1, H, W)); on the GPU with 16-byte pixel rows one launch, channels-last
    in and out."""
    if x.is_cuda and x.shape[0] % NB == 0 and (x.shape[1] * x.element_size()) % 16 == 0:
        return RegroupPartsFunction.apply(x, NB)
    return x.reshape(NB, -1, x.size(-2), x.size(-1))


def pack_image(x):
    """fp32 NCHW image batch [B, C <= 8, H, W] -> bf16 channels-last [B, 8, H, W] with zero channels behind C (no gradient)."""
    B, C, H, W = x.shape
    xc = x.detach()
    if xc.dtype != torch.float32 or not xc.is_contiguous():
        xc = xc.float().contiguous()
    y = torch.empty(B, H, W, 8, dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    check(_lib.lib().danet_pack_image(ptr(xc), ptr(y.permute(0, 2, 3, 1)), B, C, H, W, stream()), 'danet_pack_image')
    return y
