"""Fused partial-IUV ("limb") element-wise path on HIP kernels (csrc/part_ops.hip).

`part_clean`  = part drop + iuvmap_clean of the 24 partial maps (/root/reference/models/danet/danet.py:264-283)
                producing directly the zero-padded 24-channel NHWC bf16 operand of the limb regressor's first conv.
`part_losses` = part_iuv_simp + affine_grid/grid_sample of the ground truth + body_uv_losses summed over the
                24 joints (/root/reference/models/danet/iuv_estimator.py:206-246), as three raw sums.
Both take the grouped conv's output [B, 24*21, H, W] (bf16, channels_last) as it is.
"""
import torch

from . import _lib
from ._lib import ptr, check, stream
from .conv import nhwc_bf16, ARENA

NJ, NC = 24, 7


def _cpj(C):
    """channels per joint in memory: 21, or 24 for the grouped conv's zero-padded output"""
    if C not in (NJ * 21, NJ * 24):
        raise ValueError('partial IUV prediction must have %d or %d channels, got %d' % (NJ * 21, NJ * 24, C))
    return C // NJ


def padded_view6(pp):
    """[B, 24*24, H, W] group-padded prediction -> its [B,24,3,7,H,W] view (no copy); the padded tensor rides along
    as `._padded` so that part_clean / part_losses read it directly."""
    B, C, H, W = pp.shape
    v = pp.permute(0, 2, 3, 1).reshape(B, H, W, NJ, C // NJ)[..., :21].reshape(B, H, W, NJ, 3, NC).permute(0, 3, 4, 5, 1, 2)
    v._padded = pp
    return v


class PartCleanFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, keep):
        pred = nhwc_bf16(pred)
        B, C, H, W = pred.shape
        cpj = _cpj(C)
        k = None if keep is None else keep.detach().to(torch.float32).contiguous()
        x24 = torch.empty(B * NJ, H, W, 24, dtype=torch.bfloat16, device=pred.device).permute(0, 3, 1, 2)
        check(_lib.lib().danet_part_clean_forward(ptr(pred.permute(0, 2, 3, 1)), ptr(k), B, H, W, cpj, ptr(x24.permute(0, 2, 3, 1)), stream()),
              'danet_part_clean_forward')
        ctx.save_for_backward(pred, k)
        return x24

    @staticmethod
    def backward(ctx, g24):
        pred, k = ctx.saved_tensors
        B, C, H, W = pred.shape
        g24 = nhwc_bf16(g24)
        gp = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=pred.device).permute(0, 3, 1, 2)
        check(_lib.lib().danet_part_clean_backward(ptr(g24.permute(0, 2, 3, 1)), ptr(pred.permute(0, 2, 3, 1)), ptr(k), B, H, W, _cpj(C),
                                                   ptr(gp.permute(0, 2, 3, 1)), stream()), 'danet_part_clean_backward')
        return gp, None


def part_clean(pred, keep=None):
    """pred [B,504,H,W] (or its [B,24,3,7,H,W] view), keep [B,24,7] or None ->
    (part_iuv_map [B,24,3,7,H,W] bf16 view, x24 [B*24,24,H,W] bf16 channels_last: channels 21..23 are zero)."""
    if pred.dim() == 6:
        B, J, T, K, H, W = pred.shape
        pred = getattr(pred, '_padded', None) if getattr(pred, '_padded', None) is not None else pred.reshape(B, J * T * K, H, W)
    x24 = PartCleanFunction.apply(pred, keep)
    return padded_part_view(x24), x24


def padded_part_view(x24):
    """x24 [B*24, 24, H, W] (channels 21..23 zero) -> its [B,24,3,7,H,W] strided view (no copy)."""
    BJ, _, H, W = x24.shape
    return x24[:, :21].reshape(BJ // NJ, NJ, 3, NC, H, W)


class PartLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, iuv_img, theta, sample_w, sel, align, scales=None):
        pred = nhwc_bf16(pred)
        B, C, H, W = pred.shape
        img = iuv_img.detach().to(torch.float32).contiguous()
        th = theta.detach().to(torch.float32).contiguous()
        w = None if sample_w is None else sample_w.detach().to(torch.float32).contiguous()
        sel = sel.to(torch.int32).contiguous()
        ctx.cpj = _cpj(C)
        if img.shape != (B, 3, H, W) or th.shape != (B, NJ, 2, 3) or sel.shape != (NJ, 6):
            raise ValueError('part_losses: bad shapes %s %s %s' % (tuple(img.shape), tuple(th.shape), tuple(sel.shape)))
        sums = ARENA.alloc(32 * 3 * 2)              # [32][3] doubles: exact, order-independent adds of the workgroups' partial sums
        if sums is None:
            sums = torch.zeros(32 * 3 * 2, dtype=torch.float32, device=pred.device)
        check(_lib.lib().danet_part_loss_forward(ptr(pred.permute(0, 2, 3, 1)), ptr(img), ptr(th), ptr(w), ptr(sel), B, H, W, int(align), ctx.cpj,
                                                 ptr(sums), stream()), 'danet_part_loss_forward')
        ctx.save_for_backward(pred, img, th, w, sel)
        ctx.align = int(align)
        ctx.scales = scales
        if scales is not None:               # the three finished losses, one launch (glue.loss_finalize; see iuv_ops.IuvGlobalFunction)
            from .glue import loss_finalize
            out = loss_finalize(3, scales, w, B, sums=sums, rows=32)
            ctx.set_materialize_grads(False)
            return out[0:1], out[1:2], out[2:3]
        return sums.view(torch.float64).view(32, 3).sum(dim=0, dtype=torch.float64).float()

    @staticmethod
    def backward(ctx, *grads):
        pred, img, th, w, sel = ctx.saved_tensors
        B, C, H, W = pred.shape
        gp = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=pred.device).permute(0, 3, 1, 2)
        if ctx.scales is not None:
            from .glue import loss_finalize
            if all(g is None for g in grads):
                return None, None, None, None, None, None, None
            scale = loss_finalize(3, ctx.scales, w, B, grads=list(grads))
        else:
            scale = grads[0].detach().to(torch.float32).contiguous()
        check(_lib.lib().danet_part_loss_backward(ptr(pred.permute(0, 2, 3, 1)), ptr(img), ptr(th), ptr(w), ptr(sel), ptr(scale),
                                                  B, H, W, ctx.align, ctx.cpj, ptr(gp.permute(0, 2, 3, 1)), stream()), 'danet_part_loss_backward')
        return gp, None, None, None, None, None, None


def part_losses(pred, iuv_img, theta, sample_w, sel, align, scales=None):
    """-> tensor [3]: sum over (b, joint, class, pixel) of fg * smooth_l1(U), same for V, and the sum over
    (b, joint, pixel) of w_b * cross-entropy of the index map; ground truth = the 3-channel IUV image
    resampled per joint by `theta` [B,24,2,3] (sel [24,6]: DensePose parts of each joint).
    scales = ((a, b),) * 3: the three FINISHED losses sums_i * a_i / (max(sum w, 1) * b_i) as a tuple of one-element tensors instead."""
    if pred.dim() == 6:
        B, J, T, K, H, W = pred.shape
        pred = getattr(pred, '_padded', None) if getattr(pred, '_padded', None) is not None else pred.reshape(B, J * T * K, H, W)
    return PartLossFunction.apply(pred, iuv_img, theta, sample_w, sel, align, scales)


class PartJointFunction(torch.autograd.Function):
    """part_clean AND part_losses of one prediction as ONE autograd node (round 6): the prediction has two consumers -- the three losses and
    the regressor's cleaned operand -- and autograd summed their gradients with an add over three 151 MB tensors after two separate
    backward kernels; here the backward is one launch (danet_part_backward_fused) that reads the prediction once and writes its gradient once.
    Outputs: (x24, loss_pU, loss_pV, loss_pIndexUV) -- the finished losses (`scales`, see part_losses)."""

    @staticmethod
    def forward(ctx, pred, keep, iuv_img, theta, sample_w, sel, align, scales):
        from .glue import loss_finalize
        pred = nhwc_bf16(pred)
        B, C, H, W = pred.shape
        cpj = _cpj(C)
        L = _lib.lib()
        k = None if keep is None else keep.detach().to(torch.float32).contiguous()
        x24 = torch.empty(B * NJ, H, W, 24, dtype=torch.bfloat16, device=pred.device).permute(0, 3, 1, 2)
        check(L.danet_part_clean_forward(ptr(pred.permute(0, 2, 3, 1)), ptr(k), B, H, W, cpj, ptr(x24.permute(0, 2, 3, 1)), stream()), 'danet_part_clean_forward')
        img = iuv_img.detach().to(torch.float32).contiguous()
        th = theta.detach().to(torch.float32).contiguous()
        w = None if sample_w is None else sample_w.detach().to(torch.float32).contiguous()
        sel = sel.to(torch.int32).contiguous()
        if img.shape != (B, 3, H, W) or th.shape != (B, NJ, 2, 3) or sel.shape != (NJ, 6):
            raise ValueError('part_joint: bad shapes %s %s %s' % (tuple(img.shape), tuple(th.shape), tuple(sel.shape)))
        sums = ARENA.alloc(32 * 3 * 2)
        if sums is None:
            sums = torch.zeros(32 * 3 * 2, dtype=torch.float32, device=pred.device)
        check(L.danet_part_loss_forward(ptr(pred.permute(0, 2, 3, 1)), ptr(img), ptr(th), ptr(w), ptr(sel), B, H, W, int(align), cpj, ptr(sums), stream()),
              'danet_part_loss_forward')
        out = loss_finalize(3, scales, w, B, sums=sums, rows=32)
        ctx.save_for_backward(pred, k, img, th, w, sel)
        ctx.align, ctx.scales, ctx.cpj = int(align), scales, cpj
        ctx.set_materialize_grads(False)
        return x24, out[0:1], out[1:2], out[2:3]

    @staticmethod
    def backward(ctx, g24, *gl):
        from .glue import loss_finalize
        pred, k, img, th, w, sel = ctx.saved_tensors
        B, C, H, W = pred.shape
        L = _lib.lib()
        none = (None,) * 8
        have_loss = any(g is not None for g in gl)
        if g24 is None and not have_loss:
            return none
        gp = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=pred.device).permute(0, 3, 1, 2)
        if g24 is not None:
            g24 = nhwc_bf16(g24)
        if not have_loss:
            check(L.danet_part_clean_backward(ptr(g24.permute(0, 2, 3, 1)), ptr(pred.permute(0, 2, 3, 1)), ptr(k), B, H, W, ctx.cpj,
                                              ptr(gp.permute(0, 2, 3, 1)), stream()), 'danet_part_clean_backward')
            return (gp,) + none[1:]
        scale = loss_finalize(3, ctx.scales, w, B, grads=list(gl))
        if g24 is None or ctx.cpj != 24:
            check(L.danet_part_loss_backward(ptr(pred.permute(0, 2, 3, 1)), ptr(img), ptr(th), ptr(w), ptr(sel), ptr(scale),
                                             B, H, W, ctx.align, ctx.cpj, ptr(gp.permute(0, 2, 3, 1)), stream()), 'danet_part_loss_backward')
            if g24 is not None:              # (the unpadded layout: two kernels and an add, as autograd would)
                gc = torch.empty_like(gp)
                check(L.danet_part_clean_backward(ptr(g24.permute(0, 2, 3, 1)), ptr(pred.permute(0, 2, 3, 1)), ptr(k), B, H, W, ctx.cpj,
                                                  ptr(gc.permute(0, 2, 3, 1)), stream()), 'danet_part_clean_backward')
                gp = gp + gc
            return (gp,) + none[1:]
        check(L.danet_part_backward_fused(ptr(pred.permute(0, 2, 3, 1)), ptr(img), ptr(th), ptr(w), ptr(sel), ptr(scale),
                                          ptr(g24.permute(0, 2, 3, 1)), ptr(k), B, H, W, ctx.align, ctx.cpj, ptr(gp.permute(0, 2, 3, 1)), stream()),
              'danet_part_backward_fused')
        return (gp,) + none[1:]


def part_joint(pred, keep, iuv_img, theta, sample_w, sel, align, scales):
    """-> (x24, lU, lV, lI): part_clean(pred, keep)[1] and part_losses(..., scales=scales) as one autograd node."""
    if pred.dim() == 6:
        B, J, T, K, H, W = pred.shape
        pred = getattr(pred, '_padded', None) if getattr(pred, '_padded', None) is not None else pred.reshape(B, J * T * K, H, W)
    return PartJointFunction.apply(pred, keep, iuv_img, theta, sample_w, sel, align, scales)
