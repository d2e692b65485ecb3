"""Global (25-class) IUV glue on HIP kernels (csrc/iuv_ops.hip): the estimator's `iuv_img2map` + `body_uv_losses`
(/root/reference/models/danet/iuv_estimator.py:95-104,304-341, utils/iuvmap.py:103-147) and DaNet.forward's part drop +
`iuvmap_clean` + concat (models/danet/danet.py:194-205,247, utils/iuvmap.py:6-38) as ONE op per pass, and the
soft-argmax of the joint heat-maps (utils/keypoints.py:334-394).  GPU only; the tensor-op forms in iuvmap.py /
geometry.py remain as the CPU-checkable statement of the same arithmetic (tests pin both against the reference's
golden vectors)."""
import torch

from . import _lib
from ._lib import ptr, check, stream

NP, NA, MAPC = 25, 15, 80
PADDED_BASES = bool(int(__import__('os').environ.get('DANET_IUV_PADDED_BASES', '1')))       # A-B knob


def _rows(t, valid, ld):
    """[B,C,H,W] fp32 head output -> (tensor whose memory is [B*H*W][ld] floats with the `valid` channels first, ld).
    The conv epilogue's zero-padded NHWC output qualifies as it is (a view); anything else is copied into that form."""
    B, C, H, W = t.shape
    if t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(3) == ld and t.stride(2) == W * ld and t.stride(0) == H * W * ld:
        return t
    buf = torch.zeros(B, H, W, ld, dtype=torch.float32, device=t.device)
    buf[..., :valid] = t[:, :valid].permute(0, 2, 3, 1)
    return buf.permute(0, 3, 1, 2)[:, :valid]


class IuvGlobalFunction(torch.autograd.Function):
    """(u, v, index, ann, gt_img | None, w | None, keep25 | None) -> (sums[4], iuv_map [B,80,H,W] bf16 channels_last,
    argmax [B,H,W] uint8 of the raw index logits)."""

    @staticmethod
    def forward(ctx, u, v, ix, an, gt, w, keep, scales=None):
        """scales (round 6) = four (a, b) pairs: the op returns the four FINISHED losses sum_i * a_i / (max(sum w, 1) * b_i) (b_i = 0: no
        division) as separate one-element tensors instead of the raw sum vector -- one launch (glue.loss_finalize) where the selects,
        multiplications / divisions, their backward and the select-backward fills cost ~6 launches per loss and pass."""
        L = _lib.lib()
        B, _, H, W = u.shape
        # an input that IS the conv epilogue's padded output (all 32 / 16 channels: iuv_global hands those over when it can) gets
        # its gradient back at that width -- no slice-backward (a fill + a copy of the padded tensor per head) in between
        ctx.full = (u.shape[1] == 32, v.shape[1] == 32, ix.shape[1] == 32, an.shape[1] == 16)
        u, v, ix = _rows(u, NP, 32), _rows(v, NP, 32), _rows(ix, NP, 32)
        an = _rows(an, NA, 16)
        want = gt is not None
        gtc = None if gt is None else gt.detach().to(torch.float32).contiguous()
        wc = None if w is None else w.detach().to(torch.float32).contiguous()
        kc = None if keep is None else keep.detach().to(torch.float32).contiguous()
        dev = u.device
        mp = torch.empty(B, H, W, MAPC, dtype=torch.bfloat16, device=dev)
        am_raw = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        am_drop = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        sums = torch.zeros(4, dtype=torch.float64, device=dev)         # double accumulators: exact, order-independent adds of the workgroups' partial sums
        check(L.danet_iuv_global_forward(u.data_ptr(), v.data_ptr(), ix.data_ptr(), an.data_ptr(), 32, 16, ptr(gtc), ptr(wc), ptr(kc),
                                         B, H, W, int(want), ptr(mp), ptr(am_raw), ptr(am_drop), ptr(sums), stream()), 'danet_iuv_global_forward')
        ctx.save_for_backward(u, v, ix, an, gtc, wc, kc, am_drop)
        ctx.want = want
        ctx.scales = scales
        ctx.mark_non_differentiable(am_raw)
        if scales is not None:
            from .glue import loss_finalize
            out = loss_finalize(4, scales, wc, B, sums=sums, rows=1)
            ctx.set_materialize_grads(False)
            return out[0:1], out[1:2], out[2:3], out[3:4], mp.permute(0, 3, 1, 2), am_raw
        return sums.float(), mp.permute(0, 3, 1, 2), am_raw

    @staticmethod
    def backward(ctx, *grads):
        L = _lib.lib()
        u, v, ix, an, gtc, wc, kc, am_drop = ctx.saved_tensors
        B, _, H, W = u.shape
        dev = u.device
        if ctx.scales is not None:
            g4, gmap = grads[:4], grads[4]
            gsums = None
            if ctx.want and any(g is not None for g in g4):
                from .glue import loss_finalize
                gsums = loss_finalize(4, ctx.scales, wc, B, grads=list(g4))
        else:
            gsums, gmap = grads[0], grads[1]
        du = torch.empty(B, H, W, 32, dtype=torch.float32, device=dev)
        dv, di = torch.empty_like(du), torch.empty_like(du)
        da = torch.empty(B, H, W, 16, dtype=torch.float32, device=dev)
        coef = None
        if ctx.want:
            coef = torch.zeros(4, dtype=torch.float32, device=dev) if gsums is None else gsums.to(torch.float32).contiguous()
        gm = None
        if gmap is not None:
            gm = gmap.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        check(L.danet_iuv_global_backward(u.data_ptr(), v.data_ptr(), ix.data_ptr(), an.data_ptr(), 32, 16, ptr(gtc), ptr(wc), ptr(kc),
                                          ptr(am_drop), ptr(gm), ptr(coef), B, H, W, int(ctx.want and coef is not None),
                                          ptr(du), ptr(dv), ptr(di), ptr(da), stream()), 'danet_iuv_global_backward')
        f = lambda t, n, full: t.permute(0, 3, 1, 2) if full else t.permute(0, 3, 1, 2)[:, :n]        # noqa: E731
        fu = ctx.full
        return f(du, NP, fu[0]), f(dv, NP, fu[1]), f(di, NP, fu[2]), f(da, NA, fu[3]), None, None, None, None


def iuv_global(u, v, ix, an, gt=None, w=None, keep=None, scales=None):
    """sums = (sum smooth-L1 U, sum smooth-L1 V, sum CE index, sum CE ann) over the batch, weighted per sample by w
    (zeros when gt is None); iuv_map = [U_clean | V_clean | one-hot | 5 zero channels] as a bf16 channels_last
    [B,80,H,W] tensor (the body regressor's padded first-conv operand); argmax = uint8 [B,H,W] of the raw index head.
    scales = ((a, b),) * 4: the first result is the tuple of the four finished losses sums_i * a_i / (max(sum w, 1) * b_i) instead."""
    if not u.is_cuda:
        raise RuntimeError('danet_hip ops run on the GPU only (got a %s tensor); there is no CPU path' % u.device)
    if PADDED_BASES and torch.is_grad_enabled():
        # head outputs that are [:, :25] / [:, :15] views of the conv epilogue's zero-padded fp32 NHWC output carry that tensor along
        # (conv.conv2d `_padded_base`): taking IT as the differentiable input keeps autograd's slice-backward out of the backward pass
        bases = [getattr(t, '_padded_base', None) for t in (u, v, ix, an)]
        ok = all(b is not None and b.dtype == torch.float32 and b.shape[0] == t.shape[0] and b.shape[2:] == t.shape[2:] and
                 b.data_ptr() == t.data_ptr() and b.shape[1] == ld for b, t, ld in zip(bases, (u, v, ix, an), (32, 32, 32, 16)))
        if ok and all(_rows(b, n, ld) is b for b, n, ld in zip(bases, (NP, NP, NP, NA), (32, 32, 32, 16))):
            return _pack(IuvGlobalFunction.apply(bases[0], bases[1], bases[2], bases[3], gt, w, keep, scales), scales)
    return _pack(IuvGlobalFunction.apply(u, v, ix, an, gt, w, keep, scales), scales)


def _pack(out, scales):
    """(sums, map, argmax), with `scales` the four finished losses as a tuple in place of the sum vector."""
    return out if scales is None else (tuple(out[:4]), out[4], out[5])


class SoftArgmaxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hm, scale):
        L = _lib.lib()
        B, J, H, W = hm.shape
        ld = (J + 3) // 4 * 4
        rows = _rows(hm, J, ld)
        out = torch.empty(B, J, 2, dtype=torch.float32, device=hm.device)
        saved = torch.empty(B, J, 4, dtype=torch.float32, device=hm.device)
        check(L.danet_softargmax_forward(rows.data_ptr(), ld, B, J, H, W, float(scale), ptr(out), ptr(saved), stream()), 'danet_softargmax_forward')
        ctx.save_for_backward(rows, saved)
        ctx.cfg = (ld, float(scale))
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        rows, saved = ctx.saved_tensors
        B, J, H, W = rows.shape
        ld, scale = ctx.cfg
        d = torch.empty(B, H, W, J, dtype=torch.float32, device=rows.device)
        check(L.danet_softargmax_backward(rows.data_ptr(), ld, B, J, H, W, scale, ptr(saved), ptr(g.to(torch.float32).contiguous()), ptr(d), stream()),
              'danet_softargmax_backward')
        return d.permute(0, 3, 1, 2), None


def softargmax(hm, scale=1.0):
    """Expected (x, y) pixel index of softmax(scale * hm) over each joint's map: [B,J,H,W] -> [B,J,2]
    (= softmax_integral_tensor(scale * hm, J, H, W) of geometry.py)."""
    if not hm.is_cuda:
        raise RuntimeError('danet_hip ops run on the GPU only (got a %s tensor); there is no CPU path' % hm.device)
    return SoftArgmaxFunction.apply(hm, scale)
