"""Layer modules on the HIP kernels, with nn.Module parameter/buffer names identical to the
reference's torch modules (state-dict compatible)."""
import ctypes

import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import conv as _conv
from ._lib import ptr, check, stream
from .conv import Conv2d, conv2d, nhwc_bf16, nhwc_as, nhwc_act, _empty_nhwc, ARENA  # noqa: F401


def _k(L, name, dtype):
    """The entry point `name` for activations of `dtype`: the fp32 instantiation (csrc/norm_act_f32.hip, stn.hip) or the bf16 one."""
    return getattr(L, name + '_f32' if dtype == torch.float32 else name)


RELU_MASK = bool(int(os.environ.get('DANET_BN_RELU_MASK', '1')))     # A/B knob: 0 = the backward gates on the saved output y


# One-pass BatchNorm backward (csrc/norm_act.hip bn_bwd_onepass_kernel: dy and x stay in registers across a grid-wide
# barrier).  Its launches must not overlap each other, so it is only used on ONE stream: the default stream, or the one a
# trainer names in ONEPASS_STREAM (its capture stream); BatchNorms running on other (side) streams take the two-kernel path.
ONEPASS = bool(int(os.environ.get('DANET_BN_ONEPASS', '1')))
ONEPASS_STREAM = None
# Co-residency budget of a one-pass launch (workgroups; 0 = the whole device, two per compute unit; < 0 = no one-pass launches).  A
# data-parallel trainer lowers it, for the duration of its steps, to 2 * (compute units - communication channels): the all-reduce kernels of the step's earlier gradient buckets run
# on the communication stream WHILE the backward pass continues, and a grid barrier over more workgroups than fit beside them
# could wait for ever (include/danet_hip.h, danet_bn_backward_onepass).
ONEPASS_MAX_BLOCKS = 0
# > 0 while another stream of the step may be running kernels beside the one-pass stream (the regressor's body_net branch between its fork
# and its join, in the forward and in the backward pass: smpl_regressor.SideWindow).  A one-pass launch of more than ONE workgroup per
# compute unit is then not safe: each workgroup takes 78 KB of a compute unit's 160 KB of LDS, and a block of LDS that a workgroup of the
# other stream held while the first one-pass workgroup was placed can leave the remaining space in two pieces that are each too small
# for the second -- for as long as the first one spins at the barrier, i.e. until the barrier's spin bound (round 6: a 492-workgroup
# launch of limb_net's backward beside a 13 us pointwise convolution of body_net's, 0.6 s, `bn_bwd_onepass_kernel` error word 0x301ec;
# caught by bench.py's host-ahead eager step, where the two branches meet at a different phase than in the replayed graph).  While the
# window is open the budget is one workgroup per compute unit; launches that need more take the two-kernel path (four BatchNorms of
# limb_net's first layer).
SIDE_LIVE = 0
SIDE_CAP = bool(int(os.environ.get('DANET_SIDE_CAP', '1')))      # 0: A-B timing of the cap ONLY (the hazard above is real)
_ONEPASS_BAR = {}
_CUS = {}


def onepass_budget(device=None):
    """The co-residency budget a one-pass launch may use NOW (the `max_blocks` of danet_bn_backward_onepass): ONEPASS_MAX_BLOCKS, capped
    at one workgroup per compute unit while a side stream is live."""
    b = ONEPASS_MAX_BLOCKS
    if SIDE_LIVE > 0 and b >= 0 and SIDE_CAP:
        dev = torch.cuda.current_device() if device is None or getattr(device, 'index', None) is None else device.index
        cus = _CUS.get(dev)
        if cus is None:
            cus = _CUS[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
        b = cus if b == 0 else min(b, cus)
    return b


def _onepass_bar(device):
    """The barrier state of the one-pass launches on `device` (zeroed once), or None when the current stream is
    not the one these launches are confined to."""
    if not ONEPASS or ONEPASS_MAX_BLOCKS < 0:
        return None
    cur = torch.cuda.current_stream(device)
    if cur != (ONEPASS_STREAM if ONEPASS_STREAM is not None else torch.cuda.default_stream(device)):
        return None
    return _onepass_state(device)


def _onepass_state(device):
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    bar = _ONEPASS_BAR.get(device)
    if bar is None:
        bar = _ONEPASS_BAR[device] = torch.zeros(_lib.lib().danet_bn_backward_onepass_bar_words(), dtype=torch.int32, device=device)
    return bar


def onepass_poison(device):
    """The barrier's error word as a one-element int32 tensor: what optim.FusedAdam hands the Adam kernel as `poison`, so that a
    step whose one-pass launch timed out is never applied."""
    return _onepass_state(device)[2:3]


_WATCH = {}


def onepass_watch(device):
    """For barrier launches OUTSIDE a Trainer (which checks the error word itself and keeps a timed-out step from being applied): a
    non-blocking look at the error word.  Every call starts an asynchronous copy of the word into pinned memory and inspects the copy the
    PREVIOUS call started, once its event has passed -- a launch that gave up at its barrier (its results are garbage, and the stale arrival
    counts make later barriers release early) is reported by the next barrier launch or by `onepass_error()`, not silently.  Nothing is
    done while the stream is capturing."""
    if ONEPASS_STREAM is not None or torch.cuda.is_current_stream_capturing():
        return
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _WATCH.get(idx)
    if st is not None and st[1].query():
        code = int(st[0][0])
        if code != 0:
            _WATCH.pop(idx, None)
            raise RuntimeError('a grid-barrier kernel gave up waiting (error word 0x%x: csrc/grid_barrier.h); results since then are invalid -- '
                               'nn.onepass_recover() resets the barrier state' % code)
    if st is None:
        st = _WATCH[idx] = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    elif not st[1].query():
        return                                   # the previous look is still in flight
    st[0].copy_(_onepass_state(device)[2:3], non_blocking=True)
    st[1].record(torch.cuda.current_stream(device))


def _on_device(d, device):
    """d (a key of _ONEPASS_BAR, always with an index) is `device` (None = any; 'cuda' without an index = any GPU)."""
    if device is None:
        return True
    device = torch.device(device)
    return d.type == device.type and (device.index is None or d.index == device.index)


def onepass_error(device=None):
    """True if a one-pass launch gave up waiting at its barrier (its results are garbage)."""
    return any(int(b[2]) != 0 for d, b in _ONEPASS_BAR.items() if _on_device(d, device))


def onepass_recover(device=None, force=False):
    """After a barrier time-out (onepass_error): the arrival counts of the expired barrier are stale, so every later one-pass
    launch would time out as well.  Zeroes the barrier state and switches the one-pass backward off (the two-kernel path
    takes over); returns True when there was an error to recover from.  Steps computed since the time-out are garbage:
    the caller decides what to redo (trainer.Trainer re-captures its graph and raises).  force: another data-parallel rank
    reported the time-out (the all-reduced poison word): switch this rank over as well, so that all replicas keep running
    the same kernels."""
    global ONEPASS
    if not (onepass_error(device) or force):
        return False
    torch.cuda.synchronize()
    for d, b in _ONEPASS_BAR.items():
        if _on_device(d, device):
            b.zero_()
    ONEPASS = False
    return True


class BatchNormActFunction(torch.autograd.Function):
    """y = [relu](batch_norm(x) [+ res]) on NHWC bf16; training or eval statistics."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, training, momentum, eps, relu, fused_sums=None, link=None):
        L = _lib.lib()
        x = nhwc_act(x)
        dt = x.dtype
        B, C, H, W = x.shape
        M = B * H * W
        if res is not None:
            res = nhwc_as(res, dt)
            if res.shape != x.shape:
                raise ValueError('residual shape %s != %s' % (tuple(res.shape), tuple(x.shape)))
        y = _empty_nhwc(B, C, H, W, dt, x.device)
        saved = torch.empty(2, C, dtype=torch.float32, device=x.device) if training else None
        sums, sums_zero = None, False
        if training and fused_sums is not None:
            sums, sums_zero = fused_sums, 2                   # accumulated by the producing conv's epilogue
        elif training:
            sums = ARENA.alloc(L.danet_bn_ws_floats(C))
            sums_zero = sums is not None
            if sums is None:
                sums = torch.empty(L.danet_bn_ws_floats(C), dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.detach().float().contiguous()
        b = None if beta is None else beta.detach().float().contiguous()
        # ReLU gate for the backward without re-reading y (csrc/norm_act.hip ldmask): a byte mask with a residual,
        # recomputed from x without one
        # (without a residual the BatchNorm's own backward recomputes the gate; the mask then only serves conv._bn_gate)
        mask = torch.empty(M * C // 4, dtype=torch.uint8, device=x.device) \
            if (training and relu and RELU_MASK and (res is not None or _conv.FUSE_BN_BWD_REDUCE)) else None
        check(_k(L, 'danet_bn_forward', dt)(ptr(x.permute(0, 2, 3, 1)), None if res is None else ptr(res.permute(0, 2, 3, 1)),
                                 ptr(y.permute(0, 2, 3, 1)), M, C, ptr(g), ptr(b), ptr(running_mean), ptr(running_var),
                                 ptr(saved), ptr(sums), int(sums_zero), float(momentum), float(eps), int(training), int(relu),
                                 ptr(mask), stream()),
              'danet_bn_forward')
        if training:
            # the BatchNorm's own backward recomputes the gate from x when it can (no residual: no mask read); the mask then
            # only serves the consumer conv's fused reduction (conv._bn_gate)
            ctx.mask_mode = (2 if res is None else (1 if mask is not None else 0)) if RELU_MASK else 0
            ctx.save_for_backward(x, y if (relu and ctx.mask_mode == 0) else None, g, saved, b, mask)
            ctx.relu = relu
            ctx.has_res = res is not None
            ctx.link = link
            ctx.has_affine = gamma is not None
        ctx.training = training
        if _conv.TRACE is not None:
            _conv.TRACE.append(('bn', tuple(x.shape), y.float().abs().mean()))
        if training:
            # lets a consumer conv fuse this BN's backward reduction (no reference to y: no cycle)
            y._bn_ctx = (x, bool(relu), saved, mask, ctx.mask_mode)
        return y

    @staticmethod
    def backward(ctx, gy):
        if not ctx.training:
            raise RuntimeError('BatchNorm backward in eval mode is not on the hot path')
        L = _lib.lib()
        x, y, g, saved, b, mask = ctx.saved_tensors
        gy_in = gy
        dt = x.dtype
        gy = nhwc_as(gy, dt)
        B, C, H, W = x.shape
        M = B * H * W
        dx = _empty_nhwc(B, C, H, W, dt, x.device)
        dres = _empty_nhwc(B, C, H, W, dt, x.device) if ctx.has_res else None
        red = getattr(gy_in, '_bn_red', None)        # reduced by the consumer conv's data-gradient epilogue (conv.py)
        _conv.FUSION['bn_bwd_reduce_fused' if red is not None else 'bn_bwd_reduce_own'] += 1
        if red is not None:
            red_zero = 2
        else:
            red = ARENA.alloc(L.danet_bn_ws_floats(C))
            red_zero = red is not None
            if red is None:
                red = torch.empty(L.danet_bn_ws_floats(C), dtype=torch.float32, device=x.device)
        dparam = torch.empty(2, C, dtype=torch.float32, device=x.device)      # rows: d beta, d gamma
        bar = _onepass_bar(x.device) if (red_zero != 2 and (C <= 1024 or (C % 1024 == 0 and C <= 12288)) and dt == torch.bfloat16) else None
        done = False
        if bar is not None:
            if red_zero is False:
                red.zero_()
            job = (_lib.BnBwdJob * 1)()
            j = job[0]
            j.dy, j.x, j.y = gy.data_ptr(), x.data_ptr(), None if y is None else y.data_ptr()
            j.gamma, j.saved = None if g is None else g.data_ptr(), saved.data_ptr()
            j.dx, j.dres, j.dparam, j.red = dx.data_ptr(), None if dres is None else dres.data_ptr(), dparam.data_ptr(), red.data_ptr()
            j.beta, j.mask, j.mask_mode = None if b is None else b.data_ptr(), None if mask is None else mask.data_ptr(), int(ctx.mask_mode)
            j.M, j.C, j.red_state, j.relu = M, C, 1, int(ctx.relu)
            budget = onepass_budget(x.device)
            if L.danet_bn_backward_onepass_ok(ctypes.addressof(job), 1, budget):
                check(L.danet_bn_backward_onepass(ctypes.addressof(job), 1, ptr(bar), budget, stream()), 'danet_bn_backward_onepass')
                _conv.FUSION['bn_bwd_onepass'] += 1
                done = True
        if not done:
            check(_k(L, 'danet_bn_backward', dt)(ptr(gy.permute(0, 2, 3, 1)), ptr(x.permute(0, 2, 3, 1)),
                                  None if y is None else ptr(y.permute(0, 2, 3, 1)), M, C, ptr(g), ptr(saved),
                                  int(ctx.relu), ptr(dx.permute(0, 2, 3, 1)),
                                  None if dres is None else ptr(dres.permute(0, 2, 3, 1)), ptr(dparam), ptr(red), int(red_zero),
                                  int(ctx.mask_mode), ptr(mask), ptr(b), stream()),
              'danet_bn_backward')
        if _conv.TRACE is not None:
            _conv.TRACE.append(('bn_bwd' + ('+red' if red_zero == 2 else ''), tuple(x.shape), dx.float().abs().mean()))
        # unbind gives two independent-looking tensors that AccumulateGrad can keep without a clone
        dbeta, dgamma = (dparam[0], dparam[1]) if ctx.has_affine else (None, None)
        link = getattr(ctx, 'link', None)
        if link is not None and link.armed and dres is not None:
            link.dres, dres = dres, None          # the block's first convolution adds it in its data-gradient epilogue (conv.ResLink)
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters / buffers) with optional fused residual add and ReLU."""

    # per-module `num_batches_tracked += 1` is one tiny launch per layer (353 per step); a trainer may
    # switch it off and bump all counters with one multi-tensor op (bump_batch_counters)
    count_batches = True
    _ran = []                 # counters of the modules that ran in training mode while count_batches was off

    def _count(self):
        if self.track_running_stats and self.num_batches_tracked is not None:
            if BatchNorm2d.count_batches:
                self.num_batches_tracked.add_(1)
            else:
                BatchNorm2d._ran.append(self.num_batches_tracked)

    def forward(self, x, res=None, relu=False, link=None):
        training = self.training or not self.track_running_stats
        if training:
            self._count()
        momentum = 0.1 if self.momentum is None else self.momentum
        fused = getattr(x, '_bn_sums', None) if training else None
        if training:
            _conv.FUSION['bn_stats_fused' if fused is not None else 'bn_stats_own'] += 1
        return BatchNormActFunction.apply(x, res, self.weight, self.bias,
                                          self.running_mean if self.track_running_stats else None,
                                          self.running_var if self.track_running_stats else None,
                                          training, momentum, self.eps, relu, fused, link if training else None)


def _bn_forward_padded(self, x, d, relu=False, res=None, padded=None):
    """BatchNorm2d on a tensor that carries d extra, exactly-zero channels behind this module's num_features (resnet.Bottleneck.
    _forward_padded): gamma / beta are padded with zeros, so the extra channels come out as zeros again; the running statistics
    live in padded storage of which `running_mean` / `running_var` are views (state_dict keeps the reference's shapes)."""
    C = self.num_features
    training = self.training or not self.track_running_stats
    if training:
        self._count()
    rm = rv = None
    if self.track_running_stats:
        pad = getattr(self, '_stat_pad', None)
        if pad is None or pad[0].shape[0] != C + d or pad[0].device != x.device or self.running_mean.data_ptr() != pad[0].data_ptr():
            rm, rv = torch.zeros(C + d, device=x.device), torch.ones(C + d, device=x.device)
            rm[:C].copy_(self.running_mean); rv[:C].copy_(self.running_var)
            self._stat_pad = (rm, rv)
            self._buffers['running_mean'], self._buffers['running_var'] = rm[:C], rv[:C]        # views: updated in place by the kernel
        rm, rv = self._stat_pad
    momentum = 0.1 if self.momentum is None else self.momentum
    fused = getattr(x, '_bn_sums', None) if training else None
    if training:
        _conv.FUSION['bn_stats_fused' if fused is not None else 'bn_stats_own'] += 1
    gamma, beta = padded if padded is not None else (F.pad(self.weight, (0, d)), F.pad(self.bias, (0, d)))     # (padded: the caller's glue.pad_multi copies)
    return BatchNormActFunction.apply(x, res, gamma, beta, rm, rv, training, momentum, self.eps, relu, fused, None)


BatchNorm2d.forward_padded = _bn_forward_padded


class MultiBatchNormFunction(torch.autograd.Function):
    """n (<= 4) independent training-mode BatchNorm(+residual)(+ReLU) ops in ONE launch per pass
    (csrc/norm_act.hip bn_*_multi_kernel).  Tensor arguments: xs[n], ress[n] (None allowed), gammas[n], betas[n];
    `static` carries the running buffers, flags and the conv-epilogue statistics."""

    @staticmethod
    def forward(ctx, static, *tensors):
        L = _lib.lib()
        n, relu, momentum, eps, rms, rvs, fused, links = static[:8]
        done = static[8] if len(static) > 8 else None       # (out, saved, mask) per job when the producing convolution's launch applied the BatchNorm (multi_conv_bn)
        relus = list(relu) if isinstance(relu, (list, tuple)) else [bool(relu)] * n          # per job (the kernels take one flag per job)
        xs = [nhwc_act(t) for t in tensors[:n]]
        dt = xs[0].dtype
        ress = [None if t is None else nhwc_as(t, dt) for t in tensors[n:2 * n]]
        gammas = [t.detach().float().contiguous() for t in tensors[2 * n:3 * n]]
        betas = [t.detach().float().contiguous() for t in tensors[3 * n:4 * n]]
        jobs = (_lib.BnFwdJob * n)()
        ys, saveds, keep, masks = [], [], [], []
        if done is not None:
            # nothing to launch: conv -> BatchNorm ran as one call (conv.MultiConvFunction.forward); same outputs, same saved tensors
            ys, saveds, masks = [d[0] for d in done], [d[1] for d in done], [d[2] for d in done]
            _conv.FUSION['bn_forward_in_conv_launch'] += n
        for i in range(n if done is None else 0):
            B, C, H, W = xs[i].shape
            if ress[i] is not None and ress[i].shape != xs[i].shape:
                raise ValueError('residual shape %s != %s' % (tuple(ress[i].shape), tuple(xs[i].shape)))
            y = _empty_nhwc(B, C, H, W, dt, xs[i].device)
            saved = torch.empty(2, C, dtype=torch.float32, device=xs[i].device)
            sums, state = fused[i], 2
            if sums is None:
                sums, state = ARENA.alloc(L.danet_bn_ws_floats(C)), 1
                if sums is None:
                    sums = torch.zeros(L.danet_bn_ws_floats(C), dtype=torch.float32, device=xs[i].device)
            keep.append(sums)
            mask = torch.empty(B * H * W * C // 4, dtype=torch.uint8, device=xs[i].device) \
                if (relus[i] and RELU_MASK and (ress[i] is not None or _conv.FUSE_BN_BWD_REDUCE)) else None
            masks.append(mask)
            j = jobs[i]
            j.mask = None if mask is None else mask.data_ptr()
            j.x, j.res, j.y = xs[i].data_ptr(), None if ress[i] is None else ress[i].data_ptr(), y.data_ptr()
            j.gamma, j.beta = gammas[i].data_ptr(), betas[i].data_ptr()
            j.running_mean = None if rms[i] is None else rms[i].data_ptr()
            j.running_var = None if rvs[i] is None else rvs[i].data_ptr()
            j.saved, j.sums = saved.data_ptr(), sums.data_ptr()
            j.M, j.C, j.sums_state, j.relu = B * H * W, C, state, int(relus[i])
            ys.append(y)
            saveds.append(saved)
        if done is None:
            check(_k(L, 'danet_bn_forward_multi', dt)(ctypes.addressof(jobs), n, float(momentum), float(eps), stream()), 'danet_bn_forward_multi')
        modes = [((2 if r is None else (1 if m is not None else 0)) if RELU_MASK else 0) if rl else 0 for m, r, rl in zip(masks, ress, relus)]
        ctx.save_for_backward(*xs, *[y if (rl and md == 0) else None for y, md, rl in zip(ys, modes, relus)], *gammas, *saveds, *betas, *masks)
        ctx.cfg = (n, relus, [r is not None for r in ress], links, modes)
        for y, x, saved, mask, md, rl in zip(ys, xs, saveds, masks, modes, relus):
            y._bn_ctx = (x, bool(rl), saved, mask, md)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        L = _lib.lib()
        n, relus, has_res, links, modes = ctx.cfg
        sv = ctx.saved_tensors
        xs, ys, gammas, saveds, betas, masks = sv[:n], sv[n:2 * n], sv[2 * n:3 * n], sv[3 * n:4 * n], sv[4 * n:5 * n], sv[5 * n:6 * n]
        jobs = (_lib.BnBwdJob * n)()
        dt = xs[0].dtype
        dxs, dress, dparams, keep = [], [], [], []
        for i in range(n):
            B, C, H, W = xs[i].shape
            red = getattr(gys[i], '_bn_red', None)
            _conv.FUSION['bn_bwd_reduce_fused' if red is not None else 'bn_bwd_reduce_own'] += 1
            state = 2
            if red is None:
                red, state = ARENA.alloc(L.danet_bn_ws_floats(C)), 1
                if red is None:
                    red = torch.zeros(L.danet_bn_ws_floats(C), dtype=torch.float32, device=xs[i].device)
            gy = nhwc_as(gys[i], dt)
            dx = _empty_nhwc(B, C, H, W, dt, xs[i].device)
            dres = _empty_nhwc(B, C, H, W, dt, xs[i].device) if has_res[i] else None
            dparam = torch.empty(2, C, dtype=torch.float32, device=xs[i].device)
            keep += [red, gy]
            j = jobs[i]
            j.dy, j.x, j.y = gy.data_ptr(), xs[i].data_ptr(), None if ys[i] is None else ys[i].data_ptr()
            j.beta, j.mask, j.mask_mode = betas[i].data_ptr(), None if masks[i] is None else masks[i].data_ptr(), modes[i]
            j.gamma, j.saved = gammas[i].data_ptr(), saveds[i].data_ptr()
            j.dx, j.dres, j.dparam, j.red = dx.data_ptr(), None if dres is None else dres.data_ptr(), dparam.data_ptr(), red.data_ptr()
            j.M, j.C, j.red_state, j.relu = B * H * W, C, state, int(relus[i])
            dxs.append(dx)
            dress.append(dres)
            dparams.append(dparam)
        bar = _onepass_bar(xs[0].device) if (dt == torch.bfloat16 and all(jobs[i].red_state == 1 for i in range(n))) else None
        budget = onepass_budget(xs[0].device)
        if bar is not None and L.danet_bn_backward_onepass_ok(ctypes.addressof(jobs), n, budget):
            check(L.danet_bn_backward_onepass(ctypes.addressof(jobs), n, ptr(bar), budget, stream()), 'danet_bn_backward_onepass')
            _conv.FUSION['bn_bwd_onepass'] += n
        else:
            check(_k(L, 'danet_bn_backward_multi', dt)(ctypes.addressof(jobs), n, stream()), 'danet_bn_backward_multi')
        if links is not None:
            for i, lk in enumerate(links):
                if lk is not None and lk.armed and dress[i] is not None:
                    lk.dres, dress[i] = dress[i], None      # added by the block's first convolution's data gradient (conv.ResLink)
        return (None, *dxs, *dress, *[d[1] for d in dparams], *[d[0] for d in dparams])


# conv -> BatchNorm (+ residual) (+ ReLU) of the lockstep branch layers as ONE launch (csrc/conv3x3s.hip conv3x3_stream_bn_kernel).  Built, bit-identical
# to the two launches (tests/test_gpu_conv.py::test_conv_bn_one_launch_equals_two_launches) and OFF: measured on MI355X in the benched step
# (two A-B pairs in one gpurun call) 26.42 / 26.42 ms with two launches, 26.53 / 26.52 ms with one -- the workgroups' own outputs are no
# longer in their XCD's L2 when they come back for them (tools/c3s_bn_phases.py, profiles/r05_conv_bn_phases.txt: barrier released 38 us
# into the four-branch launch, statistics + 1.8 us, the apply pass + 10 us = 47 MB at 4.7 TB/s, the rate of the stand-alone
# bn_apply_multi_kernel), so the tail costs what the second launch cost and the barrier eats the launch overhead saved.
CONV_BN = bool(int(os.environ.get('DANET_CONV_BN', '0')))


def _bn_mask_wanted(relu, has_res):
    return bool(relu and RELU_MASK and (has_res or _conv.FUSE_BN_BWD_REDUCE))


def multi_conv_bn(convs, xs, bns, ress=None, relu=False, conv_links=None, bn_links=None):
    """multi_batch_norm(bns, multi_conv(convs, xs), ress, relu) -- conv -> bn -> [+ residual] -> [relu] of the lockstep branch layers
    (/root/reference/models/module/res_module.py:39-56, hr_module.py:155-177) -- with the BatchNorm applied by the convolutions'
    own launch when the set runs on the streamed 3x3 kernel (csrc/conv3x3s.hip s3_bn_tail: a grid-wide barrier after the last
    tile, then every workgroup normalises the tiles it wrote); two launches otherwise.  Results are bit-identical either way."""
    n = len(bns)
    relus = [bool(r) for r in relu] if isinstance(relu, (list, tuple)) else [bool(relu)] * n
    rs = list(ress) if ress is not None else [None] * n
    spec = None
    dev = xs[0].device
    if (CONV_BN and 1 <= n <= 4 and xs[0].is_cuda and _conv.PRECISION != 'fp32' and _conv.FUSE_BN_STATS and torch.is_grad_enabled() and
            all(b.training and b.affine and b.num_features <= 1024 for b in bns) and
            len({0.1 if b.momentum is None else b.momentum for b in bns}) == 1 and len({b.eps for b in bns}) == 1):
        bar = _onepass_bar(dev)
        # the launch crosses a grid barrier with up to `c3s_blocks` workgroups (512): under a data-parallel trainer's co-residency
        # budget (ONEPASS_MAX_BLOCKS = 2 x the compute units left beside the communication kernels) a grid that large may not be
        # resident at once -- two launches then, like the one-pass backward and the one-launch SMPL backward (round-5 advisor)
        if bar is not None and 0 < onepass_budget(dev) < _lib.lib().knob('c3s_blocks'):
            bar = None
        if bar is not None:
            spec = {'bar': bar, 'momentum': 0.1 if bns[0].momentum is None else bns[0].momentum, 'eps': bns[0].eps,
                    'jobs': [{'res': r, 'gamma': b.weight.detach().float().contiguous(), 'beta': b.bias.detach().float().contiguous(),
                              'running_mean': b.running_mean if b.track_running_stats else None,
                              'running_var': b.running_var if b.track_running_stats else None,
                              'relu': rl, 'want_mask': _bn_mask_wanted(rl, r is not None)} for b, r, rl in zip(bns, rs, relus)]}
    h = _conv.multi_conv(convs, xs, conv_links, bn=spec)
    return multi_batch_norm(bns, h, ress, relu=relu, links=bn_links)


def multi_batch_norm(bns, xs, ress=None, relu=False, links=None):
    """[bn(x, res, relu) for bn, x, res in ...] for up to 4 training-mode BatchNorm2d modules in one launch per pass;
    falls back to the per-module path otherwise (eval mode, wide layers, more than 4)."""
    n = len(bns)
    ress = list(ress) if ress is not None else [None] * n
    relus = [bool(r) for r in relu] if isinstance(relu, (list, tuple)) else [bool(relu)] * n      # one flag per BatchNorm, or one for all
    ok = 1 <= n <= 12 and all(b.training and b.affine and b.num_features <= 1024 for b in bns) and xs[0].is_cuda
    lks = list(links) if links is not None else [None] * n
    if not ok:
        return [b(x, r, rl, link=lk) for b, x, r, rl, lk in zip(bns, xs, ress, relus, lks)]
    mom = {0.1 if b.momentum is None else b.momentum for b in bns}
    eps = {b.eps for b in bns}
    if len(mom) != 1 or len(eps) != 1:
        return [b(x, r, rl, link=lk) for b, x, r, rl, lk in zip(bns, xs, ress, relus, lks)]
    for b in bns:
        b._count()
    rms = [b.running_mean if b.track_running_stats else None for b in bns]
    rvs = [b.running_var if b.track_running_stats else None for b in bns]
    fused = [getattr(x, '_bn_sums', None) for x in xs]
    for f in fused:
        _conv.FUSION['bn_stats_fused' if f is not None else 'bn_stats_own'] += 1
    done = [getattr(x, '_bn_done', None) for x in xs]
    done = done if all(d is not None for d in done) else None
    static = (n, relus, mom.pop(), eps.pop(), rms, rvs, fused, links, done)
    return list(MultiBatchNormFunction.apply(static, *xs, *ress, *[b.weight for b in bns], *[b.bias for b in bns]))


SUM_BWD_ALL = bool(int(os.environ.get('DANET_SUM_BWD_ALL', '1')))


class SumReluFunction(torch.autograd.Function):
    """y = [relu](sum_t nearest_upsample_{2^shift_t}(term_t)) -- HRNet fuse layer (hr_module.py:166-177)."""

    @staticmethod
    def forward(ctx, relu, shifts, *terms):
        L = _lib.lib()
        terms = [nhwc_act(t) for t in terms]
        dt = terms[0].dtype
        n = len(terms)
        B, C = terms[0].shape[0], terms[0].shape[1]
        H = max(t.shape[2] << s for t, s in zip(terms, shifts))
        W = max(t.shape[3] << s for t, s in zip(terms, shifts))
        for t, s in zip(terms, shifts):
            if t.shape[1] != C or (t.shape[2] << s) != H or (t.shape[3] << s) != W:
                raise ValueError('sum_relu: term %s with shift %d does not match output %dx%dx%d' % (tuple(t.shape), s, C, H, W))
        y = _empty_nhwc(B, C, H, W, dt, terms[0].device)
        ptrs = (ctypes.c_void_p * n)(*[ptr(t.permute(0, 2, 3, 1)) for t in terms])
        sh = (ctypes.c_int * n)(*shifts)
        check(_k(L, 'danet_sum_relu_forward', dt)(ptrs, sh, n, B, H, W, C, int(relu), ptr(y.permute(0, 2, 3, 1)), stream()),
              'danet_sum_relu_forward')
        ctx.save_for_backward(y if relu else None)
        ctx.cfg = (relu, tuple(shifts), B, C, H, W, dt)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        (y,) = ctx.saved_tensors
        relu, shifts, B, C, H, W, dt = ctx.cfg
        gy = nhwc_as(gy, dt)
        outs = []
        cache = {}
        need = sorted({s for i, s in enumerate(shifts) if ctx.needs_input_grad[2 + i]})
        if SUM_BWD_ALL and len(need) > 1 and need[-1] <= 3:
            # every shift in one launch: gy and y are read once (csrc/norm_act.hip sum_relu_bwd_all_kernel)
            for s in need:
                cache[s] = _empty_nhwc(B, C, H >> s, W >> s, dt, gy.device)
            dp = [None if s not in cache else ptr(cache[s].permute(0, 2, 3, 1)) for s in range(4)]
            check(_k(L, 'danet_sum_relu_backward_all', dt)(ptr(gy.permute(0, 2, 3, 1)), None if y is None else ptr(y.permute(0, 2, 3, 1)),
                                                B, H, W, C, int(relu), dp[0], dp[1], dp[2], dp[3], stream()),
                  'danet_sum_relu_backward_all')
        for i, s in enumerate(shifts):
            if not ctx.needs_input_grad[2 + i]:
                outs.append(None)
                continue
            if s not in cache:
                d = _empty_nhwc(B, C, H >> s, W >> s, dt, gy.device)
                check(_k(L, 'danet_sum_relu_backward', dt)(ptr(gy.permute(0, 2, 3, 1)), None if y is None else ptr(y.permute(0, 2, 3, 1)),
                                                B, H, W, C, s, int(relu), ptr(d.permute(0, 2, 3, 1)), stream()),
                      'danet_sum_relu_backward')
                cache[s] = d
            outs.append(cache[s])
        return (None, None) + tuple(outs)


class SumReluMultiFunction(torch.autograd.Function):
    """n (<= 4) fuse sums -- the outputs of ONE HighResolutionModule (hr_module.py:166-177) -- in one launch forward
    (csrc/norm_act.hip sum_relu_multi_kernel) and one backward (sum_relu_bwd_all_multi_kernel): the low-resolution outputs are far too
    small to fill launches of their own.  meta = [(shifts, nterms)] per output; terms flattened behind it."""

    @staticmethod
    def forward(ctx, relu, meta, *terms):
        L = _lib.lib()
        terms = [nhwc_act(t) for t in terms]
        dt = terms[0].dtype
        n = len(meta)
        jobs = (_lib.SumFwdJob * n)()
        ys, k, cfg = [], 0, []
        for i, (shifts, nt) in enumerate(meta):
            ts = terms[k:k + nt]
            k += nt
            B, C = ts[0].shape[0], ts[0].shape[1]
            H = max(t.shape[2] << s for t, s in zip(ts, shifts))
            W = max(t.shape[3] << s for t, s in zip(ts, shifts))
            for t, s in zip(ts, shifts):
                if t.shape[1] != C or (t.shape[2] << s) != H or (t.shape[3] << s) != W:
                    raise ValueError('sum_relu_multi: term %s with shift %d does not match output %dx%dx%d' % (tuple(t.shape), s, C, H, W))
            y = _empty_nhwc(B, C, H, W, dt, ts[0].device)
            j = jobs[i]
            for q, (t, s) in enumerate(zip(ts, shifts)):
                j.terms[q] = t.data_ptr()
                j.shifts[q] = int(s)
            j.nterms, j.B, j.H, j.W, j.C, j.relu, j.y = nt, B, H, W, C, int(relu), y.data_ptr()
            ys.append(y)
            cfg.append((tuple(shifts), B, C, H, W))
        check(_k(L, 'danet_sum_relu_forward_multi', dt)(ctypes.addressof(jobs), n, stream()), 'danet_sum_relu_forward_multi')
        ctx.save_for_backward(*(ys if relu else []))
        ctx.cfg = (relu, cfg, dt)
        ctx.set_materialize_grads(False)         # an output nobody differentiates through contributes no job (not a tensor of zeros)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        L = _lib.lib()
        relu, cfg, dt = ctx.cfg
        ys = ctx.saved_tensors if relu else [None] * len(cfg)
        outs = []
        jobs = (_lib.SumBwdJob * len(cfg))()
        nj, keep = 0, []
        per_out = []
        for i, (shifts, B, C, H, W) in enumerate(cfg):
            if gys[i] is None:
                per_out.append(None)
                continue
            gy = nhwc_as(gys[i], dt)
            cache = {s: _empty_nhwc(B, C, H >> s, W >> s, dt, gy.device) for s in sorted(set(shifts))}
            j = jobs[nj]
            nj += 1
            j.gy, j.y = gy.data_ptr(), None if ys[i] is None else ys[i].data_ptr()
            j.B, j.H, j.W, j.C, j.relu = B, H, W, C, int(relu)
            for s in range(4):
                j.d[s] = cache[s].data_ptr() if s in cache else None
            keep.append(gy)
            per_out.append(cache)
        if nj:
            check(_k(L, 'danet_sum_relu_backward_all_multi', dt)(ctypes.addressof(jobs), nj, stream()), 'danet_sum_relu_backward_all_multi')
        for i, (shifts, B, C, H, W) in enumerate(cfg):
            for s in shifts:
                outs.append(None if per_out[i] is None else per_out[i][s])
        return (None, None) + tuple(outs)


def sum_relu_multi(groups, relu=True):
    """[sum_relu(terms, shifts) for terms, shifts in groups] in one launch per pass when the set qualifies (<= 4 outputs of <= 4 terms,
    shifts <= 3; terms of one output that share a shift share their gradient tensor, as in SumReluFunction); otherwise output by output."""
    ok = 1 <= len(groups) <= 4 and all(1 <= len(t) <= 4 and all(0 <= s <= 3 for s in sh) for t, sh in groups) and groups[0][0][0].is_cuda \
        and SUM_BWD_ALL
    if not ok or len(groups) == 1:
        return [sum_relu(t, sh, relu) for t, sh in groups]
    meta = [(tuple(sh), len(t)) for t, sh in groups]
    flat = [x for t, _ in groups for x in t]
    return list(SumReluMultiFunction.apply(relu, meta, *flat))


class FanOutFunction(torch.autograd.Function):
    """n aliases of x for n consumers; the backward sums the n incoming gradients in ONE launch (the fuse-layer sum
    kernel without ReLU or shifts) instead of autograd's n - 1 pairwise adds: (n + 1) instead of 3 (n - 1) tensor passes.
    Used where an HRNet branch output feeds the exchange convolutions of all other branches plus its own fuse sum."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        total = None
        while gs:                                   # groups of four (the kernel's limit); a running total rides along
            grp, gs = gs[:4 if total is None else 3], gs[4 if total is None else 3:]
            if total is not None:
                grp = [total] + grp
            if len(grp) == 1:
                total = grp[0]
                continue
            L = _lib.lib()
            dt = torch.float32 if grp[0].dtype == torch.float32 else torch.bfloat16
            terms = [nhwc_as(t, dt) for t in grp]
            B, C, H, W = terms[0].shape
            y = _empty_nhwc(B, C, H, W, dt, terms[0].device)
            ptrs = (ctypes.c_void_p * len(terms))(*[ptr(t.permute(0, 2, 3, 1)) for t in terms])
            sh = (ctypes.c_int * len(terms))(*([0] * len(terms)))
            check(_k(L, 'danet_sum_relu_forward', dt)(ptrs, sh, len(terms), B, H, W, C, 0, ptr(y.permute(0, 2, 3, 1)), stream()),
                  'danet_sum_relu_forward')
            total = y
        return total, None


class FanOutMultiFunction(torch.autograd.Function):
    """FanOutFunction for the m (<= 4) branch outputs of ONE HighResolutionModule at once: n aliases of each; the backward sums each
    branch's incoming gradients, all branches in ONE launch (csrc/norm_act.hip sum_relu_multi_kernel, no ReLU, no shifts) instead of one
    launch per branch."""

    @staticmethod
    def forward(ctx, n, *xs):
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        m = len(gs) // n
        outs = [None] * m
        jobs = (_lib.SumFwdJob * m)()
        nj, keep, dt0 = 0, [], None
        for i in range(m):
            grp = [g for g in gs[i * n:(i + 1) * n] if g is not None]
            if not grp:
                continue
            if len(grp) == 1:
                outs[i] = grp[0]
                continue
            dt = torch.float32 if grp[0].dtype == torch.float32 else torch.bfloat16
            if len(grp) > 4 or (dt0 is not None and dt != dt0):
                outs[i] = FanOutFunction.backward(ctx, *grp)[0]
                continue
            dt0 = dt
            terms = [nhwc_as(t, dt) for t in grp]
            B, C, H, W = terms[0].shape
            y = _empty_nhwc(B, C, H, W, dt, terms[0].device)
            j = jobs[nj]
            nj += 1
            for q, t in enumerate(terms):
                j.terms[q] = t.data_ptr()
                j.shifts[q] = 0
            j.nterms, j.B, j.H, j.W, j.C, j.relu, j.y = len(terms), B, H, W, C, 0, y.data_ptr()
            keep.append(terms)
            outs[i] = y
        if nj:
            check(_k(_lib.lib(), 'danet_sum_relu_forward_multi', dt0)(ctypes.addressof(jobs), nj, stream()), 'danet_sum_relu_forward_multi')
        return (None,) + tuple(outs)


FAN_OUT = bool(int(os.environ.get('DANET_FAN_OUT', '1')))


def fan_out_multi(xs, n):
    """[fan_out(x, n) for x in xs] with ONE gradient-sum launch for all of them (m <= 4 tensors that qualify for fan_out)."""
    ok = FAN_OUT and n > 2 and 1 < len(xs) <= 4 and torch.is_grad_enabled() and \
        all(x.is_cuda and x.requires_grad and x.dim() == 4 and x.shape[1] % 4 == 0 for x in xs)
    if not ok:
        return [fan_out(x, n) for x in xs]
    flat = FanOutMultiFunction.apply(n, *xs)
    return [list(flat[i * n:(i + 1) * n]) for i in range(len(xs))]


def fan_out(x, n):
    """n views of x whose gradients are summed by one kernel (CUDA NHWC tensors with channels % 4 == 0; plain
    aliases otherwise)."""
    if not (FAN_OUT and n > 2 and x.is_cuda and x.requires_grad and torch.is_grad_enabled()
            and x.dim() == 4 and x.shape[1] % 4 == 0):
        return [x] * n
    return list(FanOutFunction.apply(x, n))


def sum_relu(terms, shifts=None, relu=True):
    shifts = [0] * len(terms) if shifts is None else list(shifts)
    return SumReluFunction.apply(relu, shifts, *terms)


def relu(x):
    return sum_relu([x], [0], True)


class StnGatherFunction(torch.autograd.Function):
    """x [B,C,H,W], theta [B,P,2,3] -> [B,P*C,OH,OW]; gradient to x only (theta is detached in the
    reference, iuv_estimator.py:197)."""

    @staticmethod
    def forward(ctx, x, theta, out_hw, align_corners):
        L = _lib.lib()
        x = nhwc_act(x)
        dt = x.dtype
        B, C, H, W = x.shape
        th = theta.detach().float().contiguous()
        P = th.shape[1]
        OH, OW = out_hw
        y = _empty_nhwc(B, P * C, OH, OW, dt, x.device)
        check(_k(L, 'danet_stn_gather_forward', dt)(ptr(x.permute(0, 2, 3, 1)), ptr(th), B, H, W, C, P, OH, OW, int(align_corners),
                                         ptr(y.permute(0, 2, 3, 1)), stream()), 'danet_stn_gather_forward')
        ctx.save_for_backward(th)
        ctx.cfg = (B, C, H, W, P, OH, OW, int(align_corners), dt)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        (th,) = ctx.saved_tensors
        B, C, H, W, P, OH, OW, align, dt = ctx.cfg
        gy = nhwc_as(gy, dt)
        dx = _empty_nhwc(B, C, H, W, dt, gy.device)
        check(_k(L, 'danet_stn_gather_backward', dt)(ptr(gy.permute(0, 2, 3, 1)), ptr(th), B, H, W, C, P, OH, OW, align,
                                          ptr(dx.permute(0, 2, 3, 1)), stream()), 'danet_stn_gather_backward')
        return dx, None, None, None


def stn_gather(x, theta, out_hw=None, align_corners=True):
    out_hw = (x.shape[2], x.shape[3]) if out_hw is None else out_hw
    return StnGatherFunction.apply(x, theta, out_hw, align_corners)


class MaxPool3x3S2Function(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the regressor stems (res_module.py:303) on NHWC tensors
    (csrc/pool.hip): the forward records each maximum's window position (one byte per element), the backward gathers."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        x = nhwc_act(x)
        dt = x.dtype
        B, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = _empty_nhwc(B, C, OH, OW, dt, x.device)
        idx = torch.empty(B * OH * OW * C, dtype=torch.uint8, device=x.device)
        check(_k(L, 'danet_maxpool3x3s2_forward', dt)(ptr(x.permute(0, 2, 3, 1)), ptr(y.permute(0, 2, 3, 1)), ptr(idx), B, H, W, C, stream()),
              'danet_maxpool3x3s2_forward')
        ctx.save_for_backward(idx)
        ctx.cfg = (B, C, H, W, dt)
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        (idx,) = ctx.saved_tensors
        B, C, H, W, dt = ctx.cfg
        gy = nhwc_as(gy, dt)
        dx = _empty_nhwc(B, C, H, W, dt, gy.device)
        check(_k(L, 'danet_maxpool3x3s2_backward', dt)(ptr(gy.permute(0, 2, 3, 1)), ptr(idx), ptr(dx.permute(0, 2, 3, 1)), B, H, W, C, stream()),
              'danet_maxpool3x3s2_backward')
        return dx


def maxpool3x3s2(x):
    """F.max_pool2d(x, 3, 2, 1) on the HIP kernel for device tensors whose channel count fits its 16-byte lanes."""
    lanes = 4 if _conv.PRECISION == 'fp32' else 8
    if not (x.is_cuda and x.dim() == 4 and x.shape[1] % lanes == 0):
        # no tensor-op fall-back: every stem of the path has 64 channels (res_module.py:303; SmplResNet :404)
        raise RuntimeError('maxpool3x3s2: needs a device tensor [B, C, H, W] with C a multiple of %d (csrc/pool.hip); got %s on %s'
                           % (lanes, tuple(x.shape), x.device))
    return MaxPool3x3S2Function.apply(x)


def bump_batch_counters(module=None):
    """num_batches_tracked += 1 for every BatchNorm2d that ran in training mode since the last call (with
    BatchNorm2d.count_batches switched off), as ONE multi-tensor launch -- the modules torch would have counted
    (/root/reference's nn.BatchNorm2d.forward): not the never-used stacks, not eval-mode layers, not a skipped regressor."""
    counters, BatchNorm2d._ran = BatchNorm2d._ran, []
    if counters:
        torch._foreach_add_(counters, 1)
