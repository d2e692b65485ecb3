"""Convolution ops on the HIP MFMA kernels (csrc/conv_igemm.hip, conv_wgrad.hip).

Activations are bf16 NHWC (torch channels_last, logical shape [B,C,H,W]); parameters stay fp32
in the torch layout (state-dict compatible with the reference) and are packed to bf16 once per
parameter version.  Gradients: dX through the transposed gather of the same MFMA kernel, dW
through the wgrad kernel (fp32), dbias as a channel sum.
"""
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (only for nn.init helpers / shape utils)

from . import _lib
from ._lib import ptr, check, stream

_PACK_CACHE = {}


class ZeroArena(object):
    """One fp32 buffer per device that is zeroed ONCE per step; kernels that need a zero-initialised
    accumulator (BatchNorm statistics, atomically accumulated weight gradients) take slices of it
    instead of each issuing its own memset (~800 per step otherwise).  `begin_step()` rewinds the bump
    pointer and clears what the previous step used.  Inactive (capacity 0) unless a trainer enables it."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.high = 0
        self.zeroed = 0             # extent the current step's begin_step cleared
        self.dirty = 0              # extent that may hold non-zero data in MEMORY (a recorded clear has not run: captures do not reset it)
        self.refused = []           # (offset, floats, cleared extent) of the slices a capture was refused (diagnostics)

    def enable(self, device, floats=160 * 1024 * 1024):
        self.buf = torch.zeros(floats, dtype=torch.float32, device=device)
        self.off = 0
        self.high = floats
        self.dirty = 0

    def disable(self):
        self.buf = None

    def begin_step(self):
        if self.buf is not None:
            capturing = torch.cuda.is_current_stream_capturing()
            # eager: clear everything that may be dirty in memory -- also what an ABORTED capture attempt's step or the warm-up step
            # before it left beyond the last step's own extent (round-5 advisor: a retry recorded a clear of the partial extent only)
            ext = self.high if capturing else max(self.high, self.dirty)
            if ext > 0:
                self.buf[:ext].zero_()
            self.zeroed = ext
            if not capturing:
                self.dirty = 0
            self.high = 0
            self.off = 0

    def alloc(self, n):
        """-> (tensor, is_zero).  Falls back to a fresh (non-zeroed) tensor when inactive or full."""
        n16 = (n + 15) // 16 * 16
        if self.buf is None or self.off + n16 > self.buf.numel():
            return None
        if self.off + n16 > self.zeroed and torch.cuda.is_current_stream_capturing():
            # a captured step clears, on every replay, what the step BEFORE the capture used: a slice beyond that extent would be
            # zero in the first replay only (the caller falls back to a buffer it zeroes itself)
            self.refused.append((self.off, n, self.zeroed))
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n16
        self.high = max(self.high, self.off)
        self.dirty = max(self.dirty, self.off)
        return t


ARENA = ZeroArena()


def bn_sums_total(ws, C):
    """[2, C] fp32 totals of a statistics workspace (danet_bn_ws_floats(C) floats holding [replicas][2][C] accumulators of
    danet_bn_acc_bytes() bytes -- doubles unless the library was built with DANET_BN_ACC32): the replicas added in index
    order.  For tests and tools; the kernels read the workspace themselves."""
    acc = ws.view(torch.float64) if _lib.lib().danet_bn_acc_bytes() == 8 else ws
    return acc.view(-1, 2, C).sum(0).float()
# dgrad epilogue reduces the producing BN's backward sums.  Off by default since the one-pass BatchNorm backward (nn.ONEPASS:
# dy and x read once, sums and apply in one launch) -- measured 34.7 vs 35.1 ms/step with both, 35.4 with the fused
# reduction alone (the reduction costs the 3x3 data gradients +18 us per launch, as much as it saves elsewhere)
FUSE_BN_BWD_REDUCE = bool(int(os.environ.get('DANET_FUSE_BN_BWD', '0')))
FUSE_BN_BWD_STEM = bool(int(os.environ.get('DANET_FUSE_BN_BWD_STEM', '0')))      # ... for the 7x7 stems' data gradients only (conv2d; measured neutral in round 5: 26.70 vs 26.67 ms)
FUSE_BN_STATS = True     # conv epilogue accumulates the following BatchNorm's batch statistics
USE_WGRAD3X3 = True      # 3x3/s1 weight gradients through the LDS-transpose-read kernel
MULTI_DGRAD_SUBSETS = bool(int(os.environ.get('DANET_MULTI_DGRAD_SUBSETS', '1')))     # data gradients of a multi-conv set that does not qualify as a whole: qualifying subsets in one launch each
TRACE = None           # debugging: a list that receives (tag, shape, mean |value|) for every conv / BN launch (tools/debug_flaky.py)

# How often each attribute-carried fusion was taken / missed since the last FUSION.clear().  The fusions ride on
# tensor attributes (_bn_sums, _bn_ctx, _bn_red) and ResLink objects, which vanish silently if a view or copy gets
# in between; Trainer.capture() snapshots these counts for the captured step (trainer.fusion_counts) and
# tests/test_gpu_models.py asserts them, so a refactor that breaks a fusion fails a test instead of losing time.
import collections as _collections
FUSION = _collections.Counter()
PROFILER = None        # set by bench.py: object with begin(key, flops) -> token / end(token)


class KernelProfiler(object):
    """HIP-event timing of individual conv launches on the stream they are launched on."""

    def __init__(self, only=None):
        self.only = only
        self.records = []

    def begin(self, key, flops, shape=None):
        if self.only is not None and key != self.only:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return (key, flops, e0, e1, shape)

    def end(self, tok):
        if tok is not None:
            tok[3].record()
            self.records.append(tok)

    def summary(self):
        """{key: (launches, total_seconds, total_flops)} -- call after torch.cuda.synchronize()."""
        out = {}
        for key, flops, e0, e1, _ in self.records:
            n, t, f = out.get(key, (0, 0.0, 0.0))
            out[key] = (n + 1, t + e0.elapsed_time(e1) * 1e-3, f + flops)
        return out

    def by_shape(self):
        """{(key, shape): (launches, total_seconds, total_flops)} for tools/layer_profile.py."""
        out = {}
        for key, flops, e0, e1, shape in self.records:
            n, t, f = out.get((key, shape), (0, 0.0, 0.0))
            out[(key, shape)] = (n + 1, t + e0.elapsed_time(e1) * 1e-3, f + flops)
        return out



PRECISION = 'bf16'     # 'fp32': BASELINE config C4's arithmetic: fp32 NHWC activations, convolutions on the fp32 MFMA kernels of
                       # csrc/conv_f32m.hip (the direct kernels of csrc/conv_f32.hip where those do not take a problem), normalisation /
                       # activation / resampling on the fp32 instantiation of the same HIP kernels (csrc/norm_act_f32.hip, stn.hip).


class precision(object):
    """`with conv.precision('fp32'):` -- run the enclosed forward / backward passes in the fp32 verification mode."""

    def __init__(self, mode):
        if mode not in ('bf16', 'fp32'):
            raise ValueError('precision must be bf16 or fp32')
        self.mode = mode

    def __enter__(self):
        global PRECISION
        self.prev, PRECISION = PRECISION, self.mode
        return self

    def __exit__(self, *a):
        global PRECISION
        PRECISION = self.prev
        return False


def fp32_mode():
    return PRECISION == 'fp32'


F32_MFMA = bool(int(os.environ.get('DANET_F32_MFMA', '1')))     # fp32 mode on the matrix cores (csrc/conv_f32m.hip); 0 = the direct verification kernels only


def _pack_weight_f32(weight, w, groups, mode, Cout_gp, Cin_gp):
    """fp32 fragment-major operand of csrc/conv_f32m.hip for the fp32 weight `w` (= weight.detach()); cached per Parameter
    and version like pack_weight."""
    key = (id(weight), 10 + mode, groups, Cout_gp * 65536 + Cin_gp)
    cacheable = weight is not None and isinstance(weight, nn.Parameter)
    if cacheable:
        hit = _PACK_CACHE.get(key)
        if hit is not None and hit[0] == weight._version and hit[2]() is weight:
            return hit[1]
    L = _lib.lib()
    Cout, Cin_g, R, S = w.shape                           # (a group-padded copy arrives with its padded Cout)
    wp = torch.empty(L.danet_conv_f32m_packed_elems(Cout_gp, Cin_gp, R, S, groups, mode), dtype=torch.float32, device=w.device)
    check(L.danet_conv_f32m_pack_weights(ptr(w), ptr(wp), Cout, Cin_g, R, S, groups, mode, Cout_gp, Cin_gp, stream()), 'danet_conv_f32m_pack_weights')
    if cacheable:
        _PACK_CACHE[key] = (weight._version, wp, weakref.ref(weight))
    return wp


def _pad_last(t, n):
    """[..., C] -> [..., n] with zero channels appended (a copy; t itself when C == n)."""
    return t if t.shape[-1] == n else F.pad(t, (0, n - t.shape[-1]))


class Conv2dF32Function(torch.autograd.Function):
    """fp32 convolution: forward, data and weight gradient on the fp32 MFMA kernels (csrc/conv_f32m.hip) -- channel counts
    zero-padded to multiples of 4 for its 16-byte operand loads (for grouped convolutions: the output channels of every group,
    through a padded copy of the weight) --, or on the direct verification kernels (csrc/conv_f32.hip) where that kernel family
    does not take the problem (grouped convolutions with odd input channel counts, strided data gradients whose channel count is
    no multiple of 16) or F32_MFMA is off."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, groups):
        L = _lib.lib()
        B, Cin, H, W = x.shape
        Cout, Cin_g, R, S = weight.shape
        if Cin_g * groups != Cin:
            raise ValueError('conv2d: input has %d channels, weight expects %d' % (Cin, Cin_g * groups))
        OH, OW = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
        xh = x.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        b = None if bias is None else bias.detach().to(torch.float32).contiguous()
        Cout_g = Cout // groups
        Cin_gp, Cout_gp = (Cin_g + 3) // 4 * 4, (Cout_g + 3) // 4 * 4
        Cin_p, Cout_p = Cin_gp * groups, Cout_gp * groups
        gpad = groups > 1 and Cout_gp != Cout_g            # grouped: pad every group's output channels in a copy of the weight
        mfma = F32_MFMA and (groups == 1 or Cin_gp == Cin_g) and \
            bool(L.danet_conv_f32m_ok(B, H, W, Cin_p, OH, OW, Cout_p, R, S, stride, pad, dil, groups, 0))
        if mfma:
            xh = _pad_last(xh, Cin_p)
            if gpad:
                w = F.pad(w.view(groups, Cout_g, Cin_g, R, S), (0, 0, 0, 0, 0, 0, 0, Cout_gp - Cout_g)).reshape(Cout_p, Cin_g, R, S)
                if b is not None:
                    b = F.pad(b.view(groups, Cout_g), (0, Cout_gp - Cout_g)).reshape(-1)
            wp = _pack_weight_f32(None if gpad else weight, w, groups, 0, Cout_gp, Cin_gp)
            y = torch.empty(B, OH, OW, Cout_p, dtype=torch.float32, device=x.device)
            tok = PROFILER.begin('conv_f32m_kernel', 2.0 * B * OH * OW * Cout * R * S * Cin_g, (B, H, W, Cin, Cout, R, stride, groups)) if PROFILER is not None else None
            check(L.danet_conv_f32m_forward(ptr(xh), ptr(wp), ptr(None if b is None else _pad_last(b, Cout_p)), ptr(y), B, H, W, Cin_p, OH, OW, Cout_p,
                                            R, S, stride, pad, dil, groups, 0, 0, stream()), 'danet_conv_f32m_forward')
            if tok is not None:
                PROFILER.end(tok)
            y = y.view(B, OH, OW, groups, Cout_gp)[..., :Cout_g].reshape(B, OH, OW, Cout) if gpad else y[..., :Cout]
        else:
            y = torch.empty(B, OH, OW, Cout, dtype=torch.float32, device=x.device)
            check(L.danet_conv_f32(0, ptr(xh), ptr(w), ptr(b), ptr(y), B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, stream()), 'danet_conv_f32')
        ctx.save_for_backward(xh, w)
        ctx.weight = None if gpad else weight
        ctx.cfg = (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, bias is not None, mfma, gpad)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        xh, w = ctx.saved_tensors                           # (mfma: xh / w carry the padded channels)
        (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, has_bias, mfma, gpad) = ctx.cfg
        g = gy.to(torch.float32).permute(0, 2, 3, 1).contiguous()
        gx = gw = gb = None
        dims = (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups)
        Cin_g, Cout_g = Cin // groups, Cout // groups
        Cin_gp, Cout_gp = (Cin_g + 3) // 4 * 4, (Cout_g + 3) // 4 * 4
        Cin_p, Cout_p = Cin_gp * groups, Cout_gp * groups
        if mfma and gpad:
            gp = F.pad(g.view(B, OH, OW, groups, Cout_g), (0, Cout_gp - Cout_g)).reshape(B, OH, OW, Cout_p)
        else:
            gp = _pad_last(g, Cout_p) if mfma else g
        if ctx.needs_input_grad[0]:
            if mfma and L.danet_conv_f32m_ok(B, OH, OW, Cout_p, H, W, Cin_p, R, S, stride, pad, dil, groups, 1):
                wp = _pack_weight_f32(ctx.weight, w, groups, 1, Cout_gp, Cin_gp)
                gx = torch.empty(B, H, W, Cin_p, dtype=torch.float32, device=g.device)
                tok = PROFILER.begin('conv_f32m_kernel', 2.0 * B * OH * OW * Cout * R * S * Cin_g, ('dgrad', B, H, W, Cin, Cout, R, stride, groups)) if PROFILER is not None else None
                check(L.danet_conv_f32m_forward(ptr(gp), ptr(wp), None, ptr(gx), B, OH, OW, Cout_p, H, W, Cin_p, R, S, stride, pad, dil, groups, 1, 0,
                                                stream()), 'danet_conv_f32m_forward')
                if tok is not None:
                    PROFILER.end(tok)
                gx = gx[..., :Cin]
            else:
                wr = w.view(groups, Cout_gp, Cin_g, R, S)[:, :Cout_g].reshape(Cout, Cin_g, R, S) if (mfma and gpad) else w
                xr = xh if xh.shape[-1] == Cin else xh[..., :Cin].contiguous()
                gx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=g.device)
                check(L.danet_conv_f32(1, ptr(g), ptr(wr.contiguous()), None, ptr(gx), *dims, stream()), 'danet_conv_f32')
                del xr
            gx = gx.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            if mfma:
                Co = Cout_p if gpad else Cout              # (grouped: the padded weight's gradient, sliced below)
                gw = torch.empty(Co, Cin_g, R, S, dtype=torch.float32, device=g.device)
                wdims = (B, H, W, Cin_p, OH, OW, Cout_p, R, S, stride, pad, dil, groups, Co, Cin_g)
                ws = torch.empty(L.danet_conv_f32m_wgrad_ws_floats(*wdims), dtype=torch.float32, device=g.device)
                check(L.danet_conv_f32m_wgrad(ptr(xh), ptr(gp), ptr(gw), ptr(ws), *wdims, stream()), 'danet_conv_f32m_wgrad')
                if gpad:
                    gw = gw.view(groups, Cout_gp, Cin_g, R, S)[:, :Cout_g].reshape(Cout, Cin_g, R, S)
            else:
                gw = torch.empty_like(w)
                check(L.danet_conv_f32(2, ptr(xh), ptr(g), None, ptr(gw), *dims, stream()), 'danet_conv_f32')
        if has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 1, 2))
        return gx, gw, gb, None, None, None, None


def nhwc_as(x, dtype):
    """`dtype`, channels_last-contiguous view/copy of a [B,C,H,W] tensor."""
    if x.dtype != dtype:
        x = x.to(dtype)
    if not x.permute(0, 2, 3, 1).is_contiguous():
        x = x.contiguous(memory_format=torch.channels_last)
        if not x.permute(0, 2, 3, 1).is_contiguous():       # degenerate strides (C == 1 or H*W == 1)
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return x


def nhwc_bf16(x):
    """bf16, channels_last-contiguous view/copy of a [B,C,H,W] tensor."""
    return nhwc_as(x, torch.bfloat16)


def act_dtype():
    """Element type of the activations between the kernels: bf16, or fp32 under `precision('fp32')`."""
    return torch.float32 if PRECISION == 'fp32' else torch.bfloat16


def nhwc_act(x):
    return nhwc_as(x, act_dtype())


def _empty_nhwc(B, C, H, W, dtype, device):
    return torch.empty(B, H, W, C, dtype=dtype, device=device).permute(0, 3, 1, 2)


class WeightBank(object):
    """All packed weights of a model in ONE bf16 buffer, repacked by ONE launch per step.

    A training step otherwise repacks every conv weight individually after each optimizer step
    (~600 launches of a few microseconds).  Usage (see Trainer): `start_recording()`, run one step --
    pack_weight() notes every (parameter, groups, mode) it is asked for --, `build()`, then
    `refresh()` at the start of every following step: it launches the batched kernel and points
    pack_weight's cache at the bank's views (tagged with the parameters' current versions, so a
    weight modified later in the step simply misses and is packed individually)."""

    def __init__(self):
        self.requests = None
        self.entries = []
        self.jobs = None

    def start_recording(self):
        self.requests = {}

    def note(self, weight, groups, mode, chunk, pad_to=None):
        if self.requests is not None:
            self.requests[(id(weight), mode, groups, chunk, pad_to)] = weight

    def build(self):
        global RECORDER
        reqs, self.requests = self.requests or {}, None
        if RECORDER is self:
            RECORDER = None
        if not reqs:
            return self
        L = _lib.lib()
        dev = next(iter(reqs.values())).device
        sizes = []
        for (wid, mode, groups, chunk, pad_to), w in reqs.items():
            Cout, Cin_g, R, S = _padded_dims(w, pad_to)
            sizes.append((int(L.danet_conv_packed_elems(Cout // groups, Cin_g, R, S, groups, mode, chunk)) + 63) // 64 * 64)
        self.flat = torch.zeros(sum(sizes), dtype=torch.bfloat16, device=dev)      # zeroed once: the brick launch never writes padding
        jb = int(L.danet_conv_pack_job_bytes())
        import ctypes
        host = (ctypes.c_uint8 * (jb * len(reqs)))()
        off, start, bstart = 0, 0, 0
        self.entries = []
        for ((wid, mode, groups, chunk, pad_to), w), n in zip(reqs.items(), sizes):
            assert w.dtype == torch.float32 and w.is_contiguous()
            view = self.flat[off:off + n]
            Cout, Cin_g, R, S = _padded_dims(w, pad_to)
            tot = L.danet_conv_pack_job_fill_padded(ctypes.addressof(host) + jb * len(self.entries), w.data_ptr(), view.data_ptr(), start, bstart,
                                                    Cout, Cin_g, R, S, groups, mode, chunk, w.shape[0] // groups, w.shape[1])
            assert 0 < tot <= n
            self.entries.append(((wid, mode, groups, chunk, pad_to), weakref.ref(w), view, w.data_ptr()))
            off += n
            bricks = L.danet_conv_pack_job_bricks(Cout, Cin_g, R, S, groups, mode, chunk) if pad_to is None else 0
            if bricks > 0:                 # brick launch (csrc/conv_igemm.hip pack_weights_brick_kernel)
                bstart += bricks
            else:                          # per-element launch
                start += tot
        self.total, self.total_bricks = start, bstart
        self.jobs = torch.frombuffer(host, dtype=torch.uint8).clone().to(dev)
        return self

    def refresh(self):
        if self.jobs is None:
            return
        check(_lib.lib().danet_conv_pack_weights_batched(ptr(self.jobs), len(self.entries), self.total, self.total_bricks, stream()),
              'danet_conv_pack_weights_batched')
        for key, wref, view, dptr in self.entries:
            w = wref()
            if w is not None and w.data_ptr() == dptr:
                _PACK_CACHE[key] = (w._version, view, wref)


RECORDER = None         # a WeightBank in its recording step


def _padded_dims(weight, pad_to):
    """(Cout, Cin_g, R, S) the kernels run a weight at: its own shape, or pad_to = (Cout, Cin_g) with zero-padded channels."""
    Cout, Cin_g, R, S = weight.shape
    return (Cout, Cin_g, R, S) if pad_to is None else (int(pad_to[0]), int(pad_to[1]), R, S)


def pack_weight(weight, groups, mode, chunk=0, pad_to=None):
    """Packed bf16 copy of an fp32 conv weight (mode 0: forward operand, 1: data-gradient operand; chunk > 0:
    the chunked K order of the LDS 3x3 kernel; pad_to = (Cout, Cin_g): packed at zero-padded widths straight from the
    unpadded tensor).  Cached per nn.Parameter object and version (so a parameter is re-packed once per optimizer
    step); temporaries are never cached."""
    cacheable = isinstance(weight, nn.Parameter)
    key = (id(weight), mode, groups, chunk, pad_to)
    ver = weight._version
    if cacheable:
        if RECORDER is not None and weight.dtype == torch.float32 and weight.is_contiguous():
            RECORDER.note(weight, groups, mode, chunk, pad_to)
        hit = _PACK_CACHE.get(key)
        if hit is not None and hit[0] == ver and hit[2]() is weight:
            return hit[1]
    L = _lib.lib()
    Cout, Cin_g, R, S = _padded_dims(weight, pad_to)
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    n = L.danet_conv_packed_elems(Cout // groups, Cin_g, R, S, groups, mode, chunk)
    wp = torch.empty(n, dtype=torch.bfloat16, device=weight.device)
    check(L.danet_conv_pack_weights_padded(ptr(w), ptr(wp), Cout, Cin_g, R, S, groups, mode, chunk, weight.shape[0] // groups, weight.shape[1], stream()),
          'danet_conv_pack_weights')
    if cacheable:
        _PACK_CACHE[key] = (ver, wp, weakref.ref(weight))
    return wp


def _kernel_name(kid, stream_dims=None):
    """Kernel symbol (as rocprofv3 prints it) for a danet_conv_forward_kernel id; stream_dims = (B, H, W, Cin, Cout) of a 3x3
    problem without a fused BatchNorm-backward reduction: the streamed kernel (csrc/conv3x3s.hip) takes it when it has a plan."""
    if kid % 10 == 2:
        if stream_dims is not None:
            plan = _lib.lib().danet_conv3x3_stream_plan(*stream_dims, 1)
            if plan > 0:
                return 'conv3x3_stream_kernel<%d>' % (plan % 10)
        return 'conv3x3_tile_kernel'
    if kid % 10 == 3:
        return 'conv_pw_kernel<%d, %d>' % (kid // 1000, (kid // 100) % 10)
    if kid % 10 == 4:
        return 'conv3x3_stream_kernel<%d>' % ((kid // 100) % 10)
    if kid % 10 == 5:        # csrc/conv_g3.hip: <channels per group of the gathered tensor (48 forward: NT 2, 24 data gradient: NT 3), MT>
        return 'conv_g3_kernel<%d, %d>' % (48 if (kid // 100) % 10 == 2 else 24, kid // 1000)
    if kid % 10 == 1:
        return 'conv_fast_kernel<%d, %d>' % (kid // 1000, (kid // 100) % 10)
    return 'conv_igemm_kernel<%d, %d, %s>' % (kid // 1000, (kid // 100) % 10, 'true' if (kid // 10) % 10 else 'false')


def _multi_kernel_name(jobs, n, cout_g):
    import ctypes
    L = _lib.lib()
    which = L.danet_conv_forward_multi_kernel(ctypes.addressof(jobs), n)
    if which == 3:
        return 'conv3x3_stream_kernel<%d>' % L.danet_conv_nt(cout_g)
    if which == 2:
        return 'conv3x3_tile_kernel'
    return 'conv_fast_multi_kernel<%d>' % L.danet_conv_nt(cout_g)


def conv_out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


_S3_TABLES = {}          # device index -> the tensor the streamed 3x3 kernel's tap tables live in (csrc/conv3x3s.hip)
S3_TABLE_SLOTS = 512


def stream_tables(device):
    """The library allocates no device memory: the tap-table cache of the streamed 3x3 kernel lives in a buffer this side owns,
    registered once per device (danet_conv3x3_stream_tables; without it the kernel refuses and the tile kernel runs)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _S3_TABLES.get(idx)
    if t is None:
        L = _lib.lib()
        with torch.cuda.device(idx):
            t = torch.empty(S3_TABLE_SLOTS * int(L.danet_conv3x3_stream_table_bytes()), dtype=torch.uint8, device=torch.device('cuda', idx))
            if L.danet_conv3x3_stream_tables(ptr(t), t.numel()) != S3_TABLE_SLOTS:
                raise RuntimeError('danet_conv3x3_stream_tables: workspace not accepted')
        _S3_TABLES[idx] = t
    return t


def _conv_fwd_raw(x, wp, bias, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, transposed, relu, out_fp32,
                  bn_sums=None, bn_bwd=None, addend=None):
    """bn_bwd = (bn_x, gate tensor or None, saved, red, bn_gate) (see _bn_gate): data-gradient launches also reduce the
    BatchNorm-backward sums of the BN that produced the conv's input (see include/danet_hip.h)."""
    L = _lib.lib()
    stream_tables(x.device)
    y = _empty_nhwc(B, Cout, OH, OW, torch.float32 if out_fp32 else torch.bfloat16, x.device)
    tok = None
    if PROFILER is not None:
        kid = L.danet_conv_forward_kernel(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, int(transposed), int(out_fp32))
        name = _kernel_name(kid, (B, H, W, Cin, Cout) if bn_bwd is None else None)
        tok = PROFILER.begin(name,
                             2.0 * B * OH * OW * Cout * (Cin // groups) * R * S,
                             ('dgrad' if transposed else 'fwd', B, H, W, Cin, Cout, R, stride, groups))
    check(L.danet_conv_forward(ptr(x.permute(0, 2, 3, 1)), ptr(wp), ptr(bias), ptr(y.permute(0, 2, 3, 1)),
                               B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, int(transposed),
                               int(relu), int(out_fp32), ptr(bn_sums),
                               None if bn_bwd is None else ptr(bn_bwd[0].permute(0, 2, 3, 1)),
                               None if bn_bwd is None or bn_bwd[1] is None else (ptr(bn_bwd[1]) if bn_bwd[1].dim() == 1 else ptr(bn_bwd[1].permute(0, 2, 3, 1))),
                               None if bn_bwd is None else ptr(bn_bwd[2]), None if bn_bwd is None else ptr(bn_bwd[3]),
                               None if addend is None else ptr(addend.permute(0, 2, 3, 1)),
                               0 if bn_bwd is None else bn_bwd[4], stream()), 'danet_conv_forward')
    if tok is not None:
        PROFILER.end(tok)
    if TRACE is not None:
        TRACE.append(('dgrad' if transposed else 'conv', (B, H, W, Cin, Cout, R, stride), y.float().abs().mean()))
    return y


def _conv_stem_raw(x, wp16, B, H, W, Cin, OH, OW, Cout, bn_sums=None):
    """csrc/conv_stem.hip: 7x7 / stride 2 / pad 3 forward on LDS tiles; wp16 = pack_weight(w, 1, 0, chunk=16)."""
    L = _lib.lib()
    y = _empty_nhwc(B, Cout, OH, OW, torch.bfloat16, x.device)
    tok = PROFILER.begin('conv_stem_kernel', 2.0 * B * OH * OW * Cout * Cin * 49, ('fwd', B, H, W, Cin, Cout, 7, 2, 1)) if PROFILER is not None else None
    check(L.danet_conv_stem_forward(ptr(x.permute(0, 2, 3, 1)), ptr(wp16), ptr(y.permute(0, 2, 3, 1)), B, H, W, Cin, OH, OW, Cout, ptr(bn_sums), stream()),
          'danet_conv_stem_forward')
    if tok is not None:
        PROFILER.end(tok)
    return y


C3A_WIDTHS = (16,)        # map widths csrc/conv3x3a.hip is used for (measured, tools/c3a_bench.py: 768 x 16 x 16: 29.1 / 26.5 us forward / data gradient
#                           against 32.2 / 30.4 on the LDS-tile kernel; 32 x 64 x 64: 22.5 / 21.1 against 20.6 / 18.8 -- two tiles per workgroup: slower)


def _conv3x3a_raw(x, wp16, B, H, W, transposed, bn_sums=None, bn_bwd=None, addend=None):
    """csrc/conv3x3a.hip: 3x3 / stride 1 / pad 1, 64 -> 64 channels, forward (wp16 = pack_weight(w, 1, 0, chunk=16)) or data gradient
    (transposed: x is dy, wp16 = pack_weight(w, 1, 1, chunk=16)); bn_bwd = (bn_x, gate tensor or None, saved, red, gate mode 0 / 2)."""
    L = _lib.lib()
    y = _empty_nhwc(B, 64, H, W, torch.bfloat16, x.device)
    bx = by = sv = rd = None
    gate = 0
    if bn_bwd is not None:
        gate = int(bn_bwd[4])
        by = None if bn_bwd[1] is None else (ptr(bn_bwd[1].permute(0, 2, 3, 1)) if gate == 0 else bn_bwd[1].data_ptr())
        bx, sv, rd = ptr(bn_bwd[0].permute(0, 2, 3, 1)), ptr(bn_bwd[2]), ptr(bn_bwd[3])
        if by is None:
            gate = 0
    tok = PROFILER.begin('conv3x3a_kernel', 2.0 * B * H * W * 64 * 64 * 9, ('dgrad' if transposed else 'fwd', B, H, W, 64, 64, 3, 1, 1)) if PROFILER is not None else None
    check(L.danet_conv3x3a(ptr(x.permute(0, 2, 3, 1)), ptr(wp16), ptr(y.permute(0, 2, 3, 1)), B, H, W, int(transposed), ptr(bn_sums), bx, by, sv, rd, gate,
                           None if addend is None else ptr(addend.permute(0, 2, 3, 1)), stream()), 'danet_conv3x3a')
    if tok is not None:
        PROFILER.end(tok)
    return y


def _conv_stem_dgrad_raw(gy, wp16t, B, H, W, Cin, OH, OW, Cout, bn_bwd=None):
    """csrc/conv_stem_dgrad.hip: data gradient of a 7x7 / stride 2 / pad 3 stem (64 -> 64 channels) on LDS tiles; wp16t =
    pack_weight(w, 1, 1, chunk=16); bn_bwd = (bn_x, gate tensor or None, saved, red, gate mode 0) as for _conv_fwd_raw."""
    L = _lib.lib()
    gx = _empty_nhwc(B, Cin, H, W, torch.bfloat16, gy.device)
    bx = by = sv = rd = None
    if bn_bwd is not None:
        bx, by, sv, rd = ptr(bn_bwd[0].permute(0, 2, 3, 1)), None if bn_bwd[1] is None else ptr(bn_bwd[1].permute(0, 2, 3, 1)), ptr(bn_bwd[2]), ptr(bn_bwd[3])
    tok = PROFILER.begin('conv_stem_dgrad_kernel', 2.0 * B * OH * OW * Cout * Cin * 49, ('dgrad', B, OH, OW, Cout, Cin, 7, 2, 1)) if PROFILER is not None else None
    check(L.danet_conv_stem_dgrad(ptr(gy.permute(0, 2, 3, 1)), ptr(wp16t), ptr(gx.permute(0, 2, 3, 1)), B, H, W, Cin, OH, OW, Cout, bx, by, sv, rd, stream()),
          'danet_conv_stem_dgrad')
    if tok is not None:
        PROFILER.end(tok)
    return gx


class ResLink(object):
    """Carries the residual branch's gradient of a block (`out += residual`, res_module.py:39-56) from the closing
    BatchNorm's backward to the data gradient of the block's first convolution, whose epilogue adds it -- instead of
    autograd summing the two gradients of the block input in a pass of its own.  One per block and forward pass."""
    __slots__ = ('dres', 'armed')

    def __init__(self):
        self.dres = None
        self.armed = False        # set by the convolution that will consume dres in its backward (otherwise the BatchNorm keeps autograd's path)


class Conv2dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, groups, out_fp32, bn_sums=None, bn_ctx=None, link=None, wpad=None):
        """wpad = (Cout, Cin_g): the weight runs zero-padded to these widths (packed straight from the unpadded parameter: its packed
        copies live in the WeightBank like any other layer's, and the weight gradient is cropped back to the parameter's shape)."""
        x = nhwc_bf16(x)
        B, Cin, H, W = x.shape
        Cout, Cin_g, R, S = _padded_dims(weight, wpad)
        if Cin_g * groups != Cin:
            raise ValueError('conv2d: input has %d channels, weight expects %d' % (Cin, Cin_g * groups))
        OH, OW = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
        b = None if bias is None else bias.detach().float().contiguous()
        if b is None and not out_fp32 and W in C3A_WIDTHS and _lib.lib().danet_conv3x3a_ok(B, H, W, Cin, Cout, R, S, stride, pad, dil, groups):
            # the regressor ResNets' layer1 (64 -> 64 channels, 3x3): the AGPR row-tile kernel, weights in the chunked packing
            y = _conv3x3a_raw(x, pack_weight(weight, groups, 0, 16, wpad), B, H, W, False, bn_sums)
        elif b is None and not out_fp32 and _lib.lib().danet_conv_stem_ok(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups):
            # the regressors' 7x7 / stride-2 stems over the part crops: LDS-tile kernel, weights in the chunked (16-channel slab) packing
            y = _conv_stem_raw(x, pack_weight(weight, groups, 0, 16, wpad), B, H, W, Cin, OH, OW, Cout, bn_sums)
        else:
            wp = pack_weight(weight, groups, 0, 0, wpad)
            y = _conv_fwd_raw(x, wp, b, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, False, False, out_fp32, bn_sums)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, dil, groups, bias is not None)
        ctx.wpad = wpad
        ctx.bn_ctx = bn_ctx
        ctx.link = link
        if link is not None and ctx.needs_input_grad[0]:
            link.armed = True
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        x, weight = ctx.saved_tensors
        stride, pad, dil, groups, has_bias = ctx.cfg
        wpad = ctx.wpad
        B, Cin, H, W = x.shape
        Cout, Cin_g, R, S = _padded_dims(weight, wpad)
        gy = nhwc_bf16(gy)
        OH, OW = gy.shape[2], gy.shape[3]
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            # the weight gradient is off the critical path (only the optimizer consumes it): with DEFER_WGRAD it is
            # only queued here and computed by flush_wgrads() in multi-problem launches after the backward pass
            if wpad is None:
                gw = new_wgrad(weight, (Cout, Cin_g, R, S), x.device)
                _wgrad_into(gw, x, gy, B, H, W, Cin, OH, OW, Cout, Cin_g, R, S, stride, pad, dil, groups, weight)
            else:
                # at the padded widths, then cropped to the parameter's own shape (one launch of csrc/glue.hip) -- queued like every
                # other weight gradient (round 5: these ran inside the backward chain, ~35 us of launches per layer of the heads), the
                # crop then follows the multi-problem launch and writes the tensor autograd already holds as .grad
                gwp = torch.empty(Cout, Cin_g, R, S, dtype=torch.float32, device=x.device)
                so, si = weight.shape[0] // groups, weight.shape[1]
                views = ((groups, Cout // groups, Cin_g, R * S), (groups, so, si, R * S))
                gw = new_wgrad(weight, tuple(weight.shape), x.device) if (DEFER_WGRAD and DEFER_PADDED) else None
                # (the queue keeps the ADDRESS of gw only: a second reference would make autograd clone it instead of adopting it as .grad)
                if gw is None or not _enqueue_wgrad(gwp.data_ptr(), weight, x, gy, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups,
                                                    post=(gwp, gw.data_ptr(), views[0], views[1])):
                    if gw is not None and GRAD_STORE is not None and GRAD_STORE.has(weight) and gw.data_ptr() == GRAD_STORE.grad_ptr(weight):
                        GRAD_STORE._handed.discard(id(weight))          # (the slot was taken for a gradient that is not queued after all)
                    _wgrad_into(gwp, x, gy, B, H, W, Cin, OH, OW, Cout, Cin_g, R, S, stride, pad, dil, groups, None)
                    from .glue import crop
                    gw = crop(gwp, views[0], views[1], tuple(weight.shape))
        if ctx.needs_input_grad[0]:
            bn_bwd = None
            # (ctx.bn_ctx is only there when conv2d decided for the fused reduction: FUSE_BN_BWD_REDUCE, or FUSE_BN_BWD_STEM for this layer)
            if ctx.bn_ctx is not None and \
                    L.danet_conv_forward_kernel(B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, dil, groups, 1, 0) % 10 in (1, 2):
                bn_x, saved = ctx.bn_ctx[0], ctx.bn_ctx[2]
                c3 = L.danet_conv_forward_kernel(B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, dil, groups, 1, 0) % 10 == 2
                gate_t, gate = _bn_gate(ctx.bn_ctx, x, c3)
                if bn_x.shape == x.shape:
                    n = L.danet_bn_ws_floats(Cin)
                    red = ARENA.alloc(n)
                    if red is None:
                        red = torch.zeros(n, dtype=torch.float32, device=x.device)
                    bn_bwd = (bn_x, gate_t, saved, red, gate)
            addend = None
            if ctx.link is not None and ctx.link.dres is not None:
                addend, ctx.link.dres = ctx.link.dres, None
            fused_add = addend is not None and addend.shape == x.shape and bn_bwd is None and \
                L.danet_conv_forward_kernel(B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, dil, groups, 1, 0) % 10 in (1, 2, 3)      # gather, LDS-tile 3x3, pointwise
            if (addend is None or (bn_bwd is None and addend.shape == x.shape and addend.dtype == torch.bfloat16)) and \
                    (bn_bwd is None or bn_bwd[0].dtype == torch.bfloat16) and W in C3A_WIDTHS and L.danet_conv3x3a_ok(B, H, W, Cin, Cout, R, S, stride, pad, dil, groups):
                gx = _conv3x3a_raw(gy, pack_weight(weight, groups, 1, 16, wpad), B, H, W, True, None, bn_bwd, addend)
                if addend is not None:
                    FUSION['residual_grad_fused'] += 1
                    addend = None
            elif addend is None and (bn_bwd is None or (bn_bwd[4] == 0 and bn_bwd[0].dtype == torch.bfloat16)) and \
                    L.danet_conv_stem_dgrad_ok(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups):
                # the part-crop stem (64 -> 64 channels, 7x7 / stride 2): LDS-tile kernel with the same fused BatchNorm-backward sums
                gx = _conv_stem_dgrad_raw(gy, pack_weight(weight, groups, 1, 16, wpad), B, H, W, Cin, OH, OW, Cout, bn_bwd)
            else:
                gx = _conv_fwd_raw(gy, pack_weight(weight, groups, 1, 0, wpad), None, B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, dil, groups, True, False, False,
                                   None, bn_bwd, addend if fused_add else None)
            if addend is not None:
                FUSION['residual_grad_fused' if fused_add else 'residual_grad_added'] += 1
                if not fused_add:
                    gx = gx + addend
            if bn_bwd is not None:
                gx._bn_red = bn_bwd[3]       # consumed by the producing BatchNorm's backward if gx reaches it unsummed
        if has_bias and ctx.needs_input_grad[2]:
            gb = channel_sum(gy)
            if TRACE is not None:
                TRACE.append(('bias_grad:gy', tuple(gy.shape), gy.float().abs().max()))
                TRACE.append(('bias_grad:gb', tuple(gb.shape), gb.float().abs().max()))
        return gx, gw, gb, None, None, None, None, None, None, None, None, None


def channel_sum(gy):
    """gy.sum(dim=(0, 2, 3)) in fp32 for an NHWC bf16 / fp32 tensor (bias gradients) on csrc/norm_act.hip channel_sum_kernel."""
    B, C, H, W = gy.shape
    if not gy.is_cuda or C % 4 != 0 or gy.dtype not in (torch.bfloat16, torch.float32) or not gy.permute(0, 2, 3, 1).is_contiguous():
        return gy.sum(dim=(0, 2, 3), dtype=torch.float32)
    # double accumulators: order-independent (csrc/conv_common.h); a slice of the step's zeroed arena when there is one
    # (round 6) 16 replicas for the large maps: 1 024 workgroups instead of 256 queueing on the same C addresses; summed here
    ncopy = CHSUM_COPIES if B * H * W * C >= (1 << 21) else 1
    out = ARENA.alloc(2 * C * ncopy) if CHSUM_ARENA else None
    zero = out is not None
    out = out.view(torch.float64) if zero else torch.empty(C * ncopy, dtype=torch.float64, device=gy.device)
    L = _lib.lib()
    fn = L.danet_channel_sum_f32 if gy.dtype == torch.float32 else L.danet_channel_sum
    check(fn(ptr(gy.permute(0, 2, 3, 1)), B * H * W, C, ptr(out), int(zero), ncopy, stream()), 'danet_channel_sum')
    return out.float() if ncopy == 1 else out.view(ncopy, C).sum(0).float()


CHSUM_ARENA = bool(int(os.environ.get('DANET_CHSUM_ARENA', '1')))       # A-B / debugging knob: 0 = a memset node per call
CHSUM_COPIES = int(os.environ.get('DANET_CHSUM_COPIES', '16'))            # replicas of a bias-gradient sum over a large map (1: the old single copy)


DEFER_WGRAD = False       # queue weight gradients during backward; flush_wgrads() computes them in multi-problem launches
DEFER_PADDED = bool(int(os.environ.get('DANET_DEFER_PADDED_WGRAD', '1')))       # ... those of channel-padded layers as well (A-B knob)
_WQ = []                  # 3x3 / stride 1 or 2 problems (conv_wgrad3x3.hip)
_WQG = []                 # everything else (conv_wgrad.hip)
GRAD_STORE = None         # distributed.GradStore of the running trainer: weight gradients are written into its views


def new_wgrad(weight, shape, device):
    """The tensor a weight-gradient kernel writes: the parameter's slot of the flat gradient storage (zero-copy for
    the all-reduce and the optimizer) when there is one and the parameter has no gradient yet, else a new tensor."""
    st = GRAD_STORE
    if st is not None and isinstance(weight, nn.Parameter) and weight.grad is None and st.has(weight) and tuple(shape) == tuple(weight.shape):
        v = st.take(weight)             # None: a second use of the same parameter in this backward pass (the slot is taken)
        if v is not None:
            return v
    return torch.empty(*shape, dtype=torch.float32, device=device)


def _check_adopted(queue):
    """autograd must have kept the returned (still unwritten) gradient tensor itself as .grad -- a copy, or an
    accumulation into an existing .grad (shared weights, gradient accumulation), would have read garbage."""
    for q in queue:
        gptr, weight = q[0], q[1]
        if q[-1] is not None:             # (a channel-padded layer: the kernel writes a padded temporary, .grad is the tensor it is cropped into)
            gptr = q[-1][1]
        if weight.grad is None or weight.grad.data_ptr() != gptr:
            _WQ.clear()
            _WQG.clear()
            raise RuntimeError('deferred weight gradient of a %s parameter was copied or accumulated by autograd; '
                               'set DANET_DEFER_WGRAD=0 for this model' % (tuple(weight.shape),))


def flush_wgrads(bucket=None, hold=None):
    """Compute every queued weight gradient (call after backward, before anything reads parameter .grad).  With
    `bucket` = a bucket index of GRAD_STORE only that bucket's parameters are processed (the trainer walks the buckets
    in order and all-reduces each one while the next one's launches run).  hold: a list that receives the processed queue
    entries -- a caller that flushes on a SIDE stream keeps the operands (x, dy: allocated on the step's stream) alive until the
    streams have joined again."""
    import ctypes
    L = _lib.lib()

    if bucket is not None and not isinstance(bucket, int):
        bucket = frozenset(bucket)                          # a run of buckets released together (GradStore.release_ready_group)

    def mine(q):
        if bucket is None or GRAD_STORE is None:
            return True
        b = GRAD_STORE.bucket_of.get(id(q[1]), -1)
        return b == bucket if isinstance(bucket, int) else b in bucket
    # (channel-padded layers -- post is not None -- go out in launches of their own: the multi-problem planners deal a fixed number of
    # workgroups over the jobs of a call, and a dozen tiny layers in the same call took workgroups away from every other launch: +0.3 ms)
    for wq in ([q for q in _WQ if mine(q) and q[-1] is None], [q for q in _WQ if mine(q) and q[-1] is not None]):
        if not wq:
            continue
        _check_adopted(wq)
        jobs = (_lib.Wg3Job * len(wq))()
        for j, (gptr, weight, x, gy, B, H, W, Cin, Cout, groups, stride, _post) in zip(jobs, wq):
            j.x, j.dy, j.dw = x.data_ptr(), gy.data_ptr(), gptr
            j.B, j.H, j.W, j.Cin, j.Cout, j.groups, j.stride = B, H, W, Cin, Cout, groups, stride
        n = len(wq)
        need = L.danet_conv_wgrad3x3_multi_ws_floats(ctypes.addressof(jobs), n)
        ws = torch.empty(need, dtype=torch.float32, device=wq[0][2].device)
        tok = PROFILER.begin('conv_wgrad3x3_multi', sum(2.0 * q[4] * (q[5] // q[10]) * (q[6] // q[10]) * q[8] * (q[7] // q[9]) * 9 for q in wq),
                             ('wgrad-multi', n)) if PROFILER is not None else None
        check(L.danet_conv_wgrad3x3_multi(ctypes.addressof(jobs), n, ptr(ws), need, 0.0, stream()), 'danet_conv_wgrad3x3_multi')
        if tok is not None:
            PROFILER.end(tok)
        _crop_posts(wq)
        if hold is not None:
            hold.extend(wq)
    _WQ[:] = [q for q in _WQ if not mine(q)]
    for wqg in ([q for q in _WQG if mine(q) and q[-1] is None], [q for q in _WQG if mine(q) and q[-1] is not None]):
        if not wqg:
            continue
        _check_adopted(wqg)
        jobs = (_lib.WgJob * len(wqg))()
        for j, (gptr, weight, x, gy, dims, _post) in zip(jobs, wqg):
            j.x, j.dy, j.dw = x.data_ptr(), gy.data_ptr(), gptr
            (j.B, j.H, j.W, j.Cin, j.OH, j.OW, j.Cout, j.R, j.S, j.stride, j.pad, j.dil, j.groups) = dims
        n = len(wqg)
        need = L.danet_conv_wgrad_multi_ws_floats(ctypes.addressof(jobs), n)
        zfrom = L.danet_conv_wgrad_multi_ws_zero_from(ctypes.addressof(jobs), n)
        # the leading part (partial sums of the pointwise kernel: ~290 MB of the bench step's 377 MB) needs no zeroing: only the packed
        # accumulators behind it come out of the zeroed arena when the two can be adjacent -- else a buffer of its own, tail zeroed
        ws = _wgrad_scratch(need, zfrom, wqg[0][2].device)
        tok = PROFILER.begin('conv_wgrad_multi', sum(2.0 * d[0] * d[4] * d[5] * d[6] * (d[3] // d[12]) * d[7] * d[8] for d in (q[4] for q in wqg)),
                             ('wgrad-multi', n)) if PROFILER is not None else None
        check(L.danet_conv_wgrad_multi(ctypes.addressof(jobs), n, ptr(ws), need, 0.0, stream()), 'danet_conv_wgrad_multi')
        if tok is not None:
            PROFILER.end(tok)
        _crop_posts(wqg)
        if hold is not None:
            hold.extend(wqg)
    _WQG[:] = [q for q in _WQG if not mine(q)]


def _crop_posts(queue):
    """The channel-padded layers of a flushed queue: their gradients, computed at the padded widths, are cropped into the tensors autograd
    holds as .grad (one launch per 16 layers)."""
    posts = [(q[-1], q[1].grad) for q in queue if q[-1] is not None]          # (.grad IS the tensor returned from the backward node: _check_adopted)
    if posts:
        from .glue import crop_into
        crop_into([p[0] for p, _ in posts], [p[2] for p, _ in posts], [p[3] for p, _ in posts], [g for _, g in posts])


def _enqueue_wgrad(gptr, weight, x, gy, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, post=None):
    """Queue a weight gradient for flush_wgrads (the kernel will write at address gptr); False when this problem is not queued
    (deferral off, the parameter already has a gradient, or the 7x7 stem kernel -- a chip-filling launch of its own)."""
    L = _lib.lib()
    if not (DEFER_WGRAD and isinstance(weight, nn.Parameter) and weight.grad is None):
        return False
    if L.danet_conv_wgrad_rows_ok(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups):
        return False
    if USE_WGRAD3X3 and (L.danet_conv_wgrad3x3_ok(H, W, Cin, Cout, R, S, stride, pad, dil, groups) or
                         L.danet_conv_wgrad3x3_pair_ok(B, H, W, Cin, Cout, R, S, stride, pad, dil, groups)):      # (4 x 4 maps: two images per chunk)
        # only the ADDRESS of the target is kept: holding the tensor would make autograd clone it instead of adopting it as .grad
        _WQ.append((gptr, weight, x, gy, B, H, W, Cin, Cout, groups, stride, post))      # x, gy stay alive until the flush
    else:
        _WQG.append((gptr, weight, x, gy, (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups), post))
    return True


def _wgrad_scratch(need, zero_from, device):
    """need floats of scratch whose floats [zero_from, need) are zero (one fill launch over that tail only)."""
    ws = torch.empty(need, dtype=torch.float32, device=device)
    if need > zero_from:
        ws[zero_from:].zero_()
    return ws


def _wgrad_into(gw, x, gy, B, H, W, Cin, OH, OW, Cout, Cin_g, R, S, stride, pad, dil, groups, weight=None):
    L = _lib.lib()
    if L.danet_conv_wgrad_rows_ok(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups):
        # the regressor ResNets' 7x7 stride-2 stems over the 768 part crops: a chip-filling launch of its own, never queued
        nws = L.danet_conv_wgrad_rows_ws_floats(B, OH, OW, Cin, Cout, R, S, groups)
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        tok = PROFILER.begin('conv_wgrad_rows_kernel', 2.0 * B * OH * OW * Cout * Cin_g * R * S,
                             ('wgrad', B, H, W, Cin, Cout, R, stride, groups)) if PROFILER is not None else None
        check(L.danet_conv_wgrad_rows(ptr(x.permute(0, 2, 3, 1)), ptr(gy.permute(0, 2, 3, 1)), ptr(gw), ptr(ws), nws,
                                      B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, groups, 0.0, stream()), 'danet_conv_wgrad_rows')
        if tok is not None:
            PROFILER.end(tok)
        return
    if _enqueue_wgrad(gw.data_ptr(), weight, x, gy, B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups):
        return
    if USE_WGRAD3X3 and L.danet_conv_wgrad3x3_ok(H, W, Cin, Cout, R, S, stride, pad, dil, groups):
        nws = L.danet_conv_wgrad3x3_ws_floats(B, H, W, Cin, Cout, groups, stride)
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        args = (ptr(x.permute(0, 2, 3, 1)), ptr(gy.permute(0, 2, 3, 1)), ptr(gw), ptr(ws), nws, B, H, W, Cin, Cout, groups, stride, 0.0)
        if PROFILER is None:
            check(L.danet_conv_wgrad3x3(*args, 0, stream()), 'danet_conv_wgrad3x3')
        else:                                          # the MFMA kernel and the reduction bracketed separately
            kid = L.danet_conv_wgrad3x3_kernel_id(B, H, W, Cin, Cout, groups, stride)
            tok = PROFILER.begin('conv_wgrad3x3_kernel<%d, %d>' % (kid // 10, kid % 10), 2.0 * B * OH * OW * Cout * Cin_g * 9,
                                 ('wgrad', B, H, W, Cin, Cout, R, stride, groups))
            check(L.danet_conv_wgrad3x3(*args, 1, stream()), 'danet_conv_wgrad3x3')
            PROFILER.end(tok)
            tok = PROFILER.begin('wgrad3x3_reduce_kernel', 0.0, ('wgrad-reduce', B, H, W, Cin, Cout, R, stride, groups))
            check(L.danet_conv_wgrad3x3(*args, 2, stream()), 'danet_conv_wgrad3x3')
            PROFILER.end(tok)
        return
    nws = L.danet_conv_wgrad_ws_floats_for(B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups)
    ws = ARENA.alloc(nws)
    ws_zero = ws is not None
    if ws is None:
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    tok = None
    if PROFILER is not None:
        kid = L.danet_conv_wgrad_kernel_id(Cin, Cout, groups, R * S)
        tok = PROFILER.begin('conv_wgrad_kernel<%d, %d, %d>' % (kid // 100, (kid // 10) % 10, kid % 10),
                             2.0 * B * OH * OW * Cout * Cin_g * R * S,
                             ('wgrad', B, H, W, Cin, Cout, R, stride, groups))
    check(L.danet_conv_wgrad(ptr(x.permute(0, 2, 3, 1)), ptr(gy.permute(0, 2, 3, 1)), ptr(gw), ptr(ws), nws,
                             B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups, 0.0, int(ws_zero), stream()),
          'danet_conv_wgrad')
    if tok is not None:
        PROFILER.end(tok)


PACK_IMAGE = bool(int(os.environ.get('DANET_PACK_IMAGE', '1')))      # A-B knob


def _pad_channels_nhwc(x, mult=8):
    """bf16 NHWC tensor whose channel count is padded with zeros to a multiple of `mult`
    (autograd-tracked, so the gradient is sliced back)."""
    if PACK_IMAGE and mult == 8 and x.is_cuda and x.dtype == torch.float32 and x.shape[1] < 8 and x.is_contiguous() and not x.requires_grad:
        from .glue import pack_image            # the input image: cast + channels-last + zero channels in one launch (csrc/glue.hip)
        return pack_image(x)
    x = nhwc_bf16(x)
    padc = (-x.shape[1]) % mult
    if padc == 0:
        return x
    return F.pad(x.permute(0, 2, 3, 1), (0, padc)).permute(0, 3, 1, 2)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, out_fp32=False, want_stats=False,
           keep_group_padding=False, link=None, weight_pad=None):
    """Convolution on the MFMA kernels.  Channel counts that are not a multiple of 8 (3-channel image,
    21/75-channel IUV maps, 25/15/21-channel heads) are zero-padded to the next multiple of 8 so that
    forward, dgrad and wgrad all take the 16-byte vector path; the padding is sliced off again."""
    if PRECISION == 'fp32':
        if x.shape[1] != weight.shape[1] * groups:           # a producer's zero-padded channels: drop them
            x = x[:, :weight.shape[1] * groups]
        return Conv2dF32Function.apply(x, weight, bias, stride, padding, dilation, groups)
    if weight_pad is not None:                              # the caller runs this layer at padded widths (resnet.Bottleneck._forward_padded)
        Cout, Cin_w = int(weight_pad[0]), int(weight_pad[1])
    else:
        Cout, Cin_w = weight.shape[0], weight.shape[1]
    padc = 0
    if groups == 1 and x.shape[1] != Cin_w and x.shape[1] == Cin_w + (-Cin_w) % 8:
        padc = x.shape[1] - Cin_w        # the producer already zero-padded the channels to a multiple of 8 (part_ops.part_clean): pad the weight only
    elif groups == 1 and x.shape[1] % 8 != 0:
        padc = (-x.shape[1]) % 8
        x = _pad_channels_nhwc(x)
    Cout_g = Cout // groups
    padn = (-Cout_g) % 8
    wpad = weight_pad
    if padc or padn:
        if weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous() and weight_pad is None:
            # the weight is packed at the padded widths straight from the parameter (pack_weight pad_to: its packed copies live in the
            # WeightBank like any other layer's) and its gradient is cropped back; only a bias still needs a padded copy
            wpad = (groups * (Cout_g + padn), Cin_w + padc)
            if bias is not None and padn:
                from .glue import pad_multi
                bias = pad_multi([(bias, (groups, Cout_g), (groups, Cout_g + padn), (groups * (Cout_g + padn),))])[0]
        else:
            if padc:
                weight = F.pad(weight, (0, 0, 0, 0, 0, padc))
            if padn:
                wv = weight.view(groups, Cout_g, *weight.shape[1:])
                weight = F.pad(wv, (0, 0, 0, 0, 0, 0, 0, padn)).reshape(groups * (Cout_g + padn), *weight.shape[1:])
                if bias is not None:
                    bias = F.pad(bias.view(groups, Cout_g), (0, padn)).reshape(-1)
    if padn:
        y = Conv2dFunction.apply(x, weight, bias, stride, padding, dilation, groups, out_fp32, None, None, None, wpad)
        if keep_group_padding:       # [B, groups*(Cout_g+padn), OH, OW]: the caller consumes the padded layout (part_ops)
            return y
        if groups == 1:              # a view: channels stay at the epilogue's padded pixel stride (iuv_ops reads it as it is)
            yv = y[:, :Cout]
            yv._padded_base = y      # (iuv_ops.iuv_global differentiates through the padded tensor itself when it gets this view)
            return yv
        B, _, OH, OW = y.shape
        y = y.permute(0, 2, 3, 1).reshape(B, OH, OW, groups, Cout_g + padn)[..., :Cout_g]
        return y.reshape(B, OH, OW, Cout).permute(0, 3, 1, 2)
    sums = None
    if FUSE_BN_STATS and want_stats and bias is None and not out_fp32:
        n = _lib.lib().danet_bn_ws_floats(Cout)
        sums = ARENA.alloc(n)
        if sums is None:
            sums = torch.zeros(n, dtype=torch.float32, device=x.device)
    # the BatchNorm that produced x (if any) leaves its tensors on x: the data gradient then also reduces that
    # BatchNorm's backward sums (saves one pass over dy, x, y per BatchNorm with a single consumer)
    bn_ctx = getattr(x, '_bn_ctx', None) if (FUSE_BN_BWD_REDUCE and torch.is_grad_enabled()) else None
    if bn_ctx is None and FUSE_BN_BWD_STEM and torch.is_grad_enabled() and getattr(x, '_bn_ctx', None) is not None and x.dtype == torch.bfloat16:
        # the 7x7 stems over the part crops: the BatchNorm in front of them normalises a 403 MB tensor, far beyond the one-pass
        # backward's reach, so its backward is the two-kernel form (reduce 158 us + apply 209 us) -- unless the stem's data-gradient
        # kernel (csrc/conv_stem_dgrad.hip), which has the gradient tile in registers anyway, accumulates the two sums in its epilogue
        Bx, Cx, Hx, Wx = x.shape
        R_, S_ = weight.shape[2], weight.shape[3]
        st_, pd_, dl_ = int(stride), int(padding), int(dilation)
        if _lib.lib().danet_conv_stem_dgrad_ok(Bx, Hx, Wx, Cx, conv_out_size(Hx, R_, st_, pd_, dl_), conv_out_size(Wx, S_, st_, pd_, dl_),
                                                 Cout, R_, S_, st_, pd_, dl_, groups) and x._bn_ctx[0].shape == x.shape:
            bn_ctx = x._bn_ctx
    y = Conv2dFunction.apply(x, weight, bias, stride, padding, dilation, groups, out_fp32, sums, bn_ctx, link, wpad)
    if sums is not None:
        y._bn_sums = sums              # picked up by the BatchNorm2d that consumes y (nn.BatchNorm2d.forward)
    return y


BN_GATE_MODES = bool(int(os.environ.get('DANET_BN_GATE_MODES', '1')))    # A/B knob: 0 = the fused reduction always gates on the BN output


def _bn_gate(bn_ctx, x, c3):
    """Where a fused BatchNorm-backward reduction takes that BatchNorm's ReLU gate from: (tensor or None, bn_gate).
    bn_ctx = (bn_x, relu, saved, mask, mask_mode) as left on the BatchNorm's output by nn.BatchNorm2d; x is that
    output (the conv's input).  The LDS-tile 3x3 kernel reads the byte mask (1 byte instead of 8 per lane); the gather
    kernel reads the output."""
    relu = bn_ctx[1]
    mask, mode = (bn_ctx[3], bn_ctx[4]) if len(bn_ctx) > 3 else (None, 0)
    if not relu:
        return None, 0
    if c3 and BN_GATE_MODES and mask is not None:
        return mask, 2
    return x, 0


def _conv_job(job, x, wp, y, dims, transposed, bn_sums=None, bn_bwd=None, addend=None):
    (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups) = dims
    stream_tables(x.device)
    job.x, job.wp, job.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
    job.bn_sums = None if bn_sums is None else bn_sums.data_ptr()
    job.addend = None if addend is None else addend.data_ptr()
    if bn_bwd is None:
        job.bn_x = job.bn_y = job.bn_saved = job.bn_red = None
        job.bn_gate = 0
    else:
        job.bn_gate = bn_bwd[4]
        job.bn_x, job.bn_y = bn_bwd[0].data_ptr(), None if bn_bwd[1] is None else bn_bwd[1].data_ptr()
        job.bn_saved, job.bn_red = bn_bwd[2].data_ptr(), bn_bwd[3].data_ptr()
    (job.B, job.H, job.W, job.Cin, job.OH, job.OW, job.Cout, job.R, job.S, job.stride, job.pad, job.dil, job.groups) = dims
    job.transposed = int(transposed)


class MultiConvFunction(torch.autograd.Function):
    """n (<= 4) independent bias-free convolutions in ONE launch (forward) and one launch for their data gradients
    (csrc/conv_fast.hip conv_fast_multi_kernel); weight gradients go through the usual (deferred) path.
    Tensor arguments: xs[n], weights[n]; `static` = (n, [(stride, pad, dil, groups)], want_stats, bn_ctxs)."""

    @staticmethod
    def forward(ctx, static, *tensors):
        import ctypes
        L = _lib.lib()
        n, cfgs, want_stats, bn_ctxs, links = static[:5]
        bn = static[5] if len(static) > 5 else None      # the BatchNorms that follow (nn.multi_conv_bn): applied by this launch when it can
        xs = [nhwc_bf16(t) for t in tensors[:n]]
        ws = tensors[n:2 * n]
        jobs = (_lib.ConvJob * n)()
        ys, sums_l, dims_l, keep = [], [], [], []
        for i in range(n):
            stride, pad, dil, groups = cfgs[i]
            B, Cin, H, W = xs[i].shape
            Cout, Cin_g, R, S = ws[i].shape
            OH, OW = conv_out_size(H, R, stride, pad, dil), conv_out_size(W, S, stride, pad, dil)
            dims = (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups)
            wp = pack_weight(ws[i], groups, 0)
            y = _empty_nhwc(B, Cout, OH, OW, torch.bfloat16, xs[i].device)
            sums = None
            if want_stats and FUSE_BN_STATS:
                nfl = L.danet_bn_ws_floats(Cout)
                sums = ARENA.alloc(nfl)
                if sums is None:
                    sums = torch.zeros(nfl, dtype=torch.float32, device=y.device)
            _conv_job(jobs[i], xs[i], wp, y, dims, False, sums)
            ys.append(y); sums_l.append(sums); dims_l.append(dims); keep.append(wp)
        tok = None
        if PROFILER is not None:
            tok = PROFILER.begin(_multi_kernel_name(jobs, n, dims_l[0][6] // dims_l[0][12]),
                                 sum(2.0 * d[0] * d[4] * d[5] * d[6] * (d[3] // d[12]) * d[7] * d[8] for d in dims_l), ('fwd-multi', n))
        done = None
        if bn is not None and all(s_ is not None for s_ in sums_l):
            # conv -> BatchNorm (+ residual) (+ ReLU) in ONE launch when the streamed 3x3 kernel takes the set (csrc/conv3x3s.hip
            # s3_bn_tail), else the two launches from inside the same call: either way the BatchNorm's outputs exist afterwards and
            # nn.MultiBatchNormFunction finds them on the convolution's output (`_bn_done`)
            bjobs = (_lib.BnFwdJob * n)()
            done = []
            for i in range(n):
                spec = bn['jobs'][i]
                B, Cout, OH, OW = ys[i].shape
                res = None if spec['res'] is None else nhwc_as(spec['res'], torch.bfloat16)
                if res is not None and res.shape != ys[i].shape:
                    raise ValueError('residual shape %s != %s' % (tuple(res.shape), tuple(ys[i].shape)))
                out = _empty_nhwc(B, Cout, OH, OW, torch.bfloat16, ys[i].device)
                saved = torch.empty(2, Cout, dtype=torch.float32, device=ys[i].device)
                mask = torch.empty(B * OH * OW * Cout // 4, dtype=torch.uint8, device=ys[i].device) if spec['want_mask'] else None
                j = bjobs[i]
                j.x, j.res, j.y = ys[i].data_ptr(), None if res is None else res.data_ptr(), out.data_ptr()
                j.gamma, j.beta = spec['gamma'].data_ptr(), spec['beta'].data_ptr()
                j.running_mean = None if spec['running_mean'] is None else spec['running_mean'].data_ptr()
                j.running_var = None if spec['running_var'] is None else spec['running_var'].data_ptr()
                j.saved, j.sums, j.mask = saved.data_ptr(), sums_l[i].data_ptr(), None if mask is None else mask.data_ptr()
                j.M, j.C, j.sums_state, j.relu = B * OH * OW, Cout, 2, int(spec['relu'])
                done.append((out, saved, mask))
                keep += [res, spec['gamma'], spec['beta']]
            was_fused = ctypes.c_int(0)
            check(L.danet_conv_bn_forward_multi(ctypes.addressof(jobs), n, ctypes.addressof(bjobs), float(bn['momentum']), float(bn['eps']),
                                                ptr(bn['bar']), ctypes.addressof(was_fused), stream()), 'danet_conv_bn_forward_multi')
            FUSION['conv_bn_one_launch' if was_fused.value else 'conv_bn_two_launches'] += n
            if tok is not None and was_fused.value:          # (the record names the kernel that ran: convolution + BatchNorm tail)
                tok = (tok[0].replace('conv3x3_stream_kernel', 'conv3x3_stream_bn_kernel'),) + tuple(tok[1:])
        else:
            check(L.danet_conv_forward_multi(ctypes.addressof(jobs), n, stream()), 'danet_conv_forward_multi')
        if tok is not None:
            PROFILER.end(tok)
        if TRACE is not None:
            for y, d in zip(ys, dims_l):
                TRACE.append(('conv', (d[0], d[1], d[2], d[3], d[6], d[7], d[9]), y.float().abs().mean()))
        ctx.save_for_backward(*xs, *ws)
        ctx.cfg = (n, dims_l, bn_ctxs, links)
        if links is not None:
            for i, lk in enumerate(links):
                if lk is not None and ctx.needs_input_grad[1 + i]:
                    lk.armed = True
        for y, sums in zip(ys, sums_l):
            if sums is not None:
                y._bn_sums = sums
        if done is not None:
            for y, d in zip(ys, done):
                y._bn_done = d
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        import ctypes
        L = _lib.lib()
        n, dims_l, bn_ctxs, links = ctx.cfg
        sv = ctx.saved_tensors
        xs, ws = sv[:n], sv[n:2 * n]
        gys = [nhwc_bf16(g) for g in gys]
        gws = [None] * n
        for i in range(n):                      # weight gradients first: queued (deferred) or launched per layer
            if ctx.needs_input_grad[1 + n + i]:
                (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups) = dims_l[i]
                gws[i] = new_wgrad(ws[i], (Cout, Cin // groups, R, S), xs[i].device)
                _wgrad_into(gws[i], xs[i], gys[i], B, H, W, Cin, OH, OW, Cout, Cin // groups, R, S, stride, pad, dil, groups, ws[i])
        gxs = [None] * n
        need = [i for i in range(n) if ctx.needs_input_grad[1 + i]]
        if need:
            jobs = (_lib.ConvJob * len(need))()
            keep, reds, adds = [], [], []
            for k, i in enumerate(need):
                (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups) = dims_l[i]
                wp1 = pack_weight(ws[i], groups, 1)
                gx = _empty_nhwc(B, Cin, H, W, torch.bfloat16, xs[i].device)
                bn_bwd = None
                if FUSE_BN_BWD_REDUCE and bn_ctxs[i] is not None and bn_ctxs[i][0].shape == xs[i].shape:
                    bn_x, saved = bn_ctxs[i][0], bn_ctxs[i][2]
                    nfl = L.danet_bn_ws_floats(Cin)
                    red = ARENA.alloc(nfl)
                    if red is None:
                        red = torch.zeros(nfl, dtype=torch.float32, device=gx.device)
                    gate_t, gate = _bn_gate(bn_ctxs[i], xs[i], True)       # as for the LDS-tile 3x3 kernel; revised below if not
                    bn_bwd = (bn_x, gate_t, saved, red, gate)
                addend = None
                if links is not None and links[i] is not None and links[i].dres is not None:
                    addend, links[i].dres = links[i].dres, None
                    if addend.shape != xs[i].shape:
                        raise RuntimeError('residual gradient shape %s != %s' % (tuple(addend.shape), tuple(xs[i].shape)))
                adds.append(addend)
                # data gradient = the transposed gather: roles of (H, W, Cin) and (OH, OW, Cout) swap
                _conv_job(jobs[k], gys[i], wp1, gx, (B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, dil, groups), True, None, bn_bwd, addend)
                gxs[i] = gx
                reds.append(None if bn_bwd is None else bn_bwd[3])
                keep.append(wp1)
            ok = L.danet_conv_forward_multi_ok(ctypes.addressof(jobs), len(need))
            if ok != 2 and any(j.bn_gate for j in jobs):      # not the LDS-tile 3x3 kernel: its gate modes do not apply
                for k, i in enumerate(need):
                    if jobs[k].bn_gate:
                        jobs[k].bn_y, jobs[k].bn_gate = xs[i].data_ptr(), 0
                ok = L.danet_conv_forward_multi_ok(ctypes.addressof(jobs), len(need))
            if ok:
                tok = None
                if PROFILER is not None:
                    dd = [dims_l[i] for i in need]
                    tok = PROFILER.begin(_multi_kernel_name(jobs, len(need), dd[0][3] // dd[0][12]),
                                         sum(2.0 * d[0] * d[4] * d[5] * d[6] * (d[3] // d[12]) * d[7] * d[8] for d in dd), ('dgrad-multi', len(need)))
                check(L.danet_conv_forward_multi(ctypes.addressof(jobs), len(need), stream()), 'danet_conv_forward_multi')
                if tok is not None:
                    PROFILER.end(tok)
                for k, i in enumerate(need):
                    if reds[k] is not None:
                        gxs[i]._bn_red = reds[k]
                    if adds[k] is not None:
                        FUSION['residual_grad_fused'] += 1
            else:
                # The set as a whole does not qualify (the data gradients of a fuse-layer level mix output widths -- 48 / 96 / 192 channels =
                # different channel-block counts per workgroup -- and the 48-channel strided ones have no parity classes): round 6 launches
                # every SUBSET of equal channel-block count that qualifies as one multi-problem launch (the fuse chains' ~35 single
                # data-gradient launches per step were 0.7 ms) and only the rest per layer.
                left = list(range(len(need)))
                if MULTI_DGRAD_SUBSETS and len(need) > 1 and not any(a is not None for a in adds):
                    by_nt = {}
                    for k, i in enumerate(need):
                        d = dims_l[i]
                        by_nt.setdefault(int(L.danet_conv_nt(d[3] // d[12])), []).append(k)
                    for nt_, ks in by_nt.items():
                        # drop members that do not run on the lean gather kernel by themselves, then try the rest together
                        sub = [k for k in ks if L.danet_conv_forward_multi_ok(ctypes.addressof(jobs[k]), 1) == 1]
                        if len(sub) < 2:
                            continue
                        sj = (_lib.ConvJob * len(sub))(*[jobs[k] for k in sub])
                        if L.danet_conv_forward_multi_ok(ctypes.addressof(sj), len(sub)) != 1:
                            continue
                        tok = None
                        if PROFILER is not None:
                            dd = [dims_l[need[k]] for k in sub]
                            tok = PROFILER.begin(_multi_kernel_name(sj, len(sub), dd[0][3] // dd[0][12]),
                                                 sum(2.0 * d[0] * d[4] * d[5] * d[6] * (d[3] // d[12]) * d[7] * d[8] for d in dd), ('dgrad-multi', len(sub)))
                        check(L.danet_conv_forward_multi(ctypes.addressof(sj), len(sub), stream()), 'danet_conv_forward_multi')
                        if tok is not None:
                            PROFILER.end(tok)
                        FUSION['dgrad_subset_multi'] += len(sub)
                        for k in sub:
                            left.remove(k)
                            if reds[k] is not None:
                                gxs[need[k]]._bn_red = reds[k]
                for k in left:                      # per-layer launches
                    i = need[k]
                    (B, H, W, Cin, OH, OW, Cout, R, S, stride, pad, dil, groups) = dims_l[i]
                    gxs[i] = _conv_fwd_raw(gys[i], keep[k], None, B, OH, OW, Cout, H, W, Cin, R, S, stride, pad, dil, groups, True, False, False)
                    if adds[k] is not None:
                        FUSION['residual_grad_added'] += 1
                        gxs[i] = gxs[i] + adds[k]
            if TRACE is not None:
                for i in need:
                    d = dims_l[i]
                    TRACE.append(('dgrad', (d[0], d[4], d[5], d[6], d[3], d[7], d[9]), gxs[i].float().abs().mean()))
        return (None, *gxs, *gws)


def multi_conv(convs, xs, links=None, bn=None):
    """[conv(x) for conv, x in zip(convs, xs)] for up to 4 bias-free Conv2d modules in one launch per pass; falls back
    to the per-module path when the set does not qualify (channel padding, bias, mixed tile counts, fp32 outputs).
    bn (nn.multi_conv_bn): the training-mode BatchNorms that follow, for the same launch to apply (MultiConvFunction.forward)."""
    import ctypes
    n = len(convs)
    L = _lib.lib()
    ok = PRECISION != 'fp32' and 1 <= n <= 12 and xs[0].is_cuda and all(c.bias is None and not c.out_fp32 and c.stride[0] == c.stride[1] and
                                                c.padding[0] == c.padding[1] and c.dilation[0] == c.dilation[1] for c in convs)
    if ok:
        jobs = (_lib.ConvJob * n)()
        for j, c, x in zip(jobs, convs, xs):
            B, Cin, H, W = x.shape
            Cout, Cin_g, R, S = c.weight.shape
            if Cin_g * c.groups != Cin or Cin % 8 or (Cout // c.groups) % 8:
                ok = False
                break
            st, pd, dl = c.stride[0], c.padding[0], c.dilation[0]
            (j.B, j.H, j.W, j.Cin, j.OH, j.OW, j.Cout, j.R, j.S, j.stride, j.pad, j.dil, j.groups, j.transposed) = \
                (B, H, W, Cin, conv_out_size(H, R, st, pd, dl), conv_out_size(W, S, st, pd, dl), Cout, R, S, st, pd, dl, c.groups, 0)
        ok = ok and bool(L.danet_conv_forward_multi_ok(ctypes.addressof(jobs), n))
    if not ok:
        return [c(x, link=None if links is None else links[i]) for i, (c, x) in enumerate(zip(convs, xs))]
    grad = torch.is_grad_enabled()
    cfgs = [(c.stride[0], c.padding[0], c.dilation[0], c.groups) for c in convs]
    bn_ctxs = [getattr(x, '_bn_ctx', None) if (FUSE_BN_BWD_REDUCE and grad) else None for x in xs]
    static = (n, cfgs, all(c.training for c in convs), bn_ctxs, links, bn if (bn is not None and n <= 4 and all(c.training for c in convs)) else None)
    return list(MultiConvFunction.apply(static, *xs, *[c.weight for c in convs]))


class Conv2d(nn.Conv2d):
    """nn.Conv2d with the same parameters / state-dict keys, computed by the HIP MFMA kernels."""

    def __init__(self, *args, out_fp32=False, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.padding_mode == 'zeros'
        self.out_fp32 = out_fp32

    def forward(self, x, link=None):
        s, p, d = self.stride, self.padding, self.dilation
        if s[0] != s[1] or p[0] != p[1] or d[0] != d[1]:
            raise ValueError('danet Conv2d supports square stride/padding/dilation only')
        # in training the epilogue also accumulates the BatchNorm statistics of the output (consumed by the
        # following BatchNorm2d; ignored otherwise)
        return conv2d(x, self.weight, self.bias, s[0], p[0], d[0], self.groups, self.out_fp32, want_stats=self.training, link=link)
