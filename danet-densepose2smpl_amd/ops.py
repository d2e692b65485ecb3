"""torch.autograd.Function wrappers over the C ABI (include/danet_hip.h)."""
import os

import torch

from . import _lib
from ._lib import ptr, check, stream


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


LBS_ONE_LAUNCH = True       # the SMPL forward as one kernel launch (csrc/smpl_lbs.hip smpl_fused_fwd_kernel); False: prep -> main -> finalize
_LBS_TICKETS = {}
SMPL_BWD_FUSED = bool(int(os.environ.get('DANET_LBS_BWD_FUSED', '0')))    # the SMPL backward as ONE launch (smpl_fused_bwd_kernel); default: three launches (faster)


def lbs_ticket(device, words):
    """The fused forward's arrival counters: zeroed once, then owned (and reset) by the launches on ONE stream -- a buffer
    per (device, stream), so launches that could overlap never share one."""
    if not LBS_ONE_LAUNCH or device.type != 'cuda':          # (CPU tensors: the C-ABI call below refuses them -- there is no CPU path)
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    t = _LBS_TICKETS.get(key)
    if t is None or t.numel() < words:
        with torch.cuda.device(key[0]):
            t = _LBS_TICKETS[key] = torch.zeros(max(1024, int(words)), dtype=torch.int32, device=device)
    return t


class SmplLbsFunction(torch.autograd.Function):
    """(betas [B,NB], rotmats [B,24,3,3], model buffers) -> (vertices [B,V,3], joints54 [B,54,3]).
    Gradients flow to betas and rotmats (/root/reference/models/danet/smpl_regressor.py:176)."""

    @staticmethod
    def forward(ctx, betas, rotmats, m):
        L = _lib.lib()
        betas_c, rot_c = _f32c(betas), _f32c(rotmats).view(-1, 24, 3, 3)
        B, NB = betas_c.shape
        V = m.v_template.shape[0]
        NL, NE = m.landmark_verts.numel(), m.J_regressor_extra.shape[0]
        dev = betas_c.device
        need_grad = betas.requires_grad or rotmats.requires_grad
        verts = torch.empty(B, V, 3, device=dev, dtype=torch.float32)
        j54 = torch.empty(B, 24 + NL + NE, 3, device=dev, dtype=torch.float32)
        cbuf = torch.empty(L.danet_smpl_lbs_ctx_floats(B), device=dev, dtype=torch.float32)
        vposed = torch.empty(B, V, 3, device=dev, dtype=torch.float32) if need_grad else None
        nws = L.danet_smpl_lbs_fwd_ws_floats(B, V, NE)
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        ticket = lbs_ticket(dev, L.danet_smpl_lbs_ticket_words(B))
        check(L.danet_smpl_lbs_forward(
            ptr(betas_c), ptr(rot_c), B, ptr(m.v_template), ptr(m.shapedirs), ptr(m.posedirs),
            ptr(m.J_template), ptr(m.J_shapedirs), ptr(m.lbs_weights), ptr(m.parents),
            ptr(m.J_regressor_extra), ptr(m.landmark_verts), V, NB, NL, NE,
            ptr(verts), ptr(j54), ptr(cbuf), ptr(vposed), ptr(ws), nws, ptr(ticket), stream()), 'danet_smpl_lbs_forward')
        if need_grad:
            ctx.m = m
            ctx.save_for_backward(betas_c, rot_c, cbuf, vposed)
        return verts, j54

    @staticmethod
    def backward(ctx, g_verts, g_j54):
        L = _lib.lib()
        m = ctx.m
        betas_c, rot_c, cbuf, vposed = ctx.saved_tensors
        B, NB = betas_c.shape
        V = m.v_template.shape[0]
        NL, NE = m.landmark_verts.numel(), m.J_regressor_extra.shape[0]
        dev = betas_c.device
        gv = None if g_verts is None else _f32c(g_verts)
        gj = None if g_j54 is None else _f32c(g_j54)
        g_betas = torch.empty(B, NB, device=dev, dtype=torch.float32)
        g_rot = torch.empty(B, 24, 3, 3, device=dev, dtype=torch.float32)
        nws = L.danet_smpl_lbs_bwd_ws_floats(B, V, NB)
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        # Three launches by default (126 us at B = 32 against 150 for the one-launch form, bench.py roofline_extra of round 5; neutral in
        # the step).  SMPL_BWD_FUSED: ONE launch when the grid fits the co-residency budget; the barrier state is the one-pass BatchNorm
        # backward's (same stream, never concurrent; its error word guards the optimizer step inside a Trainer -- outside one nothing
        # reads it, which is why the one-launch form is opt-in)
        from . import nn as _nn, conv as _conv
        bar = _nn._onepass_bar(dev) if SMPL_BWD_FUSED else None
        if bar is not None and L.danet_smpl_lbs_backward_fused_ok(B, V, _nn.onepass_budget(dev)):
            _conv.FUSION['smpl_bwd_fused'] += 1
        else:
            bar = None
        check(L.danet_smpl_lbs_backward(
            ptr(betas_c), ptr(rot_c), B, ptr(m.shapedirs), ptr(m.posedirs), ptr(m.J_shapedirs),
            ptr(m.lbs_weights), ptr(m.parents), ptr(m.J_regressor_extra), ptr(m.landmark_verts),
            V, NB, NL, NE, ptr(cbuf), ptr(vposed), ptr(gv), ptr(gj), ptr(g_betas), ptr(g_rot),
            ptr(ws), nws, ptr(bar), int(_nn.onepass_budget(dev)), stream()), 'danet_smpl_lbs_backward')
        return g_betas, g_rot, None


class SmplJointsFunction(torch.autograd.Function):
    """joints54 [B,54,3] -> (joints49 = j54[:, map49], joints_J19 = joints49[:, -24:][:, map19], smpl_joints = j54[:, :24]) in one launch,
    and their three gradients back into one (/root/reference/models/smpl.py:31-37: three index ops, their scatters and two accumulations)."""

    @staticmethod
    def forward(ctx, j54, map49, map19):
        L = _lib.lib()
        j = _f32c(j54)
        B, NJ = j.shape[0], j.shape[1]
        N49, N19 = map49.numel(), map19.numel()
        j49 = torch.empty(B, N49, 3, device=j.device, dtype=torch.float32)
        j19 = torch.empty(B, N19, 3, device=j.device, dtype=torch.float32)
        j24 = torch.empty(B, 24, 3, device=j.device, dtype=torch.float32)
        check(L.danet_smpl_joints_forward(ptr(j), ptr(map49), ptr(map19), B, NJ, N49, N19, ptr(j49), ptr(j19), ptr(j24), stream()), 'danet_smpl_joints_forward')
        ctx.maps = (map49, map19, B, NJ, N49, N19)
        ctx.set_materialize_grads(False)
        return j49, j19, j24

    @staticmethod
    def backward(ctx, g49, g19, g24):
        L = _lib.lib()
        map49, map19, B, NJ, N49, N19 = ctx.maps
        if g49 is None and g19 is None and g24 is None:
            return None, None, None
        f = lambda g: None if g is None else _f32c(g)      # noqa: E731
        g49, g19, g24 = f(g49), f(g19), f(g24)
        g54 = torch.empty(B, NJ, 3, device=map49.device, dtype=torch.float32)
        check(L.danet_smpl_joints_backward(ptr(g49), ptr(g19), ptr(g24), ptr(map49), ptr(map19), B, NJ, N49, N19, ptr(g54), stream()), 'danet_smpl_joints_backward')
        return g54, None, None


def smpl_joints(j54, map49, map19):
    return SmplJointsFunction.apply(j54, map49, map19)


def smpl_lbs(betas, rotmats, model):
    return SmplLbsFunction.apply(betas, rotmats, model)


def iuv_raster(verts, cam, vert_mapping, faces, tex, focal, orig, out_size, return_aux=False):
    """Forward-only (labels are rendered from detached meshes, danet.py:163-165)."""
    L = _lib.lib()
    v, c = _f32c(verts), _f32c(cam)
    B, NV = v.shape[0], v.shape[1]
    S = int(out_size)
    out = torch.empty(B, 3, S, S, device=v.device, dtype=torch.float32)
    fidx = torch.empty(B, S, S, device=v.device, dtype=torch.int32) if return_aux else None
    depth = torch.empty(B, S, S, device=v.device, dtype=torch.float32) if return_aux else None
    nws = L.danet_iuv_raster_ws_bytes(B, vert_mapping.numel(), S)
    ws = torch.empty((nws + 7) // 8, device=v.device, dtype=torch.int64)
    check(L.danet_iuv_raster_forward(ptr(v), ptr(c), B, NV, ptr(vert_mapping), vert_mapping.numel(),
                                     ptr(faces), ptr(tex), faces.shape[0], float(focal), float(orig), S,
                                     ptr(out), ptr(fidx), ptr(depth), ptr(ws), nws, stream()), 'danet_iuv_raster_forward')
    return (out, fidx, depth) if return_aux else out


def _rodrigues(theta, which):
    L = _lib.lib()
    if theta.requires_grad:
        raise RuntimeError('%s is forward-only (the reference only applies it to labels)' % which)
    th = _f32c(theta).view(-1, 3)
    R = torch.empty(th.shape[0], 3, 3, device=th.device, dtype=torch.float32)
    check(getattr(L, which)(ptr(th), th.shape[0], ptr(R), stream()), which)
    return R


def batch_rodrigues(theta):
    """/root/reference/utils/geometry.py:9-23."""
    return _rodrigues(theta, 'danet_batch_rodrigues')


def rodrigues_smplx(theta):
    return _rodrigues(theta, 'danet_rodrigues_smplx')


class Rot6dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        xc = _f32c(x).view(-1, 6)
        R = torch.empty(xc.shape[0], 3, 3, device=xc.device, dtype=torch.float32)
        check(L.danet_rot6d_to_rotmat_forward(ptr(xc), xc.shape[0], ptr(R), stream()), 'danet_rot6d_to_rotmat_forward')
        ctx.save_for_backward(xc)
        ctx.in_shape = x.shape
        return R

    @staticmethod
    def backward(ctx, gR):
        L = _lib.lib()
        (xc,) = ctx.saved_tensors
        g = _f32c(gR)
        gx = torch.empty_like(xc)
        check(L.danet_rot6d_to_rotmat_backward(ptr(xc), ptr(g), xc.shape[0], ptr(gx), stream()), 'danet_rot6d_to_rotmat_backward')
        return gx.view(ctx.in_shape)


def rot6d_to_rotmat(x):
    """/root/reference/utils/geometry.py:47-61."""
    return Rot6dFunction.apply(x)
