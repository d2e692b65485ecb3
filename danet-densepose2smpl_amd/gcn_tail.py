"""The regressor's graph tail as one launch per direction (csrc/gcn_tail.hip; /root/reference/models/danet/smpl_regressor.py:846-900).

`fused_tail(pred, rot_feats)` returns (joint_rotation[0], joint_position[0], joint_position[1], smpl_pose) for the training-mode
default configuration, or None when the configuration / batch is outside what the kernel covers (the caller then runs its torch
operations).  Parity: tests/test_gpu_gcn_tail.py compares values and every gradient with those operations."""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream

GCN_TAIL = bool(int(os.environ.get('DANET_GCN_TAIL', '1')))      # A-B knob


def _f32(t):
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _bn_of(gcn, i):
    return gcn.act[i][0]


def _layers(pred):
    """(GraphConv, BatchNorm1d) of the five graph convolutions in execution order."""
    out = [(pred.r2p_gcn.gc[0], _bn_of(pred.r2p_gcn, 0))]
    out += [(pred.refine_gcn.gc[i], _bn_of(pred.refine_gcn, i)) for i in range(3)]
    out.append((pred.p2r_gcn.gc[0], _bn_of(pred.p2r_gcn, 0)))
    return out


def applicable(pred, rot_feats):
    from .config import cfg
    if not (GCN_TAIL and rot_feats.is_cuda and pred.training and torch.is_grad_enabled()):
        return False
    if not (cfg.DANET.REFINEMENT.REFINE_ON and cfg.DANET.REFINEMENT.POS_INTERSUPV and cfg.DANET.JOINT_POSITION_WEIGHTS > 0):
        return False
    if rot_feats.dim() != 3 or tuple(rot_feats.shape[1:]) != (24, 128) or rot_feats.shape[0] > _lib.lib().danet_gcn_tail_max_batch():
        return False
    if pred.refine_gcn.num_layers != 3 or pred.r2p_gcn.num_layers != 1 or pred.p2r_gcn.num_layers != 1:
        return False
    dims = [(128, 128), (128, 256), (256, 256), (256, 128), (128, 128)]
    for (gc, bn), d in zip(_layers(pred), dims):
        if tuple(gc.weight.shape) != d or gc.bias is None or not isinstance(bn, torch.nn.BatchNorm1d) or bn.momentum is None \
                or not bn.affine or not bn.track_running_stats or bn.num_features != 24:
            return False
    bns = [bn for _, bn in _layers(pred)]
    if any(bn.momentum != bns[0].momentum or bn.eps != bns[0].eps for bn in bns):
        return False                                   # (the kernel takes one momentum / eps for the five BatchNorm1d modules)
    if any(p.dtype != torch.float32 for p in _params(pred)) or rot_feats.dtype != torch.float32:
        return False
    from . import nn as _nn
    return _nn._onepass_bar(rot_feats.device) is not None


def _params(pred):
    ps = []
    for gc, bn in _layers(pred):
        ps += [gc.weight, gc.bias, bn.weight, bn.bias]
    ps.append(pred.edge_importance)
    for i in range(2):
        ps += [pred.pose_regressors[i][1].weight, pred.pose_regressors[i][1].bias]
    for i in range(2):
        ps += [pred.coord_regressors[i][1].weight, pred.coord_regressors[i][1].bias]
    return ps


def _fill(a, x, ps, bufs):
    """Inputs of struct danet_gcn_tail_args; `ps` as _params lists them (already fp32 contiguous), bufs = the module's buffers."""
    a.x = x.data_ptr()
    for l in range(5):
        a.L[l].W, a.L[l].bias, a.L[l].gamma, a.L[l].beta = (ps[4 * l + j].data_ptr() for j in range(4))
    a.edge = ps[20].data_ptr()
    for i in range(2):
        a.Wp[i], a.bp[i] = ps[21 + 2 * i].data_ptr(), ps[22 + 2 * i].data_ptr()
        a.Wc[i], a.bc[i] = ps[25 + 2 * i].data_ptr(), ps[26 + 2 * i].data_ptr()
    a.A_r2p, a.A_p2r, a.A_mask, a.mean_pose = (b.data_ptr() for b in bufs)


class GcnTailFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, x, *params):
        from . import nn as _nn
        L = _lib.lib()
        B = x.shape[0]
        dev = x.device
        xc = _f32(x)
        ps = [_f32(p) for p in params]
        bufs = [_f32(pred.r2p_A[0]), _f32(pred.p2r_A[0]), _f32(pred.A_mask[0]), _f32(pred.mean_pose.reshape(-1))]
        ws = torch.empty(L.danet_gcn_tail_ws_floats(B), dtype=torch.float32, device=dev)
        jr0 = torch.empty(B, 216, dtype=torch.float32, device=dev)
        jp0 = torch.empty(B, 24, 3, dtype=torch.float32, device=dev)
        jp1 = torch.empty(B, 24, 3, dtype=torch.float32, device=dev)
        pose = torch.empty(B, 216, dtype=torch.float32, device=dev)
        bar = _nn._onepass_bar(dev)
        a = _lib.GcnTailArgs()
        _fill(a, xc, ps, bufs)
        bns = [bn for _, bn in _layers(pred)]
        for l, bn in enumerate(bns):
            a.L[l].running_mean, a.L[l].running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        a.ws, a.jr0, a.jp0, a.jp1, a.pose = ws.data_ptr(), jr0.data_ptr(), jp0.data_ptr(), jp1.data_ptr(), pose.data_ptr()
        a.bar, a.B, a.momentum, a.eps = bar.data_ptr(), B, float(bns[0].momentum), float(bns[0].eps)
        check(L.danet_gcn_tail_forward(ctypes.addressof(a), stream()), 'danet_gcn_tail_forward')
        _nn.onepass_watch(dev)                        # (outside a Trainer: a barrier that gave up is reported by the next launch)
        for bn in bns:                                    # torch's per-module counter (nn.BatchNorm1d.forward)
            if _nn.BatchNorm2d.count_batches:
                bn.num_batches_tracked.add_(1)
            else:
                _nn.BatchNorm2d._ran.append(bn.num_batches_tracked)
        ctx.save_for_backward(xc, ws, *ps)
        ctx.bufs = bufs
        ctx.bn = (float(bns[0].momentum), float(bns[0].eps))
        ctx.shapes = [p.shape for p in params]
        ctx.set_materialize_grads(False)
        return jr0, jp0, jp1, pose

    @staticmethod
    def backward(ctx, g_jr0, g_jp0, g_jp1, g_pose):
        from . import nn as _nn
        L = _lib.lib()
        xc, ws = ctx.saved_tensors[:2]
        ps = list(ctx.saved_tensors[2:])
        B = xc.shape[0]
        dev = xc.device
        bar = _nn._onepass_bar(dev)
        if bar is None:
            raise RuntimeError('gcn_tail backward: off the stream that owns the grid-barrier state (nn.ONEPASS_STREAM)')
        gs = [None if g is None else _f32(g) for g in (g_jr0, g_jp0, g_jp1, g_pose)]
        gx = torch.empty_like(xc)
        gp = [torch.empty_like(p) for p in ps]
        scratch = torch.empty(L.danet_gcn_tail_scratch_floats(B), dtype=torch.float32, device=dev)
        a = _lib.GcnTailArgs()
        _fill(a, xc, ps, ctx.bufs)
        a.ws = ws.data_ptr()
        a.g_jr0, a.g_jp0, a.g_jp1, a.g_pose = (None if g is None else g.data_ptr() for g in gs)
        a.gx = gx.data_ptr()
        for l in range(5):
            a.gW[l], a.gb[l], a.ggamma[l], a.gbeta[l] = (gp[4 * l + j].data_ptr() for j in range(4))
        a.gedge = gp[20].data_ptr()
        for i in range(2):
            a.gWp[i], a.gbp[i] = gp[21 + 2 * i].data_ptr(), gp[22 + 2 * i].data_ptr()
            a.gWc[i], a.gbc[i] = gp[25 + 2 * i].data_ptr(), gp[26 + 2 * i].data_ptr()
        a.scratch, a.bar, a.B = scratch.data_ptr(), bar.data_ptr(), B
        a.momentum, a.eps = ctx.bn
        check(L.danet_gcn_tail_backward(ctypes.addressof(a), stream()), 'danet_gcn_tail_backward')
        _nn.onepass_watch(dev)
        return (None, gx) + tuple(g.view(s) for g, s in zip(gp, ctx.shapes))


def fused_tail(pred, rot_feats):
    if not applicable(pred, rot_feats):
        return None
    from . import conv as _conv
    _conv.FUSION['gcn_tail'] += 1
    return GcnTailFunction.apply(pred, rot_feats, *_params(pred))
