"""SMPL-side losses of the regressor on HIP kernels (csrc/loss_ops.hip): /root/reference/models/danet/smpl_regressor.py:141-218
with its helpers :233-298 (joint_rotation*, joint_position*, keypoints_2d, keypoints_3d, smpl_pose, smpl_betas, smpl_verts,
cam) as one autograd op -- two launches forward, one backward -- instead of ~240 tensor-op launches.  The tensor-op
helpers of smpl_regressor.SMPL_Regressor remain as the CPU-checkable statement of the same arithmetic (both are pinned
against the reference's golden vectors g10)."""
import ctypes

import torch

from . import _lib
from ._lib import check, stream

NT = 10
KEYS = ('joint_rotation0', 'joint_rotation1', 'joint_position0', 'joint_position1', 'keypoints_2d', 'keypoints_3d',
        'smpl_pose', 'smpl_betas', 'smpl_verts', 'cam')


class _LossP(ctypes.Structure):
    _fields_ = [('para', ctypes.c_void_p), ('target', ctypes.c_void_p), ('jrot', ctypes.c_void_p * 2), ('jpos', ctypes.c_void_p * 2),
                ('gt_pts', ctypes.c_void_p), ('joints', ctypes.c_void_p), ('verts', ctypes.c_void_p), ('tverts', ctypes.c_void_p),
                ('kps2d', ctypes.c_void_p), ('kps3d', ctypes.c_void_p), ('has_smpl', ctypes.c_void_p), ('has_kp3d', ctypes.c_void_p),
                ('B', ctypes.c_int), ('V', ctypes.c_int), ('focal', ctypes.c_float), ('img', ctypes.c_float),
                ('op_w', ctypes.c_float), ('gt_w', ctypes.c_float), ('w', ctypes.c_float * NT), ('cnt', ctypes.c_float * NT)]


class _LossG(ctypes.Structure):
    _fields_ = [('dpara', ctypes.c_void_p), ('djrot', ctypes.c_void_p * 2), ('djpos', ctypes.c_void_p * 2),
                ('djoints', ctypes.c_void_p), ('dverts', ctypes.c_void_p)]


def _c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


class SmplLossFunction(torch.autograd.Function):
    """(para, jrot0, jrot1, jpos0, jpos1, joints, verts; constants) -> losses [10] in the order of KEYS (absent stages: 0)."""

    @staticmethod
    def forward(ctx, para, jrot0, jrot1, jpos0, jpos1, joints, verts, const):
        L = _lib.lib()
        assert L.danet_smpl_loss_param_bytes() == ctypes.sizeof(_LossP) and L.danet_smpl_loss_grad_bytes() == ctypes.sizeof(_LossG)
        target, gt_pts, tverts, kps2d, kps3d, has_smpl, has_kp3d, focal, img, op_w, gt_w, weights = const
        B = para.shape[0]
        ts = {'para': _c(para), 'target': _c(target), 'jrot0': _c(jrot0), 'jrot1': _c(jrot1), 'jpos0': _c(jpos0), 'jpos1': _c(jpos1),
              'gt_pts': _c(gt_pts), 'joints': _c(joints), 'verts': _c(verts) if weights[8] != 0 else None,
              'tverts': _c(tverts) if weights[8] != 0 else None, 'kps2d': _c(kps2d), 'kps3d': _c(kps3d),
              'has_smpl': _c(has_smpl), 'has_kp3d': _c(has_kp3d)}
        p = _LossP()
        dp = lambda t: None if t is None else t.data_ptr()      # noqa: E731
        p.para, p.target, p.gt_pts, p.joints = dp(ts['para']), dp(ts['target']), dp(ts['gt_pts']), dp(ts['joints'])
        p.jrot[0], p.jrot[1], p.jpos[0], p.jpos[1] = dp(ts['jrot0']), dp(ts['jrot1']), dp(ts['jpos0']), dp(ts['jpos1'])
        p.verts, p.tverts, p.kps2d, p.kps3d = dp(ts['verts']), dp(ts['tverts']), dp(ts['kps2d']), dp(ts['kps3d'])
        p.has_smpl, p.has_kp3d = dp(ts['has_smpl']), dp(ts['has_kp3d'])
        p.B, p.V = B, 0 if verts is None else verts.shape[1]
        p.focal, p.img, p.op_w, p.gt_w = float(focal), float(img), float(op_w), float(gt_w)
        V3 = 3.0 * (verts.shape[1] if verts is not None else 1)
        for k, (w, c) in enumerate(zip(weights, (216., 216., 1., 1., 98., 72., 216., 10., V3, 1.))):
            p.w[k], p.cnt[k] = float(w), float(c)
        dev = para.device
        ps = torch.empty(B, NT, dtype=torch.float32, device=dev)
        out = torch.empty(NT, dtype=torch.float32, device=dev)
        norm = torch.empty(NT, dtype=torch.float32, device=dev)
        check(L.danet_smpl_loss_forward(ctypes.addressof(p), ps.data_ptr(), out.data_ptr(), norm.data_ptr(), stream()), 'danet_smpl_loss_forward')
        ctx.p, ctx.keep, ctx.norm = p, ts, norm
        ctx.shapes = [None if t is None else t.shape for t in (para, jrot0, jrot1, jpos0, jpos1, joints, verts)]
        return tuple(out[i] for i in range(NT))          # ten 0-dim views: no select / select-backward launches downstream

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.lib()
        ts, p = ctx.keep, ctx.p
        dev = ctx.norm.device
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        gout = torch.stack([zero if g is None else g.to(torch.float32).reshape(()) for g in gouts])
        mk = lambda t: None if t is None else torch.empty_like(t)      # noqa: E731
        dpara, djr0, djr1, djp0, djp1, djo = mk(ts['para']), mk(ts['jrot0']), mk(ts['jrot1']), mk(ts['jpos0']), mk(ts['jpos1']), mk(ts['joints'])
        dve = mk(ts['verts'])
        g = _LossG()
        dp = lambda t: None if t is None else t.data_ptr()      # noqa: E731
        g.dpara, g.djoints, g.dverts = dp(dpara), dp(djo), dp(dve)
        g.djrot[0], g.djrot[1], g.djpos[0], g.djpos[1] = dp(djr0), dp(djr1), dp(djp0), dp(djp1)
        go = gout.to(torch.float32).contiguous()
        check(L.danet_smpl_loss_backward(ctypes.addressof(p), go.data_ptr(), ctx.norm.data_ptr(), ctypes.addressof(g), stream()), 'danet_smpl_loss_backward')
        res = []
        for t, shp in zip((dpara, djr0, djr1, djp0, djp1, djo, dve), ctx.shapes):
            res.append(None if (t is None or shp is None) else t.view(shp))
        return (*res, None)


def smpl_losses(para, joint_rotation, joint_position, joints, verts, target, gt_pts, tverts, kps2d, kps3d, has_smpl, has_kp3d,
                focal, img, op_w, gt_w, weights):
    """-> {loss name: 0-dim tensor}.  joint_rotation / joint_position: lists (<= 2 stages each) of [B,216] / [B,24,3];
    weights: dict with SMPL_POSE, JOINT_POSITION, PROJ_KPS, KPS3D, SMPL_BETAS, VERTS (the yaml weights)."""
    if not para.is_cuda:
        raise RuntimeError('danet_hip ops run on the GPU only (got a %s tensor); there is no CPU path' % para.device)
    if len(joint_rotation) > 2 or len(joint_position) > 2:
        raise ValueError('at most two regressor stages')
    jr = list(joint_rotation) + [None] * (2 - len(joint_rotation))
    jp = list(joint_position) + [None] * (2 - len(joint_position))
    w = (weights['SMPL_POSE'], weights['SMPL_POSE'], weights['JOINT_POSITION'], weights['JOINT_POSITION'], weights['PROJ_KPS'],
         weights['KPS3D'], weights['SMPL_POSE'], weights['SMPL_BETAS'], weights['VERTS'], 1.0)
    const = (target, gt_pts, tverts, kps2d, kps3d, has_smpl, has_kp3d, focal, img, op_w, gt_w, w)
    out = SmplLossFunction.apply(para, jr[0], jr[1], jp[0], jp[1], joints, verts if weights['VERTS'] != 0 else None, const)
    present = {'joint_rotation0': jr[0] is not None, 'joint_rotation1': jr[1] is not None, 'joint_position0': jp[0] is not None,
               'joint_position1': jp[1] is not None}
    return {k: out[i] for i, k in enumerate(KEYS) if present.get(k, True)}       # (out: tuple of ten 0-dim tensors)
