"""TEST INFRASTRUCTURE ONLY -- the CPU oracle.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``danet-densepose2smpl_amd``) never does.

* ``lbs_forward`` / ``lbs_backward``      -- C restatement of the SMPL layer
  (oracle/lbs_ref_impl.inc; follows /root/reference/models/smpl.py:27-46 + published smplx
  LBS).  PARITY UNPINNED (smplx + SMPL model file absent; pinned by analytic tests).
* ``raster_forward``                       -- C restatement of IUV_Renderer.verts2uvimg
  (oracle/raster_ref.c; follows /root/reference/utils/renderer.py:207-298 + published
  neural_renderer rule).  PARITY UNPINNED (neural_renderer absent; analytic tests).
* ``numpy_ref``                            -- numpy restatements of the torch-only helpers
  (utils/geometry.py, utils/iuvmap.py, utils/keypoints.py:334-394, utils/graph.py), pinned
  by golden vectors generated from the reference itself (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('lbs_ref.c', 'lbs_ref_impl.inc', 'raster_ref.c', 'Makefile')]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'liboracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _model_args(model):
    c = np.ascontiguousarray
    vt = c(model['v_template'], np.float32)
    V = vt.shape[0]
    sd = c(model['shapedirs'], np.float32).reshape(V * 3, -1)
    NB = sd.shape[1]
    pd = c(model['posedirs'], np.float32)
    jr = c(model['J_regressor'], np.float32)
    lw = c(model['lbs_weights'], np.float32)
    par = c(model['parents'], np.int32)
    jx = c(model['J_regressor_extra'], np.float32)
    lm = c(model['landmark_verts'], np.int32)
    keep = (vt, sd, pd, jr, lw, par, jx, lm)
    f, i = ctypes.c_float, ctypes.c_int32
    args = (_p(vt, f), _p(sd, f), _p(pd, f), _p(jr, f), _p(lw, f), _p(par, i), _p(jx, f), _p(lm, i))
    dims = (V, NB, lm.shape[0], jx.shape[0])
    return keep, args, dims


def lbs_forward(model, betas, pose, pose_is_rotmat, dtype=np.float64):
    """-> verts [B,V,3], joints54 [B,54,3] (24 posed joints, 21 landmarks, 9 extra)."""
    dtype = np.dtype(dtype)
    ct = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    fn = getattr(lib(), 'smpl_lbs_forward_ref_f64' if dtype == np.float64 else 'smpl_lbs_forward_ref_f32')
    betas = np.ascontiguousarray(betas, dtype)
    B = betas.shape[0]
    pose = np.ascontiguousarray(pose, dtype).reshape(B, -1)
    assert pose.shape[1] == (216 if pose_is_rotmat else 72)
    keep, margs, (V, NB, NL, NE) = _model_args(model)
    assert betas.shape[1] == NB
    verts = np.empty((B, V, 3), dtype)
    j54 = np.empty((B, 24 + NL + NE, 3), dtype)
    rc = fn(_p(betas, ct), _p(pose, ct), ctypes.c_int(int(pose_is_rotmat)), ctypes.c_int(B), *margs,
            ctypes.c_int(V), ctypes.c_int(NB), ctypes.c_int(NL), ctypes.c_int(NE), _p(verts, ct), _p(j54, ct))
    assert rc == 0
    return verts, j54


def lbs_backward(model, betas, rotmats, g_verts, g_joints54, dtype=np.float64):
    """-> g_betas [B,NB], g_rotmats [B,24,3,3]."""
    dtype = np.dtype(dtype)
    ct = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    fn = getattr(lib(), 'smpl_lbs_backward_ref_f64' if dtype == np.float64 else 'smpl_lbs_backward_ref_f32')
    betas = np.ascontiguousarray(betas, dtype)
    B = betas.shape[0]
    rot = np.ascontiguousarray(rotmats, dtype).reshape(B, 216)
    keep, margs, (V, NB, NL, NE) = _model_args(model)
    gv = None if g_verts is None else np.ascontiguousarray(g_verts, dtype)
    gj = None if g_joints54 is None else np.ascontiguousarray(g_joints54, dtype)
    gb = np.empty((B, NB), dtype)
    gr = np.empty((B, 24, 3, 3), dtype)
    rc = fn(_p(betas, ct), _p(rot, ct), ctypes.c_int(B), *margs,
            ctypes.c_int(V), ctypes.c_int(NB), ctypes.c_int(NL), ctypes.c_int(NE),
            _p(gv, ct), _p(gj, ct), _p(gb, ct), _p(gr, ct))
    assert rc == 0
    return gb, gr


def raster_forward(verts, cam, vert_mapping, faces, tex, focal, orig, S):
    """-> iuv [B,3,S,S] f32, face_idx [B,S,S] i32 (-1 bg), depth [B,S,S] f32 (inf bg)."""
    verts = np.ascontiguousarray(verts, np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    vm = np.ascontiguousarray(vert_mapping, np.int32)
    fc = np.ascontiguousarray(faces, np.int32)
    tx = np.ascontiguousarray(tex, np.float32)
    B, NV = verts.shape[0], verts.shape[1]
    out = np.empty((B, 3, S, S), np.float32)
    fidx = np.empty((B, S, S), np.int32)
    depth = np.empty((B, S, S), np.float32)
    f, i = ctypes.c_float, ctypes.c_int32
    rc = lib().iuv_raster_forward_ref(_p(verts, f), _p(cam, f), ctypes.c_int(B), ctypes.c_int(NV),
                                      _p(vm, i), ctypes.c_int(vm.shape[0]), _p(fc, i), _p(tx, f),
                                      ctypes.c_int(fc.shape[0]), ctypes.c_float(focal), ctypes.c_float(orig),
                                      ctypes.c_int(S), _p(out, f), _p(fidx, i), _p(depth, f))
    assert rc == 0
    return out, fidx, depth
