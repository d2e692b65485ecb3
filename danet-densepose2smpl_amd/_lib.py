"""ctypes binding of libdanet_hip.so (the C ABI declared in include/danet_hip.h).

There is no CPU fallback: every op of this package runs as a HIP kernel on gfx950, and using
one without the built library (or on a CPU tensor) raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DANET_LIB') or os.path.join(_HERE, 'csrc', 'libdanet_hip.so')
_lib = None

c_f = ctypes.c_void_p      # device pointers travel as void*
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_fl = ctypes.c_float

_SIGNATURES = {
    'danet_version': (c_i, []),
    'danet_last_error': (ctypes.c_char_p, []),
    'danet_smpl_lbs_debug': (c_i, [c_f]),
    'danet_smpl_lbs_ctx_floats': (c_sz, [c_i]),
    'danet_smpl_lbs_fwd_ws_floats': (c_sz, [c_i, c_i, c_i]),
    'danet_smpl_lbs_bwd_ws_floats': (c_sz, [c_i, c_i, c_i]),
    'danet_smpl_lbs_ticket_words': (c_sz, [c_i]),
    'danet_smpl_lbs_forward': (c_i, [c_f, c_f, c_i] + [c_f] * 9 + [c_i] * 4 + [c_f] * 5 + [c_sz, c_f, c_f]),
    'danet_smpl_lbs_backward': (c_i, [c_f, c_f, c_i] + [c_f] * 7 + [c_i] * 4 + [c_f] * 7 + [c_sz, c_f, c_i, c_f]),
    'danet_smpl_lbs_backward_fused_ok': (c_i, [c_i, c_i, c_i]),
    'danet_iuv_raster_ws_bytes': (c_sz, [c_i, c_i, c_i]),
    'danet_iuv_raster_forward': (c_i, [c_f, c_f, c_i, c_i, c_f, c_i, c_f, c_f, c_i, c_fl, c_fl, c_i, c_f, c_f, c_f, c_f, c_sz, c_f]),
    'danet_conv_nt': (c_i, [c_i]),
    'danet_conv_kernel_id': (c_i, [c_i] * 6),
    'danet_conv_wgrad_kernel_id': (c_i, [c_i] * 4),
    'danet_conv_packed_elems': (c_sz, [c_i] * 7),
    'danet_conv_pack_weights': (c_i, [c_f, c_f] + [c_i] * 7 + [c_f]),
    'danet_conv_pack_weights_padded': (c_i, [c_f, c_f] + [c_i] * 9 + [c_f]),
    'danet_conv_pack_job_fill_padded': (ctypes.c_long, [c_f, c_f, c_f, ctypes.c_long, ctypes.c_long] + [c_i] * 9),
    'danet_conv_pack_job_bytes': (c_sz, []),
    'danet_conv_pack_job_fill': (ctypes.c_long, [c_f, c_f, c_f, ctypes.c_long, ctypes.c_long] + [c_i] * 7),
    'danet_conv_f32': (c_i, [c_i, c_f, c_f, c_f, c_f] + [c_i] * 13 + [c_f]),
    'danet_conv_f32m_packed_elems': (c_sz, [c_i] * 6),
    'danet_conv_f32m_pack_weights': (c_i, [c_f, c_f] + [c_i] * 8 + [c_f]),
    'danet_conv_f32m_ok': (c_i, [c_i] * 14),
    'danet_conv_f32m_forward': (c_i, [c_f, c_f, c_f, c_f] + [c_i] * 15 + [c_f]),
    'danet_conv_f32m_wgrad_ws_floats': (c_sz, [c_i] * 15),
    'danet_conv_f32m_wgrad': (c_i, [c_f, c_f, c_f, c_f] + [c_i] * 15 + [c_f]),
    'danet_conv3x3_debug': (None, [c_f]),
    'danet_conv3x3_stream_plan': (c_i, [c_i] * 6),
    'danet_conv_stem_ok': (c_i, [c_i] * 13),
    'danet_conv_stem_forward': (c_i, [c_f, c_f, c_f] + [c_i] * 7 + [c_f, c_f]),
    'danet_conv_stem_dgrad_ok': (c_i, [c_i] * 13),
    'danet_conv_stem_dgrad': (c_i, [c_f, c_f, c_f] + [c_i] * 7 + [c_f] * 5),
    'danet_conv3x3a_ok': (c_i, [c_i] * 11),
    'danet_conv3x3a': (c_i, [c_f, c_f, c_f] + [c_i] * 4 + [c_f] * 5 + [c_i, c_f, c_f]),
    'danet_conv3x3_stream_table_bytes': (c_sz, []),
    'danet_conv3x3_stream_tables': (c_i, [c_f, c_sz]),
    'danet_conv_pack_weights_batched': (c_i, [c_f, c_i, ctypes.c_long, ctypes.c_long, c_f]),
    'danet_conv_pack_job_bricks': (ctypes.c_long, [c_i] * 7),
    'danet_conv_forward_multi_ok': (c_i, [c_f, c_i]),
    'danet_conv_forward_multi_kernel': (c_i, [c_f, c_i]),
    'danet_conv_forward_multi': (c_i, [c_f, c_i, c_f]),
    'danet_conv_bn_forward_multi_ok': (c_i, [c_f, c_i, c_f, c_f]),
    'danet_conv_bn_forward_multi': (c_i, [c_f, c_i, c_f, c_fl, c_fl, c_f, c_f, c_f]),
    'danet_conv_forward_kernel': (c_i, [c_i] * 15),
    'danet_conv_forward': (c_i, [c_f] * 4 + [c_i] * 16 + [c_f] * 6 + [c_i, c_f]),
    'danet_conv_wgrad_rows_ok': (c_i, [c_i] * 13),
    'danet_conv_wgrad_rows_ws_floats': (c_sz, [c_i] * 8),
    'danet_conv_wgrad_rows': (c_i, [c_f] * 4 + [c_sz] + [c_i] * 12 + [c_fl, c_f]),
    'danet_conv_wgrad3x3_ok': (c_i, [c_i] * 10),
    'danet_conv_wgrad3x3_pair_ok': (c_i, [c_i] * 11),
    'danet_conv_wgrad3x3_ws_floats': (c_sz, [c_i] * 7),
    'danet_conv_wgrad_multi_ws_floats': (c_sz, [c_f, c_i]),
    'danet_conv_wgrad_multi_ws_zero_from': (c_sz, [c_f, c_i]),
    'danet_conv_wgrad_multi': (c_i, [c_f, c_i, c_f, c_sz, c_fl, c_f]),
    'danet_conv_wgrad3x3_multi_ws_floats': (c_sz, [c_f, c_i]),
    'danet_conv_wgrad3x3_multi': (c_i, [c_f, c_i, c_f, c_sz, c_fl, c_f]),
    'danet_conv_wgrad3x3_kernel_id': (c_i, [c_i] * 7),
    'danet_conv_wgrad3x3': (c_i, [c_f] * 4 + [c_sz] + [c_i] * 7 + [c_fl, c_i, c_f]),
    'danet_conv_wgrad_ws_floats': (c_sz, [c_i] * 4),
    'danet_conv_wgrad_ws_floats_for': (c_sz, [c_i] * 13),
    'danet_conv_wgrad': (c_i, [c_f] * 4 + [c_sz] + [c_i] * 13 + [c_fl, c_i, c_f]),
    'danet_bn_forward': (c_i, [c_f, c_f, c_f, ctypes.c_int64, c_i] + [c_f] * 6 + [c_i, c_fl, c_fl, c_i, c_i, c_f, c_f]),
    'danet_bn_ws_floats': (c_sz, [c_i]),
    'danet_knob': (ctypes.c_long, [c_i, ctypes.c_long]),
    'danet_bn_acc_bytes': (c_i, []),
    'danet_channel_sum': (c_i, [c_f, ctypes.c_int64, c_i, c_f, c_i, c_i, c_f]),
    'danet_bn_forward_multi': (c_i, [c_f, c_i, c_fl, c_fl, c_f]),
    'danet_bn_backward_multi': (c_i, [c_f, c_i, c_f]),
    'danet_bn_backward_onepass_ok': (c_i, [c_f, c_i, c_i]),
    'danet_bn_backward_onepass_bar_words': (c_i, []),
    'danet_bn_backward_onepass': (c_i, [c_f, c_i, c_f, c_i, c_f]),
    'danet_bn_backward': (c_i, [c_f, c_f, c_f, ctypes.c_int64, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f]),
    'danet_sum_relu_forward': (c_i, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_i), c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    'danet_sum_relu_backward': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    'danet_sum_relu_backward_all': (c_i, [c_f, c_f] + [c_i] * 5 + [c_f] * 5),
    'danet_sum_relu_forward_multi': (c_i, [c_f, c_i, c_f]),
    'danet_sum_relu_backward_all_multi': (c_i, [c_f, c_i, c_f]),
    'danet_maxpool3x3s2_forward': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'danet_maxpool3x3s2_backward': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'danet_stn_gather_forward': (c_i, [c_f, c_f] + [c_i] * 8 + [c_f, c_f]),
    'danet_stn_gather_backward': (c_i, [c_f, c_f] + [c_i] * 8 + [c_f, c_f]),
    'danet_part_clean_forward': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    'danet_part_clean_backward': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    'danet_part_loss_forward': (c_i, [c_f] * 5 + [c_i] * 5 + [c_f, c_f]),
    'danet_part_loss_backward': (c_i, [c_f] * 6 + [c_i] * 5 + [c_f, c_f]),
    'danet_part_backward_fused': (c_i, [c_f] * 8 + [c_i] * 5 + [c_f, c_f]),
    'danet_iuv_global_forward': (c_i, [c_f] * 4 + [c_i, c_i] + [c_f] * 3 + [c_i] * 4 + [c_f] * 5),
    'danet_iuv_global_backward': (c_i, [c_f] * 4 + [c_i, c_i] + [c_f] * 6 + [c_i] * 4 + [c_f] * 5),
    'danet_softargmax_forward': (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, c_f]),
    'danet_softargmax_backward': (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, c_f, c_f]),
    'danet_pad_multi': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f]),
    'danet_loss_finalize': (c_i, [c_f, c_i, c_i, c_f, c_f, c_f, c_i, c_f, c_f, c_f]),
    'danet_smpl_joints_forward': (c_i, [c_f, c_f, c_f] + [c_i] * 4 + [c_f] * 4),
    'danet_smpl_joints_backward': (c_i, [c_f] * 5 + [c_i] * 4 + [c_f, c_f]),
    'danet_stn_theta_forward': (c_i, [c_f] * 8 + [c_i] * 4 + [c_fl, c_fl, c_f, c_f]),
    'danet_smpl_loss_param_bytes': (c_sz, []),
    'danet_smpl_loss_grad_bytes': (c_sz, []),
    'danet_smpl_loss_forward': (c_i, [c_f] * 5),
    'danet_smpl_loss_backward': (c_i, [c_f] * 5),
    'danet_adam_chunk_bytes': (c_sz, []),
    'danet_adam_step': (c_i, [c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_fl, c_fl, c_fl, c_fl, c_f, c_f, c_f]),
    'danet_batch_rodrigues': (c_i, [c_f, c_i, c_f, c_f]),
    'danet_rodrigues_smplx': (c_i, [c_f, c_i, c_f, c_f]),
    'danet_rot6d_to_rotmat_forward': (c_i, [c_f, c_i, c_f, c_f]),
    'danet_rot6d_to_rotmat_backward': (c_i, [c_f, c_f, c_i, c_f, c_f]),
    'danet_regroup_parts': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    'danet_pack_image': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    'danet_gcn_tail_ws_floats': (c_sz, [c_i]),
    'danet_gcn_tail_scratch_floats': (c_sz, [c_i]),
    'danet_gcn_tail_max_batch': (c_i, []),
    'danet_gcn_tail_debug': (c_i, [c_f]),
    'danet_gcn_tail_forward': (c_i, [c_f, c_f]),
    'danet_gcn_tail_backward': (c_i, [c_f, c_f]),
}

# fp32 instantiations (csrc/norm_act_f32.hip, stn.hip): same arguments, fp32 NHWC activations
for _n in ('danet_bn_forward', 'danet_bn_backward', 'danet_bn_forward_multi', 'danet_bn_backward_multi', 'danet_sum_relu_forward',
           'danet_sum_relu_backward', 'danet_sum_relu_backward_all', 'danet_sum_relu_forward_multi', 'danet_sum_relu_backward_all_multi', 'danet_stn_gather_forward', 'danet_stn_gather_backward',
           'danet_maxpool3x3s2_forward', 'danet_maxpool3x3s2_backward', 'danet_channel_sum'):
    _SIGNATURES[_n + '_f32'] = _SIGNATURES[_n]


class SumFwdJob(ctypes.Structure):
    """One output of danet_sum_relu_forward_multi (include/danet_hip.h)."""
    _fields_ = [('terms', ctypes.c_void_p * 4), ('shifts', c_i * 4), ('nterms', c_i), ('B', c_i), ('H', c_i), ('W', c_i), ('C', c_i), ('relu', c_i),
                ('y', ctypes.c_void_p)]


class SumBwdJob(ctypes.Structure):
    """One output's gradients of danet_sum_relu_backward_all_multi."""
    _fields_ = [('gy', ctypes.c_void_p), ('y', ctypes.c_void_p), ('B', c_i), ('H', c_i), ('W', c_i), ('C', c_i), ('relu', c_i), ('d', ctypes.c_void_p * 4)]


class Wg3Job(ctypes.Structure):
    """One problem of danet_conv_wgrad3x3_multi (include/danet_hip.h)."""
    _fields_ = [('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p),
                ('B', c_i), ('H', c_i), ('W', c_i), ('Cin', c_i), ('Cout', c_i), ('groups', c_i), ('stride', c_i)]


class WgJob(ctypes.Structure):
    """One problem of danet_conv_wgrad_multi (include/danet_hip.h)."""
    _fields_ = [('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p)] + \
               [(k, c_i) for k in ('B', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'R', 'S', 'stride', 'pad', 'dil', 'groups')]


class BnFwdJob(ctypes.Structure):
    """One tensor of danet_bn_forward_multi (csrc/norm_act.hip)."""
    _fields_ = [(k, ctypes.c_void_p) for k in ('x', 'res', 'y', 'gamma', 'beta', 'running_mean', 'running_var', 'saved', 'sums', 'mask')] + \
               [('M', ctypes.c_int64), ('C', c_i), ('sums_state', c_i), ('relu', c_i)]


class BnBwdJob(ctypes.Structure):
    """One tensor of danet_bn_backward_multi."""
    _fields_ = [(k, ctypes.c_void_p) for k in ('dy', 'x', 'y', 'gamma', 'saved', 'dx', 'dres', 'dparam', 'red', 'beta', 'mask')] + \
               [('M', ctypes.c_int64), ('C', c_i), ('red_state', c_i), ('relu', c_i), ('mask_mode', c_i)]


class GcnLayer(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ('W', 'bias', 'gamma', 'beta', 'running_mean', 'running_var')]


class GcnTailArgs(ctypes.Structure):
    """struct danet_gcn_tail_args (include/danet_hip.h)."""
    _fields_ = [('x', ctypes.c_void_p), ('L', GcnLayer * 5)] + \
               [(k, ctypes.c_void_p) for k in ('A_r2p', 'A_p2r', 'A_mask', 'edge')] + \
               [('Wp', ctypes.c_void_p * 2), ('bp', ctypes.c_void_p * 2), ('Wc', ctypes.c_void_p * 2), ('bc', ctypes.c_void_p * 2)] + \
               [(k, ctypes.c_void_p) for k in ('mean_pose', 'ws', 'jr0', 'jp0', 'jp1', 'pose', 'g_jr0', 'g_jp0', 'g_jp1', 'g_pose', 'gx')] + \
               [('gW', ctypes.c_void_p * 5), ('gb', ctypes.c_void_p * 5), ('ggamma', ctypes.c_void_p * 5), ('gbeta', ctypes.c_void_p * 5),
                ('gedge', ctypes.c_void_p),
                ('gWp', ctypes.c_void_p * 2), ('gbp', ctypes.c_void_p * 2), ('gWc', ctypes.c_void_p * 2), ('gbc', ctypes.c_void_p * 2),
                ('scratch', ctypes.c_void_p), ('bar', ctypes.c_void_p), ('B', c_i), ('momentum', ctypes.c_float), ('eps', ctypes.c_float)]


class ConvJob(ctypes.Structure):
    """One problem of danet_conv_forward_multi (include/danet_hip.h)."""
    _fields_ = [(k, ctypes.c_void_p) for k in ('x', 'wp', 'y', 'bn_sums', 'bn_x', 'bn_y', 'bn_saved', 'bn_red', 'addend')] + \
               [(k, c_i) for k in ('B', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'R', 'S', 'stride', 'pad', 'dil', 'groups', 'transposed', 'bn_gate')]


def exported_symbols():
    return list(_SIGNATURES)


# include/danet_hip.h DANET_KNOB_*: the library's only run-time switches (A-B timing, tests)
KNOBS = {'c3_enable': 1, 'c3_mt': 2, 'c3_kw': 3, 'c3_blocks': 4, 'c3_want': 5, 'c3s_enable': 6, 'c3s_blocks': 7, 'c3s_kw': 8, 'c3s_want': 9,
         'pw': 10, 'pw_wgrad': 11, 'stem': 12, 'stem_dgrad': 13, 'c3a': 14, 'bn_block_bytes': 15,
         'c3s_balance': 16, 'c3s_tile_cost': 17, 'g3': 18}


class _Library(object):
    """The loaded libdanet_hip.so (attribute access = its C entry points, include/danet_hip.h) plus the host-side spellings of
    the switch board: `knob(name, value)` and the multi-argument setters tests and tools have always called, all of them thin
    wrappers of danet_knob.  Prefer `knobs(...)` below: a context manager cannot leak a flipped switch."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        return getattr(self._cdll, name)

    def knob(self, name, value=-1):
        """Set (value >= 0; bn_block_bytes: > 0) or query (value < 0) a switch; returns the previous value."""
        return int(self._cdll.danet_knob(KNOBS[name], int(value)))

    # (enable, ...) with -1 / <= 0 = keep, returning the previous `enable`: the round-1..4 signatures
    def danet_conv3x3_set(self, enable, force_mt, force_kw, blocks, want_tiles):
        prev = self.knob('c3_enable', enable)
        if force_mt >= 0 and force_kw >= 0:
            self.knob('c3_mt', force_mt)
            self.knob('c3_kw', force_kw)
        self.knob('c3_blocks', blocks if blocks > 0 else -1)
        self.knob('c3_want', want_tiles)
        return prev

    def danet_conv3x3_stream_set(self, enable, blocks, kw, want_tiles):
        prev = self.knob('c3s_enable', enable)
        self.knob('c3s_blocks', blocks if blocks > 0 else -1)
        self.knob('c3s_kw', kw)
        self.knob('c3s_want', want_tiles)
        return prev

    def danet_conv_pw_set(self, enable):
        return self.knob('pw', enable)

    def danet_conv_pw_wgrad_set(self, enable):
        return self.knob('pw_wgrad', enable)

    def danet_conv_stem_set(self, enable):
        return self.knob('stem', enable)

    def danet_conv_stem_dgrad_set(self, enable):
        return self.knob('stem_dgrad', enable)

    def danet_conv3x3a_set(self, enable):
        return self.knob('c3a', enable)

    def danet_bn_set_block_bytes(self, nbytes):
        return self.knob('bn_block_bytes', nbytes if nbytes > 0 else -1)


import contextlib   # noqa: E402


@contextlib.contextmanager
def knobs(**settings):
    """`with _lib.knobs(c3s_enable=0, pw=0): ...` -- the named switches (KNOBS) hold the given values inside the block and their
    previous values after it, whatever happens inside."""
    L = lib()
    prev = {}
    try:
        for k, v in settings.items():
            prev[k] = L.knob(k, v)
        yield L
    finally:
        for k, v in prev.items():
            L.knob(k, v)


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7): it must be
        # mapped BEFORE this library so that both share ONE runtime (streams, device memory).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libdanet_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                'or `python danet-densepose2smpl_amd/csrc/build.py`. There is no CPU fallback.' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = _Library(l)
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().danet_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('danet_hip ops run on the GPU only (got a %s tensor); there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('danet_hip ops need contiguous tensors')
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
