"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/danet_hip.h declares; ops refuse CPU tensors (there is no CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'danet_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(danet_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from danet_densepose2smpl_amd import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.exported_symbols())
    assert lib.danet_version() >= 100


def test_workspace_queries_are_host_side():
    from danet_densepose2smpl_amd import _lib
    lib = _lib.lib()
    assert lib.danet_smpl_lbs_ctx_floats(32) == 32 * 648
    assert lib.danet_smpl_lbs_ctx_floats(4) == 8 * 648          # padded to batch groups of 8
    assert lib.danet_smpl_lbs_fwd_ws_floats(32, 6890, 9) == 208 * 32 + 108 * 32 * 27
    # per-tile partials of dA and d beta, half-tile partials of d pose-feature (216 of 96 coordinates), d v_posed [C][Bpad]
    assert lib.danet_smpl_lbs_bwd_ws_floats(32, 6890, 10) == 108 * 32 * (288 + 16) + 216 * 32 * 208 + 20670 * 32


def test_no_cpu_fallback():
    from danet_densepose2smpl_amd.smpl import SMPL
    from danet_densepose2smpl_amd.renderer import IUV_Renderer
    smpl = SMPL()
    with pytest.raises(RuntimeError, match='GPU only|no CPU'):
        smpl(betas=torch.zeros(1, 10), body_pose=torch.eye(3).expand(1, 23, 3, 3).contiguous(),
             global_orient=torch.eye(3).expand(1, 1, 3, 3).contiguous(), pose2rot=False)
    with pytest.raises(RuntimeError, match='GPU only|no CPU'):
        IUV_Renderer(256, 64).verts2uvimg(torch.zeros(1, 6890, 3), torch.ones(1, 3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'danet-densepose2smpl_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', txt, flags=re.M), f
                assert 'liboracle' not in txt, f


def test_new_ops_refuse_cpu_tensors_and_host_queries_work():
    """The ops added later in the round keep the rule: CPU tensors raise, planning queries are host-side."""
    import ctypes
    from danet_densepose2smpl_amd import _lib, conv, nn as dnn, part_ops
    from danet_densepose2smpl_amd.optim import FusedAdam
    lib = _lib.lib()
    with pytest.raises(RuntimeError, match='GPU only|no CPU'):
        conv.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 3, 3), None, 1, 1)
    with pytest.raises(RuntimeError, match='GPU only|no CPU'):
        part_ops.part_clean(torch.zeros(1, 504, 4, 4))
    with pytest.raises(RuntimeError, match='GPU'):
        FusedAdam([torch.nn.Parameter(torch.zeros(4))])
    # CPU tensors: the multi-tensor BatchNorm helper must fall back to the per-module path, which then refuses them
    bn = dnn.BatchNorm2d(8).train()
    with pytest.raises(RuntimeError, match='GPU only|no CPU'):
        dnn.multi_batch_norm([bn], [torch.zeros(1, 8, 4, 4)])
    # kernel-selection queries (no device needed)
    assert lib.danet_conv_forward_kernel(32, 64, 64, 48, 64, 64, 48, 3, 3, 1, 1, 1, 1, 0, 0) == 4312      # 3x3/s1: LDS-tile kernel, MT 4, NT 3, no K split
    assert lib.danet_conv_forward_kernel(32, 8, 8, 384, 8, 8, 384, 3, 3, 1, 1, 1, 1, 0, 0) == 4342         # small-M: 64-pixel tiles, K split four ways
    prev = lib.danet_conv3x3_set(0, -1, -1, 0, -1)
    assert prev == 1
    assert lib.danet_conv_forward_kernel(32, 64, 64, 48, 64, 64, 48, 3, 3, 1, 1, 1, 1, 0, 0) == 4311      # <4,3>, vec8, fast kernel
    assert lib.danet_conv_forward_kernel(32, 8, 8, 384, 8, 8, 384, 3, 3, 1, 1, 1, 1, 0, 0) % 10 == 1
    lib.danet_conv3x3_set(1, -1, -1, 0, -1)
    assert lib.danet_conv_forward_kernel(32, 64, 64, 48, 64, 64, 96, 3, 3, 2, 1, 1, 1, 0, 0) % 10 == 1     # strided 3x3 stays on the gather kernel
    assert lib.danet_conv_forward_kernel(2, 8, 8, 7, 8, 8, 8, 3, 3, 1, 1, 1, 1, 0, 0) % 10 == 0            # odd channels: general kernel
    jobs = (_lib.Wg3Job * 2)()
    for j, c in zip(jobs, (48, 96)):
        j.x = j.dy = j.dw = 16
        j.B, j.H, j.W, j.Cin, j.Cout, j.groups, j.stride = 4, 32, 32, c, c, 1, 1
    assert lib.danet_conv_wgrad3x3_multi_ws_floats(ctypes.addressof(jobs), 2) >= 9 * (48 * 48 + 96 * 96)
    assert lib.danet_adam_chunk_bytes() == 32 and lib.danet_conv_pack_job_bytes() >= 56
