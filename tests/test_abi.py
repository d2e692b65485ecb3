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
    assert lib.danet_smpl_lbs_bwd_ws_floats(32, 6890, 10) == 108 * 32 * (288 + 208 + 16)


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
