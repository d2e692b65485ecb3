"""IUV map glue (/root/reference/utils/iuvmap.py:6-38, 103-147) as a handful of tensor ops
instead of ~100 threshold/min launches; integer-exact one-hot planes."""
import torch
import torch.nn.functional as F

INDEX2MASK = [[0], [1, 2], [3], [4], [5], [6], [7, 9], [8, 10], [11, 13], [12, 14], [15, 17], [16, 18],
              [19, 21], [20, 22], [23, 24]]
_MERGE = {}


def _onehot(x):
    idx = torch.argmax(x, dim=1)
    return F.one_hot(idx, x.shape[1]).permute(0, 3, 1, 2).to(torch.float32)


def iuvmap_clean(U_uv, V_uv, Index_UV, AnnIndex=None):
    """argmax -> exact one-hot, U/V masked by it (gradient flows to U,V only), iuvmap.py:6-38."""
    I = _onehot(Index_UV)
    A = None if AnnIndex is None else _onehot(AnnIndex)
    return I * U_uv.float(), I * V_uv.float(), I, A


def _merge_matrix(device):
    key = str(device)
    if key not in _MERGE:
        m = torch.zeros(15, 25)
        for i, grp in enumerate(INDEX2MASK):
            m[i, grp] = 1
        _MERGE[key] = m.to(device)
    return _MERGE[key]


def iuv_img2map(uvimages):
    """3-channel IUV image -> U,V,Index [B,25,H,W] + Ann [B,15,H,W] (iuvmap.py:103-147, uv_rois=None)."""
    part = torch.round(uvimages[:, 0] * 24).long().clamp_(0, 24)
    I = F.one_hot(part, 25).permute(0, 3, 1, 2).to(torch.float32)
    U = I * uvimages[:, 1:2]
    V = I * uvimages[:, 2:3]
    A = torch.einsum('ac,bchw->bahw', _merge_matrix(uvimages.device), I)
    return U, V, I, A
