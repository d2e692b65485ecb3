"""Checkpoint files in the reference's format.

* training checkpoints: /root/reference/utils/saver.py:24-70 -- one dict holding, per model name, its state dict
  WITHOUT the `iuv2smpl.smpl.*` buffers (the SMPL tables are not learned), per optimizer name its state dict, and the
  bookkeeping keys epoch / batch_idx / batch_size / dataset_perm / total_step_count; loading updates only the keys both
  sides have.
* released / pretrained weights: /root/reference/demo.py:92-97 -- `checkpoint['model']` loaded with strict=False.
State-dict keys of this package's modules are identical to the reference's (tests/test_host_logic.py), so files move
in both directions.
"""
import os
from collections import OrderedDict

import torch

SMPL_PREFIX = 'iuv2smpl.smpl.'
BOOKKEEPING = ('epoch', 'batch_idx', 'batch_size', 'dataset_perm', 'total_step_count')


def _strip_module(sd):
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())


def save_checkpoint(path, models, optimizers=None, epoch=0, batch_idx=0, batch_size=0, dataset_perm=None, total_step_count=0):
    """models / optimizers: {name: object} as in saver.save_checkpoint(models, optimizers, ...)."""
    ckpt = {}
    for name, m in models.items():
        ckpt[name] = OrderedDict((k, v.detach().cpu()) for k, v in m.state_dict().items() if not k.startswith(SMPL_PREFIX))
    for name, o in (optimizers or {}).items():
        ckpt[name] = o.state_dict()
    ckpt.update({'epoch': epoch, 'batch_idx': batch_idx, 'batch_size': batch_size, 'dataset_perm': dataset_perm,
                 'total_step_count': total_step_count})
    os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
    torch.save(ckpt, path)
    return path


def _load_file(path, map_location, trusted):
    """torch.load with weights_only=True (tensors and plain containers only); files that need full unpickling (numpy
    permutation arrays of utils/saver.py, old torch versions) are read only when the caller vouches for them."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception:
        if not trusted:
            raise RuntimeError('%s needs full unpickling (arbitrary code may run): pass trusted=True for files you trust' % path)
        return torch.load(path, map_location=map_location, weights_only=False)


def load_checkpoint(path, models, optimizers=None, map_location='cpu', trusted=False):
    """saver.load_checkpoint: every model takes the keys it shares with the file (shape mismatches are an error, as
    in the reference's load_state_dict); optimizers load theirs if present.  Returns the bookkeeping values."""
    if not os.path.isfile(path):
        raise ValueError('checkpoint does not exist: %s' % path)
    ckpt = _load_file(path, map_location, trusted)
    for name, m in models.items():
        if name in ckpt:
            own = m.state_dict()
            own.update({k: v for k, v in _strip_module(ckpt[name]).items() if k in own})
            m.load_state_dict(own)
    for name, o in (optimizers or {}).items():
        if name in ckpt:
            o.load_state_dict(ckpt[name])
    return {k: ckpt.get(k) for k in BOOKKEEPING}


def load_pretrained(model, path, key='model', map_location='cpu', trusted=False):
    """demo.py:92-97 / eval.py: `model.load_state_dict(checkpoint['model'], strict=False)`; a bare state dict and
    DataParallel's 'module.' prefix are accepted too.  Returns (missing_keys, unexpected_keys)."""
    if not os.path.isfile(path):
        raise ValueError('pretrained model does not exist: %s' % path)
    ckpt = _load_file(path, map_location, trusted)
    sd = ckpt[key] if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict) else ckpt
    res = model.load_state_dict(_strip_module(sd), strict=False)
    return list(res.missing_keys), list(res.unexpected_keys)
