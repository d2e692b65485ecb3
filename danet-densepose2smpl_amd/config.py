"""Minimal counterpart of the reference's global Detectron-style `cfg`
(/root/reference/models/core/config.py + configs/danet_default.yaml): only the keys the hot
path reads.  Values below restate configs/danet_default.yaml."""
import copy

import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(d):
    if isinstance(d, dict):
        return AttrDict({k: _wrap(v) for k, v in d.items()})
    return d


_DEFAULT = {
    'SOLVER': {'MAX_ITER': 500000, 'TYPE': 'Adam', 'BASE_LR': 0.0001, 'STEPS': [0, 30000, 60000], 'GAMMA': 0.1},
    'DANET': {
        'INIMG_SIZE': 224, 'DECOMPOSED': True, 'INPUT_MODE': 'iuv', 'USE_6D_ROT': True,
        'GLO_NUM_LAYERS': 18, 'SMPL_MODEL_TYPE': 'neutral', 'IUV_REGRESSOR': 'hrnet', 'HEATMAP_SIZE': 56,
        'NUM_PATCHES': 24, 'INDEX_WEIGHTS': 2.0, 'PART_WEIGHTS': 0.3, 'POINT_REGRESSION_WEIGHTS': 0.5,
        'SMPL_POSE_WEIGHTS': 60.0, 'SMPL_BETAS_WEIGHTS': 0.06, 'PROJ_KPS_WEIGHTS': 300.0, 'KPS3D_WEIGHTS': 300.0,
        'VERTS_WEIGHTS': 0, 'ORTHOGONAL_WEIGHTS': 0, 'JOINT_POSITION_WEIGHTS': 1.0, 'STN_KPS_WEIGHTS': 1.0,
        'STN_HM_WEIGHTS': 0, 'STN_CENTER_JITTER': 0.1, 'STN_SCALE_JITTER': 0.2, 'STN_PART_VIS_SCORE': 0.5,
        'USE_LEARNED_RATIO': True, 'PARTDROP_RATE': 0.3, 'REFINE_STRATEGY': 'gcn',
        'REFINEMENT': {'REFINE_ON': True, 'STACK_NUM': 1, 'FEAT_DIM': 128, 'GCN_NUM_LAYER': 3, 'POS_INTERSUPV': True},
        # torch 1.1 (the reference's pinned version, requirements.txt:11) samples with align_corners=True;
        # SURVEY.md Appendix D.1.  False reproduces what the reference does under a modern torch.
        'ALIGN_CORNERS': True,
    },
    'MSRES_MODEL': {
        'EXTRA': {'DECONV_WITH_BIAS': False, 'NUM_DECONV_LAYERS': 3, 'NUM_DECONV_FILTERS': [256, 256, 256],
                  'NUM_DECONV_KERNELS': [4, 4, 4], 'NUM_LAYERS': 50},
    },
    'HR_MODEL': {
        'PRETR_SET': 'none',
        'EXTRA': {
            'STAGE2': {'NUM_MODULES': 1, 'NUM_BRANCHES': 2, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4, 4], 'NUM_CHANNELS': [48, 96], 'FUSE_METHOD': 'SUM'},
            'STAGE3': {'NUM_MODULES': 4, 'NUM_BRANCHES': 3, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4, 4, 4], 'NUM_CHANNELS': [48, 96, 192], 'FUSE_METHOD': 'SUM'},
            'STAGE4': {'NUM_MODULES': 3, 'NUM_BRANCHES': 4, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4, 4, 4, 4], 'NUM_CHANNELS': [48, 96, 192, 384], 'FUSE_METHOD': 'SUM'},
        },
    },
}

cfg = _wrap(copy.deepcopy(_DEFAULT))


def reset_cfg():
    cfg.clear()
    cfg.update(_wrap(copy.deepcopy(_DEFAULT)))
    return cfg


def _merge(a, b):
    for k, v in a.items():
        if isinstance(v, dict) and isinstance(b.get(k), dict):
            _merge(v, b[k])
        else:
            b[k] = _wrap(v)


def cfg_from_file(path):
    """Merge a yaml file (same schema as configs/danet_default.yaml) into cfg."""
    with open(path) as f:
        _merge(yaml.safe_load(f), cfg)
    return cfg


def cfg_from_dict(d):
    """Merge {'DANET.INIMG_SIZE': 256, ...} style dotted overrides or a nested dict."""
    nested = {}
    for k, v in d.items():
        node = nested
        parts = k.split('.')
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    _merge(nested, cfg)
    return cfg
