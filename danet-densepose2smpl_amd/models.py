"""Public model classes of the hot path (names as in the reference's `models` package) and the
tiny end-to-end step used by __graft_entry__.smoke()."""
import torch

from .config import cfg, cfg_from_dict, reset_cfg      # noqa: F401
from .danet import DaNet                                # noqa: F401
from .hrnet import PoseHighResolutionNet                # noqa: F401
from .iuv_estimator import IUV_Estimator                # noqa: F401
from .renderer import IUV_Renderer                      # noqa: F401
from .resnet import PoseResNet, SmplResNet, LimbResLayers, IUV_predict_layer  # noqa: F401
from .smpl import SMPL                                  # noqa: F401
from .smpl_regressor import SMPL_Regressor, DecomposedPredictor  # noqa: F401


def smoke_step(device, B=2, size=128):
    """One forward+backward+Adam step of DaNet (HRNet-W48 + SMPL + IUV render) on a tiny batch."""
    from .trainer import Trainer, synthetic_in_dict, default_options
    saved = (cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE)
    cfg_from_dict({'DANET.INIMG_SIZE': size, 'DANET.HEATMAP_SIZE': size // 4})
    try:
        torch.manual_seed(0)
        tr = Trainer(default_options(B), device=device, distributed=False)
        batch = synthetic_in_dict(tr.model, B, device, seed=1)
        _, losses = tr.train_step(batch)
        total = float(torch.stack([v.sum() for v in losses.values()]).sum())
        assert total == total and abs(total) < 1e9, 'non-finite loss %r' % total
        n_grad = sum(1 for p in tr.model.parameters() if p.grad is not None)
        return 'danet step ok (loss %.3f, %d losses, %d params with grad)' % (total, len(losses), n_grad)
    finally:
        cfg_from_dict({'DANET.INIMG_SIZE': saved[0], 'DANET.HEATMAP_SIZE': saved[1]})
