import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
mode = sys.argv[1]
if 'reset' in mode:
    reset_cfg()
cfg_from_dict({'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0., 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
dev = torch.device('cuda')
torch.manual_seed(0)
tr = Trainer(default_options(2), device=dev, distributed=False, lr=1e-30)
batch = synthetic_in_dict(tr.model, 2, dev, seed=1)
keep = []
for _ in range(2):
    out, losses = tr.train_step(batch)
    if 'keep' in mode:
        keep.append((out, losses))
    if 'float' in mode:
        e = {k: float(v.sum()) for k, v in losses.items()}
tr.capture(batch, warmup=2)
tr.train_step_graphed()
torch.cuda.synchronize()
print('ok', mode)
