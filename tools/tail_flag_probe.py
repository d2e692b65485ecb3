"""Where does the grid-barrier error word get set?  Eager steps, capture, replays: the flag and the wall time after each."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
from danet_densepose2smpl_amd import nn as dnn
reset_cfg(); cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
dev = torch.device('cuda')
tr = Trainer(default_options(32), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, 32, dev, seed=1)
def flag(tag, t0):
    torch.cuda.synchronize()
    bar = dnn._onepass_state(dev)
    print('%-12s %.3f s  error=%d  words[0:4]=%s  cnt=%s top=%d' % (tag, time.time() - t0, int(bar[2]), bar[:4].tolist(), [int(bar[16 * (1 + g)]) for g in range(8)], int(bar[16 * 17])), flush=True)
for i in range(3):
    t0 = time.time(); tr.train_step(batch); flag('eager %d' % i, t0)
t0 = time.time(); tr.capture(batch); flag('capture', t0)
for i in range(8):
    t0 = time.time(); tr.train_step_graphed(); flag('replay %d' % i, t0)
