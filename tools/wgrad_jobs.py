"""Lists the weight-gradient problems a bench-configuration step queues for its multi-problem launches (conv.flush_wgrads): the
3x3 transpose-read queue (_WQ) and the generic queue (_WQG), by shape.  usage: python tools/wgrad_jobs.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv                                 # noqa: E402
from danet_densepose2smpl_amd.config import cfg_from_dict                 # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402

cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
dev = torch.device('cuda:0')
torch.manual_seed(0)
tr = Trainer(default_options(32), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, 32, dev, seed=1)
tr.train_step(batch)
orig = conv.flush_wgrads
seen = []


def spy(*a, **k):
    if conv._WQ or conv._WQG:
        seen.append(([(q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[-1] is not None) for q in conv._WQ],
                     [tuple(q[4]) + (q[-1] is not None,) for q in conv._WQG]))
    return orig(*a, **k)


conv.flush_wgrads = spy
import danet_densepose2smpl_amd.trainer as T                               # noqa: E402
T._conv.flush_wgrads = spy
tr.train_step(batch)
torch.cuda.synchronize()
for wq, wqg in seen:
    print('3x3 queue: %d problems' % len(wq))
    for k, n in collections.Counter(wq).most_common():
        print('   %3d x (B,H,W,Cin,Cout,groups,stride,padded) = %s' % (n, k))
    print('generic queue: %d problems' % len(wqg))
    for k, n in collections.Counter(wqg).most_common():
        print('   %3d x (B,H,W,Cin,OH,OW,Cout,R,S,stride,pad,dil,groups,padded) = %s' % (n, k))
