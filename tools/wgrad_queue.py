"""List the weight-gradient problems of one train step by kernel family (tools: which layers go where)."""
import os
import sys
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict   # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402
from danet_densepose2smpl_amd import conv   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg_from_dict({'DANET.INIMG_SIZE': size, 'DANET.HEATMAP_SIZE': size // 4})
dev = torch.device('cuda:0')
tr = Trainer(default_options(B), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, B, dev, seed=1)
tr.train_step(batch)
seen = collections.Counter()
flops = collections.Counter()
orig = conv.flush_wgrads


def spy(bucket=None):
    for q in conv._WQ:
        (gptr, weight, x, gy, B_, H, W, Cin, Cout, groups, stride_) = q
        k = ('w3x3', B_, H, W, Cin, Cout, groups)
        if (id(weight), bucket) not in spy.done:
            seen[k] += 1; flops[k] += 2.0 * B_ * H * W * Cout * (Cin // groups) * 9
            spy.done.add((id(weight), bucket))
    for q in conv._WQG:
        (gptr, weight, x, gy, d) = q
        k = ('generic',) + tuple(d)
        if (id(weight), 'g') not in spy.done:
            seen[k] += 1; flops[k] += 2.0 * d[0] * d[4] * d[5] * d[6] * (d[3] // d[12]) * d[7] * d[8]
            spy.done.add((id(weight), 'g'))
    return orig(bucket)


spy.done = set()
conv.flush_wgrads = spy
import danet_densepose2smpl_amd.trainer as T   # noqa: E402
T._conv.flush_wgrads = spy
tr.train_step(batch)
tot = sum(flops.values())
print('total wgrad GFLOP %.1f' % (tot / 1e9))
for k, v in sorted(flops.items(), key=lambda kv: -kv[1])[:60]:
    print('%-70s n=%3d GFLOP %8.2f' % (str(k), seen[k], v / 1e9))
