"""Which Python lines launch the remaining torch (ATen) kernels of a train step: torch.profiler with
stacks on one eager step.  python tools/torch_ops_profile.py [B] [size]"""
import os
import sys
import collections

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict                 # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg_from_dict({'DANET.INIMG_SIZE': size, 'DANET.HEATMAP_SIZE': size // 4})
dev = torch.device('cuda:0')
torch.manual_seed(0)
tr = Trainer(default_options(B), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, B, dev, seed=1)
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=16):
    t = getattr(ev, 'self_device_time_total', 0)
    if t <= 0 or not ev.key.startswith('aten::'):
        continue
    where = 'autograd/other'
    for fr in ev.stack:
        if 'danet' in fr and 'tools/' not in fr:
            where = fr.replace(ROOT + '/', '')
            break
    a = agg[(ev.key, where)]
    a[0] += ev.count
    a[1] += t
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print('total aten device time %.2f ms in %d launches' % (sum(v[1] for _, v in rows) / 1e3, sum(v[0] for _, v in rows)))
for (name, where), (n, t) in rows[:80]:
    print('%8.1f us %5d  %-28s %s' % (t, n, name, where[:120]))

print('--- by shape')
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, 'self_device_time_total', 0)
    if t > 0 and ev.key in ('aten::copy_', 'aten::fill_', 'aten::add', 'aten::mul', 'aten::add_', 'aten::sum', 'aten::cat', 'aten::mm'):
        a = agg[(ev.key, str(ev.input_shapes)[:100])]
        a[0] += ev.count
        a[1] += t
for (name, shp), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%8.1f us %5d  %-14s %s' % (t, n, name, shp))
n = 0
for ev in prof.key_averages(group_by_stack_n=16):
    if ev.key == 'aten::copy_' and n < 3:
        print('STACK SAMPLE', ev.stack[:16])
        n += 1
