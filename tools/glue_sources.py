"""Which Python lines of the package launch the remaining ATen kernels of the FORWARD pass (torch.profiler with stacks)."""
import os
import sys
import collections
import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict                 # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402

cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
dev = torch.device('cuda:0')
tr = Trainer(default_options(32), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, 32, dev, seed=1)
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith('aten::') and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    where = 'backward / unknown'
    for fr in (ev.stack or []):
        if 'danet' in fr and 'tools/' not in fr and 'torch/' not in fr:
            where = fr.split('danet_densepose2smpl_amd/')[-1].split('danet-densepose2smpl_amd/')[-1][:90]
            break
    a = agg[(ev.name, where)]
    a[0] += 1
    a[1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print('aten launches %d, %.2f ms' % (sum(v[0] for _, v in rows), sum(v[1] for _, v in rows) / 1e3))
for (name, where), (n, t) in rows[:70]:
    print('%8.1f us %4d  %-26s %s' % (t, n, name, where))
