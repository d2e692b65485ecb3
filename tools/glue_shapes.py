"""aten ops of one eager train step by (op, input shapes) with their device time (torch.profiler, CPU + device activities): which
tensors the copy / fill / add launches touch, forward and backward.  python tools/glue_shapes.py [rows]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
reset_cfg(); cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
dev = torch.device('cuda')
tr = Trainer(default_options(32), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, 32, dev, seed=1)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if dt > 0 and e.key.startswith('aten::'):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:100]))
rows.sort(reverse=True)
print('aten ops with device time: %.2f ms in %d calls' % (sum(r[0] for r in rows) / 1e3, sum(r[1] for r in rows)))
for dt, n, k, sh in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 60]:
    print('%8.1f us %4d  %-28s %s' % (dt, n, k, sh))
