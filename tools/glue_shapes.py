"""aten ops of one eager train step by (op, input shapes): which tensors the copy / fill / add launches touch."""
import collections, os, sys
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
want = ('aten::copy_', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::sum', 'aten::constant_pad_nd', 'aten::slice_backward',
        'aten::cat', 'aten::contiguous', 'aten::_to_copy', 'aten::zeros', 'aten::zeros_like', 'aten::index', 'aten::select_backward', 'aten::max_pool2d_with_indices_backward')
agg = collections.Counter()
for e in prof.events():
    if e.name in want:
        shp = str([tuple(s) for s in (e.input_shapes or []) if s][:2])
        agg[(e.name, shp)] += 1
for (n, s), c in agg.most_common(int(sys.argv[1]) if len(sys.argv) > 1 else 70):
    print('%4d  %-26s %s' % (c, n, s[:110]))
