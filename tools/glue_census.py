"""Which lines of the package call the tensor ops behind the non-HIP (aten) kernels of one eager train step: the Python entry
points are wrapped and every call is booked to the innermost frame inside danet_densepose2smpl_amd (forward and the Python side
of the backward pass; sums that autograd itself forms for fan-out tensors do not appear).  python tools/glue_census.py [top]"""
import collections, functools, os, sys, traceback
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
reset_cfg(); cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
dev = torch.device('cuda')
tr = Trainer(default_options(32), device=dev, distributed=False)
batch = synthetic_in_dict(tr.model, 32, dev, seed=1)
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
agg = collections.Counter()
ON = [False]


def wrap(owner, name):
    orig = getattr(owner, name)

    @functools.wraps(orig)
    def w(*a, **k):
        if ON[0]:
            ON[0] = False
            try:
                fr = '?'
                for f in reversed(traceback.extract_stack()[:-1]):
                    if 'danet' in f.filename and 'tools/' not in f.filename:
                        fr = '%s:%d' % (os.path.basename(f.filename), f.lineno)
                        break
                big = max([t.numel() for t in list(a) + list(k.values()) if torch.is_tensor(t)] or [0])
                agg[(name, fr, 'big' if big > 1 << 20 else 'small')] += 1
            finally:
                ON[0] = True
        return orig(*a, **k)
    setattr(owner, name, w)


for n in ('copy_', 'clone', 'contiguous', 'zero_', 'fill_', 'float', 'bfloat16', 'to', 'sum', 'add_', 'mul_', '__add__', '__mul__', '__sub__', '__radd__', '__rmul__',
          '__truediv__', '__getitem__', '__setitem__', 'masked_fill_', 'clamp', 'abs', 'mean', 'index_select', 'gather', 'repeat', 'expand_as'):
    wrap(torch.Tensor, n)
for n in ('zeros', 'ones', 'cat', 'stack', 'zeros_like', 'ones_like', 'full', 'where', 'sum', 'bmm', 'matmul', 'einsum', '_foreach_copy_', '_foreach_add_'):
    wrap(torch, n)
for n in ('pad', 'interpolate', 'max_pool2d', 'relu', 'linear', 'smooth_l1_loss', 'cross_entropy', 'l1_loss', 'mse_loss', 'softmax', 'grid_sample', 'affine_grid'):
    wrap(F, n)
ON[0] = True
tr.train_step(batch)
torch.cuda.synchronize()
ON[0] = False
top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
print('wrapped calls:', sum(agg.values()))
for (name, fr, sz), n in agg.most_common(top):
    print('%4d  %-16s %-5s %s' % (n, name, sz, fr))
