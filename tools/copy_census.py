"""Contiguous same-dtype device copies (the ones torch turns into hipMemcpyAsync = __amd_rocclr_copyBuffer) of one eager train step,
by shape and by the innermost package frame: a TorchDispatchMode sees every aten call, the backward pass included."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
ops = collections.Counter()


def frame():
    for f in reversed(traceback.extract_stack()[:-2]):
        if 'danet' in f.filename and 'tools/' not in f.filename:
            return '%s:%d' % (os.path.basename(f.filename), f.lineno)
    return '(autograd / torch)'


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        ops[name] += 1
        if name in ('aten.copy_.default', 'aten.clone.default', 'aten._to_copy.default', 'aten.contiguous.default'):
            ts = [a for a in args if torch.is_tensor(a)]
            if ts and ts[0].is_cuda:
                src = ts[-1]
                dst = ts[0]
                same = (len(ts) == 1) or (src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous())
                if name != 'aten.copy_.default':
                    same = src.is_contiguous() and (kwargs or {}).get('dtype', src.dtype) == src.dtype
                if same:
                    agg[(name, tuple(src.shape), str(src.dtype).replace('torch.', ''), frame())] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    tr.train_step(batch)
torch.cuda.synchronize()
print('memcpy-like copies:', sum(agg.values()))
for k, v in agg.most_common(45):
    print('%4d  %s' % (v, k))
print('top aten ops:', ops.most_common(25))
