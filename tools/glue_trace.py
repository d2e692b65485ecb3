"""Call sites (file:line, two innermost frames inside the package) of every aten operator a train step dispatches on CUDA tensors:
a TorchDispatchMode over one eager bench-configuration step.  Complements tools/glue_sites.py (device time per function) with line
numbers and with the operators the autograd engine runs for built-in backward nodes.  usage: python tools/glue_trace.py [top]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from danet_densepose2smpl_amd.config import cfg                                               # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options      # noqa: E402

SKIP = ('aten.view', 'aten._unsafe_view', 'aten.detach', 'aten.t.', 'aten.transpose', 'aten.permute', 'aten.slice', 'aten.select',
        'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.alias', 'aten.as_strided', 'aten.empty', 'aten.reshape', 'aten.split',
        'aten.unbind', 'aten.narrow', 'aten.unfold', 'aten._reshape_alias', 'aten.is_', 'aten.sym_', 'aten.lift_fresh', 'aten.chunk',
        'aten.result_type', 'aten.new_empty', 'aten.stride', 'aten.size', 'aten.numel', 'aten.dim', 'aten.movedim', 'aten.flatten',
        'aten.unflatten', 'aten.view_as', 'aten.empty_like', 'aten.empty_strided', 'aten._local_scalar_dense', 'aten.item')


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            frames = [f for f in traceback.extract_stack()[:-1] if ('amd/' in f.filename) and 'tools/' not in f.filename]
            site = ' < '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(frames[-2:]))
            if not site:          # built-in backward nodes run by the autograd engine: tell them apart by their operand shapes
                site = '(engine) ' + ' '.join(str(tuple(a.shape)) + ('' if a.is_contiguous() else '*') + str(a.dtype)[6:] for a in args if torch.is_tensor(a))
            self.count[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE = 256, 64
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(32), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 32, dev, seed=3)
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t = Trace()
    with t:
        tr.train_step(batch)
    torch.cuda.synchronize()
    print('dispatched (non-view) operators: %d' % sum(t.count.values()))
    for (name, site), n in t.count.most_common(top):
        print('%4d  %-34s %s' % (n, name[:34], site))


main()
