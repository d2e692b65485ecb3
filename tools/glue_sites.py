"""Which lines of the package launch the torch tensor-op kernels of a train step (the "glue" family of tools/step_breakdown.py):
an eager bench-configuration step under torch.profiler with Python stacks, aggregated by (aten op, first frame inside the
package).  usage: python tools/glue_sites.py [top]"""
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg                                               # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options      # noqa: E402


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 70
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE = 256, 64
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(32), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, 32, dev, seed=3)
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        tr.train_step(batch)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = ev.self_device_time_total
        if not dt or not ev.name.startswith('aten::'):
            continue
        site = '?' if ev.stack else '(no python stack: autograd engine)'
        for fr in (ev.stack or []):
            if ('amd/' in fr or 'bench.py' in fr) and 'dist-packages' not in fr:
                site = fr.split('/')[-1]
                break
        if site == '?' and os.environ.get('GLUE_DEBUG'):
            print(ev.name, ev.stack[:6])
        a = agg[(ev.name, site)]
        a[0] += 1
        a[1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print('aten ops with device time: %d launches-ish, %.2f ms' % (sum(v[0] for _, v in rows), tot / 1e3))
    for (name, site), (n, dt) in rows[:top]:
        print('%-28s %-70s %4d  %8.1f us' % (name, site[:70], n, dt))


main()
