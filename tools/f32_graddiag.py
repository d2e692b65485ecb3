"""Per-parameter gradient error of the fp32 mode on the g6 HRNet golden (to locate a wrong backward): one line per parameter
whose error exceeds 1e-2, in module order."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
from conftest import golden, GOLDEN                                   # noqa: E402
sys.path.insert(0, GOLDEN)
from make_golden import formula_params                                 # noqa: E402
from danet_densepose2smpl_amd import conv, hrnet                       # noqa: E402
import test_gpu_fp32 as T                                              # noqa: E402

if __name__ == '__main__':
    T._cfg(**{'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    g = golden('g6_hrnet')
    for mfma in (True, False):
        conv.F32_MFMA = mfma
        net = hrnet.PoseHighResolutionNet(part_out_dim=7)
        formula_params(net)
        net = net.cuda().train()
        img = torch.from_numpy(g['img']).cuda().requires_grad_(True)
        with conv.precision('fp32'):
            out = net(img)
            loss = sum((out[k].float() * torch.cos(torch.arange(out[k].numel(), dtype=torch.float32, device='cuda').view_as(out[k]) * 0.37)).sum() for k in T.KEYS[:5])
            loss.backward()
        gw = {k: p.grad for k, p in net.named_parameters()}
        bad = []
        n = 0
        for k, p in net.named_parameters():
            key = 'grad__' + k.replace('.', '__')
            if key in g.files and p.grad is not None:
                n += 1
                e = T._rel(p.grad, g[key])
                if e > 1e-2:
                    bad.append((k, round(e, 4)))
        print('mfma', mfma, 'checked', n, 'bad', len(bad))
        for b in bad[-40:]:
            print('  ', b)
