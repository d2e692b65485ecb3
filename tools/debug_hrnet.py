import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from make_golden import formula_params
from oracle import torch_ref
from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
from danet_densepose2smpl_amd import hrnet
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reset_cfg(); cfg_from_dict({'DANET.INIMG_SIZE': S, 'DANET.HEATMAP_SIZE': S // 4})
ref = torch_ref.HRNet(part_out_dim=7); formula_params(ref); ref.train()
net = hrnet.PoseHighResolutionNet(part_out_dim=7); net.load_state_dict(ref.state_dict()); net = net.cuda().train()
img = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(3))
acts_r, acts_p = {}, {}
def hook(store, name):
    def f(m, i, o):
        if isinstance(o, (list, tuple)):
            for k, t in enumerate(o): store['%s[%d]' % (name, k)] = t.detach().float().cpu()
        elif torch.is_tensor(o): store[name] = o.detach().float().cpu()
    return f
names = ['conv1', 'bn1', 'conv2', 'bn2', 'layer1.0', 'layer1', 'transition1.0', 'transition1.1', 'stage2.0.branches.0.0', 'stage2.0.branches.0', 'stage2.0.branches.1',
         'stage2.0.fuse_layers.0.1', 'stage2.0.fuse_layers.1.0', 'stage2', 'stage3.0', 'stage3', 'stage4.0', 'stage4', 'final_pred.predict_u']
for n in names:
    dict(ref.named_modules())[n].register_forward_hook(hook(acts_r, n))
    dict(net.named_modules())[n].register_forward_hook(hook(acts_p, n))
with torch.no_grad():
    ref(img); net(img.cuda())
for k in acts_r:
    if k in acts_p:
        a, b = acts_p[k], acts_r[k]
        if a.shape != b.shape: print(k, 'SHAPE', a.shape, b.shape); continue
        print('%-34s shape=%-22s rel_max=%.4f rel_rms=%.4f  refmax=%.3f' % (k, tuple(b.shape), (a - b).abs().max() / (b.abs().max() + 1e-9), ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-9)), b.abs().max()))
