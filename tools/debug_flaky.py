import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
from danet_densepose2smpl_amd import trainer as trainer_mod
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
reset_cfg()
cfg_from_dict({'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.PARTDROP_RATE': 0., 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.})
dev = torch.device('cuda')
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.manual_seed(0)
    NB = int(os.environ.get('NB', '2'))
    tr = Trainer(default_options(NB), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, NB, dev, seed=1)
    recs = []
    from danet_densepose2smpl_amd import conv as _c
    traces = []
    for step in range(3):
        _c.TRACE = []
        _, losses = tr.train_step(batch)
        traces.append([(a, b, float(c)) for a, b, c in _c.TRACE]); _c.TRACE = None
        torch.cuda.synchronize()
        L = {k: float(v.sum()) for k, v in losses.items()}
        named = [(n, p) for n, p in tr.model.named_parameters() if p.grad is not None and p.dim() == 4]
        picks = named[::max(1, len(named) // 12)]
        recs.append((L, {n: p.grad.detach().clone() for n, p in picks}))
    for s in (1, 2):
        dl = {k: round(abs(recs[s][0][k] - recs[0][0][k]) / (abs(recs[0][0][k]) + 1e-9), 4) for k in recs[0][0]}
        big = {k: v for k, v in dl.items() if v > 0.02}
        rg = {n.split('.')[0] + '..' + n[-24:]: round(((recs[s][1][n] - recs[0][1][n]).norm() / (recs[0][1][n].norm() + 1e-12)).item(), 3) for n in recs[0][1]}
        print('trial', trial, 'step', s, 'vs 0: loss rel diffs >2%:', big, '| grad rel:', list(rg.values()))

    for s in (1, 2):
        bad = [(i, a[0], a[1], a[2], b[2]) for i, (a, b) in enumerate(zip(traces[0], traces[s])) if abs(a[2] - b[2]) > 0.03 * abs(a[2]) + 1e-6 or a[0] != b[0]]
        if bad:
            print('trial', trial, 'TRACE step', s, 'first divergences (index, tag, shape, step0, stepS):', bad[:4], 'of', len(bad), '/', len(traces[0]), len(traces[s]))
