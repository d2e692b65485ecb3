"""Bit-reproducibility of the train step in the PRODUCTION configuration (replica atomics, conv-epilogue statistics, one-pass
BatchNorm backward): the same batch, the same weights (lr ~ 0), `runs` eager executions and hipGraph replays.  Reports, per
loss, whether all executions agree bit for bit, and which parameters' gradients differ (with the largest relative
difference) -- what is left names the kernels whose accumulation order is still free.
usage: python tools/determinism_probe.py [B] [size] [runs]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import _lib, nn as dnn, conv as dconv                      # noqa: E402
from danet_densepose2smpl_amd.config import cfg                                          # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, cfg.DANET.PARTDROP_RATE = size, size // 4, 0.
    cfg.DANET.STN_CENTER_JITTER = cfg.DANET.STN_SCALE_JITTER = 0.
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(B), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, B, dev, seed=1)
    tr.train_step(batch)

    def snap(losses):
        torch.cuda.synchronize()
        return ({k: v.detach().float().sum().clone() for k, v in losses.items()},
                {n: p.grad.detach().float().clone() for n, p in tr.model.named_parameters() if p.grad is not None})

    ex = [snap(tr.train_step(batch)[1]) for _ in range(runs)]
    tr.capture(batch, warmup=1)
    tr.train_step_graphed()
    ex += [snap(tr.train_step_graphed()[1]) for _ in range(2)]
    kinds = ['eager'] * runs + ['graph'] * 2
    ref = ex[0]
    out = {'B': B, 'size': size, 'acc_bytes': _lib.lib().danet_bn_acc_bytes(), 'onepass': bool(dnn.ONEPASS), 'fuse_stats': bool(dconv.FUSE_BN_STATS),
           'onepass_error': bool(dnn.onepass_error())}
    loss_diff = {}
    for k in ref[0]:
        d = max(abs(float(e[0][k]) - float(ref[0][k])) / (abs(float(ref[0][k])) + 1e-30) for e in ex[1:])
        if d > 0:
            loss_diff[k] = d
    out['losses_bit_equal'] = not loss_diff
    out['loss_rel_diff'] = loss_diff
    gd = {}
    for n in ref[1]:
        worst = 0.0
        for e in ex[1:]:
            if not torch.equal(e[1][n], ref[1][n]):
                worst = max(worst, float((e[1][n] - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30)))
        if worst > 0:
            gd[n] = worst
    out['grads_total'] = len(ref[1])
    out['grads_differing'] = len(gd)
    out['grads_worst'] = {n: [v, float(ref[1][n].abs().max()), [float(e[1][n].abs().max()) for e in ex[1:]]] for n, v in sorted(gd.items(), key=lambda kv: -kv[1])[:40]}
    # eager vs graph separately (a difference only there = a path difference, not an ordering freedom)
    eg = {}
    for n in ref[1]:
        if all(torch.equal(ex[i][1][n], ref[1][n]) for i in range(runs)) and not all(torch.equal(ex[i][1][n], ref[1][n]) for i in range(runs, len(ex))):
            eg[n] = float((ex[-1][1][n] - ref[1][n]).abs().max() / (ref[1][n].abs().max() + 1e-30))
    out['grads_eager_stable_but_graph_differs'] = dict(sorted(eg.items(), key=lambda kv: -kv[1])[:10])
    out['kinds'] = kinds
    print(json.dumps(out, indent=1))


main()
