"""tools/replay_probe.py with conv.TRACE taps captured inside the graph: after each replay, the first trace entries whose value is
not finite.  usage: python tools/replay_trace.py [B] [size] [n]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from danet_densepose2smpl_amd import conv as dconv                                       # noqa: E402
from danet_densepose2smpl_amd.config import cfg                                          # noqa: E402
from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, cfg.DANET.PARTDROP_RATE = size, size // 4, 0.
    cfg.DANET.STN_CENTER_JITTER = cfg.DANET.STN_SCALE_JITTER = 0.
    dev = torch.device('cuda')
    torch.manual_seed(0)
    tr = Trainer(default_options(B), device=dev, distributed=False, lr=1e-30)
    batch = synthetic_in_dict(tr.model, B, dev, seed=1)
    tr.train_step(batch)
    tr.train_step(batch)
    torch.cuda.synchronize()
    orig_core = tr._core

    def core(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            dconv.TRACE = []
        try:
            return orig_core(*a, **k)
        finally:
            if torch.cuda.is_current_stream_capturing():
                tr._trace, dconv.TRACE = dconv.TRACE, None
    tr._core = core
    tr.capture(batch, warmup=1)
    names = {id(p): k for k, p in tr.model.named_parameters()}
    st = tr.store
    for i in range(n):
        tr.train_step_graphed()
        torch.cuda.synchronize()
        vals = [float(v) for _, _, v in tr._trace]
        bad = [(j, tr._trace[j][0], tr._trace[j][1], vals[j]) for j in range(len(vals)) if not (vals[j] == vals[j] and abs(vals[j]) != float('inf'))]
        flat_bad = (~torch.isfinite(st.flat)).nonzero().flatten().tolist()[:6]
        where = []
        for o in flat_bad:
            for p in st.params:
                po = st.offsets[id(p)]
                if po <= o < po + p.numel():
                    where.append((names[id(p)], o - po, float(st.flat[o])))
        bias = [(j, t[0], t[1], vals[j]) for j, t in enumerate(tr._trace) if t[0].startswith('bias_grad')]
        print(json.dumps({'replay': i, 'ntrace': len(vals), 'bad': bad[:6], 'flat_bad': where, 'bias': bias}))


main()
