"""Adam over all parameters in one HIP launch (csrc/adam.hip); same update as torch.optim.Adam(lr, betas=(0.9, 0.999),
eps=1e-8) -- the reference's optimizer, /root/reference/train/trainer.py:42-44 -- without weight decay / amsgrad.

The moments live in two flat fp32 buffers; a device table of <= 32768-element chunks {param, grad, moment offset, n}
drives the kernel.  With a `grad_store` (distributed.GradStore: every gradient is a view of one flat buffer) the
table is written once and never changes; which parameters received a gradient in the step comes from the store's
per-parameter mask (GradStore.used: filled from its post-accumulate-grad hooks and, with N > 1, summed over the ranks with
the last gradient bucket -- a parameter is skipped only if NO rank produced a gradient for it, so replicas stay identical).  Without a store the table is
rebuilt (host side, double-buffered pinned copy) whenever a gradient tensor's address changed, and a NULL gradient
pointer marks the skip.

Step counts are per parameter, as in torch.optim.Adam: a parameter without a gradient in a step is skipped entirely
(moments untouched) and the kernel counts the step in `idle`; its bias corrections use step - idle.  The reference
pre-trains the IUV estimator alone for 5000 steps (/root/reference/train/base_trainer.py:74): the regressor's first
update then sees step 1 -- with one global count it would be ~3x too large.  state_dict / load_state_dict carry the
per-parameter counts in torch's layout.

With world_size > 1 the gradients in the store (and hence every p.grad) are the SUM over the ranks; the average is
formed inside the update (grad_scale = 1 / world).  `param_groups[0]['lr']` is a device tensor, so a trainer can
decay the rate between hipGraph replays.  GPU only."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import ptr, check, stream

CHUNK = 32768


class FusedAdam(object):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_store=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError('FusedAdam runs on the GPU only')
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError('FusedAdam: parameters must be contiguous fp32 tensors')
        self.betas, self.eps = betas, eps
        lr_t = lr if torch.is_tensor(lr) else torch.tensor(float(lr), device=dev)
        self.param_groups = [{'lr': lr_t.to(device=dev, dtype=torch.float32).reshape(()), 'params': self.params}]
        self.offsets, tot = [], 0
        for p in self.params:
            self.offsets.append(tot)
            tot += (p.numel() + 3) // 4 * 4                   # 16-byte aligned moment slices
        self.exp_avg = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)                 # global step count
        self.idle = torch.zeros(len(self.params), dtype=torch.float32, device=dev)       # per parameter: steps without a gradient
        cb = int(_lib.lib().danet_adam_chunk_bytes())
        assert cb == 32
        self._dtype = np.dtype([('p', np.uint64), ('g', np.uint64), ('off', np.int64), ('n', np.int32), ('pad', np.int32)])
        rows = []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            for c0 in range(0, p.numel(), CHUNK):
                rows.append((i, c0, min(CHUNK, p.numel() - c0), o + c0))
        self._rows = rows
        self.nchunks = len(rows)
        # two pinned host tables, used alternately: the asynchronous H2D copy of step N may still be reading one while
        # the host fills the other for step N+1 (an event guards the reuse two steps later)
        self._hosts = [torch.empty(self.nchunks * cb, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._events = [None, None]
        self._turn = 0
        self._table = torch.empty(self.nchunks * cb, dtype=torch.uint8, device=dev)
        pidx = np.array([r[0] for r in rows], dtype=np.int64)
        self._pidx = pidx
        self._c0b = np.array([r[1] for r in rows], dtype=np.uint64) * 4
        for h in self._hosts:
            a = h.numpy().view(self._dtype)
            a['p'] = np.array([self.params[i].data_ptr() for i in pidx], dtype=np.uint64) + self._c0b
            a['off'] = np.array([r[3] for r in rows], dtype=np.int64)
            a['n'] = np.array([r[2] for r in rows], dtype=np.int32)
            a['pad'] = np.array([2 * r[0] + (1 if r[1] == 0 else 0) for r in rows], dtype=np.int32)     # parameter index, first-chunk flag
            a['g'] = 0
        self._gptrs = None
        self.grad_store = grad_store
        self.grad_scale = 1.0 if grad_store is None else float(grad_store.grad_scale)
        if grad_store is not None:
            for p in self.params:
                if not grad_store.has(p):
                    raise ValueError('FusedAdam: a parameter is missing from the gradient store')
            g = np.array([grad_store.grad_ptr(p) for p in self.params], dtype=np.uint64)
            a = self._hosts[0].numpy().view(self._dtype)
            a['g'] = g[pidx] + self._c0b
            self._table.copy_(self._hosts[0])                 # synchronous, once
            # the store's mask is in ITS parameter order: gathered into this optimizer's order every step (one tiny launch)
            self._used_perm = torch.tensor([grad_store.index_of(p) for p in self.params], dtype=torch.long, device=dev)
            self._used = torch.ones(len(self.params), dtype=torch.float32, device=dev)
        # device int the kernel tests before it touches anything: non-zero = this step's gradients are invalid (set by the trainer
        # to the one-pass BatchNorm backward's barrier error word, nn.onepass_poison) ...
        self.poison = None
        # ... and, data-parallel, the float that holds the sum of that word over all ranks (GradStore.poison): > 0 = skip, everywhere
        self.poison_sum = None

    def state_dict(self):
        """torch.optim.Adam's layout (per-parameter step / exp_avg / exp_avg_sq + one param group), so the checkpoints of
        utils/saver.py move between this optimizer and the reference's."""
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {'step': (self.step_t - self.idle[i]).detach().clone(), 'exp_avg': self.exp_avg[o:o + n].view_as(p).clone(),
                        'exp_avg_sq': self.exp_avg_sq[o:o + n].view_as(p).clone()}
        group = {'lr': float(self.param_groups[0]['lr']), 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0,
                 'amsgrad': False, 'maximize': False, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        ids = [i for g in groups for i in g['params']]
        if len(ids) != len(self.params):
            raise ValueError('FusedAdam.load_state_dict: %d parameters in the file, %d here' % (len(ids), len(self.params)))
        steps = [0.0] * len(self.params)
        for pos, (i, p, o) in enumerate(zip(ids, self.params, self.offsets)):
            st = sd['state'].get(i)
            if st is None:
                continue                    # (torch keeps no state for a parameter that never had a gradient: step 0, zero moments)
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps[pos] = float(st['step'])
        top = max(steps)
        self.step_t.fill_(top)              # global count = the most-updated parameter's; the others' deficits go to `idle`
        self.idle.copy_(torch.tensor([top - s_ for s_ in steps], dtype=torch.float32))
        self.param_groups[0]['lr'].fill_(float(groups[0]['lr']))
        self.betas, self.eps = tuple(groups[0].get('betas', self.betas)), groups[0].get('eps', self.eps)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _refresh_table(self):
        if self.grad_store is not None:
            return                                            # gradients live in the store: the table never changes
        g = np.fromiter((0 if p.grad is None else p.grad.data_ptr() for p in self.params), dtype=np.uint64, count=len(self.params))
        if self._gptrs is not None and np.array_equal(g, self._gptrs):
            return
        for p in self.params:
            if p.grad is not None and (p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.data_ptr() % 16):
                raise ValueError('FusedAdam: gradients must be contiguous, 16-byte aligned fp32 tensors')
        k = self._turn
        self._turn ^= 1
        if self._events[k] is not None:
            self._events[k].synchronize()                     # the copy that last read this host table has finished
        gp = g[self._pidx]
        self._hosts[k].numpy().view(self._dtype)['g'] = np.where(gp != 0, gp + self._c0b, 0).astype(np.uint64)
        self._table.copy_(self._hosts[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self._table.device))
        self._events[k] = ev
        self._gptrs = g

    @torch.no_grad()
    def step(self):
        self._refresh_table()
        self.step_t.add_(1.0)
        used = None
        if self.grad_store is not None:
            torch.index_select(self.grad_store.used, 0, self._used_perm, out=self._used)
            used = self._used
        check(_lib.lib().danet_adam_step(ptr(self._table), self.nchunks, ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                         ptr(self.param_groups[0]['lr']), ptr(self.step_t), ptr(used), ptr(self.idle),
                                         float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.grad_scale),
                                         ptr(self.poison), ptr(self.poison_sum), stream()), 'danet_adam_step')
        # the kernel wrote the parameters through raw pointers: bump their version counters like an in-place op would
        # (the conv weight-pack cache and autograd's saved-tensor checks key on them)
        upd = self.params if self.grad_store is not None else [p for p in self.params if p.grad is not None]
        torch._C._autograd._unsafe_set_version_counter(upd, [p._version + 1 for p in upd])
