"""Data parallelism: one process per GPU, gradients averaged with bucketed all-reduces over
RCCL/xGMI, launched on a dedicated communication stream as soon as a bucket's gradients are
complete so that they overlap with the rest of the backward pass.

The reference is single-process / single-GPU (/root/reference/train/base_trainer.py:20); this is
new.  Loss normalisers use the LOCAL batch (iuv_estimator.py:325-326,
smpl_regressor.py:235), so gradients are AVERAGED over ranks; BatchNorm stays per-device.

Bucket plan: parameters in reverse registration order (the order their gradients become ready in
backward: regressor heads -> limb/body nets -> IUV heads -> HRNet stage4 ... stem), packed into
flat fp32 buckets of ~`bucket_mb` MB.  Parameters that received no gradient in a step (the
reference's never-used rot2pos/pos2rot stacks, or the whole regressor in pretrain mode) are
reduced as zeros so that every rank issues the same collectives (static plan).
"""
import torch
import torch.distributed as dist


class GradReducer(object):
    def __init__(self, module, bucket_mb=32.0, device=None, process_group=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.device = device or next(module.parameters()).device
        params = [p for p in module.parameters() if p.requires_grad]
        self.params = list(reversed(params))
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []          # list of dicts: flat buffer, params, offsets
        cur, cur_n = [], 0
        for p in self.params:
            if cur and cur_n + p.numel() > cap:
                self._add_bucket(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._add_bucket(cur)
        self.param_bucket = {}
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self.param_bucket[id(p)] = bi
        self.use_cuda = self.device.type == 'cuda'
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.use_cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._pending = None
        self._works = []

    def _add_bucket(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=torch.float32, device=self.device)
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o += p.numel()
        self.buckets.append({'flat': flat, 'params': list(params), 'offsets': offs, 'ready': 0, 'launched': False})

    # ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """One-time broadcast of parameters and buffers (BatchNorm statistics) from rank `src`."""
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                    dist.broadcast(t.data, src=src, group=self.group)

    def prepare(self):
        """Call before backward(): arms the gradient hooks until finish()."""
        for b in self.buckets:
            b['ready'] = 0
            b['launched'] = False
        self._works = []
        self._armed = True

    def _on_grad(self, p):
        if not getattr(self, '_armed', False):      # e.g. a backward captured into a hipGraph: reduce_now() runs after the replay
            return
        bi = self.param_bucket[id(p)]
        b = self.buckets[bi]
        b['ready'] += 1
        if b['ready'] == len(b['params']):
            self._launch(bi)

    def _launch(self, bi):
        b = self.buckets[bi]
        if b['launched']:
            return
        b['launched'] = True
        flat = b['flat']
        if self.use_cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            stream = self.comm_stream
            stream.wait_event(ready)
            ctx = torch.cuda.stream(stream)
        else:
            ctx = _Null()
        with ctx:
            # pack with multi-tensor copies (a few launches per bucket instead of one per parameter)
            dst, src, zero = [], [], []
            for p, o in zip(b['params'], b['offsets']):
                v = flat[o:o + p.numel()]
                if p.grad is None:
                    zero.append(v)
                else:
                    dst.append(v.view_as(p))
                    src.append(p.grad)
            if dst:
                torch._foreach_copy_(dst, src)
            if zero:
                torch._foreach_zero_(zero)
            flat.div_(self.world)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((bi, work))

    def finish(self):
        """Call after backward(): launches buckets with unused parameters, waits for all reductions
        and scatters the averaged gradients back into p.grad."""
        for bi, b in enumerate(self.buckets):
            if not b['launched']:
                self._launch(bi)
        for bi, work in self._works:
            work.wait()
        if self.use_cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        for b in self.buckets:
            dst, src = [], []
            for p, o in zip(b['params'], b['offsets']):
                g = b['flat'][o:o + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    dst.append(p.grad)
                    src.append(g)
            if dst:
                torch._foreach_copy_(dst, src)
        self._works = []
        self._armed = False

    def reduce_now(self):
        """Average all gradients after a backward that ran without hooks (hipGraph replay): every
        bucket is packed and all-reduced on the communication stream, back to back."""
        self.prepare()
        self.finish()

    def remove(self):
        for h in self._hooks:
            h.remove()


class _Null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
