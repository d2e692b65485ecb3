"""Gradient storage and data parallelism: one process per GPU, gradients averaged with bucketed all-reduces
over RCCL/xGMI that overlap with the rest of the step.

The reference is single-process / single-GPU (/root/reference/train/base_trainer.py:20); this is new
(SURVEY.md 8e).  Loss normalisers use the LOCAL batch (iuv_estimator.py:325-326, smpl_regressor.py:235),
so gradients are AVERAGED over ranks; BatchNorm stays per-device.

`GradStore` owns ONE flat fp32 buffer; every parameter's gradient is a view of it (zero-copy: the weight-gradient
kernels write straight into the views, csrc/conv_wgrad*.hip; FusedAdam reads them through a table that never
changes).  The buffer is laid out in gradient-ready order -- reverse registration order: regressor heads ->
limb/body nets -> IUV heads -> HRNet stage4 ... stem -- and cut into buckets of ~`bucket_mb` MB.  A step
finishes its gradients bucket by bucket (the deferred weight-gradient launches of conv.flush_wgrads run per
bucket) and all-reduces each bucket as soon as it is complete, on the process group's communication stream,
while the next bucket's weight gradients are still being computed; the optimizer waits for the last one.
The collectives are issued in bucket order on every rank, whatever subset of parameters received gradients
(unused parameters are reduced as zeros), and they are plain stream work: inside a hipGraph capture they become
graph nodes with the same dependencies, so the replayed step overlaps them the same way.
The sum is not divided: `grad_scale` = 1 / world is folded into the optimizer's update (FusedAdam) or applied
by `scale_()` for other optimizers.

Early buckets.  With `arm_early(callback)` every parameter carries a post-accumulate-grad hook; as soon as all
parameters of the next bucket IN ORDER have their gradient (written, or queued as a deferred weight-gradient job) the
callback runs for that bucket from inside the backward pass -- the trainer's callback launches the bucket's weight
gradients and starts its all-reduce, so the regressor / limb-net buckets are on the wire while the HRNet backward is
still running.  Buckets are released strictly in index order (a bucket that completes before its predecessor waits
for it), so every rank issues the same sequence of collectives; buckets that do not complete during backward are
finished by the trainer's tail loop, in order as well.  Parameters that received no gradient in the previous step
(never-used modules: rot2pos / pos2rot, a skipped regressor) are not waited for -- the first step, which knows
nothing yet, releases nothing early around them; should such a parameter receive a gradient after its bucket has
gone out, end_backward() raises instead of training on an incomplete sum.

bf16 wire format (`wire_dtype=torch.bfloat16`, BASELINE config C5's 204.5 MB instead of 409 MB per step): a bucket is
rounded into a bf16 staging buffer, summed over the ranks in bf16, and widened back into the fp32 store when the
optimizer waits for it.
"""
import torch
import torch.distributed as dist


class GradStore(object):
    def __init__(self, params, bucket_mb=32.0, device=None, process_group=None, world=None, wire_dtype=torch.float32):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError('GradStore: no parameters')
        self.params = list(reversed(params))
        self.device = device or self.params[0].device
        self.group = process_group
        if world is None:
            world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.world = int(world)
        self.grad_scale = 1.0 / self.world
        cap = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.offsets, self.bucket_of, self.buckets = {}, {}, []
        off, start, first = 0, 0, 0
        for i, p in enumerate(self.params):
            n = (p.numel() + 3) // 4 * 4                         # 16-byte aligned slices (vector loads of adam.hip)
            if off > start and off - start + n > cap:
                self.buckets.append((start, off, first, i))
                start, first = off, i
            self.offsets[id(p)] = off
            self.bucket_of[id(p)] = len(self.buckets)
            off += n
        self.buckets.append((start, off, first, len(self.params)))
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.wire_dtype = wire_dtype
        self.wire = None if wire_dtype == torch.float32 else torch.zeros(off, dtype=wire_dtype, device=self.device)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        # which parameters received a gradient in the last backward pass (1 / 0, in self.params order): filled from the
        # post-accumulate-grad hooks when the trainer leaves backward_scope; the optimizer skips the others (optim.FusedAdam)
        self.used = torch.ones(len(self.params), dtype=torch.int32, device=self.device)
        self._used_host = torch.ones(len(self.params), dtype=torch.int32)
        if self.device.type == 'cuda':
            self._used_host = self._used_host.pin_memory()
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)
        self._works = []
        self.issued = []                    # bucket indices in the order their collectives were issued this step ...
        self.issued_early = 0               # ... and how many of them from inside the backward pass
        self._handed = set()
        self._early_cb = None
        self._expected = None               # ids of the parameters that received a gradient in the previous step
        self._fired, self._late = set(), []
        self._pending = None
        self._next = 0
        self._in_backward = False

    # ---- storage -------------------------------------------------------------------------------------------------
    def has(self, p):
        return id(p) in self.offsets

    def view(self, p):
        """A fresh view tensor of p's gradient slot (autograd adopts a tensor nobody else references as .grad)."""
        o = self.offsets[id(p)]
        return self.flat[o:o + p.numel()].view(p.shape)

    def grad_ptr(self, p):
        return self.flat.data_ptr() + 4 * self.offsets[id(p)]

    def take(self, p):
        """p's slot for a weight-gradient kernel to write (beta = 0) -- once per step: a parameter used twice in one backward
        pass (shared weights) gets None the second time, the caller then writes a tensor of its own and autograd sums the two
        (collect() copies the sum into the slot)."""
        if id(p) in self._handed:
            return None
        self._handed.add(id(p))
        return self.view(p)

    def begin_step(self):
        """Zero the storage (one memset): slots of parameters that receive no gradient this step stay zero."""
        self.flat.zero_()
        self.issued, self.issued_early, self._next = [], 0, 0
        self._handed = set()
        self._fired, self._late = set(), []
        self._pending = None
        if self._early_cb is not None:
            exp = self._expected
            self._pending = [sum(1 for p in self.params[i0:i1] if exp is None or id(p) in exp) for (_, _, i0, i1) in self.buckets]

    # ---- early buckets -------------------------------------------------------------------------------------------
    def arm_early(self, callback):
        """callback(bucket_index) runs during backward once the next bucket in order is complete (see the module text)."""
        self._early_cb = callback

    def index_of(self, p):
        return self.index[id(p)]

    def backward_scope(self, active, early=True):
        """The trainer brackets loss.backward() with backward_scope(True) / (False): hooks outside it are ignored; early =
        False keeps the buckets for the caller's tail loop (steps whose all-reduces run elsewhere).  Leaving the scope
        uploads which parameters received gradients (`used`; also the next step's expectation) and checks that none of them
        arrived after its bucket had been released."""
        was = self._in_backward
        self._in_backward = bool(active)
        if active and not early:
            self._pending = None
        if was and not active:
            late, self._late = self._late, []
            self._expected = set(self._fired)
            self._used_host.zero_()
            idx = [self.index[i] for i in self._fired]
            if idx:
                self._used_host[idx] = 1
            self.used.copy_(self._used_host, non_blocking=True)
            if late:
                raise RuntimeError('GradStore: %d parameter(s) received a gradient after their bucket had been all-reduced '
                                   '(the set of parameters in use changed between steps); the step is incomplete -- rerun it' % len(late))

    def _on_grad(self, p):
        if not self._in_backward:
            return
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        self._fired.add(id(p))
        if self._pending is None:
            return
        if self._expected is not None and id(p) not in self._expected:
            if b < self._next:
                self._late.append(p)
            return
        self._pending[b] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            bi = self._next
            self._next += 1
            self.issued_early += 1
            self._early_cb(bi)

    def next_bucket(self):
        """First bucket the backward pass did not release (the trainer's tail loop continues from here)."""
        return self._next if self._pending is not None else 0

    def collect(self, bi=None):
        """Gradients autograd produced elsewhere (BatchNorm / bias / Linear / GCN parameters) are copied into their
        slots (multi-tensor copy) and .grad is pointed at the slot; conv weight gradients are already there."""
        rng = range(len(self.buckets)) if bi is None else (bi,)
        dst, src, moved = [], [], []
        for b in rng:
            _, _, i0, i1 = self.buckets[b]
            for p in self.params[i0:i1]:
                g = p.grad
                if g is None or g.data_ptr() == self.grad_ptr(p):
                    continue
                v = self.view(p)
                if g.dtype != torch.float32 or g.shape != p.shape:
                    g = g.to(torch.float32).view(p.shape)
                dst.append(v)
                src.append(g)
                moved.append((p, v))
        if dst:
            with torch.no_grad():
                torch._foreach_copy_(dst, src)
            for p, v in moved:
                p.grad = v

    def attach_all(self):
        """Point every parameter's .grad at its slot (parameters without a gradient this step read as zeros)."""
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.grad_ptr(p):
                p.grad = self.view(p)

    # ---- reduction -----------------------------------------------------------------------------------------------
    def reduce_bucket(self, bi):
        """Sum bucket bi over the ranks.  Asynchronous on accelerators: the collective runs on the process group's
        communication stream after everything queued so far on the current stream; call wait() before reading."""
        if self.world == 1 and self.group is None and not (dist.is_available() and dist.is_initialized()):
            return
        s, e, _, _ = self.buckets[bi]
        self.issued.append(bi)
        if self.wire is None:
            work = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.wire[s:e].copy_(self.flat[s:e])
            work = dist.all_reduce(self.wire[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, s, e))

    def wait(self):
        for w, s, e in self._works:
            w.wait()                    # (accelerator tensors: the current stream waits, the host does not block)
            if self.wire is not None:
                self.flat[s:e].copy_(self.wire[s:e])
        self._works = []

    def reduce_all(self):
        self.collect()
        for bi in range(len(self.buckets)):
            self.reduce_bucket(bi)
        self.wait()

    def scale_(self):
        """Turn the reduced sums into averages in place (FusedAdam folds grad_scale into its update instead)."""
        if self.world > 1:
            self.flat.mul_(self.grad_scale)

    def broadcast_parameters(self, module, src=0):
        """One-time broadcast of parameters and buffers (BatchNorm statistics) from rank `src`."""
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                    dist.broadcast(t.data, src=src, group=self.group)
