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
"""
import torch
import torch.distributed as dist


class GradStore(object):
    def __init__(self, params, bucket_mb=32.0, device=None, process_group=None, world=None):
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
        self._works = []

    # ---- storage -------------------------------------------------------------------------------------------------
    def has(self, p):
        return id(p) in self.offsets

    def view(self, p):
        """A fresh view tensor of p's gradient slot (autograd adopts a tensor nobody else references as .grad)."""
        o = self.offsets[id(p)]
        return self.flat[o:o + p.numel()].view(p.shape)

    def grad_ptr(self, p):
        return self.flat.data_ptr() + 4 * self.offsets[id(p)]

    def begin_step(self):
        """Zero the storage (one memset): slots of parameters that receive no gradient this step stay zero."""
        self.flat.zero_()

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
        work = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(work)

    def wait(self):
        for w in self._works:
            w.wait()                    # (accelerator tensors: the current stream waits, the host does not block)
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
