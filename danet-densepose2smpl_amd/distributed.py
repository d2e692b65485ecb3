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

Overlap with the backward pass.  The trainer runs the backward pass in SEGMENTS (segments.py: the autograd graph is cut at
the HRNet module boundaries and at the estimator -> regressor interface) and calls `release_ready(fn)` between two
segments: every bucket whose parameters all have their gradient (written, or queued as a deferred weight-gradient job) is
handed to fn -- the trainer launches the bucket's weight gradients and starts its all-reduce -- strictly in index order (a
bucket that completes before its predecessor waits for it), so every rank issues the same sequence of collectives from
its main thread; the regressor / limb-net buckets are on the wire while the HRNet backward is still running, and buckets
that do not complete before the last segment are finished by the trainer's tail loop, in order as well.  (Rounds 2-3
released buckets from post-accumulate-grad hooks inside one loss.backward(): collectives issued from autograd's device
thread aborted the process sporadically and could not be captured; the hooks now only count.)  Parameters that received no
gradient in the previous step (never-used modules: rot2pos / pos2rot, a skipped regressor) are not waited for -- the first
step, which knows nothing yet, releases nothing early around them; should such a parameter receive a gradient after its
bucket has gone out, backward_scope(False) raises instead of training on an incomplete sum.

Which parameters are in use is a GLOBAL fact: the flat buffer ends in one float per parameter ("fired on this rank"), which
travels with the last bucket's all-reduce; the optimizer skips a parameter only if no rank produced a gradient for it
(`used`), so replicas cannot diverge on data-dependent branches.

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
        # the last bucket ends in one float per parameter: "received a gradient on this rank in this step" (see `used`), and one
        # more for "this rank's step is poisoned" (see `poison`)
        npad = (len(self.params) + 3) // 4 * 4
        self.buckets.append((start, off + npad + 4, first, len(self.params)))
        self.flat = torch.zeros(off + npad + 4, dtype=torch.float32, device=self.device)
        self.wire_dtype = wire_dtype
        self.wire = None if wire_dtype == torch.float32 else torch.zeros(off + npad + 4, dtype=wire_dtype, device=self.device)
        # Whether the step's gradients are garbage is a GLOBAL fact as well: a rank whose one-pass BatchNorm barrier timed out
        # (poison_src: its error word, a one-element integer tensor the trainer names) has already added that garbage to every
        # rank's sums when the optimizer runs.  stamp_poison() writes the word behind the mask, the last bucket's all-reduce sums it,
        # and optim.FusedAdam skips the step wherever the sum is > 0 -- every rank or none.
        self.poison = self.flat[off + npad:off + npad + 1]
        self.poison_src = None
        self.index = {id(p): i for i, p in enumerate(self.params)}
        # In how many ranks each parameter (self.params order) received a gradient in the last backward pass: written from
        # the post-accumulate-grad hooks when the trainer leaves backward_scope, summed over the ranks by the last bucket's
        # all-reduce; the optimizer skips a parameter only where this is 0 (optim.FusedAdam), on every rank alike.
        self.used = self.flat[off:off + len(self.params)]
        self.used.fill_(1.0)
        self._grads = self.flat[:off]
        self._mask_fresh = False            # this step's mask has been uploaded (backward_scope); else every parameter counts as in use
        # two pinned host copies of the mask, used alternately and rewritten only when the set of parameters in use changes:
        # the asynchronous upload of one step (and the memcpy node of a captured graph, which re-reads its buffer on every
        # replay) never sees a half-written mask
        self._mask_host = [torch.ones(len(self.params), dtype=torch.float32) for _ in range(2)]
        if self.device.type == 'cuda':
            self._mask_host = [m.pin_memory() for m in self._mask_host]
        self._mask_set = [None, None]
        self._mask_event = [None, None]
        self._mask_turn = 0
        self._mask_captured = []            # pinned masks owned by captured graphs (one per capture)
        self._mask_for_capture = []         # buffers for the next captures (prepare_capture)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._works = []
        self.issued = []                    # bucket indices in the order their collectives were issued this step ...
        self.issued_early = 0               # ... and how many of them before the backward pass had finished
        self._handed = set()
        self._expected = None               # ids of the parameters that received a gradient in the previous step
        self._fired, self._late = set(), []
        self._pending = None
        self._next = 0
        self._in_backward = False

    def close(self):
        """Detach from the parameters (hooks) -- call when a trainer replaces its store; the flat buffer is freed with the
        last reference."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

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
        self._grads.zero_()
        self._mask_fresh = False
        self.issued, self.issued_early, self._next = [], 0, 0
        self._handed = set()
        self._fired, self._late = set(), []
        exp = self._expected
        self._pending = [sum(1 for p in self.params[i0:i1] if exp is None or id(p) in exp) for (_, _, i0, i1) in self.buckets]

    def index_of(self, p):
        return self.index[id(p)]

    # ---- the backward pass ---------------------------------------------------------------------------------------
    def backward_scope(self, active, early=True):
        """The trainer brackets the backward pass with backward_scope(True) / (False): hooks outside it are ignored; early =
        False keeps every bucket for the caller's tail loop (steps whose all-reduces run elsewhere).  Leaving the scope
        uploads which parameters received gradients (`used`; also the next step's expectation) and checks that none of them
        arrived after its bucket had been released."""
        was = self._in_backward
        self._in_backward = bool(active)
        if active and not early:
            self._pending = None
        if was and not active:
            late, self._late = self._late, []
            self._expected = set(self._fired)
            self._upload_mask()
            if late:
                raise RuntimeError('GradStore: %d parameter(s) received a gradient after their bucket had been all-reduced '
                                   '(the set of parameters in use changed between steps); the step is incomplete -- rerun it' % len(late))

    def prepare_capture(self, attempts=1):
        """Before the hipGraph capture(s) of a step: the pinned host buffer each capture's usage-mask upload will read on every
        replay (pinned memory cannot be allocated while a stream is capturing -- nor, it turned out, right after a capture that
        failed: one buffer per attempt, all of them up front)."""
        while len(self._mask_for_capture) < attempts:
            m = torch.ones(len(self.params), dtype=torch.float32)
            self._mask_for_capture.append(m.pin_memory() if self.device.type == 'cuda' else m)

    def stamp_poison(self):
        """Copy this rank's poison word into the tail of the last bucket (after the backward pass, before that bucket's
        all-reduce); without a source the slot stays 0."""
        if self.poison_src is not None:
            self.poison.copy_(self.poison_src)

    def _upload_mask(self):
        fired = frozenset(self.index[i] for i in self._fired)
        if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            # a hipGraph capture: the memcpy node re-reads its host buffer on EVERY replay, so the graph gets a pinned buffer of its
            # own that no eager step ever rewrites (kept alive here), and no event is recorded (an event recorded in a capture cannot
            # be synchronized on later: hipErrorCapturedEvent) -- ADVICE r4.  The set of parameters in use is frozen into the graph
            # like the batch's host-side switches (Trainer.load_batch): capture() again when it changes (pretrain -> full model).
            if not self._mask_for_capture:
                raise RuntimeError('GradStore: call prepare_capture() before capturing a step (pinned memory cannot be allocated inside a capture)')
            m = self._mask_for_capture.pop()
            m.zero_()
            if fired:
                m[sorted(fired)] = 1.0
            self._mask_captured.append(m)
            self.used.copy_(m, non_blocking=True)
            self._mask_fresh = True
            return
        k = self._mask_turn
        if self._mask_set[k] != fired:
            k ^= 1
            if self._mask_set[k] != fired:
                if self._mask_event[k] is not None:
                    self._mask_event[k].synchronize()          # the upload that last read this buffer has finished
                m = torch.zeros(len(self.params), dtype=torch.float32)
                if fired:
                    m[sorted(fired)] = 1.0
                self._mask_host[k].copy_(m)                    # (one pass over a complete mask, never a zeroed intermediate)
                self._mask_set[k] = fired
            self._mask_turn = k
        self.used.copy_(self._mask_host[k], non_blocking=True)
        self._mask_fresh = True
        if self.device.type == 'cuda':
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._mask_event[k] = ev

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

    def release_ready(self, fn):
        """Between two segments of the backward pass (main thread): fn(bucket) for every bucket that is complete, strictly in
        index order.  The last bucket is never released here: it carries the usage mask, which is only known once the
        backward pass has finished.  Returns the number of buckets released."""
        n = 0
        if self._pending is None or not self._in_backward:
            return n
        while self._next < len(self.buckets) - 1 and self._pending[self._next] == 0:
            bi = self._next
            self._next += 1
            self.issued_early += 1
            fn(bi)
            n += 1
        return n

    def release_ready_group(self, fn):
        """release_ready for callers that take a RUN of buckets at once: fn(b0, b1) for the complete buckets b0 .. b1 - 1 (consecutive
        by construction) -- one weight-gradient flush, one multi-tensor copy and ONE all-reduce over the run's contiguous slice of the
        store instead of one of each per bucket (round 6: the regressor segment alone completes five 32 MB buckets at once; 13 bucket-sized
        flushes cost the N > 1 path 0.5 ms/step over the single flush of a one-process step).  Same order on every rank as release_ready:
        which buckets are complete after a segment depends on the model and on the global usage mask only."""
        if self._pending is None or not self._in_backward:
            return 0
        b0 = self._next
        while self._next < len(self.buckets) - 1 and self._pending[self._next] == 0:
            self._next += 1
            self.issued_early += 1
        if self._next > b0:
            fn(b0, self._next)
        return self._next - b0

    def next_bucket(self):
        """First bucket the backward pass did not release (the trainer's tail loop continues from here)."""
        return self._next if self._pending is not None else 0

    def collect(self, bi=None):
        """Gradients autograd produced elsewhere (BatchNorm / bias / Linear / GCN parameters) are copied into their
        slots (multi-tensor copy) and .grad is pointed at the slot; conv weight gradients are already there."""
        rng = range(len(self.buckets)) if bi is None else ((bi,) if isinstance(bi, int) else bi)
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
        if bi == len(self.buckets) - 1 and not self._mask_fresh:
            self.used.fill_(1.0)            # a step outside backward_scope: no mask, every parameter with a gradient pointer is updated
        if self.wire is None:
            work = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.wire[s:e].copy_(self.flat[s:e])
            work = dist.all_reduce(self.wire[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, s, e))

    def reduce_buckets(self, b0, b1):
        """reduce_bucket for the run b0 .. b1 - 1 as ONE collective: the run's slots are contiguous in the flat store."""
        if b1 - b0 == 1:
            return self.reduce_bucket(b0)
        if self.world == 1 and self.group is None and not (dist.is_available() and dist.is_initialized()):
            return
        s, e = self.buckets[b0][0], self.buckets[b1 - 1][1]
        self.issued.extend(range(b0, b1))
        if b1 == len(self.buckets) and not self._mask_fresh:
            self.used.fill_(1.0)
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

    def quiesce(self):
        """Call (after a device synchronize) before a hipGraph capture that will hold collectives.  The process group's
        watchdog thread polls the end event of every EAGER collective until it has seen it complete, at its own pace (every
        ~100 ms).  If a capture begins while such work is still on its list and the communication stream joins the capture, the
        poll fails with hipErrorCapturedEvent, the watchdog thread throws, and the process dies with SIGABRT (rounds 2-4: 'one
        start in eight', 2 of 6 in tools/ddp_abort_probe.sh; test-sized models reach their first captured collective within the
        polling interval, the full-size step does not -- which is why only some configurations ever aborted; never with this
        call: 0 of 6 with a 0.5 s pause, 0 of 12 with the explicit wait).  Blocks until the list is empty."""
        if not (dist.is_available() and dist.is_initialized()) or self.device.type != 'cuda':
            return
        try:
            pg = self.group if self.group is not None else dist.group.WORLD
            pg._get_backend(torch.device(self.device))._wait_for_pending_works()
        except (AttributeError, RuntimeError):
            import time
            time.sleep(0.5)               # (a build without the call: several polling intervals)

    def reduce_all(self):
        self.collect()
        for bi in range(len(self.buckets)):
            self.reduce_bucket(bi)
        self.wait()

    def scale_(self):
        """Turn the reduced sums into averages in place (FusedAdam folds grad_scale into its update instead)."""
        if self.world > 1:
            self.flat.mul_(self.grad_scale)

    def broadcast_parameters(self, module, src=0, chunk_mb=64.0):
        """One-time broadcast of parameters and buffers (BatchNorm statistics) from rank `src`: tensors of one dtype are
        packed into chunks of ~chunk_mb MB -- about a dozen collectives for the whole model instead of one per tensor."""
        by_dtype = {}
        for t in list(module.parameters()) + list(module.buffers()):
            if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
                by_dtype.setdefault(t.dtype, []).append(t.data)
        with torch.no_grad():
            for dt, ts in by_dtype.items():
                cap = max(1, int(chunk_mb * 1024 * 1024 / ts[0].element_size()))
                grp, n = [], 0
                for t in ts + [None]:
                    if t is None or (grp and n + t.numel() > cap):
                        flat = torch.cat([g.reshape(-1) for g in grp])
                        dist.broadcast(flat, src=src, group=self.group)
                        o = 0
                        for g in grp:
                            g.copy_(flat[o:o + g.numel()].view_as(g))
                            o += g.numel()
                        grp, n = [], 0
                    if t is not None:
                        grp.append(t)
                        n += t.numel()
