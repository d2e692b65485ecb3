"""Training step with the contract of the reference's Trainer.train_step
(/root/reference/train/trainer.py:117-244): ``train_step(in_dict) -> (output, losses)`` =
forward, sum of the loss dict (trainer.py:217-220), backward, Adam step (lr 1e-4, trainer.py:42-44).
The in_dict is the one the reference builds at trainer.py:184-212; `synthetic_in_dict` produces it
from seeded random labels.  Data loading, FitsDict, TensorBoard and
checkpoints are out of scope."""
import types

import numpy as np
import contextlib
import os

import torch

from . import ops
from .config import cfg
from .danet import DaNet
from .distributed import GradStore
from .nn import BatchNorm2d, bump_batch_counters
from . import conv as _conv
from .geometry import perspective_projection, label_prologue


DEFER_WGRAD = bool(int(os.environ.get('DANET_DEFER_WGRAD', '1')))
USE_FUSED_ADAM = bool(int(os.environ.get('DANET_FUSED_ADAM', '1')))
# Data-parallel steps run the backward pass in segments (segments.py) and release complete gradient buckets between them:
# all-reduces overlap the backward pass proper, issued by this thread in program order (eager and captured alike).
# DANET_SEGMENTS=0 switches the cuts off (every bucket after the backward pass: round 2's order); DANET_SEGMENTS=1 forces them
# on in a single-process trainer (tests, A/B timing).
SEGMENTS = os.environ.get('DANET_SEGMENTS', '')
# Experiment knob (round 5, OFF): one process launches the queued weight gradients of a finished backward segment on a SIDE stream while
# the next segment's data-gradient / BatchNorm chain continues on the step's stream -- the chain's kernels are latency-bound and leave
# most of the chip idle, the weight gradients need nothing from it.  Measured on MI355X inside the captured step (bench.py, one gpurun
# call): 27.9 ms without; with it 1 842 ms -- the one-pass BatchNorm backward's grid barrier needs all its workgroups resident at once
# and a chip-filling weight-gradient kernel beside it makes the barrier run into its spin bound --; with the two-kernel
# BatchNorm backward instead 28.86 ms against 28.47 without overlap, and 57 ms with the step's stream at high priority.  Graph
# branches do not buy concurrency on this stack: the step stays one stream.
WGRAD_OVERLAP = bool(int(os.environ.get('DANET_WGRAD_OVERLAP', '0')))
# Compute units the communication library may occupy while the backward pass runs (one workgroup per channel): the one-pass
# BatchNorm backward's grid barrier is sized to fit beside them (nn.ONEPASS_MAX_BLOCKS).  The limit is the LAUNCHER's to set --
# NCCL_MAX_NCHANNELS must be in the environment before the first collective creates the communicator: bench.py calls
# reserve_comm_channels() before init_process_group.  A Trainer never touches the environment: it reads the limit,
# and without one it does not know how many compute units the all-reduce kernels may take, so its data-parallel steps run the
# two-kernel BatchNorm backward (a warning says so).  Trade-off: 24 channels leave the all-reduce 24 of 256 compute units;
# DANET_COMM_CHANNELS chooses another number (tools/scale_sweep.sh sweeps it), NCCL_MAX_NCHANNELS set by hand wins.
COMM_CHANNELS = int(os.environ.get('DANET_COMM_CHANNELS', '24'))
# Buckets that complete together are flushed / copied / all-reduced as ONE run (GradStore.release_ready_group); 0: one bucket at a time (A-B)
GROUP_RELEASE = bool(int(os.environ.get('DANET_GROUP_RELEASE', '1')))


def reserve_comm_channels():
    """For LAUNCH scripts, before torch.distributed.init_process_group: export the channel limit (if none is set) and return the
    number of compute units kept free for collectives."""
    cur = os.environ.get('NCCL_MAX_NCHANNELS')
    if cur is None:
        os.environ['NCCL_MAX_NCHANNELS'] = str(COMM_CHANNELS)
        return COMM_CHANNELS
    return max(1, int(cur))


def comm_channel_limit():
    """The channel limit the environment names (what the communicator was, or will be, created with), or None."""
    cur = os.environ.get('NCCL_MAX_NCHANNELS')
    return None if cur is None else max(1, int(cur))


def default_options(batch_size=32):
    return types.SimpleNamespace(batch_size=batch_size, openpose_train_weight=0., gt_train_weight=1.)


def synthetic_in_dict(model, B, device, seed=1234, img_size=None, with_dp=False):
    """Seeded synthetic training batch with the reference's keys and shapes (trainer.py:184-212;
    datasets/base_dataset.py:228-298)."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    S = img_size or cfg.DANET.INIMG_SIZE
    img = torch.randn(B, 3, S, S, generator=g).to(device)
    betas = torch.randn(B, 10, generator=g).clamp(-3, 3).to(device)
    pose = (torch.randn(B, 72, generator=g) * 0.2).to(device)
    cam = torch.stack([torch.rand(B, generator=g) * 0.5 + 0.6, torch.rand(B, generator=g) * 0.2 - 0.1,
                       torch.rand(B, generator=g) * 0.2 - 0.1], dim=1).to(device)
    smpl = model.iuv2smpl.smpl
    with torch.no_grad():
        out = smpl(betas=betas, body_pose=pose[:, 3:], global_orient=pose[:, :3])
        cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * 5000. / (S * cam[:, 0] + 1e-9)], dim=-1)
        zero_c = torch.zeros(B, 2, device=device)
        kp2d = perspective_projection(out.joints, None, cam_t, 5000., zero_c) / (S / 2.)
        conf = torch.cat([torch.zeros(B, 25, 1, device=device), torch.ones(B, 24, 1, device=device)], dim=1)
        keypoints = torch.cat([kp2d, conf], dim=-1)
        pose_3d = torch.cat([out.joints[:, 25:], torch.ones(B, 24, 1, device=device)], dim=-1)
        skps = perspective_projection(out.smpl_joints, None, cam_t, 5000., zero_c) / (S / 2.)
        target_smpl_kps = torch.cat([skps, torch.ones(B, 24, 1, device=device)], dim=-1)
    ones = torch.ones(B, device=device)
    extra = {}
    if with_dp:
        # the zero DensePose blobs every non-COCO sample carries (datasets/base_dataset.py:228-237); 'dp_active' is the
        # host-side twin of the reference's `torch.sum(has_dp) > 0` test
        Sh = S // 4
        z = lambda n: torch.zeros(B, n, device=device)      # noqa: E731
        extra['dp_dict'] = {'body_uv_ann_labels': torch.zeros(B, Sh * Sh, dtype=torch.int32, device=device),
                            'body_uv_ann_weights': z(Sh * Sh), 'body_uv_X_points': z(196), 'body_uv_Y_points': z(196),
                            'body_uv_Ind_points': z(196), 'body_uv_I_points': z(196), 'body_uv_U_points': z(4900),
                            'body_uv_V_points': z(4900), 'body_uv_point_weights': z(4900), 'dp_active': False}
    return {**extra, 'img': img, 'opt_pose': pose, 'opt_betas': betas, 'keypoints': keypoints, 'pose_3d': pose_3d,
            'has_pose_3d': ones.clone(), 'valid_fit': ones.clone(), 'has_iuv': ones.clone(), 'has_dp': torch.zeros(B, device=device),
            'target_smpl_kps': target_smpl_kps, 'target_verts': out.vertices.detach(), 'target_cam': cam,
            'vis_on': False, 'pretrain_mode': False}


class Trainer(object):
    """Single-process-per-GPU trainer.  Gradients live in one flat store (distributed.GradStore); with world_size > 1
    its buckets are all-reduced (RCCL) while the rest of the backward pass and the remaining weight gradients of the step
    are still being computed (segments.py)."""

    def __init__(self, options=None, model=None, device=None, lr=None, distributed=None, smpl_model=None, bucket_mb=None, grad_wire=None):
        self.options = options or default_options()
        self.device = device or torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.model = (model or DaNet(self.options, None, pretrained=False, smpl_model=smpl_model)).to(self.device)
        self.smpl = self.model.iuv2smpl.smpl
        on_gpu = self.device.type == 'cuda'
        lr0 = lr or cfg.SOLVER.BASE_LR
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1
        self.distributed = bool(distributed)
        # a tensor learning rate keeps the manual step decay (trainer.py:120-128) effective under hipGraph replay
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.params = params
        self.store = None
        if on_gpu and USE_FUSED_ADAM:
            from .optim import FusedAdam
            if bucket_mb is None:
                bucket_mb = float(os.environ.get('DANET_BUCKET_MB', '32'))
            wire = torch.bfloat16 if (grad_wire or os.environ.get('DANET_GRAD_WIRE', 'fp32')) == 'bf16' else torch.float32
            self.store = GradStore(params, bucket_mb=bucket_mb, device=self.device,
                                   world=torch.distributed.get_world_size() if self.distributed else 1,
                                   wire_dtype=wire if self.distributed else torch.float32)
            self.optimizer = FusedAdam(params, lr=lr0, grad_store=self.store)      # one HIP launch per step (csrc/adam.hip)
        else:
            if self.distributed:
                raise RuntimeError('data-parallel training needs the GPU path (FusedAdam + GradStore)')
            self.optimizer = torch.optim.Adam(params=params, lr=torch.tensor(lr0, device=self.device) if on_gpu else lr0,
                                              weight_decay=0, **({'capturable': True, 'fused': True} if on_gpu else {}))
        self.step_count = 0
        self.bank = None
        # Every step (eager or captured) runs on ONE dedicated stream: autograd pins each parameter's
        # gradient-accumulation node to the stream it was created on, and a node created on the default
        # stream would pull that (non-capturing) stream into a later hipGraph capture.
        self.stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('DANET_MAIN_PRIORITY', '0'))) if on_gpu else None
        self.wgrad_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('DANET_WGRAD_PRIORITY', '0'))) if on_gpu else None
        self._held = []
        self.segmented = (self.distributed and SEGMENTS != '0') or SEGMENTS == '1'
        if on_gpu:
            _conv.ARENA.enable(self.device)
            from . import nn as _nn
            _nn.ONEPASS_STREAM = self.stream        # the one-pass BatchNorm backward is confined to the step's own stream
            if isinstance(self.optimizer, torch.optim.Optimizer) is False:
                # a step whose one-pass BatchNorm backward timed out at its grid barrier is skipped by the Adam kernel itself
                self.optimizer.poison = _nn.onepass_poison(self.device)
                if self.distributed:
                    # ... and, since its gradients are in every rank's sums by then, by every other rank's as well: the word travels
                    # with the last gradient bucket (GradStore.stamp_poison) and the kernel tests the all-reduced sum
                    self.store.poison_src = self.optimizer.poison
                    self.optimizer.poison_sum = self.store.poison
            # workgroup budget of the one-pass BatchNorm backward in THIS trainer's steps (installed by _core for the duration of a
            # step): 0 = the whole device; data-parallel: all-reduce kernels of earlier buckets run beside the backward pass, so the grid
            # barrier must fit next to them -- 2 * (compute units - channel limit); -1 = no limit known: two-kernel path
            self.onepass_blocks = 0
            if os.environ.get('DANET_ONEPASS_BLOCKS'):            # experiment knob: an explicit budget
                self.onepass_blocks = int(os.environ['DANET_ONEPASS_BLOCKS'])
            elif self.distributed:
                ch = comm_channel_limit()
                if ch is None:
                    import warnings
                    warnings.warn('data-parallel Trainer without NCCL_MAX_NCHANNELS in the environment: the number of compute units the '
                                  'all-reduce kernels may occupy is unknown, so the one-pass BatchNorm backward (a grid barrier over 2 workgroups '
                                  'per compute unit) is off for this trainer; call trainer.reserve_comm_channels() before init_process_group')
                    self.onepass_blocks = -1
                else:
                    cus = torch.cuda.get_device_properties(self.device).multi_processor_count
                    self.onepass_blocks = max(2, 2 * (cus - ch))
        self._graph = None
        self._static = None
        self._reduce_in_graph = True
        if self.distributed:
            self.store.broadcast_parameters(self.model)

    @property
    def reducer(self):          # (name kept for callers of the first version)
        return self.store if self.distributed else None

    @torch.no_grad()
    def prepare_batch(self, input_batch, opt_pose=None, opt_betas=None, img_res=None, focal_length=5000.):
        """The reference's step prologue (train/trainer.py:134-212) on the device -- SURVEY.md 8 row f1.
        input_batch: img, keypoints [B,49,3] (normalised to [-1,1], confidence), pose [B,72], betas [B,10], pose_3d [B,24,4],
        has_smpl, has_pose_3d, has_dp, optional smpl_2dkps / dp_dict / has_iuv_dataset (the reference's
        `dataset_name != 'dp_coco'`) / valid_fit.  opt_pose / opt_betas: the pseudo-label fits (FitsDict in the
        reference; default: the ground truth).  Two SMPL forwards (HIP), one batched least-squares solve, no host
        round trip.  Returns the in_dict `train_step` takes."""
        dev = self.device
        res = img_res or cfg.DANET.INIMG_SIZE
        B = input_batch['img'].shape[0]
        gt_pose, gt_betas = input_batch['pose'].to(dev), input_batch['betas'].to(dev)
        has_smpl = input_batch['has_smpl'].to(dev) > 0
        opt_pose = gt_pose.clone() if opt_pose is None else opt_pose.to(dev).clone()
        opt_betas = gt_betas.clone() if opt_betas is None else opt_betas.to(dev).clone()
        opt_betas = torch.where((opt_betas.abs() > 3).any(dim=-1, keepdim=True), torch.zeros_like(opt_betas), opt_betas)
        opt_pose = torch.where(has_smpl.view(B, 1), gt_pose, opt_pose)
        opt_betas = torch.where(has_smpl.view(B, 1), gt_betas, opt_betas)
        opt = self.smpl(betas=opt_betas, body_pose=opt_pose[:, 3:], global_orient=opt_pose[:, :3])
        valid_fit = input_batch['valid_fit'].to(dev) > 0 if 'valid_fit' in input_batch else has_smpl
        has_iuv = valid_fit & (input_batch['has_iuv_dataset'].to(dev) > 0 if 'has_iuv_dataset' in input_batch
                               else torch.ones(B, dtype=torch.bool, device=dev))
        has_dp = input_batch.get('has_dp', torch.zeros(B, device=dev)).to(dev)
        kps, cam, _ = label_prologue(opt.joints, opt.smpl_joints, input_batch['keypoints'].to(dev), has_iuv, has_dp,
                                     input_batch.get('smpl_2dkps', None) if input_batch.get('smpl_2dkps', None) is None
                                     else input_batch['smpl_2dkps'].to(dev), focal_length, res)
        out = {'img': input_batch['img'].to(dev), 'opt_pose': opt_pose, 'opt_betas': opt_betas,
               'keypoints': input_batch['keypoints'].to(dev), 'pose_3d': input_batch['pose_3d'].to(dev),
               'has_pose_3d': input_batch['has_pose_3d'].to(dev).float(), 'valid_fit': valid_fit.float(),
               'has_iuv': has_iuv.float(), 'has_dp': has_dp.float(), 'target_smpl_kps': kps,
               'target_verts': opt.vertices.detach(), 'target_cam': cam,
               'vis_on': bool(input_batch.get('vis_on', False)), 'pretrain_mode': bool(input_batch.get('pretrain_mode', False))}
        if 'dp_dict' in input_batch:
            out['dp_dict'] = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in input_batch['dp_dict'].items()}
            out['dp_dict'].setdefault('dp_active', bool((has_dp > 0).any()))        # (one host read per batch, outside the step)
        return out

    ONEPASS_CHECK_EVERY = 64       # steps between looks at the one-pass BatchNorm's barrier error flag (one 4-byte read each)

    def check_onepass(self):
        """The one-pass BatchNorm backward (csrc/norm_act.hip) spins at a grid-wide barrier with a bound; a time-out -- a GPU
        shared with another process, a partitioned device -- leaves garbage gradients and a stale barrier.  The Adam kernel
        reads the barrier's error word and skips such a step on the device (csrc/adam.hip `poison`), so nothing is ever
        applied; this host-side check (every ONEPASS_CHECK_EVERY steps and before a checkpoint is written) is for reporting and
        recovery: the barrier is reset, the one-pass path switched off (two-kernel BatchNorm backward from here on), a
        captured graph dropped, and the caller told.  Data-parallel: the decision is taken from the all-reduced poison word
        (GradStore.poison: the error word is sticky, so the last step's sum tells whether ANY rank has timed out since the last
        check) -- every rank recovers and raises in the same call, none is left waiting in a collective."""
        from . import nn as dnn
        if self.device.type != 'cuda':
            return
        elsewhere = bool(self.distributed and self.store is not None and self.store.poison_src is not None and float(self.store.poison) > 0)
        if dnn.onepass_recover(self.device, force=elsewhere):
            self._graph = None
            guarded = getattr(self.optimizer, 'poison', None) is not None
            raise RuntimeError('a one-pass BatchNorm launch timed out at its grid barrier' + (' on some rank' if elsewhere else '') +
                               ': gradients since then are invalid'
                               + (' -- the optimizer skipped those steps on every rank (device-side guard on the all-reduced error word), '
                                  'parameters and moments are intact. '
                                  if guarded else ' and were applied: resume from the last checkpoint. ') +
                               'The barrier was reset and the one-pass path switched off (nn.ONEPASS = False); capture() again if you '
                               'replay a graph')

    def save(self, path, epoch=0, batch_idx=0, batch_size=0, dataset_perm=None):
        """A training checkpoint in the reference's format (utils/saver.py:24-50): model, optimizer, bookkeeping with
        `total_step_count` = this trainer's step count."""
        from . import checkpoint
        self.check_onepass()
        return checkpoint.save_checkpoint(path, {'model': self.model}, {'optimizer': self.optimizer}, epoch=epoch, batch_idx=batch_idx,
                                          batch_size=batch_size, dataset_perm=dataset_perm, total_step_count=self.step_count)

    def resume(self, path, trusted=False):
        """Continue from a checkpoint (base_trainer.py:37-51): parameters, BatchNorm buffers, Adam moments and learning rate
        come from the file, and `step_count` -- which drives the step-LR decay -- from its `total_step_count`.  Works
        with a captured graph too: the tensors are updated in place and the weight repack is part of the graph.
        Returns the bookkeeping values (epoch, batch_idx, ...)."""
        from . import checkpoint
        book = checkpoint.load_checkpoint(path, {'model': self.model}, {'optimizer': self.optimizer},
                                          map_location=self.device, trusted=trusted)
        self.step_count = int(book.get('total_step_count') or 0)
        _conv._PACK_CACHE.clear()
        return book

    def _decay_lr(self):
        """manual step-LR decay (trainer.py:120-128)"""
        for i, s in enumerate(cfg.SOLVER.STEPS):
            if i > 0 and self.step_count == s:
                for group in self.optimizer.param_groups:
                    if torch.is_tensor(group['lr']):
                        group['lr'].mul_(cfg.SOLVER.GAMMA)
                    else:
                        group['lr'] *= cfg.SOLVER.GAMMA

    def _begin_step(self):
        """Per-step device housekeeping: zero the accumulator arena, repack all conv weights (one launch).
        The first GPU step records which packed weights the model asks for and builds the bank."""
        _conv.ARENA.begin_step()
        if self.device.type == 'cuda' and not os.environ.get('DANET_NO_BANK'):
            if self.bank is None:
                self.bank = _conv.WeightBank()
                self.bank.start_recording()
                _conv.RECORDER = self.bank
            elif self.bank.requests is not None:
                self.bank.build()
            self.bank.refresh()

    def _abort_cleanup(self):
        """Host-side state an aborted step (a failed graph capture) leaves behind: BatchNorm call counters of the modules
        that ran, gradients that point into the aborted pool.  (The arena cursor and the weight bank are re-armed by the
        next step's _begin_step; queued weight-gradient jobs are dropped by the caller.)"""
        BatchNorm2d._ran = []
        BatchNorm2d.count_batches = True
        _conv.DEFER_WGRAD = False
        _conv.GRAD_STORE = None
        self.optimizer.zero_grad(set_to_none=True)

    @contextlib.contextmanager
    def _on_stream(self):
        if self.stream is None:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            yield
        cur.wait_stream(self.stream)

    def _core(self, batch, reduce=True, with_optimizer=True):
        """One optimisation step on the current stream: forward; backward with the weight gradients queued -- in segments
        when data-parallel (segments.py), releasing every complete gradient bucket between two segments: the bucket's
        weight-gradient launches, its small gradients copied into the store, its all-reduce (asynchronous, on the
        communication stream: the next segment's backward kernels overlap it); then the remaining buckets the same way; Adam
        waits for the last all-reduce."""
        from . import segments
        from . import nn as _nn
        st = self.store
        prev_blocks, _nn.ONEPASS_MAX_BLOCKS = _nn.ONEPASS_MAX_BLOCKS, getattr(self, 'onepass_blocks', 0)
        _nn.SIDE_LIVE = 0                      # (a backward pass that raised inside a side-stream window leaves it open)
        try:
            return self._core_body(batch, reduce, with_optimizer, segments, st)
        finally:
            _nn.ONEPASS_MAX_BLOCKS = prev_blocks

    def _core_body(self, batch, reduce, with_optimizer, segments, st):
        self._begin_step()
        if st is not None:
            st.begin_step()
        BatchNorm2d.count_batches = False
        reduce_now = bool(self.distributed and reduce)
        overlap = bool(WGRAD_OVERLAP and DEFER_WGRAD and st is not None and not self.distributed and self.wgrad_stream is not None)
        segments.begin((self.segmented or overlap) and st is not None)
        try:
            try:
                out = self.model(batch)
            finally:
                BatchNorm2d.count_batches = True
            bump_batch_counters(self.model)
            losses = out['losses']
            self.optimizer.zero_grad(set_to_none=True)
            _conv.GRAD_STORE = st
            _conv.DEFER_WGRAD = DEFER_WGRAD
            if st is not None:
                st.backward_scope(True, early=reduce_now)
            try:
                if segments.level() > 0:
                    segments.backward(losses, (lambda k: st.release_ready_group(self._release_buckets)) if reduce_now else
                                      ((lambda k: self._flush_on_side_stream()) if overlap else None))
                else:
                    # (every loss is a 1-element tensor, models/danet/danet.py:359-364: views + one cat + one sum)
                    torch.cat([v.reshape(-1) for v in losses.values()]).sum().backward()
            finally:
                # first disarm the module globals, then close the scope (which may raise on a late gradient): an aborted step
                # must not leave queued weight-gradient jobs or pending collectives behind for the rerun
                _conv.DEFER_WGRAD = False
                _conv.GRAD_STORE = None
                if st is not None:
                    try:
                        st.backward_scope(False)
                        st.stamp_poison()
                    except RuntimeError:
                        _conv._WQ.clear()
                        _conv._WQG.clear()
                        st._works = []
                        raise
        finally:
            segments.end()
        if st is None:
            _conv.flush_wgrads()
        else:
            _conv.GRAD_STORE = st
            try:
                if reduce_now:
                    if st.next_bucket() < len(st.buckets):                   # what the backward pass did not release, as one run
                        self._release_buckets(st.next_bucket(), len(st.buckets))
                else:
                    # one process: no all-reduce to overlap with, so all queued weight gradients go out in the fewest, largest
                    # multi-problem launches (-0.15 ms against 13 bucket-sized flushes) -- or, with WGRAD_OVERLAP, segment by segment on
                    # the side stream, which joins the step's stream here
                    if overlap:
                        self._flush_on_side_stream()
                        torch.cuda.current_stream(self.device).wait_stream(self.wgrad_stream)
                        del self._held[:]
                    _conv.flush_wgrads()
                    st.collect()
                _conv.flush_wgrads()                     # (nothing left: every parameter belongs to a bucket)
            finally:
                _conv.GRAD_STORE = None
            if reduce_now:
                st.wait()
        if with_optimizer:
            self.optimizer.step()
        return out, losses

    def _flush_on_side_stream(self):
        """Launch everything queued so far (conv.flush_wgrads) on the weight-gradient stream, ordered after the work already on
        the step's stream; the operands stay referenced (self._held) until the streams join at the end of the backward pass."""
        cur = torch.cuda.current_stream(self.device)
        self.wgrad_stream.wait_stream(cur)
        with torch.cuda.stream(self.wgrad_stream):
            _conv.flush_wgrads(hold=self._held)

    def _release_bucket(self, bi):
        """A complete gradient bucket: its queued weight gradients are launched, its other gradients copied into the store, and
        its all-reduce started (GradStore.release_ready between two backward segments, or the tail loop)."""
        st = self.store
        prev, _conv.GRAD_STORE = _conv.GRAD_STORE, st
        try:
            _conv.flush_wgrads(bucket=bi)
        finally:
            _conv.GRAD_STORE = prev
        st.collect(bi)
        st.reduce_bucket(bi)

    def _release_buckets(self, b0, b1):
        """_release_bucket for the consecutive complete buckets b0 .. b1 - 1 at once: one flush of their queued weight gradients
        (fewer, larger multi-problem launches), one multi-tensor copy, one all-reduce over their contiguous slice of the store."""
        if not GROUP_RELEASE:
            for bi in range(b0, b1):
                self._release_bucket(bi)
            return
        st = self.store
        run = range(b0, b1)
        prev, _conv.GRAD_STORE = _conv.GRAD_STORE, st
        try:
            _conv.flush_wgrads(bucket=run)
        finally:
            _conv.GRAD_STORE = prev
        st.collect(run)
        st.reduce_buckets(b0, b1)

    def train_step(self, in_dict):
        self.model.train()
        self._decay_lr()
        with self._on_stream():
            out, losses = self._core(in_dict)
        self.step_count += 1
        if self.step_count % self.ONEPASS_CHECK_EVERY == 0:
            self.check_onepass()
        return out, losses

    # ------------------------------------------------------------------------------------------
    # hipGraph execution: the ~2.4k kernel launches of one step are captured once and replayed, which
    # removes the host-side launch cost (the step is launch-bound in eager mode).
    @staticmethod
    def _clone_batch(d):
        return {k: (v.clone() if torch.is_tensor(v) else Trainer._clone_batch(v) if isinstance(v, dict) else v) for k, v in d.items()}

    def capture(self, in_dict, warmup=2):
        """Capture the whole step -- forward, backward, the bucketed gradient all-reduces (N > 1) and Adam -- for batches
        shaped like `in_dict`.  The host-side switches of a batch are frozen into the graph: `pretrain_mode` / `vis_on`
        must not change between replays (checked by load_batch), and DensePose point supervision is captured as ACTIVE
        whenever the batch carries a `dp_dict` (its losses are masked by `has_dp` per sample, so batches without
        DensePose labels simply contribute zeros).  If the communication library cannot be captured the all-reduces
        and the optimizer run right after each replay instead."""
        from . import conv
        if self.device.type != 'cuda':
            raise RuntimeError('hipGraph capture needs a GPU')
        self.model.train()
        self._static = self._clone_batch(in_dict)
        if isinstance(self._static.get('dp_dict'), dict):
            self._static['dp_dict']['dp_active'] = True
        with self._on_stream():
            for _ in range(warmup):
                self._core(self._static)
        torch.cuda.synchronize(self.device)
        if self.store is not None:
            self.store.quiesce()                  # no collective of the eager steps may still be on the watchdog's list (see there)
        from . import hrnet
        if self.store is not None:
            self.store.prepare_capture(attempts=2 if self.distributed else 1)
        modes = (True,)
        if self.distributed:
            # collectives inside the graph need a backend whose calls are stream work (RCCL); any other backend (gloo: host-side
            # reductions on its own threads) cannot be captured -- and a capture that FAILS half-way is not something to rely on
            # recovering from: round 5 tried it with gloo, the step's stream stayed in capture mode after the failed capture and the
            # second capture_begin raised.  So the choice is made up front; the retry below remains for an RCCL capture that raises.
            backend = torch.distributed.get_backend(self.store.group) if torch.distributed.is_initialized() else ''
            modes = (True, False) if 'nccl' in str(backend) else (False,)
        for in_graph in modes:
            conv._PACK_CACHE.clear()              # weight packing must be part of the captured work
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            hrnet.BRANCH_STREAMS = bool(int(os.environ.get('DANET_BRANCH_STREAMS', '1')))
            conv.FUSION.clear()
            try:
                # thread_local: the communication library's watchdog thread may poll events of earlier collectives while
                # this thread captures (the default, global mode turns that into a capture error)
                with torch.cuda.graph(graph, stream=self.stream, capture_error_mode='thread_local' if self.distributed else 'global'):
                    self._static_out = self._core(self._static, reduce=in_graph, with_optimizer=in_graph)
                self._reduce_in_graph = in_graph
                self.fusion_counts = dict(conv.FUSION)      # which attribute-carried fusions the captured step contains
                # the collectives the capture recorded, in issue order, and how many of them from inside the backward pass
                self.captured_collectives = (list(self.store.issued), self.store.issued_early) if (self.store is not None and in_graph) else ([], 0)
                break
            except RuntimeError:
                if not (in_graph and self.distributed):      # (only a capture that holds collectives has a fallback: reduce after the replay)
                    raise
                self._end_stray_capture()
                torch.cuda.synchronize(self.device)
                # the aborted capture left host-side state behind: queued weight-gradient jobs that point at tensors of its
                # pool, BatchNorm call counters, the arena cursor, pending collectives
                self.store._works = []
                conv._WQ.clear()
                conv._WQG.clear()
                self._abort_cleanup()
            finally:
                hrnet.BRANCH_STREAMS = False
        self._graph = graph
        return self

    @staticmethod
    def _loaded_hip_runtime():
        """The libamdhip64 THIS process already runs on (the copy torch loaded), by the path the loader mapped it from: dlopen of
        that path returns the same handle, where a bare soname could resolve to another installation's runtime and act on nothing."""
        import ctypes
        try:
            for ln in open('/proc/self/maps'):
                path = ln.rsplit(None, 1)[-1]
                if 'libamdhip64.so' in os.path.basename(path):
                    return ctypes.CDLL(path)
        except OSError:
            pass
        return None

    def _end_stray_capture(self):
        """A capture that raised may leave the step's stream in capture mode (observed: hipStreamCaptureStatusActive after
        torch.cuda.graph's own clean-up): end it through the runtime so that the next capture can begin.  Best effort, and it says
        so when it could not (torch's own capture bookkeeping is not reset either way: a failed capture is not recoverable in
        general, DESIGN 6 lesson 13 -- this only makes the NEXT error name the real state)."""
        import ctypes
        import warnings
        try:
            if not self.stream.is_capturing():
                return
            hip = self._loaded_hip_runtime()
            if hip is None:
                warnings.warn('trainer: the step stream is still capturing after a failed capture and the loaded HIP runtime was not found; '
                              'the next capture will fail')
                return
            g = ctypes.c_void_p()
            rc = hip.hipStreamEndCapture(ctypes.c_void_p(self.stream.cuda_stream), ctypes.byref(g))
            if g.value:
                hip.hipGraphDestroy(g)
            if rc != 0 or self.stream.is_capturing():
                warnings.warn('trainer: hipStreamEndCapture on the stray capture returned %d; the stream %s capturing' %
                              (rc, 'is still' if self.stream.is_capturing() else 'stopped'))
        except Exception as e:          # noqa: BLE001 -- nothing more to try; the second capture will report the state it finds
            warnings.warn('trainer: could not end the stray capture (%r)' % (e,))

    def load_batch(self, in_dict):
        """Copy a new batch into the captured graph's static input tensors (nested dictionaries included)."""
        def rec(dst, src, path):
            for k, v in src.items():
                if torch.is_tensor(v):
                    dst[k].copy_(v, non_blocking=True)
                elif isinstance(v, dict):
                    rec(dst[k], v, path + k + '.')
                elif k != 'dp_active' and dst.get(k) != v:
                    raise ValueError('train_step_graphed: %s%s = %r differs from the captured %r; capture() again' % (path, k, v, dst.get(k)))
        rec(self._static, in_dict, '')

    def train_step_graphed(self, in_dict=None):
        if self._graph is None:
            raise RuntimeError('call capture(in_dict) first')
        if in_dict is not None:
            self.load_batch(in_dict)
        self._decay_lr()
        self._graph.replay()
        if not self._reduce_in_graph:
            with self._on_stream():
                self.store.reduce_all()
                self.optimizer.step()
        self.step_count += 1
        if self.step_count % self.ONEPASS_CHECK_EVERY == 0:
            self.check_onepass()
        return self._static_out
