"""Headline benchmark (BASELINE.json): images/sec of the full DaNet training step -- forward +
backward (+ gradient all-reduce for N > 1) + Adam -- HRNet-W48 + SMPL LBS + IUV render,
256x256 input, 32 images per GPU, synthetic data, random-init weights.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` is the whole-job aggregate (all ranks), timed over
exactly K steps between barrier + synchronize pairs, max over ranks.  `roofline` is for the
kernel instance with the largest total time (measured live with HIP events on the launch
stream during the timed steps); `cpu_baseline` times the CPU oracle (plain-torch fp32 HRNet
step + C LBS + C raster) on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0       # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
STEP_GFLOP_PER_IMAGE_256 = 180.4      # 30.07 GMAC/img forward = 60.1 GFLOP, x3 for forward + data + weight gradients (5.77 TFLOP per 32-image step)
PEAK_FP32_MFMA_TFLOPS = 157.0         # v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD (no reduced-precision path for fp32 inputs on gfx950)


def whole_step_mfma(batch, size, ms_per_step, peak_tflops):
    """The step's ALGORITHMIC convolution / linear FLOPs over its wall time against the dense MFMA peak: the number the
    north star's '>= 40 % conv MFMA roofline' is read against as a whole (the `roofline` entry is the dominant kernel alone)."""
    tflop = STEP_GFLOP_PER_IMAGE_256 * (size / 256.0) ** 2 * batch / 1e3
    ach = tflop / (ms_per_step * 1e-3)
    return {'alg_tflop_per_step': round(tflop, 3), 'achieved': round(ach, 1), 'peak': peak_tflops, 'unit': 'TFLOP/s', 'frac': round(ach / peak_tflops, 4),
            'note': 'all convolution families + every non-MFMA kernel of the step in the denominator'}


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source description) from this round's committed PMC summary
    (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on a 512 MiB copy
    in the same run), or (None, reason).  The file records the commit it was measured at."""
    for name in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            f = json.load(open(path))
            d = f['kernels'].get(kernel)
            if d and 'fetch_bytes_per_launch' in d and 'write_bytes_per_launch' in d:
                return int(d['fetch_bytes_per_launch'] + d['write_bytes_per_launch']), \
                    'profiles/%s (commit %s; bytes per launch averaged over this kernel\'s launches of one step)' % (name, f.get('commit', 'n/a'))
        except (OSError, ValueError, KeyError):
            pass
    return None, 'no PMC summary for this kernel under profiles/'


def rocprof_avg_us(kernel):
    """(average duration of `kernel` in us, source) from this round's committed `rocprofv3 --kernel-trace --stats` summary of the same
    command (profiles/r05_bench_kernel_stats.csv), to sit beside the live HIP-event figure; (None, reason) without one."""
    import csv
    for name in ('r06_bench_kernel_stats.csv', 'r05_bench_kernel_stats.csv', 'r04_bench_kernel_stats.csv', 'r03_bench_bf16_only_kernel_stats.csv'):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            for row in csv.DictReader(open(path)):
                if kernel and kernel in row['Name']:
                    return round(float(row['AverageNs']) / 1e3, 2), 'profiles/' + name
        except (OSError, KeyError, ValueError):
            pass
    return None, 'no rocprofv3 summary for this kernel under profiles/'


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU (BASELINE: 32)')
    ap.add_argument('--size', type=int, default=256, help='input resolution (BASELINE: 256)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32', action='store_true', help='skip the fp32 (BASELINE config C4) sub-record')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--dtype', choices=('bf16', 'fp32'), default='bf16',
                    help='fp32: BASELINE config C4\'s arithmetic on the verification kernels (csrc/conv_f32.hip, eager launches; a correctness configuration, not a performance one)')
    ap.add_argument('--grad-wire', choices=('fp32', 'bf16'), default=None, help='N > 1: all-reduce the gradient buckets in this type (default: DANET_GRAD_WIRE or fp32)')
    ap.add_argument('--force-ddp', action='store_true', help='diagnostic: run the N > 1 code path (GradStore buckets, segmented backward, in-graph all-reduces) on a 1-rank group')
    ap.add_argument('--dry', action='store_true',
                    help='launch-line check: build the trainer, capture, run ONE step and print the line (allreduce record included) without the '
                         'roofline / geometry / fp32 / CPU legs -- what a multi-GPU launch does first, cheap enough for a test')
    return ap.parse_args()


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(size, cpu_batch):
    """The oracle timed on the host cores on a bounded sample of the same workload: every convolutional net of the step
    (oracle/torch_ref.StepNets: HRNet-W48 + global and partial IUV heads + body / limb regressor nets) forward + backward
    + Adam in fp32, plus the C restatements of the SMPL layer (2 forward, 1 backward -- what the timed GPU step holds)
    and of the IUV raster.  The thread count is swept (all host threads, a half, a quarter: small batches do not scale to
    every core) and the best setting reported."""
    import numpy as np
    import oracle
    from oracle import torch_ref
    from danet_densepose2smpl_amd import assets
    nthr = torch.get_num_threads()
    counts = sorted({max(1, nthr), max(1, nthr // 2), max(1, nthr // 4)}, reverse=True)
    sweep, nparam, spread = torch_ref.train_step_cpu_sweep(cpu_batch, size, counts)
    best_thr, t_net = min(sweep, key=lambda r: r[1])
    model = assets.make_synthetic_smpl(0)
    vm, faces, tex = assets.densepose_render_tables(assets.make_synthetic_densepose(model, 0))
    rng = np.random.default_rng(0)
    betas = rng.normal(0, 1, (cpu_batch, 10)).astype(np.float32)
    pose = rng.normal(0, 0.2, (cpu_batch, 72)).astype(np.float32)
    t0 = time.time()
    for _ in range(2):
        verts, _ = oracle.lbs_forward(model, betas, pose, False, np.float32)
    rot = np.tile(np.eye(3, dtype=np.float32), (cpu_batch, 24, 1, 1))
    oracle.lbs_backward(model, betas, rot, verts, None, np.float32)
    cam = np.tile(np.array([[0.9, 0.0, 0.0]], np.float32), (cpu_batch, 1))
    oracle.raster_forward(verts, cam, vm, faces, tex, 5000.0, float(size), size // 4)
    t_geo = time.time() - t0
    return {'value': round(cpu_batch / (t_net + t_geo), 4), 'unit': 'images/sec', 'cores': best_thr,
            'kind': 'port', 'cpu': _cpu_model(), 'host_threads': nthr,
            'thread_sweep': [{'threads': n, 'images_per_sec': round(cpu_batch / (t + t_geo), 4)} for n, t in sweep],
            # three draws at the reported setting: `value` is the best of them, this is their range (images/sec)
            'spread': [round(cpu_batch / (spread[1] + t_geo), 4), round(cpu_batch / (spread[0] + t_geo), 4)],
            'sample': 'B=%d of the B=32 step, one timed step per thread count after one warm-up step, then two more at the fastest of %s threads (best of the three reported, range in `spread`): '
                      'oracle/torch_ref.StepNets fwd+bwd+Adam fp32 (%.2fs; %.1f M parameters; 30.06 of the step\'s 30.07 GMAC/img = 99.9 %% of its '
                      '5.77 TFLOP: only the GCN / 1x1 regressors and the loss glue are left out) + 2x C SMPL fwd, 1x C SMPL bwd, 1x C IUV '
                      'raster on one thread (%.2fs)' % (cpu_batch, counts, t_net, nparam / 1e6, t_geo)}


def geometry_rooflines(tr, B, size, dev):
    """HIP-event timings of the SMPL layer and the IUV raster at the bench batch, as achieved HBM GB/s against their
    algorithmic bytes (LBS 19.6 MB of constants once + 83.6 kB per item; backward reads the constants and
    dL/dverts again; raster ~5.0 MB at B = 32)."""
    smpl, rend = tr.model.iuv2smpl.smpl, tr.model.iuv_renderer
    g = torch.Generator(device='cpu').manual_seed(7)
    betas = torch.randn(B, 10, generator=g).to(dev).requires_grad_(True)
    rot = torch.eye(3).repeat(B, 24, 1, 1).to(dev).requires_grad_(True)
    cam = torch.tensor([[0.9, 0.0, 0.0]]).repeat(B, 1).to(dev)

    def fwd():
        return smpl(betas=betas, body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False, rotmats=rot)      # (as smpl_regressor.py calls it)

    def timed(fn, n=20):
        # replayed from a hipGraph: an event pair then brackets device time, not the host's launch latency
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # (the one-launch SMPL backward borrows the one-pass BatchNorm backward's barrier state, which belongs to the launches of ONE
        # stream: here the measuring stream, as the step's own stream inside the trainer)
        from danet_densepose2smpl_amd import nn as _dnn
        prev_stream, _dnn.ONEPASS_STREAM = _dnn.ONEPASS_STREAM, side
        try:
            with torch.cuda.stream(side):
                fn()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    fn()
                for _ in range(2):
                    gr.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    gr.replay()
                e1.record()
            torch.cuda.synchronize(dev)
        finally:
            _dnn.ONEPASS_STREAM = prev_stream
        return e0.elapsed_time(e1) * 1e-3 / n
    t_f = timed(fwd)

    def fb():
        out = fwd()
        (out.vertices.sum() + out.joints.sum()).backward()
        betas.grad = None
        rot.grad = None
    t_fb = timed(fb)
    verts = fwd().vertices.detach()
    t_r = timed(lambda: rend.verts2uvimg(verts, cam))
    const, per = 19.6e6, 83.6e3
    by_f, by_b, by_r = const + per * B, const + (82.7e3 + per) * B, 5.0e6 * B / 32.0
    peak = 8000.0
    mk = lambda name, by, t: {'kernel': name, 'bound': 'hbm', 'achieved': round(by / t / 1e9, 1), 'peak': peak, 'unit': 'GB/s',   # noqa: E731
                              'frac': round(by / t / 1e9 / peak, 4), 'us': round(t * 1e6, 1), 'alg_bytes': int(by)}
    return [mk('smpl layer forward (LBS as one launch + the joint selections as one)', by_f, t_f), mk('smpl layer backward (LBS as three launches + the selections\' gradients as one)', by_b, max(t_fb - t_f, 1e-9)),
            mk('iuv_raster forward (project+faces+resolve)', by_r, t_r)]


def fp32_record(args, tr, batch, world, dev):
    """BASELINE config C4 (the full train step in fp32, the reference's own arithmetic type): the same model and step with fp32
    NHWC activations on the fp32 MFMA kernels (csrc/conv_f32m.hip, norm_act_f32.hip), replayed from a hipGraph when the
    capture succeeds.  Returns the timing record (every rank must call it: the step holds the gradient all-reduces)."""
    from danet_densepose2smpl_amd import conv
    from danet_densepose2smpl_amd import nn as _dnn
    steps, warm = (1 if args.dry else max(10, min(args.steps, 20))), 2
    roof = None
    with conv.precision('fp32'):
        exec_mode = 'eager'
        if not args.dry:
            # the fp32 step's own roofline: the conv launches of one eager step bracketed by HIP events, against the fp32 MFMA peak
            tr.train_step(batch)
            summ, dominant = profile_step(tr, batch, dev)
            if dominant is not None:
                n, secs, flops = summ[dominant]
                ach = flops / secs / 1e12
                roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                        'traffic': None, 'kernel': dominant + ' (forward + data-gradient launches of the step, all shapes)', 'launches': n,
                        'avg_us': round(secs / n * 1e6, 2), 'alg_gflop_per_launch': round(flops / n / 1e9, 3)}
        step = lambda: tr.train_step(batch)
        if not args.no_graph:
            try:
                tr.capture(batch)
                step, exec_mode = tr.train_step_graphed, 'hipgraph'
            except Exception as e:
                sys.stderr.write('fp32: hipGraph capture failed (%r); running eagerly\n' % (e,))
        for _ in range(warm):
            out = step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.time()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        elapsed = time.time() - t0
    if _dnn.onepass_error(dev):
        raise RuntimeError('bench: a one-pass BatchNorm launch gave up at its grid barrier (nn.onepass_error)')
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = out[1] if isinstance(out, tuple) else out
    finite = bool(all(torch.isfinite(v).all() for v in losses.values())) if isinstance(losses, dict) else None
    return {'dtype': 'f32', 'value': round(world * args.batch * steps / elapsed, 2), 'unit': 'images/sec', 'ms_per_step': round(elapsed / steps * 1e3, 2),
            'steps': steps, 'warmup': warm, 'exec': exec_mode, 'finite_losses': finite, 'roofline': roof,
            'whole_step_mfma': whole_step_mfma(args.batch, args.size, elapsed / steps * 1e3, PEAK_FP32_MFMA_TFLOPS),
            'workload': 'BASELINE config C4 arithmetic: the same full train step with fp32 NHWC activations; convolutions on '
                        'v_mfma_f32_16x16x4_f32 (conv_f32m.hip), BatchNorm / fuse sums / STN on the fp32 instantiation of the HIP kernels'}


def fp32_line(args, tr, batch, world, rank, dev):
    rec = fp32_record(args, tr, batch, world, dev)
    if rank == 0:
        B = args.batch
        print(json.dumps({'metric': 'images/sec fwd+bwd HRNet-W48+SMPL+IUV 256x256 bs32/GPU', 'value': rec['value'],
                          'unit': 'images/sec', 'n_gpus': world, 'steps': rec['steps'], 'warmup': rec['warmup'],
                          'ms_per_step': rec['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'f32', 'data': 'synthetic', 'exec': rec['exec'],
                          'config': {'workload': rec['workload'], 'global_batch': B * world, 'parallelism': 'dp%d' % world},
                          'finite_losses': rec['finite_losses'], 'roofline': rec.get('roofline')}), flush=True)


def _barrier_word():
    """The grid-barrier error word(s): 0, or the code of the wait that expired (csrc/grid_barrier.h)."""
    from danet_densepose2smpl_amd import nn as _dnn
    return [int(b[2]) for b in _dnn._ONEPASS_BAR.values()]


def _dnn_error():
    from danet_densepose2smpl_amd import nn as _dnn
    return _dnn.onepass_error()


def profile_step(tr, batch, dev):
    """One eager step with every conv launch bracketed by HIP events: (summary, dominant kernel)."""
    from danet_densepose2smpl_amd import conv
    # The host needs ~3x longer to enqueue an eager step than the GPU needs to run it; ~0.4 s of queued matmuls in
    # front let the host run ahead, so that the bracketed kernels execute back to back and an event pair measures
    # the kernel, not the host's launch latency.
    fa = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    fb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    torch.mm(fa, fb)
    torch.cuda.synchronize(dev)
    for _ in range(240):
        torch.mm(fa, fb)
    conv.PROFILER = conv.KernelProfiler()
    tr.train_step(batch)
    torch.cuda.synchronize(dev)
    summ = conv.PROFILER.summary()
    conv.PROFILER = None
    del fa, fb
    dominant = max(((k, v) for k, v in summ.items() if v[2] > 0), key=lambda kv: kv[1][1])[0] if summ else None

    return summ, dominant


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or args.force_ddp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # the communicator must see the channel limit the one-pass BatchNorm barrier is sized against (trainer.reserve_comm_channels)
        from danet_densepose2smpl_amd.trainer import reserve_comm_channels
        reserve_comm_channels()
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)

    from danet_densepose2smpl_amd import conv
    from danet_densepose2smpl_amd.config import cfg_from_dict, reset_cfg
    from danet_densepose2smpl_amd.trainer import Trainer, synthetic_in_dict, default_options
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': args.size, 'DANET.HEATMAP_SIZE': args.size // 4})
    torch.manual_seed(1234)
    B = args.batch
    tr = Trainer(default_options(B), device=dev, distributed=world > 1 or args.force_ddp, grad_wire=args.grad_wire)
    batch = synthetic_in_dict(tr.model, B, dev, seed=1234 + rank)
    if args.dtype == 'fp32':
        fp32_line(args, tr, batch, world, rank, dev)
        if world > 1 or args.force_ddp:
            dist.destroy_process_group()
        return

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if args.dry:
        args.steps, args.warmup, args.no_fp32, args.no_cpu_baseline = 1, 0, True, True
    # One eager step with every conv launch bracketed by HIP events (on the launch stream) finds the
    # dominant kernel instance and gives its per-launch durations -- the same kernels, shapes and
    # data as the timed steps (which replay them from a hipGraph, where events cannot be recorded).
    def stage(tag):              # DANET_BENCH_STAGES=1: the grid-barrier error word after each stage (stderr)
        if os.environ.get('DANET_BENCH_STAGES'):
            torch.cuda.synchronize(dev)
            from danet_densepose2smpl_amd import nn as _n
            sys.stderr.write('stage %-16s t=%.3f onepass_error=%s word=%s\n' % (tag, time.time(), _dnn_error(), [hex(int(b[2])) for b in _n._ONEPASS_BAR.values()]))
    tr.train_step(batch)
    tr.train_step(batch)
    torch.cuda.synchronize(dev)
    stage('two eager steps')
    summ, dominant = {}, None
    if not args.dry:
        summ, dominant = profile_step(tr, batch, dev)
    stage('profile_step')
    use_graph = not args.no_graph
    if use_graph:
        try:
            tr.capture(batch)
            step = tr.train_step_graphed
        except Exception as e:                       # keep the bench alive: fall back to eager launches
            sys.stderr.write('hipGraph capture failed (%r); running eagerly\n' % (e,))
            use_graph = False
    if not use_graph:
        step = lambda: tr.train_step(batch)
    stage('capture')
    for _ in range(args.warmup):
        step()
    sync()
    stage('warmup')
    _flush_c_stdio()      # RCCL's start-up banner sits in the C stdio buffer of every rank: emit it now, not after the JSON line
    t0 = time.time()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.time() - t0
    elapsed_local = elapsed
    stage('timed steps')
    # what the timed steps computed must be numbers: the last step's losses and a sample of the parameters the optimizer has
    # updated W + K times by now (round 5 found replays of the round-4 graph turning NaN after a few steps: a memset node racing the
    # bias-gradient kernel -- the timing was unaffected, the training was not)
    last = tr._static_out[1] if use_graph else tr.train_step(batch)[1]
    finite = bool(all(torch.isfinite(v.float()).all() for v in last.values())) and \
        bool(all(torch.isfinite(p).all() for p in list(tr.model.parameters())[::7]))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage('before extras')
    roof = None
    if dominant is not None:
        if dominant in summ:
            n, secs, flops = summ[dominant]
            ach = flops / secs / 1e12
            traffic, tsrc = pmc_traffic(dominant)
            rp_us, rp_src = rocprof_avg_us(dominant)
            roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_BF16_DENSE_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_BF16_DENSE_TFLOPS, 4), 'traffic': traffic, 'kernel': dominant,
                    'launches': n, 'avg_us': round(secs / n * 1e6, 2), 'alg_gflop_per_launch': round(flops / n / 1e9, 3),
                    'traffic_source': tsrc,
                    # the same kernel's average under rocprofv3 --kernel-trace --stats (graph replays, committed summary) beside the
                    # live event brackets of one eager step: the brackets include the launch gaps of eager execution
                    'rocprof_avg_us': rp_us, 'rocprof_frac': None if rp_us is None else round(flops / n / (rp_us * 1e-6) / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4),
                    'rocprof_source': rp_src}
    extra = None
    if rank == 0 and not args.dry:
        try:
            extra = geometry_rooflines(tr, B, args.size, dev)
        except Exception as e:                                       # secondary figures must not take the bench line down
            extra = [{'error': repr(e)}]

    fp32 = None
    if not args.no_fp32 and world == 1:                             # (N > 1: one configuration per run; `--dtype fp32` measures C4 on N GPUs)
        try:                                                         # (after everything measured on the bf16 graph: this re-captures)
            fp32 = fp32_record(args, tr, batch, world, dev)
        except Exception as e:
            fp32 = {'error': repr(e)}
    allreduce_info = None
    if world > 1 or args.force_ddp:
        st = tr.store
        issued, early = (tr.captured_collectives if use_graph and tr._reduce_in_graph else (st.issued, st.issued_early))
        per_rank = [round(elapsed_local / args.steps * 1e3, 3)]
        if world > 1:
            gathered = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(gathered, torch.tensor([elapsed_local], device=dev, dtype=torch.float64))
            per_rank = [round(float(g.item()) / args.steps * 1e3, 3) for g in gathered]
        allreduce_info = {'mode': ('in-graph' if (use_graph and tr._reduce_in_graph) else 'after the graph replay' if use_graph else 'eager'),
                          'buckets': len(st.buckets), 'bucket_mb': round(max(e - s0 for s0, e, _, _ in st.buckets) * 4 / 2**20, 1),
                          'released_during_backward': int(early), 'wire_dtype': 'bf16' if st.wire is not None else 'f32',
                          'bytes_per_step': int(st.flat.numel() * (2 if st.wire is not None else 4)),
                          'ms_per_step_per_rank': per_rank,
                          # compute units left to the communication library while the backward pass runs (NCCL_MAX_NCHANNELS) and the
                          # workgroup budget of the one-pass BatchNorm backward's grid barrier that follows from it
                          'comm_channels_reserved': int(os.environ.get('NCCL_MAX_NCHANNELS', 0)), 'onepass_max_blocks': int(getattr(tr, 'onepass_blocks', 0)),
                          # the one-pass BatchNorm backward's barrier error word on this rank and its all-reduced sum (0 / 0.0: no launch timed out)
                          'onepass_error': bool(_dnn_error()), 'poison_sum': float(st.poison)}
    if rank == 0:
        ips = world * B * args.steps / elapsed
        line = {'metric': 'images/sec fwd+bwd HRNet-W48+SMPL+IUV 256x256 bs32/GPU', 'value': round(ips, 2), 'unit': 'images/sec',
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
                'exec': 'hipgraph' if use_graph else 'eager', 'finite_losses_and_parameters': finite, 'onepass_error': bool(_dnn_error()), 'barrier_error_word': _barrier_word(),
                'dry': bool(args.dry),
                'allreduce': allreduce_info,
                'config': {'workload': 'full DaNet train step (HRNet-W48 + global and part-wise IUV heads + regressor nets + SMPL LBS '
                                       '(2 forward + 1 backward; the 2 label-side forwards belong to the untimed batch prologue) + IUV '
                                       'render + losses, fwd+bwd+Adam), %dx%d, %d img/GPU; convs bf16 MFMA fp32-acc, '
                                       'LBS/raster/losses fp32' % (args.size, args.size, B),
                           'global_batch': B * world, 'parallelism': 'dp%d' % world},
                'roofline': roof, 'whole_step_mfma': whole_step_mfma(B, args.size, elapsed / args.steps * 1e3, PEAK_BF16_DENSE_TFLOPS),
                'roofline_extra': extra}
        if fp32 is not None:
            line['fp32'] = fp32
        if world == 1 and not args.no_cpu_baseline:
            try:
                line['cpu_baseline'] = cpu_baseline(args.size, args.cpu_batch)
            except Exception as e:                                   # the bench line must still be printed
                line['cpu_baseline'] = {'error': repr(e)}
        _flush_c_stdio()
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_ddp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
