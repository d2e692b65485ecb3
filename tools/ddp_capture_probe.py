"""Root-cause probe for the sporadic SIGABRT of collectives inside a hipGraph capture (watchdog: "operation not permitted on an
event last recorded in a capturing stream").  No model: a 1-rank RCCL group, eager all-reduces, then a capture that holds
all-reduces, under variations of WHEN things happen.  Each variant runs REPS times in a fresh process; prints abort counts.
    python tools/ddp_capture_probe.py            (driver)      python tools/ddp_capture_probe.py <variant>   (one run)"""
import os
import subprocess
import sys
import time

VARIANTS = ['eager_then_capture_now', 'eager_sleep_then_capture', 'eager_then_slow_capture', 'no_eager_capture',
            'eager_after_capture', 'eager_after_capture_sleep', 'capture_twice']
REPS = int(os.environ.get('REPS', '4'))


def one(variant):
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    bufs = [torch.ones(1 << 20, device=dev) for _ in range(8)]
    st = torch.cuda.Stream()

    def step(slow=0.0):
        works = []
        x = torch.zeros(1 << 20, device=dev)
        for i, b in enumerate(bufs):
            x = x + 1
            if slow and i == 0:
                time.sleep(slow)
            works.append(dist.all_reduce(b, async_op=True))
        for w in works:
            w.wait()
        return x

    def capture(slow=0.0):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):
            step(slow)
        return g

    with torch.cuda.stream(st):
        if variant != 'no_eager_capture':
            for _ in range(3):
                step()
        torch.cuda.synchronize()
        if variant == 'eager_sleep_then_capture':
            time.sleep(1.0)
        g = capture(slow=1.5 if variant == 'eager_then_slow_capture' else 0.0)
        g.replay()
        torch.cuda.synchronize()
        if variant in ('eager_after_capture', 'eager_after_capture_sleep'):
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            if variant == 'eager_after_capture_sleep':
                time.sleep(1.0)
            g2 = capture()
            g2.replay()
        if variant == 'capture_twice':
            g2 = capture(slow=0.5)
            g2.replay()
        torch.cuda.synchronize()
    time.sleep(0.5)
    dist.destroy_process_group()
    print('OK', variant)


def main():
    if len(sys.argv) > 1:
        return one(sys.argv[1])
    for v in VARIANTS:
        bad = 0
        msg = ''
        for r in range(REPS):
            p = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                bad += 1
                lines = [ln for ln in p.stderr.splitlines() if 'watchdog thread terminated' in ln or 'Error' in ln]
                msg = lines[0][:200] if lines else p.stderr[-200:]
        print('%-28s aborted %d / %d  %s' % (v, bad, REPS, msg), flush=True)


if __name__ == '__main__':
    main()
