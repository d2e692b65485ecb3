"""Per-step kernel time by family from a rocprofv3 kernel-stats CSV of `bench.py --no-fp32 --no-cpu-baseline` (bf16 steps only:
3 eager profiling steps + warm-up + timed replays; `steps` = how many steps the CSV covers, default from the call count of
conv3x3_stream_kernel<3> / 132).  usage: python tools/step_breakdown.py <kernel_stats.csv> [steps]"""
import csv
import sys


def family(n):
    if 'conv3x3_stream' in n:
        return 'conv3x3 stream (3x3 / s1 branch layers, fwd + dgrad)'
    if 'conv3x3_' in n or 'conv3x3a' in n:
        return 'conv3x3 tile / one / row-tile (fused BN-backward dgrads, 64-channel crop layers, fall-backs)'
    if 'wgrad' in n or 'unpack' in n:
        return 'weight gradients incl. reduce / unpack'
    if 'conv_pw_kernel' in n:
        return 'pointwise convolutions (1x1 / s1 on csrc/conv_pw.hip, fwd + dgrad)'
    if 'conv_stem' in n:
        return 'stem convolutions (7x7 / s2 forward + data gradient on LDS row tiles, csrc/conv_stem*.hip)'
    if 'conv_fast' in n or 'conv_igemm' in n:
        return 'gather convolutions (1x1, strided, 7x7, grouped, transposed)'
    if 'bn_' in n or 'sum_relu' in n or 'fuse_' in n or 'channel_sum' in n:
        return 'BatchNorm / ReLU / residual / fuse sums'
    if 'at::native' in n or 'at_cuda' in n or 'elementwise' in n or 'rocprim' in n or 'Cijk' in n or 'MIOpen' in n:
        return 'tensor-op glue (torch / hipBLASLt / MIOpen launches)'
    if 'copyBuffer' in n or 'fillBuffer' in n:
        return 'memcpy / memset nodes'
    if 'pack' in n:
        return 'weight packing'
    if 'adam' in n:
        return 'Adam'
    if 'smpl' in n or 'rodrigues' in n or 'rot6d' in n:
        return 'SMPL layer'
    if 'raster' in n:
        return 'IUV raster'
    return 'STN / IUV heads / losses / pool / other HIP'


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    if steps is None:
        c = sum(int(r['Calls']) for r in rows if 'conv3x3_stream_kernel<3>' in r['Name'] or 'conv3x3_stream_bn_kernel<3>' in r['Name'])     # (DANET_CONV_BN=1: the forward launches carry the second name)
        steps = c / 132.0 if c else 1.0
    fam = {}
    for r in rows:
        if 'Cijk' in r['Name'] and float(r['AverageNs']) > 3e5:
            continue                               # bench.py's queue filler (240 8192^3 matmuls in front of the bracketed eager step)
        f = family(r['Name'])
        c, t = fam.get(f, (0, 0.0))
        fam[f] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
    tot = sum(t for _, t in fam.values())
    print('steps covered: %.1f   kernel time per step: %.2f ms   launches per step: %.0f' % (steps, tot / steps / 1e6, sum(c for c, _ in fam.values()) / steps))
    for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print('%-68s %7.1f launches  %6.2f ms  %5.1f %%' % (f, c / steps, t / steps / 1e6, 100 * t / tot))


if __name__ == '__main__':
    main()
