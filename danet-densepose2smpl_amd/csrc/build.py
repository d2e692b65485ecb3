"""Builds libdanet_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Each .hip translation unit is compiled separately (so per-file flags apply: the rasteriser
needs -ffp-contract=off for bit-exact part ids) and linked into one shared library that sits
next to the sources, in-tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, 'libdanet_hip.so')
ARCH = 'gfx950'

MFMA_VGPR = ['-mllvm', '-amdgpu-mfma-vgpr-form']
UNITS = {
    'capi.hip': [],
    'smpl_lbs.hip': [],
    'iuv_raster.hip': ['-ffp-contract=off'],
    'geometry.hip': [],
    # MFMA_VGPR: keep MFMA accumulators / operands in VGPRs; the default AGPR form made the compiler shuttle the
    # register ring between the two files (196 v_accvgpr_* instructions in conv_fast_kernel<4,3>, ~1.4 per MFMA)
    'conv_igemm.hip': MFMA_VGPR,
    'conv_wgrad.hip': MFMA_VGPR,
    'conv_wgrad3x3.hip': MFMA_VGPR,
    'conv_wgrad_rows.hip': MFMA_VGPR,
    'conv3x3.hip': MFMA_VGPR,
    'conv3x3s.hip': MFMA_VGPR,
    'conv_fast.hip': MFMA_VGPR,
    'conv_g3.hip': MFMA_VGPR,
    'conv_pw.hip': MFMA_VGPR,
    'conv_pw_wgrad.hip': MFMA_VGPR,
    'conv3x3a.hip': [],
    'conv_stem_dgrad.hip': [],
    'conv_stem.hip': [],            # (accumulators in AGPRs: 8 x 4 tiles per wave, one workgroup per compute unit)
    'conv_f32.hip': ['-ffp-contract=off'],
    'conv_f32m.hip': MFMA_VGPR,
    'part_ops.hip': [],
    'iuv_ops.hip': [],
    'loss_ops.hip': [],
    'adam.hip': [],
    'norm_act.hip': [],
    'norm_act_f32.hip': [],
    'stn.hip': [],
    'pool.hip': [],
    'glue.hip': [],
    'gcn_tail.hip': [],
}
INCLUDES = {'norm_act_f32.hip': ['norm_act.hip']}
COMMON = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-I' + os.path.join(ROOT, 'include'), '-I' + HERE,
          '-Wall', '-Wno-unused-function'] + os.environ.get('DANET_EXTRA_CFLAGS', '').split()


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')] + \
              [os.path.join(ROOT, 'include', 'danet_hip.h'), os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src, extra in UNITS.items():
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, 'build', src.replace('.hip', '.o'))
        objs.append(o)
        incl = [os.path.join(HERE, i) for i in INCLUDES.get(src, [])]       # .hip files a unit #includes
        if force or _stale(o, [s] + incl + headers):
            jobs.append([_hipcc()] + COMMON + extra + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s\n%s' % (' '.join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
