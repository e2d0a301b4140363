"""Variant builds of the backward translation unit (build-time knobs of nfi_backward_field.inc) for A/B timing on the
GPU box with the inputs of a real training step (tools/bench_train_backward.py).

    python tools/probes/scatter_variants.py build [names]      # here: build/variants/libnfi_bwd_<name>.so
    python tools/probes/scatter_variants.py run [names]        # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = {
    'base': [],
    # round 5: moments instead of corner sums in the reduce's walk, one upstream-gradient peek per chunk, the gather of the
    # field backward at raised issue priority
    'moments': ['-DNFI_BIN_MOMENTS'], 'peek': ['-DNFI_BWD_CHUNK_PEEK'], 'prio': ['-DNFI_BWD_GATHER_PRIO'],
    'depth8': ['-DNFI_BIN_DEPTH=8'], 'depth32': ['-DNFI_BIN_DEPTH=32'], 'reload16': ['-DNFI_BIN_RELOAD'],
    'reload32': ['-DNFI_BIN_RELOAD', '-DNFI_BIN_DEPTH=32'], 'reload64': ['-DNFI_BIN_RELOAD', '-DNFI_BIN_DEPTH=64'],
    'q16': ['-DNFI_ROWS_Q16_DEFAULT'], 'q16_g2': ['-DNFI_ROWS_Q16_DEFAULT', '-DNFI_BIN_ROW_BYTES=64'],
    'q16_all': ['-DNFI_ROWS_Q16_DEFAULT', '-DNFI_BIN_ROW_BYTES=64', '-DNFI_BIN_MOMENTS', '-DNFI_BWD_CHUNK_PEEK'],
    'r5all': ['-DNFI_BIN_MOMENTS', '-DNFI_BWD_CHUNK_PEEK', '-DNFI_BWD_GATHER_PRIO'],
    # round 4: point groups of the binned scatter (rows of a group <= N MB: 268 MB per scene in the training step -> 1 / 2 /
    # 4 / 8 / 16 groups), 16- or 8-texel tiles for the grouped buckets.  The two macros existed in nfi_backward_field.inc
    # for this measurement only (profiles/r4/scatter_point_groups.log); the product has the winner (80 MB, 16) hard-wired.
    'group_off': ['-DNFI_BIN_GROUP_MB=100000'], 'group_160': ['-DNFI_BIN_GROUP_MB=160'], 'group_80': ['-DNFI_BIN_GROUP_MB=80'],
    'group_40': ['-DNFI_BIN_GROUP_MB=40'], 'group_20': ['-DNFI_BIN_GROUP_MB=20'],
    'group_80_tile8': ['-DNFI_BIN_GROUP_MB=80', '-DNFI_BIN_GROUP_TILE=8'], 'group_40_tile8': ['-DNFI_BIN_GROUP_MB=40', '-DNFI_BIN_GROUP_TILE=8'],
}


def build(names):
    import __graft_entry__ as entry
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)
    fwd_obj = os.path.join(ROOT, 'build', 'nfi_kernels.o')
    assert os.path.exists(fwd_obj), 'run python __graft_entry__.py first'
    src = os.path.join(entry.CSRC, 'nfi_backward_field.hip')
    base = dict(entry.UNITS)['nfi_backward_field.hip']

    def one(name):
        obj = os.path.join(OUT, 'bwd_%s.o' % name)
        seen, n = entry.compile_unit(src, base + VARIANTS[name], obj)
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-fPIC', '-shared',
                               fwd_obj, obj, '-o', os.path.join(OUT, 'libnfi_bwd_%s.so' % name)])
        os.remove(obj)
        return name, n
    with ThreadPoolExecutor(4) as pool:
        for name, n in pool.map(one, names):
            print('built', name, '(%d packed-fp32 instructions rewritten)' % n, flush=True)


def run(names, n='20'):
    names = names or sorted(f[11:-3] for f in os.listdir(OUT) if f.startswith('libnfi_bwd_') and f.endswith('.so'))
    for name in names * 2:                                   # two alternating rounds
        env = dict(os.environ, NFI_PROBE_LIBRARY=os.path.join(OUT, 'libnfi_bwd_%s.so' % name))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_train_backward.py'), n], env=env,
                           capture_output=True, text=True)
        print(name, r.stdout.strip() or r.stderr[-1500:], flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:] or list(VARIANTS))
    else:
        run(sys.argv[2:])            # no names: every library found under build/variants
