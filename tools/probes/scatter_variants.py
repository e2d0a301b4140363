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
    'split_mix': ['-DNFI_SPLIT_MIX=1'],
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
    for name in names:
        env = dict(os.environ, NFI_PROBE_LIBRARY=os.path.join(OUT, 'libnfi_bwd_%s.so' % name))
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_train_backward.py'), n], env=env,
                           capture_output=True, text=True)
        print(name, r.stdout.strip() or r.stderr[-1500:], flush=True)


if __name__ == '__main__':
    names = sys.argv[2:] or list(VARIANTS)
    (build if sys.argv[1] == 'build' else run)(names)
