"""A/B builds of the CURRENT tree's forward translation unit with extra compiler flags, timed on the render workloads.

    python tools/probes/ab_render.py build name=-DFLAG[,-DFLAG2] [name2=...]     # here: build/ab/libnfi_<name>.so
    python tools/probes/ab_render.py build-bwd name=-DFLAG ...                   # the same for the backward unit
    python tools/probes/ab_render.py run [names]                                 # GPU box: ms per launch, checksums
    python tools/probes/with_lib.py build/ab/libnfi_<name>.so tools/inversion_synthetic.py --hip-only    # any tool on a variant

(round-3 variants built from knobs that have left the tree: tools/probes/render_variants.py)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'build', 'ab')


def build(specs, unit=0):
    import __graft_entry__ as entry
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)
    units = [u for u, _ in entry.UNITS]
    bwd_obj = os.path.join(ROOT, 'build', units[1 - unit].replace('.hip', '.o'))      # the OTHER unit, as built for the product
    assert os.path.exists(bwd_obj), 'run python __graft_entry__.py first'
    src = os.path.join(entry.CSRC, units[unit])
    unit_flags = entry.UNITS[unit][1]

    def one(spec):
        name, _, flags = spec.partition('=')
        obj = os.path.join(OUT, '%s.o' % name)
        entry.compile_unit(src, unit_flags + [f for f in flags.split(',') if f], obj)
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-fPIC', '-shared', obj,
                               bwd_obj, '-o', os.path.join(OUT, 'libnfi_%s.so' % name)])
        os.remove(obj)
        return name
    with ThreadPoolExecutor(3) as pool:
        for name in pool.map(one, specs):
            print('built', name, flush=True)


def run(names, iters=60):
    import torch
    import bench
    from nerf_from_image_amd import _lib
    dev = torch.device('cuda:0')
    names = names or sorted(f[7:-3] for f in os.listdir(OUT) if f.startswith('libnfi_') and f.endswith('.so'))
    cases = {'chairs_b8': (8, bench.RADIUS, 0, {}), 'all_hit_b8': (8, 1.3, 0, {}), 'chairs_b1': (1, bench.RADIUS, 0, {}),
             'cfg5_b2': (2, bench.RADIUS, 0, {'R': 256, 'S': 128}), 'cfg5_b2_fp16': (2, bench.RADIUS, 2, {'R': 256, 'S': 128}),
             'chairs_b8_fp16': (8, bench.RADIUS, 2, {}),
             'all_hit_b8_fp16': (8, 1.3, 2, {}), 'chairs_b8_exact': (8, bench.RADIUS, 0, {'tuning': 8})}
    for rep in range(2):                                   # two rounds: clock / thermal drift shows as a difference between them
        for name in names:
            _lib._lib = None
            _lib.LIBRARY = os.path.join(OUT, 'libnfi_%s.so' % name)
            from nerf_from_image_amd import ops
            res = {}
            for case, (n_img, radius, tdt, kw) in cases.items():
                r, out = bench.time_render(ops, dev, n_img, radius, tdt, iters=iters, **kw)
                res[case] = (r['ms']['median'], float(out['rgb'].double().sum()))
            print('%-10s ' % name + '  '.join('%s %.4f' % (k, v[0]) for k, v in res.items()) +
                  '   checksum %.6f' % sum(v[1] for v in res.values()), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build-bwd':
        build(sys.argv[2:], unit=1)
    else:
        (build if sys.argv[1] == 'build' else run)(sys.argv[2:])
