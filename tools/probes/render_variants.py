"""Variant builds of the fused render kernels for A/B timing on the GPU box - the NEGATIVE RESULTS of round 3 (HISTORY.md
section 7).  The build-time knobs these variants switch (NFI_PLANEWISE, NFI_TILE_PAIR, NFI_SCALAR_RAY, NFI_LEAN_RAY,
NFI_SPLIT_MIX, NFI_MERGE_HIST, the tuning bits 5-8 of the work queues) were removed from the product sources in round 4;
they live in the tree of commit FROZEN below, which `build` extracts with `git show` (needs the repository's history,
i.e. this container, not the GPU box) and compiles with the current build recipe.  `run` then times the libraries it finds.

    python tools/probes/render_variants.py build                     # here: build/variants/libnfi_render_<name>.so
    python tools/probes/render_variants.py run                       # on the GPU box: ms per launch of every variant

    base        product build (3 planes of loads in flight, 2 workgroups per CU)
    pw_occ2     gather one plane at a time, 2 workgroups per CU
    pw_occ3     gather one plane at a time, 3 workgroups per CU (<= 168 VGPRs)
    occ3        all loads in flight, 3 workgroups per CU (the compiler has to fit 168 VGPRs)
    pw_scalar   pw_occ2 + the marched ray's origin / direction / near / far in scalar registers
    pw_scalar_single(_occ3)   + one field tile at a time through the decoder MLP (half the accumulators)
    lean_pw_occ3 / two_rounds_occ3 / pair3_occ3   stage-local lane ids + plane by plane / planes 0+1 then 2 / tile pairs in
                three rounds of two planes, 3 workgroups per CU          (NFI_VARIANT_SKIP=cfg5_b2 skips a case)
"""
import os
import subprocess
import sys

FROZEN = '4d6d769'             # last commit of round 3: csrc/ with the experiment knobs
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'build', 'variants')
VARIANTS = {'prev': None,      # a library built from an earlier tree, dropped into build/variants by hand
            'base': ['-DNFI_PLANEWISE=0'], 'pw_occ2': ['-DNFI_PLANEWISE=1'], 'pw_occ3': ['-DNFI_PLANEWISE=1', '-DNFI_RENDER_OCC=3'],
            'occ3': ['-DNFI_PLANEWISE=0', '-DNFI_RENDER_OCC=3'],
            'pw_scalar': ['-DNFI_PLANEWISE=1', '-DNFI_SCALAR_RAY=1'],
            'pw_scalar_single': ['-DNFI_PLANEWISE=1', '-DNFI_SCALAR_RAY=1', '-DNFI_TILE_PAIR=0'],
            'sched_max_ilp': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
            'sched_max_clause': ['-mllvm', '-amdgpu-sched-strategy=max-memory-clause'],
            'no_slp': ['-fno-slp-vectorize'], 'reg_occ2': ['-DNFI_REG_OCC=2'], 'split_mix': ['-DNFI_SPLIT_MIX=1'], 'merge_cmp': ['-DNFI_MERGE_HIST=0'],
            # end of round 3: stage-local lane ids (NFI_LEAN_RAY) + gathers that keep 64 texel registers in flight, three
            # workgroups per CU without a spill inside the ray loop (tools/vgpr_liveness.py)
            'lean': ['-DNFI_LEAN_RAY=1'],
            'lean_pw_occ3': ['-DNFI_PLANEWISE=1', '-DNFI_LEAN_RAY=1', '-DNFI_RENDER_OCC=3'],
            'two_rounds_occ3': ['-DNFI_PLANEWISE=2', '-DNFI_LEAN_RAY=1', '-DNFI_RENDER_OCC=3'],
            'pair3_occ3': ['-DNFI_PLANEWISE=3', '-DNFI_LEAN_RAY=1', '-DNFI_RENDER_OCC=3'],
            'pw_scalar_single_occ3': ['-DNFI_PLANEWISE=1', '-DNFI_SCALAR_RAY=1', '-DNFI_TILE_PAIR=0', '-DNFI_RENDER_OCC=3']}


def build(names):
    import __graft_entry__ as entry
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUT, exist_ok=True)
    # the frozen sources: csrc/ and the header as they were at FROZEN
    frozen = os.path.join(OUT, 'src_' + FROZEN)
    listing = subprocess.check_output(['git', 'ls-tree', '-r', '--name-only', FROZEN, 'nerf_from_image_amd/csrc', 'include'],
                                      cwd=ROOT, text=True).split()
    for rel in listing:
        dst = os.path.join(frozen, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, 'wb') as f:
            f.write(subprocess.check_output(['git', 'show', '%s:%s' % (FROZEN, rel)], cwd=ROOT))
    csrc = os.path.join(frozen, 'nerf_from_image_amd', 'csrc')
    src = os.path.join(csrc, 'nfi_kernels.hip')
    bwd_obj = os.path.join(OUT, 'nfi_backward_field_%s.o' % FROZEN)
    entry.compile_unit(os.path.join(csrc, 'nfi_backward_field.hip'), ['-fno-slp-vectorize'], bwd_obj)

    def one(name):
        obj = os.path.join(OUT, 'render_%s.o' % name)
        seen, n = entry.compile_unit(src, VARIANTS[name], obj)
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-fPIC', '-shared', obj,
                               bwd_obj, '-o', os.path.join(OUT, 'libnfi_render_%s.so' % name)])
        os.remove(obj)
        return name, n
    with ThreadPoolExecutor(4) as pool:
        for name, n in pool.map(one, names):
            print('built', name, '(%d packed-fp32 instructions rewritten)' % n, flush=True)


def run(names, iters=60):
    import torch
    import bench
    from nerf_from_image_amd import _lib
    dev = torch.device('cuda:0')
    rows = []
    for name in names:
        _lib._lib = None
        _lib.LIBRARY = os.path.join(OUT, 'libnfi_render_%s.so' % name)
        from nerf_from_image_amd import ops
        res = {}
        ref = None
        for case, (n_img, radius, kw) in {'chairs_b8': (8, bench.RADIUS, {}), 'all_hit_b8': (8, 1.3, {}), 'chairs_b1': (1, bench.RADIUS, {}),
                                          'cfg5_b2': (2, bench.RADIUS, {'R': 256, 'S': 128})}.items():
            if case in os.environ.get('NFI_VARIANT_SKIP', '').split(','):
                continue
            r, out = bench.time_render(ops, dev, n_img, radius, ops.TEXEL_F32, iters=iters, **kw)
            res[case] = r['ms']['median']
            res[case + '_sum'] = float(out['rgb'].double().sum())
        rows.append((name, res))
        print('%-8s ' % name + '  '.join('%s %.4f ms' % (k, v) for k, v in res.items() if not k.endswith('_sum')) +
              '   checksums ' + ' '.join('%.6f' % v for k, v in res.items() if k.endswith('_sum')), flush=True)


if __name__ == '__main__':
    names = sys.argv[2:] or list(VARIANTS)
    if sys.argv[1] == 'build':
        build(names)
    else:
        run(names)
