"""Seed sweep of the real-reference parity (tests/reference_cases.compare): N seeds per geometry - new generator weights, latents,
cameras and noise each - HIP drop-in against run.py::render + the real Generator on this GPU, and against the reference on
the CPU for the first image (all images of the geometries named in a third argument, e.g. chairs,density: where the GPU
reference itself flips single pixels - its elementwise kernels contract a*b+c, an ulp on a query point moves a sample across
a texel edge - the CPU reference over the WHOLE batch says which side moved).  JSON on stdout (profiles/r6/parity_sweep.json): per geometry the maximum over the seeds of every
error figure and the number of values over the 1e-4 budget.  Test infrastructure.   python tools/parity_sweep.py [seeds=10] [only=geometry,...] [cpu_all=geometry,...]"""
import json
import os
import sys
import tempfile

os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='nfi_miopen_db_'))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import reference_cases as rc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = set(sys.argv[2].split(',')) if len(sys.argv) > 2 and sys.argv[2] else None
    cpu_all = set(sys.argv[3].split(',')) if len(sys.argv) > 3 else set()
    dev = torch.device('cuda:0')
    out = {'seeds': n}
    for geometry, batch, res, samples in (('chairs', 8, 128, 64), ('p3d', 8, 128, 64), ('cub', 4, 128, 64), ('density', 4, 128, 64),
                                          ('carla', 2, 64, 32)):
        if only and geometry not in only:
            continue
        worst = {}
        over = 0
        masks = []
        for seed in range(n):
            sc = rc.build_scene(geometry, batch, dev, seed=5000 + 17 * seed)
            r = rc.compare(sc, res, samples, cpu_images=batch if geometry in cpu_all else 1)
            masks.append(r['mask_mean'])
            over += sum(r['pixels_over_1e-4_vs_reference_gpu'].values())
            for grp in ('vs_reference_gpu', 'vs_reference_cpu', 'reference_cpu_vs_gpu_gap'):
                for k, v in r[grp].items():
                    worst.setdefault(grp, {})[k] = max(worst.get(grp, {}).get(k, 0.0), v)
            del sc
            torch.cuda.empty_cache()
        out['%s_b%d_%dpx_%d+%d%s' % (geometry, batch, res, samples, samples, '_cpu_reference_on_all_images' if geometry in cpu_all else '')] = dict(
            worst, **{'values_over_1e-4_vs_reference_gpu_all_seeds': over, 'mask_mean_min': min(masks), 'mask_mean_max': max(masks)})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
