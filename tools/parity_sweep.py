"""Seed sweep of the real-reference parity (tests/reference_cases.compare): N seeds per geometry - new generator weights, latents,
cameras and noise each - HIP drop-in against run.py::render + the real Generator on this GPU, and against the reference on
the CPU for the first image.  JSON on stdout (profiles/r6/parity_sweep.json): per geometry the maximum over the seeds of every
error figure and the number of values over the 1e-4 budget.  Test infrastructure.   python tools/parity_sweep.py [seeds=10]"""
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
    dev = torch.device('cuda:0')
    out = {'seeds': n}
    for geometry, batch, res, samples in (('chairs', 8, 128, 64), ('p3d', 8, 128, 64), ('cub', 4, 128, 64), ('density', 4, 128, 64),
                                          ('carla', 2, 64, 32)):
        worst = {}
        over = 0
        masks = []
        for seed in range(n):
            sc = rc.build_scene(geometry, batch, dev, seed=5000 + 17 * seed)
            r = rc.compare(sc, res, samples, cpu_images=1)
            masks.append(r['mask_mean'])
            over += sum(r['pixels_over_1e-4_vs_reference_gpu'].values())
            for grp in ('vs_reference_gpu', 'vs_reference_cpu', 'reference_cpu_vs_gpu_gap'):
                for k, v in r[grp].items():
                    worst.setdefault(grp, {})[k] = max(worst.get(grp, {}).get(k, 0.0), v)
            del sc
            torch.cuda.empty_cache()
        out['%s_b%d_%dpx_%d+%d' % (geometry, batch, res, samples, samples)] = dict(
            worst, **{'values_over_1e-4_vs_reference_gpu_all_seeds': over, 'mask_mean_min': min(masks), 'mask_mean_max': max(masks)})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
