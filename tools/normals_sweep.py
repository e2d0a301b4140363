"""Seed sweep of the NORMAL MAP against the real reference (tests/reference_cases.compare with compute_normals=True: the
reference's normals are autograd gradients of the SDF w.r.t. the query points, models/generator.py:599-623): N seeds per
geometry - new generator weights, latents, cameras, noise each -, fp32 texels and (chairs) the 16-bit storages, 64 + 64 and the
128 + 128 kernel.  The CPU reference (first image) is the pinned side; the reference's own GPU path differs from it on single
pixels by up to several 1e-2 on this map (one sample across one texel edge turns a unit vector), so maximum AND mean are kept.
JSON on stdout (profiles/r6/normals_sweep.json).  Test infrastructure.       python tools/normals_sweep.py [seeds=6]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device('cuda:0')
    out = {'seeds': n, 'note': 'max / mean |d normal map| over the seeds; budget of tests/test_reference_gpu.py: 3e-3 against the CPU reference'}
    cases = (('chairs', 4, 128, 64, 'fp32'), ('p3d', 4, 128, 64, 'fp32'), ('cub', 2, 128, 64, 'fp32'), ('chairs', 1, 256, 128, 'fp32'),
             ('chairs', 4, 128, 64, 'fp16'), ('chairs', 4, 128, 64, 'bf16'))
    for geometry, batch, res, samples, texels in cases:
        worst = {'vs_reference_cpu_max': 0.0, 'vs_reference_cpu_mean': 0.0, 'vs_reference_gpu_max': 0.0, 'vs_reference_gpu_mean': 0.0,
                 'reference_cpu_vs_gpu_gap_max': 0.0, 'rgb_vs_reference_cpu_max': 0.0}
        for seed in range(n):
            sc = rc.build_scene(geometry, batch, dev, seed=7000 + 13 * seed)
            if texels != 'fp32':
                sc = rc.with_texels(sc, rc.texel_code(texels))
            r = rc.compare(sc, res, samples, cpu_images=1, grad=True, compute_normals=True)
            for k, grp, key in (('vs_reference_cpu_max', 'vs_reference_cpu', 'normals'), ('vs_reference_cpu_mean', 'mean_abs_vs_reference_cpu', 'normals'),
                                ('vs_reference_gpu_max', 'vs_reference_gpu', 'normals'), ('vs_reference_gpu_mean', 'mean_abs_vs_reference_gpu', 'normals'),
                                ('reference_cpu_vs_gpu_gap_max', 'reference_cpu_vs_gpu_gap', 'normals'), ('rgb_vs_reference_cpu_max', 'vs_reference_cpu', 'rgb')):
                worst[k] = max(worst[k], r[grp][key])
        out['%s_b%d_%dpx_%d+%d_%s_texels' % (geometry, batch, res, samples, samples, texels)] = worst
        print(geometry, batch, res, samples, texels, worst, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
