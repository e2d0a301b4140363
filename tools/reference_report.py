"""Measured parity of the HIP drop-in against the REAL reference (oracle/_ref: run.py::render + models/generator.py) on
this GPU and on the CPU: the numbers behind tests/test_reference_gpu.py, as JSON on stdout (profiles/r5/reference_parity.json).
Test infrastructure; run on the GPU box:  python tools/reference_report.py > gpurun_out/<tag>/reference_parity.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import reference_cases as rc  # noqa: E402
from oracle import reference  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rep = {'reference_root': reference.root(), 'device': torch.cuda.get_device_name(0), 'forward': {}, 'gradients': {}}
    for geometry, batch in (('chairs', 1), ('chairs', 8), ('p3d', 16), ('cub', 4)):
        t0 = time.time()
        sc = rc.build_scene(geometry, batch, dev)
        r = rc.compare(sc, 128, 64, cpu_images=2)
        r['seconds'] = time.time() - t0
        rep['forward']['%s_b%d_128px_64+64' % (geometry, batch)] = r
        del sc
        torch.cuda.empty_cache()
    sc = rc.build_scene('p3d', 4, dev)
    rep['forward']['p3d_b4_semantics'] = rc.compare(sc, 128, 64, cpu_images=1, compute_semantics=True)
    rep['forward']['p3d_b4_coords'] = rc.compare(sc, 128, 64, cpu_images=1, compute_coords=True)
    sc = rc.build_scene('carla', 2, dev)
    rep['forward']['carla_viewdir_b2_64px_32+32'] = rc.compare(sc, 64, 32, cpu_images=1)
    rep['forward']['carla_viewdir_b2_semantics'] = rc.compare(sc, 64, 32, cpu_images=1, compute_semantics=True)
    rep['forward']['carla_viewdir_b2_coords'] = rc.compare(sc, 64, 32, cpu_images=1, compute_coords=True)
    sc = rc.build_scene('chairs', 2, dev)
    rep['forward']['chairs_b2_normals_64px_32+32'] = rc.compare(sc, 64, 32, cpu_images=2, grad=True, compute_normals=True)
    for geometry in ('chairs', 'p3d', 'cub'):
        sc = rc.build_scene(geometry, 2, dev)
        rep['gradients']['%s_b2_128px_64+64' % geometry] = rc.gradients(sc, 128, 64)
        del sc
        torch.cuda.empty_cache()
    rep['gradients']['carla_viewdir_b2_64px_32+32'] = rc.gradients(rc.build_scene('carla', 2, dev), 64, 32)
    rep['training_step_cub_b4_128px_64+64'] = rc.training_step(rc.build_scene('cub', 4, dev), 128, 64)
    rep['regularisers_cub_b2'] = rc.regularisers(rc.build_scene('cub', 2, dev))
    sc = rc.build_scene('p3d', 4, dev)
    rep['inversion_p3d_b4_128px_64+64_8_steps'] = rc.inversion(sc, 128, 64, steps=8)
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
