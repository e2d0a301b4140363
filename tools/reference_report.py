"""Measured parity of the HIP drop-in against the REAL reference (oracle/_ref: run.py::render + models/generator.py) on
this GPU and on the CPU: the numbers behind tests/test_reference_gpu.py, as JSON on stdout (profiles/r6/reference_parity.json).
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
    rep = {'reference_root': reference.root(), 'device': torch.cuda.get_device_name(0), 'forward': {}, 'gradients': {},
           'configs_vs_fp32_reference': {}}
    only = set(sys.argv[1:])

    def section(key, name, fn):
        """One measurement; a failure is recorded under its name and does not cost the rest of the report."""
        if only and key not in only:
            return
        t0 = time.time()
        try:
            r = fn()
            r['seconds'] = time.time() - t0
        except Exception as e:       # noqa: BLE001
            import traceback
            r = {'error': repr(e), 'traceback': traceback.format_exc()[-1500:]}
        (rep[key] if name else rep)[name or key] = r
        torch.cuda.empty_cache()

    # the configurations BASELINE words with 16-bit storage / at cfg5's shape / with termination, vs the fp32 reference
    scenes = {}
    for name in rc.CONFIG_CASES:
        section('configs_vs_fp32_reference', name, lambda: rc.config_case(name, dev, cpu_images=1, scenes=scenes))
    scenes.clear()
    # gradients: HIP vs the fp32 reference, and both against the reference in float64; the producer's convolutions on
    # MIOpen's deterministic solvers (rc.deterministic_producer: without it d loss / d latents moves by up to 2e-3 from run to
    # run in both implementations, profiles/r6/gradient_spread.json)
    def det(fn):
        def run():
            with rc.deterministic_producer():
                return fn()
        return run
    for geometry in ('chairs', 'p3d', 'cub', 'density'):
        section('gradients', '%s_b2_128px_64+64' % geometry, det(lambda: rc.gradients(rc.build_scene(geometry, 2, dev), 128, 64)))
    section('gradients', 'carla_viewdir_b2_64px_32+32', det(lambda: rc.gradients(rc.build_scene('carla', 2, dev), 64, 32)))
    section('training_step_cub_b4_128px_64+64', None, det(lambda: rc.training_step(rc.build_scene('cub', 4, dev), 128, 64)))
    section('regularisers_cub_b2', None, det(lambda: rc.regularisers(rc.build_scene('cub', 2, dev))))
    for geometry, batch in (('chairs', 1), ('chairs', 8), ('p3d', 16), ('cub', 4), ('density', 4)):
        section('forward', '%s_b%d_128px_64+64' % (geometry, batch), lambda: rc.compare(rc.build_scene(geometry, batch, dev), 128, 64, cpu_images=2))
    if not only or 'forward' in only:
        sc = rc.build_scene('p3d', 4, dev)
        section('forward', 'p3d_b4_semantics', lambda: rc.compare(sc, 128, 64, cpu_images=1, compute_semantics=True))
        section('forward', 'p3d_b4_coords', lambda: rc.compare(sc, 128, 64, cpu_images=1, compute_coords=True))
        sc = rc.build_scene('carla', 2, dev)
        section('forward', 'carla_viewdir_b2_64px_32+32', lambda: rc.compare(sc, 64, 32, cpu_images=1))
        section('forward', 'carla_viewdir_b2_semantics', lambda: rc.compare(sc, 64, 32, cpu_images=1, compute_semantics=True))
        section('forward', 'carla_viewdir_b2_coords', lambda: rc.compare(sc, 64, 32, cpu_images=1, compute_coords=True))
        sc = rc.build_scene('chairs', 2, dev)
        section('forward', 'chairs_b2_normals_64px_32+32', lambda: rc.compare(sc, 64, 32, cpu_images=2, grad=True, compute_normals=True))
        del sc
    def inv():
        with rc.deterministic_producer():
            return rc.inversion(rc.build_scene('p3d', 4, dev), 128, 64, steps=30)
    section('inversion_p3d_b4_128px_64+64_30_steps', None, inv)
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
