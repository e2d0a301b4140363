"""cfg5 geometry (256x256 rays, 128 + 128 samples) with fp16 texels: the wide render kernel at three workgroups per CU
(the default since round 4: 168 registers) against two (tuning bit 5: 256 registers, no scratch).  GPU box:
    python tools/probes/wide_occ.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nerf_from_image_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
for n_img, radius, label in ((2, bench.RADIUS, 'chairs-like'), (2, 1.3, 'every ray hits')):
    res = {}
    for name, tuning in (('three workgroups / CU', 0), ('two workgroups / CU', 32)):
        r, out = bench.time_render(ops, dev, n_img, radius, ops.TEXEL_F16, iters=40, R=256, S=128, tuning=tuning)
        res[name] = (r, out)
        print('cfg5 fp16 %-14s %-22s %.4f ms  %.1f M rays/s' % (label, name, r['ms']['median'], r['rays_per_s'] / 1e6), flush=True)
    a, b = res['three workgroups / CU'][1], res['two workgroups / CU'][1]
    print('   images bit-identical:', all(torch.equal(a[k], b[k]) for k in ('rgb', 'depth', 'mask')))
r32, _ = bench.time_render(ops, dev, 2, bench.RADIUS, ops.TEXEL_F32, iters=40, R=256, S=128)
print('cfg5 fp32 chairs-like: %.4f ms  %.1f M rays/s' % (r32['ms']['median'], r32['rays_per_s'] / 1e6))
