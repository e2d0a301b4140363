"""Render time of the extra-map variants (coords / semantics / normals) against the plain launch, per workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from nerf_from_image_amd import ops
dev = torch.device('cuda:0')
for name, (n, rad, dt, kw) in {'cfg2_b8_fp32': (8, bench.RADIUS, ops.TEXEL_F32, {}), 'cfg5_b2_fp32': (2, bench.RADIUS, ops.TEXEL_F32, {'R': 256, 'S': 128}),
                               'cfg5_b2_fp16': (2, bench.RADIUS, ops.TEXEL_F16, {'R': 256, 'S': 128})}.items():
    base, out0 = bench.time_render(ops, dev, n, rad, dt, iters=30, **kw)
    print('%-14s plain              %.4f ms  %.1f M rays/s' % (name, base['ms']['median'], base['rays_per_s'] / 1e6))
    for label, mkw in (('coords', dict(want_coords=True)), ('semantics', dict(want_semantics=True)), ('normals', dict(want_normals=True)),
                       ('normals+semantics', dict(want_normals=True, want_semantics=True))):
        r, out = bench.time_render(ops, dev, n, rad, dt, iters=30, **mkw, **kw)
        print('%-14s %-18s %.4f ms  %.1f M rays/s  x%.2f  rgb identical %s' % (name, label, r['ms']['median'], r['rays_per_s'] / 1e6,
              r['rays_per_s'] / base['rays_per_s'], bool(torch.equal(out['rgb'], out0['rgb']))))
