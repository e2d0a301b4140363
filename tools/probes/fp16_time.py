"""Render time with fp32 / fp16 / bf16 texel storage (NFI_PROBE_LIBRARY selects a variant build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from nerf_from_image_amd import _lib
if os.environ.get('NFI_PROBE_LIBRARY'):
    _lib.LIBRARY = os.environ['NFI_PROBE_LIBRARY']
from nerf_from_image_amd import ops
dev = torch.device('cuda:0')
for name, (n, rad, dt, kw) in {'cfg2_b8_fp32': (8, bench.RADIUS, ops.TEXEL_F32, {}), 'cfg2_b8_fp16': (8, bench.RADIUS, ops.TEXEL_F16, {}),
                               'cfg2_b8_bf16': (8, bench.RADIUS, ops.TEXEL_BF16, {}), 'allhit_b8_fp32': (8, 1.3, ops.TEXEL_F32, {}), 'allhit_b8_fp16': (8, 1.3, ops.TEXEL_F16, {}),
                               'allhit_b8_bf16': (8, 1.3, ops.TEXEL_BF16, {}),
                               'cfg5_b2_fp32': (2, bench.RADIUS, ops.TEXEL_F32, {'R': 256, 'S': 128}), 'cfg5_b2_fp16': (2, bench.RADIUS, ops.TEXEL_F16, {'R': 256, 'S': 128})}.items():
    r, out = bench.time_render(ops, dev, n, rad, dt, iters=50, **kw)
    print('%-16s %.4f ms  %.1f M rays/s  sum %.6f' % (name, r['ms']['median'], r['rays_per_s'] / 1e6, float(out['rgb'].double().sum())))
