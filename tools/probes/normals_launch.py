"""The fused render WITH the normal map, cfg2 B = 8 (fp32 texels), a few dozen launches - the command tools/pmc_collect.py wraps
for the counters of the normal-map kernel (profiles/r6/pmc_render_fwd_normals.json):

  python tools/pmc_collect.py --kernel render_fwd_kernel --units 131072 --out gpurun_out/pmc_render_fwd_normals.json -- \
         python tools/probes/normals_launch.py [plain]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nerf_from_image_amd import ops

dev = torch.device('cuda:0')
kw = {} if (len(sys.argv) > 1 and sys.argv[1] == 'plain') else {'want_normals': True}
r, _ = bench.time_render(ops, dev, 8, bench.RADIUS, ops.TEXEL_F32, iters=30, **kw)
print(r['ms'], r['rays_per_s'])
