"""Prints the work-queue words of a render launch's workspace (per queue: hand-out counter at the end of the launch,
length, static share) - a check of ray_lists_kernel / RayQueue.  python tools/probes/queue_words.py [radius]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nerf_from_image_amd import ops

dev = torch.device('cuda:0')
radius = float(sys.argv[1]) if len(sys.argv) > 1 else bench.RADIUS
d = bench.synthetic_inputs(8, 1234, dev)
if radius != bench.RADIUS:
    d = dict(d, cam=bench.cameras(8, radius, torch.Generator().manual_seed(77)).to(dev))
texels = ops.planes_to_texels(d['planes'])
image = ops.decoder_pack(d['w1'], d['b1'], d['w2'], d['b2'], bench.A)
R = bench.R
nc = torch.rand(8, R, R, 64, device=dev); nf = torch.rand(8 * R * R, 64, device=dev)
out = ops.render_fwd(d['cam'], d['focal'], R, R, 64, texels, image, bench.SCENE_RANGE, bench.A, d['att'], True, d['beta'], d['alpha'],
                     noise_coarse=nc, noise_fine=nf)
torch.cuda.synchronize()
w = out['_workspace'][:(16 + 128) * 4].view(torch.int32).cpu().tolist()
print('reduce', w[:4])
for q in range(8):
    print('queue %d: counter %6d  length %6d  static %6d' % (q, w[16 + 16 * q], w[16 + 16 * q + 1], w[16 + 16 * q + 2]))
print('marched total', sum(w[16 + 16 * q + 1] for q in range(8)))
