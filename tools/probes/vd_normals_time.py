"""Render time of the view-direction decoder's launch with and without the normal map: the golden --use_viewdir case's field
(tests/golden/persp_viewdir_fine_rand: planes, 33-row decoder, output layer) under 8 cameras x 128 x 128 rays x (64 + 64)
samples and random per-ray features."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from conftest import load_golden
from parity_util import hip_field_setup
from stand_in import look_at_cameras
from nerf_from_image_amd import ops

dev = torch.device('cuda:0')
meta, t = load_golden('persp_viewdir_fine_rand')
texels, image = hip_field_setup(meta, t, dev)
B, R, S = 8, 128, 64
g = torch.Generator().manual_seed(3)
texels = texels[:1].expand(B, *texels.shape[1:]).contiguous() if texels.shape[0] < B else texels[:B]
cam = look_at_cameras(B, 1.6, g).to(dev)
focal = torch.full((B,), 1.0254, device=dev)
att = t['attention_values'][:1].expand(B, -1, -1).contiguous().to(dev)
xr = ops.pad_ray_features(torch.randn(B, R * R, 32, generator=g).to(dev))
nc, nf = torch.rand(B, R, R, S, generator=g).to(dev), torch.rand(B * R * R, S, generator=g).to(dev)


def run(**kw):
    return ops.render_fwd(cam, focal, R, R, S, texels, image, meta['scene_range'], meta['A'], attention_values=att, use_sdf=True,
                          beta=t['beta'].to(dev), alpha=t['alpha'].to(dev), noise_coarse=nc, noise_fine=nf, fine_sampling=True,
                          white_background=True, skip_missed_rays=True, ray_features=xr, **kw)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        ev[i].record(); fn()
    ev[n].record(); torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]


base = run()
plain = timed(lambda: run())
nrm = timed(lambda: run(want_normals=True))
out = run(want_normals=True)
print('view-direction decoder, 8 x 128^2 x (64+64): plain %.4f ms, + normals %.4f ms (x%.2f), rgb identical %s, mask mean %.3f' % (
    plain, nrm, plain / nrm, bool(torch.equal(out['rgb'], base['rgb'])), float(base['mask'].mean())))
