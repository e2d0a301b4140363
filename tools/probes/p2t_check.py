"""planes_to_texels against torch.permute over plane sizes (odd / tiny / non-multiples of the block) and storage types,
and its time at the bench size (8 scenes x 256^2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_from_image_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for R in (2, 3, 5, 16, 17, 48, 255, 256, 300):
    pl = torch.randn(2, 3, 32, R, R, generator=g).to(dev)
    ref = pl.permute(0, 1, 3, 4, 2).contiguous()
    for dt, td in ((ops.TEXEL_F32, torch.float32), (ops.TEXEL_BF16, torch.bfloat16), (ops.TEXEL_F16, torch.float16)):
        t = ops.planes_to_texels(pl, dt)
        assert torch.equal(t.view(ref.shape), ref.to(td)), (R, dt)
    back = ops.texels_to_planes(ops.planes_to_texels(pl))
    assert torch.equal(back, pl), R
print('planes_to_texels / texels_to_planes exact for all sizes and storage types')
pl = torch.randn(8, 3, 32, 256, 256, generator=g).to(dev)
for name, fn in (('planes_to_texels', lambda: ops.planes_to_texels(pl)), ('texels_to_planes', None)):
    if fn is None:
        t = ops.planes_to_texels(pl); fn = lambda: ops.texels_to_planes(t)
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    for i in range(50):
        ev[i].record(); fn()
    ev[50].record(); torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(50))[25]
    print('%s 8 x 3 x 256^2 x 32: %.1f us (incl. the output allocation), %.2f TB/s' % (name, ms * 1e3, 2 * pl.numel() * 4 / ms / 1e9))
