"""Timing of the regulariser operator (sdf + spatial gradient and its double backward) at the size Generator.forward
uses it: 4 scenes x 31^3 stratified points, 256^2 planes.   python tools/bench_regulariser.py [N]
(NFI_PROBE_LIBRARY=<variant .so> selects a variant build; the backward call includes its 100 MB zero-fill of g_texels)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nerf_from_image_amd import _lib  # noqa: E402
if os.environ.get('NFI_PROBE_LIBRARY'):
    _lib.LIBRARY = os.environ['NFI_PROBE_LIBRARY']
from nerf_from_image_amd import ops  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    B, R, S = 4, 256, 31
    planes = torch.randn(B, 3, 32, R, R, generator=g).to(dev)
    texels = ops.planes_to_texels(planes)
    w1 = torch.randn(64, 32, generator=g).to(dev); b1 = (0.1 * torch.randn(64, generator=g)).to(dev)
    w2 = torch.randn(11, 64, generator=g).to(dev); b2 = torch.zeros(11, device=dev)
    cell = 2.0 / S
    idx = torch.stack(torch.meshgrid(*[torch.arange(S)] * 3, indexing='ij'), -1).reshape(1, -1, 3).float()
    pts = ((idx + torch.rand(B, S ** 3, 3, generator=g)) * cell - 1.0).mul(0.55 * 0.999).to(dev)
    gd = torch.randn(B, S ** 3, generator=g).to(dev); gg = torch.randn(B, S ** 3, 3, generator=g).to(dev)

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for i in range(n):
            ev[i].record(); out = fn()
        ev[n].record(); torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
        return ms[n // 2], out
    f_ms, (sdf, grad) = timeit(lambda: ops.sdf_gradient_fwd(pts, texels, w1, b1, w2, b2, 0.55))
    b_ms, out = timeit(lambda: ops.sdf_gradient_bwd(pts, texels, w1, b1, w2, b2, 0.55, gd, gg))
    z_ms, _ = timeit(lambda: torch.zeros_like(texels))
    print('%s: %d x %d^3 points: forward %.3f ms, backward %.3f ms (of which zero-fill %.3f)  sums %.6e %.6e %.6e %.6e' % (
        os.path.basename(_lib.LIBRARY), B, S, f_ms, b_ms, z_ms, float(sdf.double().sum()), float(grad.double().sum()),
        float(out['g_texels'].double().abs().sum()), float(out['g_w1'].double().abs().sum())))


if __name__ == '__main__':
    main()
