"""Timing of the field-query backward kernel (B=4 scenes x 1M points = one pass of a 128x128x64 render)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerf_from_image_amd import _lib
if os.environ.get('NFI_PROBE_LIBRARY'):          # variant builds of tools/probes/bwd_variants.py
    _lib.LIBRARY = os.environ['NFI_PROBE_LIBRARY']
from nerf_from_image_amd import ops
from nerf_from_image_amd.field_backward import field_query_bwd


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    B, P, A, R = 4, 128 * 128 * 64, 10, 256
    planes = torch.randn(B, 3, 32, R, R, generator=g).to(dev)
    w1 = torch.randn(64, 32, generator=g).to(dev); b1 = torch.zeros(64, device=dev)
    w2 = torch.randn(1 + A, 64, generator=g).to(dev); b2 = torch.zeros(1 + A, device=dev)
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
    beta = torch.tensor([0.1], device=dev); alpha = torch.tensor([0.05], device=dev)
    texels = ops.planes_to_texels(planes); image = ops.decoder_pack(w1, b1, w2, b2, A)
    # points along rays through the cube (sorted-ish along rays like a coarse pass)
    o = torch.randn(B, 128 * 128, 1, 3, generator=g) * 0.05
    d = torch.nn.functional.normalize(torch.randn(B, 128 * 128, 1, 3, generator=g), dim=-1)
    t = torch.linspace(-0.5, 0.5, 64).view(1, 1, 64, 1)
    x = (o + d * t).reshape(B, P, 3).to(dev)
    gs = torch.randn(B, P, device=dev); gr = torch.randn(B, P, 3, device=dev)

    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    fwd = timeit(lambda: ops.field_query(x, texels, image, 0.55, A, att, True, beta, alpha))
    full = timeit(lambda: field_query_bwd(x, texels, image, w1, w2, 0.55, A, att, True, beta, alpha, gs, gr, want_points=True, scatter_mode=0))
    binned = timeit(lambda: field_query_bwd(x, texels, image, w1, w2, 0.55, A, att, True, beta, alpha, gs, gr, want_points=True, scatter_mode=1))
    nocoord = timeit(lambda: field_query_bwd(x, texels, image, w1, w2, 0.55, A, att, True, beta, alpha, gs, gr, want_points=False, scatter_mode=0))
    ponly = timeit(lambda: field_query_bwd(x, texels, image, w1, w2, 0.55, A, att, True, beta, alpha, gs, gr, points_only=True))
    # zero upstream gradient: everything but the atomics and (skipped) scatter
    z1, z3 = torch.zeros_like(gs), torch.zeros_like(gr)
    zero = timeit(lambda: field_query_bwd(x, texels, image, w1, w2, 0.55, A, att, True, beta, alpha, z1, z3, want_points=False))
    print('points %.1fM: fwd %.2f ms | bwd full(+coord) atomic scatter %.2f, binned scatter %.2f | bwd no coord grads %.2f | points_only (no atomics, no dW) %.2f | '
          'zero upstream (no atomics) %.2f' % (B * P / 1e6, fwd, full, binned, nocoord, ponly, zero))


if __name__ == '__main__':
    main()
