"""Plane-producer hand-off at the reference's size (last synthesis block of the 256x256 network: 128 feature
channels -> 96 image channels, B scenes): the fused kernel (upsample + torgb + add -> texels, one launch) against the
sequence it replaces (PyTorch upsample / modulated 1x1 conv / add as in models/stylegan.py:424-433, then
nfi_planes_to_texels), forward and backward, HIP events.   python tools/bench_handoff.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for i in range(iters):
        ev[i].record()
        fn()
    ev[iters].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return t[len(t) // 2]


def main():
    from nerf_from_image_amd import handoff, ops
    dev = torch.device('cuda:0')
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Cin, R = 128, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, R, R, generator=g).to(dev)
    s = (torch.randn(B, Cin, generator=g) / Cin ** 0.5).to(dev)
    w = torch.randn(96, Cin, 1, 1, generator=g).to(dev)
    bias = torch.randn(96, generator=g).to(dev)
    prev = torch.randn(B, 96, R // 2, R // 2, generator=g).to(dev)
    h = torch.tensor([1., 3., 3., 1.], device=dev)
    h = (h[:, None] * h[None, :]) / 64 * 4

    def reference_tail(x_, s_, w_, b_, prev_):
        up = F.conv_transpose2d(prev_.flatten(0, 1).unsqueeze(1), h[None, None], padding=1, stride=2).view(B, 96, R, R)
        y = F.conv2d(x_ * s_.reshape(B, -1, 1, 1), w_) + b_.view(1, -1, 1, 1)
        return up.add_(y)

    with torch.no_grad():
        t_ref = timed(lambda: ops.planes_to_texels(reference_tail(x, s, w, bias, prev).view(B, 3, 32, R, R)))
        t_p2t = timed(lambda: ops.planes_to_texels(x[:, :96].reshape(B, 3, 32, R, R)))
        t_fused = timed(lambda: ops.torgb_texels(x, s, w.view(96, Cin), bias, prev))
        a = ops.planes_to_texels(reference_tail(x, s, w, bias, prev).view(B, 3, 32, R, R))          # [B,3,R,R,32]
        b = ops.torgb_texels(x, s, w.view(96, Cin), bias, prev)                                      # [B,96,R,R] channels-last
        err = (a.permute(0, 2, 3, 1, 4) - b.permute(0, 2, 3, 1).reshape(B, R, R, 3, 32)).abs().max().item()
    leaves = [t.clone().requires_grad_() for t in (x, s, w, bias, prev)]
    gt = torch.randn(B, 3, R, R, 32, device=dev)                    # gradient w.r.t. planar texels
    gcl = torch.randn(B, 96, R, R, device=dev).contiguous(memory_format=torch.channels_last)

    def ref_fb():
        out = reference_tail(*leaves)
        torch.autograd.grad(out, leaves, ops.texels_to_planes(gt).view(B, 96, R, R))

    def fused_fb():
        out = handoff.torgb_upsample_add(*leaves)
        torch.autograd.grad(out, leaves, gcl)
    t_ref_fb, t_fused_fb = timed(ref_fb, 15, 3), timed(fused_fb, 15, 3)
    gb = B * R * R * (4 * Cin + 384 + 96) / 1e9
    print('hand-off at B=%d, %d -> 96 channels, %dx%d (max |fused - reference sequence| = %.2e)' % (B, Cin, R, R, err))
    print('  forward : fused kernel %.3f ms (%.2f TB/s of its compulsory %.2f GB) | PyTorch tail + planes_to_texels %.3f ms '
          '(planes_to_texels alone %.3f ms)' % (t_fused, gb / t_fused, gb, t_ref, t_p2t))
    print('  fwd+bwd : fused %.3f ms | PyTorch tail + layout kernels %.3f ms' % (t_fused_fb, t_ref_fb))


if __name__ == '__main__':
    main()
