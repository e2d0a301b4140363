"""Quick timing of the fused render on synthetic shapenet_chairs-like inputs (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerf_from_image_amd import ops


def cameras(n, radius, gen):
    v = torch.randn(n, 3, generator=gen)
    eye = radius * v / v.norm(dim=-1, keepdim=True)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    up = torch.tensor([0., 0., 1.]).expand(n, 3)
    right = torch.cross(fwd, up, dim=-1); right = right / right.norm(dim=-1, keepdim=True)
    tup = torch.cross(right, fwd, dim=-1)
    cam = torch.eye(4).repeat(n, 1, 1)
    cam[:, :3, 0] = right; cam[:, :3, 1] = tup; cam[:, :3, 2] = -fwd; cam[:, :3, 3] = eye
    return cam


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1234)
    R, S, A = 128, 64, 10
    for B, radius in ((1, 2.0), (8, 2.0), (8, 1.3)):
        planes = torch.randn(B, 3, 32, 256, 256, generator=g).to(dev)
        w1 = torch.randn(64, 32, generator=g).to(dev); b1 = torch.zeros(64, device=dev)
        w2 = torch.randn(1 + A, 64, generator=g).to(dev); b2 = torch.zeros(1 + A, device=dev)
        att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
        beta = torch.tensor([0.1], device=dev); alpha = torch.tensor([0.05], device=dev)
        cam = cameras(B, radius, g).to(dev); focal = torch.full((B,), 1.0254, device=dev)
        n = B * R * R
        noise_c = torch.rand(B, R, R, S, device=dev); noise_f = torch.rand(n, S, device=dev)
        image = ops.decoder_pack(w1, b1, w2, b2, A)
        texels = ops.planes_to_texels(planes)
        ws = None
        knobs = [int(v) for v in os.environ['NFI_TUNING'].split(',')] if os.environ.get('NFI_TUNING') else (0, 8, 16)
        for skip, tuning in [(True, t) for t in knobs]:      # default / fp32 MLP / single work counter (NFI_TUNING=a,b,.. overrides)
            def step():
                return ops.render_fwd(cam, focal, R, R, S, texels, image, 0.55, A, att, True, beta, alpha,
                                      noise_coarse=noise_c, noise_fine=noise_f, skip_missed_rays=skip, workspace=ws,
                                      tuning=tuning)
            for _ in range(3):
                out = step(); ws = out['_workspace']
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K = int(os.environ.get("NFI_ITERS", 20))
            e0.record()
            for _ in range(K):
                out = step()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / K
            print('B=%d radius=%.1f skip=%d tuning=%d: %.3f ms/step  %.2f Mrays/s  mask mean %.3f rgb mean %.3f' % (
                B, radius, skip, tuning, ms, n / ms / 1e3, out['mask'].mean().item(), out['rgb'].mean().item()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            texels = ops.planes_to_texels(planes)
        e1.record(); torch.cuda.synchronize()
        print('   planes_to_texels: %.3f ms' % (e0.elapsed_time(e1) / 10))


if __name__ == '__main__':
    main()


def phase_profile(radius=1.3):
    """Per-phase shader-cycle breakdown of the render kernel (profiling instantiation)."""
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1234)
    R, S, A, B = 128, 64, 10, 8
    planes = torch.randn(B, 3, 32, 256, 256, generator=g).to(dev)
    w1 = torch.randn(64, 32, generator=g).to(dev); b1 = torch.zeros(64, device=dev)
    w2 = torch.randn(1 + A, 64, generator=g).to(dev); b2 = torch.zeros(1 + A, device=dev)
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
    beta = torch.tensor([0.1], device=dev); alpha = torch.tensor([0.05], device=dev)
    cam = cameras(B, radius, g).to(dev); focal = torch.full((B,), 1.0254, device=dev)
    noise_c = torch.rand(B, R, R, S, device=dev); noise_f = torch.rand(B * R * R, S, device=dev)
    image = ops.decoder_pack(w1, b1, w2, b2, A); texels = ops.planes_to_texels(planes)
    prof = torch.zeros(12, dtype=torch.int64, device=dev)
    for _ in range(2):
        prof.zero_()
        ops.render_fwd(cam, focal, R, R, S, texels, image, 0.55, A, att, True, beta, alpha, noise_coarse=noise_c,
                       noise_fine=noise_f, profile_cycles=prof)
        torch.cuda.synchronize()
    p = prof.cpu().tolist()
    names = ['tile issue', 'tile wait+interp', 'tile transpose+mlp+epilogue', 'tiles', 'ray set-up', 'coarse field',
             'resample', 'fine field', 'merge', 'composite+store', 'rays', 'wave lifetime']
    rays, tiles = max(p[10], 1), max(p[3], 1)
    print('phase profile (B=8, camera radius %.1f, %d of %d rays marched): cycles per marched ray per wave' % (radius, rays, B * R * R))
    for i in (4, 5, 6, 7, 8, 9):
        print('   %-30s %9.0f' % (names[i], p[i] / rays))
    print('   %-30s %9.0f  (sum of wave lifetimes / rays marched; phases above sum to %.0f)' % (
        'total', p[11] / rays, sum(p[i] for i in (4, 5, 6, 7, 8, 9)) / rays))
    for i in (0, 1, 2):
        print('   per tile: %-20s %9.0f' % (names[i], p[i] / tiles))
    print('   tiles/ray %.2f' % (tiles / rays))


if __name__ == '__main__' and os.environ.get('NFI_PHASES'):
    for radius in (1.3, 2.0):
        phase_profile(radius)
