"""How much does the render kernel's time depend on texel-line locality?  Same scene, same number of
marched rays, cameras that differ only in direction: along a cube axis (all samples of a ray project to
the same texel of one plane and to a straight texel row of the other two) vs a generic diagonal, and with
64x64 instead of 256x256 planes (whole scene = 1.5 MB: L2-resident).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nerf_from_image_amd import ops


def look_at(eye, up):
    eye = torch.tensor(eye, dtype=torch.float32)
    fwd = -eye / eye.norm()
    up = torch.tensor(up, dtype=torch.float32)
    right = torch.linalg.cross(fwd, up); right = right / right.norm()
    tup = torch.linalg.cross(right, fwd)
    cam = torch.eye(4)
    cam[:3, 0], cam[:3, 1], cam[:3, 2], cam[:3, 3] = right, tup, -fwd, eye
    return cam


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1234)
    R, S, A, B = 128, 64, 10, 8
    w1 = torch.randn(64, 32, generator=g).to(dev); b1 = torch.zeros(64, device=dev)
    w2 = torch.randn(1 + A, 64, generator=g).to(dev); b2 = torch.zeros(1 + A, device=dev)
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
    beta = torch.tensor([0.1], device=dev); alpha = torch.tensor([0.05], device=dev)
    focal = torch.full((B,), 1.0254, device=dev)
    image = ops.decoder_pack(w1, b1, w2, b2, A)
    noise_c = torch.rand(B, R, R, S, device=dev); noise_f = torch.rand(B * R * R, S, device=dev)
    for pr in (256, 64):
        texels = ops.planes_to_texels(torch.randn(B, 3, 32, pr, pr, generator=g).to(dev))
        for name, eye, up in (('axis -z', (0.0, 0.0, 1.3), (0.0, 1.0, 0.0)), ('axis -x', (1.3, 0.0, 0.0), (0.0, 0.0, 1.0)),
                              ('diagonal', (0.7506, 0.7506, 0.7506), (0.0, 0.0, 1.0)),
                              ('generic', (1.0, 0.55, 0.62), (0.0, 0.0, 1.0))):
            cam = look_at(eye, up).repeat(B, 1, 1).to(dev)
            ws = None
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for i in range(13):
                if i == 3:
                    evs[0].record()
                out = ops.render_fwd(cam, focal, R, R, S, texels, image, 0.55, A, att, True, beta, alpha, noise_coarse=noise_c,
                                     noise_fine=noise_f, workspace=ws, taps=('hit',) if i == 0 else ())
                if i == 0:
                    marched = int(((out['hit'] & 2) != 0).sum())
                ws = out['_workspace']
            evs[1].record(); torch.cuda.synchronize()
            ms = evs[0].elapsed_time(evs[1]) / 10
            print('planes %3d^2  camera %-9s marched %6d rays  %.3f ms  %.1f M marched rays/s' % (
                pr, name, marched, ms, marched / ms / 1e3))


if __name__ == '__main__':
    main()
