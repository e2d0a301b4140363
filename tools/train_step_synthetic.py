"""Synthetic stand-in for BASELINE config 4 (GAN training on cub: orthographic camera, scene_range 2.0, black
background, supervise_alpha, SDF regularisers every G step; run.py:930-1030): one generator-side step =
render(fwd) + regulariser branch + image/mask loss + backward into the plane producer, decoder, beta/alpha, 4 images
per GPU at 128x128 with 64+64 samples.  The discriminator, data pipeline and optimiser are outside the hot path and
are replaced by an L2 loss against fixed targets.  Timed once with the HIP path (nerf_from_image_amd) and once with
the oracle (reference ATen op sequence) under PyTorch-ROCm autograd.

  python tools/train_step_synthetic.py [--batch 4 --res 128 --samples 64 --steps 10]
"""
import argparse
import copy
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def run(dev, batch=4, res=128, samples=64, steps=10, plane_res=256, verbose=True):
    from stand_in import StandInGenerator, look_at_cameras
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    from oracle import nfi_oracle as orc
    torch.manual_seed(0)
    scene_range = 2.0
    model = StandInGenerator(scene_range, attention_values=10, use_sdf=True, plane_res=plane_res).to(dev).train()
    with torch.no_grad():
        model.alpha.fill_(0.05)
    ref_model = copy.deepcopy(model)
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(1)
    cam = look_at_cameras(batch, 3.0, g).to(dev)              # ortho: focal None, extent from cam[3,3] = 1
    z = torch.randn(batch, 512, generator=g).to(dev)
    target_rgb = (torch.rand(batch, res, res, 3, generator=g) * 2 - 1).to(dev)
    target_mask = (torch.rand(batch, res, res, generator=g) > 0.5).float().to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': scene_range, 'white_background': False},
                                    strict_near_far=False)      # no host synchronisation inside the step (tools/train_bench.py)
    reg_names = ['sdf_eikonal_loss', 'sdf_distance_loss']

    def hip_step():
        rgb, _, mask, _, _, extra = render(model, res, res, cam, None, None, None, z, samples,
                                           extra_model_outputs=reg_names)
        loss = ((rgb - target_rgb) ** 2).mean() + ((mask - target_mask) ** 2).mean() + \
            0.1 * extra['sdf_eikonal_loss'].mean() + extra['sdf_distance_loss'].mean()
        return loss

    def ref_step():
        planes, att = ref_model.planes_and_values(z)
        dec = ref_model.decoder.net
        w = (dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias)
        nc = torch.rand(batch, res, res, samples, device=dev)
        nf = torch.rand(batch * res * res, samples, device=dev)
        o = orc.render(planes, *w, cam, None, res, res, samples, scene_range, white_background=False, fine_sampling=True,
                       noise_coarse=nc, noise_fine=nf, use_sdf=True, beta=ref_model.beta, alpha=ref_model.alpha,
                       attention_values=att)
        jitter = torch.rand(batch, 31, 31, 31, 3, device=dev)
        reg = orc.regularisers(planes, *w, orc.stratified_volume(batch, 32, scene_range, jitter), scene_range, True,
                               ref_model.beta, None)
        return ((o['rgb'] - target_rgb) ** 2).mean() + ((o['mask'] - target_mask) ** 2).mean() + \
            0.1 * reg['sdf_eikonal_loss'].mean() + reg['sdf_distance_loss'].mean()

    out = {}
    for name, step, m in (('hip', hip_step, model), ('reference_path', ref_step, ref_model)):
        params = [p for p in m.parameters() if p.requires_grad]
        times = []
        for i in range(steps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = step()
            grads = torch.autograd.grad(loss, params, allow_unused=True)
            torch.cuda.synchronize()
            if i >= 2:
                times.append(time.perf_counter() - t0)
        times.sort()
        out[name] = dict(ms_per_step=times[len(times) // 2] * 1e3, loss=float(loss.detach()),
                         grad_norm=float(sum((g_ ** 2).sum() for g_ in grads if g_ is not None) ** 0.5))
        if verbose:
            print('%-15s %.2f ms per generator-side step (fwd + regularisers + bwd), loss %.4f, |grad| %.4f' % (
                name, out[name]['ms_per_step'], out[name]['loss'], out[name]['grad_norm']))
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--res', type=int, default=128)
    ap.add_argument('--samples', type=int, default=64)
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    run(torch.device('cuda:0'), a.batch, a.res, a.samples, a.steps)
