"""Synthetic stand-in for BASELINE config 3 (--run_inversion on p3d_car, 30 inv_steps): no datasets or
pretrained checkpoints exist in this environment, so a seeded scene is rendered as the target, the
latent code and the camera pose are perturbed, and both are recovered with Adam(2e-3, betas=(0.9, 0.95))
(run.py:2007) for N steps - once with the HIP renderer (nerf_from_image_amd.render, gradients from the HIP
backward kernels) and once with the oracle renderer under PyTorch autograd, fed the SAME noise draws.
Reported per step: PSNR and mask IoU (lib/metrics.py:30-45, 79-94 restated) of both runs.

  python tools/inversion_synthetic.py [--res 128 --samples 64 --batch 4 --steps 30]
"""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def psnr(pred, target):
    """lib/metrics.py:30-45: images in [-1, 1] -> [0, 1], per-image PSNR."""
    p, t = (pred.clamp(-1, 1) + 1) / 2, (target.clamp(-1, 1) + 1) / 2
    mse = ((p - t) ** 2).flatten(1).mean(dim=1)
    return -10 * torch.log10(mse.clamp_min(1e-12))


def iou(pred_mask, target_mask):
    """lib/metrics.py:79-94: masks thresholded at 0.5."""
    a, b = pred_mask > 0.5, target_mask > 0.5
    inter = (a & b).flatten(1).sum(1).float()
    union = (a | b).flatten(1).sum(1).float()
    return inter / union.clamp_min(1)


def pose_matrix(cam0, delta):
    """cam0 [B,4,4] composed with a small rigid motion delta [B,6] (axis-angle, translation)."""
    w, t = delta[:, :3], delta[:, 3:]
    theta = w.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    k = w / theta
    K = torch.zeros(w.shape[0], 3, 3, dtype=w.dtype, device=w.device)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    eye = torch.eye(3, dtype=w.dtype, device=w.device).expand_as(K)
    s, c = torch.sin(theta)[..., None], torch.cos(theta)[..., None]
    R = eye + s * K + (1 - c) * (K @ K)
    M = torch.eye(4, dtype=w.dtype, device=w.device).repeat(w.shape[0], 1, 1)
    M[:, :3, :3] = R @ cam0[:, :3, :3]
    M[:, :3, 3] = cam0[:, :3, 3] + t
    return M


def run(dev, res=32, samples=32, batch=2, steps=10, plane_res=48, seed=0, verbose=False, teacher=False, hip_only=False,
        fine=True, staged=False):
    """fine=False: one pass of `samples` samples (run.py without --fine_sampling; the inversion loop then asks for
    4 x 128 = 512, run.py:2271).  staged=True (A/B): the sampler closure is handed to render() without its `.fused`
    handle, which sends the call down the stage-by-stage path (exact-fp32 decoder arithmetic there)."""
    from stand_in import StandInGenerator, look_at_cameras
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    from oracle import nfi_oracle as orc

    torch.manual_seed(seed)
    scene_range = 0.55
    model = StandInGenerator(scene_range, attention_values=10, use_sdf=True, plane_res=plane_res).to(dev).eval()
    with torch.no_grad():
        model.alpha.fill_(0.1)
    for p in model.parameters():
        p.requires_grad_(False)
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(seed + 1)
    cam_true = look_at_cameras(batch, 1.5, g).to(dev)
    focal = torch.full((batch,), 1.0254, device=dev)
    z_true = torch.randn(batch, 512, generator=g).to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=fine)
    dcfg = {'scene_range': scene_range, 'white_background': False}
    assert fine or not teacher
    target_model = model
    if staged:
        def target_model(viewdirs, model_input, request, extra_inputs={}):
            out = model(viewdirs, model_input, request, extra_inputs)
            closure = out['sampler']
            out['sampler'] = lambda x, req=['sigma', 'rgb']: closure(x, req)
            return out
    # strict_near_far off: no host synchronisation inside the step (every camera of this loop looks at the cube), as in
    # tools/train_bench.py; the default (on) reads the hit counter back per render, like the reference's boolean-mask min()
    render = nfi_render.make_render(cfg, dcfg, strict_near_far=False)
    with torch.no_grad():
        ws_true = model.mapping_network(z_true)
        # make the random scene opaque enough to have a silhouette: centre the distance output so that
        # about half of the cube is inside, and use a sharp density
        probe = (torch.rand(batch, 4096, 3, device=dev) * 2 - 1) * scene_range
        sdf = model(None, ws_true, ['sampler'])['sampler'](probe, ['sdf_distance'])['sdf_distance']
        model.decoder.net[2].bias[0] -= sdf.median()
        model.alpha.fill_(0.03)
        target_rgb, _, target_mask, _, _, _ = render(model, res, res, cam_true, focal, None, None, ws_true, samples)

    ws0 = (ws_true + 0.35 * torch.randn(ws_true.shape, generator=g).to(dev)).detach()
    delta0 = torch.cat((0.06 * torch.randn(batch, 3, generator=g), 0.04 * torch.randn(batch, 3, generator=g)), dim=1).to(dev)

    def oracle_render(ws, cam, draws):
        planes, att = model.planes_and_values(ws)
        dec = model.decoder.net
        o = orc.render(planes, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, cam, focal, res, res, samples,
                       scene_range, white_background=False, fine_sampling=fine, noise_coarse=draws[0],
                       noise_fine=draws[1] if fine else None, use_sdf=True, beta=model.beta, alpha=model.alpha,
                       attention_values=att)
        return o['rgb'], o['mask']

    def evaluate(which, ws, delta, nc, nf):
        cam = pose_matrix(cam_true, delta)
        if which == 'hip':
            draws = iter((nc, nf))
            real_rand = torch.rand
            torch.rand = lambda *a, **k: next(draws)          # inject the shared noise
            try:
                rgb, _, mask, _, _, _ = render(target_model, res, res, cam, focal, None, None, ws, samples)
            finally:
                torch.rand = real_rand
        else:
            rgb, mask = oracle_render(ws, cam, (nc, nf))
        return rgb, mask, ((rgb - target_rgb) ** 2).mean() + ((mask - target_mask) ** 2).mean()

    def optimise(which, grad_noise=0.0, shadow=None):
        """Adam on (ws, delta) with renderer `which`.  grad_noise: relative Gaussian perturbation of every gradient
        element before the update (how fast does a trajectory drift under a perturbation of that size?).  shadow: a
        second renderer evaluated at the SAME parameters every step (its loss and gradients are compared, its gradients
        are not applied) - the comparison that does not go through Adam's chaotic amplification."""
        ws = ws0.clone().requires_grad_()
        delta = delta0.clone().requires_grad_()
        opt = torch.optim.Adam([ws, delta], lr=2e-3, betas=(0.9, 0.95))
        hist, t_steps = [], []
        noise_gen = torch.Generator(device=dev).manual_seed(seed + 99)
        for step in range(steps + 1):
            nc = torch.rand((batch, res, res, samples), generator=noise_gen, device=dev)
            nf = torch.rand((batch * res * res, samples), generator=noise_gen, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rgb, mask, loss = evaluate(which, ws, delta, nc, nf)
            with torch.no_grad():
                hist.append((float(psnr(rgb, target_rgb).mean()), float(iou(mask, target_mask).mean()), float(loss)))
            if step == steps:
                break
            opt.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
            t_steps.append(time.perf_counter() - t0)
            if shadow is not None:
                ws2, delta2 = ws.detach().clone().requires_grad_(), delta.detach().clone().requires_grad_()
                _, _, loss2 = evaluate(shadow, ws2, delta2, nc, nf)
                loss2.backward()
                rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
                shadow_hist.append((float(loss2), float(loss), rel(ws2.grad, ws.grad), rel(delta2.grad, delta.grad)))
            if grad_noise:
                with torch.no_grad():
                    for q in (ws, delta):
                        q.grad.mul_(1.0 + grad_noise * torch.randn(q.grad.shape, generator=noise_gen, device=dev))
            opt.step()
        t_steps.sort()                                       # median: the first steps carry one-time module loads
        return hist, (t_steps[len(t_steps) // 2] if t_steps else 0.0)

    shadow_hist = []
    h_hip, t_hip = optimise('hip')
    if hip_only:                                             # profiling: the HIP loop alone
        print('render fwd+bwd per step (median): HIP %.2f ms  (B=%d, %dx%d, %s samples%s)' % (
            t_hip * 1e3, batch, res, res, '%d+%d' % (samples, samples) if fine else '%d single-pass' % samples,
            ', staged path' if staged else ''))
        return h_hip, None, t_hip, None
    h_ref, t_ref = optimise('oracle', shadow='hip' if teacher else None)
    if teacher:
        # conditioning of the problem itself: the oracle's gradient at the start point in float64 against float32
        import copy
        m64 = copy.deepcopy(model).double()
        gen0 = torch.Generator(device=dev).manual_seed(seed + 99)
        nc = torch.rand((batch, res, res, samples), generator=gen0, device=dev)
        nf = torch.rand((batch * res * res, samples), generator=gen0, device=dev)
        grads = {}
        for name, dt in (('f64', torch.float64), ('f32', torch.float32), ('hip', None)):
            ws, delta = ws0.clone().to(dt or torch.float32).requires_grad_(), delta0.clone().to(dt or torch.float32).requires_grad_()
            if name == 'hip':
                _, _, l = evaluate('hip', ws, delta, nc, nf)
            else:
                m = m64 if dt == torch.float64 else model
                planes, att = m.planes_and_values(ws)
                dec = m.decoder.net
                o = orc.render(planes, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, pose_matrix(cam_true.to(dt), delta),
                               focal.to(dt), res, res, samples, scene_range, white_background=False, fine_sampling=True,
                               noise_coarse=nc.to(dt), noise_fine=nf.to(dt), use_sdf=True, beta=m.beta, alpha=m.alpha,
                               attention_values=att)
                l = ((o['rgb'] - target_rgb.to(dt)) ** 2).mean() + ((o['mask'] - target_mask.to(dt)) ** 2).mean()
            l.backward()
            grads[name] = (ws.grad.double(), delta.grad.double(), float(l))
        rel = lambda a, b: float((a - b).norm() / b.norm())
        conditioning = {k: (rel(grads[k][0], grads['f64'][0]), rel(grads[k][1], grads['f64'][1]), grads[k][2]) for k in ('f32', 'hip')}
        # (HIP loss, oracle loss, relative gradient error on ws, on the pose delta) along the ORACLE's trajectory, and the
        # oracle's own trajectory under a 1e-4 relative perturbation of its gradients
        h_pert, _ = optimise('oracle', grad_noise=1e-4)
        return h_hip, h_ref, t_hip, t_ref, shadow_hist, h_pert, conditioning
    if verbose:
        print('step   HIP psnr   iou     loss      | oracle(PyTorch-ROCm) psnr   iou     loss')
        for i, (a, b) in enumerate(zip(h_hip, h_ref)):
            print('%4d   %7.3f  %.4f  %.6f |            %7.3f  %.4f  %.6f' % (i, a[0], a[1], a[2], b[0], b[1], b[2]))
        print('render fwd+bwd per step (median): HIP %.2f ms, oracle %.2f ms (B=%d, %dx%d, %d+%d samples)' % (
            t_hip * 1e3, t_ref * 1e3, batch, res, res, samples, samples))
    return h_hip, h_ref, t_hip, t_ref


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=128)
    ap.add_argument('--samples', type=int, default=64)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--plane-res', type=int, default=256)
    ap.add_argument('--hip-only', action='store_true')
    ap.add_argument('--no-fine', action='store_true', help='one pass of --samples samples (run.py without --fine_sampling)')
    ap.add_argument('--staged', action='store_true', help='A/B: force the stage-by-stage path')
    a = ap.parse_args()
    run(torch.device('cuda:0'), a.res, a.samples, a.batch, a.steps, a.plane_res, verbose=True, hip_only=a.hip_only,
        fine=not a.no_fine, staged=a.staged)
