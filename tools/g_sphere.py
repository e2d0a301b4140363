"""The "G-sphere" workload of SURVEY.md 8(d): a default-initialised reference Generator renders an almost empty scene (its SDF
is not sphere-initialised), so for opacity-dependent measurements - importance sampling, early termination, compaction - the
reference's own `pretrain_sdf` procedure (run.py:824-866: Adam on all generator parameters, loss = sdf_distance_loss +
eikonal * sdf_eikonal_loss against the unit sphere, fresh z every iteration) is run for a fixed number of iterations first.

What is measured on the result (p3d_car-like geometry: scene_range 1.4, the unit sphere seen from distance 2, black background):
  * parity of the drop-in against the untouched reference on the pretrained generator (tests/reference_cases.compare);
  * render-only rays/s of the HIP path, exact and with fine-pass termination at eps 1e-5 / 1e-3, with the deviation of the
    latter from the exact launch - the surface is opaque now, so this is the workload on which termination could pay;
  * the reference renderer on the same planes under PyTorch-ROCm.
The pre-training itself runs on the attached model (`hip_regularisers=True, fused_handoff=True`; its gradient parity is
tests/test_reference_gpu.py's business) and is timed.  JSON on stdout (profiles/r6/g_sphere.json).

TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/reference.py).      python tools/g_sphere.py [iterations=300] [batch=8]
"""
import copy
import json
import os
import sys
import tempfile
import time
import types

os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='nfi_miopen_db_'))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import reference_cases as rc  # noqa: E402
from oracle import reference  # noqa: E402


GEOMETRY = 'p3d'


def pretrained_state(dev, iters, batch, curve=None):
    """run.py:824-866 on the attached model: `iters` Adam(lr_g) steps on all generator parameters, loss = sdf_distance_loss +
    0.1 sdf_eikonal_loss against the unit sphere, fresh z every step.  Returns the state dict."""
    import nerf_from_image_amd.generator as nfi_gen
    g = rc.GEOMETRY[GEOMETRY]
    m = reference.modules()
    torch.manual_seed(1234)
    gen = m.generator.Generator(512, g['scene_range'], attention_values=10, use_sdf=True, disable_stylegan_noise=True).to(dev)
    model = nfi_gen.attach(gen, hip_regularisers=True, fused_handoff=True).train().requires_grad_(True)
    opt = torch.optim.Adam(model.parameters(), lr=0.0025)
    gz = torch.Generator(device=dev).manual_seed(7)
    for i in range(iters):
        z = torch.randn((batch, 512), device=dev, generator=gz)
        losses = model(None, z, ['sdf_distance_loss', 'sdf_eikonal_loss'])
        loss_dist, loss_eik = losses['sdf_distance_loss'].mean(), losses['sdf_eikonal_loss'].mean()
        (loss_dist + 0.1 * loss_eik).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if curve is not None and (i % 50 == 0 or i == iters - 1):
            curve.append({'iteration': i, 'sdf_distance_loss': float(loss_dist), 'sdf_eikonal_loss': float(loss_eik)})
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def scene_from_state(state, dev, batch):
    """The pretrained weights in an UNTOUCHED reference Generator and in its attached twin, p3d_car-like cameras: a
    tests/reference_cases.py scene namespace."""
    import nerf_from_image_amd.generator as nfi_gen
    g = rc.GEOMETRY[GEOMETRY]
    m = reference.modules()
    ref_gen = m.generator.Generator(512, g['scene_range'], attention_values=10, use_sdf=True, disable_stylegan_noise=True).to(dev)
    ref_gen.load_state_dict(state)
    with torch.no_grad():
        ref_gen.alpha.fill_(0.05)            # (pretrain_sdf does not move alpha / beta; a trained model's are small: opaque surface)
        ref_gen.beta.fill_(0.1)
    ref_gen = ref_gen.eval().requires_grad_(False)
    cpu = torch.Generator().manual_seed(1235)
    z = torch.randn(batch, 512, generator=cpu).to(dev)
    with torch.no_grad():
        ws = ref_gen.mapping_network(z, None)
    cam = rc.cameras(batch, g['radius'], cpu).to(dev)
    focal = torch.full((batch,), g['focal']).to(dev)
    return types.SimpleNamespace(geometry=GEOMETRY, g=g, gen=ref_gen, hip=nfi_gen.attach(copy.deepcopy(ref_gen)), z=z, ws=ws, cam=cam,
                                 focal=focal, bbox=None, args=reference.render_args(), batch=batch, dev=dev,
                                 dcfg={'scene_range': g['scene_range'], 'white_background': False})


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device('cuda:0')
    from nerf_from_image_amd import ops
    g = rc.GEOMETRY[GEOMETRY]
    rep = {'iterations': iters, 'batch': batch, 'geometry': 'p3d_car-like: scene_range 1.4, camera distance 2, focal 1, black background'}
    curve = []
    torch.cuda.synchronize()
    t0 = time.time()
    state = pretrained_state(dev, iters, batch, curve)
    torch.cuda.synchronize()
    rep['pretrain'] = {'seconds': time.time() - t0, 'curve': curve}
    sc = scene_from_state(state, dev, batch)
    ref_gen, ws, cam, focal = sc.gen, sc.ws, sc.cam, sc.focal
    par = rc.compare(sc, 128, 64, cpu_images=1)
    rep['parity_vs_the_reference'] = {k: par[k] for k in ('mask_mean', 'vs_reference_gpu', 'vs_reference_cpu', 'reference_cpu_vs_gpu_gap',
                                                          'pixels_over_1e-4_vs_reference_gpu')}

    # ---- render only, on the pretrained generator's planes ----
    R, S = 128, 64
    with torch.no_grad():
        planes = ref_gen.synthesis_network(ws[:, :14]).view(batch, 3, 32, 256, 256).contiguous()
        att = rc._attention(ref_gen, ws)
    dec = ref_gen.decoder.net
    texels = ops.planes_to_texels(planes)
    image = ops.decoder_pack(dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, 10)
    gn = torch.Generator(device=dev).manual_seed(99)
    nc = torch.rand((batch, R, R, S), device=dev, generator=gn)
    nf = torch.rand((batch * R * R, S), device=dev, generator=gn)

    def render(eps):
        return ops.render_fwd(cam, focal, R, R, S, texels, image, g['scene_range'], 10, att, True, ref_gen.beta, ref_gen.alpha,
                              noise_coarse=nc, noise_fine=nf, fine_sampling=True, white_background=False, skip_missed_rays=True,
                              termination_eps=eps)

    def timed(fn, n):
        for _ in range(3):
            fn()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        for i in range(n):
            evs[i].record()
            fn()
        evs[n].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        return per[n // 2]
    exact = render(0.0)
    ro, rd = ops.raygen(R, R, focal, cam, normalize=True)
    hit = ops.near_far(ro, rd, g['scene_range'], strict=False)[2]
    rows = {}
    for eps in (0.0, 1e-5, 1e-3):
        ms = timed(lambda: render(eps), 40)
        out = render(eps)
        rows['eps_%g' % eps] = {'ms': ms, 'rays_per_s': batch * R * R / (ms * 1e-3),
                                'max_abs_drgb_vs_exact': float((out['rgb'] - exact['rgb']).abs().max()),
                                'max_abs_dmask_vs_exact': float((out['mask'] - exact['mask']).abs().max())}
    for k in ('eps_1e-05', 'eps_0.001'):
        rows[k]['x_exact_rate'] = rows[k]['rays_per_s'] / rows['eps_0']['rays_per_s']
    rep['render_only_hip'] = dict(rows, rays_hitting_the_cube_fraction=float(hit.float().mean()), mask_mean=float(exact['mask'].mean()),
                                  mean_transmittance_behind_the_surface=float(1.0 - exact['mask'][exact['mask'] > 0.5].mean()))
    # the reference renderer on the same planes (PyTorch-ROCm, TorchScript stages as run.py runs them)
    ren, _ = reference.load_render(sc.args, sc.dcfg)
    planes96 = planes.view(batch, 96, 256, 256)

    def ref_once():
        with torch.no_grad(), rc.frozen_producer(ref_gen, planes96):
            ren(ref_gen, R, R, cam, focal, None, None, ws, S, extra_model_inputs={'attention_values': att})
    ms_ref = timed(ref_once, 5)
    rep['render_only_reference_pytorch_rocm'] = {'ms': ms_ref, 'rays_per_s': batch * R * R / (ms_ref * 1e-3)}
    rep['x_reference'] = rep['render_only_hip']['eps_0']['rays_per_s'] / rep['render_only_reference_pytorch_rocm']['rays_per_s']
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
