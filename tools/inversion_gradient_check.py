"""Gradient of the synthetic inversion loss at the start point: HIP path against the float64 oracle, over variants."""
import sys, os, types, copy, torch
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools')); sys.path.insert(0, os.path.join(R, 'tests'))
from stand_in import StandInGenerator, look_at_cameras
import nerf_from_image_amd.generator as nfi_gen
import nerf_from_image_amd.render as nfi_render
from oracle import nfi_oracle as orc
from inversion_synthetic import pose_matrix
dev = torch.device('cuda:0')


def case(alpha=0.03, white=False, res=32, samples=32, plane_res=48, batch=2, seed=0, fine=True, centre=True, what='both'):
    torch.manual_seed(seed)
    scene_range = 0.55
    model = StandInGenerator(scene_range, attention_values=10, use_sdf=True, plane_res=plane_res).to(dev).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(seed + 1)
    cam_true = look_at_cameras(batch, 1.5, g).to(dev)
    focal = torch.full((batch,), 1.0254, device=dev)
    z_true = torch.randn(batch, 512, generator=g).to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=fine)
    render = nfi_render.make_render(cfg, {'scene_range': scene_range, 'white_background': white})
    with torch.no_grad():
        ws_true = model.mapping_network(z_true)
        if centre:
            probe = (torch.rand(batch, 4096, 3, device=dev) * 2 - 1) * scene_range
            sdf = model(None, ws_true, ['sampler'])['sampler'](probe, ['sdf_distance'])['sdf_distance']
            model.decoder.net[2].bias[0] -= sdf.median()
        model.alpha.fill_(alpha)
    ws0 = (ws_true + 0.35 * torch.randn(ws_true.shape, generator=g).to(dev)).detach()
    delta0 = torch.cat((0.06 * torch.randn(batch, 3, generator=g), 0.04 * torch.randn(batch, 3, generator=g)), dim=1).to(dev)
    gen0 = torch.Generator(device=dev).manual_seed(seed + 99)
    nc = torch.rand((batch, res, res, samples), generator=gen0, device=dev)
    nf = torch.rand((batch * res * res, samples), generator=gen0, device=dev)
    w_rgb = torch.randn(batch, res, res, 3, generator=gen0, device=dev)
    w_mask = torch.randn(batch, res, res, generator=gen0, device=dev)
    m64 = copy.deepcopy(model).double()
    out = {}
    for name, dt in (('f64', torch.float64), ('hip', torch.float32)):
        ws, delta = ws0.clone().to(dt).requires_grad_(), delta0.clone().to(dt).requires_grad_()
        cam = pose_matrix(cam_true.to(dt), delta)
        if name == 'hip':
            draws = iter((nc, nf))
            real_rand = torch.rand
            torch.rand = lambda *a, **k: next(draws)
            try:
                rgb, _, mask, _, _, _ = render(model, res, res, cam, focal, None, None, ws, samples)
            finally:
                torch.rand = real_rand
        else:
            planes, att = m64.planes_and_values(ws)
            dec = m64.decoder.net
            o = orc.render(planes, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, cam, focal.to(dt), res, res, samples,
                           scene_range, white_background=white, fine_sampling=fine, noise_coarse=nc.to(dt),
                           noise_fine=nf.to(dt) if fine else None, use_sdf=True, beta=m64.beta, alpha=m64.alpha, attention_values=att)
            rgb, mask = o['rgb'], o['mask']
        for lname, l in (('rgb', (rgb * w_rgb.to(dt)).sum()), ('mask', (mask * w_mask.to(dt)).sum())):
            gw, gd = torch.autograd.grad(l, (ws, delta), retain_graph=True)
            out[(name, lname)] = (gw.double(), gd.double(), rgb.detach().double(), mask.detach().double())
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    r = {l: (rel(out[('hip', l)][0], out[('f64', l)][0]), rel(out[('hip', l)][1], out[('f64', l)][1])) for l in ('rgb', 'mask')}
    fw = (float((out[('hip', 'rgb')][2] - out[('f64', 'rgb')][2]).abs().max()), float((out[('hip', 'rgb')][3] - out[('f64', 'rgb')][3]).abs().max()))
    return r, fw


variants = [dict(), dict(alpha=0.2), dict(white=True), dict(fine=False), dict(plane_res=64), dict(res=16), dict(samples=64),
            dict(centre=False), dict(alpha=0.2, white=True, fine=False), dict(batch=1)]
for v in variants:
    r, fw = case(**v)
    print('%-45s rgb-loss grad rel err ws %.2e delta %.2e | mask-loss ws %.2e delta %.2e | forward rgb %.1e mask %.1e' % (
        v, r['rgb'][0], r['rgb'][1], r['mask'][0], r['mask'][1], fw[0], fw[1]))
