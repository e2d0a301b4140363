"""The reference-shaped host API (Generator.forward -> sampler closure, run.py::render,
lib/nerf_utils.py functions) on the GPU, against the oracle fed with the same noise."""
import types

import pytest
import torch

from parity_util import err
from stand_in import StandInGenerator, look_at_cameras
import nerf_from_image_amd.generator as nfi_gen
import nerf_from_image_amd.nerf_utils as nu
import nerf_from_image_amd.render as nfi_render
from oracle import nfi_oracle as orc

pytestmark = pytest.mark.gpu


def close(a, b, tol, what):
    e = err(a, b)
    assert e['nonfinite'] == 0 and e['max'] <= tol, (what, e)


class RandTap:
    """Records torch.rand draws made on the GPU so the CPU oracle can replay them."""

    def __enter__(self):
        self.draws = []
        self._rand = torch.rand

        def rand(*a, **k):
            out = self._rand(*a, **k)
            self.draws.append(out.detach().cpu())
            return out
        torch.rand = rand
        return self

    def __exit__(self, *a):
        torch.rand = self._rand


@pytest.fixture(scope='module')
def setup(gpu_device):
    torch.manual_seed(1234)
    model = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=48).to(gpu_device).eval()
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(9)
    B = 2
    cam = look_at_cameras(B, 1.6, g).to(gpu_device)
    focal = torch.full((B,), 1.0254, device=gpu_device)
    z = torch.randn(B, 512, generator=g).to(gpu_device)
    return model, cam, focal, z


def oracle_for(model, z, cam, focal, H, W, S, cfg, dcfg, draws, bbox=None, want_semantics=False):
    with torch.no_grad():
        planes, att = model.planes_and_values(z)
        dec = model.decoder.net
        cpu = lambda t: None if t is None else t.detach().cpu()
        return orc.render(cpu(planes), cpu(dec[0].weight), cpu(dec[0].bias), cpu(dec[2].weight), cpu(dec[2].bias),
                          cpu(cam), cpu(focal), H, W, S, dcfg['scene_range'], white_background=dcfg['white_background'],
                          fine_sampling=cfg.fine_sampling, bbox=cpu(bbox),
                          noise_coarse=draws[0] if draws else None,
                          noise_fine=draws[1] if len(draws) > 1 else None, use_sdf=True, beta=cpu(model.beta),
                          alpha=cpu(model.alpha), attention_values=cpu(att), want_semantics=want_semantics)


def test_sampler_closure_matches_reference_semantics(setup):
    model, cam, focal, z = setup
    with torch.no_grad():
        out = model(None, z, ['sampler', 'attention_values'])
        sampler = out['sampler']
        g = torch.Generator().manual_seed(2)
        x = ((torch.rand(2, 5, 6, 7, 3, generator=g) * 2 - 1) * 0.7).to(cam.device)
        res = sampler(x, ['sigma', 'rgb', 'semantics', 'sdf_distance', 'coords'])
        planes, att = model.planes_and_values(z)
        dec = model.decoder.net
        ref = orc.field_query(planes.cpu(), dec[0].weight.cpu(), dec[0].bias.cpu(), dec[2].weight.cpu(),
                              dec[2].bias.cpu(), x.cpu(), 0.55, True, model.beta.cpu(), model.alpha.cpu(), att.cpu())
    assert res['sigma'].shape == (2, 5 * 6 * 7) and res['rgb'].shape == (2, 210, 3)
    assert res['semantics'].shape == (2, 210, 10) and res['sdf_distance'].shape == (2, 210, 1)
    assert res['coords'] is x
    close(out['attention_values'], att, 0, 'attention_values passthrough')
    close(res['sigma'], ref['sigma'], 1e-4 / 0.05, 'sigma')
    close(res['rgb'], ref['rgb'], 1e-4, 'rgb')
    close(res['semantics'], ref['semantics'], 1e-5, 'semantics')
    close(res['sdf_distance'][..., 0], ref['sdf'], 1e-5, 'sdf')


@pytest.mark.parametrize('fine,randomize,white', [(True, True, True), (True, False, False), (False, True, True)])
def test_render_dropin_fused(setup, fine, randomize, white):
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=fine)
    dcfg = {'scene_range': 0.55, 'white_background': white}
    nfi_render.configure(cfg, dcfg)
    H = W = 24
    S = 32
    with torch.no_grad(), RandTap() as tap:
        rgb, depth, mask, normals, sem, extra = nfi_render.render(model, H, W, cam, focal, None, None, z, S,
                                                                  randomize=randomize)
    assert normals is None and sem is None and extra == {}
    assert rgb.shape == (2, H, W, 3) and depth.shape == (2, H, W) and mask.shape == (2, H, W)
    if randomize:
        assert tuple(tap.draws[0].shape) == (2, H, W, S)
        if fine:
            assert tuple(tap.draws[1].shape) == (2 * H * W, S)
    o = oracle_for(model, z, cam, focal, H, W, S, cfg, dcfg, tap.draws)
    close(rgb, o['rgb'], 1e-4, 'rgb'); close(depth, o['depth'], 1e-4, 'depth'); close(mask, o['mask'], 1e-4, 'mask')


def test_graphed_render_replays_the_eager_call(setup):
    """nerf_from_image_amd.graphs.GraphedRender: the whole render() call - plane producer, hand-off, noise, set-up, fused
    kernel - captured in a HIP graph.  Deterministic sampling (randomize=False): bit-identical to the eager call, also for
    NEW cameras and latents copied into the captured buffers; random sampling: every replay draws fresh noise."""
    from nerf_from_image_amd.graphs import GraphedRender
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dcfg = {'scene_range': 0.55, 'white_background': True}
    render = nfi_render.make_render(cfg, dcfg, strict_near_far=False)
    H = W = 24
    S = 32
    graphed = GraphedRender(render, model, H, W, cam, focal, None, None, z, S, randomize=False, compute_semantics=True)
    with torch.no_grad():
        eager = render(model, H, W, cam, focal, None, None, z, S, randomize=False, compute_semantics=True)
    out = graphed(cam, focal, None, None, z)
    for k in (0, 1, 2, 4):
        assert torch.equal(out[k], eager[k]), k
    g = torch.Generator().manual_seed(3)
    cam2 = cam.clone()
    cam2[:, :3, 3] += 0.05 * torch.randn(2, 3, generator=g).to(cam.device)
    z2 = torch.randn(z.shape, generator=g).to(z.device)
    with torch.no_grad():
        eager2 = render(model, H, W, cam2, focal, None, None, z2, S, randomize=False, compute_semantics=True)
    out2 = graphed(cam2, focal, None, None, z2)
    for k in (0, 1, 2, 4):
        assert torch.equal(out2[k], eager2[k]) and not torch.equal(eager2[k], eager[k]), k
    with pytest.raises(ValueError):
        graphed(cam2[:1], focal[:1], None, None, z2[:1])                  # shapes are the graph's
    # random sampling: consecutive replays are consecutive draws
    noisy = GraphedRender(render, model, H, W, cam, focal, None, None, z, S)
    a = noisy(cam, focal, None, None, z)[0].clone()
    b = noisy(cam, focal, None, None, z)[0].clone()
    assert not torch.equal(a, b) and float((a - b).abs().mean()) < 0.05
    # the default (strict) render function reads the hit count back on the host: it cannot be captured, and says so
    with pytest.raises(ValueError, match='strict_near_far=False'):
        GraphedRender(nfi_render.make_render(cfg, dcfg), model, H, W, cam, focal, None, None, z, S)


class OneRenderLaunch:
    """The extra maps must come out of ONE fused render launch: counts ops.render_fwd calls and makes the stage ops of
    the staged path (field query, resampling, compositing) fail."""

    def __enter__(self):
        from nerf_from_image_amd import ops
        self.ops, self.calls = ops, 0
        self.saved = {k: getattr(ops, k) for k in ('render_fwd', 'field_query', 'composite', 'resample')}

        def counted(*a, **k):
            self.calls += 1
            return self.saved['render_fwd'](*a, **k)

        def forbidden(*a, **k):
            raise AssertionError('a stage kernel of the staged path was launched')
        ops.render_fwd = counted
        ops.field_query = ops.composite = ops.resample = forbidden
        return self

    def __exit__(self, *a):
        for k, v in self.saved.items():
            setattr(self.ops, k, v)


@pytest.mark.parametrize('S', [200, 512])
def test_render_dropin_single_pass_beyond_128_samples(setup, S):
    """One pass of more than 128 samples without fine sampling (run.py:2271: 512): ONE fused launch for the plain maps;
    an extra map over such a pass keeps the staged path and agrees with it."""
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=False)
    dcfg = {'scene_range': 0.55, 'white_background': True}
    render = nfi_render.make_render(cfg, dcfg)
    H, W = 16, 24
    with torch.no_grad(), RandTap() as tap, OneRenderLaunch() as one:
        rgb, depth, mask, normals, sem, _ = render(model, H, W, cam, focal, None, None, z, S)
    assert one.calls == 1 and normals is None and sem is None
    assert tuple(tap.draws[0].shape) == (2, H, W, S) and len(tap.draws) == 1
    o = oracle_for(model, z, cam, focal, H, W, S, cfg, dcfg, tap.draws)
    close(rgb, o['rgb'], 1e-4, 'rgb'); close(depth, o['depth'], 1e-4, 'depth'); close(mask, o['mask'], 1e-4, 'mask')
    with torch.no_grad():
        torch.manual_seed(5)
        a = render(model, H, W, cam, focal, None, None, z, S)
        torch.manual_seed(5)
        b = render(model, H, W, cam, focal, None, None, z, S, compute_coords=True)      # staged
    assert b[4].shape == (2, H, W, 3)
    close(a[0], b[0], 1e-5, 'fused vs staged rgb'); close(a[2], b[2], 1e-5, 'fused vs staged mask')
    cfg_f = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    with pytest.raises(NotImplementedError):
        nfi_render.make_render(cfg_f, dcfg)(model, H, W, cam, focal, None, None, z, S)
    # termination_eps acts on the fine pass: a configuration without one ignores it
    with torch.no_grad():
        torch.manual_seed(5)
        c = nfi_render.make_render(cfg, dcfg, termination_eps=1e-5)(model, H, W, cam, focal, None, None, z, S)
    assert torch.equal(c[0], a[0]) and torch.equal(c[2], a[2])


def test_render_dropin_fused_semantics(setup):
    """compute_semantics without a gradient (every inversion eval batch, run.py:2036-2051): one fused launch, the
    semantic map against the oracle, rgb / depth / mask identical to the call without it."""
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dcfg = {'scene_range': 0.55, 'white_background': True}
    render = nfi_render.make_render(cfg, dcfg)
    H, W, S = 20, 28, 32
    with torch.no_grad(), RandTap() as tap, OneRenderLaunch() as one:
        rgb, depth, mask, normals, sem, _ = render(model, H, W, cam, focal, None, None, z, S, compute_semantics=True)
    assert one.calls == 1 and normals is None and sem.shape == (2, H, W, 10)
    o = oracle_for(model, z, cam, focal, H, W, S, cfg, dcfg, tap.draws, want_semantics=True)
    close(rgb, o['rgb'], 1e-4, 'rgb'); close(mask, o['mask'], 1e-4, 'mask'); close(depth, o['depth'], 1e-4, 'depth')
    close(sem, o['semantics'], 1e-5, 'semantic map')
    draws = iter(tap.draws)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: next(draws).to(cam.device)
    try:
        with torch.no_grad():
            rgb0, depth0, mask0, _, none, _ = render(model, H, W, cam, focal, None, None, z, S)
    finally:
        torch.rand = real_rand
    assert none is None and torch.equal(rgb0, rgb) and torch.equal(depth0, depth) and torch.equal(mask0, mask)


def test_render_dropin_coords_and_force_no_cam_grad(setup):
    """compute_coords: the query points are composited in the semantics slot (run.py:337-338); force_no_cam_grad
    (run.py:211-214): same pixels, no gradient into the camera while the latent still gets one."""
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dcfg = {'scene_range': 0.55, 'white_background': False}
    render = nfi_render.make_render(cfg, dcfg)
    H, W, S = 12, 20, 32
    with torch.no_grad(), RandTap() as tap, OneRenderLaunch() as one:
        rgb, depth, mask, normals, coords_map, _ = render(model, H, W, cam, focal, None, None, z, S, compute_coords=True)
    assert one.calls == 1 and normals is None and coords_map.shape == (2, H, W, 3)
    with torch.no_grad():
        planes, att = model.planes_and_values(z)
        dec = model.decoder.net
        cpu = lambda t: t.detach().cpu()
        o = orc.render(cpu(planes), cpu(dec[0].weight), cpu(dec[0].bias), cpu(dec[2].weight), cpu(dec[2].bias), cpu(cam),
                       cpu(focal), H, W, S, 0.55, white_background=False, noise_coarse=tap.draws[0],
                       noise_fine=tap.draws[1], use_sdf=True, beta=cpu(model.beta), alpha=cpu(model.alpha),
                       attention_values=cpu(att), want_coords=True)
    close(rgb, o['rgb'], 1e-4, 'rgb'); close(mask, o['mask'], 1e-4, 'mask')
    close(coords_map, o['semantics'], 1e-5, 'composited coordinates')
    # gradients: with force_no_cam_grad the focal length and the camera's rotation get none, the latent does
    cam_g, focal_g, z_g = cam.clone().requires_grad_(), focal.clone().requires_grad_(), z.clone().requires_grad_()
    draws = iter(tap.draws)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: next(draws).to(cam.device)
    try:
        rgb2, _, mask2, _, _, _ = render(model, H, W, cam_g, focal_g, None, None, z_g, S, force_no_cam_grad=True)
    finally:
        torch.rand = real_rand
    close(rgb2, o['rgb'], 1e-4, 'rgb (force_no_cam_grad)')
    g_cam, g_focal, g_z = torch.autograd.grad(rgb2.sum() + mask2.sum(), [cam_g, focal_g, z_g], allow_unused=True)
    # (the focal length gets none; the camera's TRANSLATION does - the reference builds the fine pass's points from the
    #  undetached ray origins, run.py:286-288 - its rotation does not)
    assert g_focal is None or float(g_focal.abs().max()) == 0.0
    assert g_cam is not None and float(g_cam[:, :3, :3].abs().max()) == 0.0 and float(g_cam[:, :3, 3].abs().max()) > 0.0
    assert g_z is not None and torch.isfinite(g_z).all() and g_z.abs().sum() > 0
    # and without it the camera does get one
    draws = iter(tap.draws)
    torch.rand = lambda *a, **k: next(draws).to(cam.device)
    try:
        rgb3, _, mask3, _, _, _ = render(model, H, W, cam_g, focal_g, None, None, z_g, S)
    finally:
        torch.rand = real_rand
    g_cam, g_focal = torch.autograd.grad(rgb3.sum() + mask3.sum(), [cam_g, focal_g])
    assert g_cam.abs().sum() > 0 and g_focal.abs().sum() > 0


def test_nerf_utils_api(setup):
    model, cam, focal, z = setup
    H, W, S = 16, 16, 16
    with torch.no_grad():
        ro, rd = nu.get_ray_bundle(H, W, focal, cam, None)
        o_ro, o_rd = orc.ray_bundle(H, W, focal.cpu(), cam.cpu())
        close(ro, o_ro.expand_as(ro.cpu()), 0, 'ro'); close(rd, o_rd, 2e-7, 'rd')
        rdn = torch.nn.functional.normalize(rd, dim=-1)
        near, far = nu.compute_near_far_planes(ro, rdn, 0.55)
        o_near, o_far, _ = orc.near_far(ro.cpu(), rdn.cpu(), 0.55)
        close(near, o_near, 0, 'near'); close(far, o_far, 0, 'far')
        q, t = nu.compute_query_points_from_rays(ro, rdn, near, far, S, randomize=False)
        o_t = orc.stratified_depths(o_near, o_far, S, None)
        close(t, o_t, 0, 'depths'); close(q, orc.points_on_rays(ro.cpu(), rdn.cpu(), o_t), 0, 'points')
        sigma = torch.rand(2, H, W, S, device=cam.device) * 30
        rgbs = torch.rand(2, H, W, S, 3, device=cam.device) * 2 - 1
        w = nu.render_volume_density_weights_only(sigma, ro, rdn, t)
        close(w, orc.ray_weights(sigma.cpu(), rdn.cpu(), t.cpu()), 1e-6, 'weights')
        sem = torch.rand(2, H, W, S, 4, device=cam.device)
        rgb_map, depth_map, mask, nmap, smap = nu.render_volume_density(sigma, rgbs, ro, rdn, t, None, sem, True)
        o_rgb, o_dep, o_acc, o_sem, _ = orc.composite(sigma.cpu(), rgbs.cpu(), rdn.cpu(), t.cpu(), sem.cpu(), True)
        close(rgb_map, o_rgb, 1e-5, 'rgb map'); close(depth_map, o_dep, 1e-5, 'depth'); close(mask, o_acc, 1e-5, 'mask')
        close(smap, o_sem, 1e-5, 'semantic map'); assert nmap is None
        bins = torch.sort(torch.rand(50, 33, device=cam.device), dim=-1)[0]
        wts = torch.rand(50, 32, device=cam.device)
        s_det = nu.sample_pdf(bins, wts, 24, deterministic=True)
        ref, _, _ = orc.inverse_cdf(bins.cpu(), wts.cpu(), orc.deterministic_u(50, 24, wts.cpu()))
        close(s_det, ref, 1e-5, 'sample_pdf')


def test_gradient_request_without_a_backward_fails_loudly(setup):
    """An op without a HIP backward must raise when a gradient is asked for, never fall back to ATen."""
    model, cam, focal, z = setup
    sigma = (torch.rand(2, 4, 4, 16, device=cam.device) * 5).requires_grad_()
    ro, rd = nu.get_ray_bundle_normalized(4, 4, focal, cam, None)
    t = torch.sort(torch.rand(2, 4, 4, 16, device=cam.device), dim=-1)[0] + 0.5
    w = nu.render_volume_density_weights_only(sigma, ro, rd, t)          # run.py calls it under no_grad only
    with pytest.raises(NotImplementedError):
        w.sum().backward()
    # a view-direction decoder with 16-bit texels has no backward either
    import copy
    from nerf_from_image_amd import ops
    torch.manual_seed(5)
    vd = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32, use_viewdir=True).to(cam.device).eval()
    nfi_gen.attach(vd, texel_dtype=ops.TEXEL_BF16)
    cfg = types.SimpleNamespace(use_viewdir=True, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 0.55, 'white_background': True})
    zz = z.clone().requires_grad_()
    rgb, *_ = render(vd, 8, 8, cam, focal, None, None, zz, 16)
    with pytest.raises(NotImplementedError):
        rgb.sum().backward()


@pytest.mark.parametrize('tdt_name', ['bf16', 'fp16'])
def test_gradients_with_16_bit_texel_storage(setup, tdt_name):
    """BASELINE cfg2 'bf16' / cfg5 'fp16' plane storage in training: forward and gradients equal the oracle evaluated
    on the ROUNDED planes (arithmetic stays fp32; the gradient passes through the rounding unchanged)."""
    import copy
    from nerf_from_image_amd import ops
    model, cam, focal, z = setup
    tdt, tt = (ops.TEXEL_BF16, torch.bfloat16) if tdt_name == 'bf16' else (ops.TEXEL_F16, torch.float16)
    m2 = nfi_gen.attach(copy.deepcopy(model), texel_dtype=tdt)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dcfg = {'scene_range': 0.55, 'white_background': True}
    render = nfi_render.make_render(cfg, dcfg)
    H, W, S = 10, 12, 32
    g = torch.Generator().manual_seed(8)
    w_rgb = torch.randn(2, H, W, 3, generator=g)
    cam_g = cam.clone().requires_grad_()
    params = [m2.synthesis_network.basis, m2.decoder.net[0].weight, m2.decoder.net[2].weight, m2.beta, cam_g]
    with RandTap() as tap:
        rgb, _, mask, _, _, _ = render(m2, H, W, cam_g, focal, None, None, z, S)
    got = torch.autograd.grad((rgb * w_rgb.to(cam.device)).sum() + mask.sum(), params)
    # oracle in float64 on planes rounded to the storage type, straight-through gradient into the producer
    m64 = copy.deepcopy(model).cpu().double()
    planes, att = m64.planes_and_values(z.cpu().double())
    rounded = planes + (planes.detach().float().to(tt).double() - planes.detach())
    dec = m64.decoder.net
    cam64 = cam.detach().cpu().double().requires_grad_()
    o = orc.render(rounded, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, cam64, focal.cpu().double(), H, W, S,
                   0.55, white_background=True, noise_coarse=tap.draws[0].double(), noise_fine=tap.draws[1].double(),
                   use_sdf=True, beta=m64.beta, alpha=m64.alpha, attention_values=att)
    close(rgb, o['rgb'].float(), 2e-4, 'rgb on rounded planes')
    ref = torch.autograd.grad((o['rgb'] * w_rgb.double()).sum() + o['mask'].sum(),
                              [m64.synthesis_network.basis, dec[0].weight, dec[2].weight, m64.beta, cam64])
    for name, a, b in zip(('plane producer', 'w1', 'w2', 'beta', 'camera'), got, ref):
        scale = b.abs().max().item()
        assert (a.cpu().double() - b).abs().max().item() <= 2e-3 * scale, (name, (a.cpu().double() - b).abs().max().item(), scale)


def test_normals_sampler_and_render(setup):
    """`normals` = normalize(d sdf / d x) (generator.py:599-623) from the HIP coordinate-gradient path, per point
    and composited by render(compute_normals=True) (weights detached + white background, nerf_utils.py:146-159)."""
    model, cam, focal, z = setup
    dev = cam.device
    with torch.no_grad():
        sampler = model(None, z, ['sampler'])['sampler']
        planes, att = model.planes_and_values(z)
    g = torch.Generator().manual_seed(4)
    x = ((torch.rand(2, 300, 3, generator=g) * 2 - 1) * 0.5)
    res = sampler(x.to(dev), ['sigma', 'rgb', 'normals'])
    dec = model.decoder.net
    cpu = lambda t: t.detach().cpu()

    def oracle_normals(pts):
        p = pts.clone().requires_grad_()
        q = orc.field_query(cpu(planes), cpu(dec[0].weight), cpu(dec[0].bias), cpu(dec[2].weight), cpu(dec[2].bias), p,
                            0.55, True, cpu(model.beta), cpu(model.alpha), cpu(att))
        gx, = torch.autograd.grad(q['sdf'].sum(), p)
        return torch.nn.functional.normalize(gx, dim=-1), q
    n_ref, q_ref = oracle_normals(x)
    assert res['normals'].shape == (2, 300, 3)
    close(res['normals'], n_ref, 2e-3, 'normals')              # unit vectors; fp32 finite differences of texels
    close(res['rgb'], q_ref['rgb'], 1e-4, 'rgb alongside normals')
    assert not res['sigma'].requires_grad

    # ---- composited normal map
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dcfg = {'scene_range': 0.55, 'white_background': True}
    render = nfi_render.make_render(cfg, dcfg)
    H, W, S = 16, 16, 32
    with torch.no_grad(), RandTap() as tap, OneRenderLaunch() as one:
        rgb, depth, mask, normal_map, sem, _ = render(model, H, W, cam, focal, None, None, z, S, compute_normals=True)
    assert one.calls == 1 and sem is None and normal_map.shape == (2, H, W, 3)
    o = oracle_for(model, z, cam, focal, H, W, S, cfg, dcfg, tap.draws)
    close(rgb, o['rgb'], 1e-4, 'rgb')
    n_c, _ = oracle_normals(orc.points_on_rays(o['ro'], o['rd'], o['t_coarse']).reshape(2, -1, 3))
    n_f, _ = oracle_normals(orc.points_on_rays(o['ro'], o['rd'], o['t_fine']).reshape(2, -1, 3))
    n_all = torch.cat((n_c.view(2, H, W, S, 3), n_f.view(2, H, W, S, 3)), dim=-2)
    n_sorted = n_all.gather(-2, o['perm'].unsqueeze(-1).expand(-1, -1, -1, -1, 3))
    ref_map = (o['weights'][..., None] * n_sorted).sum(dim=-2) + (1. - o['mask'][..., None])
    close(normal_map, ref_map, 3e-3, 'normal map')
    # the first eval batch asks for normals AND semantics (run.py:2036-2051): still one launch, same images
    draws = iter(tap.draws)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: next(draws).to(cam.device)
    try:
        with torch.no_grad(), OneRenderLaunch() as one:
            rgb2, depth2, mask2, nmap2, sem2, _ = render(model, H, W, cam, focal, None, None, z, S, compute_normals=True,
                                                         compute_semantics=True)
    finally:
        torch.rand = real_rand
    assert one.calls == 1 and torch.equal(rgb2, rgb) and torch.equal(mask2, mask) and torch.equal(nmap2, normal_map)
    close(sem2.sum(-1), mask2, 1e-5, 'semantic map sums to the mask')
    # and the staged path (what a call with a gradient still takes) gives the same normal map
    from nerf_from_image_amd import ops as _ops
    real_fwd = _ops.render_fwd
    draws = iter(tap.draws)
    torch.rand = lambda *a, **k: next(draws).to(cam.device)
    try:
        zg = z.clone().requires_grad_()
        _, _, _, nmap_staged, _, _ = render(model, H, W, cam, focal, None, None, zg, S, compute_normals=True)
    finally:
        torch.rand = real_rand
        _ops.render_fwd = real_fwd
    close(nmap_staged, normal_map, 2e-3, 'staged vs fused normal map')


def test_parallel_model_dispatch(setup):
    """ParallelModel.forward (run.py:569-617): EMA selection, resolution multiplier, closure hand-off."""
    from nerf_from_image_amd.parallel import ParallelModel
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 0.55, 'white_background': True})
    pm = ParallelModel(8, model=None, model_ema=model, render=render, depth_samples_per_ray=16)
    with torch.no_grad():
        out = pm(cam, focal, None, None, z, use_ema=True, res_multiplier=2)
        assert out[0].shape == (2, 16, 16, 3) and out[2].shape == (2, 16, 16)
        seen = {}

        def closure(self_, rgb, alpha, sem, extra, weight):
            seen['args'] = (self_ is pm, tuple(rgb.shape), tuple(alpha.shape), sem, extra, weight)
            return rgb.mean() * weight
        loss = pm(cam, focal, None, None, z, use_ema=True, closure=closure, closure_params={'weight': 2.0})
    assert seen['args'] == (True, (2, 8, 8, 3), (2, 8, 8), None, {}, 2.0) and loss.dim() == 0


def _viewdir_oracle(model, z, cam, focal, H, W, S, draws, white, double=False):
    """Oracle render of a --use_viewdir stand-in model (ray feature from the model's own mapper on the oracle's
    normalised ray directions, run.py:216-219)."""
    conv = (lambda t: t.cpu().double()) if double else (lambda t: t.detach().cpu())   # double: keep the graph
    planes, att = model.planes_and_values(z)
    dec = model.decoder.net
    ro, rd = orc.ray_bundle(H, W, conv(focal), conv(cam))
    rd = orc.unit_dirs(rd)
    import copy
    mapper = model.viewdir_mapper if double else copy.deepcopy(model.viewdir_mapper).cpu()
    x = mapper(rd.unsqueeze(-2)).reshape(cam.shape[0], H * W, 32)
    return orc.render(conv(planes), conv(dec[0].weight), conv(dec[0].bias), conv(dec[2].weight), conv(dec[2].bias),
                      conv(cam), conv(focal), H, W, S, 0.55, white_background=white, fine_sampling=len(draws) > 1,
                      noise_coarse=draws[0] if not double else draws[0].double(),
                      noise_fine=None if len(draws) < 2 else (draws[1] if not double else draws[1].double()), use_sdf=True,
                      beta=conv(model.beta),
                      alpha=conv(model.alpha), attention_values=conv(att),
                      viewdir=dict(x=x, w3=conv(mapper.output.weight), b3=conv(mapper.output.bias)))


def test_render_with_view_directions(gpu_device):
    """args.use_viewdir (carla): render() computes the rays first, hands their directions to the model
    (run.py:216-222), and the per-ray mapper feature + its output layer run inside the HIP kernels; fused
    inference path and the differentiable path (fused render + stash as one node) against the oracle."""
    torch.manual_seed(77)
    model = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=48, use_viewdir=True).to(gpu_device).eval()
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(19)
    B, H, W, S = 2, 16, 20, 32
    cam = look_at_cameras(B, 1.6, g).to(gpu_device)
    focal = torch.full((B,), 1.0254, device=gpu_device)
    z = torch.randn(B, 512, generator=g).to(gpu_device)
    cfg = types.SimpleNamespace(use_viewdir=True, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 0.55, 'white_background': False})
    # gradients to the mapper, its output layer and the camera.  (The comparison is against a
    # float64 oracle: a sample within fp32 rounding of a texel boundary has a different bilinear slope there - about
    # one such sample per ~1e5 is expected - so the noise draws of this test are pinned by the seed above.)
    params = [model.viewdir_mapper.fc0.weight, model.viewdir_mapper.fc6.bias, model.viewdir_mapper.output.weight,
              model.viewdir_mapper.output.bias, model.decoder.net[2].weight]
    cam_g = cam.clone().requires_grad_()
    w_rgb = torch.randn(B, H, W, 3, generator=g).to(gpu_device)
    with RandTap() as tap, OneRenderLaunch() as one:       # the view-direction decoder with a gradient: the one-node path too
        rgb2, _, mask2, _, _, _ = render(model, H, W, cam_g, focal, None, None, z, S)
        got = torch.autograd.grad((rgb2 * w_rgb).sum() + mask2.sum(), params + [cam_g])
    assert one.calls == 1
    import copy
    m64 = copy.deepcopy(model).cpu().double()
    cam64 = cam.detach().cpu().double().requires_grad_()
    o2 = _viewdir_oracle(m64, z.cpu().double(), cam64, focal, H, W, S, tap.draws, False, double=True)
    p64 = [m64.viewdir_mapper.fc0.weight, m64.viewdir_mapper.fc6.bias, m64.viewdir_mapper.output.weight,
           m64.viewdir_mapper.output.bias, m64.decoder.net[2].weight]
    ref = torch.autograd.grad((o2['rgb'] * w_rgb.cpu().double()).sum() + o2['mask'].sum(), p64 + [cam64])
    for name, a, b in zip(['fc0.weight', 'fc6.bias', 'output.weight', 'output.bias', 'decoder w2', 'camera'], got, ref):
        scale = b.abs().max().item()
        assert (a.cpu().double() - b).abs().max().item() <= 2e-3 * scale, (name, (a.cpu().double() - b).abs().max().item(), scale)

    # fused inference path
    with torch.no_grad(), RandTap() as tap:
        rgb, depth, mask, _, _, _ = render(model, H, W, cam, focal, None, None, z, S)
    with torch.no_grad():
        o = _viewdir_oracle(model, z, cam, focal, H, W, S, tap.draws, False)
    close(rgb, o['rgb'], 1e-4, 'fused rgb'); close(mask, o['mask'], 1e-4, 'fused mask'); close(depth, o['depth'], 1e-4, 'depth')
    assert o['mask'].mean() > 0.05


def test_single_pass_kernel_other_instantiations(setup, gpu_device):
    """render_fwd_long_kernel (one pass of 129..512 samples) with 16-bit texel storage - against the stage kernels on the
    SAME storage (an extra map over such a pass keeps the staged path) - and with the view-direction decoder, inference
    and gradient, against the oracle."""
    import copy
    from nerf_from_image_amd import ops
    model, cam, focal, z = setup
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=False)
    dcfg = {'scene_range': 0.55, 'white_background': True}
    H, W, S = 12, 16, 200
    for tdt in (ops.TEXEL_F16, ops.TEXEL_BF16):
        m2 = nfi_gen.attach(copy.deepcopy(model), texel_dtype=tdt)
        render = nfi_render.make_render(cfg, dcfg)
        with torch.no_grad():
            torch.manual_seed(3)
            with OneRenderLaunch() as one:
                a = render(m2, H, W, cam, focal, None, None, z, S)
            assert one.calls == 1
            torch.manual_seed(3)
            b = render(m2, H, W, cam, focal, None, None, z, S, compute_coords=True)       # staged
        close(a[0], b[0], 2e-5, 'rgb, 16-bit texels'); close(a[1], b[1], 2e-5, 'depth'); close(a[2], b[2], 2e-5, 'mask')
    # view-direction decoder
    torch.manual_seed(78)
    vd = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=48, use_viewdir=True).to(gpu_device).eval()
    nfi_gen.attach(vd)
    cfg_v = types.SimpleNamespace(use_viewdir=True, use_sdf=True, attention_values=10, fine_sampling=False)
    render = nfi_render.make_render(cfg_v, {'scene_range': 0.55, 'white_background': False})
    S = 160
    with torch.no_grad(), RandTap() as tap, OneRenderLaunch() as one:
        rgb, depth, mask, _, _, _ = render(vd, H, W, cam, focal, None, None, z, S)
    assert one.calls == 1 and len(tap.draws) == 1
    with torch.no_grad():
        o = _viewdir_oracle(vd, z, cam, focal, H, W, S, tap.draws, False)
    close(rgb, o['rgb'], 1e-4, 'viewdir rgb'); close(mask, o['mask'], 1e-4, 'viewdir mask'); close(depth, o['depth'], 1e-4, 'depth')
    w_rgb = torch.randn(2, H, W, 3, device=gpu_device)
    params = [vd.viewdir_mapper.output.weight, vd.decoder.net[2].weight]
    with RandTap() as tap, OneRenderLaunch() as one:
        rgb2, _, mask2, _, _, _ = render(vd, H, W, cam, focal, None, None, z, S)
        got = torch.autograd.grad((rgb2 * w_rgb).sum() + mask2.sum(), params)
    assert one.calls == 1
    m64 = copy.deepcopy(vd).cpu().double()
    o2 = _viewdir_oracle(m64, z.cpu().double(), cam.cpu().double(), focal, H, W, S, tap.draws, False, double=True)
    ref = torch.autograd.grad((o2['rgb'] * w_rgb.cpu().double()).sum() + o2['mask'].sum(),
                              [m64.viewdir_mapper.output.weight, m64.decoder.net[2].weight])
    for name, a_, b_ in zip(['output.weight', 'decoder w2'], got, ref):
        scale = b_.abs().max().item()
        assert (a_.cpu().double() - b_).abs().max().item() <= 2e-3 * scale, (name, (a_.cpu().double() - b_).abs().max().item(), scale)


def test_bbox_overlay(setup):
    """'bbox' in the model request + 'coords' in the sampler request: sigma gets +100 on the cube's wire frame."""
    model, cam, focal, z = setup
    g = torch.Generator().manual_seed(4)
    x = ((torch.rand(2, 5, 5, 33, 3, generator=g) * 2 - 1) * 0.6).to(cam.device)
    with torch.no_grad():
        plain = model(None, z, ['sampler'])['sampler'](x, ['sigma'])['sigma']
        res = model(None, z, ['sampler', 'bbox'])['sampler'](x, ['sigma', 'coords'])
    outside = ((x.view(2, -1, 3) / 0.55).abs() > 1).any(dim=-1).float()
    ref = orc.bbox_overlay(x.cpu(), plain.cpu(), outside.cpu(), 0.55)
    assert torch.equal(res['coords'], x)
    assert torch.equal(res['sigma'].cpu(), ref)
    assert (ref != plain.cpu()).float().mean() > 0.01


def test_regulariser_outputs(gpu_device):
    """Generator.forward's regulariser branch (generator.py:505-585) on the HIP path of a bare container: eikonal
    (d sdf/dx as an operator output, its backward = the reference's double backward), distance, total variation and
    entropy terms, and their gradients w.r.t. the plane producer and the decoder, against float64 autograd of the
    oracle with the same two random draws."""
    import copy
    torch.manual_seed(3)
    model = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32).to(gpu_device).train()
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2, 512, generator=g).to(gpu_device)
    names = ['sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss']
    draws = {}
    real_rand, real_randn_like = torch.rand, torch.randn_like

    def rand(*a, **k):
        # the stratified jitter is the one uniform draw of the branch (generator.sample_volume_stratified).
        # plane_res 32 and 31 strata: the texel coordinate of a point is (cell index + jitter), so keeping the jitter
        # off 0 and 1 keeps every point off the texel boundaries, where d sdf/dx jumps and an fp32 kernel and a
        # float64 oracle may legitimately pick different cells
        draws['jitter'] = real_rand(*a, **k).clamp_(1e-3, 1 - 1e-3)
        return draws['jitter']

    def randn_like(t, **k):
        draws['perturb'] = real_randn_like(t, **k)
        return draws['perturb']
    torch.rand, torch.randn_like = rand, randn_like
    try:
        out = model(None, z, names)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    assert set(out) == set(names) and tuple(draws['jitter'].shape) == (2, 31, 31, 31, 3)
    dec = model.decoder.net
    params = [model.synthesis_network.basis, model.synthesis_network.proj.weight, dec[0].weight, dec[0].bias,
              dec[2].weight, dec[2].bias, model.beta]
    weights = [1.0, 0.7, 3.0, 0.01]
    got = torch.autograd.grad(sum(w * out[n].sum() for w, n in zip(weights, names)), params)

    m64 = copy.deepcopy(model).cpu().double()
    planes64, _ = m64.planes_and_values(z.cpu().double())
    d64 = m64.decoder.net
    p64 = [m64.synthesis_network.basis, m64.synthesis_network.proj.weight, d64[0].weight, d64[0].bias, d64[2].weight,
           d64[2].bias, m64.beta]
    bins = orc.stratified_volume(2, 32, 0.55, draws['jitter'].cpu().double())
    ref = orc.regularisers(planes64, *p64[2:6], bins, 0.55, True, m64.beta, draws['perturb'].cpu().double())
    for n in names:
        close(out[n], ref[n], 2e-4 * float(ref[n].detach().abs().max()) + 1e-6, n)
    ref_g = torch.autograd.grad(sum(w * ref[n].sum() for w, n in zip(weights, names)), p64)
    for name, a, b in zip(['basis', 'proj', 'w1', 'b1', 'w2', 'b2', 'beta'], got, ref_g):
        scale = b.abs().max().item()
        assert (a.cpu().double() - b).abs().max().item() <= 2e-3 * scale + 1e-9, (name, (a.cpu().double() - b).abs().max().item(), scale)


def test_regulariser_outputs_with_view_direction_decoder(gpu_device):
    """--use_viewdir models (33-output decoder): all four regulariser terms (the total-variation one used to fail on
    the decoder shape) against the float64 oracle, which like the reference uses output row 0 only."""
    import copy
    torch.manual_seed(3)
    model = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32, use_viewdir=True).to(gpu_device).train()
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2, 512, generator=g).to(gpu_device)
    names = ['sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss']
    draws = {}
    real_rand, real_randn_like = torch.rand, torch.randn_like

    def rand(*a, **k):
        draws['jitter'] = real_rand(*a, **k).clamp_(1e-3, 1 - 1e-3)
        return draws['jitter']

    def randn_like(t, **k):
        draws['perturb'] = real_randn_like(t, **k)
        return draws['perturb']
    torch.rand, torch.randn_like = rand, randn_like
    try:
        out = model(None, z, names)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    assert set(out) == set(names)
    dec = model.decoder.net
    assert dec[2].weight.shape == (33, 64)
    params = [model.synthesis_network.basis, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, model.beta]
    weights = [1.0, 0.7, 3.0, 0.01]
    got = torch.autograd.grad(sum(w * out[n].sum() for w, n in zip(weights, names)), params)
    m64 = copy.deepcopy(model).cpu().double()
    planes64, _ = m64.planes_and_values(z.cpu().double())
    d64 = m64.decoder.net
    p64 = [m64.synthesis_network.basis, d64[0].weight, d64[0].bias, d64[2].weight, d64[2].bias, m64.beta]
    bins = orc.stratified_volume(2, 32, 0.55, draws['jitter'].cpu().double())
    ref = orc.regularisers(planes64, *p64[1:5], bins, 0.55, True, m64.beta, draws['perturb'].cpu().double())
    for n in names:
        close(out[n], ref[n], 2e-4 * float(ref[n].detach().abs().max()) + 1e-6, n)
    ref_g = torch.autograd.grad(sum(w * ref[n].sum() for w, n in zip(weights, names)), p64)
    for name, a, b in zip(['basis', 'w1', 'b1', 'w2', 'b2', 'beta'], got, ref_g):
        scale = b.abs().max().item()
        assert (a.cpu().double() - b).abs().max().item() <= 2e-3 * scale + 1e-9, (name, (a.cpu().double() - b).abs().max().item(), scale)
    assert got[3][1:].abs().max().item() == 0.0          # only the distance row of the 33-output layer is involved


def test_wrapped_module_with_hip_regularisers(gpu_device):
    """A module that has its own forward (like the reference Generator) attached with hip_regularisers=True gives the
    same regulariser losses as the bare-container path (same parameters, same random draws)."""
    class WithForward(StandInGenerator):
        def forward(self, viewdir, c, request_model_outputs=['sampler'], model_inputs={}):
            ws = self.mapping_network(c)
            self.synthesis_network(ws[:, :14])
            out = {}
            if 'attention_values' in request_model_outputs:
                out['attention_values'] = self.texture_mapper(ws[:, 14])
            if 'sampler' in request_model_outputs:
                out['sampler'] = None
            return out
    torch.manual_seed(11)
    wrapped = WithForward(0.55, attention_values=10, use_sdf=True, plane_res=32).to(gpu_device).train()
    torch.manual_seed(11)
    bare = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32).to(gpu_device).train()
    bare.load_state_dict(wrapped.state_dict())
    nfi_gen.attach(wrapped, hip_regularisers=True)
    nfi_gen.attach(bare)
    z = torch.randn(2, 512, device=gpu_device)
    names = ['sdf_eikonal_loss', 'sdf_distance_loss', 'entropy_loss']
    torch.manual_seed(99)
    a = wrapped(None, z, names)
    torch.manual_seed(99)
    b = bare(None, z, names)
    assert set(a) == set(names)
    for n in names:
        assert torch.equal(a[n], b[n]), n
    ga = torch.autograd.grad(sum(a[n].sum() for n in names), wrapped.decoder.net[0].weight)[0]
    gb = torch.autograd.grad(sum(b[n].sum() for n in names), bare.decoder.net[0].weight)[0]
    close(ga, gb, 1e-5 * float(gb.abs().max()) + 1e-9, 'w1 gradient')
