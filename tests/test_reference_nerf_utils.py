"""nerf_from_image_amd.nerf_utils, function by function, against the REAL lib/nerf_utils.py module (oracle/_ref or the
checkout): the same names and parameters (CPU test), and on MI355X the same values and gradients on the same tensors - the
reference's TorchScript functions on PyTorch-ROCm on one side, the HIP kernels behind the C ABI on the other.

Tolerances are fp32 evaluation-order differences (FMA contraction, reduction order), stated per check."""
import inspect

import pytest
import torch

from oracle import reference

FUNCTIONS = ('cumprod_exclusive', 'get_ray_bundle', 'compute_query_points_from_rays', 'render_volume_density',
             'render_volume_density_weights_only', 'sample_pdf', 'compute_near_far_planes')


def _require_reference():
    if not reference.available():
        pytest.skip('reference sources not staged: run oracle/make_ref.py (or __graft_entry__.build()) where /root/reference exists')


def test_every_function_of_the_reference_module_exists_with_its_parameters():
    """A caller written against lib/nerf_utils.py (positional or keyword) binds unchanged: every public function is there,
    its parameters come first, in the reference's order, under the reference's names and with the reference's defaults;
    what the drop-in adds (strict=, noise=) follows them and has a default."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.nerf_utils_unscripted()
    public = sorted(n for n, f in vars(ref).items() if inspect.isfunction(f) and f.__module__ == ref.__name__ and not n.startswith('_'))
    assert public == sorted(FUNCTIONS), public            # (the list above is the whole module)
    for name in FUNCTIONS:
        theirs = list(inspect.signature(getattr(ref, name)).parameters.values())
        ours = list(inspect.signature(getattr(nu, name)).parameters.values())
        assert len(ours) >= len(theirs), name
        for a, b in zip(theirs, ours):
            assert (a.name, a.kind) == (b.name, b.kind), (name, a, b)
            assert a.default == b.default or (a.default is inspect.Parameter.empty) == (b.default is inspect.Parameter.empty), (name, a, b)
            if a.default is not inspect.Parameter.empty:
                assert a.default == b.default, (name, a, b)
        for extra in ours[len(theirs):]:
            assert extra.default is not inspect.Parameter.empty, (name, extra)


# ------------------------------------------------------------------------------------------------------- on the GPU
def _close(a, b, tol, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, tol * scale)


def _leaf(t):
    return t.detach().clone().requires_grad_(True)


def _cameras(batch, dev, seed):
    import reference_cases as rc
    cpu = torch.Generator().manual_seed(seed)
    return rc.cameras(batch, 1.3, cpu).to(dev), (1.1 + 0.2 * torch.rand(batch, generator=cpu)).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['perspective', 'perspective_bbox_center', 'orthographic_bbox'])
def test_get_ray_bundle(gpu_device, model):
    """lib/nerf_utils.py:28-91, its three branches: pinhole, pinhole with crop box and principal point, orthographic
    (focal_length=None) with crop box.  Values and the gradients to the camera matrix and the focal length."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.modules().nerf_utils
    B, H, W = 3, 48, 40
    cam, focal = _cameras(B, gpu_device, 11)
    g = torch.Generator(device=gpu_device).manual_seed(5)
    bbox = center = None
    if model != 'perspective':
        import reference_cases as rc
        bbox = rc.crop_boxes(B, torch.Generator().manual_seed(6)).to(gpu_device)           # [B,2,2], data/datasets.py:318-340
    if model == 'perspective_bbox_center':
        center = (torch.rand((B, 2), device=gpu_device, generator=g) - 0.5) * 0.2
    if model == 'orthographic_bbox':
        focal = None
    cot = torch.randn((2, B, H, W, 3), device=gpu_device, generator=g)
    out = []
    for fn in (ref.get_ray_bundle, nu.get_ray_bundle):
        c = _leaf(cam)
        f = None if focal is None else _leaf(focal)
        ro, rd = fn(H, W, f, c, bbox, center)
        ((ro * cot[0]).sum() + (rd * cot[1]).sum()).backward()
        out.append((ro.detach(), rd.detach(), c.grad, None if f is None else f.grad))
    (ro_r, rd_r, gc_r, gf_r), (ro, rd, gc, gf) = out
    _close(ro, ro_r, 1e-6, 'ray origins')
    _close(rd, rd_r, 2e-6, 'ray directions')
    _close(gc, gc_r, 2e-5, 'd/d tform_cam2world')               # sums over H*W rays: reduction order
    if gf_r is not None:
        _close(gf, gf_r, 2e-5, 'd/d focal_length')


def _rays(dev, seed, B=2, R=40, scene_range=1.0):
    import nerf_from_image_amd.nerf_utils as nu
    cam, focal = _cameras(B, dev, seed)
    with torch.no_grad():
        ro, rd = nu.get_ray_bundle(R, R, focal, cam, None)
        rd = torch.nn.functional.normalize(rd, dim=-1)
    return ro, rd


@pytest.mark.gpu
@pytest.mark.parametrize('scene_range', [1.0, 1.4, 0.35])
def test_compute_near_far_planes(gpu_device, scene_range):
    """lib/nerf_utils.py:225-273 incl. the batch-wide fill of the rays that miss the cube (0.35: most of them do), the 0.1
    clamp and the 1e-3 minimum extent."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.modules().nerf_utils
    ro, rd = _rays(gpu_device, 21)
    near_r, far_r = ref.compute_near_far_planes(ro, rd, scene_range)
    near, far = nu.compute_near_far_planes(ro, rd, scene_range)
    _close(near, near_r, 2e-6, 'near')
    _close(far, far_r, 2e-6, 'far')
    assert not near.requires_grad and not far.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize('randomize', [False, True])
def test_compute_query_points_from_rays(gpu_device, randomize):
    """lib/nerf_utils.py:94-120: the stratified depths (the jitter drawn from the same Philox stream - same shape, same
    seed), the points, and the gradients of the points to the rays."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.modules().nerf_utils
    ro, rd = _rays(gpu_device, 31)
    near, far = nu.compute_near_far_planes(ro, rd, 1.0)
    S = 48
    cot = torch.randn((*ro.shape[:-1], S, 3), device=gpu_device, generator=torch.Generator(device=gpu_device).manual_seed(3))
    out = []
    for fn in (ref.compute_query_points_from_rays, nu.compute_query_points_from_rays):
        a, b = _leaf(ro), _leaf(rd)
        torch.manual_seed(77)
        pts, depth = fn(a, b, near, far, S, randomize)
        (pts * cot).sum().backward()
        out.append((pts.detach(), depth.detach(), a.grad, b.grad))
    (p_r, d_r, ga_r, gb_r), (p, d, ga, gb) = out
    _close(d, d_r, 1e-6, 'depth values')
    _close(p, p_r, 1e-6, 'query points')
    _close(ga, ga_r, 1e-5, 'd/d ray_origins')
    _close(gb, gb_r, 1e-5, 'd/d ray_directions')
    assert (d[..., 1:] >= d[..., :-1]).all()


def _samples(dev, seed, S=64, C=3, K=10):
    """Depths along real rays, a density with a surface in it, colours, unit normals, softmax semantics."""
    import nerf_from_image_amd.nerf_utils as nu
    ro, rd = _rays(dev, seed)
    near, far = nu.compute_near_far_planes(ro, rd, 1.0)
    torch.manual_seed(seed)
    with torch.no_grad():
        pts, depth = nu.compute_query_points_from_rays(ro, rd, near, far, S, True)
    g = torch.Generator(device=dev).manual_seed(seed)
    sigma = torch.nn.functional.softplus(8.0 * (0.6 - pts.norm(dim=-1))) * 6.0 + 0.05 * torch.rand(depth.shape, device=dev, generator=g)
    rgb = torch.rand((*depth.shape, C), device=dev, generator=g)
    normals = torch.nn.functional.normalize(torch.randn((*depth.shape, 3), device=dev, generator=g), dim=-1)
    sem = torch.softmax(torch.randn((*depth.shape, K), device=dev, generator=g), dim=-1)
    return ro, rd * 1.0, depth, sigma, rgb, normals, sem


@pytest.mark.gpu
def test_render_volume_density_weights_only_and_cumprod(gpu_device):
    """lib/nerf_utils.py:20-25, 164-180."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.modules().nerf_utils
    ro, rd, depth, sigma, *_ = _samples(gpu_device, 41)
    _close(nu.render_volume_density_weights_only(sigma, ro, rd, depth),
           ref.render_volume_density_weights_only(sigma, ro, rd, depth), 2e-6, 'weights')
    x = torch.rand((5, 7, 33), device=gpu_device) + 0.5
    _close(nu.cumprod_exclusive(x), ref.cumprod_exclusive(x), 1e-6, 'cumprod_exclusive')
    # non-unit directions scale the intervals (the norm factor of lib/nerf_utils.py:176)
    _close(nu.render_volume_density_weights_only(sigma, ro, rd * 1.7, depth),
           ref.render_volume_density_weights_only(sigma, ro, rd * 1.7, depth), 2e-6, 'weights, |d| = 1.7')


@pytest.mark.gpu
@pytest.mark.parametrize('white_background', [True, False])
@pytest.mark.parametrize('maps', ['rgb', 'normals', 'semantics', 'normals+semantics'])
def test_render_volume_density(gpu_device, white_background, maps):
    """lib/nerf_utils.py:123-161, every output and the gradients to sigma, rgb, semantics and the ray directions (the
    normal map is composited with detached weights and gets the white background, the semantic map neither)."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.modules().nerf_utils
    ro, rd, depth, sigma, rgb, normals, sem = _samples(gpu_device, 51)
    rd = rd * 1.3
    use_n, use_s = 'normals' in maps, 'semantics' in maps
    g = torch.Generator(device=gpu_device).manual_seed(9)
    shp = depth.shape[:-1]
    cot = {k: torch.randn((*shp, n), device=gpu_device, generator=g) for k, n in (('rgb', 3), ('mask', 1), ('normal', 3), ('sem', 10))}
    out = []
    for fn in (ref.render_volume_density, nu.render_volume_density):
        s, c, d = _leaf(sigma), _leaf(rgb), _leaf(rd)
        n = _leaf(normals) if use_n else None
        e = _leaf(sem) if use_s else None
        rgb_map, depth_map, mask, normal_map, sem_map = fn(s, c, ro, d, depth, n, e, white_background)
        assert (normal_map is None) == (not use_n) and (sem_map is None) == (not use_s)
        loss = (rgb_map * cot['rgb']).sum() + (mask * cot['mask'][..., 0]).sum()
        if use_n:
            loss = loss + (normal_map * cot['normal']).sum()
        if use_s:
            loss = loss + (sem_map * cot['sem']).sum()
        loss.backward()
        assert not depth_map.requires_grad
        out.append(dict(rgb=rgb_map.detach(), depth=depth_map.detach(), mask=mask.detach(),
                        normal=None if normal_map is None else normal_map.detach(), sem=None if sem_map is None else sem_map.detach(),
                        g_sigma=s.grad, g_rgb=c.grad, g_rd=d.grad, g_normals=None if n is None else n.grad,
                        g_sem=None if e is None else e.grad))
    r, h = out
    for k in ('rgb', 'depth', 'mask', 'normal', 'sem'):
        if r[k] is not None:
            _close(h[k], r[k], 3e-6, k + ' map')
    for k in ('g_sigma', 'g_rgb', 'g_rd', 'g_normals', 'g_sem'):
        if r[k] is not None:
            assert h[k] is not None, k
            _close(h[k], r[k], 2e-5, k)


@pytest.mark.gpu
@pytest.mark.parametrize('surface', ['translucent', 'opaque'])
@pytest.mark.parametrize('deterministic', [True, False])
def test_sample_pdf(gpu_device, deterministic, surface):
    """lib/nerf_utils.py:183-222 the way run.py:262-270 calls it: bins = interval mid-points, weights without the first
    and last sample; u from the same Philox stream, or the linspace.

    The inverse CDF amplifies the rounding of the cumulative sum (ATen's parallel scan and the kernel's associate
    differently: a few 1e-7) by width / pdf of the bin the sample lands in, so the tolerance is stated per sample as
    1e-5 + width * 6e-7 / pdf.  Bins whose mass sits at the reference's own `denom < 1e-5` switch (lib/nerf_utils.py:217-218:
    the empty bins behind an opaque surface, pdf = 1e-5 / sum) flip between t = 0 and t in [0, 1] with that rounding - in the
    reference against itself as well, CPU vs GPU - so a sample there is held to one bin width.  Such samples carry no weight
    in the composite."""
    _require_reference()
    import nerf_from_image_amd.nerf_utils as nu
    ref = reference.modules().nerf_utils
    ro, rd, depth, sigma, *_ = _samples(gpu_device, 61)
    if surface == 'translucent':
        sigma = sigma.clamp(max=3.0) + 0.2
    w = nu.render_volume_density_weights_only(sigma, ro, rd, depth).reshape(-1, depth.shape[-1])
    z = depth.reshape(-1, depth.shape[-1])
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    wm = w[:, 1:-1].contiguous()
    torch.manual_seed(123)
    theirs = ref.sample_pdf(bins, wm, 64, deterministic)
    torch.manual_seed(123)
    ours = nu.sample_pdf(bins, wm, 64, deterministic)
    assert ours.shape == theirs.shape
    pdf = (wm + 1e-5) / (wm + 1e-5).sum(dim=-1, keepdim=True)
    idx = (torch.searchsorted(bins.contiguous(), theirs.contiguous(), right=True) - 1).clamp(0, pdf.shape[-1] - 1)
    mass = torch.gather(pdf, 1, idx)
    width = (bins[:, 1:] - bins[:, :-1]).max(dim=-1, keepdim=True).values
    at_switch = torch.gather(torch.nn.functional.pad((pdf < 1.1e-5).float(), (1, 1)).unfold(1, 3, 1).amax(dim=-1), 1, idx) > 0
    tol = torch.where(at_switch, width * 1.001, 1e-5 + width * 6e-7 / mass)
    err = (ours - theirs).abs()
    assert bool((err <= tol).all()), (float((err - tol).max()), int((err > tol).sum()))
    if surface == 'translucent':
        assert not bool(at_switch.any()) and float(err.max()) <= 2e-4, float(err.max())
    else:
        assert float(at_switch.float().mean()) < 0.2          # (the check above is not vacuous)
    assert float(err.median()) <= 1e-6


def _same_leading_parameters(theirs, ours, what):
    theirs, ours = list(theirs.parameters.values()), list(ours.parameters.values())
    assert len(ours) >= len(theirs), what
    for a, b in zip(theirs, ours):
        assert (a.name, a.kind) == (b.name, b.kind), (what, a, b)
        assert (a.default is inspect.Parameter.empty) == (b.default is inspect.Parameter.empty), (what, a, b)
        if a.default is not inspect.Parameter.empty:
            assert a.default == b.default, (what, a, b)
    for extra in ours[len(theirs):]:
        assert extra.default is not inspect.Parameter.empty or extra.kind is inspect.Parameter.VAR_KEYWORD, (what, extra)


def test_render_and_generator_forward_take_the_reference_parameters():
    """run.py::render (run.py:176-189) and Generator.forward (models/generator.py:407-412): the drop-in's render() and the
    forward that attach() installs bind the same positional and keyword arguments, defaults included."""
    _require_reference()
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    ren, _ = reference.load_render(reference.render_args(), {'scene_range': 1.0, 'white_background': True})
    _same_leading_parameters(inspect.signature(ren), inspect.signature(nfi_render.render), 'render')
    bound = nfi_render.make_render(reference.render_args(), {'scene_range': 1.0, 'white_background': True})
    _same_leading_parameters(inspect.signature(ren), inspect.signature(bound), 'make_render(...)')
    fwd = inspect.signature(reference.modules().generator.Generator.forward)
    _same_leading_parameters(fwd, inspect.signature(nfi_gen.hip_forward), 'Generator.forward (hip)')
    _same_leading_parameters(fwd, inspect.signature(nfi_gen.wrapped_forward), 'Generator.forward (wrapped)')
