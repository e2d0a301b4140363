"""Backward kernels against autograd of the oracle (the reference gets these gradients from
PyTorch autograd).  The oracle is differentiated in float64 on CPU; the
HIP gradients (fp32) must agree to 1e-4 of the gradient's scale."""
import pytest
import torch

from conftest import load_golden
from nerf_from_image_amd import nerf_utils as nu
from oracle import nfi_oracle as orc

pytestmark = pytest.mark.gpu


def rel_close(got, ref, what, tol=2e-4):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().clamp_min(1e-12)
    err = (got - ref).abs().max() / scale
    assert torch.isfinite(got).all() and err <= tol, (what, float(err), float(scale))


def oracle_merge_composite(rd, t_a, s_a, c_a, t_b, s_b, c_b, e_a, e_b, white):
    t = torch.cat((t_a, t_b), -1) if t_b is not None else t_a
    s = torch.cat((s_a, s_b), -1) if t_b is not None else s_a
    c = torch.cat((c_a, c_b), -2) if t_b is not None else c_a
    e = (torch.cat((e_a, e_b), -2) if t_b is not None else e_a) if e_a is not None else None
    if t_b is not None:
        t, perm = torch.sort(t, dim=-1, stable=True)
        s = s.gather(-1, perm)
        c = c.gather(-2, perm.unsqueeze(-1).expand(*perm.shape, 3))
        if e is not None:
            e = e.gather(-2, perm.unsqueeze(-1).expand(*perm.shape, e.shape[-1]))
    rgb_map, depth_map, acc, sem_map, _ = orc.composite(s, c, rd, t, e, white)
    return rgb_map, acc, sem_map


@pytest.mark.parametrize('two_lists,white,n_a,n_b,extras', [(True, True, 64, 64, 0), (True, False, 16, 16, 5),
                                                            (False, True, 32, 0, 3), (True, True, 40, 24, 0),
                                                            (True, False, 128, 128, 2), (True, True, 96, 70, 0)])
def test_composite_backward(gpu_device, two_lists, white, n_a, n_b, extras):
    g = torch.Generator().manual_seed(17 + n_a + extras)
    N = (3, 5, 7)
    rd = torch.nn.functional.normalize(torch.randn(*N, 3, generator=g), dim=-1) * (1 + 0.1 * torch.rand(*N, 1, generator=g))
    t_a = torch.sort(torch.rand(*N, n_a, generator=g) * 2 + 0.5, dim=-1)[0]
    s_a = torch.rand(*N, n_a, generator=g) * 8 * (torch.rand(*N, n_a, generator=g) > 0.3)
    c_a = torch.rand(*N, n_a, 3, generator=g) * 2 - 1
    t_b = s_b = c_b = e_a = e_b = None
    if two_lists:
        t_b = torch.rand(*N, n_b, generator=g) * 2 + 0.5
        s_b = torch.rand(*N, n_b, generator=g) * 20
        c_b = torch.rand(*N, n_b, 3, generator=g) * 2 - 1
    if extras:
        e_a = torch.rand(*N, n_a, extras, generator=g)
        e_b = torch.rand(*N, n_b, extras, generator=g) if two_lists else None
    w_rgb, w_mask = torch.randn(*N, 3, generator=g), torch.randn(*N, generator=g)
    w_ex = torch.randn(*N, extras, generator=g) if extras else None

    leaves = [x for x in (rd, s_a, c_a, s_b, c_b, e_a, e_b) if x is not None]
    # ---- oracle, float64
    ref_in = [x.double().requires_grad_() for x in leaves]
    it = iter(ref_in)
    r_rd, r_sa, r_ca = next(it), next(it), next(it)
    r_sb, r_cb = (next(it), next(it)) if two_lists else (None, None)
    r_ea = next(it) if extras else None
    r_eb = next(it) if (extras and two_lists) else None
    rgb_map, acc, sem = oracle_merge_composite(r_rd, t_a.double(), r_sa, r_ca, None if t_b is None else t_b.double(),
                                               r_sb, r_cb, r_ea, r_eb, white)
    loss = (rgb_map * w_rgb.double()).sum() + (acc * w_mask.double()).sum()
    if extras:
        loss = loss + (sem * w_ex.double()).sum()
    ref_g = torch.autograd.grad(loss, ref_in)
    # ---- HIP
    dev = gpu_device
    hip_in = [x.to(dev).requires_grad_() for x in leaves]
    it = iter(hip_in)
    h_rd, h_sa, h_ca = next(it), next(it), next(it)
    h_sb, h_cb = (next(it), next(it)) if two_lists else (None, None)
    h_ea = next(it) if extras else None
    h_eb = next(it) if (extras and two_lists) else None
    if two_lists:
        rgb_h, dep_h, mask_h, _, ex_h = nu.merge_and_composite(h_rd, t_a.to(dev), h_sa, h_ca, t_b.to(dev), h_sb, h_cb,
                                                              None, None, h_ea, h_eb, white_background=white)
    else:
        rgb_h, dep_h, mask_h, _, ex_h = nu.render_volume_density(h_sa, h_ca, None, h_rd, t_a.to(dev), None, h_ea, white)
    rel_close(rgb_h, rgb_map, 'forward rgb', 1e-5)
    assert not dep_h.requires_grad
    loss_h = (rgb_h * w_rgb.to(dev)).sum() + (mask_h * w_mask.to(dev)).sum()
    if extras:
        loss_h = loss_h + (ex_h * w_ex.to(dev)).sum()
    hip_g = torch.autograd.grad(loss_h, hip_in)
    names = ['rd', 'sigma_a', 'rgb_a'] + (['sigma_b', 'rgb_b'] if two_lists else []) + \
            (['extra_a'] + (['extra_b'] if two_lists else []) if extras else [])
    for n, a, b in zip(names, hip_g, ref_g):
        rel_close(a, b, 'grad ' + n)


def test_points_backward(gpu_device):
    g = torch.Generator().manual_seed(5)
    ro, rd = torch.randn(2, 4, 6, 3, generator=g), torch.randn(2, 4, 6, 3, generator=g)
    t = torch.rand(2, 4, 6, 40, generator=g) + 0.5
    w = torch.randn(2, 4, 6, 40, 3, generator=g)
    a, b = ro.double().requires_grad_(), rd.double().requires_grad_()
    ref = torch.autograd.grad((orc.points_on_rays(a, b, t.double()) * w.double()).sum(), (a, b))
    ha, hb = ro.to(gpu_device).requires_grad_(), rd.to(gpu_device).requires_grad_()
    x = nu.points_on_rays(ha, hb, t.to(gpu_device))
    got = torch.autograd.grad((x * w.to(gpu_device)).sum(), (ha, hb))
    rel_close(got[0], ref[0], 'g_ro'); rel_close(got[1], ref[1], 'g_rd')
    # through the stratified-sampling entry point too (depth is not differentiable)
    near = torch.full((2, 4, 6), 0.7).to(gpu_device); far = torch.full((2, 4, 6), 2.0).to(gpu_device)
    q, depth = nu.compute_query_points_from_rays(ha, hb, near, far, 24, randomize=True)
    assert q.requires_grad and not depth.requires_grad
    gq = torch.autograd.grad(q.sum(), (ha, hb))
    rel_close(gq[0], torch.full_like(ro, 24.0), 'g_ro (stratified)')
    rel_close(gq[1], depth.sum(-1, keepdim=True).expand_as(rd), 'g_rd (stratified)')


@pytest.mark.parametrize('name', ['persp_white_fine_rand', 'persp_bbox_black_fine_rand', 'ortho_fine_det'])
def test_raygen_backward(gpu_device, name):
    meta, t = load_golden(name)
    H, W = meta['H'], meta['W']
    g = torch.Generator().manual_seed(3)
    w_o, w_d = torch.randn(meta['B'], H, W, 3, generator=g), torch.randn(meta['B'], H, W, 3, generator=g)
    cam = t['cam2world'].double().requires_grad_()
    focal = t['focal'].double().requires_grad_() if 'focal' in t else None
    bbox = t['bbox'].double() if 'bbox' in t else None
    ro, rd = orc.ray_bundle(H, W, focal, cam, bbox)
    rd = orc.unit_dirs(rd)
    leaves = [cam] + ([focal] if focal is not None else [])
    ref = torch.autograd.grad((ro * w_o.double()).sum() + (rd * w_d.double()).sum(), leaves)
    dev = gpu_device
    hcam = t['cam2world'].to(dev).requires_grad_()
    hfocal = t['focal'].to(dev).requires_grad_() if 'focal' in t else None
    hbbox = t['bbox'].to(dev) if 'bbox' in t else None
    hro, hrd = nu.get_ray_bundle_normalized(H, W, hfocal, hcam, hbbox)
    hleaves = [hcam] + ([hfocal] if hfocal is not None else [])
    got = torch.autograd.grad((hro * w_o.to(dev)).sum() + (hrd * w_d.to(dev)).sum(), hleaves)
    rel_close(got[0][:, :3], ref[0][:, :3], 'g_cam2world')
    if meta['ortho']:
        # normalised directions do not depend on the homogeneous scale: the reference gradient is 0,
        # so compare on the scale of the whole matrix gradient
        rel_close(got[0], ref[0], 'g_cam2world incl. [3,3]')
    if hfocal is not None:
        rel_close(got[1], ref[1], 'g_focal')
    # un-normalised variant
    cam2 = t['cam2world'].double().requires_grad_()
    ro2, rd2 = orc.ray_bundle(H, W, None if focal is None else t['focal'].double(), cam2, bbox)
    ref2 = torch.autograd.grad((rd2 * w_d.double()).sum(), cam2)[0]
    hcam2 = t['cam2world'].to(dev).requires_grad_()
    _, hrd2 = nu.get_ray_bundle(H, W, None if hfocal is None else hfocal.detach(), hcam2, hbbox)
    got2 = torch.autograd.grad((hrd2 * w_d.to(dev)).sum(), hcam2)[0]
    rel_close(got2[:, :3, :3], ref2[:, :3, :3], 'g_cam2world (raw directions)')


# --------------------------------------------------------------------------------------------
# field query backward (sampler closure) and the whole differentiable render
# --------------------------------------------------------------------------------------------
import types  # noqa: E402

from stand_in import StandInGenerator, look_at_cameras, _Decoder  # noqa: E402
import nerf_from_image_amd.generator as nfi_gen  # noqa: E402
import nerf_from_image_amd.render as nfi_render  # noqa: E402


@pytest.mark.parametrize('A,use_sdf,P', [(10, True, 200), (0, False, 70), (10, True, 64)])
def test_field_query_backward(gpu_device, A, use_sdf, P):
    _field_query_backward_case(gpu_device, A, use_sdf, P, None)


@pytest.mark.parametrize('scale', [1e-7, 3e4, 'mixed'])
def test_field_query_backward_at_any_gradient_scale(gpu_device, scale):
    """The backward is linear in the upstream gradient, whose scale is the caller's business (the inversion loop's
    mean-squared error over 10^4 pixels hands down 1e-6 per sample).  The kernel's split-fp16 MFMA operands resolve
    2^-24 ABSOLUTE, so without the per-point power-of-two normalisation tiny upstream gradients lost most of their bits
    (found in round 2: 3-8 % gradient error in tools/inversion_synthetic.py while every unit-scale test passed).
    'mixed': every point its own scale over 13 decades; g_points is then checked per point."""
    _field_query_backward_case(gpu_device, 10, True, 200, scale)


def _field_query_backward_case(dev, A, use_sdf, P, scale):
    g = torch.Generator().manual_seed(100 + A + P)
    B, R = 2, 24
    r = float(torch.tensor(0.55, dtype=torch.float32))      # the fp32 value the kernels divide by (face points!)
    low = torch.randn(B * 3, 32, 6, 6, generator=g)
    planes = torch.nn.functional.interpolate(low, size=(R, R), mode='bilinear', align_corners=True)
    planes = (planes + 0.1 * torch.randn(B * 3, 32, R, R, generator=g)).view(B, 3, 32, R, R)
    dec = _Decoder(1 + A if A > 0 else 4, g)
    x = (torch.rand(B, P, 3, generator=g) * 2 - 1) * r * 1.1          # some points outside the cube
    x[0, 0] = torch.tensor([r, 0.1, -r])                               # on the faces: clamped coordinates
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1) if A > 0 else None
    beta, alpha = torch.tensor([0.12]), torch.tensor([0.3])
    w_sig, w_rgb = torch.randn(B, P, generator=g), torch.randn(B, P, 3, generator=g)
    w_sdf = torch.randn(B, P, generator=g)
    w_sem = torch.randn(B, P, A, generator=g) if A > 0 else None
    if scale is not None:
        ps = 10.0 ** (torch.rand(B, P, generator=g) * 13 - 9) if scale == 'mixed' else torch.full((B, P), float(scale))
        w_sig, w_rgb, w_sdf = w_sig * ps, w_rgb * ps[..., None], w_sdf * ps
        w_sem = w_sem * ps[..., None] if A > 0 else None

    # ---- oracle in float64
    dd = lambda t: None if t is None else t.double().requires_grad_()
    o_x, o_pl, o_att, o_be, o_al = dd(x), dd(planes), dd(att), dd(beta), dd(alpha)
    o_w = [dd(p.detach()) for p in (dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias)]
    q = orc.field_query(o_pl, o_w[0], o_w[1], o_w[2], o_w[3], o_x, r, use_sdf, o_be if use_sdf else None,
                        o_al if use_sdf else None, o_att)
    loss = (q['sigma'] * w_sig.double()).sum() + (q['rgb'] * w_rgb.double()).sum() + (q['sdf'] * w_sdf.double()).sum()
    if A > 0:
        loss = loss + (q['semantics'] * w_sem.double()).sum()
    leaves = [o_x, o_pl] + o_w + ([o_att] if A > 0 else []) + ([o_be, o_al] if use_sdf else [])
    ref = torch.autograd.grad(loss, leaves)

    # ---- HIP through the sampler closure
    dec = dec.to(dev)
    h_pl = planes.to(dev).requires_grad_()
    h_x = x.to(dev).requires_grad_()
    h_att = att.to(dev).requires_grad_() if A > 0 else None
    h_be = beta.to(dev).requires_grad_() if use_sdf else None
    h_al = alpha.to(dev).requires_grad_() if use_sdf else None
    sampler = nfi_gen.make_sampler(h_pl, dec, r, A, h_att, use_sdf, h_be, h_al)
    req = ['sigma', 'rgb', 'sdf_distance'] + (['semantics'] if A > 0 else [])
    res = sampler(h_x, req)
    rel_close(res['sigma'], q['sigma'], 'forward sigma', 2e-4)
    loss_h = (res['sigma'] * w_sig.to(dev)).sum() + (res['rgb'] * w_rgb.to(dev)).sum() + \
             (res['sdf_distance'][..., 0] * w_sdf.to(dev)).sum()
    if A > 0:
        loss_h = loss_h + (res['semantics'] * w_sem.to(dev)).sum()
    h_w = [dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias]
    h_leaves = [h_x, h_pl] + h_w + ([h_att] if A > 0 else []) + ([h_be, h_al] if use_sdf else [])
    got = torch.autograd.grad(loss_h, h_leaves)
    names = ['points', 'planes', 'w1', 'b1', 'w2', 'b2'] + (['attention_values'] if A > 0 else []) + \
            (['beta', 'alpha'] if use_sdf else [])
    for n, a, b in zip(names, got, ref):
        rel_close(a, b, 'grad ' + n, 5e-4)
    if scale == 'mixed':
        # per point: the coordinate gradient relative to that point's own magnitude
        a, b = got[0].double().cpu(), ref[0]
        err = (a - b).norm(dim=-1) / b.norm(dim=-1).clamp_min(1e-300)
        live = b.norm(dim=-1) > 0
        assert float(err[live].max()) < 2e-3, float(err[live].max())


@pytest.mark.parametrize('A,use_sdf,N,S', [(10, True, 12, 16), (0, False, 7, 10), (10, True, 5, 64)])
def test_field_query_backward_viewdir(gpu_device, A, use_sdf, N, S):
    """--use_viewdir decoder (33 outputs + per-ray feature + third layer): every gradient, incl. the ray feature
    (-> ViewDirectionMapper) and the mapper's output layer, against float64 autograd of the oracle."""
    dev = gpu_device
    g = torch.Generator().manual_seed(300 + A + N)
    B, R = 2, 24
    P = N * S
    r = float(torch.tensor(0.55, dtype=torch.float32))
    low = torch.randn(B * 3, 32, 6, 6, generator=g)
    planes = torch.nn.functional.interpolate(low, size=(R, R), mode='bilinear', align_corners=True)
    planes = (planes + 0.1 * torch.randn(B * 3, 32, R, R, generator=g)).view(B, 3, 32, R, R)
    dec = _Decoder(33, g)
    n3 = A if A > 0 else 3
    out_layer = torch.nn.Linear(32, n3)
    with torch.no_grad():
        out_layer.weight.copy_(torch.randn(n3, 32, generator=g))
        out_layer.bias.copy_(0.3 * torch.randn(n3, generator=g))
    x = (torch.rand(B, N, S, 3, generator=g) * 2 - 1) * r * 1.1
    xr = torch.randn(B, N, 1, 32, generator=g)
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1) if A > 0 else None
    beta, alpha = torch.tensor([0.12]), torch.tensor([0.3])
    w_sig, w_rgb = torch.randn(B, P, generator=g), torch.randn(B, P, 3, generator=g)
    w_sdf = torch.randn(B, P, generator=g)
    w_sem = torch.randn(B, P, A, generator=g) if A > 0 else None

    dd = lambda t: None if t is None else t.detach().double().requires_grad_()
    o_x, o_pl, o_att, o_be, o_al, o_xr = dd(x), dd(planes), dd(att), dd(beta), dd(alpha), dd(xr)
    o_w = [dd(p) for p in (dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias,
                           out_layer.weight, out_layer.bias)]
    q = orc.field_query(o_pl, o_w[0], o_w[1], o_w[2], o_w[3], o_x, r, use_sdf, o_be if use_sdf else None,
                        o_al if use_sdf else None, o_att, viewdir=dict(x=o_xr, w3=o_w[4], b3=o_w[5]))
    loss = (q['sigma'] * w_sig.double()).sum() + (q['rgb'] * w_rgb.double()).sum() + (q['sdf'] * w_sdf.double()).sum()
    if A > 0:
        loss = loss + (q['semantics'] * w_sem.double()).sum()
    leaves = [o_x, o_pl] + o_w + [o_xr] + ([o_att] if A > 0 else []) + ([o_be, o_al] if use_sdf else [])
    ref = torch.autograd.grad(loss, leaves)

    dec, out_layer = dec.to(dev), out_layer.to(dev)
    h_pl, h_x, h_xr = planes.to(dev).requires_grad_(), x.to(dev).requires_grad_(), xr.to(dev).requires_grad_()
    h_att = att.to(dev).requires_grad_() if A > 0 else None
    h_be = beta.to(dev).requires_grad_() if use_sdf else None
    h_al = alpha.to(dev).requires_grad_() if use_sdf else None
    sampler = nfi_gen.make_sampler(h_pl, dec, r, A, h_att, use_sdf, h_be, h_al, viewdir=(h_xr, out_layer))
    req = ['sigma', 'rgb', 'sdf_distance'] + (['semantics'] if A > 0 else [])
    res = sampler(h_x, req)
    rel_close(res['sigma'], q['sigma'], 'forward sigma', 2e-4)
    rel_close(res['rgb'], q['rgb'], 'forward rgb', 2e-4)
    loss_h = (res['sigma'] * w_sig.to(dev)).sum() + (res['rgb'] * w_rgb.to(dev)).sum() + \
             (res['sdf_distance'][..., 0] * w_sdf.to(dev)).sum()
    if A > 0:
        loss_h = loss_h + (res['semantics'] * w_sem.to(dev)).sum()
    h_w = [dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias, out_layer.weight, out_layer.bias]
    h_leaves = [h_x, h_pl] + h_w + [h_xr] + ([h_att] if A > 0 else []) + ([h_be, h_al] if use_sdf else [])
    got = torch.autograd.grad(loss_h, h_leaves)
    names = ['points', 'planes', 'w1', 'b1', 'w2', 'b2', 'w3', 'b3', 'ray_feature'] + \
            (['attention_values'] if A > 0 else []) + (['beta', 'alpha'] if use_sdf else [])
    for n, a, b in zip(names, got, ref):
        rel_close(a, b, 'grad ' + n, 5e-4)


@pytest.mark.parametrize('fine,ortho,S', [(True, False, 32), (False, False, 32), (True, True, 32), (False, False, 512),
                                          (True, False, 96),       # 96 + 96: the two-slot render kernel's stash
                                          (False, False, 128)])    # run.py without --fine_sampling: one pass of 128
def test_render_backward_end_to_end(gpu_device, fine, ortho, S):
    """d(rgb, mask)/d(planes producer params, decoder, beta, alpha, attention values, camera, focal) through
    nfi_render.render - up to 128 samples per pass with fine sampling, up to 512 in a single pass (the inversion loop
    without --fine_sampling, run.py:2271), the fused render + training stash as ONE autograd node (asserted: one render
    launch, no stage kernel of the staged path) - against autograd of the oracle with the same noise."""
    from test_host_api_gpu import OneRenderLaunch, RandTap
    dev = gpu_device
    torch.manual_seed(7)
    model = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32).to(dev)
    with torch.no_grad():
        model.alpha.fill_(0.2)
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(21)
    B, H, W = 2, 12, 10                # S = 512: one pass at the inversion loop's sample count without fine sampling
    cam0 = look_at_cameras(B, 1.5, g)
    focal0 = None if ortho else torch.full((B,), 1.1)
    if ortho:
        cam0[:, :3, 3] *= 2.0
    z = torch.randn(B, 512, generator=g).to(dev)
    w_rgb, w_mask = torch.randn(B, H, W, 3, generator=g), torch.randn(B, H, W, generator=g)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=fine)
    dcfg = {'scene_range': 0.55 if not ortho else 1.2, 'white_background': True}
    model.scene_range = dcfg['scene_range']
    render = nfi_render.make_render(cfg, dcfg)

    cam = cam0.to(dev).requires_grad_()
    focal = None if ortho else focal0.to(dev).requires_grad_()
    with RandTap() as tap, OneRenderLaunch() as one:
        rgb, depth, mask, _, _, _ = render(model, H, W, cam, focal, None, None, z, S)
        loss = (rgb * w_rgb.to(dev)).sum() + (mask * w_mask.to(dev)).sum()
        params = [model.decoder.net[0].weight, model.decoder.net[0].bias, model.decoder.net[2].weight,
                  model.decoder.net[2].bias, model.beta, model.alpha, model.synthesis_network.basis,
                  model.texture_mapper.lin.weight, cam] + ([] if ortho else [focal])
        got = torch.autograd.grad(loss, params)
    assert one.calls == 1

    # ---- oracle with identical noise: float64 (reference value) and float32 (the precision the
    # reference runs at; its distance to float64 calibrates how much rounding alone moves a gradient
    # that is a sum of cancelling terms and depends on discontinuous sample placement)
    import copy

    def oracle_grads(dtype):
        ref_model = copy.deepcopy(model).cpu().to(dtype)
        planes, att = ref_model.planes_and_values(z.cpu().to(dtype))
        dec = ref_model.decoder.net
        ocam = cam0.to(dtype).requires_grad_()
        ofocal = None if ortho else focal0.to(dtype).requires_grad_()
        draws = [d.to(dtype) for d in tap.draws]
        o = orc.render(planes, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, ocam, ofocal, H, W, S,
                       dcfg['scene_range'], white_background=True, fine_sampling=fine, noise_coarse=draws[0],
                       noise_fine=draws[1] if fine else None, use_sdf=True, beta=ref_model.beta, alpha=ref_model.alpha,
                       attention_values=att)
        oloss = (o['rgb'] * w_rgb.to(dtype)).sum() + (o['mask'] * w_mask.to(dtype)).sum()
        oparams = [dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, ref_model.beta, ref_model.alpha,
                   ref_model.synthesis_network.basis, ref_model.texture_mapper.lin.weight, ocam] + \
                  ([] if ortho else [ofocal])
        return o, torch.autograd.grad(oloss, oparams)
    o, ref = oracle_grads(torch.float64)
    _, ref32 = oracle_grads(torch.float32)
    rel_close(rgb, o['rgb'], 'forward rgb', 2e-4)
    names = ['w1', 'b1', 'w2', 'b2', 'beta', 'alpha', 'plane producer', 'texture mapper', 'cam2world'] + \
            ([] if ortho else ['focal'])
    for n, a, b, b32 in zip(names, got, ref, ref32):
        if n == 'cam2world':
            a, b, b32 = a[:, :3], b[:, :3], b32[:, :3]
        scale = b.abs().max().clamp_min(1e-12)
        noise = float((b32.double() - b).abs().max() / scale)
        rel_close(a, b, 'grad ' + n, max(2e-3, 4 * noise))


@pytest.mark.parametrize('P,res', [(5000, 24), (333, 9), (70000, 64), (40000, 256), (9000, 600), (3, 2), (20000, 17),
                                   (1 << 20, 128),         # 128 MB of gradient rows per scene: two point groups, 8-texel tiles
                                   (3 << 20, 256)])        # 384 MB: eight groups, 16-texel tiles (the training step's shape)
def test_binned_scatter_matches_atomic_scatter(gpu_device, P, res):
    """scatter_mode 1 (two-level counting sort by plane tile and texel cell + register accumulation with carried
    corners) against scatter_mode 0 (atomics per point): identical gradients up to fp32 summation order; ragged P,
    points outside the cube, zero upstream gradients (skipped by the binning), empty cells, a crowd of a quarter of the
    points in a few cells (buckets split into several 4096-entry chunks), plane sides that are not multiples of the
    16-texel tile, the 32-texel tiles of planes above 512^2 and the smallest plane; scenes whose gradient rows exceed the
    Infinity-Cache window and are cut into point groups (group-major bucket ids, work items handed out in order)."""
    from nerf_from_image_amd import field_backward as fb, ops as hops
    dev = gpu_device
    g = torch.Generator().manual_seed(500 + P)
    B, A, r = 2, 10, 0.55
    planes = torch.randn(B, 3, 32, res, res, generator=g).to(dev)
    dec = _Decoder(1 + A, g).to(dev)
    w1, b1, w2, b2 = dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias
    x = ((torch.rand(B, P, 3, generator=g) * 2 - 1) * r * 1.15).to(dev)
    x[0, : P // 4] *= 0.05                                           # a crowd of points in a few cells
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
    beta, alpha = torch.tensor([0.12], device=dev), torch.tensor([0.3], device=dev)
    g_sig = torch.randn(B, P, generator=g).to(dev)
    g_rgb = torch.randn(B, P, 3, generator=g).to(dev)
    g_sig[1, P // 2:] = 0; g_rgb[1, P // 2:] = 0                      # points with no gradient at all
    texels = hops.planes_to_texels(planes)
    image = hops.decoder_pack(w1, b1, w2, b2, A)
    res_by_mode = []
    for mode in (0, 1):
        res_by_mode.append(fb.field_query_bwd(x, texels, image, w1, w2, r, A, att, True, beta, alpha, g_sig, g_rgb,
                                              want_points=True, scatter_mode=mode))
    a, b = res_by_mode
    assert a['g_texels'].abs().max() > 0
    for k in a:
        rel_close(b[k], a[k], 'binned vs atomic ' + k, 2e-5)


def test_backward_outputs_without_atomics_are_bit_reproducible(gpu_device):
    """g_points (and the normals path) involve no atomics, so repeated launches must agree bit for bit.  Regression
    test for a round-2 finding: with the coordinate-gradient arithmetic compiled to packed fp32 (v_pk_mul_f32 /
    v_pk_add_f32 with cross-half op_sel, LLVM's SLP vectoriser) one product of the last plane came out as zero for
    the wave's lanes 48..63 once in about 1e5 tiles, depending on timing - points 12..15 of a 16-point tile, z
    component only.  nfi_backward_field.hip is therefore built without SLP vectorisation (tools/determinism_probe.py
    counts events over thousands of launches: 54 in 1500 before, 0 in 3000 after)."""
    from nerf_from_image_amd import field_backward as fb, ops as hops
    dev = gpu_device
    g = torch.Generator().manual_seed(70500)
    B, A, r, P, res = 2, 10, 0.55, 70000, 64
    planes = torch.randn(B, 3, 32, res, res, generator=g).to(dev)
    dec = _Decoder(1 + A, g).to(dev)
    w1, b1, w2, b2 = dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias
    x = ((torch.rand(B, P, 3, generator=g) * 2 - 1) * r * 1.15).to(dev)
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
    beta, alpha = torch.tensor([0.12], device=dev), torch.tensor([0.3], device=dev)
    gs, gr = torch.randn(B, P, generator=g).to(dev), torch.randn(B, P, 3, generator=g).to(dev)
    texels, image = hops.planes_to_texels(planes), hops.decoder_pack(w1, b1, w2, b2, A)

    def run(mode, **kw):
        return fb.field_query_bwd(x, texels, image, w1, w2, r, A, att, True, beta, alpha, gs, gr, scatter_mode=mode, **kw)
    ref = run(0, want_points=True)['g_points'].clone()
    ref_n = run(0, points_only=True, normalize_points=True)['g_points'].clone()
    for _ in range(8):
        assert torch.equal(run(1, want_points=True)['g_points'], ref)
        assert torch.equal(run(0, want_points=True)['g_points'], ref)
        assert torch.equal(run(0, points_only=True, normalize_points=True)['g_points'], ref_n)


def test_render_backward_without_camera_gradient(gpu_device):
    """force_no_cam_grad (run.py:211-214) and a camera that needs no gradient: the other gradients must be the ones of the
    full backward; a camera without requires_grad takes the field backward without its coordinate-gradient pass."""
    from test_host_api_gpu import RandTap
    dev = gpu_device
    torch.manual_seed(9)
    model = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32).to(dev)
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(22)
    B, H, W, S = 2, 16, 16, 32
    cam0 = look_at_cameras(B, 1.5, g).to(dev)
    focal = torch.full((B,), 1.1, device=dev)
    z = torch.randn(B, 512, generator=g).to(dev)
    w_rgb = torch.randn(B, H, W, 3, generator=g).to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 0.55, 'white_background': False})
    params = [model.decoder.net[0].weight, model.decoder.net[2].weight, model.beta, model.synthesis_network.basis]

    def grads(cam, force):
        torch.manual_seed(33)                      # identical noise draws
        rgb, _, mask, _, _, _ = render(model, H, W, cam, focal, None, None, z, S, force_no_cam_grad=force)
        return rgb, torch.autograd.grad((rgb * w_rgb).sum() + mask.sum(), params + ([cam] if cam.requires_grad and not force else []))
    cam_g = cam0.clone().requires_grad_()
    rgb_a, full = grads(cam_g, False)
    rgb_b, forced = grads(cam_g, True)
    rgb_c, plain = grads(cam0.clone(), False)
    assert torch.equal(rgb_a, rgb_b) and torch.equal(rgb_a, rgb_c)
    assert float(full[-1].abs().max()) > 0
    for a, b, c in zip(full[:4], forced, plain):         # (float atomics: equal up to summation order)
        scale = max(a.abs().max().item(), 1e-12)
        assert (b - c).abs().max().item() <= 1e-4 * scale
        assert (a - b).abs().max().item() <= 1e-4 * scale
    # force_no_cam_grad detaches the coarse points, the depths and the directions - run.py:286-288 then builds the FINE points
    # from the undetached origins, so the camera's translation (and nothing else of it) still gets a gradient, as in the
    # reference (tests/test_reference_gpu.py::test_force_no_cam_grad_matches_the_real_reference)
    g_forced, = torch.autograd.grad(render(model, H, W, cam_g, focal, None, None, z, S, force_no_cam_grad=True)[0].sum(), [cam_g])
    assert float(g_forced[:, :3, :3].abs().max()) == 0.0 and float(g_forced[:, :3, 3].abs().max()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,spr', [(32, 48, 128), (16, 24, 64), (40, 8, 192)])
def test_ray_order_hint_changes_nothing(gpu_device, H, W, spr):
    """nfi_field_bwd_args.rays_per_row: the points are [rays][samples] of an H x W image; the kernel then walks the rays in
    16 x 16- (or 8 x 8-) pixel tiles, one tile per XCD at a time.  Same coordinate gradients bit for bit (they are written
    per point), same plane / parameter gradients up to fp32 summation order; also for images that only divide into 8-pixel
    tiles."""
    from nerf_from_image_amd import field_backward as fb, ops as hops
    dev = gpu_device
    g = torch.Generator().manual_seed(77 + H)
    B, A, r, res = 2, 10, 0.55, 64
    P = H * W * spr
    planes = torch.randn(B, 3, 32, res, res, generator=g).to(dev)
    dec = _Decoder(1 + A, g).to(dev)
    w1, b1, w2, b2 = dec.net[0].weight, dec.net[0].bias, dec.net[2].weight, dec.net[2].bias
    x = ((torch.rand(B, P, 3, generator=g) * 2 - 1) * r).to(dev)
    att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
    beta, alpha = torch.tensor([0.12], device=dev), torch.tensor([0.3], device=dev)
    g_sig, g_rgb = torch.randn(B, P, generator=g).to(dev), torch.randn(B, P, 3, generator=g).to(dev)
    g_sig[0, : P // 3] = 0; g_rgb[0, : P // 3] = 0                     # whole tiles without a gradient
    texels, image = hops.planes_to_texels(planes), hops.decoder_pack(w1, b1, w2, b2, A)
    a = fb.field_query_bwd(x, texels, image, w1, w2, r, A, att, True, beta, alpha, g_sig, g_rgb, want_points=True)
    b = fb.field_query_bwd(x, texels, image, w1, w2, r, A, att, True, beta, alpha, g_sig, g_rgb, want_points=True,
                           ray_order=(spr, W))
    assert torch.equal(a['g_points'], b['g_points'])
    assert a['g_texels'].abs().max() > 0
    for k in a:
        # (beta / alpha: one fp32 sum over all points of terms that largely cancel with these random gradients - the
        #  order of the walk shows at 1e-4 of the RESULT, 1e-7 of the terms)
        rel_close(b[k], a[k], 'ray-order hint ' + k, 1e-3 if k in ('g_beta', 'g_alpha') else 2e-5)


@pytest.mark.gpu
def test_planes_to_texels_all_sizes(gpu_device):
    """The layout kernels against torch.permute: tiny / odd planes, sizes that are not multiples of the 256-pixel block,
    fp32 / bf16 / fp16 storage (round to nearest even), and the round trip."""
    from nerf_from_image_amd import ops as hops
    g = torch.Generator().manual_seed(0)
    for R in (2, 3, 5, 16, 17, 48, 255, 300):
        pl = torch.randn(2, 3, 32, R, R, generator=g).to(gpu_device)
        ref = pl.permute(0, 1, 3, 4, 2).contiguous()
        for dt, td in ((hops.TEXEL_F32, torch.float32), (hops.TEXEL_BF16, torch.bfloat16), (hops.TEXEL_F16, torch.float16)):
            assert torch.equal(hops.planes_to_texels(pl, dt).view(ref.shape), ref.to(td)), (R, dt)
        assert torch.equal(hops.texels_to_planes(hops.planes_to_texels(pl)), pl), R
