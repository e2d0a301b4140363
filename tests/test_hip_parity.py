"""GPU parity tests: every HIP stage and the fused renderer against the oracle, through the C ABI.

Tolerances (fp32):
  * indices / masks / permutations: bit-exact on identical float inputs;
  * elementwise stages that ATen evaluates op by op (near/far, stratified depths, query points,
    smoothing): bit-exact on identical inputs;
  * rendered rgb / depth / mask and per-sample rgb: 1e-4 absolute (north star);
  * sigma: RELATIVE bound |d sigma| <= 3e-5 * max(1, |sigma|) per sample (sigma reaches 1/alpha = 50; the
    absolute 1e-4 of the north star holds wherever sigma <= 3.3).  sigma = laplace_cdf(-d/beta)/alpha amplifies
    the decoder's 1e-7-level rounding differences by up to 0.5/(alpha*beta); measured on MI355X (tools/
    parity_report.py, profiles/r2/parity_report.json): <= 1.5e-5 relative, <= 2.8e-4 absolute at sigma ~ 50; the
    reference's own CPU-vs-GPU difference is of the same size (tools/gpu_diag.py);
  * index flips against the reference END TO END (searchsorted on a cdf computed by a different summation order):
    measured 0 on every randomised golden case, 2e-3 / 5.6e-3 on the two deterministic-u cases (linspace u lands
    exactly on cdf break points of flat pdfs); the asserts are about twice the measured rates.
"""
import pytest
import torch

from conftest import golden_case_names, load_golden
from parity_util import err, hip_field_setup, hip_render, oracle_normal_map, oracle_render, viewdir_of
from nerf_from_image_amd import ops
from oracle import nfi_oracle as orc

pytestmark = pytest.mark.gpu
ATOL = 1e-4


SIGMA_RTOL = 3e-5


def sigma_close(a, b, what):
    """|a - b| <= SIGMA_RTOL * max(1, |b|) elementwise (relative for large sigma, absolute below 1)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    viol = (a - b).abs() / b.abs().clamp_min(1.0)
    assert torch.isfinite(viol).all() and float(viol.max()) <= SIGMA_RTOL, (what, float(viol.max()), float((a - b).abs().max()))


@pytest.fixture(scope='module', params=[n for n in golden_case_names() if 's512' not in n])
def case(request, gpu_device):
    meta, t = load_golden(request.param)
    o = oracle_render(meta, t, 'cpu')
    return request.param, meta, t, o, gpu_device


def _fine_case_names():
    import json
    import os
    from conftest import GOLDEN
    cases = json.load(open(os.path.join(GOLDEN, 'cases.json')))
    return [n for n in golden_case_names() if 's512' not in n and cases[n]['fine']]


@pytest.fixture(scope='module', params=_fine_case_names())
def fine_case(request, gpu_device):
    """The golden cases WITH a fine pass (the single-pass cases have no resampling stage to test: they are left out of
    the parametrisation instead of being skipped)."""
    meta, t = load_golden(request.param)
    o = oracle_render(meta, t, 'cpu')
    return request.param, meta, t, o, gpu_device


def close(a, b, tol, what):
    e = err(a, b)
    assert e['nonfinite'] == 0 and e['max'] <= tol, (what, e)


def exact(a, b, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape and torch.equal(a, b), (what, err(a.float(), b.float()))


def test_texel_layout_roundtrip(gpu_device):
    g = torch.Generator().manual_seed(3)
    planes = torch.randn(2, 3, 32, 20, 20, generator=g).to(gpu_device)
    tex = ops.planes_to_texels(planes)
    exact(tex, planes.permute(0, 1, 3, 4, 2).contiguous(), 'texels')
    exact(ops.texels_to_planes(tex), planes, 'roundtrip')
    tb = ops.planes_to_texels(planes, ops.TEXEL_BF16)
    exact(tb, planes.permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16), 'bf16 texels')
    th = ops.planes_to_texels(planes, ops.TEXEL_F16)
    exact(th, planes.permute(0, 1, 3, 4, 2).contiguous().to(torch.float16), 'fp16 texels')


def test_rays_and_planes(case):
    name, meta, t, o, dev = case
    g = lambda k: t[k].to(dev) if k in t else None
    ro, rd = ops.raygen(meta['H'], meta['W'], g('focal'), g('cam2world'), g('bbox'), g('center'), normalize=True)
    exact(ro, o['ro'], 'ray origins')
    close(rd, o['rd'], 2e-7, 'ray directions')          # <= 1 ulp
    ro_r, rd_r = ops.raygen(meta['H'], meta['W'], g('focal'), g('cam2world'), g('bbox'), g('center'), normalize=False)
    ro_o, rd_o = orc.ray_bundle(meta['H'], meta['W'], t.get('focal'), t['cam2world'], t.get('bbox'), t.get('center'))
    close(rd_r, rd_o, 2e-7, 'raw directions')
    near, far, hit = ops.near_far(o['ro'].contiguous().to(dev), o['rd'].to(dev), meta['scene_range'])
    exact(near, o['near'], 'near'); exact(far, o['far'], 'far'); exact(hit, o['hit'], 'hit')
    pts, dep = ops.stratified_points(o['ro'].contiguous().to(dev), o['rd'].to(dev), o['near'].to(dev), o['far'].to(dev),
                                     meta['S'], g('noise_coarse'))
    exact(dep, o['t_coarse'], 'stratified depths')
    exact(pts, orc.points_on_rays(o['ro'], o['rd'], o['t_coarse']), 'query points')


def test_no_ray_hits_raises(gpu_device):
    ro = torch.tensor([[5., 5., 5.]] * 70, device=gpu_device)
    rd = torch.tensor([[0., 0., 1.]] * 70, device=gpu_device)
    with pytest.raises(RuntimeError):
        ops.near_far(ro, rd, 0.5)


def test_field_query(case):
    name, meta, t, o, dev = case
    g = lambda k: t[k].to(dev) if k in t else None
    texels, image = hip_field_setup(meta, t, dev)
    B = meta['B']
    x = orc.points_on_rays(o['ro'], o['rd'], o['t_coarse']).reshape(B, -1, 3)
    q = ops.field_query(x.to(dev), texels, image, meta['scene_range'], meta['A'], g('attention_values'), meta['sdf'],
                        g('beta'), g('alpha'), want_sdf=True, want_semantics=meta['A'] > 0, want_outside=True,
                        ray_features=ops.pad_ray_features(g('viewdir_x')) if 'viewdir_x' in t else None,
                        samples_per_ray=meta['S'])
    exact(q['outside'].float(), o['outside_coarse'].reshape(B, -1), 'outside mask')
    close(q['sdf'], o['sdf_coarse'].reshape(B, -1), 1e-5, 'sdf')
    sigma_close(q['sigma'], o['sigma_coarse'].reshape(B, -1), 'sigma')
    close(q['rgb'], o['rgb_coarse'].reshape(B, -1, 3), ATOL, 'rgb')
    if meta['A'] > 0:
        ref = orc.field_query(t['planes'], t['w1'], t['b1'], t['w2'], t['b2'], x.view(B, -1, meta['S'], 3),
                              meta['scene_range'], meta['sdf'], t.get('beta'), t.get('alpha'), t['attention_values'],
                              viewdir_of(t))
        close(q['semantics'], ref['semantics'], 1e-5, 'semantics')


def test_field_query_ragged_and_far_points(gpu_device):
    """P not a multiple of 64, points far outside the cube and exactly on its faces."""
    dev = gpu_device
    meta, t = load_golden('persp_white_fine_rand')
    texels, image = hip_field_setup(meta, t, dev)
    g = torch.Generator().manual_seed(11)
    r = meta['scene_range']
    x = (torch.rand(2, 77, 3, generator=g) * 2 - 1) * r * 1.5
    x[0, :3] = torch.tensor([[r, -r, r], [r, 0.0, 0.0], [-r, -r, -r]])
    x[1, 5] = torch.tensor([1e6, -1e6, 3.0])
    q = ops.field_query(x.to(dev), texels, image, r, meta['A'], t['attention_values'].to(dev), True,
                        t['beta'].to(dev), t['alpha'].to(dev), want_sdf=True, want_outside=True)
    ref = orc.field_query(t['planes'], t['w1'], t['b1'], t['w2'], t['b2'], x, r, True, t['beta'], t['alpha'],
                          t['attention_values'])
    exact(q['outside'].float(), ref['outside'], 'outside mask')
    sigma_close(q['sigma'], ref['sigma'], 'sigma')
    close(q['rgb'], ref['rgb'], ATOL, 'rgb')
    close(q['sdf'], ref['sdf'], 1e-5, 'sdf')


def test_sampling_stages(fine_case):
    name, meta, t, o, dev = fine_case
    S = meta['S']
    n = o['weights_coarse'].shape[0]
    w = ops.ray_weights(o['sigma_coarse'].to(dev), o['rd'].to(dev), o['t_coarse'].to(dev))
    close(w.flatten(0, 2), o['weights_coarse'], 1e-6, 'coarse weights')
    u = t['noise_fine'].to(dev) if 'noise_fine' in t else orc.deterministic_u(n, S, o['weights_coarse']).to(dev)
    fine, taps = ops.resample(o['sigma_coarse'].to(dev), o['rd'].to(dev), o['t_coarse'].to(dev), u, want_taps=True)
    close(taps['smooth'], o['weights_smooth'], 1e-6, 'smoothed weights')
    close(taps['cdf'], o['cdf'], 1e-6, 'cdf')
    # indices: bit-exact against searchsorted on the kernel's own cdf (identical float inputs) ...
    exact(taps['inds'], torch.searchsorted(taps['cdf'].contiguous(), u.contiguous(), right=True), 'searchsorted indices')
    # ... and within the flip budget against the reference end to end.  Deterministic u (linspace)
    # lands exactly on cdf break points of flat pdfs, where a 1-ulp difference in the pdf
    # normalisation (a float sum whose order no two implementations share) flips the index while
    # the sample itself moves by an ulp; the budget is wider there and the samples are checked below.
    flips = (taps['inds'].cpu() != o['inds']).float().mean().item()
    # randomize=False (linspace u): NO caller in the reference uses it - every render(...) call in run.py leaves
    # `randomize` at its default True, eval included (run.py:185, 203-209; 1250-1264, 1444-1454, 1639-1646, 2036-2051,
    # 2264-2275) - so the 1.2e-2 bound covers an API corner: linspace u lands exactly on the cdf break points of flat pdfs,
    # where a last-bit difference in the cumsum flips searchsorted.  SURVEY's 1e-5 budget is met on the randomised cases.
    assert flips <= (1e-4 if meta['randomize'] else 1.2e-2), flips      # measured: 0 / 5.6e-3 (see the module docstring)
    close(fine, o['t_fine'].flatten(0, 2), 1e-5, 'fine depths')
    # stand-alone sample_pdf on the oracle's exact inputs
    mid = (.5 * (o['t_coarse'][..., 1:] + o['t_coarse'][..., :-1])).flatten(0, 2)
    smp, inds, cdf = ops.sample_pdf(mid.to(dev), o['weights_smooth'][..., 1:-1].contiguous().to(dev), u, True, True)
    exact(inds, torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True), 'sample_pdf indices')
    close(smp, o['t_fine'].flatten(0, 2), 1e-5, 'sample_pdf samples')


def test_merge_and_composite(case):
    name, meta, t, o, dev = case
    d = lambda k: o[k].to(dev)
    if meta['fine']:
        rgb, dep, msk, _, taps = ops.composite(d('rd'), d('t_coarse'), d('sigma_coarse'), d('rgb_coarse'), d('t_fine'),
                                               d('sigma_fine'), d('rgb_fine'), white_background=meta['white'],
                                               want_taps=True)
        exact(taps['depth_sorted'], o['t_sorted'], 'sorted depths')
        ties = (o['t_sorted'][..., 1:] == o['t_sorted'][..., :-1]).any(-1)
        exact(taps['perm'].cpu()[~ties], o['perm'][~ties], 'sort permutation (tie-free rays)')
    else:
        rgb, dep, msk, _, taps = ops.composite(d('rd'), d('t_coarse'), d('sigma_coarse'), d('rgb_coarse'),
                                               white_background=meta['white'], want_taps=True)
    close(taps['weights'], o['weights'], 1e-6, 'weights')
    close(rgb, o['rgb'], 1e-5, 'rgb map'); close(dep, o['depth'], 1e-5, 'depth map'); close(msk, o['mask'], 1e-5, 'mask')


def test_composite_extra_attribute(gpu_device):
    """semantics compositing through the extra-attribute slot, against the oracle."""
    meta, t = load_golden('persp_white_fine_rand')
    o = oracle_render(meta, t, 'cpu')
    dev = gpu_device
    B, S, A = meta['B'], meta['S'], meta['A']
    x_c = orc.points_on_rays(o['ro'], o['rd'], o['t_coarse'])
    x_f = orc.points_on_rays(o['ro'], o['rd'], o['t_fine'])
    f = lambda x: orc.field_query(t['planes'], t['w1'], t['b1'], t['w2'], t['b2'], x, meta['scene_range'], True, t['beta'],
                                  t['alpha'], t['attention_values'])['semantics'].view(*x.shape[:-1], A)
    sem_c, sem_f = f(x_c), f(x_f)
    d = lambda k: o[k].to(dev)
    _, _, _, sem_map, _ = ops.composite(d('rd'), d('t_coarse'), d('sigma_coarse'), d('rgb_coarse'), d('t_fine'),
                                        d('sigma_fine'), d('rgb_fine'), extra_a=sem_c.to(dev), extra_b=sem_f.to(dev),
                                        white_background=True)
    close(sem_map, t['ref_semantics'], 1e-5, 'semantic map')


def staged_render(meta, t, dev):
    """One pass of more than 128 samples (no fine sampling): the stage kernels through the nerf_utils-level ops."""
    g = lambda k: t[k].to(dev) if k in t else None
    texels, image = hip_field_setup(meta, t, dev)
    ro, rd = ops.raygen(meta['H'], meta['W'], g('focal'), g('cam2world'), g('bbox'), g('center'), normalize=True)
    near, far, hit = ops.near_far(ro, rd, meta['scene_range'])
    pts, dep = ops.stratified_points(ro, rd, near, far, meta['S'], g('noise_coarse'))
    q = ops.field_query(pts.reshape(meta['B'], -1, 3), texels, image, meta['scene_range'], meta['A'], g('attention_values'),
                        meta['sdf'], g('beta'), g('alpha'))
    shp = dep.shape
    rgb, depth, mask, _, taps = ops.composite(rd, dep, q['sigma'].view(*shp), q['rgb'].view(*shp, 3),
                                              white_background=meta['white'], want_taps=True)
    w = ops.ray_weights(q['sigma'].view(*shp), rd, dep)
    return dict(rgb=rgb, depth=depth, mask=mask, t_coarse=dep, sigma_coarse=q['sigma'].view(*shp), weights=taps['weights'],
                hit=hit, ray_weights=w)


def test_single_pass_beyond_128_samples(gpu_device):
    """512 samples in one pass (run.py without --fine_sampling, inversion: ray_multiplier 4, run.py:2271): the stage
    kernels and the fused single-pass kernel (render_fwd_long_kernel), with its taps, the training stash and the
    skipped rays."""
    meta, t = load_golden('persp_s512_coarse_only_rand')
    o = oracle_render(meta, t, 'cpu')
    r = staged_render(meta, t, gpu_device)
    exact(r['t_coarse'], o['t_coarse'], 'depths')
    sigma_close(r['sigma_coarse'], o['sigma_coarse'], 'sigma')
    close(r['weights'], o['weights'], 1e-6, 'weights')
    close(r['ray_weights'], o['weights'], 1e-6, 'ray_weights kernel')
    for k in ('rgb', 'depth', 'mask'):
        close(r[k], o[k], 1e-5, k)
        close(r[k], t['ref_' + k], 1e-5, k + ' vs committed reference output')
    f = hip_render(meta, t, gpu_device, taps=ops.TAP_NAMES)
    exact(f['t_coarse'], o['t_coarse'], 'fused depths')
    sigma_close(f['sigma_coarse'], o['sigma_coarse'], 'fused sigma')
    close(f['weights'], o['weights'], 1e-6, 'fused weights')
    exact(f['t_sorted'], f['t_coarse'], 'a single pass is composited in the order given')
    for k in ('rgb', 'depth', 'mask'):
        close(f[k], o[k], ATOL, 'fused ' + k)
        close(f[k], t['ref_' + k], ATOL, 'fused %s vs committed reference output' % k)
    # the plain launch (split-fp16 decoder arithmetic like every fused inference launch), rays that miss the cube skipped
    p = hip_render(meta, t, gpu_device, skip_missed_rays=True)
    for k in ('rgb', 'depth', 'mask'):
        close(p[k], f[k], 1e-5, 'skip_missed_rays ' + k)
    # the training stash is the tapped per-sample state, ray-major
    st = hip_render(meta, t, gpu_device, stash=True, skip_missed_rays=True)
    hit = (f['hit'] & 2).bool().to(gpu_device)
    exact(st['stash_t'][hit], f['t_coarse'][hit], 'stash depths')
    exact(st['stash_sigma'][hit], f['sigma_coarse'][hit], 'stash sigma')
    exact(st['stash_rgb'][hit], f['rgb_coarse'][hit], 'stash rgb')
    assert float(st['stash_sigma'][~hit].abs().sum()) == 0.0
    exact(st['rgb'], p['rgb'], 'stash launch image')
    # what stays out of the single-pass kernel is refused, not mis-rendered
    with pytest.raises(RuntimeError):
        hip_render(meta, t, gpu_device, want_coords=True)
    m2 = dict(meta, fine=True)
    with pytest.raises(RuntimeError):
        hip_render(m2, t, gpu_device)                    # with fine sampling a pass holds at most 128 samples


def test_fused_render(case):
    name, meta, t, o, dev = case
    r = hip_render(meta, t, dev, taps=ops.TAP_NAMES)
    for k in ('rgb', 'depth', 'mask'):
        close(r[k], o[k], ATOL, 'fused ' + k)
        close(r[k], t['ref_' + k], ATOL, 'fused %s vs committed reference output' % k)
    close(r['t_coarse'], o['t_coarse'], 1e-5, 't_coarse')
    sigma_close(r['sigma_coarse'], o['sigma_coarse'], 'sigma_coarse')
    exact((r['hit'] & 1).bool(), o['hit'], 'hit mask')
    if meta['fine']:
        close(r['t_fine'], o['t_fine'], 1e-4, 't_fine')
        close(r['t_sorted'], o['t_sorted'], 1e-4, 't_sorted')
        mism = (r['perm'].cpu().long() != o['perm']).float().mean().item()
        assert mism <= 2e-4, ('sort permutation flip budget (measured: 0 on every golden case)', mism)
    # skipping rays that miss the cube is exact
    r2 = hip_render(meta, t, dev, skip_missed_rays=True)
    for k in ('rgb', 'depth', 'mask'):
        exact(r2[k], r[k], 'skip_missed_rays ' + k)


@pytest.mark.parametrize('name', ['persp_s96_black_fine_det', 'persp_s128_fine_rand'])
def test_wide_kernel_semantics_table_precision(gpu_device, name):
    """The 128 + 128 kernel (64 < S <= 128: render_fwd_wide_kernel) parks the per-sample softmax probabilities as unorm16
    (tile_epilogue<SEMP < 0>: round to nearest, |error| <= 2^-17 = 7.7e-6 per sample, probabilities under 7.6e-6 become 0)
    where the 64 + 64 kernel keeps fp32 - the composited map is a convex combination of them (weights sum to the mask <= 1),
    so its error is bounded by the same 7.7e-6 plus the fp32 kernel's own ~1e-6.  Explicit bound for the wide kernel: 1e-5
    against the oracle AND the committed reference output, every pixel's map sums to the mask within 2e-5 (A = 10 roundings
    of either sign), and rgb / depth / mask are untouched by the table's format."""
    meta, t = load_golden(name)
    assert meta['S'] > 64 and meta['A'] > 0
    o = oracle_render(meta, t, 'cpu')
    plain = hip_render(meta, t, gpu_device, skip_missed_rays=True)
    r = hip_render(meta, t, gpu_device, skip_missed_rays=True, want_semantics=True)
    for k in ('rgb', 'depth', 'mask'):
        exact(r[k], plain[k], 'semantics launch (wide kernel), %s' % k)
    e = err(r['semantics'], o['semantics'])
    assert e['nonfinite'] == 0 and e['max'] <= 1e-5, ('semantic map, wide kernel (unorm16 table)', e)     # measured <= 4e-6
    close(r['semantics'], t['ref_semantics'], 1e-5, 'semantic map vs committed reference output')
    close(r['semantics'].sum(-1), r['mask'], 2e-5, 'sum of the semantic map = mask')
    assert float(r['semantics'].min()) >= 0.0


def test_fused_render_extra_maps(case):
    """compute_semantics / compute_coords inside the fused kernel (run.py:312-338, lib/nerf_utils.py:147-159): the maps
    come out of the SAME launch, rgb / depth / mask stay bit-identical to the plain render, the maps match the oracle
    and the committed reference output."""
    name, meta, t, o, dev = case
    vd = 'viewdir_x' in t          # --use_viewdir (carla): semantics / coords from the same launch too (fp32 texels), normals staged
    plain = hip_render(meta, t, dev, skip_missed_rays=True)
    want_sem = meta['A'] > 0 and not meta.get('coords')
    for texel_dtype in ((ops.TEXEL_F32,) if vd else (ops.TEXEL_F32, ops.TEXEL_F16)):
        base = plain if texel_dtype == ops.TEXEL_F32 else hip_render(meta, t, dev, skip_missed_rays=True, texel_dtype=texel_dtype)
        r = hip_render(meta, t, dev, skip_missed_rays=True, texel_dtype=texel_dtype, want_semantics=want_sem, want_coords=True)
        for k in ('rgb', 'depth', 'mask'):
            exact(r[k], base[k], 'extra-map launch, %s' % k)
        if want_sem:
            assert r['semantics'].shape == (meta['B'], meta['H'], meta['W'], meta['A'])
            # probabilities composited with the weights: every pixel's map sums to the mask
            close(r['semantics'].sum(-1), r['mask'], 1e-5, 'sum of the semantic map = mask')
        if texel_dtype != ops.TEXEL_F32:
            continue                      # (16-bit planes: the oracle comparison belongs to the fp32 storage)
        if want_sem:
            close(r['semantics'], o['semantics'], 1e-5, 'semantic map')
            close(r['semantics'], t['ref_semantics'], 1e-5, 'semantic map vs committed reference output')
        # the coords map against the oracle's own compositing of its query points
        w, ts = o['weights'], (o['t_sorted'] if meta['fine'] else o['t_coarse'])
        pts = o['ro'].unsqueeze(-2) + o['rd'].unsqueeze(-2) * ts.unsqueeze(-1)
        close(r['coords'], (w.unsqueeze(-1) * pts).sum(-2), 1e-5, 'coords map')
        if meta.get('coords'):
            close(r['coords'], o['semantics'], 1e-5, 'coords map (oracle render)')
            close(r['coords'], t['ref_coords_map'], 1e-5, 'coords map vs committed reference output')
    if vd:
        if meta['sdf']:
            # the normal map with the view-direction decoder in the SAME launch since round 6 (the distance is row 0 of its
            # second layer; run.py:1444-1454 on carla): against autograd of the oracle's distance, like the plain decoder
            rn = hip_render(meta, t, dev, skip_missed_rays=True, want_normals=True, want_semantics=want_sem, want_coords=True)
            for k in ('rgb', 'depth', 'mask'):
                exact(rn[k], plain[k], 'normal-map launch (view-direction decoder), %s' % k)
            close(rn['normals'], oracle_normal_map(meta, t, o), 3e-5, 'normal map, view-direction decoder')
        with pytest.raises(RuntimeError):
            hip_render(meta, t, dev, skip_missed_rays=True, texel_dtype=ops.TEXEL_F16, want_coords=True)
        return
    if meta['sdf']:
        # the normal map (compute_normals): the kernel's analytic d sdf / d x against autograd of the oracle's distance,
        # composited with the oracle's weights (lib/nerf_utils.py:149-151, 159); unit vectors from fp32 texel differences
        ref_map = oracle_normal_map(meta, t, o)
        rn = hip_render(meta, t, dev, skip_missed_rays=True, want_normals=True, want_semantics=want_sem, want_coords=True)
        for k in ('rgb', 'depth', 'mask'):
            exact(rn[k], plain[k], 'normal-map launch, %s' % k)
        close(rn['normals'], ref_map, 3e-5, 'normal map')                     # measured <= 1.2e-5 (profiles/r5/parity_report.json)
        if want_sem:
            close(rn['semantics'], o['semantics'], 1e-5, 'semantic map next to the normals')
        # 16-bit texel STORAGE (fp16, and since round 6 bf16 in the fused kernel too): parity is against the oracle on the
        # SAME rounded planes - a unit vector made of differences of rounded texels has nothing to do with the unrounded
        # planes' normal (that comparison, mean 2.6e-4 / max 0.06, was a sanity check, not parity)
        for tdt, torch_dt in ((ops.TEXEL_F16, torch.float16), (ops.TEXEL_BF16, torch.bfloat16)):
            t16 = dict(t, planes=t['planes'].to(torch_dt).to(torch.float32))
            o16 = oracle_render(meta, t16, 'cpu')
            ref16 = oracle_normal_map(meta, t16, o16)
            rn16 = hip_render(meta, t, dev, skip_missed_rays=True, texel_dtype=tdt, want_normals=True)
            base16 = hip_render(meta, t, dev, skip_missed_rays=True, texel_dtype=tdt)
            for k in ('rgb', 'depth', 'mask'):
                exact(rn16[k], base16[k], 'normal-map launch on 16-bit texels, %s' % k)
            close(rn16['normals'], ref16, 6e-5, 'normal map, %s planes, oracle on the same rounded planes' % torch_dt)
    else:
        with pytest.raises((RuntimeError, ValueError)):
            hip_render(meta, t, dev, skip_missed_rays=True, want_normals=True)       # normals need the SDF decoder
    if meta['fine'] and meta['sdf'] and meta['A'] > 0:
        # the same maps from a SINGLE pass (run.py without --fine_sampling): no merge, the weights are in source order already
        m1 = dict(meta, fine=False)
        o1 = oracle_render(dict(m1, coords=False), t, 'cpu')
        r1p = hip_render(m1, t, dev, skip_missed_rays=True)
        r1m = hip_render(m1, t, dev, skip_missed_rays=True, want_semantics=True, want_coords=True, want_normals=True)
        for k in ('rgb', 'depth', 'mask'):
            exact(r1m[k], r1p[k], 'single pass, extra-map launch, %s' % k)
            close(r1m[k], o1[k], ATOL, 'single pass ' + k)
        close(r1m['semantics'], o1['semantics'], 1e-5, 'single pass semantic map')
        pts1 = o1['ro'].unsqueeze(-2) + o1['rd'].unsqueeze(-2) * o1['t_coarse'].unsqueeze(-1)
        close(r1m['coords'], (o1['weights'].unsqueeze(-1) * pts1).sum(-2), 1e-5, 'single pass coords map')
        ref1 = oracle_normal_map(m1, t, o1)
        close(r1m['normals'], ref1, 3e-5, 'single pass normal map')
    # evaluating every ray instead of skipping the missed ones changes nothing (their weights are exactly 0)
    r0 = hip_render(meta, t, dev, skip_missed_rays=False, want_semantics=want_sem, want_coords=True)
    r1 = hip_render(meta, t, dev, skip_missed_rays=True, want_semantics=want_sem, want_coords=True)
    for k in ('coords',) + (('semantics',) if want_sem else ()):
        close(r0[k], r1[k], 1e-6, 'skip_missed_rays ' + k)


def test_fused_render_bf16_texels(case):
    """bf16 plane storage (BASELINE config 2 variant): parity against the fp32 reference at the
    tolerance bf16 rounding of the planes allows."""
    name, meta, t, o, dev = case
    r = hip_render(meta, t, dev, texel_dtype=ops.TEXEL_BF16)
    close(r['mask'], o['mask'], 0.08, 'bf16 mask')
    close(r['rgb'], o['rgb'], 0.08, 'bf16 rgb')
    assert err(r['rgb'], o['rgb'])['mean'] < 5e-3


@pytest.mark.parametrize('dtype,torch_dtype', [(ops.TEXEL_BF16, torch.bfloat16), (ops.TEXEL_F16, torch.float16)])
def test_fused_render_half_texels_against_rounded_planes(case, dtype, torch_dtype):
    """16-bit plane storage (BASELINE cfg2 bf16 / cfg5 fp16): the kernel must equal the fp32 reference
    evaluated on the SAME rounded planes at the fp32 tolerance, i.e. the storage rounding of the
    planes is the only deviation (arithmetic stays fp32)."""
    name, meta, t, o, dev = case
    r = hip_render(meta, t, dev, texel_dtype=dtype)
    t2 = dict(t)
    t2['planes'] = t['planes'].to(torch_dtype).to(torch.float32)
    o2 = oracle_render(meta, t2)
    tol = 2e-4
    close(r['mask'], o2['mask'], tol, 'mask')
    close(r['rgb'], o2['rgb'], tol, 'rgb')
