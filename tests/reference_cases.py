"""The LIVE reference (real models/generator.py::Generator incl. its StyleGAN2 plane producer, run.py::render AST-sliced)
beside the HIP drop-in on the same device, same weights, same cameras, same noise.  Used by tests/test_reference_gpu.py,
tools/reference_report.py and bench.py's reference legs (never by the product, never inside a timed region of the HIP
path).  The reference sources come from oracle/reference.py (the checkout, or the copy oracle/make_ref.py staged)."""
import contextlib
import copy
import math
import types

import torch

from oracle import reference

# geometry of the run.py data sets the BASELINE configurations name; all with the SDF decoder, A = 10
GEOMETRY = {
    # shapenet_chairs (cfg2): perspective, focal 131.25 / 128, camera radius 2.0, white background
    'chairs': dict(scene_range=0.55, white=True, radius=2.0, focal=1.0254, bbox=False),
    # p3d_car (cfg3): perspective, focal 1, camera distance 2, black background, eval renders pass a crop bbox
    'p3d': dict(scene_range=1.4, white=False, radius=2.0, focal=1.0, bbox=True),
    # cub (cfg4): orthographic (focal None), black background
    'cub': dict(scene_range=2.0, white=False, radius=3.0, focal=None, bbox=False),
    # carla (--use_viewdir): perspective, white background, the view-direction decoder
    'carla': dict(scene_range=1.0, white=True, radius=2.2, focal=1.2, bbox=False, viewdir=True),
}


def cameras(n, radius, gen, ortho=False):
    """Cameras on a sphere looking at the origin (OpenGL convention, what get_ray_bundle expects)."""
    v = torch.randn(n, 3, generator=gen)
    eye = radius * v / v.norm(dim=-1, keepdim=True)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    up = torch.tensor([0., 0., 1.]).expand(n, 3)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    tup = torch.cross(right, fwd, dim=-1)
    cam = torch.eye(4).repeat(n, 1, 1)
    cam[:, :3, 0], cam[:, :3, 1], cam[:, :3, 2], cam[:, :3, 3] = right, tup, -fwd, eye
    if ortho:
        cam[:, 3, 3] = 1.0 + 0.2 * torch.rand(n, generator=gen)        # the ortho zoom run.py keeps in cam[3,3]
    return cam


def crop_boxes(n, gen):
    """bbox [B,2,2] as data/datasets.py:318-340 builds it: row 0 = (x, y) start of the square crop in [-1, 1] with y
    flipped, row 1 = extent x 2."""
    size = 1.2 + 0.6 * torch.rand(n, 1, generator=gen)
    start = -0.5 * size + 0.2 * (torch.rand(n, 2, generator=gen) - 0.5)
    return torch.stack([start, size.expand(n, 2)], dim=1).contiguous()


def build_scene(geometry, batch, dev, seed=1234, alpha=0.05, beta=0.1):
    """A default-initialised reference Generator with its SDF centred so that it renders surfaces (a random-init
    generator renders an almost empty scene: SURVEY.md 8(d)), its twin with the HIP sampler attached, and seeded
    cameras / latents.  Returns a namespace."""
    import nerf_from_image_amd.generator as nfi_gen
    g = GEOMETRY[geometry]
    m = reference.modules()
    vd = bool(g.get('viewdir'))
    torch.manual_seed(seed)
    gen = m.generator.Generator(512, g['scene_range'], attention_values=10, use_viewdir=vd, use_sdf=True,
                                disable_stylegan_noise=True)
    cpu = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        gen.alpha.fill_(alpha)
        gen.beta.fill_(beta)
        if vd:
            # the mapper's output layer is zero-initialised (generator.py:217-219): give it weights, or every colour
            # would be the same constant in both implementations
            gen.viewdir_mapper.output.weight.copy_(0.5 * torch.randn(gen.viewdir_mapper.output.weight.shape, generator=cpu))
            gen.viewdir_mapper.output.bias.copy_(0.1 * torch.randn(gen.viewdir_mapper.output.bias.shape, generator=cpu))
    gen = gen.to(dev).eval().requires_grad_(False)
    z = torch.randn(batch, 512, generator=cpu).to(dev)
    with torch.no_grad():
        ws = gen.mapping_network(z, None)
        # centre the distance output: shift its bias by the lower quartile of the SDF over the cube (all scenes of the
        # batch): a quarter of the volume is inside a surface
        pts = ((torch.rand(batch, 20000, 3, generator=cpu) * 2 - 1) * g['scene_range']).to(dev)
        if vd:
            out = gen(torch.zeros(batch, 1, 1, 1, 3, device=dev), ws, ['sampler'])
            sdf = out['sampler'](pts.view(batch, 1, 1, -1, 3), ['sdf_distance'])['sdf_distance']
        else:
            sdf = gen(None, ws, ['sampler'])['sampler'](pts, ['sdf_distance'])['sdf_distance']
        gen.decoder.net[2].bias[0] -= sdf.flatten().quantile(0.25)
    hip = nfi_gen.attach(copy.deepcopy(gen))
    ortho = g['focal'] is None
    cam = cameras(batch, g['radius'], cpu, ortho).to(dev)
    focal = None if ortho else torch.full((batch,), g['focal']).to(dev)
    bbox = crop_boxes(batch, cpu).to(dev) if g['bbox'] else None
    args = reference.render_args(fine_sampling=True, use_sdf=True, attention_values=10, use_viewdir=vd)
    dcfg = {'scene_range': g['scene_range'], 'white_background': g['white']}
    return types.SimpleNamespace(geometry=geometry, g=g, gen=gen, hip=hip, z=z, ws=ws, cam=cam, focal=focal, bbox=bbox,
                                 args=args, dcfg=dcfg, batch=batch, dev=dev)


class ReplayNoise:
    """torch.rand / torch.rand_like hand out pre-drawn tensors in call order (render draws [B,H,W,S] for the stratified
    jitter, then [B*H*W,S] for the inverse-CDF samples: lib/nerf_utils.py:115, 202), on whatever device is asked."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self._rand, self._rand_like = torch.rand, torch.rand_like
        self.i = 0

        def take(shape, device):
            d = self.draws[self.i]
            assert tuple(d.shape) == tuple(shape), ('draw %d' % self.i, tuple(d.shape), tuple(shape))
            self.i += 1
            return d.to(device).clone()

        def rand(*size, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else size
            return take(size, kw.get('device', 'cpu'))

        def rand_like(t, **kw):
            return take(t.shape, t.device)
        torch.rand, torch.rand_like = rand, rand_like
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like = self._rand, self._rand_like


def draw_noise(sc, res, samples, seed=99):
    gn = torch.Generator(device=sc.dev).manual_seed(seed)
    return [torch.rand((sc.batch, res, res, samples), device=sc.dev, generator=gn),
            torch.rand((sc.batch * res * res, samples), device=sc.dev, generator=gn)]


@contextlib.contextmanager
def frozen_producer(gen, planes96):
    """The plane producer of `gen` returns `planes96` ([B,96,R,R]) instead of running: lets a CPU copy of the reference
    render the very planes the GPU produced (MIOpen and the CPU convolutions differ in the last bits)."""
    orig = gen.synthesis_network.forward
    gen.synthesis_network.forward = lambda ws, **kw: planes96
    try:
        yield
    finally:
        gen.synthesis_network.forward = orig


def reference_render(sc, res, samples, noise, device=None, images=None, grad=False, **render_kw):
    """run.py::render on the real Generator.  device='cpu': a CPU copy fed with the planes / colour table the GPU
    produced; images: slice of the batch.  Returns the 6-tuple."""
    ren, _ = reference.load_render(sc.args, sc.dcfg, unscripted_stages=noise is not None)
    sl = slice(None) if images is None else images
    pick = (lambda t: None if t is None else t[sl])
    gen, ws, cam, focal, bbox = sc.gen, sc.ws[sl], sc.cam[sl], pick(sc.focal), pick(sc.bbox)
    n = cam.shape[0]
    nz = None if noise is None else [noise[0][sl], noise[1].view(sc.batch, -1, samples)[sl].reshape(-1, samples)]
    ctx = contextlib.nullcontext()
    extra_in = {}
    if device is not None and torch.device(device) != cam.device:
        with torch.no_grad():
            planes = sc.gen.synthesis_network(ws[:, :14]).cpu()
            extra_in = {'attention_values': _attention(sc.gen, ws).cpu()}
        gen = copy.deepcopy(sc.gen).to(device)
        ws, cam, focal, bbox = ws.to(device), cam.to(device), pick_to(focal, device), pick_to(bbox, device)
        ctx = frozen_producer(gen, planes)
    with ctx, (ReplayNoise(nz) if nz is not None else contextlib.nullcontext()), \
            (contextlib.nullcontext() if grad else torch.no_grad()):
        return ren(gen, res, res, cam, focal, None, bbox, ws, samples, extra_model_inputs=extra_in, **render_kw)


def pick_to(t, device):
    return None if t is None else t.to(device)


def _attention(gen, ws):
    """The colour table Generator.forward computes (models/generator.py:452-464)."""
    return gen(None, ws, ['attention_values', 'sampler'])['attention_values']


def hip_render(sc, res, samples, noise, grad=False, ws=None, cam=None, focal=None, **render_kw):
    import nerf_from_image_amd.render as nfi_render
    ren = nfi_render.make_render(sc.args, sc.dcfg)
    with (ReplayNoise(noise) if noise is not None else contextlib.nullcontext()), \
            (contextlib.nullcontext() if grad else torch.no_grad()):
        return ren(sc.hip, res, res, sc.cam if cam is None else cam, sc.focal if focal is None else focal, None, sc.bbox,
                   sc.ws if ws is None else ws, samples, **render_kw)


def max_err(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def compare(sc, res, samples, cpu_images=2, **render_kw):
    """HIP vs the reference on this GPU (whole batch) and vs the reference on the CPU (first `cpu_images` images, same
    planes).  Returns a dict of max |error| per output and the CPU-vs-GPU gap of the reference itself."""
    noise = draw_noise(sc, res, samples)
    ours = hip_render(sc, res, samples, noise, **render_kw)
    ref_gpu = reference_render(sc, res, samples, noise, **render_kw)
    names = ['rgb', 'depth', 'mask', 'normals', 'extra']
    rep = {'mask_mean': float(ref_gpu[2].mean()), 'vs_reference_gpu': {}, 'vs_reference_cpu': {}, 'reference_cpu_vs_gpu_gap': {},
           'pixels_over_1e-4_vs_reference_gpu': {}}
    for k, a, b in zip(names, ours[:5], ref_gpu[:5]):
        if a is None and b is None:
            continue
        assert a is not None and b is not None and a.shape == b.shape, (k, None if a is None else a.shape, None if b is None else b.shape)
        rep['vs_reference_gpu'][k] = max_err(a, b)
        rep['pixels_over_1e-4_vs_reference_gpu'][k] = int(((a - b).abs() > 1e-4).sum())
    if cpu_images:
        sl = slice(0, min(cpu_images, sc.batch))
        ref_cpu = reference_render(sc, res, samples, noise, device='cpu', images=sl, **render_kw)
        for k, a, b, c in zip(names, ours[:5], ref_cpu[:5], ref_gpu[:5]):
            if a is None:
                continue
            rep['vs_reference_cpu'][k] = max_err(a[sl], b)
            rep['reference_cpu_vs_gpu_gap'][k] = max_err(c[sl], b)
    return rep


def gradients(sc, res, samples, seed=5):
    """Forward + backward of  sum(rgb * w_rgb) + sum(mask * w_mask)  w.r.t. the latents ws, the camera matrix and the
    focal length, in both implementations (same noise).  Returns relative L2 errors of the gradients."""
    noise = draw_noise(sc, res, samples)
    gw = torch.Generator(device=sc.dev).manual_seed(seed)
    w_rgb = torch.randn((sc.batch, res, res, 3), device=sc.dev, generator=gw)
    w_mask = torch.randn((sc.batch, res, res), device=sc.dev, generator=gw)

    def leaves():
        ws = sc.ws.detach().clone().requires_grad_()
        cam = sc.cam.detach().clone().requires_grad_()
        focal = None if sc.focal is None else sc.focal.detach().clone().requires_grad_()
        return ws, cam, focal

    def run(which):
        ws, cam, focal = leaves()
        if which == 'hip':
            out = hip_render(sc, res, samples, noise, grad=True, ws=ws, cam=cam, focal=focal)
        else:
            keep = sc.ws, sc.cam, sc.focal
            sc.ws, sc.cam, sc.focal = ws, cam, focal
            try:
                out = reference_render(sc, res, samples, noise, grad=True)
            finally:
                sc.ws, sc.cam, sc.focal = keep
        loss = (out[0] * w_rgb).sum() + (out[2] * w_mask).sum()
        loss.backward()
        return float(loss.detach()), ws.grad, cam.grad, None if focal is None else focal.grad
    for mod in (sc.gen, sc.hip):
        mod.requires_grad_(False)
    l_h, gws_h, gcam_h, gf_h = run('hip')
    l_r, gws_r, gcam_r, gf_r = run('ref')
    rep = {'loss_hip': l_h, 'loss_reference': l_r, 'g_ws': rel_err(gws_h, gws_r), 'g_cam': rel_err(gcam_h[:, :3], gcam_r[:, :3])}
    if gf_h is not None:
        rep['g_focal'] = rel_err(gf_h, gf_r)
    return rep


def generator_from_tensors(planes, w1, b1, w2, b2, beta, alpha, scene_range, device):
    """The real reference Generator carrying GIVEN field tensors (bench.py's synthetic workload): decoder weights copied
    into its TriplanarDecoder (raw EqualizedLinear weights - the class applies the 1/sqrt(fan_in) gains itself), beta /
    alpha set, and its plane producer frozen to `planes` [B,3,32,R,R] ("render only": the StyleGAN2 synthesis is not part
    of the rendered-rays metric).  Call it with ws = dummy_ws(B) and extra_model_inputs={'attention_values': att}."""
    m = reference.modules()
    A = w2.shape[0] - 1
    gen = m.generator.Generator(512, scene_range, attention_values=A, use_sdf=True, disable_stylegan_noise=True)
    with torch.no_grad():
        gen.decoder.net[0].weight.copy_(w1)
        gen.decoder.net[0].bias.copy_(b1)
        gen.decoder.net[2].weight.copy_(w2)
        gen.decoder.net[2].bias.copy_(b2)
        gen.beta.copy_(beta.view(1))
        gen.alpha.copy_(alpha.view(1))
    gen = gen.to(device).eval().requires_grad_(False)
    planes96 = planes.to(device).reshape(planes.shape[0], 96, planes.shape[-2], planes.shape[-1])
    gen.synthesis_network.forward = lambda ws, **kw: planes96[:ws.shape[0]]
    return gen


def dummy_ws(batch, device):
    return torch.zeros(batch, 15, 512, device=device)


def _psnr_iou(rgb, mask, t_rgb, t_mask):
    """lib/metrics.py psnr (images in [-1, 1]: peak-to-peak 2) and iou of the thresholded masks, batch means."""
    mse = ((rgb - t_rgb) ** 2).flatten(1).mean(1) / 4.0
    psnr = float((-10.0 * torch.log10(mse.clamp_min(1e-10))).mean())
    a, b = mask > 0.5, t_mask > 0.5
    iou = float(((a & b).flatten(1).sum(1).float() / (a | b).flatten(1).sum(1).clamp_min(1).float()).mean())
    return psnr, iou


def inversion(sc, res, samples, steps=8, lr=2e-3, seed=11):
    """The --run_inversion loop's shape (run.py:2232-2299) on the REAL Generator with a synthetic target: the target is the
    reference's own render of the scene; latents and camera start perturbed; Adam(lr, betas 0.9 / 0.95, run.py:2007) on
    ws + camera matrix (+ focal) for `steps` steps, fresh noise every step.

    Adam normalises the update, so free-running trajectories are chaotic; what is compared without chaos is, at every
    point of the REFERENCE's trajectory, the loss and the gradients of the HIP path at the same parameters and noise.
    The HIP path also runs its own trajectory (its own Adam), for the PSNR / IoU figures.  Returns a dict."""
    g = torch.Generator().manual_seed(seed)
    dev = sc.dev
    with torch.no_grad():
        tgt = reference_render(sc, res, samples, draw_noise(sc, res, samples, seed=1000))
    t_rgb, t_mask = tgt[0].detach(), tgt[2].detach()
    ws0 = sc.ws + 0.25 * torch.randn(sc.ws.shape, generator=g).to(dev) * sc.ws.std()
    cam0 = sc.cam.clone()
    cam0[:, :3, 3] += 0.03 * torch.randn(sc.batch, 3, generator=g).to(dev)
    focal0 = None if sc.focal is None else sc.focal * (1.0 + 0.02 * torch.randn(sc.batch, generator=g).to(dev))
    for mod in (sc.gen, sc.hip):
        mod.requires_grad_(False)

    def make_params():
        p = [ws0.clone().requires_grad_(), cam0.clone().requires_grad_()]
        if focal0 is not None:
            p.append(focal0.clone().requires_grad_())
        return p

    def forward(which, params, noise):
        ws, cam = params[0], params[1]
        focal = params[2] if len(params) > 2 else None
        if which == 'hip':
            out = hip_render(sc, res, samples, noise, grad=True, ws=ws, cam=cam, focal=focal)
        else:
            keep = sc.ws, sc.cam, sc.focal
            sc.ws, sc.cam, sc.focal = ws, cam, focal
            try:
                out = reference_render(sc, res, samples, noise, grad=True)
            finally:
                sc.ws, sc.cam, sc.focal = keep
        loss = ((out[0] - t_rgb) ** 2).mean() + ((out[2] - t_mask) ** 2).mean()
        return loss, out

    def grads_of(which, params, noise):
        for p in params:
            p.grad = None
        loss, out = forward(which, params, noise)
        loss.backward()
        return float(loss.detach()), [p.grad.clone() for p in params], _psnr_iou(out[0].detach(), out[2].detach(), t_rgb, t_mask)

    p_ref, p_hip = make_params(), make_params()
    opt_ref = torch.optim.Adam(p_ref, lr=lr, betas=(0.9, 0.95))
    opt_hip = torch.optim.Adam(p_hip, lr=lr, betas=(0.9, 0.95))
    along, traj_ref, traj_hip = [], [], []
    for step in range(steps):
        noise = draw_noise(sc, res, samples, seed=2000 + step)
        # the HIP path at the reference trajectory's parameters (shadow copies: no optimiser state involved)
        shadow = [p.detach().clone().requires_grad_() for p in p_ref]
        l_h, g_h, _ = grads_of('hip', shadow, noise)
        l_r, g_r, m_r = grads_of('ref', p_ref, noise)
        along.append({'loss_rel': abs(l_h - l_r) / abs(l_r), 'g_ws': rel_err(g_h[0], g_r[0]), 'g_cam': rel_err(g_h[1][:, :3], g_r[1][:, :3]),
                      **({'g_focal': rel_err(g_h[2], g_r[2])} if len(g_r) > 2 else {})})
        for p, gr in zip(p_ref, g_r):
            p.grad = gr
        opt_ref.step()
        traj_ref.append((l_r,) + m_r)
        l_o, g_o, m_o = grads_of('hip', p_hip, noise)
        opt_hip.step()
        traj_hip.append((l_o,) + m_o)
    return {'along_reference_trajectory': along, 'reference': traj_ref, 'hip': traj_hip}


REGULARISER_NAMES = ['sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss']


def regularisers(sc, seed=321, weights=(1.0, 0.7, 3.0, 0.01)):
    """The G step's regulariser branch (generator.py:505-585) on the real Generator in training mode: the reference's own
    forward against `attach(model, hip_regularisers=True)`, same seed (= the same two draws), losses and gradients w.r.t. the
    latents (through the synthesis network), the decoder and beta.  Returns relative errors."""
    import nerf_from_image_amd.generator as nfi_gen

    def run(model):
        model = model.train().requires_grad_(True)
        ws = sc.ws.detach().clone().requires_grad_()
        torch.manual_seed(seed)
        out = model(None, ws, REGULARISER_NAMES)
        assert set(out) == set(REGULARISER_NAMES)
        loss = sum(w * out[n].sum() for w, n in zip(weights, REGULARISER_NAMES))
        dec = model.decoder.net
        grads = torch.autograd.grad(loss, [ws, dec[0].weight, dec[0].bias, dec[2].weight, model.beta])
        return {n: out[n].detach() for n in REGULARISER_NAMES}, grads
    ref_out, ref_g = run(copy.deepcopy(sc.gen))
    hip_out, hip_g = run(nfi_gen.attach(copy.deepcopy(sc.gen), hip_regularisers=True))
    rep = {'loss_rel': {n: max_err(hip_out[n], ref_out[n]) / float(ref_out[n].abs().max()) for n in REGULARISER_NAMES},
           'loss_reference': {n: ref_out[n].tolist() for n in REGULARISER_NAMES},
           'grad_rel_l2': {name: rel_err(a, b) for name, a, b in zip(['ws', 'w1', 'b1', 'w2', 'beta'], hip_g, ref_g)}}
    return rep


def training_step(sc, res, samples, seed=77, reg_weight=0.1):
    """One generator-side training step in cfg4's shape on the REAL Generator (training mode, latents through the mapping
    network): render + image / alpha loss (run.py:980-1010), regulariser forward (974-979, 1011-1028), one backward - the
    reference's own render + forward against the drop-in render + `attach(..., hip_regularisers=True)`, same noise and
    seed.  Compares the loss and the gradient of EVERY generator parameter.  Returns a dict."""
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    noise = draw_noise(sc, res, samples, seed=seed)
    g = torch.Generator().manual_seed(seed)
    t_rgb = (torch.rand(sc.batch, res, res, 3, generator=g) * 2 - 1).to(sc.dev)
    t_mask = (torch.rand(sc.batch, res, res, generator=g) > 0.5).float().to(sc.dev)
    ref_render, _ = reference.load_render(sc.args, sc.dcfg, unscripted_stages=True)
    hip_render_fn = nfi_render.make_render(sc.args, sc.dcfg)

    def run(model, render_fn):
        model = model.train().requires_grad_(True)
        for p_ in model.parameters():
            p_.grad = None
        with ReplayNoise(noise):
            out = render_fn(model, res, res, sc.cam, sc.focal, None, sc.bbox, sc.z, samples)
        loss = ((out[0] - t_rgb) ** 2).mean() + ((out[2] - t_mask) ** 2).mean()
        torch.manual_seed(seed)
        reg = model(None, sc.z, ['sdf_eikonal_loss', 'sdf_distance_loss'])
        loss = loss + reg_weight * (reg['sdf_eikonal_loss'].mean() + reg['sdf_distance_loss'].mean())
        loss.backward()
        return float(loss.detach()), {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
    l_r, g_r = run(copy.deepcopy(sc.gen), ref_render)
    l_h, g_h = run(nfi_gen.attach(copy.deepcopy(sc.gen), hip_regularisers=True), hip_render_fn)
    assert set(g_r) == set(g_h), set(g_r) ^ set(g_h)
    num = sum(float((g_h[n].double() - g_r[n].double()).pow(2).sum()) for n in g_r)
    den = sum(float(g_r[n].double().pow(2).sum()) for n in g_r)
    per = {n: rel_err(g_h[n], g_r[n]) for n in g_r}
    big = {n: e for n, e in per.items() if float(g_r[n].double().norm()) >= 1e-3 * den ** 0.5}
    worst = max(big, key=big.get)
    return {'loss_hip': l_h, 'loss_reference': l_r, 'n_parameter_tensors': len(g_r), 'grad_rel_l2_all_parameters': (num / den) ** 0.5,
            'worst_tensor_among_the_significant': worst, 'worst_tensor_rel_l2': big[worst], 'significant_tensors': len(big)}
