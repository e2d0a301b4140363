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
    # the generator's OTHER branches on the chairs cameras: NeRF density softplus(d - 1) instead of the SDF
    # (models/generator.py:637-641) and the direct colour head wide_sigmoid_rescaled(features), A = 0 (665-666)
    'density': dict(scene_range=0.55, white=True, radius=2.0, focal=1.0254, bbox=False, use_sdf=False, attention_values=0),
    # --use_class (class-conditional data sets): Generator(num_classes=...) with model_input = (z, labels) (generator.py:428-446)
    'classes': dict(scene_range=0.55, white=True, radius=2.0, focal=1.0254, bbox=False, num_classes=5),
    # cub with the crop box its loader passes (orthographic branch of get_ray_bundle WITH bbox, lib/nerf_utils.py:72-77)
    'cub_bbox': dict(scene_range=2.0, white=False, radius=3.0, focal=None, bbox=True),
    # --use_encoder: Generator(use_encoder=True) with model_input = (z, image) (generator.py:357-358, 423-426: ResidualEncoder
    # -> conditional mapping network)
    'encoder': dict(scene_range=0.55, white=True, radius=2.0, focal=1.0254, bbox=False, use_encoder=True),
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


def build_scene(geometry, batch, dev, seed=1234, alpha=0.05, beta=0.1, fine_sampling=True, stylegan_noise=False):
    """A default-initialised reference Generator with its SDF centred so that it renders surfaces (a random-init
    generator renders an almost empty scene: SURVEY.md 8(d)), its twin with the HIP sampler attached, and seeded
    cameras / latents.  Returns a namespace."""
    import nerf_from_image_amd.generator as nfi_gen
    g = GEOMETRY[geometry]
    m = reference.modules()
    vd = bool(g.get('viewdir'))
    use_sdf, n_att = bool(g.get('use_sdf', True)), int(g.get('attention_values', 10))
    torch.manual_seed(seed)
    n_cls = g.get('num_classes')
    enc = bool(g.get('use_encoder'))
    gen = m.generator.Generator(512, g['scene_range'], attention_values=n_att, use_viewdir=vd, use_sdf=use_sdf,
                                disable_stylegan_noise=not stylegan_noise, num_classes=n_cls, use_encoder=enc)
    cpu = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        if use_sdf:
            gen.alpha.fill_(alpha)
            gen.beta.fill_(beta)
        if stylegan_noise:
            # noise_strength is zero-initialised (stylegan.py:322): a trained model's is not, and only then do the per-layer
            # draws reach the planes
            for mod in gen.modules():
                if hasattr(mod, 'noise_strength'):
                    mod.noise_strength.fill_(0.1)
        if vd:
            # the mapper's output layer is zero-initialised (generator.py:217-219): give it weights, or every colour
            # would be the same constant in both implementations
            gen.viewdir_mapper.output.weight.copy_(0.5 * torch.randn(gen.viewdir_mapper.output.weight.shape, generator=cpu))
            gen.viewdir_mapper.output.bias.copy_(0.1 * torch.randn(gen.viewdir_mapper.output.bias.shape, generator=cpu))
    gen = gen.to(dev).eval().requires_grad_(False)
    z = torch.randn(batch, 512, generator=cpu).to(dev)
    labels = torch.randint(n_cls, (batch,), generator=cpu).to(dev) if n_cls else None
    image = (torch.rand(batch, 3, 128, 128, generator=cpu) * 2 - 1).to(dev) if enc else None
    with torch.no_grad():
        ws = gen.mapping_network(z, gen.emb(image) if enc else (gen.class_embedding(labels) if n_cls else None))
        # centre the distance output: shift its bias by the lower quartile of the SDF over the cube (all scenes of the
        # batch): a quarter of the volume is inside a surface
        pts = ((torch.rand(batch, 20000, 3, generator=cpu) * 2 - 1) * g['scene_range']).to(dev)
        if vd:
            out = gen(torch.zeros(batch, 1, 1, 1, 3, device=dev), ws, ['sampler'])
            sdf = out['sampler'](pts.view(batch, 1, 1, -1, 3), ['sdf_distance'])['sdf_distance']
        else:
            sdf = gen(None, (z, image) if enc else ws, ['sampler'])['sampler'](pts, ['sdf_distance'])['sdf_distance']
        if use_sdf:
            gen.decoder.net[2].bias[0] -= sdf.flatten().quantile(0.25)
        else:
            # density branch, sigma = softplus(d - 1): half of the volume well above 1, the rest well below
            # (d' = k (d - median) + 1: the output row's weight times k, its bias moved accordingly)
            k, q75 = 20.0, sdf.flatten().quantile(0.5)
            gen.decoder.net[2].weight[0] *= k
            gen.decoder.net[2].bias[0] = k * (gen.decoder.net[2].bias[0] - q75) + 1.0
    hip = nfi_gen.attach(copy.deepcopy(gen))
    ortho = g['focal'] is None
    cam = cameras(batch, g['radius'], cpu, ortho).to(dev)
    focal = None if ortho else torch.full((batch,), g['focal']).to(dev)
    bbox = crop_boxes(batch, cpu).to(dev) if g['bbox'] else None
    args = reference.render_args(fine_sampling=fine_sampling, use_sdf=use_sdf, attention_values=n_att, use_viewdir=vd)
    dcfg = {'scene_range': g['scene_range'], 'white_background': g['white']}
    return types.SimpleNamespace(geometry=geometry, g=g, gen=gen, hip=hip, z=z, ws=ws, cam=cam, focal=focal, bbox=bbox,
                                 args=args, dcfg=dcfg, batch=batch, dev=dev, labels=labels, image=image)


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
    draws = [torch.rand((sc.batch, res, res, samples), device=sc.dev, generator=gn)]
    if sc.args.fine_sampling:                    # (the inverse-CDF draw exists with fine sampling only, run.py:261-281)
        draws.append(torch.rand((sc.batch * res * res, samples), device=sc.dev, generator=gn))
    return draws


def as_double(sc):
    """The reference side of the scene in float64 (module, latents, cameras): the ground truth the fp32 implementations
    are measured against.  fp64 atomics sum 1e7 addends of either sign to ~1e-13 relative whatever their order, so this
    side is deterministic at the scale of every bound in tests/test_reference_gpu.py - unlike the fp32 reference, whose
    grid_sampler backward scatters with fp32 atomics in arrival order."""
    twin = copy.copy(sc)
    twin.gen = copy.deepcopy(sc.gen).double()
    twin.hip = None
    twin.z, twin.ws, twin.cam = sc.z.double(), sc.ws.double(), sc.cam.double()
    twin.focal, twin.bbox = (None if sc.focal is None else sc.focal.double()), (None if sc.bbox is None else sc.bbox.double())
    return twin


def deterministic_producer():
    """The producer's convolutions restricted to MIOpen's deterministic solvers (torch.backends.cudnn.deterministic).  Left
    to itself MIOpen picks split-K weight-gradient kernels that sum with fp32 atomics in arrival order, and d loss / d
    latents then moves by 1e-5 ... 2e-3 (relative L2) from one run to the next IN BOTH IMPLEMENTATIONS - the renderer's share of
    a comparison drowns in it.  With the flag every gradient comparison of this file repeats to all printed digits
    (profiles/r6/gradient_spread.json: three runs each way)."""
    return torch.backends.cudnn.flags(enabled=True, deterministic=True, benchmark=False)


@contextlib.contextmanager
def default_dtype(dtype):
    """torch's default dtype for the duration: the reference builds a few tensors without naming one (arange(S) / S as
    the lerp weight, lib/nerf_utils.py:104-106, which torch.lerp refuses next to float64 planes)."""
    keep = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(keep)


@contextlib.contextmanager
def float32_draws_in_float64():
    """Inside a float64 run of the reference's regulariser branch (generator.py:505-585): the stratified volume samples
    (lib/ops.py sample_volume_stratified draws float32 whatever the module's dtype; grid_sample then refuses the mixed
    dtypes) come out as the SAME float32 draws cast to float64, and randn_like of a float64 tensor (the total-variation
    perturbation) draws in float32 and casts - same Philox stream, same values as the fp32 run."""
    m = reference.modules()
    gen_ops = m.generator.ops
    orig_sample, orig_randn_like = gen_ops.sample_volume_stratified, torch.randn_like

    def sample32(*a, **k):
        with default_dtype(torch.float32):
            return orig_sample(*a, **k).double()
    gen_ops.sample_volume_stratified = sample32
    torch.randn_like = lambda t, **kw: (orig_randn_like(t.float(), **kw).double() if t.dtype == torch.float64 else orig_randn_like(t, **kw))
    try:
        yield
    finally:
        gen_ops.sample_volume_stratified, torch.randn_like = orig_sample, orig_randn_like


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
    nz = None if noise is None else [noise[0][sl]] + [n.view(sc.batch, -1, samples)[sl].reshape(-1, samples) for n in noise[1:]]
    ctx = contextlib.nullcontext()
    extra_in = dict(render_kw.pop('extra_model_inputs', {}))
    cut = (lambda v: v[sl] if torch.is_tensor(v) else v)               # ('freeze_noise' is a bool)
    if images is not None and (device is None or torch.device(device) == cam.device):
        extra_in = {k: cut(v) for k, v in extra_in.items()}
    if device is not None and torch.device(device) != cam.device:
        with torch.no_grad():
            planes = sc.gen.synthesis_network(ws[:, :14], **({'noise_mode': 'const'} if extra_in.get('freeze_noise') else {})).cpu()
            extra_in.pop('freeze_noise', None)                           # (the CPU copy's producer is frozen altogether)
            if sc.gen.attention_values > 0:
                # the colour table the GPU model ends up with under the caller's model inputs (override / bias), as the
                # CPU copy's override
                given = {k: cut(v) for k, v in extra_in.items()}
                extra_in = {'attention_values': sc.gen(None, ws, ['attention_values', 'sampler'], given)['attention_values'].cpu()}
            else:
                extra_in = {k: cut(v).to(device) for k, v in extra_in.items()}
        gen = copy.deepcopy(sc.gen).to(device)
        ws, cam, focal, bbox = ws.to(device), cam.to(device), pick_to(focal, device), pick_to(bbox, device)
        ctx = frozen_producer(gen, planes)
    with ctx, (ReplayNoise(nz) if nz is not None else contextlib.nullcontext()), \
            (contextlib.nullcontext() if grad else torch.no_grad()), \
            (default_dtype(torch.float64) if cam.dtype == torch.float64 else contextlib.nullcontext()):
        return ren(gen, res, res, cam, focal, None, bbox, ws, samples, extra_model_inputs=extra_in, **render_kw)


def pick_to(t, device):
    return None if t is None else t.to(device)


def _attention(gen, ws):
    """The colour table Generator.forward computes (models/generator.py:452-464)."""
    return gen(None, ws, ['attention_values', 'sampler'])['attention_values']


def with_texels(sc, texel_dtype):
    """The same scene with the HIP twin's planes stored as `texel_dtype` (ops.TEXEL_F32 / TEXEL_F16 / TEXEL_BF16): the
    reference side (sc.gen) is untouched, so everything compared against it is compared against fp32 planes."""
    import nerf_from_image_amd.generator as nfi_gen
    twin = copy.copy(sc)
    twin.hip = nfi_gen.attach(copy.deepcopy(sc.gen), texel_dtype=texel_dtype)
    return twin


def hip_render(sc, res, samples, noise, grad=False, ws=None, cam=None, focal=None, hip_options=None, **render_kw):
    import nerf_from_image_amd.render as nfi_render
    ren = nfi_render.make_render(sc.args, sc.dcfg, **(hip_options or {}))
    with (ReplayNoise(noise) if noise is not None else contextlib.nullcontext()), \
            (contextlib.nullcontext() if grad else torch.no_grad()):
        return ren(sc.hip, res, res, sc.cam if cam is None else cam, sc.focal if focal is None else focal, None, sc.bbox,
                   sc.ws if ws is None else ws, samples, **render_kw)


def max_err(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def mean_err(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().mean())


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def compare(sc, res, samples, cpu_images=2, hip_options=None, **render_kw):
    """HIP vs the reference on this GPU (whole batch) and vs the reference on the CPU (first `cpu_images` images, same
    planes).  Returns a dict of max |error| (and mean |error|) per output and the CPU-vs-GPU gap of the reference itself.
    hip_options: render options of the drop-in only (termination_eps ...); the reference always renders exactly, in fp32."""
    noise = draw_noise(sc, res, samples)
    ours = hip_render(sc, res, samples, noise, hip_options=hip_options, **render_kw)
    ref_gpu = reference_render(sc, res, samples, noise, **render_kw)
    names = ['rgb', 'depth', 'mask', 'normals', 'extra']
    rep = {'mask_mean': float(ref_gpu[2].mean()), 'vs_reference_gpu': {}, 'vs_reference_cpu': {}, 'reference_cpu_vs_gpu_gap': {},
           'pixels_over_1e-4_vs_reference_gpu': {}, 'mean_abs_vs_reference_gpu': {}, 'mean_abs_vs_reference_cpu': {}}
    for k, a, b in zip(names, ours[:5], ref_gpu[:5]):
        if a is None and b is None:
            continue
        assert a is not None and b is not None and a.shape == b.shape, (k, None if a is None else a.shape, None if b is None else b.shape)
        rep['vs_reference_gpu'][k] = max_err(a, b)
        rep['mean_abs_vs_reference_gpu'][k] = mean_err(a, b)
        rep['pixels_over_1e-4_vs_reference_gpu'][k] = int(((a - b).abs() > 1e-4).sum())
    if cpu_images:
        sl = slice(0, min(cpu_images, sc.batch))
        ref_cpu = reference_render(sc, res, samples, noise, device='cpu', images=sl, **render_kw)
        for k, a, b, c in zip(names, ours[:5], ref_cpu[:5], ref_gpu[:5]):
            if a is None:
                continue
            rep['vs_reference_cpu'][k] = max_err(a[sl], b)
            rep['mean_abs_vs_reference_cpu'][k] = mean_err(a[sl], b)
            rep['reference_cpu_vs_gpu_gap'][k] = max_err(c[sl], b)
    return rep


# The configurations BASELINE.json words with 16-bit storage or at cfg5's shape, each against the REAL reference in fp32
# (run.py:176-350; res / ray multipliers run.py:598-605): name -> (geometry, images, rays per side, samples per pass,
# texel storage of the HIP twin, termination_eps of the HIP twin).
CONFIG_CASES = {
    'cfg2_b8_128px_64+64_bf16_texels': ('chairs', 8, 128, 64, 'bf16', 0.0),
    'cfg2_b8_128px_64+64_bf16_texels_term1e-5': ('chairs', 8, 128, 64, 'bf16', 1e-5),
    'cfg2_b8_128px_64+64_fp16_texels': ('chairs', 8, 128, 64, 'fp16', 0.0),
    'cfg2_b8_128px_64+64_fp32_texels_term1e-5': ('chairs', 8, 128, 64, 'fp32', 1e-5),
    'cfg5_b2_256px_128+128_fp32_texels': ('chairs', 2, 256, 128, 'fp32', 0.0),
    'cfg5_b2_256px_128+128_fp32_texels_term1e-5': ('chairs', 2, 256, 128, 'fp32', 1e-5),
    'cfg5_b2_256px_128+128_fp16_texels': ('chairs', 2, 256, 128, 'fp16', 0.0),
    'cfg5_b2_256px_128+128_fp16_texels_term1e-5': ('chairs', 2, 256, 128, 'fp16', 1e-5),
}


def texel_code(name):
    from nerf_from_image_amd import ops
    return {'fp32': ops.TEXEL_F32, 'fp16': ops.TEXEL_F16, 'bf16': ops.TEXEL_BF16}[name]


def config_case(name, dev, cpu_images=1, scenes=None):
    """compare() of one CONFIG_CASES entry.  scenes: a dict the built fp32 scenes are kept in between cases (the
    producer's weights and the cameras are the same for every storage type)."""
    geometry, batch, res, samples, texels, eps = CONFIG_CASES[name]
    key = (geometry, batch)
    sc = (scenes or {}).get(key) or build_scene(geometry, batch, dev)
    if scenes is not None:
        scenes[key] = sc
    twin = sc if texels == 'fp32' else with_texels(sc, texel_code(texels))
    rep = compare(twin, res, samples, cpu_images=cpu_images, hip_options={'termination_eps': eps} if eps else None)
    rep['texels'], rep['termination_eps'] = texels, eps
    return rep


def parallel_model(render_fn, model, resolution, samples):
    """run.py's own ParallelModel (560-617), AST-sliced, with the module-level names it reads: `render` and
    `depth_samples_per_ray`."""
    env = {'nn': torch.nn, 'torch': torch, 'render': render_fn, 'depth_samples_per_ray': samples}
    reference.slice_functions('run.py', ['ParallelModel'], env)
    return env['ParallelModel'](resolution, model=model, model_ema=model)


def gradients(sc, res, samples, seed=5, float64=True, **render_kw):
    """Forward + backward of  sum(rgb * w_rgb) + sum(mask * w_mask)  w.r.t. the latents ws, the camera matrix and the
    focal length, in both implementations (same noise).  Returns relative L2 errors of the gradients: HIP against the
    fp32 reference, and (float64=True) HIP and the fp32 reference each against the reference run in FLOAT64 on the same
    device (`as_double`) - the comparator that does not move from run to run.
    render_kw (compute_semantics=True ...): passed to both renders; the extra map in slot 4 then joins the loss with random
    weights of its own (the staged path WITH a gradient)."""
    noise = draw_noise(sc, res, samples)
    gw = torch.Generator(device=sc.dev).manual_seed(seed)
    w_rgb = torch.randn((sc.batch, res, res, 3), device=sc.dev, generator=gw)
    w_mask = torch.randn((sc.batch, res, res), device=sc.dev, generator=gw)
    w_extra = torch.randn((sc.batch, res, res, 16), device=sc.dev, generator=gw)

    def loss_of(out, cast=lambda t: t):
        loss = (out[0] * cast(w_rgb)).sum() + (out[2] * cast(w_mask)).sum()
        if out[4] is not None:
            loss = loss + (out[4] * cast(w_extra[..., :out[4].shape[-1]])).sum()
        return loss

    def leaves():
        ws = sc.ws.detach().clone().requires_grad_()
        cam = sc.cam.detach().clone().requires_grad_()
        focal = None if sc.focal is None else sc.focal.detach().clone().requires_grad_()
        return ws, cam, focal

    def run(which):
        ws, cam, focal = leaves()
        kept = {}

        def keep_planes(mod, inp, out):         # d loss / d planes: what the renderer hands the producer's backward
            if out.requires_grad:
                out.retain_grad()
                kept['planes'] = out
        if which == 'hip':
            h = sc.hip.synthesis_network.register_forward_hook(keep_planes)
            out = hip_render(sc, res, samples, noise, grad=True, ws=ws, cam=cam, focal=focal, **render_kw)
            loss = loss_of(out)
        elif which == 'ref':
            keep = sc.ws, sc.cam, sc.focal
            sc.ws, sc.cam, sc.focal = ws, cam, focal
            h = sc.gen.synthesis_network.register_forward_hook(keep_planes)
            try:
                out = reference_render(sc, res, samples, noise, grad=True, **render_kw)
            finally:
                sc.ws, sc.cam, sc.focal = keep
            loss = loss_of(out)
        else:                              # the reference in float64: the deterministic ground truth
            sc64 = as_double(sc)
            sc64.gen.requires_grad_(False)
            ws, cam = ws.detach().double().requires_grad_(), cam.detach().double().requires_grad_()
            focal = None if focal is None else focal.detach().double().requires_grad_()
            sc64.ws, sc64.cam, sc64.focal = ws, cam, focal
            h = sc64.gen.synthesis_network.register_forward_hook(keep_planes)
            out = reference_render(sc64, res, samples, [n.double() for n in noise], grad=True, **render_kw)
            loss = loss_of(out, lambda t: t.double())
        h.remove()
        loss.backward()
        g_planes.append(kept['planes'].grad.detach().clone())
        if which == 'ref':
            planes32.append(kept['planes'].detach())
        return float(loss.detach()), ws.grad, cam.grad, None if focal is None else focal.grad

    def run_renderer_only_float64():
        """The float64 reference on the SAME planes and colour table the fp32 implementations rendered (the producer's own
        fp32 rounding - MIOpen's, 1e-6 relative, amplified ~1e3 by the texel differences every coordinate gradient is made
        of - is then common to all three and drops out): d loss / d planes, camera, focal of the RENDERER alone."""
        sc64 = as_double(sc)
        sc64.gen.requires_grad_(False)
        planes = planes32[0].double().requires_grad_()
        cam = sc.cam.detach().double().requires_grad_()
        focal = None if sc.focal is None else sc.focal.detach().double().requires_grad_()
        sc64.cam, sc64.focal = cam, focal
        extra = {}
        if sc.gen.attention_values > 0:
            with torch.no_grad():
                extra = {'attention_values': _attention(sc.gen, sc.ws).double()}
        with frozen_producer(sc64.gen, planes):
            out = reference_render(sc64, res, samples, [n.double() for n in noise], grad=True, extra_model_inputs=extra, **render_kw)
        loss = loss_of(out, lambda t: t.double())
        loss.backward()
        return planes.grad, cam.grad, None if focal is None else focal.grad
    g_planes, planes32 = [], []
    for mod in (sc.gen, sc.hip):
        mod.requires_grad_(False)
    l_h, gws_h, gcam_h, gf_h = run('hip')
    l_r, gws_r, gcam_r, gf_r = run('ref')
    rep = {'loss_hip': l_h, 'loss_reference': l_r, 'g_ws': rel_err(gws_h, gws_r), 'g_cam': rel_err(gcam_h[:, :3], gcam_r[:, :3])}
    if gf_h is not None:
        rep['g_focal'] = rel_err(gf_h, gf_r)
    rep['g_planes'] = rel_err(g_planes[0], g_planes[1])
    if float64:
        l_d, gws_d, gcam_d, gf_d = run('ref64')
        rep['loss_reference_float64'] = l_d
        rep['hip_vs_float64'] = {'g_ws': rel_err(gws_h, gws_d), 'g_cam': rel_err(gcam_h[:, :3], gcam_d[:, :3]),
                                 'g_planes': rel_err(g_planes[0], g_planes[2])}
        rep['reference_vs_float64'] = {'g_ws': rel_err(gws_r, gws_d), 'g_cam': rel_err(gcam_r[:, :3], gcam_d[:, :3]),
                                       'g_planes': rel_err(g_planes[1], g_planes[2])}
        if gf_h is not None:
            rep['hip_vs_float64']['g_focal'] = rel_err(gf_h, gf_d)
            rep['reference_vs_float64']['g_focal'] = rel_err(gf_r, gf_d)
        gp_f, gcam_f, gf_f = run_renderer_only_float64()
        rep['renderer_only_hip_vs_float64'] = {'g_planes': rel_err(g_planes[0], gp_f), 'g_cam': rel_err(gcam_h[:, :3], gcam_f[:, :3])}
        rep['renderer_only_reference_vs_float64'] = {'g_planes': rel_err(g_planes[1], gp_f), 'g_cam': rel_err(gcam_r[:, :3], gcam_f[:, :3])}
        if gf_h is not None:
            rep['renderer_only_hip_vs_float64']['g_focal'] = rel_err(gf_h, gf_f)
            rep['renderer_only_reference_vs_float64']['g_focal'] = rel_err(gf_r, gf_f)
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


def regularisers(sc, seed=321, weights=(1.0, 0.7, 3.0, 0.01), float64=True):
    """The G step's regulariser branch (generator.py:505-585) on the real Generator in training mode: the reference's own
    forward against `attach(model, hip_regularisers=True)`, same seed (= the same two draws), losses and gradients w.r.t. the
    latents (through the synthesis network), the decoder and beta.  Returns relative errors."""
    import nerf_from_image_amd.generator as nfi_gen

    def run(model, double=False):
        model = model.train().requires_grad_(True)
        ws = sc.ws.detach().clone().requires_grad_() if not double else sc.ws.detach().double().requires_grad_()
        torch.manual_seed(seed)
        with (float32_draws_in_float64() if double else contextlib.nullcontext()):
            out = model(None, ws, REGULARISER_NAMES)
            assert set(out) == set(REGULARISER_NAMES)
            loss = sum(w * out[n].sum() for w, n in zip(weights, REGULARISER_NAMES))
            dec = model.decoder.net
            grads = torch.autograd.grad(loss, [ws, dec[0].weight, dec[0].bias, dec[2].weight, model.beta])
        return {n: out[n].detach() for n in REGULARISER_NAMES}, grads
    names = ['ws', 'w1', 'b1', 'w2', 'beta']
    ref_out, ref_g = run(copy.deepcopy(sc.gen))
    hip_out, hip_g = run(nfi_gen.attach(copy.deepcopy(sc.gen), hip_regularisers=True))
    rep = {'loss_rel': {n: max_err(hip_out[n], ref_out[n]) / float(ref_out[n].abs().max()) for n in REGULARISER_NAMES},
           'loss_reference': {n: ref_out[n].tolist() for n in REGULARISER_NAMES},
           'grad_rel_l2': {name: rel_err(a, b) for name, a, b in zip(names, hip_g, ref_g)}}
    if float64:
        d_out, d_g = run(copy.deepcopy(sc.gen).double(), double=True)
        rep['hip_vs_float64'] = {name: rel_err(a, b) for name, a, b in zip(names, hip_g, d_g)}
        rep['reference_vs_float64'] = {name: rel_err(a, b) for name, a, b in zip(names, ref_g, d_g)}
        rep['loss_rel_hip_vs_float64'] = {n: max_err(hip_out[n], d_out[n]) / float(d_out[n].abs().max()) for n in REGULARISER_NAMES}
    return rep


def _summary(g_a, g_b):
    """Relative L2 error over ALL tensors of g_a against g_b, and the worst tensor among those that carry at least 1e-3 of
    the whole gradient's norm."""
    num = sum(float((g_a[n].double() - g_b[n].double()).pow(2).sum()) for n in g_b)
    den = sum(float(g_b[n].double().pow(2).sum()) for n in g_b)
    per = {n: rel_err(g_a[n], g_b[n]) for n in g_b}
    big = {n: e for n, e in per.items() if float(g_b[n].double().norm()) >= 1e-3 * den ** 0.5}
    worst = max(big, key=big.get)
    return {'all_parameters': (num / den) ** 0.5, 'worst_tensor': worst, 'worst_tensor_rel_l2': big[worst], 'significant_tensors': len(big)}


def training_step(sc, res, samples, seed=77, reg_weight=0.1, float64=True, fused_handoff=False, one_forward=False,
                  path_length=False):
    """One generator-side training step in cfg4's shape on the REAL Generator (training mode, latents through the mapping
    network): render + image / alpha loss (run.py:980-1010), regulariser forward (974-979, 1011-1028), one backward - the
    reference's own render + forward against the drop-in render + `attach(..., hip_regularisers=True)`, same noise and
    seed.  Compares the loss and the gradient of EVERY generator parameter.  Returns a dict.
    one_forward: the way run.py's G loop calls it (run.py:966-986): ONE Generator.forward inside render() serves the sampler AND
    the regularisers (`extra_model_outputs=['sdf_eikonal_loss', 'total_variation_loss', 'entropy_loss']`); the forward's own
    stratified volume draw (lib/ops.py sample_volume_stratified: rand_like) sits between render's two draws."""
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    noise = draw_noise(sc, res, samples, seed=seed)
    g = torch.Generator().manual_seed(seed)
    t_rgb = (torch.rand(sc.batch, res, res, 3, generator=g) * 2 - 1).to(sc.dev)
    t_mask = (torch.rand(sc.batch, res, res, generator=g) > 0.5).float().to(sc.dev)
    ref_render, _ = reference.load_render(sc.args, sc.dcfg, unscripted_stages=True)
    hip_render_fn = nfi_render.make_render(sc.args, sc.dcfg)
    bins_noise = torch.rand((sc.batch, 31, 31, 31, 3), generator=g).to(sc.dev)      # sample_volume_stratified's draw (nstrata 32)

    def run(model, render_fn, double=False):
        model = model.train().requires_grad_(True)
        for p_ in model.parameters():
            p_.grad = None
        cast = (lambda t: None if t is None else t.double()) if double else (lambda t: t)
        if one_forward:
            # (+ the path-length regulariser of run.py:968-969, 1029-1043: a double backward through the synthesis network - the
            #  attached forward runs the last block unfused for it)
            names = ['sdf_eikonal_loss', 'total_variation_loss', 'entropy_loss'] + (['path_length'] if path_length else [])
            draws = [cast(noise[0]), cast(bins_noise), cast(noise[1])]
            torch.manual_seed(seed)                                     # (the total-variation perturbation: randn_like)
            with ReplayNoise(draws), (default_dtype(torch.float64) if double else contextlib.nullcontext()), \
                    (float32_draws_in_float64() if double else contextlib.nullcontext()):
                out = render_fn(model, res, res, cast(sc.cam), cast(sc.focal), None, cast(sc.bbox), cast(sc.z), samples,
                                extra_model_outputs=names)
                loss = ((out[0] - cast(t_rgb)) ** 2).mean() + ((out[2] - cast(t_mask)) ** 2).mean()
                loss = loss + reg_weight * (out[5]['sdf_eikonal_loss'].mean() + 5.0 * out[5]['total_variation_loss'].mean() +
                                            0.1 * out[5]['entropy_loss'].mean())
                if path_length:
                    ppl = out[5]['path_length']
                    loss = loss + 2.0 * (ppl - ppl.mean().detach()).square().mean()
                loss.backward()
            return float(loss.detach()), {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
        with ReplayNoise([cast(n) for n in noise]), (default_dtype(torch.float64) if double else contextlib.nullcontext()):
            out = render_fn(model, res, res, cast(sc.cam), cast(sc.focal), None, cast(sc.bbox), cast(sc.z), samples)
        loss = ((out[0] - cast(t_rgb)) ** 2).mean() + ((out[2] - cast(t_mask)) ** 2).mean()
        torch.manual_seed(seed)
        with (float32_draws_in_float64() if double else contextlib.nullcontext()):
            reg = model(None, cast(sc.z), ['sdf_eikonal_loss', 'sdf_distance_loss'])
            loss = loss + reg_weight * (reg['sdf_eikonal_loss'].mean() + reg['sdf_distance_loss'].mean())
            loss.backward()
        return float(loss.detach()), {n: p_.grad.detach().clone() for n, p_ in model.named_parameters() if p_.grad is not None}
    l_r, g_r = run(copy.deepcopy(sc.gen), ref_render)
    l_h, g_h = run(nfi_gen.attach(copy.deepcopy(sc.gen), hip_regularisers=True, fused_handoff=fused_handoff), hip_render_fn)
    assert set(g_r) == set(g_h), set(g_r) ^ set(g_h)
    s_ = _summary(g_h, g_r)
    rep = {'loss_hip': l_h, 'loss_reference': l_r, 'n_parameter_tensors': len(g_r), 'grad_rel_l2_all_parameters': s_['all_parameters'],
           'worst_tensor_among_the_significant': s_['worst_tensor'], 'worst_tensor_rel_l2': s_['worst_tensor_rel_l2'],
           'significant_tensors': s_['significant_tensors']}
    strengths = [n for n in g_r if n.endswith('noise_strength')]
    if strengths:
        rep['noise_strength_tensors'] = len(strengths)
        rep['noise_strength_grad_rel_l2'] = rel_err(torch.stack([g_h[n] for n in strengths]), torch.stack([g_r[n] for n in strengths]))
    if float64:
        l_d, g_d = run(copy.deepcopy(sc.gen).double(), ref_render, double=True)
        assert set(g_d) == set(g_h)
        rep['loss_reference_float64'] = l_d
        rep['hip_vs_float64'] = _summary(g_h, g_d)
        rep['reference_vs_float64'] = _summary(g_r, g_d)
    return rep
