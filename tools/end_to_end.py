"""End-to-end timings on the REAL reference classes (SURVEY.md 8(d) metric (ii)): the reference's own Generator - mapping
network, StyleGAN2 synthesis network, texture mapper: the plane producer, PyTorch-ROCm / MIOpen in BOTH implementations -
rendered by run.py::render (reference) or by the drop-in with the attach()ed HIP sampler, same weights and cameras.

Legs (what run.py executes, with the parts that cannot exist offline named):
  render     render() INCLUDING Generator.forward (run.py:221 -> models/generator.py:475-477), BASELINE cfg2: chairs
             geometry, B images x 128 x 128 rays x (64 + 64) samples, latents ws given (the eval / inversion callers), no grad.
  inversion  one optimisation step of --run_inversion (run.py:2264-2299) in BASELINE cfg3's per-GPU shape: p3d_car-like
             geometry, 4 images, Adam(2e-3, betas 0.9 / 0.95, run.py:2007) on latents + camera + focal, MSE on image and
             mask against a synthetic target.  NOT included: the 15-way augmentation and the LPIPS network (needs the
             `lpips` package and its weights: absent offline).
  gstep      the generator side of one GAN iteration (run.py:955-1074) in BASELINE cfg4's per-GPU shape: cub-like geometry
             (orthographic), 4 images, model.train(), latents from z through the mapping network, ONE Generator.forward
             serving sampler + eikonal regulariser (+ path length with --path-length), L1 image / alpha loss in the place
             of the discriminator (NOT included: D), clip_grad_norm_, Adam(0, 0.99), beta / alpha clamp.

Timing: HIP events around every step on the current stream, median.  With --markers the step emits marker kernels (one
otherwise unused elementwise op per phase boundary) so that tools/phase_split.py can cut a rocprofv3 kernel trace of the
same command into renderer / producer / other GPU time.

TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/reference.py): never part of the product or of bench.py's timed region.
  python tools/end_to_end.py                                  # every leg, both implementations, JSON on stdout
  python tools/end_to_end.py --leg gstep --impl hip --markers --sidecar /tmp/gstep_hip.json
"""
import argparse
import json
import os
import sys
import tempfile

# the producer's timings are MIOpen's: start from an empty user database (a fresh box's state) whatever ran on this box before
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='nfi_miopen_db_'))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# phase boundaries and the elementwise op whose kernel marks each (none of them occurs in the producer, the renderers,
# the losses or the optimiser; the kernel name contains '<op>_kernel_cuda')
MARKER_OPS = {'step_begin': 'erfinv', 'model_begin': 'digamma', 'synth_end': 'lgamma', 'model_end': 'erfc',
              'render_end': 'sinc', 'loss_bwd_end': 'frac', 'reg_bwd_begin': 'expm1', 'producer_bwd_begin': 'asinh',
              'bwd_end': 'atanh'}
REGULARISER_OUTPUTS = ('sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss', 'path_length')


class Markers:
    def __init__(self, dev, enabled):
        self.enabled = enabled
        self.src = torch.full((1,), 0.25, device=dev)
        self.dst = torch.empty_like(self.src)

    def emit(self, name):
        if self.enabled:
            getattr(torch, MARKER_OPS[name])(self.src, out=self.dst)

    def on_grad(self, tensor, name):
        """The marker is launched when the gradient of `tensor` is ready, i.e. right before the node that produced it
        runs its backward."""
        if self.enabled and torch.is_tensor(tensor) and tensor.requires_grad:
            tensor.register_hook(lambda g: self.emit(name))

    def instrument(self, model):
        """Generator.forward = [model_begin .. synth_end] producer proper, [synth_end .. model_end] the rest of the forward
        (regulariser branch, path length; with the drop-in also the texel hand-off and the decoder pack).  Backward:
        autograd runs nodes in reverse creation order, so the renderer's nodes (created by render() after the forward
        returned) run first, then the regulariser branch's, then the producer's."""
        if not self.enabled:
            return

        def synth_done(mod, inp, out):
            self.emit('synth_end')
            self.on_grad(out, 'producer_bwd_begin')

        def model_done(mod, inp, out):
            self.emit('model_end')
            for k in REGULARISER_OUTPUTS:
                if k in out:
                    self.on_grad(out[k], 'reg_bwd_begin')
        model.synthesis_network.register_forward_hook(synth_done)
        model.register_forward_pre_hook(lambda mod, inp: self.emit('model_begin'))
        model.register_forward_hook(model_done)


def _time_steps(step, iters, warmup):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    for i in range(iters):
        evs[i].record()
        step()
    evs[iters].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(iters))
    return {'ms_median': per[len(per) // 2], 'ms_min': per[0], 'ms_max': per[-1], 'iters': iters}


def _scene(geometry, batch, dev, impl, texels, fused_handoff=False, hip_regularisers=False, channels_last=False):
    import copy
    import reference_cases as rc
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    from oracle import reference
    sc = rc.build_scene(geometry, batch, dev)
    if impl == 'hip':
        model = nfi_gen.attach(copy.deepcopy(sc.gen), texel_dtype=rc.texel_code(texels), hip_regularisers=hip_regularisers,
                               fused_handoff=fused_handoff)
        ren = nfi_render.make_render(sc.args, sc.dcfg)
    else:
        model = sc.gen
        ren, _ = reference.load_render(sc.args, sc.dcfg)           # scripted stage functions, as run.py runs them
    if channels_last:
        # experiment: the producer's convolution weights (and with them its activations) in NHWC memory format - MIOpen's
        # implicit-GEMM kernels are NHWC and bracket every NCHW call with batched_transpose launches
        model.synthesis_network.to(memory_format=torch.channels_last)
    return sc, model, ren


def leg_render(dev, impl, batch=8, texels='fp32', iters=20, warmup=3, markers=False, res=128, samples=64, channels_last=False,
               graph=False):
    sc, model, ren = _scene('chairs', batch, dev, impl, texels, channels_last=channels_last)
    mk = Markers(dev, markers)
    mk.instrument(model)
    graphed = None
    if graph:
        # the whole call as ONE HIP graph (nerf_from_image_amd/graphs.py; drop-in only: the reference's render reads the hit
        # mask back on the host, lib/nerf_utils.py:258, which cannot be captured)
        assert impl == 'hip' and not markers
        import nerf_from_image_amd.render as nfi_render
        from nerf_from_image_amd.graphs import GraphedRender
        graphed = GraphedRender(nfi_render.make_render(sc.args, sc.dcfg, strict_near_far=False), model, res, res, sc.cam, sc.focal,
                                None, None, sc.ws, samples)

    def step():
        if graphed is not None:
            graphed(sc.cam, sc.focal, None, None, sc.ws)
            return
        mk.emit('step_begin')
        with torch.no_grad():
            ren(model, res, res, sc.cam, sc.focal, None, None, sc.ws, samples)
        mk.emit('render_end')
    r = _time_steps(step, iters, warmup)
    r.update(rays_per_s=batch * res * res / (r['ms_median'] * 1e-3), images=batch, rays=batch * res * res)
    return r


def leg_inversion(dev, impl, batch=4, texels='fp32', iters=12, warmup=3, markers=False, res=128, samples=64):
    import reference_cases as rc
    sc, model, ren = _scene('p3d', batch, dev, impl, texels)
    model.requires_grad_(False)
    with torch.no_grad():
        tgt = rc.reference_render(sc, res, samples, None)
    t_rgb, t_mask = tgt[0].detach(), tgt[2].detach()
    g = torch.Generator().manual_seed(11)
    ws = (sc.ws + 0.25 * torch.randn(sc.ws.shape, generator=g).to(dev) * sc.ws.std()).requires_grad_()
    cam = sc.cam.clone()
    cam[:, :3, 3] += 0.03 * torch.randn(batch, 3, generator=g).to(dev)
    cam.requires_grad_()
    focal = (sc.focal * (1.0 + 0.02 * torch.randn(batch, generator=g).to(dev))).requires_grad_()
    opt = torch.optim.Adam([ws, cam, focal], lr=2e-3, betas=(0.9, 0.95))
    mk = Markers(dev, markers)
    mk.instrument(model)

    def step():
        mk.emit('step_begin')
        opt.zero_grad()
        out = ren(model, res, res, cam, focal, None, sc.bbox, ws, samples)
        mk.emit('render_end')
        mk.on_grad(out[0], 'loss_bwd_end')
        mk.on_grad(out[2], 'loss_bwd_end')
        loss = F.mse_loss(out[0], t_rgb) + F.mse_loss(out[2], t_mask)
        loss.backward()
        mk.emit('bwd_end')
        opt.step()
    r = _time_steps(step, iters, warmup)
    r.update(images=batch, rays=batch * res * res)
    return r


def leg_gstep(dev, impl, batch=4, texels='fp32', iters=12, warmup=3, markers=False, res=128, samples=64, fused_handoff=False,
              path_length=False, hip_regularisers=True):
    sc, model, ren = _scene('cub', batch, dev, impl, texels, fused_handoff=fused_handoff, hip_regularisers=hip_regularisers)
    model.train().requires_grad_(True)
    g = torch.Generator().manual_seed(77)
    t_img = torch.cat([torch.rand(batch, res, res, 3, generator=g) * 2 - 1, (torch.rand(batch, res, res, 1, generator=g) > 0.5).float()],
                      dim=-1).to(dev)
    params = [p for p in model.parameters()]
    opt = torch.optim.Adam(params, lr=0.0025, betas=(0., 0.99))
    want = ['sdf_eikonal_loss'] + (['path_length'] if path_length else [])
    mk = Markers(dev, markers)
    mk.instrument(model)

    def step():
        mk.emit('step_begin')
        opt.zero_grad()
        rgb, _, acc, _, _, extra = ren(model, res, res, sc.cam, sc.focal, None, sc.bbox, sc.z, samples, extra_model_outputs=want)
        mk.emit('render_end')
        mk.on_grad(rgb, 'loss_bwd_end')
        mk.on_grad(acc, 'loss_bwd_end')
        loss = F.l1_loss(torch.cat((rgb, acc.unsqueeze(-1)), dim=-1), t_img) * 10
        loss = loss + 0.1 * extra['sdf_eikonal_loss'].mean()
        if path_length:
            ppl = extra['path_length']
            loss = loss + 2.0 * (ppl - ppl.mean().detach()).square().mean()
        loss.backward()
        mk.emit('bwd_end')
        torch.nn.utils.clip_grad_norm_(params, 100.0)
        opt.step()
        model.beta.data.clamp_(min=1e-3)
        model.alpha.data.clamp_(min=1e-3)
    r = _time_steps(step, iters, warmup)
    r.update(images=batch, rays=batch * res * res)
    return r


LEGS = {'render': leg_render, 'inversion': leg_inversion, 'gstep': leg_gstep}


def summary(dev, quick=False):
    """Every leg, both implementations (bench.py's extras block calls this after its timed region)."""
    it = dict(iters=6, warmup=2) if quick else {}
    out = {'render_incl_synthesis': {}, 'sample': __doc__.split('Timing:')[0].split('Legs')[1]}
    for b in (1, 4, 8):
        ref = leg_render(dev, 'reference', batch=b, **it)
        row = {'reference_rays_per_s': ref['rays_per_s'], 'reference_ms': ref['ms_median']}
        for tx in ('fp32', 'bf16'):
            h = leg_render(dev, 'hip', batch=b, texels=tx, **it)
            row['hip_%s_texels_rays_per_s' % tx] = h['rays_per_s']
            row['hip_%s_texels_ms' % tx] = h['ms_median']
            row['x_reference_%s_texels' % tx] = h['rays_per_s'] / ref['rays_per_s']
        try:
            h = leg_render(dev, 'hip', batch=b, graph=True, **it)
            row['hip_fp32_texels_hip_graph_rays_per_s'] = h['rays_per_s']
            row['hip_fp32_texels_hip_graph_ms'] = h['ms_median']
            row['x_reference_hip_graph'] = h['rays_per_s'] / ref['rays_per_s']
        except Exception as e:       # noqa: BLE001 (a producer that cannot be captured is reported, not fatal)
            row['hip_graph_error'] = repr(e)[:300]
        out['render_incl_synthesis']['b%d' % b] = row
        torch.cuda.empty_cache()
    ref = leg_inversion(dev, 'reference', **it)
    hip = leg_inversion(dev, 'hip', **it)
    out['inversion_step_4_images'] = {'reference_ms': ref['ms_median'], 'hip_ms': hip['ms_median'], 'x_reference': ref['ms_median'] / hip['ms_median']}
    torch.cuda.empty_cache()
    out['g_step_4_images'] = {}
    for pl in (False, True):
        ref = leg_gstep(dev, 'reference', path_length=pl, **it)
        row = {'reference_ms': ref['ms_median']}
        for name, kw in (('hip', {}), ('hip_fused_handoff', {'fused_handoff': True}), ('hip_reference_regularisers', {'hip_regularisers': False})):
            h = leg_gstep(dev, 'hip', path_length=pl, **kw, **it)
            row[name + '_ms'] = h['ms_median']
            row['x_reference_' + name] = ref['ms_median'] / h['ms_median']
        out['g_step_4_images']['with_path_length' if pl else 'eikonal_only'] = row
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--leg', choices=sorted(LEGS))
    ap.add_argument('--impl', choices=['hip', 'reference'], default='hip')
    ap.add_argument('--batch', type=int)
    ap.add_argument('--texels', default='fp32', choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--markers', action='store_true')
    ap.add_argument('--fused-handoff', action='store_true')
    ap.add_argument('--path-length', action='store_true')
    ap.add_argument('--reference-regularisers', action='store_true', help='hip: leave the regulariser branch to the original forward')
    ap.add_argument('--sidecar', help='JSON written for tools/phase_split.py (marker table, steps, warm-up)')
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--channels-last', action='store_true', help='render leg: the synthesis network in NHWC memory format (experiment)')
    ap.add_argument('--graph', action='store_true', help='render leg, hip: the whole call captured in a HIP graph (GraphedRender)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    if a.leg is None:
        print(json.dumps(summary(dev, quick=a.quick), indent=1))
        return
    kw = dict(texels=a.texels, iters=a.iters, warmup=a.warmup, markers=a.markers)
    if a.batch:
        kw['batch'] = a.batch
    if a.leg == 'render':
        kw.update(channels_last=a.channels_last, graph=a.graph)
    if a.leg == 'gstep':
        kw.update(fused_handoff=a.fused_handoff, path_length=a.path_length, hip_regularisers=not a.reference_regularisers)
    r = LEGS[a.leg](dev, a.impl, **kw)
    r.update(leg=a.leg, impl=a.impl, **{k: v for k, v in kw.items() if k not in ('iters', 'warmup')})
    if a.sidecar:
        json.dump({'markers': MARKER_OPS, 'iters': a.iters, 'warmup': a.warmup, 'result': r}, open(a.sidecar, 'w'))
    print(json.dumps(r))


if __name__ == '__main__':
    main()
