"""The radiance-field side of the reference's ``models/generator.py`` on HIP kernels.

What is replaced: the ``sampler`` closure that ``Generator.forward`` returns
(models/generator.py:587-681), i.e. TriplanarDecoder.forward (301-331: three bilinear plane
gathers, mean, Linear(32,64)-Softplus-Linear(64,1+A) with equalized-lr gains,
models/stylegan.py:173-180), the SDF->density conversion (laplace_cdf 30-33, 629-636) or the
softplus density (637-641), and the colour head (661-679).

What is NOT replaced: the plane producer (mapping network, StyleGAN2 synthesis network,
AttentionMapper) stays the reference's own PyTorch modules; ``hip_forward`` calls them exactly
where ``Generator.forward`` does (407-503) and hands their output to the kernels.  So a reference
``Generator`` instance keeps its parameters, ``state_dict`` keys and attributes and only gets a new
``forward``:

    from models.generator import Generator            # the reference's class
    import nerf_from_image_amd.generator as nfi_gen
    model = nfi_gen.attach(Generator(512, scene_range, attention_values=10, use_sdf=True))

Any module with the attributes listed in ``REQUIRED_ATTRS`` works the same way (the GPU-box tests
use a stand-in plane producer, the reference checkout is not available there).
"""
import math
import types

import torch

from . import ops
from .autograd import differentiable
from .field_backward import make_field_bwd, surface_normals

REQUIRED_ATTRS = ('scene_range', 'attention_values', 'use_sdf', 'use_viewdir', 'use_encoder', 'num_classes',
                  'mapping_network', 'synthesis_network', 'decoder')

_MODEL_OUTPUTS = ('sampler', 'sdf_eikonal_loss', 'sdf_distance_loss', 'path_length', 'total_variation_loss',
                  'entropy_loss', 'attention_values', 'bbox')
_MODEL_INPUTS = ('freeze_noise', 'attention_values', 'attention_values_bias')
_SAMPLER_OUTPUTS = ('sdf_distance', 'sigma', 'rgb', 'normals', 'semantics', 'coords')


class FusedField:
    """Everything the fused renderer needs about one batch of scenes (``sampler.fused``)."""

    def __init__(self, texels, decoder_image, attention_values, n_attention, use_sdf, beta, alpha, scene_range,
                 planes=None, decoder_params=None):
        self.texels = texels                    # [B,3,R,R,32] channel-last
        self.decoder_image = decoder_image      # MFMA operand image of the decoder
        self.attention_values = attention_values
        self.n_attention = n_attention
        self.use_sdf = use_sdf
        self.beta = beta
        self.alpha = alpha
        self.scene_range = scene_range
        self.planes = planes                    # autograd handles (None in inference)
        self.decoder_params = decoder_params

    @property
    def requires_grad(self):
        ts = [self.planes, self.attention_values, self.beta, self.alpha] + list(self.decoder_params or ())
        return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def decoder_parameters(decoder):
    """(w1, b1, w2, b2) of a TriplanarDecoder-shaped module: decoder.net[0] / decoder.net[2]."""
    l0, l2 = decoder.net[0], decoder.net[2]
    return l0.weight, l0.bias, l2.weight, l2.bias


class _distance_head:
    """Decoder view exposing only output row 0 (the distance): net[2] becomes a 4-row layer whose rows 1..3 are zero
    (the kernels' A == 0 shape).  Built with differentiable tensor ops, so gradients reach decoder.net[2].weight[0]."""

    def __init__(self, decoder):
        l0, l2 = decoder.net[0], decoder.net[2]
        w = torch.cat([l2.weight[:1], torch.zeros_like(l2.weight[:3])], dim=0)
        b = torch.cat([l2.bias[:1], torch.zeros_like(l2.bias[:3])], dim=0)
        self.net = (l0, None, types.SimpleNamespace(weight=w, bias=b))
        self.training = getattr(decoder, 'training', False)


def texels_of(planes, texel_dtype=ops.TEXEL_F32, cache=None):
    """The hand-off (generator.py:475-477 -> the kernels): a producer whose [B,96,R,R] output is channels-last in memory
    is read in place (interleaved texel layout, no kernel, no copy); an NCHW producer goes through the one transposition
    launch nfi_planes_to_texels.

    cache: a dict that lives for ONE forward call.  One forward asks twice (the sampler and the regulariser branch) and
    the planes cannot change in between; nothing is kept beyond the call, so a planes tensor that a caller updates
    through ``.data`` (EMA / inversion code in the style of run.py's ``z_.data[:] = ...``: no _version bump) can never
    meet a stale transposed copy."""
    p = planes.detach()
    v = ops.planes_view_as_texels(p) if p.dtype == torch.float32 else None
    if v is not None:
        return v if texel_dtype == ops.TEXEL_F32 else v.to(ops._TEXEL_TORCH[texel_dtype])
    if cache is None:
        return ops.planes_to_texels(p, texel_dtype)
    key = (id(planes), texel_dtype)
    if key not in cache:
        cache[key] = (planes, ops.planes_to_texels(p, texel_dtype))      # (the tensor is held so that its id stays its own)
    return cache[key][1]


def make_sampler(planes, decoder, scene_range, n_attention, attention_values, use_sdf, beta, alpha,
                 texel_dtype=ops.TEXEL_F32, request_model_outputs=(), viewdir=None, texel_cache=None):
    """Builds the ``sampler(x_in, request_sampler_outputs)`` closure over HIP kernels.

    planes [B,3,32,R,R] (view of the synthesis output), decoder: module with .net[0]/.net[2].
    viewdir (--use_viewdir): (ray_feature [B,H,W,1,32] = output of ViewDirectionMapper.fc6, output_layer =
    the mapper's `output` EqualizedLinear); the closure of generator.py:243-251 is then part of the kernels."""
    w1, b1, w2, b2 = decoder_parameters(decoder)
    texels = texels_of(planes, texel_dtype, texel_cache)
    ray_feature = w3 = b3 = ray_pad = None
    if viewdir is not None:
        ray_feature, out_layer = viewdir
        w3, b3 = out_layer.weight, out_layer.bias
        image = ops.decoder_pack_viewdir(w1.detach(), b1.detach(), w2.detach(), b2.detach(), w3.detach(), b3.detach(),
                                         n_attention, texel_dtype)
        ray_pad = ops.pad_ray_features(ray_feature.detach().reshape(ray_feature.shape[0], -1, 32))   # [B,N,48]
    else:
        image = ops.decoder_pack(w1.detach(), b1.detach(), w2.detach(), b2.detach(), n_attention, texel_dtype)
    fused = FusedField(texels, image, attention_values, n_attention, use_sdf, beta, alpha, scene_range,
                       planes=planes, decoder_params=(w1, b1, w2, b2) + ((ray_feature, w3, b3) if viewdir is not None else ()))
    fused.ray_features = ray_pad
    fused.bbox_overlay = 'bbox' in request_model_outputs      # the coords request then edits sigma (generator.py:645-659)

    def sampler(x_in, request_sampler_outputs=['sigma', 'rgb'], mlp_split_fp16=False):
        # mlp_split_fp16 (not in the reference's signature; render() sets it): the decoder arithmetic of the fused
        # renderer instead of exact fp32 - the same in both of render()'s paths, and what the backward recomputes
        for output in request_sampler_outputs:
            assert output in _SAMPLER_OUTPUTS
        want_normals = 'normals' in request_sampler_outputs
        if want_normals:
            # generator.py:599-602: SDF only, eval only; every other output is then detached
            assert use_sdf and not getattr(decoder, 'training', False)
        bs = x_in.shape[0]
        pts = x_in.reshape(bs, -1, 3)
        spr = 0
        if ray_pad is not None:
            # generator.py:243-247: the ray feature is broadcast over x_in's sample axis
            spr = x_in.shape[-2]
            assert pts.shape[1] == ray_pad.shape[1] * spr, (tuple(x_in.shape), tuple(ray_pad.shape))
        want_sem = 'semantics' in request_sampler_outputs
        if want_sem:
            assert n_attention > 0
        want_sdf = 'sdf_distance' in request_sampler_outputs

        def fwd(p, pl, a_w1, a_b1, a_w2, a_b2, att, be, al, *vd_unused):
            q = ops.field_query(p, texels, image, scene_range, n_attention, att, use_sdf, be, al,
                                want_sdf=want_sdf, want_semantics=want_sem, ray_features=ray_pad, samples_per_ray=spr,
                                mlp_precision=1 if (mlp_split_fp16 and ray_pad is None) else 0)
            return tuple(q[k] for k in ('sigma', 'rgb') + (('sdf',) if want_sdf else ()) +
                         (('semantics',) if want_sem else ()))
        bwd = None
        if texel_dtype == ops.TEXEL_F32 or ray_pad is None:      # (the view-direction decoder's backward is fp32-texel only)
            bwd = make_field_bwd(texels, image, scene_range, n_attention, use_sdf, want_sdf, want_sem, ray_pad, spr)
        # what an output does not depend on gets no gradient (None, not zeros - as in the reference's graph): the colour
        # table only enters rgb, beta / alpha only sigma
        det = (lambda t, used: t if (t is None or used) else t.detach())
        want_sigma = 'sigma' in request_sampler_outputs
        args = (pts, planes, w1, b1, w2, b2, det(attention_values, 'rgb' in request_sampler_outputs) if n_attention > 0 else None,
                det(beta, want_sigma) if use_sdf else None, det(alpha, want_sigma) if use_sdf else None)
        if ray_pad is not None:
            args = args + (ray_feature, w3, b3)
        if want_normals:
            args = tuple(None if t is None else t.detach() for t in args)
        res = differentiable('field_query', fwd, *args, bwd=bwd)
        out = {}
        if want_normals:
            out['normals'] = surface_normals(pts.detach(), texels, image, w1.detach(), w2.detach(), scene_range,
                                             n_attention, None if attention_values is None else attention_values.detach(),
                                             use_sdf, beta.detach(), alpha.detach(),
                                             viewdir=None if ray_pad is None else dict(
                                                 ray_features=ray_pad, samples_per_ray=spr, w3=w3.detach())
                                             ).view(x_in.shape)        # (the gradient w.r.t. x_in: its shape, generator.py:614-621)
        i = 2
        if want_sdf:
            out['sdf_distance'] = res[i].unsqueeze(-1)
            i += 1
        if 'sigma' in request_sampler_outputs:
            out['sigma'] = res[0]
        if 'coords' in request_sampler_outputs:
            out['coords'] = x_in
            if 'bbox' in request_model_outputs:
                # visualisation overlay (generator.py:645-659): the reference adds it to sigma in place
                pts_d = pts.detach()
                out['sigma'] = differentiable('bbox_overlay', lambda s_: ops.bbox_overlay(pts_d, s_, scene_range), res[0],
                                              bwd=lambda inputs, outputs, grads, needs: (grads[0],))
        if want_sem:
            out['semantics'] = res[i]
        if 'rgb' in request_sampler_outputs:
            out['rgb'] = res[1]
        return out

    sampler.fused = fused
    return sampler


def sample_volume_stratified(batch_size, nstrata, scene_range, device=None):
    """One jittered point per cell of an n^3 grid (n = nstrata - 1) over the scene cube, as lib/ops.py:20-26 draws them:
    cell (i, j, k) of the [n,n,n] index grid holds (x, y, z) = (j, i, k) (the reference's 'xy' meshgrid), the jitter is
    ONE uniform draw of shape [B,n,n,n,3] (what its rand_like of the expanded index grid consumes), and the point is
    ((cell + jitter) / n * 2 - 1) * scene_range."""
    n = nstrata - 1
    idx = torch.arange(n, device=device).float()
    cell = torch.empty((n, n, n, 3), dtype=torch.float32, device=device)
    cell[..., 0] = idx.view(1, n, 1)
    cell[..., 1] = idx.view(n, 1, 1)
    cell[..., 2] = idx.view(1, 1, n)
    jitter = torch.rand((batch_size, n, n, n, 3), dtype=torch.float32, device=device)
    return ((cell + jitter) / n * 2 - 1).reshape(batch_size, n ** 3, 3) * scene_range


def sdf_and_gradient(points, planes, decoder, scene_range, texel_cache=None):
    """(sdf [B,P], d sdf / d points [B,P,3]) at fixed points as ONE autograd node (HIP forward + HIP backward): the
    gradient output replaces torch.autograd.grad(..., create_graph=True) of generator.py:534-540, its backward is the
    double backward lib/ops.grid_sample2d exists for.  Differentiable w.r.t. planes and the decoder parameters."""
    w1, b1, w2, b2 = decoder_parameters(decoder)
    pts = points.detach()
    texels = texels_of(planes, cache=texel_cache)

    def fwd(pl, a_w1, a_b1, a_w2, a_b2):
        return ops.sdf_gradient_fwd(pts, texels, a_w1, a_b1, a_w2, a_b2, scene_range)

    def bwd(inputs, outputs, grads, needs):
        pl, a_w1, a_b1, a_w2, a_b2 = inputs
        g = ops.sdf_gradient_bwd(pts, texels, a_w1, a_b1, a_w2, a_b2, scene_range, grads[0], grads[1])
        return (ops.texel_grad_to_planes(g['g_texels']) if needs[0] else None, g['g_w1'], g['g_b1'], g['g_w2'], g['g_b2'])
    return differentiable('sdf_gradient', fwd, planes, w1, b1, w2, b2, bwd=bwd)


def regulariser_outputs(self, planes, request_model_outputs, texel_cache=None):
    """generator.py:505-585 on HIP kernels: eikonal / distance / total-variation / entropy terms of the SDF."""
    out = {}
    assert torch.is_grad_enabled()
    bins_in = sample_volume_stratified(planes.shape[0], 32, self.scene_range, device=planes.device)
    if 'sdf_eikonal_loss' in request_model_outputs:
        assert self.use_sdf and self.training
    d, g = sdf_and_gradient(bins_in, planes, self.decoder, self.scene_range, texel_cache)
    if 'sdf_eikonal_loss' in request_model_outputs:
        out['sdf_eikonal_loss'] = ((g.norm(dim=-1) - 1) ** 2).flatten(1).mean(dim=1)
    if 'sdf_distance_loss' in request_model_outputs:
        assert self.use_sdf
        with torch.no_grad():
            target = bins_in.norm(dim=-1) - 1                         # unit sphere
        out['sdf_distance_loss'] = torch.nn.functional.mse_loss(d.flatten(1), target.flatten(1), reduction='none').mean(dim=1)
    want_tv = 'total_variation_loss' in request_model_outputs
    if want_tv or 'entropy_loss' in request_model_outputs:
        d_p = None
        if want_tv:
            coords = (bins_in / self.scene_range).view(planes.shape[0], 1, -1, 3)
            coords_p = coords + torch.randn_like(coords) * 0.004
            # the reference evaluates self.decoder only and keeps channel 0 (generator.py:559-566): no view-direction
            # closure is involved, so a --use_viewdir decoder (33 outputs) is queried through its distance row alone
            dec = _distance_head(self.decoder) if self.use_viewdir else self.decoder
            n_att = 0 if self.use_viewdir else self.attention_values
            smp = make_sampler(planes, dec, self.scene_range, n_att,
                               torch.zeros((planes.shape[0], max(n_att, 1), 3), device=planes.device),
                               self.use_sdf, self.beta if self.use_sdf else None, self.alpha if self.use_sdf else None,
                               texel_cache=texel_cache)
            d_p = smp((coords_p * self.scene_range).detach(), ['sdf_distance'])['sdf_distance'][..., 0]
        if self.use_sdf:
            beta = self.beta
            cdf = lambda z: 0.5 + 0.5 * torch.sign(z) * (1 - torch.exp(-z.abs() / beta))      # laplace_cdf, 30-33
            if want_tv:
                out['total_variation_loss'] = torch.nn.functional.l1_loss(cdf(-d), cdf(-d_p), reduction='none').flatten(1).mean(dim=1)
            if 'entropy_loss' in request_model_outputs:
                out['entropy_loss'] = (0.5 * torch.exp(-d.abs() / beta) / beta).flatten(1).mean(dim=1)      # laplace_pdf, 24-27
        else:
            tv = torch.sigmoid(d - 1)
            if want_tv:
                out['total_variation_loss'] = torch.nn.functional.l1_loss(tv, torch.sigmoid(d_p - 1), reduction='none').flatten(1).mean(dim=1)
            if 'entropy_loss' in request_model_outputs:
                out['entropy_loss'] = (tv * (1 - tv)).flatten(1).mean(dim=1)
    return out


def hip_forward(self, viewdir, c, request_model_outputs=['sampler'], model_inputs={}):
    """Replacement for Generator.forward (models/generator.py:407-686): same arguments, same
    returned dict; the plane producer is called as in the reference, the field is HIP."""
    for output in request_model_outputs:
        assert output in _MODEL_OUTPUTS
    for k in model_inputs.keys():
        assert k in _MODEL_INPUTS

    # ---- latent handling (generator.py:423-446) ----
    label = None
    if self.use_encoder:
        z, image = c
        ws = self.mapping_network(z, self.emb(image))
        batch = z.shape[0]
    else:
        if self.num_classes:
            if isinstance(c, (list, tuple)):
                c, label_idx = c
                assert c.dim() == 2
                label = self.class_embedding(label_idx)
            else:
                assert c.dim() == 3
        batch = c.shape[0]
        if c.dim() == 3:
            num_ws = self.mapping_network.backbone.num_ws
            ws = c.expand(-1, num_ws, -1).contiguous() if c.shape[1] == 1 else c
        else:
            ws = self.mapping_network(c, label)
    if 'path_length' in request_model_outputs:
        assert torch.is_grad_enabled()
        ws = ws.contiguous().requires_grad_()

    # ---- colour table + planes (generator.py:448-477) ----
    attention_values = None
    if self.attention_values > 0:
        assert ws.shape[1] == 15
        w_tex, w_syn = ws[:, 14], ws[:, :14]
        if 'attention_values' in model_inputs:
            attention_values = model_inputs['attention_values']
        elif 'sampler' in request_model_outputs:
            attention_values = self.texture_mapper(w_tex)
            if 'attention_values_bias' in model_inputs:
                attention_values = attention_values + model_inputs['attention_values_bias']
    else:
        w_syn = ws
    kwargs = {'noise_mode': 'const'} if model_inputs.get('freeze_noise') else {}
    if 'path_length' in request_model_outputs:
        # second-order output: keep the (once-differentiable) fused hand-off node out of this call's graph
        from . import handoff
        with handoff.unfused(self.synthesis_network):
            planes = self.synthesis_network(w_syn, **kwargs)
    else:
        planes = self.synthesis_network(w_syn, **kwargs)
    planes = planes.view(batch, 3, 32, planes.shape[-2], planes.shape[-1])

    model_outputs = {}
    texel_cache = {}                 # the transposed texels of THIS forward (sampler + regulariser branch), see texels_of
    if 'attention_values' in request_model_outputs:
        assert self.attention_values > 0
        model_outputs['attention_values'] = attention_values
    if 'path_length' in request_model_outputs:
        # path-length regulariser of the plane producer (generator.py:484-499); lives on the
        # producer side of the hand-off, plain autograd through the synthesis network
        scale = 1.0 / math.sqrt(planes.shape[-2] * planes.shape[-1])
        target = (planes * (torch.randn_like(planes) * scale)).sum()
        if self.attention_values > 0:
            target = target + (attention_values * torch.randn_like(attention_values)).sum()
        grad, = torch.autograd.grad(target, inputs=ws, create_graph=True)
        model_outputs['path_length'] = grad.square().sum(dim=-1).mean(dim=-1).sqrt()

    if any(r in request_model_outputs for r in ('sdf_eikonal_loss', 'total_variation_loss', 'entropy_loss')):
        # (as in the reference, 'sdf_distance_loss' is only produced together with one of these three, 505-506 / 542)
        model_outputs.update(regulariser_outputs(self, planes, request_model_outputs, texel_cache))

    if 'sampler' in request_model_outputs:
        vd = None
        if self.use_viewdir and viewdir is not None:
            # generator.py:468-469: the per-ray MLP stays PyTorch; only its output and its last layer are handed over
            with _capture_ray_feature(self.viewdir_mapper) as cap:
                self.viewdir_mapper(viewdir)
            vd = (cap['x'], self.viewdir_mapper.output)
        model_outputs['sampler'] = make_sampler(
            planes, self.decoder, self.scene_range, self.attention_values, attention_values, self.use_sdf,
            self.beta if self.use_sdf else None, self.alpha if self.use_sdf else None,
            texel_dtype=getattr(self, 'nfi_texel_dtype', ops.TEXEL_F32), request_model_outputs=request_model_outputs,
            viewdir=vd, texel_cache=texel_cache)
    return model_outputs


class _capture_ray_feature:
    """Context manager: forward hook on ViewDirectionMapper.fc6 recording its output (the per-ray feature the
    mapper's closure adds to the decoder features, generator.py:237-247)."""

    def __init__(self, mapper):
        self.mapper = mapper
        self.store = {}

    def __enter__(self):
        self.hook = self.mapper.fc6.register_forward_hook(lambda m, i, o: self.store.__setitem__('x', o))
        return self.store

    def __exit__(self, *a):
        self.hook.remove()


def wrapped_forward(self, viewdir, c, request_model_outputs=['sampler'], model_inputs={}):
    """Forward for a module that HAS the reference's own Generator.forward (models/generator.py:407-686):
    the original forward runs unchanged for everything that is not the hot path - latent handling,
    mapping / synthesis / texture networks, path-length and the SDF regulariser branch (505-585, plain
    PyTorch incl. its double backward) - while a forward hook captures the planes it produced; only the
    returned ``sampler`` closure is replaced by the HIP one.  The plane producer therefore runs once."""
    want_sampler = 'sampler' in request_model_outputs
    req = list(request_model_outputs)
    # opt-in (attach(..., hip_regularisers=True)): the regulariser branch on the HIP kernels as well; the names are
    # then withheld from the original forward (which still runs the plane producer) and filled in afterwards
    reg_all = ('sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss')
    hip_reg = getattr(self, 'nfi_hip_regularisers', False) and any(
        r in req for r in ('sdf_eikonal_loss', 'total_variation_loss', 'entropy_loss'))
    if hip_reg:
        req = [r for r in req if r not in reg_all]
    added_att = False
    if want_sampler and self.attention_values > 0 and 'attention_values' not in req:
        req.append('attention_values')           # the original forward returns the (possibly overridden) table
        added_att = True
    captured = {}
    hook = self.synthesis_network.register_forward_hook(lambda mod, inp, out: captured.__setitem__('planes', out))
    use_vd = bool(self.use_viewdir) and viewdir is not None
    import contextlib
    from . import handoff
    with contextlib.ExitStack() as stack:
        stack.callback(hook.remove)
        if 'path_length' in req:
            # differentiated twice (generator.py:484-499): the fused hand-off node is first-order only
            stack.enter_context(handoff.unfused(self.synthesis_network))
        cap = stack.enter_context(_capture_ray_feature(self.viewdir_mapper)) if use_vd else None
        model_outputs = self._nfi_original_forward(viewdir, c, req, model_inputs)
    planes = captured.get('planes')
    if planes is not None:
        planes = planes.view(planes.shape[0], 3, 32, planes.shape[-2], planes.shape[-1])
    texel_cache = {}
    if hip_reg:
        model_outputs.update(regulariser_outputs(self, planes, request_model_outputs, texel_cache))
    if want_sampler:
        att = model_outputs.get('attention_values') if self.attention_values > 0 else None
        model_outputs['sampler'] = make_sampler(
            planes, self.decoder, self.scene_range, self.attention_values, att, self.use_sdf,
            self.beta if self.use_sdf else None, self.alpha if self.use_sdf else None,
            texel_dtype=getattr(self, 'nfi_texel_dtype', ops.TEXEL_F32), request_model_outputs=request_model_outputs,
            viewdir=(cap['x'], self.viewdir_mapper.output) if use_vd else None, texel_cache=texel_cache)
    if added_att:
        del model_outputs['attention_values']
    return model_outputs


def attach(model, texel_dtype=ops.TEXEL_F32, hip_regularisers=False, fused_handoff=False):
    """Gives a reference-style Generator the HIP sampler.  Returns the same module.

    fused_handoff: also fuse the tail of the last synthesis block (upsample + torgb + add, stylegan.py:424-433) into
    the HIP kernel that writes texels directly (nerf_from_image_amd.handoff): no NCHW <-> channel-last pass remains.

    hip_regularisers: also serve sdf_eikonal / sdf_distance / total_variation / entropy losses of a module that has
    its own forward from the HIP kernels (bare containers always do).

    A module whose class implements ``forward`` itself (the reference ``Generator``) keeps that
    forward for the non-hot-path outputs and only gets its sampler swapped (``wrapped_forward``);
    a bare container of the required sub-modules gets the restated ``hip_forward``."""
    missing = [a for a in REQUIRED_ATTRS if not hasattr(model, a)]
    if missing:
        raise AttributeError('attach(): module lacks %s' % missing)
    model.nfi_texel_dtype = texel_dtype
    model.nfi_hip_regularisers = bool(hip_regularisers)
    if fused_handoff:
        from . import handoff
        handoff.fuse_last_block(model.synthesis_network)
    if type(model).forward is not torch.nn.Module.forward and not hasattr(model, '_nfi_original_forward'):
        model._nfi_original_forward = model.forward          # bound method of the reference class
        model.forward = types.MethodType(wrapped_forward, model)
    elif not hasattr(model, '_nfi_original_forward'):
        model.forward = types.MethodType(hip_forward, model)
    return model
