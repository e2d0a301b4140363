"""Tensor-level wrapper of nfi_field_query_bwd and the autograd closure of the sampler's field query."""
import torch

from . import _lib, ops
from .autograd import zeros_like_or


BINNED_SCATTER_MIN_POINTS = 1 << 16


def field_query_bwd(points, texels, decoder_image, w1, w2, scene_range, n_attention, attention_values, use_sdf, beta,
                    alpha, g_sigma, g_rgb, g_sdf=None, g_semantics=None, want_points=False, points_only=False,
                    normalize_points=False, viewdir=None, scatter_mode=None, ray_order=None):
    """Returns dict(g_texels [B,3,R,R,32], g_w1, g_b1, g_w2, g_b2, g_attention_values?, g_beta?, g_alpha?, g_points?).
    viewdir: None or dict(ray_features=padded [B,N,48], samples_per_ray, w3) for the --use_viewdir decoder
    (decoder_image from ops.decoder_pack_viewdir, w2 [33,64]); adds g_ray_features [B,N,32], g_w3, g_b3.
    ray_order: None or (samples_per_ray, rays_per_row) - the points are [rays][samples] with the rays in row-major image
    order (the fused render's stash): a locality hint for the kernel's walk, no effect on the result."""
    f = ops._f32c
    points = f(points, 'points')
    B, P = points.shape[0], points.shape[1]
    dev = points.device
    if texels.dtype != torch.float32 and viewdir is not None:
        raise NotImplementedError('field_query_bwd: the view-direction decoder needs fp32 texels')
    n_out = 1 + n_attention if n_attention > 0 else 4
    n3 = n_attention if n_attention > 0 else 3
    if viewdir is not None:
        n_out = 33
    lib = _lib.load()
    out = {}
    if not points_only:
        # fp32 gradient image in the texels' layout (16-bit texel storage: the gradient w.r.t. the rounded planes)
        out = {'g_texels': torch.zeros(texels.shape, dtype=torch.float32, device=dev),
               'g_w1': torch.zeros((64, 32), dtype=torch.float32, device=dev),
               'g_b1': torch.zeros((64,), dtype=torch.float32, device=dev),
               'g_w2': torch.zeros((n_out, 64), dtype=torch.float32, device=dev),
               'g_b2': torch.zeros((n_out,), dtype=torch.float32, device=dev)}
        if n_attention > 0:
            out['g_attention_values'] = torch.zeros((B, n_attention, 3), dtype=torch.float32, device=dev)
        if use_sdf:
            out['g_beta'] = torch.zeros((1,), dtype=torch.float32, device=dev)
            out['g_alpha'] = torch.zeros((1,), dtype=torch.float32, device=dev)
    if want_points or points_only:
        out['g_points'] = torch.zeros((B, P, 3), dtype=torch.float32, device=dev)
    vd_args = {}
    if viewdir is not None:
        rf = f(viewdir['ray_features'], 'ray_features')
        vd_args = dict(ray_features=rf, samples_per_ray=int(viewdir['samples_per_ray']), w3=f(viewdir['w3'], 'w3'))
        if not points_only:
            out['g_w3'] = torch.zeros((n3, 32), dtype=torch.float32, device=dev)
            out['g_b3'] = torch.zeros((n3,), dtype=torch.float32, device=dev)
            vd_args['g_ray_features'] = torch.zeros_like(rf)
    if ray_order is not None and viewdir is None:
        vd_args = dict(samples_per_ray=int(ray_order[0]), rays_per_row=int(ray_order[1]))
    if scatter_mode is None:
        # binned plane-gradient scatter pays once the counting sort is amortised (dense renders); tiny queries
        # (tests, the regulariser's 31^3 probes) scatter straight from the kernel
        scatter_mode = 1 if (P >= BINNED_SCATTER_MIN_POINTS and not points_only) else 0
    fields = dict(
            n_scenes=B, points_per_scene=P,
            points=points, texels=texels, plane_res=ops.texel_res(texels), texel_dtype=ops.texel_dtype_of(texels),
            texel_layout=ops.texel_layout_of(texels),
            decoder_image=decoder_image, w1=f(w1, 'w1'), w2=f(w2, 'w2'), n_attention=n_attention,
            attention_values=f(attention_values, 'attention_values') if n_attention > 0 else None,
            use_sdf=int(use_sdf), beta=f(beta, 'beta') if use_sdf else None, alpha=f(alpha, 'alpha') if use_sdf else None,
            scene_range=float(scene_range), g_sigma=f(g_sigma, 'g_sigma'), g_rgb=f(g_rgb, 'g_rgb'),
            g_sdf=f(g_sdf, 'g_sdf'), g_semantics=f(g_semantics, 'g_semantics'),
            points_only=int(points_only), normalize_g_points=int(normalize_points), scatter_mode=int(scatter_mode),
            **vd_args, **out)
    n_ws = _lib.struct_query('nfi_field_bwd_workspace_bytes', 'nfi_field_bwd_args', **fields)
    ws = torch.empty(((n_ws + 3) // 4,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_field_query_bwd', 'nfi_field_bwd_args', ops._stream(points), workspace=ws,
                         workspace_bytes=ws.numel() * 4, **fields)
    if 'g_ray_features' in vd_args:
        out['g_ray_features'] = vd_args['g_ray_features'][..., 1:33]
    return out


def make_field_bwd(texels, decoder_image, scene_range, n_attention, use_sdf, want_sdf, want_sem, viewdir_pad=None,
                   samples_per_ray=0):
    """Backward closure for ``autograd.differentiable('field_query', ...)``.
    inputs = (points, planes, w1, b1, w2, b2, attention_values, beta, alpha[, ray_feature, w3, b3]);
    outputs = (sigma, rgb[, sdf][, semantics]).  viewdir_pad: padded ray features of the --use_viewdir decoder."""
    def bwd(inputs, outputs, grads, needs):
        pts, planes, w1, b1, w2, b2, att, be, al = inputs[:9]
        vd = None
        if viewdir_pad is not None:
            vd = dict(ray_features=viewdir_pad, samples_per_ray=samples_per_ray, w3=inputs[10])
        g_sigma = zeros_like_or(grads[0], outputs[0])
        g_rgb = zeros_like_or(grads[1], outputs[1])
        i = 2
        g_sdf = g_sem = None
        if want_sdf:
            g_sdf = grads[i]
            i += 1
        if want_sem:
            g_sem = grads[i]
        g = field_query_bwd(pts, texels, decoder_image, w1, w2, scene_range, n_attention, att, use_sdf, be, al,
                            g_sigma, g_rgb, g_sdf, g_sem, want_points=bool(needs[0]), viewdir=vd)
        g_planes = ops.texel_grad_to_planes(g['g_texels']) if needs[1] else None
        base = (g.get('g_points'), g_planes, g['g_w1'], g['g_b1'], g['g_w2'], g['g_b2'],
                g.get('g_attention_values'), g.get('g_beta'), g.get('g_alpha'))
        if vd is None:
            return base
        return base + (g['g_ray_features'].reshape(inputs[9].shape), g['g_w3'], g['g_b3'])
    return bwd


def surface_normals(points, texels, decoder_image, w1, w2, scene_range, n_attention, attention_values, use_sdf, beta,
                    alpha, viewdir=None):
    """normalize(d sdf / d x) per point [B,P,3]: the `normals` output of the sampler closure
    (models/generator.py:599-623), from the coordinate-gradient path of the backward kernel."""
    B, P = points.shape[0], points.shape[1]
    dev = points.device
    zeros = torch.zeros((B, P), dtype=torch.float32, device=dev)
    g = field_query_bwd(points, texels, decoder_image, w1, w2, scene_range, n_attention, attention_values, use_sdf,
                        beta, alpha, zeros, torch.zeros((B, P, 3), dtype=torch.float32, device=dev),
                        g_sdf=torch.ones((B, P), dtype=torch.float32, device=dev), points_only=True,
                        normalize_points=True, viewdir=viewdir)
    return g['g_points']
