"""Drop-in for ``render`` in the reference's run.py (176-350): same 16-parameter signature, same
6-tuple, same module-level ``args`` / ``dataset_config`` globals.

    import nerf_from_image_amd.render as nfi_render
    nfi_render.configure(args, dataset_config)          # once, where run.py builds them
    render = nfi_render.render                           # replaces run.py's own def

Three execution paths, all HIP through the C ABI:
  * fused          - one persistent launch for the whole pipeline (no gradient); the composited `semantics`, `coords` and
                     `normals` maps (compute_semantics / compute_coords / compute_normals: run.py:1250-1264, 1444-1454,
                     1639-1646, 2036-2051) come out of the same launch;
  * fused + stash  - the SAME launch when a gradient is needed (training / inversion; with or without fine sampling, plain
                     or view-direction decoder): the kernel also
                     writes a per-sample stash (depths, sigma, rgb of the 2S samples of every ray, ray-major), and the
                     whole render is ONE autograd node whose backward is compositing backward on the stash -> one field
                     backward launch over the 2S points of every ray (+ its binned plane-gradient scatter) -> ray /
                     camera backward.  Replaces the ~20 launches of the staged graph;
  * staged         - one launch per stage through ``nerf_utils`` and the ``sampler`` closure: extra maps with a gradient /
                     the 'bbox' overlay / extra maps of the view-direction decoder on 16-bit texels, or over a single pass
                     of more than 128 samples.
A single pass of up to 512 samples (run.py's inversion without --fine_sampling: depth_samples_per_ray * 4, run.py:2271)
takes the fused / fused + stash paths too (render_fwd_long_kernel).
Randomness follows the reference, in its order: ``torch.rand`` of [B,H,W,S] for the stratified
jitter (nerf_utils.py:115) BEFORE the model is called (its synthesis network draws noise of its own
in training), then ``torch.rand`` of [B*H*W,S] for the inverse-CDF draws (nerf_utils.py:202), even
in eval (``randomize`` defaults to True and no caller overrides it).

Options (``configure(..., **options)`` / ``make_render(..., **options)``; per bound render function, nothing
process-wide - the reference calls render from one thread per GPU):
  termination_eps   0 = off (default).  eps in (0,1): the fused inference kernel does not evaluate fine samples behind the
                    depth at which the COARSE transmittance has fallen below eps and compacts the rest by wave ballot
                    (coarse pass, pdf and sample indices untouched; |d rgb| <= ~eps; ops.render_fwd(termination_eps=...));
                    without fine sampling, for calls that ask for extra maps and with the view-direction decoder it has
                    nothing to act on and is ignored;
  strict_near_far   True (default, the reference's behaviour): every path raises when no ray of the batch meets the scene
                    cube (lib/nerf_utils.py:258 fails on min() of an empty selection; its boolean indexing synchronises
                    the host there too).  The fused paths read the hit counter of the ray set-up back BEFORE the render
                    kernel is launched: the host waits for the set-up kernels only, and a batch without a hit does not
                    pay for a render.  'after': the fused paths launch once and read the counter back behind the render
                    (no device idle between set-up and render; the host waits for the render).  'deferred': no
                    synchronisation at all, on any path - the counter travels to pinned host memory
                    behind the render and is looked at by the NEXT strict call on that device (or ops.flush_strict()),
                    which raises for the earlier batch; for serving / pipelined loops.  False: never raises, such a
                    batch renders as background.  With row_window the check is the whole image's: it runs only when
                    row_window_sync has summed the hit count over the bands (a band of background rows is legitimate);
                    without row_window_sync a windowed call does not check;
  row_window        None, or (row_offset, rows): render only these image rows (fused inference path; one image sharded
                    over the ranks of a node, parallel.shard_rows); the outputs then have `rows` rows;
  row_window_sync   False, or True / a process group: the ranks of the group render bands of the SAME image, and the
                    batch-wide miss-fill of lib/nerf_utils.py:258-259 is reduced over them between the ray set-up and the
                    render kernel (parallel.allreduce_ray_setup), so that every band is bit-identical to the same rows
                    of the full render.  Without it the fill is the band's own: identical only where no marched ray
                    misses the exact cube (the default cameras) - stated in include/nfi_hip.h.
"""
import types

import torch

from . import nerf_utils, ops
from .autograd import differentiable, zeros_like_or
from .field_backward import field_query_bwd

args = None
dataset_config = None
_DEFAULTS = dict(termination_eps=0.0, strict_near_far=True, row_window=None, row_window_sync=False)
options = types.SimpleNamespace(**_DEFAULTS)


def _options(kw):
    unknown = set(kw) - set(_DEFAULTS)
    if unknown:
        raise TypeError('unknown render option(s) %s (known: %s)' % (sorted(unknown), sorted(_DEFAULTS)))
    return types.SimpleNamespace(**dict(_DEFAULTS, **kw))


def configure(new_args, new_dataset_config, **new_options):
    """Installs the globals run.py::render reads: args.{use_viewdir,use_sdf,attention_values,
    fine_sampling} and dataset_config['scene_range'|'white_background'], plus the options above."""
    global args, dataset_config, options
    args, dataset_config, options = new_args, new_dataset_config, _options(new_options)


def make_render(new_args, new_dataset_config, **new_options):
    """A render function bound to its own args / dataset_config / options (multi-config processes, one per thread)."""
    opts = _options(new_options)

    def bound(target_model, height, width, tform_cam2world, focal_length, center, bbox, model_input,
              depth_samples_per_ray, randomize=True, compute_normals=False, compute_semantics=False,
              compute_coords=False, extra_model_outputs=[], extra_model_inputs={}, force_no_cam_grad=False):
        return _render(new_args, new_dataset_config, opts, target_model, height, width, tform_cam2world, focal_length,
                       center, bbox, model_input, depth_samples_per_ray, randomize, compute_normals, compute_semantics,
                       compute_coords, extra_model_outputs, extra_model_inputs, force_no_cam_grad)
    bound.options = opts                 # (read by graphs.GraphedRender: a strict render cannot be captured)
    return bound


def render(target_model, height, width, tform_cam2world, focal_length, center, bbox, model_input,
           depth_samples_per_ray, randomize=True, compute_normals=False, compute_semantics=False,
           compute_coords=False, extra_model_outputs=[], extra_model_inputs={}, force_no_cam_grad=False):
    if args is None or dataset_config is None:
        raise RuntimeError('nerf_from_image_amd.render.configure(args, dataset_config) has not been called')
    return _render(args, dataset_config, options, target_model, height, width, tform_cam2world, focal_length, center,
                   bbox, model_input, depth_samples_per_ray, randomize, compute_normals, compute_semantics, compute_coords,
                   extra_model_outputs, extra_model_inputs, force_no_cam_grad)


def _strict(opts):
    v = opts.strict_near_far
    return v if v in ('deferred', 'after') else bool(v)


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def _render_with_stash(fused, height, width, S, cam, focal, bbox, center, noise_c, noise_f, white, cam_grad, fine=True,
                       strict=False):
    """The fused render as ONE autograd node (see the module docstring).  Gradients follow the reference's graph:
    rgb_map / mask -> sigma, rgb of every sample and (through dists * ||rd||) the ray directions; depth_map and the
    depth samples carry none; the field -> planes, decoder, colour table, beta, alpha and - unless the camera is
    detached - the query points -> rays -> tform_cam2world / focal_length.  fine=False: one pass of S samples
    (run.py without --fine_sampling).  With the view-direction decoder (fused.ray_features: --use_viewdir, carla) the
    per-ray feature of the ViewDirectionMapper and its output layer are inputs of the node too: their gradients go
    back into the PyTorch mapper and, through its view directions, into the camera."""
    texels, image = fused.texels, fused.decoder_image
    A, use_sdf, scene_range = fused.n_attention, fused.use_sdf, fused.scene_range
    w1, b1, w2, b2 = fused.decoder_params[:4]
    vd = fused.ray_features is not None
    B = cam.shape[0]
    n_list = (2 if fine else 1) * S
    keep = {}

    def fwd(a_cam, a_focal, pl, a_w1, a_b1, a_w2, a_b2, att, be, al, *vd_in):
        out = ops.render_fwd(a_cam, a_focal, height, width, S, texels, image, scene_range, A, att, use_sdf, be, al,
                             bbox=bbox, center=center, noise_coarse=noise_c, noise_fine=noise_f, fine_sampling=fine,
                             white_background=bool(white), skip_missed_rays=True, stash=True,
                             ray_features=fused.ray_features, strict=strict)
        keep.update({k: out[k] for k in ('stash_t', 'stash_sigma', 'stash_rgb', 'ray_origins', 'ray_directions')})
        return out['rgb'], out['depth'], out['mask']

    def bwd(inputs, out_meta, grads, needs):
        a_cam, a_focal, pl, a_w1, a_b1, a_w2, a_b2, att, be, al = inputs[:10]
        g_rgb = zeros_like_or(grads[0], out_meta[0])
        g_mask = None if grads[2] is None else grads[2].contiguous()
        st_t, rd = keep['stash_t'], keep['ray_directions']
        full = cam_grad is True
        cb = ops.composite_bwd_stash(rd, st_t, keep['stash_sigma'], keep['stash_rgb'], g_rgb, g_mask,
                                     white_background=bool(white), want_rd=full, fine=fine)
        pts = ops.points_on_rays(keep['ray_origins'], rd, st_t).view(B, -1, 3)
        viewdir = dict(ray_features=fused.ray_features, samples_per_ray=n_list, w3=inputs[11]) if vd else None
        g = field_query_bwd(pts, texels, image, a_w1, a_w2, scene_range, A, att, use_sdf, be, al,
                            cb['g_sigma'].view(B, -1), cb['g_rgb'].view(B, -1, 3), want_points=bool(cam_grad),
                            viewdir=viewdir, ray_order=(n_list, width))
        g_cam = g_focal = None
        if full:
            g_ro, g_rd = ops.points_bwd(g['g_points'].view(*st_t.shape, 3), st_t)
            g_rd = g_rd + cb['g_ray_directions']
            g_cam, g_focal = ops.raygen_bwd(height, width, a_focal, a_cam, bbox, center, True, g_ro, g_rd)
        elif cam_grad == 'fine_origins':
            # force_no_cam_grad: only the origins of the FINE samples (stash columns S ...) are attached in the reference
            g_pts = g['g_points'].view(*st_t.shape, 3).clone()
            g_pts[..., :S, :] = 0.0
            g_ro, _ = ops.points_bwd(g_pts, st_t, want_rd=False)
            g_cam, _ = ops.raygen_bwd(height, width, a_focal, a_cam, bbox, center, True, g_ro, None)
        g_planes = ops.texel_grad_to_planes(g['g_texels']) if needs[2] else None
        base = (g_cam, g_focal, g_planes, g['g_w1'], g['g_b1'], g['g_w2'], g['g_b2'], g.get('g_attention_values'),
                g.get('g_beta'), g.get('g_alpha'))
        if not vd:
            return base
        return base + (g['g_ray_features'].reshape(inputs[10].shape), g['g_w3'], g['g_b3'])

    cam_in = cam if cam_grad else cam.detach()
    focal_in = focal if (cam_grad is True or focal is None) else focal.detach()
    extra_in = tuple(fused.decoder_params[4:7]) if vd else ()          # (ray_feature [B,H,W,1,32], w3, b3)
    res = differentiable('render', fwd, cam_in, focal_in, fused.planes, w1, b1, w2, b2,
                         fused.attention_values if A > 0 else None, fused.beta if use_sdf else None,
                         fused.alpha if use_sdf else None, *extra_in, bwd=bwd, non_differentiable_outputs=(1,))
    return res


def _render(cfg, dcfg, opts, target_model, height, width, tform_cam2world, focal_length, center, bbox, model_input,
            depth_samples_per_ray, randomize=True, compute_normals=False, compute_semantics=False,
            compute_coords=False, extra_model_outputs=[], extra_model_inputs={}, force_no_cam_grad=False):
    S = depth_samples_per_ray
    if S > 512 or (S > 128 and cfg.fine_sampling):
        raise NotImplementedError('depth_samples_per_ray: at most 128 per pass with fine sampling, 512 without '
                                  '(run.py asks for 64 + 64, or 128 / 512 in one pass), got %d' % S)
    scene_range = dcfg['scene_range']
    white = dcfg['white_background']
    if compute_normals:
        assert cfg.use_sdf
    if compute_semantics:
        assert cfg.attention_values > 0

    B = tform_cam2world.shape[0]
    dev = tform_cam2world.device
    plain = not (compute_normals or compute_semantics or compute_coords)
    cam_grad = (not force_no_cam_grad) and _needs_grad(tform_cam2world, focal_length, bbox, center)
    if force_no_cam_grad and cfg.fine_sampling and _needs_grad(tform_cam2world, bbox, center):
        # run.py:211-214 detaches the COARSE query points, the depths and the ray directions - and then builds the fine pass's
        # points from the undetached ray origins (run.py:286-288): a camera that requires grad still receives
        # d loss / d origin of the fine samples (its translation; with an orthographic camera also its rotation).
        # Reproduced as is.
        cam_grad = "fine_origins"
    rows = height if opts.row_window is None else int(opts.row_window[1])

    # Order of the random draws as in the reference: the stratified jitter (rand_like inside
    # compute_query_points_from_rays, run.py:203-209) comes BEFORE target_model is called (whose synthesis network
    # draws its own noise in training), the inverse-CDF draw after it.
    rays = None
    viewdirs = None
    if cfg.use_viewdir:
        # run.py:192-219: the model needs the normalised ray directions before it can build the sampler
        rays = nerf_utils.get_ray_bundle_normalized(height, width, focal_length, tform_cam2world, bbox, center)
        viewdirs = (rays[1].detach() if force_no_cam_grad else rays[1]).unsqueeze(-2)
    noise_c = torch.rand((B, rows, width, S), dtype=torch.float32, device=dev) if randomize else None

    model_outputs = target_model(viewdirs, model_input, ['sampler'] + extra_model_outputs, extra_model_inputs)
    sampler = model_outputs['sampler']
    del model_outputs['sampler']
    fused = getattr(sampler, 'fused', None)
    ray_features = getattr(fused, 'ray_features', None)

    def inverse_cdf_draws():
        if not cfg.fine_sampling:
            return None
        if randomize:
            return torch.rand([B * rows * width, S], dtype=torch.float32, device=dev)
        return None            # the kernels take linspace(0, 1, S) themselves (nerf_utils.py:196-200)

    # semantics / coords / normals are composited by the fused kernel itself (any texel storage; with the view-direction
    # decoder: fp32 texels); the 'bbox' overlay (which edits sigma, generator.py:645-659) and 16-bit texels with the
    # view-direction decoder keep the staged path
    # (with the view-direction decoder the kernel composites semantics / coords / normals on fp32 texels)
    fused_maps = plain or (not getattr(fused, 'bbox_overlay', False) and
                           (ray_features is None or fused.texels.dtype == torch.float32))
    if compute_normals and fused is not None:
        # the sampler's own condition (generator.py:599-602; torch.is_grad_enabled() is autograd's business there)
        assert fused.use_sdf and not getattr(target_model, 'training', False)
    # (one pass of 129..512 samples - no fine sampling, checked above - has its own fused kernel: plain maps only)
    if fused is not None and fused_maps and not cam_grad and not fused.requires_grad and (S <= 128 or plain):
        # ---------------- fused inference path (the kernel generates the rays itself) ----------------
        window = None if opts.row_window is None else (int(opts.row_window[0]), height)
        extras = not plain
        ws = None
        if window is not None and opts.row_window_sync:
            from . import parallel
            ws = ops.render_setup(tform_cam2world.detach(), None if focal_length is None else focal_length.detach(), rows,
                                  width, scene_range, bbox=None if bbox is None else bbox.detach(),
                                  center=None if center is None else center.detach(), row_window=window)
            parallel.allreduce_ray_setup(ws, None if opts.row_window_sync is True else opts.row_window_sync)
        out = ops.render_fwd(
            tform_cam2world.detach(), None if focal_length is None else focal_length.detach(), rows, width, S,
            fused.texels, fused.decoder_image, scene_range, fused.n_attention,
            None if fused.attention_values is None else fused.attention_values.detach(), fused.use_sdf,
            None if fused.beta is None else fused.beta.detach(), None if fused.alpha is None else fused.alpha.detach(),
            bbox=None if bbox is None else bbox.detach(), center=None if center is None else center.detach(),
            noise_coarse=noise_c, noise_fine=inverse_cdf_draws(), fine_sampling=bool(cfg.fine_sampling),
            white_background=bool(white), skip_missed_rays=True, ray_features=ray_features,
            termination_eps=opts.termination_eps if (cfg.fine_sampling and not extras and ray_features is None) else 0.0,
            row_window=window, want_semantics=compute_semantics and not compute_coords, want_coords=compute_coords,
            want_normals=compute_normals, workspace=ws, rays_ready=ws is not None,
            # a band's own hit count says nothing about the image (the top rows of a centred object are all background):
            # the check applies to the whole image - every call without a window, windowed calls only once
            # row_window_sync has summed the count over the bands
            strict=_strict(opts) if (window is None or ws is not None) else False)
        # run.py:337-338: coords take the semantics slot of render_volume_density when both are asked for
        extra_map = out['coords'] if compute_coords else (out['semantics'] if compute_semantics else None)
        return out['rgb'], out['depth'], out['mask'], out.get('normals'), extra_map, model_outputs
    if opts.row_window is not None:
        raise NotImplementedError('row_window is an option of the fused inference path (no gradient, no extra maps)')

    if fused is not None and plain and (ray_features is None or fused.texels.dtype == torch.float32):
        # ---------------- fused render + stash as one differentiable node ----------------
        # (with or without fine sampling, plain or view-direction decoder; the view-direction rays of run.py:216-222 were
        #  computed above for the model - the kernel regenerates the same rays, and the camera gradient of the viewdirs
        #  flows through the PyTorch mapper into that first ray op)
        det = (lambda t: None if t is None else t.detach())
        rgb_map, depth_map, mask = _render_with_stash(
            fused, height, width, S, tform_cam2world, focal_length, det(bbox), det(center), noise_c, inverse_cdf_draws(),
            white, cam_grad, fine=bool(cfg.fine_sampling), strict=_strict(opts))
        return rgb_map, depth_map, mask, None, None, model_outputs

    # ---------------- staged path (extra maps with a gradient or over a pass of more than 128 samples) ----------------
    ray_origins, ray_directions = rays if rays is not None else nerf_utils.get_ray_bundle_normalized(
        height, width, focal_length, tform_cam2world, bbox, center)
    with torch.no_grad():
        # (the staged path has one place to look at the counter - behind the near / far launch: 'after' is True here)
        near, far = nerf_utils.compute_near_far_planes(ray_origins.detach(), ray_directions.detach(), scene_range,
                                                       strict=True if _strict(opts) == 'after' else _strict(opts))
    query_points, depth_values = nerf_utils.compute_query_points_from_rays(
        ray_origins, ray_directions, near, far, S, randomize=randomize, noise=noise_c)
    if force_no_cam_grad:
        query_points, depth_values = query_points.detach(), depth_values.detach()
        ray_directions = ray_directions.detach()

    req = ['sigma', 'rgb']
    if compute_normals:
        req.append('normals')
    if compute_semantics:
        req.append('semantics')
    if compute_coords:
        req.append('coords')
    shp = query_points.shape[:-1]

    def unpack(o):
        sig = o['sigma'].view(*shp)
        col = o['rgb'].view(*shp, 3)
        nor = o['normals'].view(*shp, -1) if compute_normals else None
        sem = o['semantics'].view(*shp, -1) if compute_semantics else None
        coo = o['coords'].view(*shp, -1) if compute_coords else None
        return sig, col, nor, sem, coo
    # our own closure (it carries .fused): the fused renderer's split-fp16 decoder arithmetic in this path too
    hip_kw = {'mlp_split_fp16': True} if hasattr(sampler, 'fused') else {}
    sigma, rgb, normals, semantics, coords = unpack(sampler(query_points, req, **hip_kw))

    extra = coords if coords is not None else semantics      # run.py:337-338: coords hijack the semantics slot
    if cfg.fine_sampling:
        with torch.no_grad():
            if randomize:
                u = torch.rand([B * height * width, S], dtype=torch.float32, device=dev)
            else:
                u = torch.linspace(0.0, 1.0, steps=S, dtype=torch.float32, device=dev).expand(B * height * width, S)
            z_samples, _ = ops.resample(sigma.detach(), ray_directions.detach(), depth_values.detach(), u)
            z_samples = z_samples.view(*depth_values.shape[:3], S)
        # (force_no_cam_grad: the directions are detached above, the ORIGINS are not - run.py:286-288 builds the fine points from
        #  them as they are, so their gradient reaches the camera)
        query_fine = nerf_utils.points_on_rays(ray_origins, ray_directions, z_samples)
        sigma_f, rgb_f, normals_f, semantics_f, coords_f = unpack(sampler(query_fine, req, **hip_kw))
        extra_f = coords_f if coords_f is not None else semantics_f
        rgb_map, depth_map, mask, normal_map, extra_map = nerf_utils.merge_and_composite(
            ray_directions, depth_values, sigma, rgb, z_samples, sigma_f, rgb_f, normals, normals_f, extra, extra_f,
            white_background=white)
    else:
        rgb_map, depth_map, mask, normal_map, extra_map = nerf_utils.render_volume_density(
            sigma, rgb, ray_origins, ray_directions, depth_values, normals, extra, white_background=white)
    return rgb_map, depth_map, mask, normal_map, extra_map, model_outputs
