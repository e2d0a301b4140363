"""Drop-in for ``render`` in the reference's run.py (176-350): same 16-parameter signature, same
6-tuple, same module-level ``args`` / ``dataset_config`` globals.

    import nerf_from_image_amd.render as nfi_render
    nfi_render.configure(args, dataset_config)          # once, where run.py builds them
    render = nfi_render.render                           # replaces run.py's own def

Two execution paths, both HIP through the C ABI:
  * fused   - one persistent launch for the whole pipeline (no gradient, no per-sample extras);
  * staged  - one launch per stage through ``nerf_utils`` and the ``sampler`` closure, used when a
              gradient or normals/semantics/coords maps are requested.
Randomness follows the reference, in its order: ``torch.rand`` of [B,H,W,S] for the stratified
jitter (nerf_utils.py:115) BEFORE the model is called (its synthesis network draws noise of its own
in training), then ``torch.rand`` of [B*H*W,S] for the inverse-CDF draws (nerf_utils.py:202), even
in eval (``randomize`` defaults to True and no caller overrides it).
"""
import torch

from . import nerf_utils, ops

args = None
dataset_config = None
# opt-in, NOT parity: transmittance threshold below which the fused inference kernel stops marching a ray
# (0 = exact path; see ops.render_fwd(fast_termination=...) and DESIGN.md)
FAST_TERMINATION = 0.0


def configure(new_args, new_dataset_config):
    """Installs the globals run.py::render reads: args.{use_viewdir,use_sdf,attention_values,
    fine_sampling} and dataset_config['scene_range'|'white_background']."""
    global args, dataset_config
    args, dataset_config = new_args, new_dataset_config


def make_render(new_args, new_dataset_config):
    """A render function bound to its own args/dataset_config (for multi-config processes)."""
    def bound(*a, **k):
        return _render(new_args, new_dataset_config, *a, **k)
    return bound


def render(target_model, height, width, tform_cam2world, focal_length, center, bbox, model_input,
           depth_samples_per_ray, randomize=True, compute_normals=False, compute_semantics=False,
           compute_coords=False, extra_model_outputs=[], extra_model_inputs={}, force_no_cam_grad=False):
    if args is None or dataset_config is None:
        raise RuntimeError('nerf_from_image_amd.render.configure(args, dataset_config) has not been called')
    return _render(args, dataset_config, target_model, height, width, tform_cam2world, focal_length, center, bbox,
                   model_input, depth_samples_per_ray, randomize, compute_normals, compute_semantics, compute_coords,
                   extra_model_outputs, extra_model_inputs, force_no_cam_grad)


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def _render(cfg, dcfg, target_model, height, width, tform_cam2world, focal_length, center, bbox, model_input,
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

    # Order of the random draws as in the reference: the stratified jitter (rand_like inside
    # compute_query_points_from_rays, run.py:203-209) comes BEFORE target_model is called (whose synthesis network
    # draws its own noise in training), the inverse-CDF draw after it.
    rays = None
    viewdirs = None
    if cfg.use_viewdir:
        # run.py:192-219: the model needs the normalised ray directions before it can build the sampler
        rays = nerf_utils.get_ray_bundle_normalized(height, width, focal_length, tform_cam2world, bbox, center)
        viewdirs = (rays[1].detach() if force_no_cam_grad else rays[1]).unsqueeze(-2)
    noise_c = torch.rand((B, height, width, S), dtype=torch.float32, device=dev) if randomize else None

    model_outputs = target_model(viewdirs, model_input, ['sampler'] + extra_model_outputs, extra_model_inputs)
    sampler = model_outputs['sampler']
    del model_outputs['sampler']
    fused = getattr(sampler, 'fused', None)

    if fused is not None and plain and not cam_grad and not fused.requires_grad and S <= 128:
        # ---------------- fused inference path (the kernel generates the rays itself) ----------------
        noise_f = None
        if cfg.fine_sampling and randomize:
            noise_f = torch.rand([B * height * width, S], dtype=torch.float32, device=dev)
        out = ops.render_fwd(
            tform_cam2world.detach(), None if focal_length is None else focal_length.detach(), height, width, S,
            fused.texels, fused.decoder_image, scene_range, fused.n_attention,
            None if fused.attention_values is None else fused.attention_values.detach(), fused.use_sdf,
            None if fused.beta is None else fused.beta.detach(), None if fused.alpha is None else fused.alpha.detach(),
            bbox=None if bbox is None else bbox.detach(), center=None if center is None else center.detach(),
            noise_coarse=noise_c, noise_fine=noise_f, fine_sampling=bool(cfg.fine_sampling),
            white_background=bool(white), skip_missed_rays=True, ray_features=getattr(fused, 'ray_features', None),
            fast_termination=FAST_TERMINATION)
        return out['rgb'], out['depth'], out['mask'], None, None, model_outputs

    # ---------------- staged path (differentiable / extra maps) ----------------
    ray_origins, ray_directions = rays if rays is not None else nerf_utils.get_ray_bundle_normalized(
        height, width, focal_length, tform_cam2world, bbox, center)
    with torch.no_grad():
        near, far = nerf_utils.compute_near_far_planes(ray_origins.detach(), ray_directions.detach(), scene_range)
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
        query_fine = nerf_utils.points_on_rays(ray_origins, ray_directions, z_samples)
        if force_no_cam_grad:
            query_fine = query_fine.detach()
        sigma_f, rgb_f, normals_f, semantics_f, coords_f = unpack(sampler(query_fine, req, **hip_kw))
        extra_f = coords_f if coords_f is not None else semantics_f
        rgb_map, depth_map, mask, normal_map, extra_map = nerf_utils.merge_and_composite(
            ray_directions, depth_values, sigma, rgb, z_samples, sigma_f, rgb_f, normals, normals_f, extra, extra_f,
            white_background=white)
    else:
        rgb_map, depth_map, mask, normal_map, extra_map = nerf_utils.render_volume_density(
            sigma, rgb, ray_origins, ray_directions, depth_values, normals, extra, white_background=white)
    return rgb_map, depth_map, mask, normal_map, extra_map, model_outputs
