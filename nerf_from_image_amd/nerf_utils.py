"""Drop-in for the reference's ``lib/nerf_utils.py`` (same names, same positional/keyword
arguments, same shapes), every function running as HIP kernels through the C ABI, forward and
backward.

Reference lines: cumprod_exclusive 20-25, get_ray_bundle 28-91, compute_query_points_from_rays
94-120, render_volume_density 123-161, render_volume_density_weights_only 164-180, sample_pdf
183-222, compute_near_far_planes 225-273 (all in lib/nerf_utils.py).

Randomness: like the reference, ``compute_query_points_from_rays(randomize=True)`` draws
``torch.rand`` of shape [..., S] and ``sample_pdf(deterministic=False)`` draws ``torch.rand([N, K])``
on the rays' device (same shapes and order, hence the same Philox stream as the reference's
PyTorch-ROCm run); the noise is then handed to the kernels.

Gradients: as in the reference's autograd graph, depth samples and depth_map carry none;
rgb/mask/extra maps differentiate w.r.t. sigma, rgb, extras and the ray directions; the normal map is
composited with DETACHED weights (lib/nerf_utils.py:146-147): its cotangent reaches the normals (and, on a
white background, the mask) but not sigma; query points differentiate w.r.t. ray origins/directions; rays
w.r.t. tform_cam2world and focal_length.  Every function is held against the live reference module in
tests/test_reference_nerf_utils.py (signatures, values, gradients).
"""
from typing import Optional

import torch

from . import ops
from .autograd import differentiable, zeros_like_or

def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """tf.math.cumprod(..., exclusive=True) along the last dim.  Kept for API completeness only:
    nothing in this package calls it (the running transmittance lives inside the weights and
    compositing kernels), so it is the one helper left as a plain tensor expression."""
    ones = torch.ones_like(tensor[..., :1])
    return torch.cat((ones, torch.cumprod(tensor[..., :-1], dim=-1)), dim=-1)


def _ray_bundle(name, height, width, focal_length, tform_cam2world, bbox, center, normalize):
    def fwd(cam, focal, bb, cen):
        return ops.raygen(height, width, focal, cam, bb, cen, normalize=normalize)

    def bwd(inputs, outputs, grads, needs):
        cam, focal, bb, cen = inputs
        g_ro, g_rd = grads
        if g_ro is None and g_rd is None:
            return None, None, None, None
        g_cam, g_focal = ops.raygen_bwd(height, width, focal, cam, bb, cen, normalize,
                                        None if g_ro is None else g_ro.contiguous(),
                                        None if g_rd is None else g_rd.contiguous())
        return g_cam, g_focal, None, None
    return differentiable(name, fwd, tform_cam2world, focal_length, bbox, center, bwd=bwd)


def get_ray_bundle(height: int, width: int, focal_length: Optional[torch.Tensor], tform_cam2world: torch.Tensor,
                   bbox: Optional[torch.Tensor], center: Optional[torch.Tensor] = None):
    """Returns (ray_origins, ray_directions), each [B,H,W,3]; directions are NOT normalised
    (the caller normalises, run.py:196).  focal_length=None selects the orthographic model."""
    return _ray_bundle('get_ray_bundle', height, width, focal_length, tform_cam2world, bbox, center, False)


def get_ray_bundle_normalized(height: int, width: int, focal_length: Optional[torch.Tensor],
                              tform_cam2world: torch.Tensor, bbox: Optional[torch.Tensor],
                              center: Optional[torch.Tensor] = None):
    """get_ray_bundle followed by F.normalize(ray_directions, dim=-1) (run.py:193-196) in one launch."""
    return _ray_bundle('get_ray_bundle_normalized', height, width, focal_length, tform_cam2world, bbox, center, True)


def compute_near_far_planes(ray_origins: torch.Tensor, ray_directions: torch.Tensor, scene_range: float,
                            strict: bool = True):
    """Slab test against [-scene_range, scene_range]^3 with the reference's miss-fill, clamps and
    its failure when no ray hits.  No gradient (the reference detaches its inputs).

    strict (not in the reference's signature; default = its behaviour): raise when no ray of the batch meets the scene
    cube - the reference fails on min() of an empty selection (lib/nerf_utils.py:258).  The check reads one counter
    back from the device, i.e. one host synchronisation per call; a training loop that cannot see such a batch may pass
    False (render option strict_near_far)."""
    near, far, _ = ops.near_far(ray_origins.detach(), ray_directions.detach(), scene_range, strict=strict)
    return near, far


def _points_bwd(depth):
    def bwd(inputs, outputs, grads, needs):
        g_points = grads[0]
        if g_points is None:
            return None, None
        return ops.points_bwd(g_points.contiguous(), depth, want_ro=True, want_rd=True)
    return bwd


def compute_query_points_from_rays(ray_origins: torch.Tensor, ray_directions: torch.Tensor, near_thresh: torch.Tensor,
                                   far_thresh: torch.Tensor, num_samples: int, randomize: bool = True,
                                   noise: Optional[torch.Tensor] = None):
    """Returns (query_points [...,S,3], depth_values [...,S]).  noise (extension): the [...,S] jitter already drawn
    by the caller (render draws it before the model call to keep the reference's RNG order)."""
    if near_thresh.dim() != ray_origins.dim() - 1:
        raise NotImplementedError('per-batch scalar near/far planes are not used by run.py::render and not supported')
    if randomize and noise is None:
        noise = torch.rand((*near_thresh.shape, num_samples), dtype=torch.float32, device=near_thresh.device)
    if not randomize:
        noise = None
    # depth first (no gradient), then the points as a differentiable function of the rays
    _, depth = ops.stratified_points(ray_origins.detach(), ray_directions.detach(), near_thresh.detach(),
                                     far_thresh.detach(), num_samples, noise, want_points=False)
    return points_on_rays(ray_origins, ray_directions, depth), depth


def points_on_rays(ray_origins: torch.Tensor, ray_directions: torch.Tensor, depth_values: torch.Tensor):
    """ray_origins[..., None, :] + ray_directions[..., None, :] * depth[..., :, None] (run.py:286-288)."""
    depth = depth_values.detach()

    def fwd(ro, rd):
        return ops.points_on_rays(ro, rd, depth)
    return differentiable('points_on_rays', fwd, ray_origins, ray_directions, bwd=_points_bwd(depth))


def render_volume_density_weights_only(sigma_a: torch.Tensor, ray_origins: torch.Tensor, ray_directions: torch.Tensor,
                                       depth_values: torch.Tensor) -> torch.Tensor:
    """Per-sample weights.  run.py calls this under no_grad (run.py:261); no backward is provided."""
    def fwd(sig, rd, dep):
        return ops.ray_weights(sig, rd, dep)
    return differentiable('render_volume_density_weights_only', fwd, sigma_a, ray_directions, depth_values)


def sample_pdf(bins, weights, num_samples: int, deterministic: bool = False) -> torch.Tensor:
    """bins [N,M], weights [N,M-1] -> samples [N,num_samples] (no gradient is defined through the
    index search; the reference calls it under no_grad, run.py:261)."""
    n = bins.shape[0]
    if deterministic:
        u = torch.linspace(0.0, 1.0, steps=num_samples, dtype=weights.dtype, device=weights.device)
        u = u.expand(n, num_samples)
    else:
        u = torch.rand([n, num_samples], dtype=weights.dtype, device=weights.device)
    samples, _, _ = ops.sample_pdf(bins.detach(), weights.detach(), u)
    return samples


def _composite(name, ray_directions, depth_a, sigma_a, rgb_a, depth_b, sigma_b, rgb_b, normals_a, normals_b,
               extra_a, extra_b, white_background):
    """Shared body of render_volume_density (one list) and merge_and_composite (two lists)."""
    two = depth_b is not None
    if two and torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (normals_a, normals_b)):
        raise NotImplementedError('merge_and_composite: normals that require grad are not supported (the sampler returns '
                                  'them detached, models/generator.py:612-621)')
    ex_a = [t for t in (normals_a, extra_a) if t is not None]
    ex_b = [t for t in (normals_b, extra_b) if t is not None] if two else []
    n_norm = normals_a.shape[-1] if normals_a is not None else 0
    n_ex = len(ex_a)
    da = depth_a.detach()
    db = depth_b.detach() if two else None

    def split(args):
        rd, sa, ca = args[0], args[1], args[2]
        sb, cb = (args[3], args[4]) if two else (None, None)
        ex = args[5:] if two else args[3:]
        ea = eb = None
        if n_ex:
            ea = torch.cat(ex[:n_ex], dim=-1) if n_ex > 1 else ex[0]
            if two:
                eb = torch.cat(ex[n_ex:], dim=-1) if n_ex > 1 else ex[n_ex]
        return rd, sa, ca, sb, cb, ea, eb

    def fwd(*args):
        rd, sa, ca, sb, cb, ea, eb = split(args)
        rgb_map, depth_map, mask, extra_map, _ = ops.composite(rd, da, sa, ca, db, sb, cb, extra_a=ea, extra_b=eb,
                                                               white_background=white_background)
        return (rgb_map, depth_map, mask) + ((extra_map,) if extra_map is not None else ())

    def bwd(inputs, outputs, grads, needs):
        rd, sa, ca, sb, cb, ea, eb = split(inputs)
        g_rgb = zeros_like_or(grads[0], outputs[0])
        g_mask = None if grads[2] is None else grads[2].contiguous()
        g_ex = g_normal_map = None
        if n_ex and len(grads) > 3 and grads[3] is not None:
            g_ex = grads[3].contiguous()
            if n_norm:
                # the reference composites normals with weights.detach() (lib/nerf_utils.py:146-147): what arrives on the
                # normal channels must not reach sigma / the directions through the weights (the backward kernel's extras
                # path differentiates through them, which is right for semantics / coords only)
                g_normal_map = g_ex[..., :n_norm]
                g_ex = g_ex.clone()
                g_ex[..., :n_norm] = 0
        g = ops.composite_bwd(rd, da, sa, ca, g_rgb, g_mask, db, sb, cb, ea, eb, g_ex, white_background, want_rd=True)
        out = [g['g_ray_directions'], g['g_sigma_a'], g['g_rgb_a']]
        if two:
            out += [g['g_sigma_b'], g['g_rgb_b']]

        def unsplit(ge, parts):
            if ge is None:
                return [None] * len(parts)
            if len(parts) == 1:
                return [ge]
            sizes = [p.shape[-1] for p in parts]
            return list(torch.split(ge, sizes, dim=-1))
        if n_ex:
            out += unsplit(g.get('g_extra_a'), ex_a)
            if two:
                out += unsplit(g.get('g_extra_b'), ex_b)
            elif n_norm and needs[3] and g_normal_map is not None:
                # d normal_map / d normals = the (detached) weights; one list only (render_volume_density)
                out[3] = ops.ray_weights(sa, rd, da)[..., None] * g_normal_map[..., None, :]
        return tuple(out)

    tensors = [ray_directions, sigma_a, rgb_a] + ([sigma_b, rgb_b] if two else []) + ex_a + ex_b
    out = differentiable(name, fwd, *tensors, bwd=bwd, non_differentiable_outputs=(1,))
    rgb_map, depth_map, mask = out[0], out[1], out[2]
    normal_map = extra_map = None
    if n_ex:
        em = out[3]
        if normals_a is not None:
            normal_map = em[..., :n_norm]
            if white_background:
                normal_map = normal_map + (1. - mask[..., None])
        if extra_a is not None:
            extra_map = em[..., n_norm:]
    return rgb_map, depth_map, mask, normal_map, extra_map


def render_volume_density(sigma_a: torch.Tensor, rgb: torch.Tensor, ray_origins: torch.Tensor,
                          ray_directions: torch.Tensor, depth_values: torch.Tensor,
                          normals: Optional[torch.Tensor] = None, semantics: Optional[torch.Tensor] = None,
                          white_background: bool = True):
    """Returns (rgb_map, depth_map, mask, normal_map, semantic_map)."""
    return _composite('render_volume_density', ray_directions, depth_values, sigma_a, rgb, None, None, None,
                      normals, None, semantics, None, white_background)


def merge_and_composite(ray_directions, depth_a, sigma_a, rgb_a, depth_b, sigma_b, rgb_b, normals_a=None,
                        normals_b=None, extra_a=None, extra_b=None, white_background=True):
    """The sort/merge of run.py:283-335 fused with render_volume_density: the coarse (a) and fine (b)
    sample lists are merged by depth inside the kernel (stable, a first on ties) and composited.
    Returns (rgb_map, depth_map, mask, normal_map, extra_map)."""
    return _composite('merge_and_composite', ray_directions, depth_a, sigma_a, rgb_a, depth_b, sigma_b, rgb_b,
                      normals_a, normals_b, extra_a, extra_b, white_background)
