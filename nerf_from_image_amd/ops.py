"""Tensor-level wrappers over the C ABI (one function per entry point of include/nfi_hip.h).

PyTorch is used here for device memory and streams only: every function checks
its inputs, allocates the outputs as torch tensors, and passes raw device
pointers plus ``torch.cuda.current_stream()`` to libnfi_hip.so.  Nothing here
computes on the CPU and nothing falls back to ATen ops.
"""
import torch

from . import _lib

TEXEL_F32 = 0
TEXEL_BF16 = 1
TEXEL_F16 = 2
_TEXEL_TORCH = {TEXEL_F32: torch.float32, TEXEL_BF16: torch.bfloat16, TEXEL_F16: torch.float16}


def texel_dtype_of(texels):
    for k, v in _TEXEL_TORCH.items():
        if texels.dtype == v:
            return k
    raise TypeError('texels must be float32, bfloat16 or float16, got %s' % texels.dtype)


TEXELS_PLANAR = 0          # [B,3,R,R,32]
TEXELS_INTERLEAVED = 1     # [B,R,R,3,32] = channels-last [B,96,R,R]


def texel_layout_of(texels):
    """Layout of a texel tensor from its shape: [B,3,R,R,32] planar, [B,R,R,3,32] interleaved (R > 3)."""
    if texels.dim() != 5 or texels.shape[4] != 32:
        raise ValueError('texels must be [B,3,R,R,32] or [B,R,R,3,32], got %s' % (tuple(texels.shape),))
    if texels.shape[1] == 3 and texels.shape[2] == texels.shape[3]:
        return TEXELS_PLANAR
    if texels.shape[3] == 3 and texels.shape[1] == texels.shape[2]:
        return TEXELS_INTERLEAVED
    raise ValueError('texels must be [B,3,R,R,32] or [B,R,R,3,32], got %s' % (tuple(texels.shape),))


def texel_res(texels):
    return texels.shape[2]          # R sits at index 2 in both layouts


def planes_view_as_texels(planes):
    """planes [B,3,32,R,R] (a view of the producer's [B,96,R,R] output): if that output is channels-last in memory
    (torch.channels_last, or written by torgb_texels), returns the zero-copy interleaved texel view [B,R,R,3,32];
    otherwise None (the caller then runs planes_to_texels)."""
    if planes.dim() != 5 or planes.shape[1] != 3 or planes.shape[2] != 32 or planes.shape[3] <= 3:
        return None
    v = planes.permute(0, 3, 4, 1, 2)
    return v if v.is_contiguous() else None


def texel_grad_to_planes(g_texels):
    """Gradient image in texel layout -> gradient of the [B,3,32,R,R] planes view (a kernel for the planar layout,
    a zero-copy strided view for the interleaved one)."""
    if texel_layout_of(g_texels) == TEXELS_INTERLEAVED:
        return g_texels.permute(0, 3, 4, 1, 2)
    return texels_to_planes(g_texels)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _f32c(t, name):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('%s must live on the GPU (no CPU path in nerf_from_image_amd)' % name)
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    return t.contiguous()


# --------------------------------------------------------------------------- #
def planes_to_texels(planes, texel_dtype=TEXEL_F32):
    """planes [B,3,32,R,R] fp32 -> texels [B,3,R,R,32] (fp32 or bf16)."""
    planes = _f32c(planes, 'planes')
    B, three, C, R, R2 = planes.shape
    if three != 3 or C != 32 or R != R2:
        raise ValueError('planes must be [B,3,32,R,R], got %s' % (tuple(planes.shape),))
    dt = _TEXEL_TORCH[texel_dtype]
    texels = torch.empty((B, 3, R, R, 32), dtype=dt, device=planes.device)
    lib = _lib.load()
    with torch.cuda.device(planes.device):
        _lib.check(lib.nfi_planes_to_texels(_lib.ptr(planes), _lib.ptr(texels), B, R, texel_dtype, _stream(planes)),
                   'nfi_planes_to_texels')
    return texels


def texels_to_planes(texels):
    """fp32 texels [B,3,R,R,32] -> planes [B,3,32,R,R] (adjoint layout change for gradients)."""
    texels = _f32c(texels, 'texels')
    B, three, R, R2, C = texels.shape
    planes = torch.empty((B, 3, C, R, R), dtype=torch.float32, device=texels.device)
    lib = _lib.load()
    with torch.cuda.device(texels.device):
        _lib.check(lib.nfi_texels_to_planes(_lib.ptr(texels), _lib.ptr(planes), B, R, _stream(texels)),
                   'nfi_texels_to_planes')
    return planes


def decoder_pack(w1, b1, w2, b2, n_attention, texel_dtype=TEXEL_F32):
    """Raw TriplanarDecoder parameters -> lane-ordered MFMA operand image (fp32 [3152])."""
    w1, b1, w2, b2 = (_f32c(t, n) for t, n in ((w1, 'w1'), (b1, 'b1'), (w2, 'w2'), (b2, 'b2')))
    n_out = 1 + n_attention if n_attention > 0 else 4
    if tuple(w1.shape) != (64, 32) or tuple(b1.shape) != (64,) or tuple(w2.shape) != (n_out, 64) \
            or tuple(b2.shape) != (n_out,):
        raise ValueError('decoder shapes must be [64,32],[64],[%d,64],[%d]; got %s %s %s %s' % (
            n_out, n_out, tuple(w1.shape), tuple(b1.shape), tuple(w2.shape), tuple(b2.shape)))
    lib = _lib.load()
    image = torch.empty((lib.nfi_decoder_image_floats(),), dtype=torch.float32, device=w1.device)
    with torch.cuda.device(w1.device):
        _lib.check(lib.nfi_decoder_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), n_attention,
                                        texel_dtype, _lib.ptr(image), _stream(w1)), 'nfi_decoder_pack')
    return image


RAY_FEATURE_PITCH = 48


def decoder_pack_viewdir(w1, b1, w2, b2, w3, b3, n_attention, texel_dtype=TEXEL_F32):
    """--use_viewdir decoder (33 outputs) + ViewDirectionMapper.output (w3, b3) -> operand image (fp32 [6016])."""
    ts = [_f32c(t, n) for t, n in ((w1, 'w1'), (b1, 'b1'), (w2, 'w2'), (b2, 'b2'), (w3, 'w3'), (b3, 'b3'))]
    n3 = n_attention if n_attention > 0 else 3
    want = [(64, 32), (64,), (33, 64), (33,), (n3, 32), (n3,)]
    if [tuple(t.shape) for t in ts] != want:
        raise ValueError('view-direction decoder shapes must be %s; got %s' % (want, [tuple(t.shape) for t in ts]))
    lib = _lib.load()
    image = torch.empty((lib.nfi_decoder_image_floats_viewdir(),), dtype=torch.float32, device=ts[0].device)
    with torch.cuda.device(ts[0].device):
        _lib.check(lib.nfi_decoder_pack_viewdir(*[_lib.ptr(t) for t in ts], n_attention, texel_dtype, _lib.ptr(image),
                                                _stream(ts[0])), 'nfi_decoder_pack_viewdir')
    return image


def pad_ray_features(x):
    """ViewDirectionMapper.fc6 output [..., 32] -> [..., 48] rows [0, x, 0 x 15] (the kernels' accumulator-row order)."""
    x = _f32c(x, 'ray_features')
    if x.shape[-1] != 32:
        raise ValueError('ray features must have 32 channels, got %s' % (tuple(x.shape),))
    return torch.nn.functional.pad(x, (1, RAY_FEATURE_PITCH - 33)).contiguous()


# --------------------------------------------------------------------------- #
def raygen(height, width, focal, cam2world, bbox=None, center=None, normalize=False):
    cam2world = _f32c(cam2world, 'tform_cam2world')
    B = cam2world.shape[0]
    dev = cam2world.device
    focal, bbox, center = _f32c(focal, 'focal_length'), _f32c(bbox, 'bbox'), _f32c(center, 'center')
    ro = torch.empty((B, height, width, 3), dtype=torch.float32, device=dev)
    rd = torch.empty_like(ro)
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_raygen', 'nfi_raygen_args', _stream(cam2world), n_scenes=B, height=height, width=width,
                         cam2world=cam2world, focal=focal, bbox=bbox, center=center, normalize=int(normalize),
                         ray_origins=ro, ray_directions=rd)
    return ro, rd


def near_far(ray_origins, ray_directions, scene_range, strict=True):
    """Returns near, far (finished planes), hit (bool).  strict: True - raise when no ray hits, as the reference does
    (costs a device->host read of one counter); 'deferred' - no synchronisation, the counter is looked at by the next
    strict call on the device / flush_strict() (as in render_fwd); False - never raises."""
    ro, rd = _f32c(ray_origins, 'ray_origins'), _f32c(ray_directions, 'ray_directions')
    shape = ro.shape[:-1]
    n = ro.numel() // 3
    dev = ro.device
    near_raw = torch.empty((n,), dtype=torch.float32, device=dev)
    far_raw = torch.empty_like(near_raw)
    near, far = torch.empty_like(near_raw), torch.empty_like(near_raw)
    hit = torch.empty((n,), dtype=torch.uint8, device=dev)
    red = torch.empty((4,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_near_far', 'nfi_near_far_args', _stream(ro), n_rays=n, ray_origins=ro,
                         ray_directions=rd, scene_range=float(scene_range), near_raw=near_raw, far_raw=far_raw,
                         hit=hit, reduce=red, near_plane=near, far_plane=far)
    if strict == 'deferred':
        _raise_pending(dev)
        _defer_hit_count(red[2:3], dev)
    elif strict and int(red[2].item()) == 0:
        raise RuntimeError(_NO_HIT)
    return near.view(shape), far.view(shape), (hit & 1).bool().view(shape)


def stratified_points(ray_origins, ray_directions, near, far, num_samples, noise=None, want_points=True):
    ro, rd = _f32c(ray_origins, 'ray_origins'), _f32c(ray_directions, 'ray_directions')
    near, far, noise = _f32c(near, 'near'), _f32c(far, 'far'), _f32c(noise, 'noise')
    shape = ro.shape[:-1]
    n = ro.numel() // 3
    depth = torch.empty((*shape, num_samples), dtype=torch.float32, device=ro.device)
    points = torch.empty((*shape, num_samples, 3), dtype=torch.float32, device=ro.device) if want_points else None
    with torch.cuda.device(ro.device):
        _lib.call_struct('nfi_stratified_points', 'nfi_stratified_args', _stream(ro), n_rays=n,
                         n_samples=num_samples, ray_origins=ro, ray_directions=rd, near_plane=near, far_plane=far,
                         noise=noise, depth=depth, points=points)
    return points, depth


def points_on_rays(ray_origins, ray_directions, depth):
    ro, rd, depth = _f32c(ray_origins, 'ray_origins'), _f32c(ray_directions, 'ray_directions'), _f32c(depth, 'depth')
    S = depth.shape[-1]
    n = depth.numel() // S
    points = torch.empty((*depth.shape, 3), dtype=torch.float32, device=depth.device)
    lib = _lib.load()
    with torch.cuda.device(depth.device):
        _lib.check(lib.nfi_points_on_rays(_lib.ptr(ro), _lib.ptr(rd), _lib.ptr(depth), n, S, _lib.ptr(points),
                                          _stream(depth)), 'nfi_points_on_rays')
    return points


# --------------------------------------------------------------------------- #
def field_query(points, texels, decoder_image, scene_range, n_attention, attention_values=None, use_sdf=True,
                beta=None, alpha=None, want_sdf=False, want_semantics=False, want_outside=False,
                ray_features=None, samples_per_ray=0, mlp_precision=0):
    """points [B,P,3] -> dict(sigma [B,P], rgb [B,P,3], sdf?, semantics?, outside?).
    mlp_precision: 0 exact fp32 MFMA, 1 split-fp16 operands (the fused renderer's arithmetic; not with ray_features).
    ray_features: padded [B, P/samples_per_ray, 48] (pad_ray_features) with a decoder_pack_viewdir image."""
    points = _f32c(points, 'points')
    B, P = points.shape[0], points.shape[1]
    if ray_features is not None:
        ray_features = _f32c(ray_features, 'ray_features')
        if samples_per_ray <= 0 or P % samples_per_ray or \
                tuple(ray_features.shape) != (B, P // samples_per_ray, RAY_FEATURE_PITCH):
            raise ValueError('ray_features must be [B, P/samples_per_ray, 48]; got %s for P=%d, S=%d' % (
                tuple(ray_features.shape), P, samples_per_ray))
    dev = points.device
    tdt = texel_dtype_of(texels)
    att = _f32c(attention_values, 'attention_values') if n_attention > 0 else None
    out = {'sigma': torch.empty((B, P), dtype=torch.float32, device=dev),
           'rgb': torch.empty((B, P, 3), dtype=torch.float32, device=dev)}
    if want_sdf:
        out['sdf'] = torch.empty((B, P), dtype=torch.float32, device=dev)
    if want_semantics:
        out['semantics'] = torch.empty((B, P, n_attention), dtype=torch.float32, device=dev)
    if want_outside:
        out['outside'] = torch.empty((B, P), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_field_query_fwd', 'nfi_field_args', _stream(points), n_scenes=B, points_per_scene=P,
                         points=points, texels=texels, plane_res=texel_res(texels), texel_dtype=tdt,
                         texel_layout=texel_layout_of(texels),
                         decoder_image=decoder_image, n_attention=n_attention, attention_values=att,
                         use_sdf=int(use_sdf), beta=_f32c(beta, 'beta') if use_sdf else None,
                         alpha=_f32c(alpha, 'alpha') if use_sdf else None, scene_range=float(scene_range),
                         sigma=out['sigma'], rgb=out['rgb'], sdf=out.get('sdf'), semantics=out.get('semantics'),
                         outside=out.get('outside'), ray_features=ray_features, samples_per_ray=int(samples_per_ray),
                         mlp_precision=int(mlp_precision))
    return out


def bbox_overlay(points, sigma, scene_range):
    """sigma + 100 on the wire frame of the scene cube (generator.py:645-659)."""
    import numpy as np
    points, sigma = _f32c(points, 'points'), _f32c(sigma, 'sigma')
    n = sigma.numel()
    if points.numel() != 3 * n:
        raise ValueError('bbox_overlay: points %s do not match sigma %s' % (tuple(points.shape), tuple(sigma.shape)))
    out = torch.empty_like(sigma)
    thr = float(np.float32(scene_range - 5e-2))          # `x < scene_range - eps` against an fp32 tensor
    with torch.cuda.device(sigma.device):
        _lib.check(_lib.load().nfi_bbox_overlay(_lib.ptr(points), n, float(scene_range), thr, _lib.ptr(sigma),
                                                _lib.ptr(out), _stream(sigma)), 'nfi_bbox_overlay')
    return out


# --------------------------------------------------------------------------- #
def ray_weights(sigma, ray_directions, depth):
    sigma, rd, depth = _f32c(sigma, 'sigma'), _f32c(ray_directions, 'ray_directions'), _f32c(depth, 'depth')
    S = depth.shape[-1]
    n = depth.numel() // S
    w = torch.empty_like(depth)
    with torch.cuda.device(depth.device):
        _lib.call_struct('nfi_ray_weights', 'nfi_weights_args', _stream(depth), n_rays=n, n_samples=S, sigma=sigma,
                         ray_directions=rd, depth=depth, weights=w)
    return w


def _u_and_stride(u, n, k):
    """u: [n,k] dense, or a stride-0 broadcast of one row."""
    if u.dim() == 2 and u.stride(0) == 0 and u.stride(1) == 1:
        return u, 0
    u = u.contiguous()
    return u, k


def sample_pdf(bins, weights, u, want_inds=False, want_cdf=False):
    bins, weights = _f32c(bins, 'bins'), _f32c(weights, 'weights')
    n, M = bins.shape
    K = u.shape[-1]
    u, stride = _u_and_stride(u, n, K)
    dev = bins.device
    samples = torch.empty((n, K), dtype=torch.float32, device=dev)
    inds = torch.empty((n, K), dtype=torch.int64, device=dev) if want_inds else None
    cdf = torch.empty((n, M), dtype=torch.float32, device=dev) if want_cdf else None
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_sample_pdf', 'nfi_sample_pdf_args', _stream(bins), n_rays=n, n_bins=M, n_samples=K,
                         bins=bins, weights=weights, u=u, u_row_stride=stride, samples=samples, inds=inds, cdf=cdf)
    return samples, inds, cdf


def resample(sigma, ray_directions, depth, u, want_taps=False):
    sigma, rd, depth = _f32c(sigma, 'sigma'), _f32c(ray_directions, 'ray_directions'), _f32c(depth, 'depth')
    S = depth.shape[-1]
    n = depth.numel() // S
    u, stride = _u_and_stride(u.reshape(-1, S) if u.stride(0) != 0 else u, n, S)
    dev = depth.device
    fine = torch.empty((n, S), dtype=torch.float32, device=dev)
    taps = {}
    if want_taps:
        taps = dict(weights=torch.empty((n, S), dtype=torch.float32, device=dev),
                    smooth=torch.empty((n, S), dtype=torch.float32, device=dev),
                    cdf=torch.empty((n, S - 1), dtype=torch.float32, device=dev),
                    inds=torch.empty((n, S), dtype=torch.int64, device=dev))
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_resample', 'nfi_resample_args', _stream(depth), n_rays=n, n_samples=S, sigma=sigma,
                         ray_directions=rd, depth=depth, u=u, u_row_stride=stride, fine_depth=fine, **taps)
    return fine, taps


def composite(ray_directions, depth_a, sigma_a, rgb_a, depth_b=None, sigma_b=None, rgb_b=None, extra_a=None,
              extra_b=None, white_background=True, want_taps=False):
    rd = _f32c(ray_directions, 'ray_directions')
    depth_a, sigma_a, rgb_a = _f32c(depth_a, 'depth'), _f32c(sigma_a, 'sigma'), _f32c(rgb_a, 'rgb')
    na = depth_a.shape[-1]
    n = depth_a.numel() // na
    nb = 0
    if depth_b is not None:
        depth_b, sigma_b, rgb_b = _f32c(depth_b, 'depth_b'), _f32c(sigma_b, 'sigma_b'), _f32c(rgb_b, 'rgb_b')
        nb = depth_b.shape[-1]
    n_extra = 0
    if extra_a is not None:
        extra_a = _f32c(extra_a, 'extra_a')
        n_extra = extra_a.shape[-1]
        extra_b = _f32c(extra_b, 'extra_b')
    dev = depth_a.device
    shape = depth_a.shape[:-1]
    rgb_map = torch.empty((*shape, 3), dtype=torch.float32, device=dev)
    depth_map = torch.empty(shape, dtype=torch.float32, device=dev)
    mask = torch.empty(shape, dtype=torch.float32, device=dev)
    extra_map = torch.empty((*shape, n_extra), dtype=torch.float32, device=dev) if n_extra else None
    taps = {}
    if want_taps:
        taps = dict(weights=torch.empty((*shape, na + nb), dtype=torch.float32, device=dev),
                    depth_sorted=torch.empty((*shape, na + nb), dtype=torch.float32, device=dev),
                    perm=torch.empty((*shape, na + nb), dtype=torch.int64, device=dev))
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_composite_fwd', 'nfi_composite_args', _stream(depth_a), n_rays=n, n_a=na, n_b=nb,
                         ray_directions=rd, depth_a=depth_a, sigma_a=sigma_a, rgb_a=rgb_a, depth_b=depth_b,
                         sigma_b=sigma_b, rgb_b=rgb_b, n_extra=n_extra, extra_a=extra_a, extra_b=extra_b,
                         white_background=int(white_background), rgb_map=rgb_map, depth_map=depth_map, mask=mask,
                         extra_map=extra_map, **taps)
    return rgb_map, depth_map, mask, extra_map, taps


# --------------------------------------------------------------------------- #
TAP_NAMES = ('ray_origins', 'ray_directions', 'near_plane', 'far_plane', 'hit', 't_coarse', 'sigma_coarse',
             'rgb_coarse', 't_fine', 'sigma_fine', 'rgb_fine', 't_sorted', 'weights', 'perm')


def render_setup(cam2world, focal, height, width, scene_range, bbox=None, center=None, workspace=None, row_window=None):
    """The ray set-up of render_fwd alone (nfi_render_setup) into `workspace` (allocated when None; returned).  A
    render_fwd(..., workspace=that, rays_ready=True) with the same cameras / shape / scene_range / row_window then launches
    the render kernel only - e.g. with the set-up of the next batch running on another stream meanwhile.  (Not for calls
    that ask for the ray_origins / ray_directions taps or the training stash: those write rays to their own tensors.)"""
    lib = _lib.load()
    cam2world = _f32c(cam2world, 'tform_cam2world')
    dev = cam2world.device
    B = cam2world.shape[0]
    n = B * height * width
    ws_bytes = lib.nfi_render_workspace_bytes(n)
    if workspace is None or workspace.numel() < ws_bytes:
        workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_render_setup', 'nfi_render_args', _stream(cam2world), n_scenes=B, height=height, width=width,
                         scene_range=float(scene_range), cam2world=cam2world, focal=_f32c(focal, 'focal_length'),
                         bbox=_f32c(bbox, 'bbox'), center=_f32c(center, 'center'), workspace=workspace,
                         workspace_bytes=workspace.numel(), row_offset=0 if row_window is None else int(row_window[0]),
                         full_height=0 if row_window is None else int(row_window[1]))
    return workspace


def render_fwd(cam2world, focal, height, width, num_samples, texels, decoder_image, scene_range, n_attention,
               attention_values=None, use_sdf=True, beta=None, alpha=None, bbox=None, center=None,
               noise_coarse=None, noise_fine=None, fine_sampling=True, white_background=True, taps=(),
               skip_missed_rays=True, workspace=None, events=None, tuning=0, profile_cycles=None, ray_features=None,
               termination_eps=0.0, row_window=None, clock_probe=None, stash=False, rays_ready=False,
               want_semantics=False, want_coords=False, want_normals=False, strict=False):
    """Fused forward render.  Returns dict(rgb [B,H,W,3], depth, mask [B,H,W], + requested taps).
    row_window: None, or (row_offset, full_height): `height` rows starting at row_offset of an image full_height rows tall
    (bit-identical to those rows of the full render; noise / outputs / taps are sized for the window).
    clock_probe: None or a uint64 / int64 [2] device tensor receiving {shader cycles, 100 MHz ticks} of the render kernel.
    stash: also return the training stash 'stash_t' [B,H,W,2S], 'stash_sigma' [B,H,W,2S], 'stash_rgb' [B,H,W,2S,3]
    (coarse samples in [..., :S], fine in [..., S:], source order; skipped rays hold zeros; without fine sampling the
    rows hold the S samples of the single pass) and 'ray_origins' /
    'ray_directions' - what composite_bwd(list_row_stride=2S) + field_query_bwd need for the backward of the render.
    ray_features: padded [B,H,W,48] per-ray view-direction features (decoder_pack_viewdir image).
    termination_eps: 0 = off; eps in (0,1): fine samples behind the depth at which the COARSE transmittance has fallen
    below eps are not evaluated and the rest is compacted by wave ballot (the coarse pass, the pdf and every sample index
    are untouched; |d rgb| <= ~eps, see include/nfi_hip.h).
    strict: True - raise when no ray of the batch meets the scene cube, as the reference does (lib/nerf_utils.py:258 fails
    on min() of an empty selection): the ray set-up runs as its own launch and its hit counter is read back before the
    render kernel is launched (one host synchronisation, on the set-up only - the reference synchronises at the same
    place, its boolean indexing at nerf_utils.py:258; the device idles for that host round trip, and `events` /
    `clock_probe` then bracket the render kernel WITHOUT the ray set-up); 'after' - one launch, the counter is read
    back behind the render kernel (no gap between set-up and render; the host waits for the whole render);
    'deferred' - no synchronisation: the counter is copied to pinned memory behind the render and checked by the next
    strict call / flush_strict().
    want_semantics / want_coords: also return 'semantics' [B,H,W,A] (composited softmax probabilities, run.py:312-335)
    / 'coords' [B,H,W,3] (composited query points, run.py:337-338) / 'normals' [B,H,W,3] (composited unit normals of the
    SDF, + 1 - mask on a white background, lib/nerf_utils.py:149-151, 159; any texel storage; with ray_features: fp32
    texels) from the SAME launch.
    Precision of 'semantics': with num_samples <= 64 the per-sample probabilities wait for the compositing in fp32; the
    64 < num_samples <= 128 kernel parks them as unorm16 (|error| <= 2^-17 = 7.7e-6 per sample, values below that become
    0), so its map is within 1e-5 of the reference's instead of 1e-6 (tests/test_hip_parity.py,
    test_wide_kernel_semantics_table_precision)."""
    cam2world = _f32c(cam2world, 'tform_cam2world')
    B = cam2world.shape[0]
    dev = cam2world.device
    n = B * height * width
    S = num_samples
    lib = _lib.load()
    tdt = texel_dtype_of(texels)
    out = {'rgb': torch.empty((B, height, width, 3), dtype=torch.float32, device=dev),
           'depth': torch.empty((B, height, width), dtype=torch.float32, device=dev),
           'mask': torch.empty((B, height, width), dtype=torch.float32, device=dev)}
    n2 = 2 * S if fine_sampling else S
    shapes = {'ray_origins': ((B, height, width, 3), torch.float32), 'ray_directions': ((B, height, width, 3), torch.float32),
              'near_plane': ((B, height, width), torch.float32), 'far_plane': ((B, height, width), torch.float32),
              'hit': ((B, height, width), torch.uint8),
              't_coarse': ((B, height, width, S), torch.float32), 'sigma_coarse': ((B, height, width, S), torch.float32),
              'rgb_coarse': ((B, height, width, S, 3), torch.float32),
              't_fine': ((B, height, width, S), torch.float32), 'sigma_fine': ((B, height, width, S), torch.float32),
              'rgb_fine': ((B, height, width, S, 3), torch.float32),
              't_sorted': ((B, height, width, n2), torch.float32), 'weights': ((B, height, width, n2), torch.float32),
              'perm': ((B, height, width, n2), torch.int32)}
    tap_t = {}
    for name in taps:
        if name not in shapes:
            raise KeyError('unknown tap %s' % name)
        if not fine_sampling and name in ('t_fine', 'sigma_fine', 'rgb_fine', 'perm'):
            continue
        shp, dt = shapes[name]
        tap_t[name] = torch.zeros(shp, dtype=dt, device=dev)
    if stash:
        for name in ('ray_origins', 'ray_directions'):
            tap_t.setdefault(name, torch.empty(shapes[name][0], dtype=torch.float32, device=dev))
        tap_t['stash_t'] = torch.empty((B, height, width, n2), dtype=torch.float32, device=dev)
        tap_t['stash_sigma'] = torch.empty((B, height, width, n2), dtype=torch.float32, device=dev)
        tap_t['stash_rgb'] = torch.empty((B, height, width, n2, 3), dtype=torch.float32, device=dev)
    if want_semantics:
        if n_attention <= 0:
            raise ValueError('render_fwd: semantics need attention values (n_attention > 0)')
        tap_t['semantics'] = torch.empty((B, height, width, n_attention), dtype=torch.float32, device=dev)
    if want_coords:
        tap_t['coords'] = torch.empty((B, height, width, 3), dtype=torch.float32, device=dev)
    if want_normals:
        if not use_sdf:
            raise ValueError('render_fwd: the normals map needs the SDF decoder (use_sdf)')
        tap_t['normals'] = torch.empty((B, height, width, 3), dtype=torch.float32, device=dev)
    ws_bytes = lib.nfi_render_workspace_bytes(n)
    check_after = False
    if strict == 'deferred':
        _raise_pending(dev)                      # an EARLIER batch of this device that met no ray
    elif strict == 'after':
        check_after = True
    elif strict:
        if not rays_ready and not stash and not any(name in tap_t for name in ('ray_origins', 'ray_directions', 'hit')):
            # the ray set-up as its own launch, its hit count read back BEFORE the render kernel goes out: the host waits
            # for the set-up only, and a batch without a hit never pays for a render
            workspace = render_setup(cam2world, focal, height, width, scene_range, bbox=bbox, center=center,
                                     workspace=workspace, row_window=row_window)
            rays_ready = True
        if rays_ready and workspace is not None and workspace.numel() >= ws_bytes:
            _raise_if_no_hit(workspace)
        else:
            check_after = True                   # (stash / ray taps: the set-up writes rays to the caller's tensors)
    if rays_ready:
        # the kernel would march whatever the workspace holds: refuse anything that is not nfi_render_setup's own result
        if workspace is None or workspace.numel() < ws_bytes:
            raise ValueError('render_fwd(rays_ready=True) needs the workspace render_setup filled (%d bytes)' % ws_bytes)
        if stash or any(name in tap_t for name in ('ray_origins', 'ray_directions', 'hit')):
            raise ValueError('render_fwd(rays_ready=True) cannot be combined with the stash or the ray_origins / '
                             'ray_directions / hit taps (they redirect the ray set-up away from the workspace)')
    if workspace is None or workspace.numel() < ws_bytes:
        workspace = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    if fine_sampling:
        if noise_fine is None:
            noise_fine = torch.linspace(0.0, 1.0, steps=S, dtype=torch.float32, device=dev).expand(n, S)
        u, ustride = _u_and_stride(noise_fine if noise_fine.dim() == 2 else noise_fine.reshape(-1, S), n, S)
    else:
        u, ustride = None, 0
    if ray_features is not None:
        ray_features = _f32c(ray_features, 'ray_features')
        if ray_features.numel() != n * RAY_FEATURE_PITCH:
            raise ValueError('ray_features must be [B,H,W,48], got %s' % (tuple(ray_features.shape),))
    with torch.cuda.device(dev):
        _lib.call_struct(
            'nfi_render_fwd', 'nfi_render_args', _stream(cam2world), n_scenes=B, height=height, width=width,
            n_samples=S, fine_sampling=int(fine_sampling), white_background=int(white_background),
            scene_range=float(scene_range), cam2world=cam2world, focal=_f32c(focal, 'focal_length'),
            bbox=_f32c(bbox, 'bbox'), center=_f32c(center, 'center'), texels=texels, plane_res=texel_res(texels),
            texel_layout=texel_layout_of(texels), texel_dtype=tdt, decoder_image=decoder_image, n_attention=n_attention,
            attention_values=_f32c(attention_values, 'attention_values') if n_attention > 0 else None,
            use_sdf=int(use_sdf), beta=_f32c(beta, 'beta') if use_sdf else None,
            alpha=_f32c(alpha, 'alpha') if use_sdf else None, noise_coarse=_f32c(noise_coarse, 'noise_coarse'),
            noise_fine=u, noise_fine_row_stride=ustride, rgb=out['rgb'], depth=out['depth'], mask=out['mask'],
            workspace=workspace, workspace_bytes=workspace.numel(), skip_missed_rays=int(skip_missed_rays),
            event_start=None if events is None else events[0], event_stop=None if events is None else events[1],
            tuning=int(tuning), profile_cycles=profile_cycles, ray_features=ray_features,
            termination_eps=float(termination_eps), clock_probe=clock_probe,
            row_offset=0 if row_window is None else int(row_window[0]),
            full_height=0 if row_window is None else int(row_window[1]), rays_ready=int(bool(rays_ready)), **tap_t)
    out.update(tap_t)
    out['_workspace'] = workspace
    if strict == 'deferred':
        _defer_hit_count(workspace, dev)
    elif check_after:
        _raise_if_no_hit(workspace)
    return out


_NO_HIT = ('compute_near_far_planes: no ray intersects the scene cube (the reference fails on min() of an empty '
           'selection here)')
_PENDING = {}          # device index -> dict(host=pinned int32 ring, slot, queue=[(event, slot)]); one thread per device


def _raise_if_no_hit(workspace):
    if int(workspace[8:12].view(torch.int32).item()) == 0:      # the hit count of the ray set-up (csrc: reduce[2])
        raise RuntimeError(_NO_HIT)


def _dev_key(dev):
    """The device's index; torch.device('cuda') (index None) is the current device - the key the tensors' own device has."""
    return torch.cuda.current_device() if dev.index is None else dev.index


def _defer_hit_count(source, dev):
    """strict='deferred': the hit count goes to pinned host memory behind the render (async copy + event), no wait.
    source: the render workspace (uint8; the count is its third int32) or the int32 [1] count itself."""
    st = _PENDING.setdefault(_dev_key(dev), {'host': torch.zeros(64, dtype=torch.int32).pin_memory(), 'slot': 0, 'queue': []})
    if len(st['queue']) >= 64:
        _raise_pending(dev, wait=True)
    slot = st['slot']
    st['slot'] = (slot + 1) % 64
    count = source[8:12].view(torch.int32) if source.dtype == torch.uint8 else source
    st['host'][slot:slot + 1].copy_(count, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    st['queue'].append((ev, slot))


def _raise_pending(dev, wait=False):
    st = _PENDING.get(_dev_key(dev))
    while st and st['queue']:
        ev, slot = st['queue'][0]
        if not ev.query():
            if not wait:
                return
            ev.synchronize()
        st['queue'].pop(0)
        if int(st['host'][slot]) == 0:
            st['queue'].clear()
            raise RuntimeError(_NO_HIT + ' [an earlier batch, strict_near_far="deferred"]')


def flush_strict(device=None):
    """strict='deferred': waits for the hit counts still in flight on `device` (default: the current one) and raises if
    one of those batches met no ray.  Call it where a loop synchronises anyway (end of an epoch, before a report)."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    _raise_pending(dev, wait=True)


# --------------------------------------------------------------------------- #
# regulariser branch: distance + spatial gradient as one operator
# --------------------------------------------------------------------------- #
def sdf_gradient_fwd(points, texels, w1, b1, w2, b2, scene_range):
    """points [B,P,3] (inside the cube) -> (sdf [B,P], d sdf / d points [B,P,3])."""
    points = _f32c(points, 'points')
    if texels.dtype != torch.float32:
        raise TypeError('sdf_gradient: fp32 texels only')
    B, P = points.shape[0], points.shape[1]
    sdf = torch.empty((B, P), dtype=torch.float32, device=points.device)
    grad = torch.empty((B, P, 3), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.call_struct('nfi_sdf_gradient_fwd', 'nfi_sdf_gradient_args', _stream(points), n_scenes=B, points_per_scene=P,
                         points=points, texels=texels, plane_res=texel_res(texels), texel_layout=texel_layout_of(texels),
                         scene_range=float(scene_range),
                         w1=_f32c(w1, 'w1'), b1=_f32c(b1, 'b1'), w2=_f32c(w2, 'w2'), b2=_f32c(b2, 'b2'), sdf=sdf,
                         gradient=grad)
    return sdf, grad


def sdf_gradient_bwd(points, texels, w1, b1, w2, b2, scene_range, g_sdf, g_gradient):
    """Backward (= the reference's double backward) of sdf_gradient_fwd.  Returns dict(g_texels, g_w1, g_b1, g_w2
    [n_out,64] with row 0 filled, g_b2 [n_out] with entry 0 filled)."""
    points = _f32c(points, 'points')
    B, P = points.shape[0], points.shape[1]
    dev = points.device
    w2, b2 = _f32c(w2, 'w2'), _f32c(b2, 'b2')
    out = {'g_texels': torch.zeros_like(texels), 'g_w1': torch.zeros((64, 32), dtype=torch.float32, device=dev),
           'g_b1': torch.zeros((64,), dtype=torch.float32, device=dev), 'g_w2': torch.zeros_like(w2),
           'g_b2': torch.zeros_like(b2)}
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_sdf_gradient_bwd', 'nfi_sdf_gradient_args', _stream(points), n_scenes=B, points_per_scene=P,
                         points=points, texels=texels, plane_res=texel_res(texels), texel_layout=texel_layout_of(texels),
                         scene_range=float(scene_range),
                         w1=_f32c(w1, 'w1'), b1=_f32c(b1, 'b1'), w2=w2, b2=b2, g_sdf=_f32c(g_sdf, 'g_sdf'),
                         g_gradient=_f32c(g_gradient, 'g_gradient'), **out)
    return out


# --------------------------------------------------------------------------- #
# backward wrappers
# --------------------------------------------------------------------------- #
def composite_bwd_stash(ray_directions, stash_t, stash_sigma, stash_rgb, g_rgb_map, g_mask=None, white_background=True,
                        want_rd=True, fine=True):
    """Compositing backward on the training stash of render_fwd (rows of 2S entries: coarse | fine; fine=False: rows of
    the S samples of a single pass).  Returns dict(g_sigma [..., 2S], g_rgb [..., 2S, 3], g_ray_directions?) in the
    stash's layout."""
    rd = _f32c(ray_directions, 'ray_directions')
    t, sg, col = _f32c(stash_t, 'stash_t'), _f32c(stash_sigma, 'stash_sigma'), _f32c(stash_rgb, 'stash_rgb')
    S2 = t.shape[-1]
    n = t.numel() // S2
    if not fine:
        g = composite_bwd(rd, t, sg, col, g_rgb_map, g_mask, white_background=white_background, want_rd=want_rd)
        out = dict(g_sigma=g['g_sigma_a'], g_rgb=g['g_rgb_a'])
        if want_rd:
            out['g_ray_directions'] = g['g_ray_directions']
        return out
    S = S2 // 2
    out = dict(g_sigma=torch.empty_like(sg), g_rgb=torch.empty_like(col))
    if want_rd:
        out['g_ray_directions'] = torch.empty_like(rd)
    f4 = 4              # bytes per float: list b starts S entries into every row
    with torch.cuda.device(rd.device):
        _lib.call_struct('nfi_composite_bwd', 'nfi_composite_bwd_args', _stream(rd), n_rays=n, n_a=S, n_b=S,
                         list_row_stride=S2, ray_directions=rd, depth_a=t, sigma_a=sg, rgb_a=col,
                         depth_b=t.data_ptr() + S * f4, sigma_b=sg.data_ptr() + S * f4, rgb_b=col.data_ptr() + 3 * S * f4,
                         n_extra=0, white_background=int(white_background), g_rgb_map=_f32c(g_rgb_map, 'g_rgb_map'),
                         g_mask=_f32c(g_mask, 'g_mask'), g_sigma_a=out['g_sigma'], g_rgb_a=out['g_rgb'],
                         g_sigma_b=out['g_sigma'].data_ptr() + S * f4, g_rgb_b=out['g_rgb'].data_ptr() + 3 * S * f4,
                         g_ray_directions=out.get('g_ray_directions'))
    return out


def composite_bwd(ray_directions, depth_a, sigma_a, rgb_a, g_rgb_map, g_mask=None, depth_b=None, sigma_b=None,
                  rgb_b=None, extra_a=None, extra_b=None, g_extra_map=None, white_background=True, want_rd=True):
    rd = _f32c(ray_directions, 'ray_directions')
    depth_a, sigma_a, rgb_a = _f32c(depth_a, 'depth'), _f32c(sigma_a, 'sigma'), _f32c(rgb_a, 'rgb')
    na = depth_a.shape[-1]
    n = depth_a.numel() // na
    nb = 0
    if depth_b is not None:
        depth_b, sigma_b, rgb_b = _f32c(depth_b, 'depth_b'), _f32c(sigma_b, 'sigma_b'), _f32c(rgb_b, 'rgb_b')
        nb = depth_b.shape[-1]
    n_extra = 0
    if extra_a is not None and g_extra_map is not None:
        extra_a, extra_b = _f32c(extra_a, 'extra_a'), _f32c(extra_b, 'extra_b')
        n_extra = extra_a.shape[-1]
    out = dict(g_sigma_a=torch.empty_like(sigma_a), g_rgb_a=torch.empty_like(rgb_a))
    if nb:
        out.update(g_sigma_b=torch.empty_like(sigma_b), g_rgb_b=torch.empty_like(rgb_b))
    if n_extra:
        out['g_extra_a'] = torch.empty_like(extra_a)
        if nb:
            out['g_extra_b'] = torch.empty_like(extra_b)
    if want_rd:
        out['g_ray_directions'] = torch.empty_like(rd)
    with torch.cuda.device(rd.device):
        _lib.call_struct('nfi_composite_bwd', 'nfi_composite_bwd_args', _stream(rd), n_rays=n, n_a=na, n_b=nb,
                         ray_directions=rd, depth_a=depth_a, sigma_a=sigma_a, rgb_a=rgb_a, depth_b=depth_b,
                         sigma_b=sigma_b, rgb_b=rgb_b, n_extra=n_extra, extra_a=extra_a if n_extra else None,
                         extra_b=extra_b if n_extra else None, white_background=int(white_background),
                         g_rgb_map=_f32c(g_rgb_map, 'g_rgb_map'), g_mask=_f32c(g_mask, 'g_mask'),
                         g_extra_map=_f32c(g_extra_map, 'g_extra_map') if n_extra else None, **out)
    return out


def points_bwd(g_points, depth, want_ro=True, want_rd=True):
    g_points, depth = _f32c(g_points, 'g_points'), _f32c(depth, 'depth')
    S = depth.shape[-1]
    n = depth.numel() // S
    shape = depth.shape[:-1]
    g_ro = torch.empty((*shape, 3), dtype=torch.float32, device=depth.device) if want_ro else None
    g_rd = torch.empty((*shape, 3), dtype=torch.float32, device=depth.device) if want_rd else None
    lib = _lib.load()
    with torch.cuda.device(depth.device):
        _lib.check(lib.nfi_points_bwd(_lib.ptr(g_points), _lib.ptr(depth), n, S, _lib.ptr(g_ro), _lib.ptr(g_rd),
                                      _stream(depth)), 'nfi_points_bwd')
    return g_ro, g_rd


def raygen_bwd(height, width, focal, cam2world, bbox, center, normalize, g_ro, g_rd):
    cam2world = _f32c(cam2world, 'tform_cam2world')
    B = cam2world.shape[0]
    focal, bbox, center = _f32c(focal, 'focal_length'), _f32c(bbox, 'bbox'), _f32c(center, 'center')
    g_cam = torch.empty((B, 4, 4), dtype=torch.float32, device=cam2world.device)
    g_focal = torch.empty((B,), dtype=torch.float32, device=cam2world.device) if focal is not None else None
    lib = _lib.load()
    a = _lib.make_args('nfi_raygen_args', n_scenes=B, height=height, width=width, cam2world=cam2world, focal=focal,
                       bbox=bbox, center=center, normalize=int(normalize))
    import ctypes
    with torch.cuda.device(cam2world.device):
        _lib.check(lib.nfi_raygen_bwd(ctypes.byref(a), _lib.ptr(_f32c(g_ro, 'g_ro')), _lib.ptr(_f32c(g_rd, 'g_rd')),
                                      _lib.ptr(g_cam), _lib.ptr(g_focal), _stream(cam2world)), 'nfi_raygen_bwd')
    return g_cam, g_focal


# --------------------------------------------------------------------------- #
# neighbours of the renderer in the inversion loop
# --------------------------------------------------------------------------- #
def _warp_args(img_shape, rot, scale, translation, white_background):
    n, c, h, w = img_shape
    rot, translation = _f32c(rot, 'rot'), _f32c(translation, 'translation')
    scale = _f32c(scale, 'scale')
    if rot.numel() != n or translation.numel() != 2 * n or (scale is not None and scale.numel() != n):
        raise ValueError('affine_warp: rot [N], scale [N] or None, translation [N,2] expected for N=%d images' % n)
    return dict(n_images=n, channels=c, height=h, width=w, rot=rot, scale=scale, translation=translation,
                white_background=int(bool(white_background)))


def affine_warp(img, rot, scale, translation, white_background=False):
    """img [N,C,H,W] warped by the per-image rotation / scale / translation of augment_impl (run.py:720-769)."""
    img = _f32c(img, 'img')
    if img.dim() != 4:
        raise ValueError('affine_warp: img must be [N,C,H,W]')
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _lib.call_struct('nfi_affine_warp_fwd', 'nfi_warp_args', _stream(img), image=img, warped=out,
                         **_warp_args(img.shape, rot, scale, translation, white_background))
    return out


def affine_warp_bwd(g_out, rot, scale, translation, white_background=False):
    """Adjoint of affine_warp w.r.t. the image: g_out [N,C,H,W] -> g_img [N,C,H,W]."""
    g_out = _f32c(g_out, 'g_out')
    g_img = torch.empty_like(g_out)
    with torch.cuda.device(g_out.device):
        _lib.call_struct('nfi_affine_warp_bwd', 'nfi_warp_args', _stream(g_out), g_warped=g_out, g_image=g_img,
                         **_warp_args(g_out.shape, rot, scale, translation, white_background))
    return g_img


def image_metrics(pred=None, target=None, mask_pred=None, mask_real=None, check_range=True):
    """Per-image PSNR (lib/metrics.py:30-45) of pred/target [B,...] in [0,1] and / or mask IoU (79-94) of
    mask_pred/mask_real [B,...].  Returns (psnr [B] or None, iou [B] or None, out_of_range int32 [1] or None)."""
    first = pred if pred is not None else mask_pred
    if first is None:
        raise ValueError('image_metrics: nothing to measure')
    b, dev = first.shape[0], first.device
    kw = dict(n_images=b)
    psnr = iou = flag = None
    if pred is not None:
        pred, target = _f32c(pred, 'pred'), _f32c(target, 'target')
        if pred.shape != target.shape:
            raise ValueError('image_metrics: pred %s vs target %s' % (tuple(pred.shape), tuple(target.shape)))
        psnr = torch.empty((b,), dtype=torch.float32, device=dev)
        kw.update(pred=pred, target=target, elements_per_image=pred.numel() // b, psnr=psnr)
    if mask_pred is not None:
        mask_pred, mask_real = _f32c(mask_pred, 'mask_pred'), _f32c(mask_real, 'mask_real')
        if mask_pred.shape != mask_real.shape or mask_pred.shape[0] != b:
            raise ValueError('image_metrics: mask shapes %s vs %s' % (tuple(mask_pred.shape), tuple(mask_real.shape)))
        iou = torch.empty((b,), dtype=torch.float32, device=dev)
        kw.update(mask_pred=mask_pred, mask_real=mask_real, elements_per_mask=mask_pred.numel() // b, iou=iou)
    if check_range:
        flag = torch.empty((1,), dtype=torch.int32, device=dev)
        kw['out_of_range'] = flag
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_image_metrics', 'nfi_metrics_args', _stream(first), **kw)
    return psnr, iou, flag


# --------------------------------------------------------------------------- #
# plane-producer hand-off
# --------------------------------------------------------------------------- #
def torgb_texels(x, styles, weight, bias, previous_image=None):
    """Tail of the last synthesis block (stylegan.py:383-435) written as texels: x [B,Cin,R,R], styles [B,Cin],
    weight [96,Cin], bias [96], previous_image [B,96,R/2,R/2] or None -> a [B,96,R,R] tensor in channels-last memory
    format (its storage IS the interleaved texel image [B,R,R,3,32])."""
    x, styles, weight, bias = _f32c(x, 'x'), _f32c(styles, 'styles'), _f32c(weight, 'weight'), _f32c(bias, 'bias')
    prev = _f32c(previous_image, 'previous_image')
    B, Cin, R, R2 = x.shape
    if R != R2 or tuple(styles.shape) != (B, Cin) or tuple(weight.shape) != (96, Cin) or tuple(bias.shape) != (96,) or \
            (prev is not None and tuple(prev.shape) != (B, 96, R // 2, R // 2)):
        raise ValueError('torgb_texels: shapes x %s styles %s weight %s bias %s prev %s' % (
            tuple(x.shape), tuple(styles.shape), tuple(weight.shape), tuple(bias.shape),
            None if prev is None else tuple(prev.shape)))
    out = torch.empty((B, 96, R, R), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _lib.call_struct('nfi_torgb_texels_fwd', 'nfi_torgb_args', _stream(x), n_scenes=B, in_channels=Cin, resolution=R,
                         x=x, styles=styles, weight=weight, bias=bias, previous_image=prev, texels=out)
    return out


def torgb_texels_bwd(g_out, x, styles, weight, previous_image=None, want_weight=True, want_prev=True):
    """Backward of torgb_texels.  g_out: [B,96,R,R] (any strides; brought to channels-last).  Returns dict(g_x, g_styles,
    g_weight?, g_bias?, g_previous_image?)."""
    x, styles, weight = _f32c(x, 'x'), _f32c(styles, 'styles'), _f32c(weight, 'weight')
    prev = _f32c(previous_image, 'previous_image')
    B, Cin, R, _ = x.shape
    if not g_out.is_cuda or g_out.dtype != torch.float32:
        raise TypeError('torgb_texels_bwd: g_out must be a float32 GPU tensor')
    g = g_out.contiguous(memory_format=torch.channels_last)
    dev = x.device
    out = {'g_x': torch.empty_like(x), 'g_styles': torch.empty((B, Cin), dtype=torch.float32, device=dev)}
    if want_weight:
        out['g_weight'] = torch.empty((96, Cin), dtype=torch.float32, device=dev)
        out['g_bias'] = torch.empty((96,), dtype=torch.float32, device=dev)
    if want_prev and prev is not None:
        out['g_previous_image'] = torch.empty_like(prev)
    with torch.cuda.device(dev):
        _lib.call_struct('nfi_torgb_texels_bwd', 'nfi_torgb_args', _stream(x), n_scenes=B, in_channels=Cin, resolution=R,
                         x=x, styles=styles, weight=weight, previous_image=prev, g_texels=g, **out)
    return out
