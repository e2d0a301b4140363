/*
 * nfi_hip.h — C ABI of libnfi_hip.so, the MI355X (gfx950) implementation of the
 * nerf-from-image volumetric-rendering hot path.
 *
 * The reference (google-research/nerf-from-image) has no FFI: the boundary of this
 * path is a set of Python callables.  Each entry point below names the reference
 * callable it replaces (file:line relative to the reference checkout); the Python
 * binding in nerf_from_image_amd/ calls these through ctypes and re-creates the
 * reference's call surface on top (lib/nerf_utils.py functions, the Generator
 * `sampler` closure, run.py::render).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a CALLER-OWNED DEVICE pointer (HIP, same device as the
 *     stream) unless the field says "host"; nothing is allocated, nothing is
 *     synchronised, every launch goes to the caller's stream;
 *   - all entry points are re-entrant (no global mutable state): the reference
 *     calls this path concurrently from one thread per GPU (nn.DataParallel,
 *     run.py:636-640);
 *   - return value: 0 on success, a negative nfi_status otherwise;
 *     nfi_last_error() returns a thread-local message for the last failure;
 *   - tensors are dense row-major fp32 unless stated; "N" is the number of rays
 *     B*H*W, "S" the samples per ray and pass.
 */
#ifndef NFI_HIP_H
#define NFI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nfi_stream_t; /* hipStream_t */

typedef enum nfi_status {
  NFI_OK = 0,
  NFI_ERR_INVALID_ARGUMENT = -1,
  NFI_ERR_UNSUPPORTED = -2,
  NFI_ERR_WORKSPACE_TOO_SMALL = -3,
  NFI_ERR_LAUNCH = -4
} nfi_status;

enum { NFI_PLANE_CHANNELS = 32, NFI_HIDDEN = 64, NFI_MAX_ATTENTION = 14, NFI_MAX_SAMPLES = 128 };
/* NFI_MAX_SAMPLES: samples per ray and PASS of the two-pass (coarse + fine) pipeline and of the fused kernel.  A single
 * pass without fine sampling (run.py:512-514: 128 samples; the inversion loop asks for ray_multiplier = 4, run.py:2271)
 * can hold up to NFI_MAX_SAMPLES_SINGLE_PASS samples in nfi_ray_weights / nfi_composite_fwd / nfi_composite_bwd. */
enum { NFI_MAX_SAMPLES_SINGLE_PASS = 512 };

/* texel storage type of the channel-last plane image */
enum { NFI_TEXEL_F32 = 0, NFI_TEXEL_BF16 = 1, NFI_TEXEL_F16 = 2 };
/* texel layout (`texel_layout` fields; 0 = default).  A texel = the 32 channels of one plane at one (y,x):
 *   PLANAR       [B,3,R,R,32]   written by nfi_planes_to_texels from the reference's NCHW planes;
 *   INTERLEAVED  [B,R,R,3,32]   = a channels-last (NHWC) [B,96,R,R] image, i.e. what a producer that emits
 *                               torch.channels_last - or nfi_torgb_texels_fwd - leaves in memory: read in place,
 *                               no hand-off kernel; gradient images use the layout of their texels. */
enum { NFI_TEXELS_PLANAR = 0, NFI_TEXELS_INTERLEAVED = 1 };

const char* nfi_last_error(void);
int nfi_version(void);

/* ------------------------------------------------------------------------------------------
 * Triplane hand-off: NCHW planes from the synthesis network -> channel-last texels.
 * Replaces the implicit layout the reference gathers from (models/generator.py:475-477,
 * 501-503: planes.view(B,3,32,R,R); F.grid_sample reads NCHW).  A texel (32 channels) becomes
 * one 128-byte line (fp32) / 64-byte line (bf16).
 *   planes  [B,3,32,R,R] fp32      texels  [B,3,R,R,32] fp32|bf16
 * ------------------------------------------------------------------------------------------ */
int nfi_planes_to_texels(const float* planes, void* texels, int n_scenes, int plane_res,
                         int texel_dtype, nfi_stream_t stream);
/* adjoint of the above (fp32 only): texel-layout gradient -> NCHW gradient */
int nfi_texels_to_planes(const float* texels, float* planes, int n_scenes, int plane_res,
                         nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Decoder operand image.  Replaces the per-call weight scaling of EqualizedLinear
 * (models/stylegan.py:173-180: W * 1/sqrt(in), b * 1) for TriplanarDecoder.net
 * (models/generator.py:295-299): folds the gains, the /3 of the plane mean
 * (generator.py:328) and the exp2/log2 change of base into a lane-ordered MFMA operand image.
 *   w1 [64,32]  b1 [64]  w2 [n_out,64]  b2 [n_out]   (raw parameters, n_out = 1+A, or 4 if A==0)
 *   image: nfi_decoder_image_floats() floats
 * ------------------------------------------------------------------------------------------ */
size_t nfi_decoder_image_floats(void);
int nfi_decoder_pack(const float* w1, const float* b1, const float* w2, const float* b2,
                     int n_attention, int texel_dtype, float* image, nfi_stream_t stream);

/* View-direction variant (--use_viewdir; ViewDirectionMapper, models/generator.py:189-253; decoder output
 * widened to 32 features, 376-377; closure applied per sample, 243-251 / 662-663):
 *   w2 [33,64] b2 [33] (row 0 = distance, rows 1..32 = features);  w3 [n3,32] b3 [n3] = the mapper's
 *   `output` EqualizedLinear (raw parameters), n3 = A, or 3 if A == 0.
 * The per-ray output of ViewDirectionMapper.fc6 is handed to the field / render entry points as
 * ray_features [rays, NFI_RAY_FEATURE_PITCH] = [0, x_0 .. x_31, 0 x 15] (padded by the caller). */
enum { NFI_RAY_FEATURE_PITCH = 48 };
size_t nfi_decoder_image_floats_viewdir(void);
int nfi_decoder_pack_viewdir(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                             const float* b3, int n_attention, int texel_dtype, float* image, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Camera rays + scene-box intersection.
 * Replaces nerf_utils.get_ray_bundle (lib/nerf_utils.py:28-91), F.normalize (run.py:196)
 * and nerf_utils.compute_near_far_planes (lib/nerf_utils.py:225-273).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_raygen_args {
  int n_scenes, height, width;
  const float* cam2world;   /* [B,4,4] */
  const float* focal;       /* [B] or NULL -> orthographic model (nerf_utils.py:66-89) */
  const float* bbox;        /* [B,2,2] or NULL */
  const float* center;      /* [B,2] or NULL (perspective only) */
  int normalize;            /* 1: ray directions L2-normalised (run.py:196) */
  float* ray_origins;       /* [N,3] out */
  float* ray_directions;    /* [N,3] out */
} nfi_raygen_args;
int nfi_raygen(const nfi_raygen_args* a, nfi_stream_t stream);

typedef struct nfi_near_far_args {
  int64_t n_rays;
  const float* ray_origins;    /* [N,3] */
  const float* ray_directions; /* [N,3] */
  float scene_range;
  float* near_raw;  /* [N] out: slab entry distance before the miss-fill / clamps */
  float* far_raw;   /* [N] out */
  uint8_t* hit;     /* [N] out: 1 if the ray's line meets the cube */
  uint32_t* reduce; /* [4] out: order-preserving keys of min(near|hit), max(far|hit), hit count, 0 */
  /* optional: finished planes (miss-fill with the batch-wide min/max, clamp >= 0.1,
   * far-near >= 1e-3; nerf_utils.py:258-268).  NULL to skip. */
  float* near_plane; /* [N] out or NULL */
  float* far_plane;  /* [N] out or NULL */
} nfi_near_far_args;
int nfi_near_far(const nfi_near_far_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Stratified depths + query points.  Replaces nerf_utils.compute_query_points_from_rays
 * (lib/nerf_utils.py:94-120).  noise: the reference's torch.rand_like draw, or NULL.
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_stratified_args {
  int64_t n_rays;
  int n_samples;
  const float* ray_origins;    /* [N,3] */
  const float* ray_directions; /* [N,3] */
  const float* near_plane;     /* [N] */
  const float* far_plane;      /* [N] */
  const float* noise;          /* [N,S] in [0,1) or NULL */
  float* depth;                /* [N,S] out */
  float* points;               /* [N,S,3] out */
} nfi_stratified_args;
int nfi_stratified_points(const nfi_stratified_args* a, nfi_stream_t stream);
/* x = o + d*t for given depths (the fine query points, run.py:286-288).
 * ray_origins/ray_directions [N,3], depth [N,S] -> points [N,S,3] */
int nfi_points_on_rays(const float* ray_origins, const float* ray_directions, const float* depth,
                       int64_t n_rays, int n_samples, float* points, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Field query: the `sampler` closure of Generator.forward (models/generator.py:587-681) with
 * TriplanarDecoder.forward (301-331), laplace_cdf (30-33) and the colour head (661-679).
 *   points [B,P,3] world coordinates -> sigma [B,P], rgb [B,P,3], optional sdf [B,P]
 *   (raw decoder output 0), semantics [B,P,A] (softmax probabilities), outside [B,P] (u8).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_field_args {
  int n_scenes;
  int64_t points_per_scene;
  const float* points;           /* [B,P,3] */
  const void* texels;            /* [B,3,R,R,32] */
  int plane_res;
  int texel_dtype;
  const float* decoder_image;    /* from nfi_decoder_pack (same texel_dtype, same n_attention) */
  int n_attention;               /* A; 0 -> rgb = wide sigmoid of 3 features */
  const float* attention_values; /* [B,A,3] (A>0) */
  int use_sdf;                   /* 1: sigma = laplace_cdf(-d,beta)/alpha; 0: softplus(d-1) */
  const float* beta;             /* device scalar (use_sdf) */
  const float* alpha;            /* device scalar (use_sdf) */
  float scene_range;
  float* sigma;                  /* [B,P] out */
  float* rgb;                    /* [B,P,3] out */
  float* sdf;                    /* [B,P] out or NULL */
  float* semantics;              /* [B,P,A] out or NULL */
  uint8_t* outside;              /* [B,P] out or NULL */
  /* view-direction decoder: NULL, or [B, P/samples_per_ray, NFI_RAY_FEATURE_PITCH]; decoder_image must then
   * come from nfi_decoder_pack_viewdir and point p belongs to ray p / samples_per_ray (x_in [B,H,W,S,3]) */
  const float* ray_features;
  int samples_per_ray;
  int texel_layout;              /* NFI_TEXELS_PLANAR (0) / NFI_TEXELS_INTERLEAVED */
  int mlp_precision;             /* 0: exact fp32 MFMA (default); 1: split-fp16 operands with fp32 accumulation, the fused
                                  * renderer's decoder arithmetic (sigma within 3e-5 relative of mode 0) */
} nfi_field_args;
int nfi_field_query_fwd(const nfi_field_args* a, nfi_stream_t stream);

/* 'bbox' visualisation overlay of the sampler closure (models/generator.py:645-659, only with 'coords' in the
 * sampler request and 'bbox' in the model request): sigma_out = sigma_in + 100 for points inside the scene cube
 * that lie within 5e-2 of at least one face pair on every axis pair (the wire frame of the cube).
 *   points [n,3], sigma_in / sigma_out [n] (may alias); threshold = (float)(scene_range - 5e-2) from the caller. */
int nfi_bbox_overlay(const float* points, int64_t n_points, float scene_range, float threshold, const float* sigma_in,
                     float* sigma_out, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray weights.  Replaces nerf_utils.render_volume_density_weights_only
 * (lib/nerf_utils.py:164-180, cumprod_exclusive 20-25).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_weights_args {
  int64_t n_rays;
  int n_samples;               /* <= NFI_MAX_SAMPLES_SINGLE_PASS */
  const float* sigma;          /* [N,S] */
  const float* ray_directions; /* [N,3] */
  const float* depth;          /* [N,S] */
  float* weights;              /* [N,S] out */
} nfi_weights_args;
int nfi_ray_weights(const nfi_weights_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Inverse-CDF sampling.  Replaces nerf_utils.sample_pdf (lib/nerf_utils.py:183-222).
 *   bins [N,M], weights [N,M-1], u [N,K] (row stride u_row_stride floats; 0 broadcasts one
 *   row, which is how the deterministic linspace(0,1,K) of the reference is passed)
 *   -> samples [N,K]; optional inds int64 [N,K] (searchsorted(cdf,u,right=True)), cdf [N,M].
 *   M <= 128, K <= 128.
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_sample_pdf_args {
  int64_t n_rays;
  int n_bins;      /* M */
  int n_samples;   /* K */
  const float* bins;
  const float* weights;
  const float* u;
  int64_t u_row_stride;
  float* samples;
  int64_t* inds;   /* or NULL */
  float* cdf;      /* or NULL */
} nfi_sample_pdf_args;
int nfi_sample_pdf(const nfi_sample_pdf_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hierarchical resampling as run.py does it between the two field queries: coarse weights
 * (run.py:262) -> EG3D smoothing (264-272) -> sample_pdf on the bin mid-points (274-281).
 *   sigma,depth [N,S]; u as above with K = S  ->  fine depths [N,S] (unsorted, like the reference)
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_resample_args {
  int64_t n_rays;
  int n_samples;               /* S <= 64 */
  const float* sigma;
  const float* ray_directions;
  const float* depth;
  const float* u;
  int64_t u_row_stride;
  float* fine_depth;           /* [N,S] out */
  float* weights;              /* [N,S] out or NULL (coarse weights) */
  float* smooth;               /* [N,S] out or NULL */
  float* cdf;                  /* [N,S-1] out or NULL */
  int64_t* inds;               /* [N,S] out or NULL */
} nfi_resample_args;
int nfi_resample(const nfi_resample_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Alpha compositing.  Replaces nerf_utils.render_volume_density (lib/nerf_utils.py:123-161)
 * and, when depth_b != NULL, the sort/merge of run.py:283-335 in front of it: the two depth
 * lists are merged ascending (stable: list a first on ties) and attributes follow.
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_composite_args {
  int64_t n_rays;
  int n_a;                      /* samples in list a (n_b == 0: <= 512; merged lists: <= 128 each) */
  int n_b;                      /* 0: plain compositing of list a, assumed in ray order */
  const float* ray_directions;  /* [N,3] */
  const float* depth_a; const float* sigma_a; const float* rgb_a;  /* [N,n_a], [N,n_a], [N,n_a,3] */
  const float* depth_b; const float* sigma_b; const float* rgb_b;  /* [N,n_b] ... or NULL */
  int n_extra;                  /* channels of an extra attribute (semantics / normals / coords), 0 = none */
  const float* extra_a;         /* [N,n_a,n_extra] */
  const float* extra_b;         /* [N,n_b,n_extra] */
  int white_background;
  float* rgb_map;    /* [N,3] out */
  float* depth_map;  /* [N] out */
  float* mask;       /* [N] out */
  float* extra_map;  /* [N,n_extra] out or NULL */
  float* weights;    /* [N,n_a+n_b] out or NULL (in merged order) */
  float* depth_sorted; /* [N,n_a+n_b] out or NULL */
  int64_t* perm;     /* [N,n_a+n_b] out or NULL: merged position -> index into cat(a,b) */
} nfi_composite_args;
int nfi_composite_fwd(const nfi_composite_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the per-ray stages (the reference gets these from autograd).
 * Gradients flow to sigma, rgb, extras and (through dists*||rd||) the ray directions; depth samples
 * and depth_map carry none (lib/nerf_utils.py:145, run.py:197-200, 261).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_composite_bwd_args {
  int64_t n_rays;
  int n_a; int n_b;             /* as nfi_composite_args */
  const float* ray_directions;
  const float* depth_a; const float* sigma_a; const float* rgb_a;
  const float* depth_b; const float* sigma_b; const float* rgb_b;
  int n_extra; const float* extra_a; const float* extra_b;
  int white_background;
  const float* g_rgb_map;    /* [N,3] upstream gradient */
  const float* g_mask;       /* [N] or NULL */
  const float* g_extra_map;  /* [N,n_extra] or NULL */
  float* g_sigma_a; float* g_rgb_a;   /* [N,n_a], [N,n_a,3] out */
  float* g_sigma_b; float* g_rgb_b;   /* [N,n_b], [N,n_b,3] out (n_b > 0) */
  float* g_extra_a; float* g_extra_b; /* out or NULL */
  float* g_ray_directions;            /* [N,3] out or NULL */
  /* 0: lists a and b are dense ([N,n_a], [N,n_b]).  > 0: every per-sample array (inputs and gradients) has this many
   * entries per ray, e.g. n_a + n_b with depth_b = depth_a + n_a for the stash layout of nfi_render_fwd */
  int list_row_stride;
} nfi_composite_bwd_args;
int nfi_composite_bwd(const nfi_composite_bwd_args* a, nfi_stream_t stream);
/* x = o + d*t: g_points [N,S,3], depth [N,S] -> g_ray_origins [N,3], g_ray_directions [N,3] (either may be NULL) */
int nfi_points_bwd(const float* g_points, const float* depth, int64_t n_rays, int n_samples,
                   float* g_ray_origins, float* g_ray_directions, nfi_stream_t stream);
/* nfi_raygen backward: g_ray_origins / g_ray_directions [N,3] (either may be NULL) ->
 * g_cam2world [B,4,4] (zeroed inside), g_focal [B] or NULL.  a->ray_origins/ray_directions are ignored. */
int nfi_raygen_bwd(const nfi_raygen_args* a, const float* g_ray_origins, const float* g_ray_directions,
                   float* g_cam2world, float* g_focal, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Field query backward: autograd of the sampler closure (models/generator.py:587-681), i.e.
 * grid_sample backward (scatter-add into the planes + coordinate gradients), the decoder MLP, the
 * Laplace-CDF density and the colour head, with the forward recomputed per tile.
 *   upstream: g_sigma [B,P], g_rgb [B,P,3], optional g_sdf [B,P], g_semantics [B,P,A]
 *   results : g_texels [B,3,R,R,32] (ACCUMULATED: caller zeroes it; convert with nfi_texels_to_planes),
 *             g_w1 [64,32], g_b1 [64], g_w2 [n_out,64], g_b2 [n_out], g_attention_values [B,A,3],
 *             g_beta [1], g_alpha [1] (all ACCUMULATED into caller-zeroed buffers, raw-parameter
 *             gradients with the equalized-lr gains applied), g_points [B,P,3] or NULL (written).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_field_bwd_args {
  int n_scenes;
  int64_t points_per_scene;          /* <= 2^30 (<= 2^25 with scatter_mode 1): the kernels address a scene's points by 32-bit offsets */
  const float* points;
  const void* texels; int plane_res; int texel_dtype;   /* any storage type; g_texels is fp32 (view-direction decoder: fp32 texels only) */
  const float* decoder_image;                            /* forward operand image */
  const float* w1; const float* w2;                      /* raw decoder weights (backward operands) */
  int n_attention; const float* attention_values;
  int use_sdf; const float* beta; const float* alpha;
  float scene_range;
  const float* g_sigma; const float* g_rgb; const float* g_sdf; const float* g_semantics;
  float* g_texels; float* g_points;
  float* g_w1; float* g_b1; float* g_w2; float* g_b2;
  float* g_attention_values; float* g_beta; float* g_alpha;
  void* workspace; size_t workspace_bytes;               /* >= nfi_decoder_bwd_image_floats()*4 bytes */
  /* points_only = 1: only g_points is produced (no plane scatter, no parameter gradients; all
   * other g_* outputs may be NULL).  normalize_g_points = 1: each g_points row is L2-normalised
   * (eps 1e-12).  Together with g_sdf == 1 this yields the analytic surface normals
   * normalize(d sdf / d x) of models/generator.py:599-623 without autograd. */
  int points_only; int normalize_g_points;
  /* view-direction decoder (decoder_image from nfi_decoder_pack_viewdir; w2 is [33,64], g_w2 [33,64], g_b2 [33]):
   * ray_features [B, P/samples_per_ray, NFI_RAY_FEATURE_PITCH], w3 raw [n3,32]; gradients g_ray_features (same
   * padded shape, zero-initialised by the caller, columns 1..32 accumulate), g_w3 [n3,32], g_b3 [n3]. */
  const float* ray_features; int samples_per_ray; const float* w3;
  float* g_ray_features; float* g_w3; float* g_b3;
  /* plane-gradient scatter.  0: fp32 atomics straight from the backward kernel (384 atomic dwords per point).
   * 1: binned - the kernel writes the per-point feature gradient (128 B) to the workspace, the points are counting-
   * sorted by texel cell per plane (by 16x16-texel tile through global memory, by cell inside LDS), and the sorted
   * entries are reduced in registers with one set of line-coalesced atomics per run of a cell; needs
   * nfi_field_bwd_workspace_bytes(a) of workspace (177 B per point).  Same result up to fp32 summation order. */
  int scatter_mode;
  int texel_layout;              /* of texels AND g_texels */
  /* Order hint (0: none): the points are [rays][samples_per_ray] with the rays in row-major order of an image
   * rays_per_row wide (what nfi_render_fwd's training stash holds).  The kernel then walks the rays in 16 x 16- (or 8 x 8-)
   * pixel tiles, one tile at a time per XCD, for L2 locality of the texel gather; results do not depend on it beyond the
   * summation order of the parameter gradients.  Ignored unless samples_per_ray is a multiple of 64 and the image
   * divides into whole tiles. */
  int rays_per_row;
} nfi_field_bwd_args;
size_t nfi_field_bwd_workspace_bytes(const nfi_field_bwd_args* a);  /* for the scatter_mode / decoder of *a */
size_t nfi_decoder_bwd_image_floats(void);          /* workspace floats, plain decoder */
size_t nfi_decoder_bwd_image_floats_viewdir(void);  /* workspace floats, view-direction decoder */
int nfi_field_query_bwd(const nfi_field_bwd_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Distance field + its spatial gradient as one differentiable operator: the regulariser branch of
 * Generator.forward (models/generator.py:505-585).  The reference gets d(sdf)/d(x) from
 * torch.autograd.grad(..., create_graph=True) through a double-differentiable grid_sample
 * (lib/ops.py:58-120); here the gradient is a second OUTPUT, so nfi_sdf_gradient_bwd is that double backward.
 *   points [B,P,3] (inside the cube: stratified samples, lib/ops.py:20-26), texels fp32 [B,3,R,R,32],
 *   raw decoder parameters w1 [64,32] b1 [64] w2 [n_out,64] b2 [n_out] (row 0 = distance head)
 *   fwd: sdf [B,P], gradient [B,P,3] = d sdf / d points
 *   bwd: upstream g_sdf [B,P] / g_gradient [B,P,3] (either may be NULL) -> g_texels [B,3,R,R,32], g_w1 [64,32],
 *        g_b1 [64], g_w2 [64] (row 0), g_b2 [1]; all ACCUMULATED into (zero-initialise them).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_sdf_gradient_args {
  int n_scenes;
  int64_t points_per_scene;
  const float* points;
  const float* texels; int plane_res;
  float scene_range;
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* sdf; float* gradient;                       /* forward outputs */
  const float* g_sdf; const float* g_gradient;       /* backward inputs */
  float* g_texels; float* g_w1; float* g_b1; float* g_w2; float* g_b2;
  int texel_layout;              /* of texels AND g_texels */
} nfi_sdf_gradient_args;
int nfi_sdf_gradient_fwd(const nfi_sdf_gradient_args* a, nfi_stream_t stream);
int nfi_sdf_gradient_bwd(const nfi_sdf_gradient_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused forward render: run.py::render (176-350) from cameras + texels to pixels in one
 * persistent launch (plus the ray set-up launch), no per-sample HBM round trips.
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_render_args {
  int n_scenes, height, width;
  int n_samples;                 /* S per pass, <= NFI_MAX_SAMPLES; without fine_sampling a single pass of up to
                                  * NFI_MAX_SAMPLES_SINGLE_PASS (run.py:2271), with stage taps / stash / viewdir but
                                  * without the extra maps, the cycle profile and termination_eps */
  int fine_sampling;             /* args.fine_sampling (run.py:259) */
  int white_background;          /* dataset_config['white_background'] (run.py:348) */
  float scene_range;             /* dataset_config['scene_range'] (run.py:200) */
  /* camera (as nfi_raygen_args) */
  const float* cam2world; const float* focal; const float* bbox; const float* center;
  /* field (as nfi_field_args) */
  const void* texels; int plane_res; int texel_dtype;
  const float* decoder_image; int n_attention; const float* attention_values;
  int use_sdf; const float* beta; const float* alpha;
  /* the reference's two random draws (nerf_utils.py:115, 202); NULL = deterministic */
  const float* noise_coarse;     /* [N,S] or NULL */
  const float* noise_fine;       /* u: [N,S] with row stride below, never NULL when fine_sampling */
  int64_t noise_fine_row_stride;
  /* outputs */
  float* rgb;        /* [N,3] */
  float* depth;      /* [N] */
  float* mask;       /* [N] */
  float* semantics;  /* [N,A] or NULL: composited softmax probabilities, sum_k w_k softmax(features_k) over the merged
                      * samples (compute_semantics: run.py:231-233, 312-335; lib/nerf_utils.py:153-156); n_attention > 0 */
  /* optional stage taps (any may be NULL) */
  float* ray_origins; float* ray_directions; float* near_plane; float* far_plane; uint8_t* hit;
  float* t_coarse; float* sigma_coarse; float* rgb_coarse;   /* [N,S], [N,S], [N,S,3] */
  float* t_fine; float* sigma_fine; float* rgb_fine;
  float* t_sorted; float* weights; int32_t* perm;            /* [N,2S] */
  /* workspace: nfi_render_workspace_bytes(n_scenes*height*width) bytes */
  void* workspace; size_t workspace_bytes;
  /* 1: rays whose line misses the scene cube inflated by 1e-4 skip both passes (exact:
   * every sample of such a ray is outside the cube, sigma==0).  0: evaluate every ray. */
  int skip_missed_rays;
  /* optional hipEvent_t pair recorded on the stream immediately before / after the render kernel
   * (excludes the ray set-up launch): live per-launch kernel timing for bench.py.  NULL = off. */
  void* event_start; void* event_stop;
  /* tuning knob, 0 = default.  bit 2: hand rays out in scanline order instead of 8x8 pixel tiles
   * (results identical).  bit 3: evaluate the decoder MLP with exact-fp32 MFMA instead of the
   * split-fp16 (hi+lo, 22 significand bits) MFMA; both meet the 1e-4 parity budget (fp32 texels only: with 16-bit texel
   * storage the texels, not the MLP operands, set the precision - the call is refused).  bit 4: ONE device-wide work
   * counter instead of the per-XCD queues over square pixel blocks (results identical; the per-XCD queues take the
   * largest of 32 / 16 / 8 pixels that divides both image sides, two positions per atomic, and fall back to the single
   * counter when not even 8 does). */
  int tuning;
  /* optional uint64[12] device array: per-phase shader-cycle sums over all waves (profiling build of
   * the kernel; NULL = off): field tile {issue, wait+interp, mlp, count}, ray set-up, coarse field,
   * resample, fine field, merge, composite, rays marched, wave lifetime; fp32 texels only */
  void* profile_cycles;
  /* view-direction decoder: NULL, or [N, NFI_RAY_FEATURE_PITCH] (decoder_image from nfi_decoder_pack_viewdir;
   * the MLP then runs in exact fp32) */
  const float* ray_features;
  /* Ray termination in the FINE pass (0 = off; BASELINE cfg5: "wavefront early-termination + sample compaction").  eps in
   * (0,1): the coarse pass - hence the resampling pdf, every sample index and every depth - is untouched; fine samples
   * that lie behind the first coarse sample in front of which the COARSE transmittance prod(1 - alpha_j + 1e-10) has
   * fallen below eps are not evaluated (sigma = 0; they stay in the merge with their depth), and the surviving fine
   * samples are compacted to the low lanes by wave ballot + popcount so that whole 16-point field tiles drop out.
   * What is dropped carries at most the merged transmittance at that depth (about eps) of compositing weight:
   * |d rgb|, |d mask| <= ~eps; tests hold eps = 1e-5 to the 1e-4 parity budget at full size.  Not available together
   * with stage taps, extra maps, the cycle profile, the view-direction decoder or the exact-fp32 MLP. */
  float termination_eps;
  int texel_layout;              /* NFI_TEXELS_PLANAR (0) / NFI_TEXELS_INTERLEAVED */
  /* optional uint64[2] device array (NULL = off): shader cycles (s_memtime) and 100 MHz reference ticks
   * (s_memrealtime) between the start and the end of workgroup 0 / wave 0 of the render kernel, i.e. the shader clock
   * DURING this launch = cycles / ticks x 1e8 Hz (bench.py prices the kernel's issue rate against it) */
  void* clock_probe;
  /* pixel-row window: this call renders rows [row_offset, row_offset + height) of an image that is full_height rows
   * tall (full_height 0 = height: the whole image).  Rays, samples and pixels are bit-identical to the same rows of the
   * full render (the pixel coordinate of get_ray_bundle, lib/nerf_utils.py:36-39, is (row_offset + row) / full_height)
   * - with ONE exception: the batch-wide miss-fill of lib/nerf_utils.py:258-259 (min near / max far over the hit rays,
   * the first two cells of the workspace after nfi_render_setup) is taken over the WINDOW's rays, so rays that are
   * marched although they miss the exact cube (inside the 1e-4 inflated one, or any missed ray with skip_missed_rays =
   * 0) get the window's fill, not the image's.  A caller that shards one image max-reduces those two cells (and sums
   * the third, the hit count) over the windows between nfi_render_setup and nfi_render_fwd(rays_ready = 1):
   * parallel.allreduce_ray_setup.  Outputs, noise and taps are sized for the window.  One image sharded over the ranks of a node: SURVEY.md 8(e),
   * run.py:598-605 (res_multiplier renders). */
  int row_offset; int full_height;
  /* training stash (all three or none): the per-sample state the backward needs, ray-major with the coarse samples in
   * [0,S) and the fine samples in [S,2S) of every row - stash_t [N,2S], stash_sigma [N,2S], stash_rgb [N,2S,3], in
   * SOURCE order (not merged); without fine_sampling the rows hold the S samples of the single pass ([N,S], [N,S],
   * [N,S,3]).  nfi_composite_bwd (list_row_stride = 2S, or one list of S) and nfi_field_query_bwd over the points of
   * every ray then replace autograd of run.py:193-348.  Unlike the debug
   * taps the stash keeps the missed-ray skip: rays that are skipped get an all-zero row.  Mutually exclusive with the
   * t_coarse ... rgb_fine taps. */
  float* stash_t; float* stash_sigma; float* stash_rgb;
  /* 1: the workspace already holds the ray set-up of these cameras / this image window (nfi_render_setup with the same
   * camera, shape, scene_range and workspace arguments, ordered before this call): nfi_render_fwd then launches the render
   * kernel only.  Lets a caller run the set-up of the NEXT batch on another stream while this one renders. */
  int rays_ready;
  /* [N,3] or NULL: composited query points, sum_k w_k (o + d t_k) over the merged samples - the sampler's 'coords'
   * output (models/generator.py:643) in the semantics slot of render_volume_density (compute_coords: run.py:234-235,
   * 337-338; the encoder-training loop asks for it every iteration, run.py:1639-1646).  Like `semantics` it comes out of
   * the SAME render launch (rgb / depth / mask bit-identical to a call without it); neither is available together with
   * stage taps, the cycle profile or the exact-fp32 MLP; with the view-direction decoder (ray_features) both exist for fp32
   * texels.  (The 128 + 128 kernel parks the probabilities of `semantics` as unorm16 between the passes: |error| <= 7.7e-6
   * per sample under a convex combination.) */
  float* coords;
  /* [N,3] or NULL: the composited normal map, sum_k w_k normalize(d sdf / d x)_k over the merged samples, + (1 - mask)
   * on a white background (compute_normals: run.py:228-230, 241-245, 296-300; lib/nerf_utils.py:149-151, 159; the
   * sampler's 'normals' output, models/generator.py:599-618, here the analytic derivative of the decoder's distance
   * instead of autograd).  use_sdf only; any texel storage; same launch, same restrictions as `coords` (with the
   * view-direction decoder: fp32 texels - the distance is row 0 of its second layer). */
  float* normals;
} nfi_render_args;
size_t nfi_render_workspace_bytes(int64_t n_rays);
int nfi_render_fwd(const nfi_render_args* a, nfi_stream_t stream);
/* The ray set-up of nfi_render_fwd alone (get_ray_bundle + normalize + the scene-cube test of
 * lib/nerf_utils.py:28-91, 237-268 into the workspace, the batch-wide miss-fill reduction and the work counters cleared).
 * Reads only the camera / shape / scene_range / workspace fields (and ray_origins / ray_directions / hit if given). */
int nfi_render_setup(const nfi_render_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Camera pose from a canonical-coordinate map: replaces the .cpu().numpy() + OpenCV solvePnPGeneric round trip of
 * lib/pose_estimation.py:30-131 (compute_pose_pnp) that run.py:1709-1740 (estimate_poses_batch) makes for every
 * inversion batch.  Per image and focal proposal: Hartley-normalised DLT start, polar decomposition, Levenberg-Marquardt
 * on the reprojection error; per image the proposal with the smallest RMS reprojection error (OpenCV's definition,
 * sqrt(sum |r|^2 / 2N)) among the solutions with t_z > 0; images without a solution (or with fewer than 4 foreground
 * pixels, as in the reference) get the reference's dummy pose (t = (0,0,-10), focal 1, error 10).  4 or 5 pixels,
 * coplanar points and linear starts that end behind the camera: refinement from the 24 axis-aligned rotations instead.
 * PARITY UNPINNED: OpenCV is not available offline, no golden vectors exist (see csrc/nfi_pnp.inc).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_pnp_args {
  int n_images; int height; int width;
  const float* coords;          /* [B,H,W,3] canonical object coordinates per pixel */
  const float* mask;            /* [B,H,W]: a pixel is foreground where mask > mask_threshold (run.py:1710: 0.9) */
  float mask_threshold;
  int n_focal; const float* focal_proposals;   /* [n_focal] device array (pose_estimation.py:134-143) */
  int refine_iterations;        /* Levenberg-Marquardt passes (0: linear start only = refine=False) */
  void* workspace; size_t workspace_bytes;     /* nfi_pnp_workspace_bytes(n_images, n_focal), 8-byte aligned */
  float* world2cam;             /* [B,4,4] out: flip @ [R|t], flip = diag(1,-1,-1,1) (pose_estimation.py:119-127) */
  float* focal;                 /* [B] out: the chosen proposal */
  float* error;                 /* [B] out: its RMS reprojection error */
} nfi_pnp_args;
size_t nfi_pnp_workspace_bytes(int n_images, int n_focal);
int nfi_pose_pnp(const nfi_pnp_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Plane-producer hand-off: the tail of the LAST StyleGAN2 synthesis block
 * (models/stylegan.py:383-435: img = upsample2d(img_prev); y = torgb(x, w); img += y) fused into one kernel that
 * writes the result as texels in the INTERLEAVED layout [B,R,R,3,32] (= a channels-last [B,96,R,R] tensor), so
 * that neither nfi_planes_to_texels nor nfi_texels_to_planes is needed.
 *   x [B,Cin,R,R] NCHW activations of conv1; styles [B,Cin] = torgb.affine(w) * torgb.weight_gain (stylegan.py:365);
 *   weight [96,Cin] = torgb.weight (1x1, no demodulation); bias [96]; previous_image [B,96,R/2,R/2] or NULL.
 *   fwd: texels [B,R,R,96].
 *   bwd: g_texels [B,R,R,96] -> g_x [B,Cin,R,R], g_styles [B,Cin], optional g_weight [96,Cin] + g_bias [96] (both
 *        or neither), optional g_previous_image [B,96,R/2,R/2]; all written (zeroed inside where accumulated).
 * Cin: multiple of 16, <= 256; R: multiple of 8.
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_torgb_args {
  int n_scenes, in_channels, resolution;
  const float* x; const float* styles; const float* weight; const float* bias; const float* previous_image;
  float* texels;                                         /* forward output */
  const float* g_texels;                                 /* backward input */
  float* g_x; float* g_styles; float* g_weight; float* g_bias; float* g_previous_image;
} nfi_torgb_args;
int nfi_torgb_texels_fwd(const nfi_torgb_args* a, nfi_stream_t stream);
int nfi_torgb_texels_bwd(const nfi_torgb_args* a, nfi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Neighbours of the renderer in the inversion loop.
 *
 * 2-D augmentation warp: the image part of augment_impl (run.py:720-769): mat = [[cos r, -sin r, tx], [sin r, cos r,
 * -ty]], scaled by `scale`, translation column re-projected (run.py:745-752), F.affine_grid(align_corners=False) +
 * F.grid_sample(bilinear, padding zeros, align_corners=False); white_background: (img - 1) is warped and 1 added
 * back (run.py:755-764).  The inversion loss warps cat(prediction, target) 15 times per step (run.py:2216-2231).
 *   image / warped [N,C,H,W]; rot [N], scale [N] or NULL (= 1), translation [N,2] (the host's random draws).
 *   bwd: g_warped [N,C,H,W] -> g_image [N,C,H,W] (zeroed inside; the adjoint of the forward w.r.t. the image).
 * ------------------------------------------------------------------------------------------ */
typedef struct nfi_warp_args {
  int n_images, channels, height, width;
  const float* image; float* warped;            /* forward */
  const float* g_warped; float* g_image;        /* backward */
  const float* rot; const float* scale; const float* translation;
  int white_background;
} nfi_warp_args;
int nfi_affine_warp_fwd(const nfi_warp_args* a, nfi_stream_t stream);
int nfi_affine_warp_bwd(const nfi_warp_args* a, nfi_stream_t stream);

/* Image metrics of the inversion / evaluation loops: lib/metrics.py psnr (30-45) and iou (79-94), per image.
 *   pred / target: n_images x elements_per_image values in [0,1] (any layout, identical for both) -> psnr [n_images]
 *   = min(60, -10 log10(mean((clamp(pred) - clamp(target))^2)));
 *   mask_pred / mask_real: n_images x elements_per_mask -> iou [n_images] = (|a & b| + 1e-6) / (|a | b| + 1e-6) of
 *   the masks thresholded at 0.5.  out_of_range (int, optional): set to 1 if any input violates the reference's
 *   range_check (-0.1 < v < 1.1; lib/metrics.py:22-27), 0 otherwise.  Either metric may be skipped (NULL output). */
typedef struct nfi_metrics_args {
  int n_images;
  const float* pred; const float* target; int64_t elements_per_image;
  const float* mask_pred; const float* mask_real; int64_t elements_per_mask;
  float* psnr; float* iou; int* out_of_range;
} nfi_metrics_args;
int nfi_image_metrics(const nfi_metrics_args* a, nfi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFI_HIP_H */
