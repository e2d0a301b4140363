// nfi_device.hpp — device-side building blocks shared by every kernel of libnfi_hip.so.
//
// Target: gfx950 (MI355X, CDNA4) only.  One wavefront (64 lanes) owns one ray; a ray's samples
// live one (or two) per lane, so scans, neighbour differences, the inverse-CDF search and the
// 2S-key merge are intra-wavefront operations (DPP/ds_bpermute shuffles + a per-wave LDS slab),
// never HBM round trips.  The triplane gather + decoder MLP runs on 16-point tiles shaped for
// v_mfma_f32_16x16x4_f32 (exact fp32, so the 1e-4 parity budget is spent on nothing).
//
// The translation unit is compiled with -ffp-contract=off: the reference evaluates every
// elementwise step as a separate ATen kernel (un-fused mul/add, true division), and sample
// indices/masks are discontinuous in those values.  FMAs appear only where written (fmaf),
// i.e. where ATen itself fuses (torch.lerp, the L2 norm) or where only a tolerance applies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// build-time experiment knobs of the fused render kernels (product values below; tools/probes/render_variants.py)
#ifndef NFI_RENDER_OCC
#define NFI_RENDER_OCC 2         // workgroups of 4 waves per CU the render kernels are compiled and launched for
#endif

namespace nfi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;
constexpr int kC = 32;        // plane channels
constexpr int kHidden = 64;   // decoder hidden width
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kSoftplusThr2 = 28.853900817779268f;  // 20 * log2(e): softplus threshold in the log2 domain

// ---- decoder operand image (floats), built by nfi_decoder_pack -------------------------------
// W1F [8 k-steps][64 lanes][4 n-tiles]   A operand of layer 1 (rows = hidden units)
// W2F [4 n-tiles][64 lanes][4 regs]      A operand of layer 2 (rows = outputs), k-step = (nt, r)
// B1F [4 groups][4 n-tiles][4 regs]      layer-1 bias in accumulator layout, * log2(e)
// B2F [4 groups][4 regs]                 layer-2 bias in accumulator layout
constexpr int kW1F = 0;
constexpr int kW2F = kW1F + 8 * 64 * 4;
constexpr int kB1F = kW2F + 4 * 64 * 4;
constexpr int kB2F = kB1F + 64;
// fp16 hi/lo split of the same operands for the renderer's split-precision MLP (dwords = half pairs):
// W1H [4 n-tiles][hi,lo][64 lanes][4 dwords]   A operand of v_mfma_f32_16x16x32_f16, K = 32 channels
// W2H [2 k-instr][hi,lo][64 lanes][4 dwords]   K = 32 of the 64 hidden units per instruction
constexpr int kW1H = kB2F + 16;                         // 3152
constexpr int kW2H = kW1H + 4 * 2 * 64 * 4;             // 5200
constexpr int kImageFloats = kW2H + 2 * 2 * 64 * 4;     // 6224
// LDS copy of the image: a kernel stages EITHER the fp32 fragments (W1F, W2F) OR the fp16 ones, which
// then overlay the fp32 fragment area; the biases keep their place.  kLdsImageFloats floats + VF.
constexpr int kW1H_lds = 0;                             // fp16 W1 fragments overlay W1F
constexpr int kW2H_lds = 4 * 2 * 64 * 4;                // fp16 W2 fragments overlay W2F
constexpr int kLdsImageFloats = kW1H;                   // 3152
constexpr int kVF = kLdsImageFloats;                    // per-scene attention values [16 rows][4] appended in LDS
constexpr int kFieldLdsFloats = kVF + 64;               // 3216

// ---- view-direction variant (--use_viewdir, models/generator.py:189-253, 376-377, 662-663) ----------
// decoder = Linear(32,64) -> Softplus -> Linear(64, 1+32); per sample  y = leaky_relu(x_ray + f, 0.2),
// colour logits = Linear(32, A or 3)(y).  Image built by nfi_decoder_pack_viewdir (exact-fp32 MFMA only):
// W1F/B1F as above; W2V [3 row tiles][4 n-tiles][64 lanes][4 regs] (rows 0..32 of the 33 outputs, row 0 =
// distance); B2V [3][4 groups][4 regs]; W3V [3 row tiles][64 lanes][4 regs] = A operand of the third
// layer over K = the 48 (padded) second-layer rows, with W3V[row 0][k 0] = 1 passing the distance
// through; B3V [4 groups][4 regs].
constexpr int kVdW1F = 0;
constexpr int kVdB1F = kVdW1F + 8 * 64 * 4;            // 2048
constexpr int kVdW2 = kVdB1F + 64;                     // 2112
constexpr int kVdB2 = kVdW2 + 3 * 4 * 64 * 4;          // 5184
constexpr int kVdW3 = kVdB2 + 48;                      // 5232
constexpr int kVdB3 = kVdW3 + 3 * 64 * 4;              // 6000
constexpr int kVdImageFloats = kVdB3 + 16;             // 6016
constexpr int kVdVF = kVdImageFloats;
constexpr int kVdFieldLdsFloats = kVdVF + 64;          // 6080
constexpr int kRayFeatPad = 48;                        // per-ray feature row: [0, x_0..x_31, 0 x 15]

// ---- division of a 32-bit unsigned by a launch-invariant divisor ------------------------------------------------
// q = x / d as a multiply-high, two adds and two shifts (Granlund-Montgomery, round-up form: exact for every 32-bit x).
// The render kernels decode a queue position into (scene, block row, block column) once per ray with three such
// divisions; as plain `/` they are ~25 VALU instructions each (the operands are wave-uniform, but there is no scalar
// divide), this way they are scalar-ALU work.
struct FastDiv {
  uint32_t mul, shift;      // mul == 0: d == 1
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{0u, 0u};
  if (d <= 1u) return f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;                                   // l = ceil(log2 d), 1..32
  f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
  f.shift = l - 1u;
  return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, FastDiv f) {
  const uint32_t t = __umulhi(x, f.mul);
  const uint32_t q = (t + ((x - t) >> 1)) >> f.shift;
  return f.mul == 0u ? x : q;
}

// ---- small wave helpers ---------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// Ordering point between the lanes of ONE wave around its private LDS slab: LDS operations of a wave execute in
// issue order, so a compiler fence plus a wave barrier (no s_waitcnt) is all a write -> read-by-other-lanes needs.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
// order-preserving float -> uint key (total order; -0 < +0; NaNs at the ends)
__device__ __forceinline__ uint32_t ordered_key(float f) {
  uint32_t b = f2bits(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ordered_key_inv(uint32_t k) {
  uint32_t b = (k >> 31) ? (k ^ 0x80000000u) : ~k;
  return bits2f(b);
}

// ---- cross-lane primitives without LDS traffic ------------------------------------------------
// gfx950 has DPP row operations (incl. row_bcast:15/31, wave_shl/shr) and v_permlane16/32_swap:
// reductions, scans and neighbour exchanges stay on the VALU instead of going through
// ds_bpermute (LDS crossbar, ~100 cycles of dependent latency per hop).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return bits2f((uint32_t)__builtin_amdgcn_update_dpp((int)f2bits(old), (int)f2bits(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const uint64_t ob = __builtin_bit_cast(uint64_t, old), vb = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)ob, (int)(uint32_t)vb, CTRL, ROW_MASK, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(ob >> 32), (int)(uint32_t)(vb >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRowHalfMirror = 0x141, kDppRowMirror = 0x140;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;
#define NFI_DPP_ROW_SHR(n) (0x110 + (n))

// value of lane l^16 / l^32 paired with the own value, as (x, y) with x+y == v[l] + v[l^16]
__device__ __forceinline__ void swap16(float v, float& x, float& y) {
  auto p = __builtin_amdgcn_permlane16_swap(f2bits(v), f2bits(v), false, false);
  x = bits2f(p[0]); y = bits2f(p[1]);
}
__device__ __forceinline__ void swap32(float v, float& x, float& y) {
  auto p = __builtin_amdgcn_permlane32_swap(f2bits(v), f2bits(v), false, false);
  x = bits2f(p[0]); y = bits2f(p[1]);
}
__device__ __forceinline__ float sum_xor16(float v) { float x, y; swap16(v, x, y); return x + y; }
__device__ __forceinline__ float sum_xor32(float v) { float x, y; swap32(v, x, y); return x + y; }
__device__ __forceinline__ float max_xor16(float v) { float x, y; swap16(v, x, y); return fmaxf(x, y); }
__device__ __forceinline__ float max_xor32(float v) { float x, y; swap32(v, x, y); return fmaxf(x, y); }
// value held by row 0 (lanes 0-15) copied to the same column of all four rows
__device__ __forceinline__ float bcast_row0(float v) {
  float x, y;
  swap16(v, x, y);        // x = [r0, r0, r2, r2]
  swap32(x, x, y);        // x = [r0, r0, r0, r0]
  return x;
}
// neighbours in lane order (lane 63 / lane 0 receive `edge`)
__device__ __forceinline__ float lane_next(float v, float edge) { return dpp_f32<kDppWaveShl1>(edge, v); }
__device__ __forceinline__ float lane_prev(float v, float edge) { return dpp_f32<kDppWaveShr1>(edge, v); }

__device__ __forceinline__ float row_allreduce_sum(float v) {
  v += dpp_f32<kDppQuadXor1>(0.0f, v);
  v += dpp_f32<kDppQuadXor2>(0.0f, v);
  v += dpp_f32<kDppRowHalfMirror>(0.0f, v);
  v += dpp_f32<kDppRowMirror>(0.0f, v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return sum_xor32(sum_xor16(row_allreduce_sum(v))); }
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<kDppQuadXor1>(v, v));
  v = fmaxf(v, dpp_f32<kDppQuadXor2>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowHalfMirror>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowMirror>(v, v));
  return max_xor32(max_xor16(v));
}
// inclusive scans over the 64 lanes in double: 4 in-row Hillis-Steele steps + 2 row broadcasts
__device__ __forceinline__ double wave_incl_scan_add(double v, int lane) {
  (void)lane;
  v += dpp_f64<NFI_DPP_ROW_SHR(1)>(0.0, v);
  v += dpp_f64<NFI_DPP_ROW_SHR(2)>(0.0, v);
  v += dpp_f64<NFI_DPP_ROW_SHR(4)>(0.0, v);
  v += dpp_f64<NFI_DPP_ROW_SHR(8)>(0.0, v);
  v += dpp_f64<kDppRowBcast15, 0xa>(0.0, v);
  v += dpp_f64<kDppRowBcast31, 0xc>(0.0, v);
  return v;
}
// the same in fp32 (the renderer's opt-in fast mode; exact for small integer counts: the merge's histogram)
__device__ __forceinline__ float wave_incl_scan_add_f32(float v) {
  v += dpp_f32<NFI_DPP_ROW_SHR(1)>(0.0f, v);
  v += dpp_f32<NFI_DPP_ROW_SHR(2)>(0.0f, v);
  v += dpp_f32<NFI_DPP_ROW_SHR(4)>(0.0f, v);
  v += dpp_f32<NFI_DPP_ROW_SHR(8)>(0.0f, v);
  v += dpp_f32<kDppRowBcast15, 0xa>(0.0f, v);
  v += dpp_f32<kDppRowBcast31, 0xc>(0.0f, v);
  return v;
}
__device__ __forceinline__ double wave_incl_scan_mul(double v, int lane) {
  (void)lane;
  v *= dpp_f64<NFI_DPP_ROW_SHR(1)>(1.0, v);
  v *= dpp_f64<NFI_DPP_ROW_SHR(2)>(1.0, v);
  v *= dpp_f64<NFI_DPP_ROW_SHR(4)>(1.0, v);
  v *= dpp_f64<NFI_DPP_ROW_SHR(8)>(1.0, v);
  v *= dpp_f64<kDppRowBcast15, 0xa>(1.0, v);
  v *= dpp_f64<kDppRowBcast31, 0xc>(1.0, v);
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// Element layout for per-ray arrays of up to SPL*64 entries: element e lives in slot e/64 of
// lane e%64.
template <int SPL>
__device__ __forceinline__ void next_elem(const float (&x)[SPL], float (&y)[SPL], int lane) {
  (void)lane;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    float edge = 0.0f;
    if (j + 1 < SPL) edge = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(x[j + 1 < SPL ? j + 1 : j]), 0));
    y[j] = lane_next(x[j], edge);
  }
}

// Exclusive running product over the SPL*64 elements.  ATen's CPU cumprod accumulates a float
// row in double (acc_type<float,false>), so a double scan reproduces it to the last bit except
// on exact rounding ties; elements past the ray's length must hold 1.
template <int SPL>
__device__ __forceinline__ void excl_cumprod(const float (&x)[SPL], float (&out)[SPL], int lane) {
  double carry = 1.0;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    double inc = wave_incl_scan_mul((double)x[j], lane);
    double exc = dpp_f64<kDppWaveShr1>(1.0, inc);
    out[j] = (float)(carry * exc);
    carry = carry * readlane_f64(inc, 63);
  }
}
template <int SPL>
__device__ __forceinline__ void incl_cumsum(const float (&x)[SPL], float (&out)[SPL], int lane) {
  double carry = 0.0;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    double inc = wave_incl_scan_add((double)x[j], lane);
    out[j] = (float)(carry + inc);
    carry = carry + readlane_f64(inc, 63);
  }
}

// ||d||_2 as ATen's CPU norm kernel evaluates it for a 3-vector: an fma chain, then sqrt.
__device__ __forceinline__ float norm3(float x, float y, float z) {
  return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));  // sqrtf is correctly rounded; __fsqrt_rn lowers to a bare v_sqrt_f32 (1 ulp)
}

// torch.lerp(a, b, w) bit for bit (ATen: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w), both fused)
__device__ __forceinline__ float aten_lerp(float a, float b, float w) {
  float d = b - a;
  return (w < 0.5f) ? fmaf(w, d, a) : fmaf(-d, 1.0f - w, b);
}

// ---- camera rays (lib/nerf_utils.py:28-91, run.py:196) ----------------------------------------
struct CameraParams {
  const float* cam2world;  // [B,4,4]
  const float* focal;      // [B] or null (ortho)
  const float* bbox;       // [B,2,2] or null
  const float* center;     // [B,2] or null
  int height, width;       // of the FULL image (pixel coordinates); rays are indexed over `rows` x width
  int normalize;
  int rows, row0;          // window of image rows this launch covers: [row0, row0 + rows)  (whole image: height, 0)
};

__device__ __forceinline__ void make_ray(const CameraParams& c, int b, int row, int col, float (&o)[3],
                                         float (&d)[3]) {
  const float* M = c.cam2world + (size_t)b * 16;
  float u = (float)col / (float)c.width;
  float v = (float)row / (float)c.height;
  float cd[3], co[3];
  if (c.focal) {
    if (c.center) {
      u = (u - 0.5f * (2.0f * c.center[b * 2 + 0] - 1.0f)) - 0.5f;
      v = (v - 0.5f * (2.0f * c.center[b * 2 + 1] - 1.0f)) - 0.5f;
    } else {
      u = u - 0.5f;
      v = v - 0.5f;
    }
    if (c.bbox) {
      const float* bb = c.bbox + b * 4;  // [[x0,y0],[w,h]]
      u = (bb[2] * (u + 0.5f) + bb[0]) * 0.5f;
      v = -(bb[3] * (-v + 0.5f) + bb[1]) * 0.5f;
    }
    float f = c.focal[b];
    u = u / f;
    v = v / f;
    cd[0] = u; cd[1] = -v; cd[2] = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d[k] = (cd[0] * M[k * 4 + 0] + cd[1] * M[k * 4 + 1]) + cd[2] * M[k * 4 + 2];
      o[k] = M[k * 4 + 3];
    }
  } else {
    u = (u - 0.5f) * 2.0f;
    v = (v - 0.5f) * 2.0f;
    if (c.bbox) {
      const float* bb = c.bbox + b * 4;
      u = bb[2] * (u / 2.0f + 0.5f) + bb[0];
      v = -(bb[3] * (-v / 2.0f + 0.5f) + bb[1]);
    }
    co[0] = u; co[1] = -v; co[2] = 0.0f;
    cd[0] = 0.0f; cd[1] = 0.0f; cd[2] = -1.0f;
    float w = M[15];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      o[k] = ((co[0] * M[k * 4 + 0] + co[1] * M[k * 4 + 1]) + co[2] * M[k * 4 + 2]) + M[k * 4 + 3];
      d[k] = ((cd[0] * M[k * 4 + 0] + cd[1] * M[k * 4 + 1]) + cd[2] * M[k * 4 + 2]) / w;
    }
  }
  if (c.normalize) {
    float n = fmaxf(norm3(d[0], d[1], d[2]), 1e-12f);
    d[0] = d[0] / n; d[1] = d[1] / n; d[2] = d[2] / n;
  }
}

// ---- scene-cube slab test (lib/nerf_utils.py:237-256) -----------------------------------------
__device__ __forceinline__ bool slab_test(const float (&o)[3], const float (&d)[3], float r, float& near,
                                          float& far) {
  float lo[3], hi[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float inv = 1.0f / d[k];
    bool neg = inv < 0.0f;
    lo[k] = ((neg ? r : -r) - o[k]) * inv;
    hi[k] = ((neg ? -r : r) - o[k]) * inv;
  }
  bool hit = !((lo[0] > hi[1]) || (lo[1] > hi[0]));
  // torch.max / torch.min propagate NaN; fmaxf would not, so spell the selects out
  near = (lo[0] > lo[1] || lo[0] != lo[0]) ? lo[0] : lo[1];
  far = (hi[0] < hi[1] || hi[0] != hi[0]) ? hi[0] : hi[1];
  hit = hit && !((near > hi[2]) || (lo[2] > far));
  near = (near > lo[2] || near != near) ? near : lo[2];
  far = (far < hi[2] || far != far) ? far : hi[2];
  return hit;
}

// miss-fill + clamps (lib/nerf_utils.py:258-268)
__device__ __forceinline__ void finish_planes(bool hit, float fill_near, float fill_far, float& near, float& far) {
  near = hit ? near : fill_near;
  far = hit ? far : fill_far;
  near = (near < 0.1f) ? 0.1f : near;  // clamp_(min): NaN stays NaN
  far = (far < 0.1f) ? 0.1f : far;
  if ((far - near) < 1e-3f) far = near + 1e-3f;
}

// ---- triplane field on 16-point tiles ---------------------------------------------------------
// Texel layouts (one scene = 3*R*R texels of 32 channels either way):
//   planar       [3][R][R][32]   what nfi_planes_to_texels writes from an NCHW producer
//   interleaved  [R][R][3][32]   = a channels-last [96,R,R] image: a producer that emits NHWC (nfi_torgb_texels_fwd,
//                                or any torch.channels_last synthesis output) is read in place, no hand-off kernel
// Only the three strides below differ; every kernel takes them from here.
struct FieldParams {
  __amdgpu_buffer_rsrc_t rsrc;   // texels of ONE scene
  uint32_t plane_bytes;          // distance between the three planes
  uint32_t pix_bytes;            // distance between x-adjacent texels
  uint32_t row_bytes;            // distance between y-adjacent texels = R * pix_bytes
  uint32_t row_pix_bytes;        // row_bytes + pix_bytes (the diagonal neighbour)
  uint32_t scene_bytes;          // 3*R*R*texel_bytes
  int res;                       // R
  float res_m1;                  // R-1
  int n_attention;               // A (0: direct rgb)
  int use_sdf;
  float inv_alpha;               // 1/alpha
  float beta;
  float neg_log2e_over_beta;      // -log2(e)/beta: exp(-|d|/beta) = exp2(|d| * this)
  const float* lds;              // LDS: decoder operand image (shared by the block)
  const float* vf;               // LDS: this scene's attention values in accumulator layout [16 rows][4]
  // normal maps (field_wave<..., NRM>): W1' transposed as the fp16 hi / lo A operands of the contraction over the hidden
  // units (v_mfma_f32_16x16x32_f16), [2 channel tiles][2 k-steps][hi | lo][64 lanes] x 8 halves, and row 0 of W2' in
  // accumulator layout [4 groups][4 n-tiles][4 regs]
  const float* w1t;
  const float* w2r0;
};
constexpr int kW1TFloats = 2 * 2 * 2 * 64 * 4, kNrmLdsFloats = kW1TFloats + 64 + 4;     // (+ the scale-back factor)

__device__ __forceinline__ void split_f16x8(const float (&x)[8], f16x8& hi, f16x8& lo);

// The two tables above from the fp32 section of the decoder operand image (global), by the threads of a block:
//   W1F[s][(g', m')][nt] = W1'[unit 16 nt + m'][channel c(s, g')], c(s, g') = s < 4 ? 4 g' + s : 16 + 4 g' + (s - 4)
//   A operand of G^T[16 channels x 16 points] += W1'^T[channels x 32 units] GH[32 units x points], lane (i = lane & 15,
//   kg = lane >> 4), element e: W1'[unit 32 kk + 16 (e >> 2) + 4 kg + (e & 3)][channel 16 ct + i] - the k-slot order in which
//   the accumulator layout of layer 1 (rows 4 g + r of n-tiles 2 kk, 2 kk + 1) is the B operand (as in layer 2)
//   W2F[nt][(g, m)][r] = W2'[row m][unit 16 nt + 4 g + r]  ->  w2r0[g][nt][r] = W2F[nt][(g, 0)][r]
// (w1f / w2f: float offsets of the W1 fragments and of the second layer's fragments in `image` - kW1F / kW2F of the plain
//  decoder image, kVdW1F / kVdW2 of the view-direction one, whose tile 0 holds the distance row)
__device__ __forceinline__ void pow2_normaliser(float amax, float& s, float& inv);
__device__ __forceinline__ void stage_normal_operands(float* dst, const float* image, int w1f = kW1F, int w2f = kW2F) {
  for (int idx = threadIdx.x; idx < 256; idx += blockDim.x) {
    const int l = idx & 63, kk = (idx >> 6) & 1, ct = idx >> 7;
    const int kg = l >> 4, c = 16 * ct + (l & 15);
    const int gq = (c & 15) >> 2, sq = (c < 16 ? 0 : 4) + (c & 3);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int u = 32 * kk + 16 * (e >> 2) + 4 * kg + (e & 3);
      x[e] = image[w1f + (((sq * 64 + (16 * gq + (u & 15))) << 2) + (u >> 4))];
    }
    f16x8 hi, lo;
    split_f16x8(x, hi, lo);
    reinterpret_cast<u32x4*>(dst)[((ct * 2 + kk) * 2 + 0) * 64 + l] = __builtin_bit_cast(u32x4, hi);
    reinterpret_cast<u32x4*>(dst)[((ct * 2 + kk) * 2 + 1) * 64 + l] = __builtin_bit_cast(u32x4, lo);
  }
  // the distance row of the second layer under ONE power-of-two scale for the whole launch: GH = sigmoid(h) * W2'[0] is
  // bounded by |W2'[0]|, so the largest entry is brought to [2^12, 2^13) - the hi + lo fp16 split then resolves 2^-24 of a
  // point's GH as long as its largest component is within 2^-12 of the row's (fp16 subnormals start 2^-24 below 1), which
  // is what the per-point normaliser of round 6's first form bought with 16 max + 2 cross-lane steps + 24 multiplies a tile
  float amax = 0.0f;
  for (int i = 0; i < 64; ++i) amax = fmaxf(amax, fabsf(image[w2f + ((((i >> 2) & 3) * 64 + 16 * (i >> 4)) << 2) + (i & 3)]));
  float ssc, s_inv;
  pow2_normaliser(amax, ssc, s_inv);
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    const int r = i & 3, nt = (i >> 2) & 3, g = i >> 4;
    dst[kW1TFloats + i] = image[w2f + ((nt * 64 + 16 * g) << 2) + r] * ssc * 4096.0f;
  }
  if (threadIdx.x == 0) dst[kW1TFloats + 64] = s_inv * (1.0f / 4096.0f);
}

// per-point gather set-up in sample layout: unnormalised, border-clamped plane coordinates.
// grid_sample(align_corners=True): u = ((p+1)/2)*(R-1); border: clamp to [0,R-1].  The left
// texel index is clamped to R-2 so that its right/lower neighbour always exists; at u == R-1 the
// fraction becomes 1 and the value is the same texel the reference reads with weight 1.
__device__ __forceinline__ void plane_coord(float p, float res_m1, int res, int& i0, float& fr) {
  float u = ((p + 1.0f) / 2.0f) * res_m1;
  u = fminf(fmaxf(u, 0.0f), res_m1);
  float fl = floorf(u);
  fl = fminf(fl, (float)(res - 2));
  i0 = (int)fl;
  fr = u - fl;
}

template <int TEX>
__device__ __forceinline__ void load_texel8(const FieldParams& P, uint32_t voff, uint32_t soff, int imm, float (&t)[8]);

template <>
__device__ __forceinline__ void load_texel8<0>(const FieldParams& P, uint32_t voff, uint32_t soff, int imm,
                                               float (&t)[8]) {
  // fp32 texel = 128 B; this lane (group g) owns channels [4g,4g+4) and [16+4g,16+4g+4)
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(P.rsrc, voff + imm, soff, 0);
  u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(P.rsrc, voff + imm + 64, soff, 0);
  t[0] = bits2f(a.x); t[1] = bits2f(a.y); t[2] = bits2f(a.z); t[3] = bits2f(a.w);
  t[4] = bits2f(b.x); t[5] = bits2f(b.y); t[6] = bits2f(b.z); t[7] = bits2f(b.w);
}
template <>
__device__ __forceinline__ void load_texel8<1>(const FieldParams& P, uint32_t voff, uint32_t soff, int imm,
                                               float (&t)[8]) {
  // bf16 texel = 64 B; this lane owns channels [8g,8g+8)
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(P.rsrc, voff + imm, soff, 0);
  t[0] = bits2f(a.x << 16); t[1] = bits2f(a.x & 0xFFFF0000u);
  t[2] = bits2f(a.y << 16); t[3] = bits2f(a.y & 0xFFFF0000u);
  t[4] = bits2f(a.z << 16); t[5] = bits2f(a.z & 0xFFFF0000u);
  t[6] = bits2f(a.w << 16); t[7] = bits2f(a.w & 0xFFFF0000u);
}

template <>
__device__ __forceinline__ void load_texel8<2>(const FieldParams& P, uint32_t voff, uint32_t soff, int imm,
                                               float (&t)[8]) {
  // fp16 texel = 64 B; same channel ownership as bf16
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(P.rsrc, voff + imm, soff, 0);
  const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 h = __builtin_bit_cast(f16x2, w[i]);
    t[2 * i] = (float)h.x; t[2 * i + 1] = (float)h.y;
  }
}

struct TileOut {
  float sdf, sigma, r, g, b;
};

// One tile = 16 points, processed in three steps:
//   tile_issue    - 24 x buffer_load_dwordx4 (3 planes x 4 bilinear corners x 2 half-lines); the lane
//                   owns 16-byte chunk q of each half-line
//   tile_bilinear - consumes the 96 texel registers into the lane's 8 interpolated features
//   tile_mlp      - decoder MLP on MFMA + density / colour epilogue (MFMA lane layout)
template <int TEX>
struct TileTex {
  float v[3][4][8];
};

// xi: packed integer texel coordinates of point j (x | y<<10 | z<<20)
template <int TEX>
__device__ __forceinline__ void tile_issue(const FieldParams& P, int g, uint32_t xi, TileTex<TEX>& T) {
  const uint32_t x0 = xi & 1023u, y0 = (xi >> 10) & 1023u, z0 = (xi >> 20) & 1023u;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const uint32_t a0 = (pl == 2) ? y0 : x0;      // W index: x, x, y
    const uint32_t b0 = (pl == 0) ? y0 : z0;      // H index: y, z, z
    // (the x / y / diagonal neighbours ride on the scalar offset of the buffer instruction: no extra vector math)
    const uint32_t voff = (uint32_t)pl * P.plane_bytes + __umul24(b0 * (uint32_t)P.res + a0, P.pix_bytes) + (uint32_t)g * 16u;
    load_texel8<TEX>(P, voff, 0, 0, T.v[pl][0]);
    load_texel8<TEX>(P, voff, P.pix_bytes, 0, T.v[pl][1]);
    load_texel8<TEX>(P, voff, P.row_bytes, 0, T.v[pl][2]);
    load_texel8<TEX>(P, voff, P.row_pix_bytes, 0, T.v[pl][3]);
  }
}

template <int TEX>
__device__ __forceinline__ void tile_bilinear(const TileTex<TEX>& T, float fx, float fy, float fz, float (&feat)[8]) {
#pragma unroll
  for (int s = 0; s < 8; ++s) feat[s] = 0.0f;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const float fa = (pl == 2) ? fy : fx;
    const float fb = (pl == 0) ? fy : fz;
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float acc = feat[s];
      acc = fmaf(w00, T.v[pl][0][s], acc);
      acc = fmaf(w10, T.v[pl][1][s], acc);
      acc = fmaf(w01, T.v[pl][2][s], acc);
      acc = fmaf(w11, T.v[pl][3][s], acc);
      feat[s] = acc;
    }
  }
}

// fp16 texel storage: the texels stay PACKED (48 registers instead of 96) and the blend reads each half in place with
// v_fma_mix_f32 - fma(w, float(texel), acc) in one instruction, the same arithmetic as conversion + fma - so 16-bit
// storage costs no conversion instructions (round 2-3: 96 v_cvt_f32_f16 per tile on top of the blend made fp16 texels
// slower than fp32 ones in an instruction-bound kernel).  (gfx950 has no bf16 form of the instruction: TileTex<1> below.)
template <>
struct TileTex<2> {
  uint32_t r[3][4][4];
};
template <>
__device__ __forceinline__ void tile_issue<2>(const FieldParams& P, int g, uint32_t xi, TileTex<2>& T) {
  const uint32_t x0 = xi & 1023u, y0 = (xi >> 10) & 1023u, z0 = (xi >> 20) & 1023u;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const uint32_t a0 = (pl == 2) ? y0 : x0, b0 = (pl == 0) ? y0 : z0;
    const uint32_t voff = (uint32_t)pl * P.plane_bytes + __umul24(b0 * (uint32_t)P.res + a0, P.pix_bytes) + (uint32_t)g * 16u;
    const uint32_t soff[4] = {0u, P.pix_bytes, P.row_bytes, P.row_pix_bytes};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(P.rsrc, voff, soff[c], 0);
      T.r[pl][c][0] = a.x; T.r[pl][c][1] = a.y; T.r[pl][c][2] = a.z; T.r[pl][c][3] = a.w;
    }
  }
}
template <>
__device__ __forceinline__ void tile_bilinear<2>(const TileTex<2>& T, float fx, float fy, float fz, float (&feat)[8]) {
#pragma unroll
  for (int s = 0; s < 8; ++s) feat[s] = 0.0f;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const float fa = (pl == 2) ? fy : fx;
    const float fb = (pl == 0) ? fy : fz;
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w[4] = {ga * gb, fa * gb, ga * fb, fa * fb};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(feat[2 * i]) : "v"(w[c]), "v"(T.r[pl][c][i]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(feat[2 * i + 1]) : "v"(w[c]), "v"(T.r[pl][c][i]));
      }
    }
  }
}

// bf16 texel storage, packed as well (round 5).  gfx950 has no bf16 form of v_fma_mix, and left to itself the compiler
// widens the texels as they arrive (96 registers again, round 4's attempt: 219 registers, scratch when capped).  Here the
// widening shift / mask and the fma are ONE asm statement per texel half with an early-clobber temporary: the widened
// value never lives beyond it, so the tile is 48 packed registers and the inference kernel fits three workgroups per CU
// like the fp16 one.  Same arithmetic and order as conversion at load time + fmaf (bit-identical images).
template <>
struct TileTex<1> {
  uint32_t r[3][4][4];
};
template <>
__device__ __forceinline__ void tile_issue<1>(const FieldParams& P, int g, uint32_t xi, TileTex<1>& T) {
  const uint32_t x0 = xi & 1023u, y0 = (xi >> 10) & 1023u, z0 = (xi >> 20) & 1023u;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const uint32_t a0 = (pl == 2) ? y0 : x0, b0 = (pl == 0) ? y0 : z0;
    const uint32_t voff = (uint32_t)pl * P.plane_bytes + __umul24(b0 * (uint32_t)P.res + a0, P.pix_bytes) + (uint32_t)g * 16u;
    const uint32_t soff[4] = {0u, P.pix_bytes, P.row_bytes, P.row_pix_bytes};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(P.rsrc, voff, soff[c], 0);
      T.r[pl][c][0] = a.x; T.r[pl][c][1] = a.y; T.r[pl][c][2] = a.z; T.r[pl][c][3] = a.w;
    }
  }
}
template <>
__device__ __forceinline__ void tile_bilinear<1>(const TileTex<1>& T, float fx, float fy, float fz, float (&feat)[8]) {
#pragma unroll
  for (int s = 0; s < 8; ++s) feat[s] = 0.0f;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const float fa = (pl == 2) ? fy : fx;
    const float fb = (pl == 0) ? fy : fz;
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w[4] = {ga * gb, fa * gb, ga * fb, fa * fb};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t tmp;
        asm("v_lshlrev_b32 %1, 16, %3\n\tv_fmac_f32 %0, %2, %1" : "+v"(feat[2 * i]), "=&v"(tmp) : "v"(w[c]), "v"(T.r[pl][c][i]));
        asm("v_and_b32 %1, 0xffff0000, %3\n\tv_fmac_f32 %0, %2, %1" : "+v"(feat[2 * i + 1]), "=&v"(tmp) : "v"(w[c]), "v"(T.r[pl][c][i]));
      }
    }
  }
}

// The same gather + blend one plane at a time (identical arithmetic and order: plane 0, 1, 2; corners 00, 10, 01, 11).
template <int TEX>
__device__ __forceinline__ void tile_gather_planewise(const FieldParams& P, int g, uint32_t xi, float fx, float fy, float fz,
                                                      float (&feat)[8]) {
  const uint32_t x0 = xi & 1023u, y0 = (xi >> 10) & 1023u, z0 = (xi >> 20) & 1023u;
#pragma unroll
  for (int s = 0; s < 8; ++s) feat[s] = 0.0f;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const uint32_t a0 = (pl == 2) ? y0 : x0, b0 = (pl == 0) ? y0 : z0;
    const uint32_t voff = (uint32_t)pl * P.plane_bytes + __umul24(b0 * (uint32_t)P.res + a0, P.pix_bytes) + (uint32_t)g * 16u;
    float tv[4][8];
    load_texel8<TEX>(P, voff, 0, 0, tv[0]);
    load_texel8<TEX>(P, voff, P.pix_bytes, 0, tv[1]);
    load_texel8<TEX>(P, voff, P.row_bytes, 0, tv[2]);
    load_texel8<TEX>(P, voff, P.row_pix_bytes, 0, tv[3]);
    const float fa = (pl == 2) ? fy : fx, fb = (pl == 0) ? fy : fz;
    const float ga = 1.0f - fa, gb = 1.0f - fb;
    const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float acc = feat[s];
      acc = fmaf(w00, tv[0][s], acc);
      acc = fmaf(w10, tv[1][s], acc);
      acc = fmaf(w01, tv[2][s], acc);
      acc = fmaf(w11, tv[3][s], acc);
      feat[s] = acc;
    }
    // Pin the order: without it the IR optimiser sinks this plane's (pure) arithmetic below the next planes' loads - all
    // 24 loads then go out first and the 96 texel registers are back (what the round-3 'plane-wise' experiments actually
    // measured).  The empty volatile asm makes the eight sums exist here, its memory clobber keeps the next loads behind it.
    asm volatile("" : "+v"(feat[0]), "+v"(feat[1]), "+v"(feat[2]), "+v"(feat[3]), "+v"(feat[4]), "+v"(feat[5]), "+v"(feat[6]), "+v"(feat[7]) : : "memory");
    __builtin_amdgcn_sched_barrier(0);       // (and the machine scheduler from undoing it)
  }
}

// One plane of a tile: the four corner loads, and their blend into the running sums (the arithmetic and its order are
// those of tile_bilinear: plane 0, 1, 2; corners 00, 10, 01, 11 - every gather form below gives the same bits).
template <int TEX>
__device__ __forceinline__ void plane_issue(const FieldParams& P, int g, uint32_t xi, int pl, float (&dst)[4][8]) {
  const uint32_t x0 = xi & 1023u, y0 = (xi >> 10) & 1023u, z0 = (xi >> 20) & 1023u;
  const uint32_t a0 = (pl == 2) ? y0 : x0, b0 = (pl == 0) ? y0 : z0;
  const uint32_t voff = (uint32_t)pl * P.plane_bytes + __umul24(b0 * (uint32_t)P.res + a0, P.pix_bytes) + (uint32_t)g * 16u;
  load_texel8<TEX>(P, voff, 0, 0, dst[0]);
  load_texel8<TEX>(P, voff, P.pix_bytes, 0, dst[1]);
  load_texel8<TEX>(P, voff, P.row_bytes, 0, dst[2]);
  load_texel8<TEX>(P, voff, P.row_pix_bytes, 0, dst[3]);
}
__device__ __forceinline__ void plane_blend(int pl, float fx, float fy, float fz, const float (&src)[4][8], float (&feat)[8]) {
  const float fa = (pl == 2) ? fy : fx, fb = (pl == 0) ? fy : fz;
  const float ga = 1.0f - fa, gb = 1.0f - fb;
  const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float acc = feat[s];
    acc = fmaf(w00, src[0][s], acc);
    acc = fmaf(w10, src[1][s], acc);
    acc = fmaf(w01, src[2][s], acc);
    acc = fmaf(w11, src[3][s], acc);
    feat[s] = acc;
  }
}
// the running sums exist HERE and later loads stay behind them (see tile_gather_planewise)
__device__ __forceinline__ void pin_sums(float (&feat)[8]) {
  asm volatile("" : "+v"(feat[0]), "+v"(feat[1]), "+v"(feat[2]), "+v"(feat[3]), "+v"(feat[4]), "+v"(feat[5]), "+v"(feat[6]), "+v"(feat[7]) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Two rounds: planes 0 and 1 in flight together (64 texel registers), then plane 2 (32).  One dependent gather round
// trip fewer per tile than plane by plane, 32 registers fewer than all at once.
template <int TEX>
__device__ __forceinline__ void tile_gather_two_rounds(const FieldParams& P, int g, uint32_t xi, float fx, float fy, float fz,
                                                       float (&feat)[8]) {
  float tv[2][4][8];
#pragma unroll
  for (int s = 0; s < 8; ++s) feat[s] = 0.0f;
  plane_issue<TEX>(P, g, xi, 0, tv[0]);
  plane_issue<TEX>(P, g, xi, 1, tv[1]);
  plane_blend(0, fx, fy, fz, tv[0], feat);
  plane_blend(1, fx, fy, fz, tv[1], feat);
  pin_sums(feat);
  plane_issue<TEX>(P, g, xi, 2, tv[0]);
  plane_blend(2, fx, fy, fz, tv[0], feat);
}

// The normal map's gather (fp32 texels): features AND derivative features (tile_bilinear_deriv's arithmetic) with two planes
// in flight at any time - planes 0 and 1 go out together, plane 2 takes plane 0's registers as soon as they are consumed: 64
// texel registers + 24 derivative sums instead of 96 + 24 (which spills), one round trip exposed instead of three.
__device__ __forceinline__ void plane_blend_deriv(int pl, float fx, float fy, float fz, const float (&src)[4][8], float (&feat)[8],
                                                  float (&dfeat)[3][8]) {
  // written on channel PAIRS: the compiler's SLP pass leaves this blend scalar (288 instructions per tile; as v_pk_fma_f32 /
  // v_pk_add_f32 it is 144), the plain blend it packs by itself
  typedef float v2f __attribute__((ext_vector_type(2)));
  const float fa = (pl == 2) ? fy : fx, fb = (pl == 0) ? fy : fz;
  const float ga = 1.0f - fa, gb = 1.0f - fb;
  const float w00 = ga * gb, w10 = fa * gb, w01 = ga * fb, w11 = fa * fb;
  const v2f W00 = {w00, w00}, W10 = {w10, w10}, W01 = {w01, w01}, W11 = {w11, w11};
  const v2f FA = {fa, fa}, FB = {fb, fb}, GA = {ga, ga}, GB = {gb, gb};
  const int ia = (pl == 2) ? 1 : 0, ib = (pl == 0) ? 1 : 2;
#pragma unroll
  for (int s = 0; s < 8; s += 2) {
    const v2f t0 = {src[0][s], src[0][s + 1]}, t1 = {src[1][s], src[1][s + 1]}, t2 = {src[2][s], src[2][s + 1]},
              t3 = {src[3][s], src[3][s + 1]};
    v2f acc = {feat[s], feat[s + 1]};
    acc = __builtin_elementwise_fma(W00, t0, acc);
    acc = __builtin_elementwise_fma(W10, t1, acc);
    acc = __builtin_elementwise_fma(W01, t2, acc);
    acc = __builtin_elementwise_fma(W11, t3, acc);
    feat[s] = acc.x; feat[s + 1] = acc.y;
    v2f da = {dfeat[ia][s], dfeat[ia][s + 1]}, db = {dfeat[ib][s], dfeat[ib][s + 1]};
    da = __builtin_elementwise_fma(FB, t3 - t2, __builtin_elementwise_fma(GB, t1 - t0, da));
    db = __builtin_elementwise_fma(FA, t3 - t1, __builtin_elementwise_fma(GA, t2 - t0, db));
    dfeat[ia][s] = da.x; dfeat[ia][s + 1] = da.y;
    dfeat[ib][s] = db.x; dfeat[ib][s + 1] = db.y;
  }
}
// plane 0's sums exist HERE and later loads stay behind them (see tile_gather_planewise)
__device__ __forceinline__ void pin_deriv_sums(float (&feat)[8], float (&dfeat)[3][8]) {
  asm volatile("" : "+v"(feat[0]), "+v"(feat[1]), "+v"(feat[2]), "+v"(feat[3]), "+v"(feat[4]), "+v"(feat[5]), "+v"(feat[6]), "+v"(feat[7]),
               "+v"(dfeat[0][0]), "+v"(dfeat[0][1]), "+v"(dfeat[0][2]), "+v"(dfeat[0][3]), "+v"(dfeat[0][4]), "+v"(dfeat[0][5]),
               "+v"(dfeat[0][6]), "+v"(dfeat[0][7]), "+v"(dfeat[1][0]), "+v"(dfeat[1][1]), "+v"(dfeat[1][2]), "+v"(dfeat[1][3]),
               "+v"(dfeat[1][4]), "+v"(dfeat[1][5]), "+v"(dfeat[1][6]), "+v"(dfeat[1][7]) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// density / colour epilogue on the decoder outputs o (lane (g,j): rows 4g..4g+3 of point j; row 0 =
// distance/density, rows 1.. = colour logits pre-scaled by log2e)
// SEMP: where the softmax probabilities of a point go.  0: sem[n] is a global row [A] (the sampler closure's
// 'semantics' output); < 0: the same table with unorm16 entries, pitch -SEMP (tile_epilogue); > 0: sem[n] addresses the
// point's column of a per-wave LDS table [A][SEMP] (attribute-major, the
// fused renderer composites it after the merge; the pitch makes the four channel groups' stores conflict-free).
template <bool ATT, int N, int SEMP = 0>
__device__ __forceinline__ void tile_epilogue(const FieldParams& P, int lane, const f32x4 (&o)[N], const float (&outside)[N],
                                              float* const (&sem)[N], TileOut (&res)[N]) {
  const int g = lane >> 4;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const float sdf = bcast_row0(o[n].x);     // group 0's output row 0 -> all four channel groups
    res[n].sdf = sdf;
    if (P.use_sdf) {
      // sigma = (1/alpha) * (0.5 + 0.5*sign(-d)*(1 - exp(-|d|/beta))) * (1 - outside)
      float e = __builtin_amdgcn_exp2f(fabsf(sdf) * P.neg_log2e_over_beta);
      float sgn = (sdf < 0.0f) ? 0.5f : ((sdf > 0.0f) ? -0.5f : 0.0f);
      float cdf = 0.5f + sgn * (1.0f - e);
      res[n].sigma = P.inv_alpha * (cdf * (1.0f - outside[n]));
    } else {
      float d = sdf - 1.0f;
      float sp = (d > 20.0f) ? d : log1pf(__expf(d));
      res[n].sigma = sp * (1.0f - outside[n]);
    }
    if constexpr (ATT) {
      // softmax over features 1..A spread over (group, reg); rows are pre-scaled by log2e
      const int A = P.n_attention;
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        bool valid = (row >= 1) && (row <= A);
        m = valid ? fmaxf(m, o[n][r]) : m;
      }
      m = max_xor32(max_xor16(m));
      const f32x4* vf = reinterpret_cast<const f32x4*>(P.vf) + g * 4;
      float se = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f;
      float e4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        bool valid = (row >= 1) && (row <= A);
        float e = valid ? __builtin_amdgcn_exp2f(o[n][r] - m) : 0.0f;
        e4[r] = e;
        f32x4 v = vf[r];
        se += e;
        sr = fmaf(e, v.x, sr);
        sg = fmaf(e, v.y, sg);
        sb = fmaf(e, v.z, sb);
      }
      se = sum_xor32(sum_xor16(se)); sr = sum_xor32(sum_xor16(sr)); sg = sum_xor32(sum_xor16(sg)); sb = sum_xor32(sum_xor16(sb));
      float inv = __builtin_amdgcn_rcpf(se);
      inv = inv * (2.0f - se * inv);      // one Newton step: the quotient is then within 1 ulp
      res[n].r = sr * inv; res[n].g = sg * inv; res[n].b = sb * inv;
      if (sem[n]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row = 4 * g + r;
          if constexpr (SEMP > 0) {
            // explicit LDS address space: a ds_write, ordered with the wave's other LDS traffic (a flat store is not)
            auto* q = (__attribute__((address_space(3))) float*)sem[n];
            if (row >= 1 && row <= A) q[(row - 1) * SEMP] = e4[r] * inv;
          } else if constexpr (SEMP < 0) {
            // 16-bit table (the 128 + 128 kernel: half the LDS, two workgroups per CU): unorm16, p = q / 65535,
            // |error| <= 7.7e-6 per sample - and the composited map is a convex combination of the samples
            auto* q = (__attribute__((address_space(3))) unsigned short*)sem[n];
            const float pq = fminf(__builtin_rintf(e4[r] * inv * 65535.0f), 65535.0f);
            if (row >= 1 && row <= A) q[(row - 1) * (-SEMP)] = (unsigned short)(int)pq;
          } else {
            if (row >= 1 && row <= A) sem[n][row - 1] = e4[r] * inv;
          }
        }
      }
    } else {
      // rgb = sigmoid(f)*2.004 - 1.002, features (rows 1..3 of group 0) pre-scaled by log2e
      float r1 = bcast_row0(o[n].y), r2 = bcast_row0(o[n].z), r3 = bcast_row0(o[n].w);
      res[n].r = (1.0f / (1.0f + __builtin_amdgcn_exp2f(-r1))) * 2.004f - 1.002f;
      res[n].g = (1.0f / (1.0f + __builtin_amdgcn_exp2f(-r2))) * 2.004f - 1.002f;
      res[n].b = (1.0f / (1.0f + __builtin_amdgcn_exp2f(-r3))) * 2.004f - 1.002f;
    }
  }
}

// Decoder MLP + density / colour epilogue for N tiles at once (N = 2 in the renderer: the two tiles'
// MFMA chains, softplus blocks and cross-lane reductions are independent, so a wave has twice the
// instruction-level parallelism against the dependent latencies that dominate this phase, and the
// operand fragments are read from LDS once for both).
// Returns (in every lane of the four groups) the decoder outputs for point j of each tile.
//   outside[n]: 1.0f if the point is outside the scene cube.
//   sem[n]: if non-null, softmax probabilities are written to sem[n][A] (global) for this point.
// x[0..7] -> hi = fp16(x) (round toward zero), lo = fp16(x - hi): hi + lo carries 22 significand bits for values of
// order one.  The resolution is ABSOLUTE, 2^-24 (the fp16 subnormal spacing; subnormals are preserved by the MFMA in
// the default kernel mode): fine for activations, whose products accumulate with O(1) terms, NOT for operands of
// arbitrary scale such as gradients - those are brought to [1, 2) first (pow2_normaliser below).
__device__ __forceinline__ void split_f16x8(const float (&x)[8], f16x8& hi, f16x8& lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    auto h = __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]);
    float r0, r1;
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(x[2 * i]));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(x[2 * i + 1]));
    auto l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
    lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
  }
}

// Power-of-two factor s that brings a non-negative magnitude amax into [1, 2), and inv = 1 / s.  Used per point on the
// gradient operands of the split-fp16 MFMAs (exact scaling; the fp32 accumulator is scaled back with inv).  Zero and
// fp32-subnormal magnitudes are left alone (s = 1), inf / nan keep propagating.
__device__ __forceinline__ void pow2_normaliser(float amax, float& s, float& inv) {
  uint32_t e = ((uint32_t)f2bits(amax) >> 23) & 0xffu;
  e = e == 0u ? 127u : (e > 253u ? 253u : e);
  s = bits2f((int)((254u - e) << 23));
  inv = bits2f((int)(e << 23));
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 nfi_fp16x4;
__device__ __forceinline__ f16x4 lds_read_tr16(const char* p) {
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) nfi_fp16x4*)(p)));
}
__device__ __forceinline__ f16x8 lds_read_tr16x2(const char* p, int second) {
  const f16x4 a = lds_read_tr16(p), b = lds_read_tr16(p + second);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float row_allreduce_max(float v) {
  v = fmaxf(v, dpp_f32<kDppQuadXor1>(v, v));
  v = fmaxf(v, dpp_f32<kDppQuadXor2>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowHalfMirror>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowMirror>(v, v));
  return v;
}
// Sticky power-of-two scale of a set of weight-gradient accumulators whose A operands are tile-scaled fp16 (see
// nfi_backward_field.inc).  am_pt: the lane's point's largest entry (uniform over the four 16-lane rows); es: biased
// exponent of the scale (0: not set yet); et_max: biased exponent of the largest tile maximum seen.  A tile's largest
// entry lands in [2^8, 2^9) when the scale is (re)set, which happens when it would leave [2^-2, 2^14) under the current
// one.  The scale never rises more than 2^40 above what the largest tile seen so far asks for: the accumulators then stay
// far from overflow when they are rescaled, and a tile 2^40 below the largest one contributes nothing to an fp32 sum
// anyway.  Returns the factor the accumulators of the set have to be multiplied by (1: nothing to do).
__device__ __forceinline__ float sticky_scale(float am_pt, int& es, int& et_max) {
  const int et = (int)(((uint32_t)__builtin_amdgcn_readfirstlane((int)f2bits(row_allreduce_max(am_pt))) >> 23) & 0xffu);
  float fac = 1.0f;
  if (et == 0 || et == 255) { if (es == 0) es = 127; return fac; }
  et_max = max(et_max, et);
  const int want = min(max(min(262 - et, 302 - et_max), 1), 254);
  if (es == 0) { es = want; return fac; }
  const int u = et + es - 254;
  if (u > 13 || (u < -2 && want > es)) {
    fac = bits2f((uint32_t)min(max(want - es + 127, 1), 254) << 23);
    es = want;
  }
  return fac;
}
// fp16 pair 2^x, x = es + e_pt - 254: a scale 2^(es - 127) over the point's own normaliser (inv_pt = 2^(e_pt - 127) is what
// pow2_normaliser returned as `inv`); 0 below the fp16 normal range
__device__ __forceinline__ uint32_t ratio_f16x2(int es, float inv_pt) {
  const int x = es + (int)((f2bits(inv_pt) >> 23) & 0xffu) - 254;
  const uint32_t hb = x < -14 ? 0u : (uint32_t)(min(x, 15) + 15) << 10;
  return hb | (hb << 16);
}

// G^T[32 channels x 16 points] = W1'^T[32 x 64] * GH[64 x 16] (rows 16 ct + 4 g + r: the channel ownership of `feat`), the
// operand of the normal map (field_wave).  gh: d(distance) / d(h2) of the lane's hidden units = sigmoid(h) * W2'[0][unit] in
// accumulator layout.  Split fp16 like the layers themselves (round 6: 12 K = 32 MFMAs instead of 32 exact-fp32 ones, which
// run at the fp32 VECTOR rate on gfx950): a gradient has no natural scale and the hi + lo split resolves 2^-24 absolute,
// so GH is scaled per point by a power of two and the accumulator scaled back exactly.
__device__ __forceinline__ void normal_contraction(const FieldParams& P, int lane, const f32x4 (&gh)[4], f32x4& Gout0, f32x4& Gout1) {
  const u32x4* w1t = reinterpret_cast<const u32x4*>(P.w1t);
  // four accumulators (one per channel tile and K half, summed at the end): chains of 3 dependent MFMAs instead of 6
  f32x4 G[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const f32x4 ga = gh[2 * kk], gb = gh[2 * kk + 1];
    const float xs[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
    f16x8 sh, sl;
    split_f16x8(xs, sh, sl);
    const f16x8 ah = __builtin_bit_cast(f16x8, w1t[((0 * 2 + kk) * 2 + 0) * 64 + lane]);
    const f16x8 al = __builtin_bit_cast(f16x8, w1t[((0 * 2 + kk) * 2 + 1) * 64 + lane]);
    const f16x8 bh = __builtin_bit_cast(f16x8, w1t[((1 * 2 + kk) * 2 + 0) * 64 + lane]);
    const f16x8 bl = __builtin_bit_cast(f16x8, w1t[((1 * 2 + kk) * 2 + 1) * 64 + lane]);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    G[kk][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, sh, zero, 0, 0, 0);
    G[kk][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, sh, zero, 0, 0, 0);
    G[kk][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, sl, G[kk][0], 0, 0, 0);
    G[kk][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, sl, G[kk][1], 0, 0, 0);
    G[kk][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, sh, G[kk][0], 0, 0, 0);
    G[kk][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, sh, G[kk][1], 0, 0, 0);
  }
  const f32x4 G0 = G[0][0] + G[1][0], G1 = G[0][1] + G[1][1];
  // (the launch-wide scale of the staged W2' row is NOT taken back: every axis of a point's gradient carries it, and a
  //  power of two passes through the normalisation exactly - P.w2r0[64] holds it for whoever needs the magnitude)
  Gout0 = G0; Gout1 = G1;
}

// PREC 0: exact fp32 MFMA (v_mfma_f32_16x16x4_f32, 48 per tile).
// PREC 1: split fp16 (v_mfma_f32_16x16x32_f16, 18 per tile): every operand is hi + lo in fp16 and the
//         products hi*hi + hi*lo + lo*hi are accumulated in fp32 - 2^-21 relative per product instead of
//         fp32's 2^-24, an order of magnitude below the 1e-4 parity budget, at 1/5 of the matrix-pipe time.
//         On gfx950 the f32-input MFMA runs at the f32 VECTOR rate and does not overlap with VALU work of
//         other waves (tools/probes/mfma_valu_overlap.hip), so this time comes straight off the kernel.
// NRM: Gout[n][ct] receives G = W1'^T (sigmoid(h) * W2'[0]) of tile n - d(distance) / d(feature) up to the positive factors
// the normalisation drops -, rows 16 ct + 4 g + r of point j: the operand of the normal map (field_wave).
template <bool ATT, int N, int PREC, int SEMP = 0, bool NRM = false>
__device__ __forceinline__ void tile_mlp(const FieldParams& P, int lane, const float (&feat)[N][8],
                                         const float (&outside)[N], float* const (&sem)[N], TileOut (&res)[N],
                                         f32x4 (&Gout)[N][2]) {
  const int g = lane >> 4;
  const f32x4* ldsv = reinterpret_cast<const f32x4*>(P.lds);
  f32x4 o[N];
  if constexpr (PREC == 1) {
    const u32x4* ldsu = reinterpret_cast<const u32x4*>(P.lds);
    f16x8 fh[N], fl[N];
#pragma unroll
    for (int n = 0; n < N; ++n) split_f16x8(feat[n], fh[n], fl[n]);
    f32x4 acc1[N][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 b = ldsv[(kB1F >> 2) + g * 4 + nt];
      const f16x8 wh = __builtin_bit_cast(f16x8, ldsu[(kW1H_lds >> 2) + (nt * 2 + 0) * 64 + lane]);
      const f16x8 wl = __builtin_bit_cast(f16x8, ldsu[(kW1H_lds >> 2) + (nt * 2 + 1) * 64 + lane]);
#pragma unroll
      for (int n = 0; n < N; ++n) acc1[n][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fh[n], b, 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) acc1[n][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fl[n], acc1[n][nt], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) acc1[n][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fh[n], acc1[n][nt], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      f32x4 gh[4];      // NRM: d(distance) / d(h2) of the lane's hidden units = sigmoid(h) * W2'[0][unit] (accumulator layout)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // softplus in base 2.  v_med3(sp, h, 128) == (h > thr ? h : sp) of nn.Softplus wherever the two differ
          // in fp32: for h beyond ~25 log2(1 + 2^h) rounds to h, beyond 128 2^h overflows and the median is h.
          float h = acc1[n][nt][r];
          float e = __builtin_amdgcn_exp2f(h);
          float sp = __builtin_amdgcn_logf(1.0f + e);
          acc1[n][nt][r] = __builtin_amdgcn_fmed3f(sp, h, 128.0f);
          if constexpr (NRM) {
            // d softplus2 / d h2 = e / (1 + e) = 1 - 1 / (1 + e): 1 where e overflows, absolute error 2^-24 (GH is a sum's operand)
            // (times the staged W2' entry in the same instruction: w - w / (1 + e))
            const float w2u = reinterpret_cast<const f32x4*>(P.w2r0)[g * 4 + nt][r];
            gh[nt][r] = fmaf(-w2u, __builtin_amdgcn_rcpf(1.0f + e), w2u);
          }
        }
      // (right behind this tile's softplus, so that GH never outlives it)
      if constexpr (NRM) normal_contraction(P, lane, gh, Gout[n][0], Gout[n][1]);
    }
    const f32x4 b2 = ldsv[(kB2F >> 2) + g];
#pragma unroll
    for (int n = 0; n < N; ++n) o[n] = b2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const f16x8 wh = __builtin_bit_cast(f16x8, ldsu[(kW2H_lds >> 2) + (kk * 2 + 0) * 64 + lane]);
      const f16x8 wl = __builtin_bit_cast(f16x8, ldsu[(kW2H_lds >> 2) + (kk * 2 + 1) * 64 + lane]);
      f16x8 sh[N], sl[N];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const float x[8] = {acc1[n][2 * kk][0], acc1[n][2 * kk][1], acc1[n][2 * kk][2], acc1[n][2 * kk][3],
                            acc1[n][2 * kk + 1][0], acc1[n][2 * kk + 1][1], acc1[n][2 * kk + 1][2], acc1[n][2 * kk + 1][3]};
        split_f16x8(x, sh[n], sl[n]);
      }
#pragma unroll
      for (int n = 0; n < N; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, sh[n], o[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, sl[n], o[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, sh[n], o[n], 0, 0, 0);
    }
  } else {
    // ---- layer 1: H^T[64 x 16] = W1'[64 x 32] * F^T[32 x 16], bias pre-loaded, log2 domain ----
    f32x4 acc1[N][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 b = ldsv[(kB1F >> 2) + g * 4 + nt];
#pragma unroll
      for (int n = 0; n < N; ++n) acc1[n][nt] = b;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const f32x4 w = ldsv[(kW1F >> 2) + s * 64 + lane];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        acc1[n][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, feat[n][s], acc1[n][0], 0, 0, 0);
        acc1[n][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, feat[n][s], acc1[n][1], 0, 0, 0);
        acc1[n][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, feat[n][s], acc1[n][2], 0, 0, 0);
        acc1[n][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, feat[n][s], acc1[n][3], 0, 0, 0);
      }
    }
    // softplus in base 2: sp2 = log2(1 + 2^h2)  (= softplus(h)/ln2; ln2 is folded into W2')
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float h = acc1[n][nt][r];
          float e = __builtin_amdgcn_exp2f(h);
          float sp = __builtin_amdgcn_logf(1.0f + e);
          acc1[n][nt][r] = (h > kSoftplusThr2) ? h : sp;
        }
    static_assert(!NRM, "the normal map's contraction is built for the split-fp16 decoder (PREC 1)");
    // ---- layer 2: O^T[16 x 16] = W2'[16 x 64] * SP^T[64 x 16]; two accumulators per tile ----
    f32x4 o0[N], o1[N];
    const f32x4 b2 = ldsv[(kB2F >> 2) + g];
#pragma unroll
    for (int n = 0; n < N; ++n) { o0[n] = b2; o1[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 w = ldsv[(kW2F >> 2) + nt * 64 + lane];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        o0[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, acc1[n][nt][0], o0[n], 0, 0, 0);
        o1[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, acc1[n][nt][1], o1[n], 0, 0, 0);
        o0[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, acc1[n][nt][2], o0[n], 0, 0, 0);
        o1[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, acc1[n][nt][3], o1[n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) o[n] = o0[n] + o1[n];   // lane (j,g): outputs 4g..4g+3 of point j; row 0 = sdf/density
  }

  tile_epilogue<ATT, N, SEMP>(P, lane, o, outside, sem, res);
}
template <bool ATT, int N, int PREC, int SEMP = 0>
__device__ __forceinline__ void tile_mlp(const FieldParams& P, int lane, const float (&feat)[N][8],
                                         const float (&outside)[N], float* const (&sem)[N], TileOut (&res)[N]) {
  f32x4 unused[N][2];
  tile_mlp<ATT, N, PREC, SEMP, false>(P, lane, feat, outside, sem, res, unused);
}

// View-direction variant of the decoder (exact fp32 MFMA): layer 1 as above, layer 2 with 33 outputs
// in three 16-row tiles, y = leaky_relu(x_ray + f, 0.2) on the 32 feature rows (row 0 = distance goes
// through), layer 3 over K = those 48 accumulator rows - the accumulator layout of one layer is the B
// operand of the next, as between layers 1 and 2.  xr[n][t]: rows 16t+4g..+3 of the padded ray feature
// of tile n's point j.  P.lds holds the nfi_decoder_pack_viewdir image.
template <bool ATT, int N, int SEMP = 0, bool NRM = false>
__device__ __forceinline__ void tile_mlp_vd(const FieldParams& P, int lane, const float (&feat)[N][8],
                                            const f32x4 (&xr)[N][3], const float (&outside)[N], float* const (&sem)[N],
                                            TileOut (&res)[N], f32x4 (&Gout)[N][2]) {
  const int g = lane >> 4;
  const f32x4* ldsv = reinterpret_cast<const f32x4*>(P.lds);
  f32x4 acc1[N][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 b = ldsv[(kVdB1F >> 2) + g * 4 + nt];
#pragma unroll
    for (int n = 0; n < N; ++n) acc1[n][nt] = b;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const f32x4 w = ldsv[(kVdW1F >> 2) + s * 64 + lane];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      acc1[n][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, feat[n][s], acc1[n][0], 0, 0, 0);
      acc1[n][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, feat[n][s], acc1[n][1], 0, 0, 0);
      acc1[n][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, feat[n][s], acc1[n][2], 0, 0, 0);
      acc1[n][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, feat[n][s], acc1[n][3], 0, 0, 0);
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    f32x4 gh[4];        // NRM: sigmoid(h) * W2'[0][unit] - the distance is row 0 of the second layer here too
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float h = acc1[n][nt][r];
        float e = __builtin_amdgcn_exp2f(h);
        float sp = __builtin_amdgcn_logf(1.0f + e);
        acc1[n][nt][r] = (h > kSoftplusThr2) ? h : sp;
        if constexpr (NRM) {
          const float w2u = reinterpret_cast<const f32x4*>(P.w2r0)[g * 4 + nt][r];                // (as in tile_mlp)
          gh[nt][r] = fmaf(-w2u, __builtin_amdgcn_rcpf(1.0f + e), w2u);
        }
      }
    if constexpr (NRM) normal_contraction(P, lane, gh, Gout[n][0], Gout[n][1]);
  }
  f32x4 o2[N][3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const f32x4 b = ldsv[(kVdB2 >> 2) + t * 4 + g];
#pragma unroll
    for (int n = 0; n < N; ++n) o2[n][t] = b;
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const f32x4 w = ldsv[(kVdW2 >> 2) + (t * 4 + nt) * 64 + lane];
#pragma unroll
      for (int n = 0; n < N; ++n) {
        o2[n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, acc1[n][nt][0], o2[n][t], 0, 0, 0);
        o2[n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, acc1[n][nt][1], o2[n][t], 0, 0, 0);
        o2[n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, acc1[n][nt][2], o2[n][t], 0, 0, 0);
        o2[n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, acc1[n][nt][3], o2[n][t], 0, 0, 0);
      }
    }
  f32x4 o[N], oB[N];
  const f32x4 b3 = ldsv[(kVdB3 >> 2) + g];
#pragma unroll
  for (int n = 0; n < N; ++n) { o[n] = b3; oB[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const f32x4 w = ldsv[(kVdW3 >> 2) + t * 64 + lane];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      f32x4 y;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = o2[n][t][r] + xr[n][t][r];
        y[r] = (v >= 0.0f) ? v : v * 0.2f;                       // F.leaky_relu(x + f, 0.2)
      }
      if (t == 0 && g == 0) y[0] = o2[n][0][0];                   // row 0: the distance passes through
      o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, y[0], o[n], 0, 0, 0);
      oB[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, y[1], oB[n], 0, 0, 0);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, y[2], o[n], 0, 0, 0);
      oB[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, y[3], oB[n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) o[n] = o[n] + oB[n];
  tile_epilogue<ATT, N, SEMP>(P, lane, o, outside, sem, res);
}

struct SampleOut {
  float sdf, sigma, r, g, b;
  float nx, ny, nz;     // field_wave<..., NRM>: normalize(d sdf / d x) of the sample (models/generator.py:609-618)
};

// Field query for the (up to) 64 points a wave holds one per lane.  px,py,pz: WORLD coordinates;
// valid: lane holds a point.  SKIP (fused renderer): tiles whose 16 points are all outside the cube
// (or invalid) are skipped: sigma is exactly 0 there (the reference multiplies by (1-mask),
// generator.py:633) and rgb is reported as 0 (its compositing weight is exactly 0).  Without SKIP
// (the sampler closure) outside points get the border-clamped colour/distance the reference returns.
// sem_base: null, or global pointer to this wave's [64][A] semantics rows (written for valid points); with SEMP > 0
// the wave's LDS table [A][SEMP] at the column of this call's point 0 (tile_epilogue).
// stage: 16 x 36 floats of LDS owned by this wave (feature-tile transpose).
// prof: null, or 4 cycle accumulators {tile set-up + load issue, load wait + interpolation,
// transpose + MLP + epilogue, tiles} filled with s_memtime deltas (profiling builds only)
// VD: view-direction decoder; xray = padded per-ray features [rays][kRayFeatPad], ray_idx = this lane's ray.
// NRM (fused renderer, compute_normals): the sample's unit normal normalize(d sdf / d x) - the decoder's distance
// differentiated analytically: G = W1'^T (sigmoid(h) * W2'[0]) on 12 split-fp16 MFMAs per tile, times d feature / d axis of the
// bilinear footprint, which comes out of the tile's ONE gather in forward mode (plane_blend_deriv; 16-bit texels are widened
// at the load there), one tile per turn.  (Round 6's first form took G's inner products with the corner differences from a
// second, cache-hot gather behind the decoder: 0.56 x the plain rate where this is 0.72 x - profiles/r6/
// extra_maps_render_times.log.)  The positive factors common to the three axes (the plane mean's 1/3, the base-2 scalings)
// drop out of the normalisation.  Border-clamped coordinates carry no gradient, like grid_sample's.
template <int TEX, bool ATT, bool SKIP, int PREC = 0, bool VD = false, int SEMP = 0, bool NRM = false>
__device__ __forceinline__ SampleOut field_wave(const FieldParams& P, float scene_range, int lane, float px, float py,
                                                float pz, bool valid, float* sem_base, bool* outside_flag,
                                                float* stage, unsigned long long* prof = nullptr,
                                                const float* xray = nullptr, int ray_idx = 0) {
  // reference: x / scene_range, mask = any(|x| > 1)   (true division, generator.py:604-607)
  float qx = px / scene_range, qy = py / scene_range, qz = pz / scene_range;
  bool out = (fabsf(qx) > 1.0f) || (fabsf(qy) > 1.0f) || (fabsf(qz) > 1.0f);
  if (outside_flag) *outside_flag = out;
  int x0, y0, z0;
  float fx, fy, fz;
  plane_coord(qx, P.res_m1, P.res, x0, fx);
  plane_coord(qy, P.res_m1, P.res, y0, fy);
  plane_coord(qz, P.res_m1, P.res, z0, fz);
  if (!valid) { x0 = y0 = z0 = 0; fx = fy = fz = 0.0f; }  // keep NaN/garbage out of the address math
  const int xi = (int)((uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)z0 << 20));
  // bit 0: outside, bit 1: valid, bits 2-4 (NRM): the axis' unnormalised coordinate lies strictly inside (0, R-1)
  int flags = (out ? 1 : 0) | (valid ? 2 : 0);
  if constexpr (NRM) {
    auto inside = [&](float q) { float u = ((q + 1.0f) / 2.0f) * P.res_m1; return (u > 0.0f && u < P.res_m1) ? 1 : 0; };
    flags |= (inside(qx) << 2) | (inside(qy) << 3) | (inside(qz) << 4);
  }
  const uint64_t live = SKIP ? __ballot(valid && !out) : __ballot(valid);
  uint32_t tm = 0;   // wave-uniform 4-bit mask of tiles with work
#pragma unroll
  for (int t = 0; t < 4; ++t) tm |= (((live >> (16 * t)) & 0xFFFFull) != 0 ? 1u : 0u) << t;

  SampleOut so;
  so.sdf = 0.0f; so.sigma = 0.0f; so.r = 0.0f; so.g = 0.0f; so.b = 0.0f;
  so.nx = 0.0f; so.ny = 0.0f; so.nz = 0.0f;
  if (tm == 0) return so;
  const int j = lane & 15, g = lane >> 4;

  // Two lane layouts are used per tile (16 points):
  //   load layout  L: lane = 4*p + q  (p = point, q = 16-byte chunk) - the 4 lanes of a point are
  //                   CONSECUTIVE, so a wave load reads 16 x 64 contiguous bytes and the texture
  //                   addresser coalesces each lane quad into one line access (4x fewer accesses
  //                   than with the MFMA layout, whose consecutive lanes are different points);
  //   MFMA layout  M: lane = 16*g + j (j = point, g = channel group) - fixed by v_mfma_*.
  // The 16 x 32 interpolated features are moved L -> M through a small LDS tile (2 b128 writes + 2
  // b128 reads per lane);
  // chunk q of L and channel group g of M hold the same channels, so the W1 operand image is
  // independent of this choice.
  const int lp = lane >> 2, lq = lane & 3;        // L layout
  // gather + interpolation + L->M transpose of one tile; returns the tile's flags for this lane's point
  auto gather_tile = [&](int t, float (&feat)[8]) -> int {
    const int srcL = 16 * t + lp, srcM = 16 * t + j;
    const float cfx = __shfl(fx, srcL, 64), cfy = __shfl(fy, srcL, 64), cfz = __shfl(fz, srcL, 64);
    const uint32_t cxi = (uint32_t)__shfl(xi, srcL, 64);
    const int fcur = __shfl(flags, srcM, 64);
    float featL[8];
    TileTex<TEX> T;
    tile_issue<TEX>(P, lq, cxi, T);
    tile_bilinear<TEX>(T, cfx, cfy, cfz, featL);
    // pitch 36 floats: conflict-free for both the 16-byte writes and the 16-byte reads
    __builtin_amdgcn_sched_barrier(0);
    f32x4* wr = reinterpret_cast<f32x4*>(stage + lp * 36 + lq * 4);
    wr[0] = f32x4{featL[0], featL[1], featL[2], featL[3]};
    wr[4] = f32x4{featL[4], featL[5], featL[6], featL[7]};
    wave_lds_fence();
    const f32x4* rd = reinterpret_cast<const f32x4*>(stage + j * 36 + g * 4);
    const f32x4 lo = rd[0], hi = rd[4];
    feat[0] = lo.x; feat[1] = lo.y; feat[2] = lo.z; feat[3] = lo.w;
    feat[4] = hi.x; feat[5] = hi.y; feat[6] = hi.z; feat[7] = hi.w;
    wave_lds_fence();
    __builtin_amdgcn_sched_barrier(0);
    return fcur;
  };
  // a point's semantics: row [A] of the global output, or its column of the wave's LDS table (SEMP > 0)
  // (SEMP < 0: a table of 16-bit entries - the pointer only carries the column's address to tile_epilogue)
  const size_t sem_pt = SEMP != 0 ? (size_t)1 : (size_t)P.n_attention;
  auto sem_col = [&](int t) -> float* {
    if constexpr (SEMP < 0) return reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(sem_base) + (16 * t + j));
    else return sem_base + (size_t)(16 * t + j) * sem_pt;
  };
  // the tail of a tile's normals, given this lane's partial (8-channel) coordinate gradients in the L layout
  auto finish_normals = [&](int t, float (&gcoord)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {                             // sum over the 4 lanes (channel chunks) of the point
      gcoord[a] += dpp_f32<kDppQuadXor1>(0.0f, gcoord[a]);
      gcoord[a] += dpp_f32<kDppQuadXor2>(0.0f, gcoord[a]);
    }
    const int flL = __shfl(flags, 16 * t + lp, 64);
    float gx = ((flL >> 2) & 1) ? gcoord[0] : 0.0f, gy = ((flL >> 3) & 1) ? gcoord[1] : 0.0f,
          gz = ((flL >> 4) & 1) ? gcoord[2] : 0.0f;
    // F.normalize(x_grad, dim=-1): the common factor (R - 1) / 2 / scene_range is applied first so that the eps clamp
    // sees the reference's magnitude
    // (and the launch-wide power of two the staged W2' row carries is taken back: exact)
    const float sc = ((P.res_m1 * 0.5f) / scene_range) * P.w2r0[64];
    gx *= sc; gy *= sc; gz *= sc;
    // (x * (1 / n) for F.normalize's x / n: one reciprocal instead of three IEEE divisions, an ulp inside the map's 3e-5)
    // (1 / max(n, eps) as rsq(max(n^2, eps^2)): one transcendental instead of sqrt + reciprocal)
    const float rn = __builtin_amdgcn_rsqf(fmaxf(fmaf(gx, gx, fmaf(gy, gy, gz * gz)), 1e-24f));
    gx *= rn; gy *= rn; gz *= rn;
    // the point's lanes 4 p .. 4 p + 3 all hold the result; sample lane 16 t + p takes it from lane 4 p
    const float sx = __shfl(gx, 4 * j, 64), sy = __shfl(gy, 4 * j, 64), sz = __shfl(gz, 4 * j, 64);
    if (g == t) { so.nx = sx; so.ny = sy; so.nz = sz; }
  };
#pragma unroll 1
  while (tm != 0) {
    if constexpr (NRM) {
      // ---- the normal map's own schedule: ONE tile per turn; the derivative features d feature / d (x, y, z) come out of
      // the tile's (only) gather, wait in registers while the decoder runs, and meet G = d sdf / d feature behind it - one turn
      // later, between the NEXT tile's load issue and its blend, where the gather's latency would otherwise be idle ----
      int tprev = -1;                       // wave-uniform: the tile whose normals are still owed
      float dL[3][8];                       // its derivative features (L layout)
      f32x4 Gp1[2];                         // its G = d sdf / d feature (M layout)
      auto normals_of_previous = [&]() {
        f32x4* wr = reinterpret_cast<f32x4*>(stage + j * 36 + g * 4);      // G: M -> L layout through the stage tile
        wr[0] = Gp1[0]; wr[4] = Gp1[1];
        wave_lds_fence();
        float gfL[8];
        {
          const f32x4* rd = reinterpret_cast<const f32x4*>(stage + lp * 36 + lq * 4);
          const f32x4 lo = rd[0], hi = rd[4];
          gfL[0] = lo.x; gfL[1] = lo.y; gfL[2] = lo.z; gfL[3] = lo.w;
          gfL[4] = hi.x; gfL[5] = hi.y; gfL[6] = hi.z; gfL[7] = hi.w;
        }
        wave_lds_fence();
        // the lane's 8-channel part of G . d feature / d axis, on channel pairs (v_pk_fma_f32: 12 instead of 24 instructions)
        typedef float v2f __attribute__((ext_vector_type(2)));
        float gcoord[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          v2f acc = {0.0f, 0.0f};
#pragma unroll
          for (int s8 = 0; s8 < 8; s8 += 2)
            acc = __builtin_elementwise_fma((v2f){gfL[s8], gfL[s8 + 1]}, (v2f){dL[a][s8], dL[a][s8 + 1]}, acc);
          gcoord[a] = acc.x + acc.y;
        }
        finish_normals(tprev, gcoord);
      };
#pragma unroll 1
      while (tm != 0) {
        const int t = __builtin_ctz(tm);
        tm &= tm - 1;
        float feat1[1][8];
        __builtin_amdgcn_s_setprio(3);
        const int srcL = 16 * t + lp, srcM = 16 * t + j;
        const float cfx = __shfl(fx, srcL, 64), cfy = __shfl(fy, srcL, 64), cfz = __shfl(fz, srcL, 64);
        const uint32_t cxi = (uint32_t)__shfl(xi, srcL, 64);
        const int f1 = __shfl(flags, srcM, 64);
        float tv[2][4][8];
        plane_issue<TEX>(P, lq, cxi, 0, tv[0]);
        plane_issue<TEX>(P, lq, cxi, 1, tv[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (tprev >= 0) normals_of_previous();
        __builtin_amdgcn_sched_barrier(0);
        float featL[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) { featL[s8] = 0.0f; dL[0][s8] = 0.0f; dL[1][s8] = 0.0f; dL[2][s8] = 0.0f; }
        plane_blend_deriv(0, cfx, cfy, cfz, tv[0], featL, dL);
        pin_deriv_sums(featL, dL);                                         // plane 2's loads stay behind plane 0's sums
        plane_issue<TEX>(P, lq, cxi, 2, tv[0]);
        plane_blend_deriv(1, cfx, cfy, cfz, tv[1], featL, dL);
        plane_blend_deriv(2, cfx, cfy, cfz, tv[0], featL, dL);
        __builtin_amdgcn_sched_barrier(0);
        {
          f32x4* wr = reinterpret_cast<f32x4*>(stage + lp * 36 + lq * 4);
          wr[0] = f32x4{featL[0], featL[1], featL[2], featL[3]};
          wr[4] = f32x4{featL[4], featL[5], featL[6], featL[7]};
          wave_lds_fence();
          const f32x4* rd = reinterpret_cast<const f32x4*>(stage + j * 36 + g * 4);
          const f32x4 lo = rd[0], hi = rd[4];
          feat1[0][0] = lo.x; feat1[0][1] = lo.y; feat1[0][2] = lo.z; feat1[0][3] = lo.w;
          feat1[0][4] = hi.x; feat1[0][5] = hi.y; feat1[0][6] = hi.z; feat1[0][7] = hi.w;
          wave_lds_fence();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        const float outs1[1] = {(f1 & 1) ? 1.0f : 0.0f};
        float* const sems1[1] = {(sem_base && (f1 & 2)) ? sem_col(t) : nullptr};
        TileOut to1[1];
        f32x4 G1[1][2];
        if constexpr (VD) {
          static_assert(PREC == 0, "the view-direction decoder exists in exact fp32 only");
          const int r1 = __shfl(ray_idx, 16 * t + j, 64);
          f32x4 xr1[1][3];
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) xr1[0][c3] = *reinterpret_cast<const f32x4*>(xray + (size_t)r1 * kRayFeatPad + 16 * c3 + 4 * g);
          tile_mlp_vd<ATT, 1, SEMP, true>(P, lane, feat1, xr1, outs1, sems1, to1, G1);
        } else {
          tile_mlp<ATT, 1, PREC, SEMP, true>(P, lane, feat1, outs1, sems1, to1, G1);
        }
        Gp1[0] = G1[0][0]; Gp1[1] = G1[0][1];
        tprev = t;
        if (g == t) { so.sdf = to1[0].sdf; so.sigma = to1[0].sigma; so.r = to1[0].r; so.g = to1[0].g; so.b = to1[0].b; }
      }
      if (tprev >= 0) normals_of_previous();
      return so;
    }
    unsigned long long c0 = prof ? __builtin_readcyclecounter() : 0;
    const int ta = __builtin_ctz(tm);
    tm &= tm - 1;
    const bool pair = tm != 0;                    // wave-uniform
    const int tb = pair ? __builtin_ctz(tm) : ta;
    tm &= tm - 1;                                 // (no-op when tm is already 0)
    float feat[2][8];
    // the gather half of the loop body (address arithmetic, 24 loads per tile, interpolation) at raised issue priority,
    // the MFMA half at the default: the waves of a SIMD then tend to take turns instead of both queueing for the
    // same pipe (measured, images identical: fp16 texels 3 workgroups per CU -2 % chairs-like / -3 % every ray hits, fp32
    // -0.2 .. -0.7 %; raising the MLP half instead: nothing)
    __builtin_amdgcn_s_setprio(3);
    const int fa = gather_tile(ta, feat[0]);
    int fb = fa;
    if (pair) {
      fb = gather_tile(tb, feat[1]);
    } else {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) feat[1][s8] = feat[0][s8];
    }
    unsigned long long c2 = prof ? __builtin_readcyclecounter() : 0;
    __builtin_amdgcn_s_setprio(0);
    const float outs[2] = {(fa & 1) ? 1.0f : 0.0f, (fb & 1) ? 1.0f : 0.0f};
    float* const sems[2] = {(sem_base && (fa & 2)) ? sem_col(ta) : nullptr, (sem_base && pair && (fb & 2)) ? sem_col(tb) : nullptr};
    TileOut to[2];
    f32x4 Gp[2][2];          // (only the normal map's schedule above asks the decoder for d distance / d feature)
    if constexpr (VD) {
      static_assert(PREC == 0, "the view-direction decoder exists in exact fp32 only");
      const int ra = __shfl(ray_idx, 16 * ta + j, 64), rb = __shfl(ray_idx, 16 * tb + j, 64);
      f32x4 xr[2][3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        xr[0][t] = *reinterpret_cast<const f32x4*>(xray + (size_t)ra * kRayFeatPad + 16 * t + 4 * g);
        xr[1][t] = *reinterpret_cast<const f32x4*>(xray + (size_t)rb * kRayFeatPad + 16 * t + 4 * g);
      }
      tile_mlp_vd<ATT, 2, SEMP, false>(P, lane, feat, xr, outs, sems, to, Gp);
    } else {
      tile_mlp<ATT, 2, PREC, SEMP, false>(P, lane, feat, outs, sems, to, Gp);
    }
    if (g == ta) { so.sdf = to[0].sdf; so.sigma = to[0].sigma; so.r = to[0].r; so.g = to[0].g; so.b = to[0].b; }
    if (pair && g == tb) { so.sdf = to[1].sdf; so.sigma = to[1].sigma; so.r = to[1].r; so.g = to[1].g; so.b = to[1].b; }
    if (prof) {
      asm volatile("" :: "v"(so.sigma), "v"(so.r));
      unsigned long long c3 = __builtin_readcyclecounter();
      prof[1] += c2 - c0; prof[2] += c3 - c2; prof[3] += pair ? 2 : 1;
    }
  }
  return so;
}

// ---- per-ray weights (lib/nerf_utils.py:164-180) ------------------------------------------------
// sigma/t hold n elements (invalid slots: sigma = 0).  dnorm = ||ray direction||.
// T: the exclusive transmittance in front of every sample.
template <int SPL>
__device__ __forceinline__ void ray_weights(const float (&sigma)[SPL], const float (&t)[SPL], int n, float dnorm,
                                            int lane, float (&w)[SPL], float (&T)[SPL]) {
  float tn[SPL], om[SPL], alpha[SPL];
  next_elem<SPL>(t, tn, lane);
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    int e = j * 64 + lane;
    float delta = (e < n - 1) ? (tn[j] - t[j]) : 0.0f;
    delta = delta * dnorm;
    float a = 1.0f - expf(-sigma[j] * delta);
    if (e >= n) a = 0.0f;
    alpha[j] = a;
    om[j] = (1.0f - a) + 1e-10f;
  }
  excl_cumprod<SPL>(om, T, lane);
#pragma unroll
  for (int j = 0; j < SPL; ++j) w[j] = alpha[j] * T[j];
}
template <int SPL>
__device__ __forceinline__ void ray_weights(const float (&sigma)[SPL], const float (&t)[SPL], int n, float dnorm,
                                            int lane, float (&w)[SPL]) {
  float T[SPL];
  ray_weights<SPL>(sigma, t, n, dnorm, lane, w, T);
}

// count of cdf entries <= u (searchsorted right=True) over the M sorted floats at cdf[] (LDS)
__device__ __forceinline__ int upper_bound_lds(const float* cdf, int M, float u) {
  int pos = 0;
#pragma unroll
  for (int step = 128; step >= 1; step >>= 1) {
    int idx = pos + step;
    if (idx <= M && cdf[idx - 1] <= u) pos = idx;
  }
  return pos;
}

}  // namespace nfi
