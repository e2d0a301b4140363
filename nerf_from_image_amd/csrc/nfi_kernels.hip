// nfi_kernels.hip — kernels and C ABI of libnfi_hip.so (gfx950 / MI355X only).
// See include/nfi_hip.h for the contract and nfi_device.hpp for the building blocks.
#include "nfi_device.hpp"
#include "../../include/nfi_hip.h"
#include "nfi_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>

using namespace nfi;

// error plumbing, argument checks and the LDS staging helpers shared with nfi_backward_field.hip: nfi_host.hpp
thread_local char nfi_err_buf[256] = "";
extern "C" const char* nfi_last_error(void) { return nfi_err_buf; }
extern "C" int nfi_version(void) { return 100; }


// ------------------------------------------------------------------------------------------------
// planes [B,3,32,R,R] <-> texels [B,3,R,R,32]
// ------------------------------------------------------------------------------------------------
// 256 pixels x 32 channels per block through LDS.  Loads: 16 bytes per lane along the pixels (a wave reads 1 KB of one
// channel row per instruction).  tile[c][p + 8 (c >> 3)] with a row pitch of 288 floats: the 16-byte LDS writes stay
// aligned and contiguous, and the transposing reads - lane (p, q) gathers channels 8q .. 8q + 7 of pixel p - hit bank
// (p + 8 q) mod 32: conflict free.  Stores: every lane 32 contiguous bytes of its texel (4 lanes = one 128-B texel).
constexpr int kP2tPixels = 256, kP2tPitch = 288;
template <int TEX>
__global__ __launch_bounds__(256) void planes_to_texels_kernel(const float* __restrict__ planes, void* __restrict__ texels,
                                                               int hw) {
  __shared__ __attribute__((aligned(16))) float tile[kC][kP2tPitch];
  const int img = blockIdx.y;                 // b*3 + plane
  const int px0 = blockIdx.x * kP2tPixels;
  const float* src = planes + (size_t)img * kC * hw;
  {
    const int c4 = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    if ((hw & 3) == 0 && px0 + kP2tPixels <= hw) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + 4 * i) * hw + px0 + 4 * c4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = r0 + 4 * i;
        *reinterpret_cast<f32x4*>(&tile[c][4 * c4 + 8 * (c >> 3)]) = v[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = r0 + 4 * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int px = px0 + 4 * c4 + e;
          tile[c][4 * c4 + e + 8 * (c >> 3)] = (px < hw) ? src[(size_t)c * hw + px] : 0.0f;
        }
      }
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 3;              // channel octet
#pragma unroll
  for (int it = 0; it < kP2tPixels / 64; ++it) {
    const int p = it * 64 + (threadIdx.x >> 2);
    if (px0 + p >= hw) continue;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tile[q * 8 + k][p + 8 * q];
    if (TEX == 0) {
      float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(texels) + ((size_t)img * hw + px0 + p) * kC + q * 8);
      dst[0] = make_float4(v[0], v[1], v[2], v[3]);
      dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else if (TEX == 2) {
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f16x2 h = {(_Float16)v[2 * k], (_Float16)v[2 * k + 1]};   // round-to-nearest-even fp32 -> fp16
        w[k] = __builtin_bit_cast(uint32_t, h);
      }
      uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(texels) + ((size_t)img * hw + px0 + p) * kC + q * 8);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // round-to-nearest-even fp32 -> bf16
        uint32_t a = f2bits(v[2 * k]), b = f2bits(v[2 * k + 1]);
        a = (a + 0x7FFFu + ((a >> 16) & 1u)) >> 16;
        b = (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
        w[k] = a | (b << 16);
      }
      uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(texels) + ((size_t)img * hw + px0 + p) * kC + q * 8);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

__global__ __launch_bounds__(256) void texels_to_planes_kernel(const float* __restrict__ texels, float* __restrict__ planes,
                                                               int hw) {
  __shared__ float tile[kC][65];
  const int img = blockIdx.y;
  const int px0 = blockIdx.x * 64;
  const int p = threadIdx.x >> 2, q = threadIdx.x & 3;
  if (px0 + p < hw) {
    const float4* src = reinterpret_cast<const float4*>(texels + ((size_t)img * hw + px0 + p) * kC + q * 8);
    float4 a = src[0], b = src[1];
    tile[q * 8 + 0][p] = a.x; tile[q * 8 + 1][p] = a.y; tile[q * 8 + 2][p] = a.z; tile[q * 8 + 3][p] = a.w;
    tile[q * 8 + 4][p] = b.x; tile[q * 8 + 5][p] = b.y; tile[q * 8 + 6][p] = b.z; tile[q * 8 + 7][p] = b.w;
  }
  __syncthreads();
  float* dst = planes + (size_t)img * kC * hw;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int c = ty; c < kC; c += 4) {
    int px = px0 + tx;
    if (px < hw) dst[(size_t)c * hw + px] = tile[c][tx];
  }
}

extern "C" int nfi_planes_to_texels(const float* planes, void* texels, int n_scenes, int plane_res, int texel_dtype,
                                    nfi_stream_t stream) {
  REQUIRE(planes && texels, "planes_to_texels: null pointer");
  REQUIRE(n_scenes > 0 && plane_res >= 2 && plane_res <= 1024, "planes_to_texels: plane_res must be in [2,1024]");
  REQUIRE(texel_dtype >= NFI_TEXEL_F32 && texel_dtype <= NFI_TEXEL_F16, "planes_to_texels: bad texel dtype");
  int hw = plane_res * plane_res;
  dim3 grid((hw + kP2tPixels - 1) / kP2tPixels, n_scenes * 3);
  if (texel_dtype == NFI_TEXEL_F32)
    hipLaunchKernelGGL(planes_to_texels_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, planes, texels, hw);
  else if (texel_dtype == NFI_TEXEL_BF16)
    hipLaunchKernelGGL(planes_to_texels_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, planes, texels, hw);
  else
    hipLaunchKernelGGL(planes_to_texels_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, planes, texels, hw);
  return check_launch("planes_to_texels");
}

extern "C" int nfi_texels_to_planes(const float* texels, float* planes, int n_scenes, int plane_res, nfi_stream_t stream) {
  REQUIRE(planes && texels, "texels_to_planes: null pointer");
  REQUIRE(n_scenes > 0 && plane_res >= 2 && plane_res <= 1024, "texels_to_planes: plane_res must be in [2,1024]");
  int hw = plane_res * plane_res;
  dim3 grid((hw + 63) / 64, n_scenes * 3);
  hipLaunchKernelGGL(texels_to_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, texels, planes, hw);
  return check_launch("texels_to_planes");
}

// ------------------------------------------------------------------------------------------------
// decoder operand image
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int feat_channel(int tex, int s, int g) {
  // which plane channel sits in feature register s of channel group g (see load_texel8)
  return tex == 0 ? ((s < 4) ? 4 * g + s : 16 + 4 * g + (s - 4)) : 8 * g + s;
}

__global__ __launch_bounds__(256) void decoder_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           int n_out, int tex, float* __restrict__ image) {
  const float gain1 = 0.17677669529663687f;  // 1/sqrt(32)  (models/stylegan.py:171)
  const float gain2 = 0.125f;                // 1/sqrt(64)
  // one element per thread over ceil(kImageFloats / 256) workgroups (one workgroup walking the image took 11 us: 25
  // dependent round trips to the weights)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kImageFloats; i += gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (i < kW2F) {
      int k = i - kW1F;
      int nt = k & 3, lane = (k >> 2) & 63, s = k >> 8;
      int row = 16 * nt + (lane & 15);
      int ch = feat_channel(tex, s, lane >> 4);
      // gain, the /3 of the three-plane mean (generator.py:328) and log2(e) for the base-2 softplus
      v = (w1[row * kC + ch] * gain1) * (kLog2e / 3.0f);
    } else if (i < kB1F) {
      int k = i - kW2F;
      int r = k & 3, lane = (k >> 2) & 63, nt = k >> 8;
      int row = lane & 15;
      int hid = 16 * nt + 4 * (lane >> 4) + r;
      if (row < n_out) {
        float w = w2[row * kHidden + hid] * gain2;
        // softplus was computed in base 2 (missing factor ln2).  Row 0 (distance / density) stays in
        // natural units; feature rows are wanted times log2(e) for the base-2 softmax/sigmoid, and
        // ln2*log2e == 1.
        v = (row == 0) ? w * kLn2 : w;
      }
    } else if (i < kB2F) {
      int k = i - kB1F;
      int r = k & 3, nt = (k >> 2) & 3, g = k >> 4;
      v = b1[16 * nt + 4 * g + r] * kLog2e;
    } else if (i < kW1H) {
      int k = i - kB2F;
      int row = k;  // 4g + r
      if (row < n_out) v = (row == 0) ? b2[0] : b2[row] * kLog2e;
    } else {
      // fp16 hi/lo operand images: one dword = two consecutive k-elements of one lane
      float w2v[2];
      int hl;
      if (i < kW2H) {
        int k = i - kW1H;
        int d = k & 3, lane = (k >> 2) & 63;
        hl = (k >> 8) & 1;
        int nt = k >> 9;
        int row = 16 * nt + (lane & 15);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int ch = feat_channel(tex, 2 * d + h, lane >> 4);
          w2v[h] = (w1[row * kC + ch] * gain1) * (kLog2e / 3.0f);
        }
      } else {
        int k = i - kW2H;
        int d = k & 3, lane = (k >> 2) & 63;
        hl = (k >> 8) & 1;
        int kk = k >> 9;
        int row = lane & 15;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int e = 2 * d + h;
          int hid = 16 * (2 * kk + (e >> 2)) + 4 * (lane >> 4) + (e & 3);
          float w = 0.0f;
          if (row < n_out) { w = w2[row * kHidden + hid] * gain2; if (row == 0) w *= kLn2; }
          w2v[h] = w;
        }
      }
      auto hi = __builtin_amdgcn_cvt_pkrtz(w2v[0], w2v[1]);
      auto lo = __builtin_amdgcn_cvt_pkrtz(w2v[0] - (float)hi[0], w2v[1] - (float)hi[1]);
      v = bits2f(hl == 0 ? __builtin_bit_cast(uint32_t, hi) : __builtin_bit_cast(uint32_t, lo));
    }
    image[i] = v;
  }
}

extern "C" size_t nfi_decoder_image_floats(void) { return (size_t)kImageFloats; }

extern "C" int nfi_decoder_pack(const float* w1, const float* b1, const float* w2, const float* b2, int n_attention,
                                int texel_dtype, float* image, nfi_stream_t stream) {
  REQUIRE(w1 && b1 && w2 && b2 && image, "decoder_pack: null pointer");
  REQUIRE(n_attention >= 0 && n_attention <= NFI_MAX_ATTENTION, "decoder_pack: attention_values must be in [0,14]");
  REQUIRE(texel_dtype >= NFI_TEXEL_F32 && texel_dtype <= NFI_TEXEL_F16, "decoder_pack: bad texel dtype");
  int n_out = n_attention > 0 ? 1 + n_attention : 4;
  hipLaunchKernelGGL(decoder_pack_kernel, dim3((kImageFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2,
                     n_out, texel_dtype, image);
  return check_launch("decoder_pack");
}

__global__ __launch_bounds__(256) void decoder_pack_vd_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                                              const float* __restrict__ w3, const float* __restrict__ b3,
                                                              int n3, int tex, float* __restrict__ image) {
  const float gain1 = 0.17677669529663687f;  // 1/sqrt(32)
  const float gain2 = 0.125f;                // 1/sqrt(64)
  const float gain3 = 0.17677669529663687f;  // 1/sqrt(32)
  for (int i = threadIdx.x; i < kVdImageFloats; i += blockDim.x) {
    float v = 0.0f;
    if (i < kVdB1F) {
      int k = i - kVdW1F;
      int nt = k & 3, lane = (k >> 2) & 63, s = k >> 8;
      int row = 16 * nt + (lane & 15);
      int ch = feat_channel(tex, s, lane >> 4);
      v = (w1[row * kC + ch] * gain1) * (kLog2e / 3.0f);
    } else if (i < kVdW2) {
      int k = i - kVdB1F;
      int r = k & 3, nt = (k >> 2) & 3, g = k >> 4;
      v = b1[16 * nt + 4 * g + r] * kLog2e;
    } else if (i < kVdB2) {
      int k = i - kVdW2;
      int r = k & 3, lane = (k >> 2) & 63, nt = (k >> 8) & 3, t = k >> 10;
      int row = 16 * t + (lane & 15);
      int hid = 16 * nt + 4 * (lane >> 4) + r;
      if (row < 33) v = (w2[row * kHidden + hid] * gain2) * kLn2;   // every row stays in natural units
    } else if (i < kVdW3) {
      int row = i - kVdB2;                                           // 16t + 4g + r
      if (row < 33) v = b2[row];
    } else if (i < kVdB3) {
      int k = i - kVdW3;
      int r = k & 3, lane = (k >> 2) & 63, t = k >> 8;
      int out = lane & 15;
      int kk = 16 * t + 4 * (lane >> 4) + r;                          // second-layer row feeding this K slot
      if (out == 0) v = (kk == 0) ? 1.0f : 0.0f;                      // the distance passes through
      else if (out <= n3 && kk >= 1 && kk <= 32) v = (w3[(out - 1) * 32 + (kk - 1)] * gain3) * kLog2e;
    } else {
      int out = i - kVdB3;
      if (out >= 1 && out <= n3) v = b3[out - 1] * kLog2e;
    }
    image[i] = v;
  }
}

extern "C" size_t nfi_decoder_image_floats_viewdir(void) { return (size_t)kVdImageFloats; }

extern "C" int nfi_decoder_pack_viewdir(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                        const float* b3, int n_attention, int texel_dtype, float* image,
                                        nfi_stream_t stream) {
  REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && image, "decoder_pack_viewdir: null pointer");
  REQUIRE(n_attention >= 0 && n_attention <= NFI_MAX_ATTENTION, "decoder_pack_viewdir: attention_values must be in [0,14]");
  REQUIRE(texel_dtype >= NFI_TEXEL_F32 && texel_dtype <= NFI_TEXEL_F16, "decoder_pack_viewdir: bad texel dtype");
  static_assert(NFI_RAY_FEATURE_PITCH == kRayFeatPad, "ray feature pitch");
  int n3 = n_attention > 0 ? n_attention : 3;
  hipLaunchKernelGGL(decoder_pack_vd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, n3,
                     texel_dtype, image);
  return check_launch("decoder_pack_viewdir");
}

// ------------------------------------------------------------------------------------------------
// rays + scene-cube planes
// ------------------------------------------------------------------------------------------------
// reduce[0] = ~key(min near | hit) (so that zero-initialised memory + atomicMax works),
// reduce[1] = key(max far | hit), reduce[2] = hit count.
// Per-thread accumulators (a thread may have seen several rays): kmin/kmax keys as above, cnt = its hit count.
// The three atomics per block all go to the same three addresses, and same-address device atomics serialise at
// ~12 ns on this chip, so the ray kernels run a grid-stride loop over at most kRayBlocks blocks.
constexpr int kRayBlocks = 256;
__device__ __forceinline__ void block_reduce_keys(uint32_t kmin, uint32_t kmax, uint32_t cnt, uint32_t* reduce) {
  __shared__ uint32_t s_min[4], s_max[4], s_cnt[4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    uint32_t a = (uint32_t)__shfl_xor((int)kmin, d, 64), b = (uint32_t)__shfl_xor((int)kmax, d, 64);
    kmin = a > kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
    cnt += (uint32_t)__shfl_xor((int)cnt, d, 64);
  }
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_min[wave] = kmin; s_max[wave] = kmax; s_cnt[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) {
      kmin = s_min[w] > kmin ? s_min[w] : kmin;
      kmax = s_max[w] > kmax ? s_max[w] : kmax;
      cnt += s_cnt[w];
    }
    if (cnt) {
      atomicMax(&reduce[0], kmin);
      atomicMax(&reduce[1], kmax);
      atomicAdd(&reduce[2], cnt);
    }
  }
}

struct RaygenOut {
  float* ro; float* rd;          // [N,3] or null
  float* near_raw; float* far_raw; uint8_t* hit;  // [N] or null (null near_raw -> no slab test)
  uint32_t* reduce;
  float scene_range;
};

__global__ __launch_bounds__(256) void raygen_kernel(CameraParams cam, int n_scenes, RaygenOut out) {
  const int hw = cam.rows * cam.width;
  const int64_t n = (int64_t)n_scenes * hw;
  uint32_t kmin = 0u, kmax = 0u, cnt = 0u;
  for (int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ray < n; ray += (int64_t)gridDim.x * blockDim.x) {
    int b = (int)(ray / hw);
    int pix = (int)(ray - (int64_t)b * hw);
    int row = pix / cam.width, col = pix - row * cam.width;
    float o[3], d[3];
    make_ray(cam, b, cam.row0 + row, col, o, d);
    if (out.ro) { out.ro[ray * 3 + 0] = o[0]; out.ro[ray * 3 + 1] = o[1]; out.ro[ray * 3 + 2] = o[2]; }
    if (out.rd) { out.rd[ray * 3 + 0] = d[0]; out.rd[ray * 3 + 1] = d[1]; out.rd[ray * 3 + 2] = d[2]; }
    if (out.near_raw) {
      float near = 0.0f, far = 0.0f;
      const bool hit = slab_test(o, d, out.scene_range, near, far);
      // second, inflated test: lets the fused renderer skip rays that provably never enter the cube
      float n2, f2;
      bool hit_wide = slab_test(o, d, out.scene_range * 1.0001f, n2, f2);
      out.near_raw[ray] = near;
      out.far_raw[ray] = far;
      out.hit[ray] = (uint8_t)((hit ? 1 : 0) | (hit_wide ? 2 : 0));
      if (hit) {
        const uint32_t a = ~ordered_key(near), c = ordered_key(far);   // maximise the complement of the near key
        kmin = a > kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
        ++cnt;
      }
    }
  }
  if (out.near_raw) block_reduce_keys(kmin, kmax, cnt, out.reduce);
}

__global__ __launch_bounds__(256) void slab_kernel(const float* __restrict__ ro, const float* __restrict__ rd, int64_t n,
                                                   float scene_range, float* __restrict__ near_raw,
                                                   float* __restrict__ far_raw, uint8_t* __restrict__ hit_out,
                                                   uint32_t* reduce) {
  uint32_t kmin = 0u, kmax = 0u, cnt = 0u;
  for (int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ray < n; ray += (int64_t)gridDim.x * blockDim.x) {
    float o[3] = {ro[ray * 3], ro[ray * 3 + 1], ro[ray * 3 + 2]};
    float d[3] = {rd[ray * 3], rd[ray * 3 + 1], rd[ray * 3 + 2]};
    float near = 0.0f, far = 0.0f;
    const bool hit = slab_test(o, d, scene_range, near, far);
    float n2, f2;
    bool hit_wide = slab_test(o, d, scene_range * 1.0001f, n2, f2);
    near_raw[ray] = near;
    far_raw[ray] = far;
    hit_out[ray] = (uint8_t)((hit ? 1 : 0) | (hit_wide ? 2 : 0));
    if (hit) {
      const uint32_t a = ~ordered_key(near), c = ordered_key(far);
      kmin = a > kmin ? a : kmin;
      kmax = c > kmax ? c : kmax;
      ++cnt;
    }
  }
  block_reduce_keys(kmin, kmax, cnt, reduce);
}

__global__ __launch_bounds__(256) void finish_planes_kernel(const float* __restrict__ near_raw,
                                                            const float* __restrict__ far_raw,
                                                            const uint8_t* __restrict__ hit, const uint32_t* __restrict__ reduce,
                                                            int64_t n, float* __restrict__ near_plane,
                                                            float* __restrict__ far_plane) {
  const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n) return;
  float fill_near = ordered_key_inv(~reduce[0]), fill_far = ordered_key_inv(reduce[1]);
  float a = near_raw[ray], b = far_raw[ray];
  finish_planes((hit[ray] & 1) != 0, fill_near, fill_far, a, b);
  near_plane[ray] = a;
  far_plane[ray] = b;
}

extern "C" int nfi_raygen(const nfi_raygen_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->cam2world && a->ray_origins && a->ray_directions, "raygen: null pointer");
  REQUIRE(a->n_scenes > 0 && a->height > 0 && a->width > 0, "raygen: bad shape");
  CameraParams cam{a->cam2world, a->focal, a->bbox, a->focal ? a->center : nullptr, a->height, a->width, a->normalize, a->height, 0};
  RaygenOut out{a->ray_origins, a->ray_directions, nullptr, nullptr, nullptr, nullptr, 0.0f};
  int64_t n = (int64_t)a->n_scenes * a->height * a->width;
  hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, kRayBlocks)), dim3(256), 0, (hipStream_t)stream, cam,
                     a->n_scenes, out);
  return check_launch("raygen");
}

extern "C" int nfi_near_far(const nfi_near_far_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->ray_origins && a->ray_directions && a->near_raw && a->far_raw && a->hit && a->reduce,
          "near_far: null pointer");
  REQUIRE(a->n_rays > 0, "near_far: no rays");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(a->reduce, 0, 16, s) != hipSuccess) return fail(NFI_ERR_LAUNCH, "near_far: memset failed");
  unsigned blocks = (unsigned)((a->n_rays + 255) / 256);
  hipLaunchKernelGGL(slab_kernel, dim3(std::min<unsigned>(blocks, kRayBlocks)), dim3(256), 0, s, a->ray_origins, a->ray_directions, a->n_rays,
                     a->scene_range, a->near_raw, a->far_raw, a->hit, a->reduce);
  if (a->near_plane && a->far_plane)
    hipLaunchKernelGGL(finish_planes_kernel, dim3(blocks), dim3(256), 0, s, a->near_raw, a->far_raw, a->hit, a->reduce,
                       a->n_rays, a->near_plane, a->far_plane);
  return check_launch("near_far");
}

// ------------------------------------------------------------------------------------------------
// stratified depths + query points (lib/nerf_utils.py:94-120)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float stratified_depth(float near, float far, int k, int S, float noise, bool jitter) {
  float t = aten_lerp(near, far, (float)k / (float)S);
  if (jitter) t = t + noise * ((far - near) / (float)S);
  return t;
}

__global__ __launch_bounds__(256) void stratified_kernel(nfi_stratified_args a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = a.n_rays * a.n_samples;
  if (idx >= total) return;
  const int64_t ray = idx / a.n_samples;
  const int k = (int)(idx - ray * a.n_samples);
  float near = a.near_plane[ray], far = a.far_plane[ray];
  float t = stratified_depth(near, far, k, a.n_samples, a.noise ? a.noise[idx] : 0.0f, a.noise != nullptr);
  a.depth[idx] = t;
  if (a.points) {
#pragma unroll
    for (int c = 0; c < 3; ++c) a.points[idx * 3 + c] = a.ray_origins[ray * 3 + c] + a.ray_directions[ray * 3 + c] * t;
  }
}

__global__ __launch_bounds__(256) void points_on_rays_kernel(const float* __restrict__ ro, const float* __restrict__ rd,
                                                             const float* __restrict__ depth, int64_t total, int S,
                                                             float* __restrict__ points) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t ray = idx / S;
  const float t = depth[idx];
#pragma unroll
  for (int c = 0; c < 3; ++c) points[idx * 3 + c] = ro[ray * 3 + c] + rd[ray * 3 + c] * t;
}

extern "C" int nfi_points_on_rays(const float* ray_origins, const float* ray_directions, const float* depth,
                                  int64_t n_rays, int n_samples, float* points, nfi_stream_t stream) {
  REQUIRE(ray_origins && ray_directions && depth && points, "points_on_rays: null pointer");
  REQUIRE(n_rays > 0 && n_samples > 0, "points_on_rays: bad shape");
  int64_t total = n_rays * n_samples;
  hipLaunchKernelGGL(points_on_rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     ray_origins, ray_directions, depth, total, n_samples, points);
  return check_launch("points_on_rays");
}

extern "C" int nfi_stratified_points(const nfi_stratified_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->ray_origins && a->ray_directions && a->near_plane && a->far_plane && a->depth, "stratified: null pointer");
  REQUIRE(a->n_rays > 0 && a->n_samples > 0, "stratified: bad shape");
  int64_t total = a->n_rays * a->n_samples;
  hipLaunchKernelGGL(stratified_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
  return check_launch("stratified_points");
}

// ------------------------------------------------------------------------------------------------
// field query (sampler closure)
// ------------------------------------------------------------------------------------------------
struct FieldKernelParams {
  const float* points; int64_t P;
  const void* texels; int res; int tex;
  const float* image; int A; const float* att;
  int use_sdf; const float* beta; const float* alpha; float scene_range;
  float* sigma; float* rgb; float* sdf; float* sem; uint8_t* outside;
  const float* xray; int spr;
  int layout;
};

// PREC: 0 = exact fp32 MFMA (the sampler closure's default: bitwise an fmaf chain), 1 = split-fp16 operands as in the
// fused renderer (render()'s differentiable path: the same decoder arithmetic in both of its paths, and what the
// backward kernel recomputes)
template <int TEX, bool ATT, bool VD = false, int PREC = 0>
__global__ __launch_bounds__(256) void field_query_kernel(FieldKernelParams k) {
  constexpr int kImg = VD ? kVdImageFloats : kLdsImageFloats;
  __shared__ __attribute__((aligned(16))) float lds[kImg + 64];
  __shared__ __attribute__((aligned(16))) float stages[4][16 * 36];
  const int scene = blockIdx.y;
  stage_field_lds(lds, k.image, k.att ? k.att + (size_t)scene * k.A * 3 : nullptr, k.A, kImg);
  if (PREC == 1) {
    // fp16 fragments overlay the fp32 fragment area; biases stay where they are
    __syncthreads();
    for (int i = threadIdx.x; i < kB1F; i += blockDim.x) lds[i] = k.image[kW1H + i];
  }
  __syncthreads();
  const size_t tb = TEX == 0 ? 128 : 64;
  const char* tex_scene = reinterpret_cast<const char*>(k.texels) + (size_t)scene * 3 * k.res * k.res * tb;
  FieldParams P = make_field_params(tex_scene, k.res, TEX, k.A, k.use_sdf, k.beta, k.alpha, lds, kImg, k.layout);
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_chunks = (k.P + 63) / 64;
  const float* xray_scene = VD ? k.xray + (size_t)scene * (size_t)(k.P / k.spr) * kRayFeatPad : nullptr;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < n_chunks; chunk += (int64_t)gridDim.x * 4) {
    int64_t p = chunk * 64 + lane;
    bool valid = p < k.P;
    size_t gi = (size_t)scene * k.P + (valid ? p : 0);
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (valid) { px = k.points[gi * 3]; py = k.points[gi * 3 + 1]; pz = k.points[gi * 3 + 2]; }
    bool out;
    float* sem = k.sem ? k.sem + ((size_t)scene * k.P + chunk * 64) * k.A : nullptr;
    SampleOut so = field_wave<TEX, ATT, false, PREC, VD>(P, k.scene_range, lane, px, py, pz, valid, sem, &out, stages[wave],
                                                      nullptr, xray_scene, VD && valid ? (int)(p / k.spr) : 0);
    if (valid) {
      k.sigma[gi] = so.sigma;
      k.rgb[gi * 3] = so.r; k.rgb[gi * 3 + 1] = so.g; k.rgb[gi * 3 + 2] = so.b;
      if (k.sdf) k.sdf[gi] = so.sdf;
      if (k.outside) k.outside[gi] = out ? 1 : 0;
    }
  }
}

extern "C" int nfi_field_query_fwd(const nfi_field_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->points && a->sigma && a->rgb, "field_query: null pointer");
  REQUIRE(a->n_scenes > 0 && a->points_per_scene > 0, "field_query: bad shape");
  int rc = check_field_common(a->texels, a->plane_res, a->texel_dtype, a->decoder_image, a->n_attention,
                               a->attention_values, a->use_sdf, a->beta, a->alpha, a->texel_layout);
  if (rc) return rc;
  REQUIRE(!a->semantics || a->n_attention > 0, "field_query: semantics need attention_values > 0");
  // (semantics rows are stored per valid point only: field_wave guards them with the lane's valid flag)
  FieldKernelParams k{a->points, a->points_per_scene, a->texels, a->plane_res, a->texel_dtype, a->decoder_image,
                      a->n_attention, a->attention_values, a->use_sdf, a->beta, a->alpha, a->scene_range,
                      a->sigma, a->rgb, a->sdf, a->semantics, a->outside, a->ray_features, a->samples_per_ray, a->texel_layout};
  REQUIRE(!a->ray_features || (a->samples_per_ray > 0 && a->points_per_scene % a->samples_per_ray == 0),
          "field_query: with ray_features, points_per_scene must be a multiple of samples_per_ray");
  REQUIRE(a->mlp_precision == 0 || (a->mlp_precision == 1 && !a->ray_features),
          "field_query: mlp_precision must be 0 (exact fp32) or 1 (split fp16; not with the view-direction decoder)");
  int64_t chunks = (a->points_per_scene + 63) / 64;
  int64_t blocks = (chunks + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks, (unsigned)a->n_scenes);
  hipStream_t s = (hipStream_t)stream;
  bool att = a->n_attention > 0;
#define NFI_LAUNCH_FIELD(TEX)                                                                          \
  do {                                                                                                \
    if (a->ray_features) {                                                                            \
      if (att) hipLaunchKernelGGL((field_query_kernel<TEX, true, true>), grid, dim3(256), 0, s, k);   \
      else hipLaunchKernelGGL((field_query_kernel<TEX, false, true>), grid, dim3(256), 0, s, k);      \
    } else if (a->mlp_precision == 1) {                                                               \
      if (att) hipLaunchKernelGGL((field_query_kernel<TEX, true, false, 1>), grid, dim3(256), 0, s, k);  \
      else hipLaunchKernelGGL((field_query_kernel<TEX, false, false, 1>), grid, dim3(256), 0, s, k);     \
    } else {                                                                                          \
      if (att) hipLaunchKernelGGL((field_query_kernel<TEX, true>), grid, dim3(256), 0, s, k);         \
      else hipLaunchKernelGGL((field_query_kernel<TEX, false>), grid, dim3(256), 0, s, k);            \
    }                                                                                                 \
  } while (0)
  if (a->texel_dtype == NFI_TEXEL_F32) NFI_LAUNCH_FIELD(0);
  else if (a->texel_dtype == NFI_TEXEL_BF16) NFI_LAUNCH_FIELD(1);
  else NFI_LAUNCH_FIELD(2);
#undef NFI_LAUNCH_FIELD
  return check_launch("field_query_fwd");
}

__global__ __launch_bounds__(256) void bbox_overlay_kernel(const float* __restrict__ points, int64_t n, float scene_range,
                                                           float thr, const float* __restrict__ sigma_in,
                                                           float* __restrict__ sigma_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = points[i * 3], y = points[i * 3 + 1], z = points[i * 3 + 2];
  const bool outside = (fabsf(x / scene_range) > 1.0f) || (fabsf(y / scene_range) > 1.0f) || (fabsf(z / scene_range) > 1.0f);
  const bool ix = fabsf(x) < thr, iy = fabsf(y) < thr, iz = fabsf(z) < thr;
  // generator.py:650-658: product of (1 - both-inside) over the axis pairs xy, xz, yz (the last one twice)
  float m = 1.0f;
  m *= 1.0f - ((ix && iy) ? 1.0f : 0.0f);
  m *= 1.0f - ((ix && iz) ? 1.0f : 0.0f);
  m *= 1.0f - ((iy && iz) ? 1.0f : 0.0f);
  m *= 1.0f - ((iy && iz) ? 1.0f : 0.0f);
  m *= 1.0f - (outside ? 1.0f : 0.0f);
  sigma_out[i] = sigma_in[i] + 100.0f * m;
}

extern "C" int nfi_bbox_overlay(const float* points, int64_t n_points, float scene_range, float threshold,
                                const float* sigma_in, float* sigma_out, nfi_stream_t stream) {
  REQUIRE(points && sigma_in && sigma_out && n_points > 0, "bbox_overlay: null pointer / empty");
  hipLaunchKernelGGL(bbox_overlay_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     points, n_points, scene_range, threshold, sigma_in, sigma_out);
  return check_launch("bbox_overlay");
}

// ------------------------------------------------------------------------------------------------
// per-wave LDS slab used by the sampling / compositing stages
// ------------------------------------------------------------------------------------------------
template <int N>
struct __attribute__((aligned(16))) WaveSlabT {
  static constexpr int kN = N;
  float cdf[N];
  float bins[N];
  uint32_t key[N];
  float srt[5][N];   // depth, sigma, r, g, b in merged order
  float piv[N / 8];  // every eighth cdf entry (cdf[8p + 7], +inf past the end): build_cdf / invert_cdf
};
using WaveSlab = WaveSlabT<128>;       // up to 64 + 64 samples per ray
using WaveSlabWide = WaveSlabT<256>;   // up to 128 + 128

// ---- inverse CDF on a ray held one element per (slot, lane) -----------------------------------
// bins[e] e<M, weights[e] e<M-1 (already padded with 0 past M-1).  Writes cdf/bins to the slab and
// returns, for each of this lane's u values, the sample and the searchsorted index.
template <int SPL, class Slab>
__device__ __forceinline__ void build_cdf(Slab& slab, const float (&bins)[SPL], const float (&wts)[SPL], int M,
                                          int lane) {
  float q[SPL];
  float tot = 0.0f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    int e = j * 64 + lane;
    q[j] = (e < M - 1) ? (wts[j] + 1e-5f) : 0.0f;
    tot += q[j];
  }
  tot = wave_sum(tot);
  float pdf[SPL], inc[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) pdf[j] = (j * 64 + lane < M - 1) ? q[j] / tot : 0.0f;
  incl_cumsum<SPL>(pdf, inc, lane);
  // cdf[0] = 0, cdf[e+1] = inclusive sum through e
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    int e = j * 64 + lane;
    if (e < M - 1) slab.cdf[e + 1] = inc[j];
    if (e < M) slab.bins[e] = bins[j];
    // pivot table for the two-level search of invert_cdf: entry p = cdf[8p + 7] (+inf past the end)
    if (((e + 1) & 7) == 7) slab.piv[(e + 1) >> 3] = (e + 1 < M) ? inc[j] : INFINITY;
  }
  if (lane == 0) slab.cdf[0] = 0.0f;
  wave_lds_fence();
}

// NP: pivots to look at = ceil(largest M / 8) of the caller (a multiple of 4; build_cdf<SPL> writes 8 SPL of them)
template <int NP, class Slab>
__device__ __forceinline__ float invert_cdf(const Slab& slab, int M, float u, int& ind) {
  // searchsorted(cdf, u, right=True) = #{cdf <= u} over the ascending cdf, as an 8-way two-level search: the pivots
  // (every eighth entry: two or four 16-byte broadcast reads) give the block, the block's eight entries the position -
  // two dependent LDS round trips instead of the seven or eight of a binary search
  {
    const f32x4* pv = reinterpret_cast<const f32x4*>(slab.piv);
    int b = 0;
#pragma unroll
    for (int i = 0; i < NP / 4; ++i) {
      const f32x4 p = pv[i];
      b += (p.x <= u ? 1 : 0) + (p.y <= u ? 1 : 0) + (p.z <= u ? 1 : 0) + (p.w <= u ? 1 : 0);
    }
    const f32x4* cv = reinterpret_cast<const f32x4*>(slab.cdf + 8 * b);
    const f32x4 c0 = cv[0], c1 = cv[1];
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) cnt += (8 * b + e < M && c[e] <= u) ? 1 : 0;
    ind = 8 * b + cnt;
  }
  int lo = ind - 1 < 0 ? 0 : ind - 1;
  int hi = ind > M - 1 ? M - 1 : ind;
  float c_lo = slab.cdf[lo], c_hi = slab.cdf[hi];
  float b_lo = slab.bins[lo], b_hi = slab.bins[hi];
  float den = c_hi - c_lo;
  den = (den < 1e-5f) ? 1.0f : den;
  float fr = (u - c_lo) / den;
  return b_lo + fr * (b_hi - b_lo);
}

// EG3D smoothing of S weights held one per lane (run.py:264-272); lanes >= S return garbage
__device__ __forceinline__ float smooth_weights(float w, int S, int lane) {
  float wp = lane_prev(w, -INFINITY);
  float wn = lane_next(w, -INFINITY);
  if (lane >= S - 1) wn = -INFINITY;
  float m0 = fmaxf(wp, w);   // max(w[k-1], w[k])
  float m1 = fmaxf(w, wn);   // max(w[k], w[k+1])
  return (m0 + m1) / 2.0f + 0.01f;
}

// hierarchical resampling for S <= 64: coarse weights -> smooth -> pdf over smooth[1..S-2] on the
// S-1 bin mid-points -> S fine depths.  Returns the fine depth for this lane's u.
struct ResampleTaps { float w, smooth, T; int ind; };
__device__ __forceinline__ float resample_ray(WaveSlab& slab, float sigma, float t, int S, float dnorm, float u, int lane,
                                              ResampleTaps* taps) {
  float sg[1] = {lane < S ? sigma : 0.0f}, tt[1] = {t}, w[1], T[1];
  ray_weights<1>(sg, tt, S, dnorm, lane, w, T);
  float sm = smooth_weights(w[0], S, lane);
  float tn = lane_next(t, 0.0f);
  float mid[1] = {0.5f * (tn + t)};                       // bins e = 0..S-2
  float wts[1] = {lane_next(sm, 0.0f)};                    // weights e = 0..S-3  <- smooth[e+1]
  build_cdf<1>(slab, mid, wts, S - 1, lane);
  int ind;
  float z = invert_cdf<8>(slab, S - 1, u, ind);
  if (taps) { taps->w = w[0]; taps->smooth = sm; taps->ind = ind; taps->T = T[0]; }
  return z;
}

// The same for 64 < S <= 128 (two elements per lane, element e = slot*64 + lane).
template <int SP>
__device__ __forceinline__ void smooth_weights_wide(const float (&w)[SP], int S, int lane, float (&sm)[SP]) {
#pragma unroll
  for (int j = 0; j < SP; ++j) {
    const float prev_edge = j > 0 ? bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(w[j > 0 ? j - 1 : 0]), 63)) : -INFINITY;
    const float next_edge = j + 1 < SP ? bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(w[j + 1 < SP ? j + 1 : j]), 0)) : -INFINITY;
    const float wp = lane_prev(w[j], prev_edge);
    float wn = lane_next(w[j], next_edge);
    if (j * 64 + lane >= S - 1) wn = -INFINITY;
    const float m0 = fmaxf(wp, w[j]), m1 = fmaxf(w[j], wn);
    sm[j] = (m0 + m1) / 2.0f + 0.01f;
  }
}

template <int SP, class Slab>
__device__ __forceinline__ void resample_ray_wide(Slab& slab, const float (&sigma)[SP], const float (&t)[SP], int S, float dnorm,
                                                  const float (&u)[SP], int lane, float (&z)[SP], float (&w)[SP],
                                                  float (&sm)[SP], int (&ind)[SP], float (&T)[SP]) {
  float sg[SP];
#pragma unroll
  for (int j = 0; j < SP; ++j) sg[j] = (j * 64 + lane < S) ? sigma[j] : 0.0f;
  ray_weights<SP>(sg, t, S, dnorm, lane, w, T);
  smooth_weights_wide<SP>(w, S, lane, sm);
  float tn[SP], mid[SP], wts[SP];
  next_elem<SP>(t, tn, lane);
  next_elem<SP>(sm, wts, lane);                               // weights e = 0..S-3  <- smooth[e+1]
#pragma unroll
  for (int j = 0; j < SP; ++j) mid[j] = 0.5f * (tn[j] + t[j]);  // bins e = 0..S-2
  build_cdf<SP>(slab, mid, wts, S - 1, lane);
#pragma unroll
  for (int j = 0; j < SP; ++j) z[j] = invert_cdf<8 * SP>(slab, S - 1, u[j], ind[j]);
}

// ---- merge + composite ---------------------------------------------------------------------------
// Every lane owns up to 2 input samples (element e = slot*64+lane of cat(a,b)); computes the
// stable ascending rank of each key among the n keys, scatters (depth, sigma, rgb) to the slab in
// merged order.  rank_out: merged position of each of this lane's elements.
template <int NS, class Slab>
__device__ __forceinline__ void merge_scatter(Slab& slab, const float (&dep)[NS], const float (&sig)[NS],
                                              const float (&cr)[NS], const float (&cg)[NS], const float (&cb)[NS],
                                              const int (&eidx)[NS], int n, int (&rank_out)[NS]) {
  // eidx[j]: index of this lane's j-th element in cat(a, b) (>= n: no element)
  uint32_t key[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    key[j] = ordered_key(dep[j]);
    if (eidx[j] < n) slab.key[eidx[j]] = key[j];
  }
  wave_lds_fence();
  int rank[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) rank[j] = 0;
  const uint4* kv = reinterpret_cast<const uint4*>(slab.key);
  const int n4 = (n + 3) >> 2;
  for (int i = 0; i < n4; ++i) {
    uint4 q = kv[i];
    uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int idx = 4 * i + c;
      bool in = idx < n;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        bool before = (qq[c] < key[j]) || (qq[c] == key[j] && idx < eidx[j]);
        rank[j] += (in && before) ? 1 : 0;
      }
    }
  }
  wave_lds_fence();
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    rank_out[j] = rank[j];
    if (eidx[j] < n) {
      int r = rank[j];
      slab.srt[0][r] = dep[j]; slab.srt[1][r] = sig[j];
      slab.srt[2][r] = cr[j]; slab.srt[3][r] = cg[j]; slab.srt[4][r] = cb[j];
    }
  }
  wave_lds_fence();
}

// Merge for the fused renderer, lane = sample index k < S: the coarse list is (checked to be)
// ascending, so only the S fine keys have to be ranked by counting:
//   rank(coarse k) = k + #{fine j : z_j <  t_k}
//   rank(fine  k)  = #{coarse j : t_j <= z_k} + #{fine j : z_j < z_k or (z_j == z_k and j < k)}
// i.e. the stable ascending order of cat(coarse, fine) (coarse first on ties), ~4x fewer compares
// than ranking all 2S keys against all 2S keys.  Falls back to the general count when the coarse
// depths are not ascending (possible only through 1-ulp rounding of the jittered depths).
struct MergeIn { float t, sigma, r, g, b; };
__device__ __forceinline__ void merge_pair_scatter(WaveSlab& slab, const MergeIn& c, const MergeIn& f, int S, int lane,
                                                   int& rank_c, int& rank_f) {
  const bool valid = lane < S;
  const uint32_t kc = valid ? ordered_key(c.t) : 0xFFFFFFFFu;
  const uint32_t kf = valid ? ordered_key(f.t) : 0xFFFFFFFFu;
  const uint32_t kc_next = f2bits(lane_next(bits2f(kc), bits2f(0xFFFFFFFFu)));
  const bool ascending = __all(lane >= S - 1 || kc <= kc_next);
  slab.key[lane] = kf;            // fine keys   [0,64)   (padding 0xFFFFFFFF ranks after everything)
  slab.key[64 + lane] = kc;       // coarse keys [64,128)
  uint32_t* kpiv = reinterpret_cast<uint32_t*>(slab.piv);          // every eighth coarse key (the pivot row is free after the resampling)
  if ((lane & 7) == 7) kpiv[lane >> 3] = kc;
  wave_lds_fence();
  const uint4* kv = reinterpret_cast<const uint4*>(slab.key);
  const int n4 = (S + 3) >> 2;
  int cnt_a = 0, cnt_b = 0, cnt_c = 0;
  if (ascending) {
    // #{fine < z_k} against all fine keys; the coarse side needs no second comparison per key: with the coarse keys
    // ascending, a fine key f is below coarse key k exactly when #{coarse <= f} <= k, so #{fine < t_k} is the running sum
    // over the histogram of the fine keys' upper bounds (one LDS add per lane + one wave scan instead of 64 compares)
    // (all 16 quads, whatever S: the padding keys 0xFFFFFFFF never count, and a fixed trip count lets the 16 LDS reads
    //  go out together instead of one LDS latency per turn)
    uint4 qv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qv[i] = kv[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t qq[4] = {qv[i].x, qv[i].y, qv[i].z, qv[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) cnt_b += (qq[e] < kf) ? 1 : 0;
    }
    // #{coarse <= z_k} over the ascending coarse keys: 8-way two-level search (pivots, then the block's eight keys), two
    // dependent LDS round trips instead of a binary search's seven
    {
      const uint4* pv = reinterpret_cast<const uint4*>(kpiv);
      const uint4 p0 = pv[0], p1 = pv[1];
      const int b = (p0.x <= kf ? 1 : 0) + (p0.y <= kf ? 1 : 0) + (p0.z <= kf ? 1 : 0) + (p0.w <= kf ? 1 : 0) +
                    (p1.x <= kf ? 1 : 0) + (p1.y <= kf ? 1 : 0) + (p1.z <= kf ? 1 : 0) + (p1.w <= kf ? 1 : 0);
      const uint4* cv = reinterpret_cast<const uint4*>(slab.key + 64 + 8 * b);
      const uint4 c0 = cv[0], c1 = cv[1];
      const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      int cnt = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) cnt += (8 * b + e < 64 && c[e] <= kf) ? 1 : 0;
      cnt_c = 8 * b + cnt;
    }
    uint32_t* hist = reinterpret_cast<uint32_t*>(slab.bins);      // (free after the resampling) entries 0 .. S
    hist[lane] = 0u;
    if (lane == 0) hist[64] = 0u;
    wave_lds_fence();
    if (valid) atomicAdd(&hist[cnt_c], 1u);
    wave_lds_fence();
    cnt_a = (int)wave_incl_scan_add_f32((float)hist[lane]);
  } else {
    for (int i = 0; i < n4; ++i) {
      const uint4 q = kv[i];
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cnt_a += (qq[e] < kc) ? 1 : 0;
        cnt_b += (qq[e] < kf) ? 1 : 0;
      }
    }
  }
  // fine depths are almost never equal.  Equal keys have equal counts of smaller keys and distinct keys distinct counts,
  // so a tie shows as two lanes claiming the same slot of a scratch row (the cdf row is free after the resampling):
  // one LDS write + read instead of a third comparison per key.  Only then the count is redone with the stable
  // tie-break of torch.sort (earlier index first)
  uint32_t* claim = reinterpret_cast<uint32_t*>(slab.cdf);
  if (valid) claim[cnt_b] = (uint32_t)lane;
  wave_lds_fence();
  const bool lost = valid && claim[cnt_b] != (uint32_t)lane;
  wave_lds_fence();
  if (__any(lost)) {
    cnt_b = 0;
    for (int i = 0; i < n4; ++i) {
      const uint4 q = kv[i];
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = 4 * i + e;
        cnt_b += ((qq[e] < kf) || (qq[e] == kf && idx < lane)) ? 1 : 0;
      }
    }
  }
  if (ascending) {
    rank_c = lane + cnt_a;
  } else {
    int cnt_d = 0;   // coarse j before coarse k
    for (int i = 0; i < n4; ++i) {
      const uint4 q = kv[16 + i];
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = 4 * i + e;
        cnt_c += (idx < S && qq[e] <= kf) ? 1 : 0;
        cnt_d += (idx < S && (qq[e] < kc || (qq[e] == kc && idx < lane))) ? 1 : 0;
      }
    }
    rank_c = cnt_d + cnt_a;
  }
  rank_f = cnt_c + cnt_b;
  wave_lds_fence();
  if (valid) {
    slab.srt[0][rank_c] = c.t; slab.srt[1][rank_c] = c.sigma; slab.srt[2][rank_c] = c.r; slab.srt[3][rank_c] = c.g; slab.srt[4][rank_c] = c.b;
    slab.srt[0][rank_f] = f.t; slab.srt[1][rank_f] = f.sigma; slab.srt[2][rank_f] = f.r; slab.srt[3][rank_f] = f.g; slab.srt[4][rank_f] = f.b;
  }
  wave_lds_fence();
}

// The same for two elements per lane and list (64 < S <= 128, element e = slot*64 + lane).  Returns false - nothing
// written - when the coarse depths are not ascending; the caller then uses the general merge_scatter.
template <class Slab>
__device__ __forceinline__ bool merge_pair_scatter_wide(Slab& slab, const float (&dep)[4], const float (&sig)[4],
                                                        const float (&cr)[4], const float (&cg)[4], const float (&cb)[4],
                                                        int S, int lane, int (&rank)[4]) {
  // dep[0..1] coarse slots, dep[2..3] fine slots
  uint32_t kc[2], kf[2];
  bool val[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    val[j] = j * 64 + lane < S;
    kc[j] = val[j] ? ordered_key(dep[j]) : 0xFFFFFFFFu;
    kf[j] = val[j] ? ordered_key(dep[2 + j]) : 0xFFFFFFFFu;
  }
  // ascending check of the coarse list across the two slots
  const uint32_t first1 = (uint32_t)__builtin_amdgcn_readlane((int)kc[1], 0);
  const uint32_t n0 = f2bits(lane_next(bits2f(kc[0]), bits2f(first1)));
  const uint32_t n1 = f2bits(lane_next(bits2f(kc[1]), bits2f(0xFFFFFFFFu)));
  const bool ok0 = (lane >= S - 1) || kc[0] <= n0, ok1 = (64 + lane >= S - 1) || kc[1] <= n1;
  if (!__all(ok0 && ok1)) return false;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    slab.key[j * 64 + lane] = kf[j];           // fine keys   [0,128)
    slab.key[128 + j * 64 + lane] = kc[j];     // coarse keys [128,256)
  }
  uint32_t* kpiv = reinterpret_cast<uint32_t*>(slab.piv);          // every eighth coarse key (16 pivots)
#pragma unroll
  for (int j = 0; j < 2; ++j) if ((lane & 7) == 7) kpiv[(j * 64 + lane) >> 3] = kc[j];
  wave_lds_fence();
  const uint4* kv = reinterpret_cast<const uint4*>(slab.key);
  const int n4 = (S + 3) >> 2;
  int cnt_a[2] = {0, 0}, cnt_b[2] = {0, 0}, cnt_c[2];
  // all 32 quads of the fine keys in two batches of 16 LDS reads (padding keys never count; see merge_pair_scatter)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint4 qv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qv[i] = kv[16 * h + i];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t qq[4] = {qv[i].x, qv[i].y, qv[i].z, qv[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          cnt_b[j] += (qq[e] < kf[j]) ? 1 : 0;
        }
    }
  }
  {
    // #{coarse <= z} over the ascending coarse keys: two-level 8-way search (merge_pair_scatter)
    const uint4* pv = reinterpret_cast<const uint4*>(kpiv);
    const uint4 pq[4] = {pv[0], pv[1], pv[2], pv[3]};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int b = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        b += (pq[i].x <= kf[j] ? 1 : 0) + (pq[i].y <= kf[j] ? 1 : 0) + (pq[i].z <= kf[j] ? 1 : 0) + (pq[i].w <= kf[j] ? 1 : 0);
      const uint4* cv = reinterpret_cast<const uint4*>(slab.key + 128 + 8 * b);
      const uint4 c0 = cv[0], c1 = cv[1];
      const uint32_t c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      int cnt = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) cnt += (8 * b + e < 128 && c[e] <= kf[j]) ? 1 : 0;
      cnt_c[j] = 8 * b + cnt;
    }
  }
  {
    // #{fine < t_e} for the ascending coarse keys = running sum over the histogram of the fine keys' upper bounds
    // (merge_pair_scatter): entries 0 .. S of the bins row, element e = slot * 64 + lane
    uint32_t* hist = reinterpret_cast<uint32_t*>(slab.bins);
    hist[lane] = 0u; hist[64 + lane] = 0u;
    if (lane == 0) hist[128] = 0u;
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 2; ++j) if (val[j]) atomicAdd(&hist[cnt_c[j]], 1u);
    wave_lds_fence();
    const float s0 = wave_incl_scan_add_f32((float)hist[lane]);
    const float t0 = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(s0), 63));
    const float s1 = t0 + wave_incl_scan_add_f32((float)hist[64 + lane]);
    cnt_a[0] = (int)s0; cnt_a[1] = (int)s1;
  }
  // ties among the fine keys = two elements claiming the same slot (see merge_pair_scatter)
  uint32_t* claim = reinterpret_cast<uint32_t*>(slab.cdf);
#pragma unroll
  for (int j = 0; j < 2; ++j) if (val[j]) claim[cnt_b[j]] = (uint32_t)(j * 64 + lane);
  wave_lds_fence();
  bool lost = false;
#pragma unroll
  for (int j = 0; j < 2; ++j) lost = lost || (val[j] && claim[cnt_b[j]] != (uint32_t)(j * 64 + lane));
  wave_lds_fence();
  if (__any(lost)) {
    cnt_b[0] = cnt_b[1] = 0;            // equal fine depths exist: stable tie-break (earlier index first)
    for (int i = 0; i < n4; ++i) {
      const uint4 q = kv[i];
      const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int idx = 4 * i + e;
          cnt_b[j] += ((qq[e] < kf[j]) || (qq[e] == kf[j] && idx < j * 64 + lane)) ? 1 : 0;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    rank[j] = j * 64 + lane + cnt_a[j];
    rank[2 + j] = cnt_c[j] + cnt_b[j];
  }
  wave_lds_fence();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (val[j]) {
      const int rc = rank[j], rf = rank[2 + j];
      slab.srt[0][rc] = dep[j]; slab.srt[1][rc] = sig[j]; slab.srt[2][rc] = cr[j]; slab.srt[3][rc] = cg[j]; slab.srt[4][rc] = cb[j];
      slab.srt[0][rf] = dep[2 + j]; slab.srt[1][rf] = sig[2 + j]; slab.srt[2][rf] = cr[2 + j]; slab.srt[3][rf] = cg[2 + j]; slab.srt[4][rf] = cb[2 + j];
    }
  }
  wave_lds_fence();
  return true;
}

struct CompositeOut { float r, g, b, depth, mask; };

// composite the n samples sitting in merged order in the slab (lib/nerf_utils.py:123-161)
template <int NS, class Slab>
__device__ __forceinline__ CompositeOut composite_slab(const Slab& slab, int n, float dnorm, int white, int lane,
                                                       float (&w_out)[NS]) {
  float dep[NS], sig[NS], w[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int e = j * 64 + lane;
    dep[j] = (e < n) ? slab.srt[0][e] : 0.0f;
    sig[j] = (e < n) ? slab.srt[1][e] : 0.0f;
  }
  ray_weights<NS>(sig, dep, n, dnorm, lane, w);
  float r = 0.0f, g = 0.0f, b = 0.0f, d = 0.0f, m = 0.0f;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int e = j * 64 + lane;
    if (e < n) {
      r += w[j] * slab.srt[2][e]; g += w[j] * slab.srt[3][e]; b += w[j] * slab.srt[4][e];
      d += w[j] * dep[j]; m += w[j];
    }
    w_out[j] = w[j];
  }
  CompositeOut o;
  o.r = wave_sum(r); o.g = wave_sum(g); o.b = wave_sum(b); o.depth = wave_sum(d); o.mask = wave_sum(m);
  if (white) { float bg = 1.0f - o.mask; o.r += bg; o.g += bg; o.b += bg; }
  return o;
}

// ------------------------------------------------------------------------------------------------
// stand-alone stage kernels (one wave per ray)
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void ray_weights_kernel(nfi_weights_args a) {
  const int lane = lane_id();
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.n_rays) return;
  const int S = a.n_samples;
  float sig[NS], t[NS], w[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int e = j * 64 + lane;
    sig[j] = e < S ? a.sigma[ray * S + e] : 0.0f;
    t[j] = e < S ? a.depth[ray * S + e] : 0.0f;
  }
  float dn = norm3(a.ray_directions[ray * 3], a.ray_directions[ray * 3 + 1], a.ray_directions[ray * 3 + 2]);
  ray_weights<NS>(sig, t, S, dn, lane, w);
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int e = j * 64 + lane;
    if (e < S) a.weights[ray * S + e] = w[j];
  }
}

extern "C" int nfi_ray_weights(const nfi_weights_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->sigma && a->ray_directions && a->depth && a->weights, "ray_weights: null pointer");
  REQUIRE(a->n_rays > 0 && a->n_samples > 0 && a->n_samples <= NFI_MAX_SAMPLES_SINGLE_PASS, "ray_weights: n_samples must be in [1,512]");
  const dim3 grid((unsigned)((a->n_rays + 3) / 4));
  if (a->n_samples <= 128) hipLaunchKernelGGL(ray_weights_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(ray_weights_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  return check_launch("ray_weights");
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(nfi_sample_pdf_args a) {
  __shared__ WaveSlab slabs[4];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
  if (ray >= a.n_rays) return;
  WaveSlab& slab = slabs[wave];
  const int M = a.n_bins, K = a.n_samples;
  float bins[2], wts[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int e = j * 64 + lane;
    bins[j] = e < M ? a.bins[ray * M + e] : 0.0f;
    wts[j] = e < M - 1 ? a.weights[ray * (M - 1) + e] : 0.0f;
  }
  build_cdf<2>(slab, bins, wts, M, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int e = j * 64 + lane;
    if (e < K) {
      float u = a.u[ray * a.u_row_stride + e];
      int ind;
      float z = invert_cdf<16>(slab, M, u, ind);
      a.samples[ray * K + e] = z;
      if (a.inds) a.inds[ray * K + e] = ind;
    }
    if (a.cdf && e < M) a.cdf[ray * M + e] = slab.cdf[e];
  }
}

extern "C" int nfi_sample_pdf(const nfi_sample_pdf_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->bins && a->weights && a->u && a->samples, "sample_pdf: null pointer");
  REQUIRE(a->n_rays > 0 && a->n_bins >= 2 && a->n_bins <= 128 && a->n_samples >= 1 && a->n_samples <= 128,
          "sample_pdf: need 2 <= bins <= 128 and 1 <= samples <= 128");
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((a->n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  return check_launch("sample_pdf");
}

__global__ __launch_bounds__(256) void resample_kernel(nfi_resample_args a) {
  __shared__ WaveSlab slabs[4];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
  if (ray >= a.n_rays) return;
  WaveSlab& slab = slabs[wave];
  const int S = a.n_samples;
  float sigma = lane < S ? a.sigma[ray * S + lane] : 0.0f;
  float t = lane < S ? a.depth[ray * S + lane] : 0.0f;
  float u = lane < S ? a.u[ray * a.u_row_stride + lane] : 0.0f;
  float dn = norm3(a.ray_directions[ray * 3], a.ray_directions[ray * 3 + 1], a.ray_directions[ray * 3 + 2]);
  ResampleTaps taps;
  float z = resample_ray(slab, sigma, t, S, dn, u, lane, &taps);
  if (lane < S) {
    a.fine_depth[ray * S + lane] = z;
    if (a.weights) a.weights[ray * S + lane] = taps.w;
    if (a.smooth) a.smooth[ray * S + lane] = taps.smooth;
    if (a.inds) a.inds[ray * S + lane] = taps.ind;
  }
  if (a.cdf && lane < S - 1) a.cdf[ray * (S - 1) + lane] = slab.cdf[lane];
}

__global__ __launch_bounds__(256) void resample_wide_kernel(nfi_resample_args a) {
  __shared__ WaveSlab slabs[4];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
  if (ray >= a.n_rays) return;
  WaveSlab& slab = slabs[wave];
  const int S = a.n_samples;
  float sigma[2], t[2], u[2], z[2], w[2], sm[2];
  int ind[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = j * 64 + lane;
    sigma[j] = e < S ? a.sigma[ray * S + e] : 0.0f;
    t[j] = e < S ? a.depth[ray * S + e] : 0.0f;
    u[j] = e < S ? a.u[ray * a.u_row_stride + e] : 0.0f;
  }
  float dn = norm3(a.ray_directions[ray * 3], a.ray_directions[ray * 3 + 1], a.ray_directions[ray * 3 + 2]);
  float Tc[2];
  resample_ray_wide<2>(slab, sigma, t, S, dn, u, lane, z, w, sm, ind, Tc);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = j * 64 + lane;
    if (e < S) {
      a.fine_depth[ray * S + e] = z[j];
      if (a.weights) a.weights[ray * S + e] = w[j];
      if (a.smooth) a.smooth[ray * S + e] = sm[j];
      if (a.inds) a.inds[ray * S + e] = ind[j];
    }
    if (a.cdf && e < S - 1) a.cdf[ray * (S - 1) + e] = slab.cdf[e];
  }
}

extern "C" int nfi_resample(const nfi_resample_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->sigma && a->ray_directions && a->depth && a->u && a->fine_depth, "resample: null pointer");
  REQUIRE(a->n_rays > 0 && a->n_samples >= 4 && a->n_samples <= NFI_MAX_SAMPLES, "resample: n_samples must be in [4,128]");
  const dim3 grid((unsigned)((a->n_rays + 3) / 4));
  if (a->n_samples <= 64) hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(resample_wide_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
  return check_launch("resample");
}

template <int NS>
__global__ __launch_bounds__(256) void composite_kernel(nfi_composite_args a) {
  __shared__ WaveSlabT<64 * NS> slabs[4];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
  if (ray >= a.n_rays) return;
  auto& slab = slabs[wave];
  const int na = a.n_a, nb = a.n_b, n = na + nb;
  float dep[NS], sig[NS], cr[NS], cg[NS], cb[NS];
  int eidx[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int e = j * 64 + lane;
    dep[j] = sig[j] = cr[j] = cg[j] = cb[j] = 0.0f;
    eidx[j] = e;
    if (e < na) {
      size_t i = (size_t)ray * na + e;
      dep[j] = a.depth_a[i]; sig[j] = a.sigma_a[i];
      cr[j] = a.rgb_a[i * 3]; cg[j] = a.rgb_a[i * 3 + 1]; cb[j] = a.rgb_a[i * 3 + 2];
    } else if (e < n) {
      size_t i = (size_t)ray * nb + (e - na);
      dep[j] = a.depth_b[i]; sig[j] = a.sigma_b[i];
      cr[j] = a.rgb_b[i * 3]; cg[j] = a.rgb_b[i * 3 + 1]; cb[j] = a.rgb_b[i * 3 + 2];
    }
  }
  int rank[NS];
  if (nb > 0) {
    merge_scatter<NS>(slab, dep, sig, cr, cg, cb, eidx, n, rank);
  } else {
    // list a is composited in the order given (render_volume_density does not sort)
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      int e = j * 64 + lane;
      rank[j] = e;
      if (e < n) { slab.srt[0][e] = dep[j]; slab.srt[1][e] = sig[j]; slab.srt[2][e] = cr[j]; slab.srt[3][e] = cg[j]; slab.srt[4][e] = cb[j]; }
    }
    wave_lds_fence();
  }
  float dn = norm3(a.ray_directions[ray * 3], a.ray_directions[ray * 3 + 1], a.ray_directions[ray * 3 + 2]);
  float w[NS];
  CompositeOut o = composite_slab<NS>(slab, n, dn, a.white_background, lane, w);
  if (lane == 0) {
    a.rgb_map[ray * 3] = o.r; a.rgb_map[ray * 3 + 1] = o.g; a.rgb_map[ray * 3 + 2] = o.b;
    a.depth_map[ray] = o.depth;
    a.mask[ray] = o.mask;
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    int e = j * 64 + lane;
    if (e < n) {
      if (a.weights) a.weights[(size_t)ray * n + e] = w[j];
      if (a.depth_sorted) a.depth_sorted[(size_t)ray * n + e] = slab.srt[0][e];
      if (a.perm) a.perm[(size_t)ray * n + rank[j]] = e;
    }
  }
  // extra attribute (semantics / normals / coords): weights are in merged order, attributes in input order
  if (a.n_extra > 0 && a.extra_map) {
    wave_lds_fence();
    // park the merged-order weights in the slab, then every lane accumulates its own elements
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      int e = j * 64 + lane;
      if (e < n) slab.cdf[e] = w[j];
    }
    wave_lds_fence();
    float we[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) { int e = j * 64 + lane; we[j] = e < n ? slab.cdf[rank[j]] : 0.0f; }
    for (int c = 0; c < a.n_extra; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        int e = j * 64 + lane;
        if (e < na) acc += we[j] * a.extra_a[((size_t)ray * na + e) * a.n_extra + c];
        else if (e < n) acc += we[j] * a.extra_b[((size_t)ray * nb + (e - na)) * a.n_extra + c];
      }
      acc = wave_sum(acc);
      if (lane == 0) a.extra_map[ray * a.n_extra + c] = acc;
    }
  }
}

extern "C" int nfi_composite_fwd(const nfi_composite_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->ray_directions && a->depth_a && a->sigma_a && a->rgb_a && a->rgb_map && a->depth_map && a->mask,
          "composite: null pointer");
  REQUIRE(a->n_rays > 0 && a->n_a > 0 && a->n_b >= 0 && a->n_a + a->n_b <= NFI_MAX_SAMPLES_SINGLE_PASS,
          "composite: need 1 <= n_a + n_b <= 512");
  REQUIRE(a->n_b == 0 || (a->n_a <= NFI_MAX_SAMPLES && a->n_b <= NFI_MAX_SAMPLES), "composite: merged lists hold at most 128 samples each");
  REQUIRE(a->n_b == 0 || (a->depth_b && a->sigma_b && a->rgb_b), "composite: list b missing");
  REQUIRE(a->n_extra == 0 || (a->extra_a && (a->n_b == 0 || a->extra_b)), "composite: extra attribute missing");
  const dim3 grid((unsigned)((a->n_rays + 3) / 4));
  if (a->n_a + a->n_b <= 128) hipLaunchKernelGGL(composite_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else if (a->n_a + a->n_b <= 256) hipLaunchKernelGGL(composite_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(composite_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, *a);
  return check_launch("composite_fwd");
}

#include "nfi_backward_rays.inc"
// nfi_backward_field.inc is its own translation unit (nfi_backward_field.hip: built without SLP vectorisation)
#include "nfi_regulariser.inc"
#include "nfi_neighbours.inc"
#include "nfi_handoff.inc"
#include "nfi_pnp.inc"

// ------------------------------------------------------------------------------------------------
// fused forward render
// ------------------------------------------------------------------------------------------------
struct RenderKernelParams {
  int n_scenes, hw, S;
  int fine, white;
  float scene_range;
  // ray set-up results
  const float* ro; const float* rd; const float* near_raw; const float* far_raw; const uint8_t* hit;
  const uint32_t* reduce;
  uint32_t* counter;   // work counter (zeroed with reduce[])
  int width;           // image width (tile order of the work queue)
  int tile_order;      // 1: hand rays out in 8x8 pixel tiles instead of scanlines
  int xcd_blocks;      // 1: every XCD marches its own square pixel blocks (own queue, steals when it runs dry)
  uint32_t* xcd_counter;   // 8 counters, 64 bytes apart
  int xcd_block_shift;     // log2 of the block side: 3, 4 or 5
  int fetch_batch;         // positions taken per atomic
  // field
  const void* texels; int res; int layout;
  const float* image; int A; const float* att;
  int use_sdf; const float* beta; const float* alpha;
  // noise
  const float* noise_c; const float* noise_f; int64_t noise_f_stride;
  // outputs
  float* rgb; float* depth; float* mask;
  // taps
  float* near_plane; float* far_plane;
  float* t_coarse; float* sigma_coarse; float* rgb_coarse;
  float* t_fine; float* sigma_fine; float* rgb_fine;
  float* t_sorted; float* weights; int32_t* perm;
  int skip_missed;
  unsigned long long* prof;
  const float* xray;   // view-direction decoder: padded per-ray features [N][kRayFeatPad], or null
  float term_eps;      // kRenderTerm kernels: coarse transmittance below which fine samples are no longer evaluated
  float* semantics;    // kRenderExtra kernels: composited softmax probabilities [N][A], or null
  float* coords;       // kRenderExtra kernels: composited query points [N][3], or null
  float* normals;      // kRenderNormals kernels: composited unit normals [N][3]
  unsigned long long* clock_probe;   // null, or {shader cycles, 100 MHz ticks} lived by workgroup 0 / wave 0
  FastDiv div_hw, div_bps, div_bw;   // division by rays per image, blocks per scene, blocks per image row (nfi_device.hpp)
  int tap_stride;      // entries per ray in the per-sample tap arrays: S, or 2S for the training stash (fine half at +S)
  int stash;           // 1: the taps are the training stash: missed rays keep being skipped and get an all-zero row
};

// start / end of the clock probe (nfi_render_args.clock_probe): one wave of the persistent grid lives as long as the launch
struct ClockProbe {
  unsigned long long c0 = 0, r0 = 0;
  bool on = false;
  __device__ __forceinline__ void start(const RenderKernelParams& k) {
    on = k.clock_probe && blockIdx.x == 0 && threadIdx.x == 0;
    if (on) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  }
  __device__ __forceinline__ void stop(const RenderKernelParams& k) const {
    if (on) {
      k.clock_probe[0] = __builtin_readcyclecounter() - c0;
      k.clock_probe[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
  }
};

__device__ __forceinline__ float uniform_f32(float v) {
  return bits2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2bits(v)));
}

// Persistent kernel: one wave per ray, rays handed out by one device-scope counter (scene-major,
// so the chip works on one scene's 25 MB of texels at a time).  The ray index two steps ahead is
// being fetched and the next ray's inputs are loaded while the current ray is marched.
// OCC = waves per SIMD the register budget is held to; TAPS = write the optional stage taps.
struct RayInputs {
  float ox, oy, oz, dx, dy, dz, near, far, noise, u;
  uint32_t hit;
};

// Work hand-out of the persistent render kernels.  One device-scope counter for all waves serialises at ~12 ns per
// fetch on this chip (131 k rays -> 1.6 ms: with it the counter, not the field, set the kernel time), so every XCD
// has its own counter, served by its own L2: the image is cut into square pixel blocks, block b belongs to XCD b % 8
// (all XCDs stay on the same scene), positions inside a block walk 8x8 sub-tiles, and an XCD that runs dry steals
// from the next one's queue.  fetch() is wave-uniform and returns a RAY ID (n_rays: no work left); without the
// per-XCD queues (image sides not multiples of the block side, or tuning bit 4) it returns a position of the single
// queue, which ray_of() maps to a ray.
struct RayQueue {
  const RenderKernelParams& k;
  int lane;
  uint32_t n_rays, xcd, bsh, bw, bps, n_blocks, q_cur, pos_next, pos_end, batch;
  bool dry;
  __device__ __forceinline__ RayQueue(const RenderKernelParams& kk, int l) : k(kk), lane(l) {
    n_rays = (uint32_t)k.n_scenes * (uint32_t)k.hw;
    xcd = k.xcd_blocks ? (__builtin_amdgcn_s_getreg(6164) & 7u) : 0u;     // hwreg(HW_REG_XCC_ID, 0, 4)
    bsh = (uint32_t)k.xcd_block_shift;                                   // log2 of the block side (3..5)
    bw = (uint32_t)k.width >> bsh;
    bps = bw * (((uint32_t)k.hw / (uint32_t)k.width) >> bsh);
    n_blocks = bps * (uint32_t)k.n_scenes;
    q_cur = xcd; pos_next = 0; pos_end = 0; dry = false;
    batch = (uint32_t)k.fetch_batch;
  }
  __device__ __forceinline__ uint32_t ray_at(uint32_t q, uint32_t pos) const {
    const uint32_t b = (pos >> (2 * bsh)) * 8u + q;
    const uint32_t in = pos & ((1u << (2 * bsh)) - 1u), scene = fastdiv(b, k.div_bps), bb = b - scene * bps;
    const uint32_t by = fastdiv(bb, k.div_bw), bx = bb - by * bw;
    const uint32_t st = in >> 6, sty = st >> (bsh - 3), stx = st & ((1u << (bsh - 3)) - 1u);   // 8x8 sub-tile of the block
    const uint32_t y = (by << bsh) + (sty << 3) + ((in >> 3) & 7u), x = (bx << bsh) + (stx << 3) + (in & 7u);
    return scene * (uint32_t)k.hw + y * (uint32_t)k.width + x;
  }
  __device__ __forceinline__ uint32_t fetch() {
    if (!k.xcd_blocks) {
      uint32_t p = 0;
      if (lane == 0) p = atomicAdd(k.counter, 1u);
      return (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
    }
    if (pos_next == pos_end && !dry) {
      dry = true;
      for (uint32_t i = 0; i < 8; ++i) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(k.xcd_counter + q_cur * 16, batch);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((base >> (2 * bsh)) * 8u + q_cur < n_blocks) { pos_next = base; pos_end = base + batch; dry = false; break; }
        q_cur = (q_cur + 1) & 7u;        // this queue is empty: steal from the next XCD's
      }
    }
    if (dry) return n_rays;
    return ray_at(q_cur, pos_next++);
  }
};

// Variants of the persistent render kernels (one template argument; they exclude each other):
//   kRenderPlain  rgb / depth / mask only - the inference kernel
//   kRenderTaps   + the optional stage taps or the training stash
//   kRenderProf   + per-phase cycle counters (S <= 64)
//   kRenderTerm   ray termination in the FINE pass (nfi_render_args.termination_eps): the coarse pass, hence the pdf and
//                 every sample index, is untouched; fine samples that lie behind the first coarse sample in front of
//                 which the coarse transmittance has fallen below eps are not evaluated (sigma = 0; they stay in the
//                 merge with their depth) and the surviving ones are compacted to the low lanes (wave ballot +
//                 popcount) so that whole 16-point tiles drop out
//   kRenderExtra  + the composited per-sample attributes of run.py:312-338 / lib/nerf_utils.py:147-159: `semantics`
//                 (softmax probabilities, parked per sample in a per-wave LDS table [A][pitch] by the field epilogue and
//                 composited with the merged weights brought back to source order) and `coords` (the query point
//                 o + d t of every merged sample)
//   kRenderNormals  kRenderExtra + the composited `normals` map (SDF only; lib/nerf_utils.py:149-151, 159): every sample's
//                 normalize(d sdf / d x) from the analytic derivative of the decoder (field_wave<..., NRM>), composited
//                 with the merged weights in source order like the semantics
constexpr int kRenderPlain = 0, kRenderTaps = 1, kRenderProf = 2, kRenderTerm = 3, kRenderExtra = 4, kRenderNormals = 5;
constexpr int kSemPitch = 2 * 64 + 4;                 // pitch = 4 (mod 64) dwords: conflict-free stores from the MFMA layout
// the 128 + 128 kernel parks the probabilities as unorm16 (tile_epilogue<SEMP < 0>): 41.6 KB of fp32 tables left room for ONE
// workgroup per CU next to the 47 KB of slabs and operands; 21 KB let two in (cfg5 + semantics 0.59 -> see DESIGN 4.2a).
// Pitch in 16-bit entries: 4 rows = 528 dwords = 16 (mod 64), the four row groups' stores fall into different banks
constexpr int kSemPitchWide = 2 * 128 + 8;
// dynamic LDS of the kRenderExtra / kRenderNormals kernels: [normal operands: kNrmLdsFloats, kRenderNormals only]
// [semantics tables: 4 waves x A x pitch floats, when asked for]
extern __shared__ __attribute__((aligned(16))) float nfi_dyn_lds[];

typedef __attribute__((address_space(3))) float lds_float;

// sum_e w_e (o + d t_e) over the merged samples (coords = x_in of the sampler, generator.py:643; run.py:337 puts it in the
// semantics slot of render_volume_density)
template <int NS, class Slab>
__device__ __forceinline__ void composite_coords(const Slab& slab, const float (&w)[NS], int n, int lane, float ox, float oy,
                                                 float oz, float dx, float dy, float dz, float* out3) {
  float cx = 0.0f, cy = 0.0f, cz = 0.0f;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int e = j * 64 + lane;
    if (e < n) {
      const float t = slab.srt[0][e];
      cx += w[j] * (ox + dx * t); cy += w[j] * (oy + dy * t); cz += w[j] * (oz + dz * t);
    }
  }
  cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz);
  if (lane == 0) { out3[0] = cx; out3[1] = cy; out3[2] = cz; }
}

template <int TEX, bool ATT, int OCC, int MODE, int PREC, bool VD = false>
__global__ __launch_bounds__(256, OCC) void render_fwd_kernel(RenderKernelParams k) {
  constexpr bool TAPS = MODE == kRenderTaps, PROF = MODE == kRenderProf, TERM = MODE == kRenderTerm;
  constexpr bool EXTRA = MODE == kRenderExtra || MODE == kRenderNormals, NRM = MODE == kRenderNormals;
  constexpr int SEMP = (EXTRA && ATT) ? kSemPitch : 0;
  constexpr int kImg = VD ? kVdImageFloats : kLdsImageFloats;
  __shared__ __attribute__((aligned(16))) float lds[kImg];
  __shared__ __attribute__((aligned(16))) float vfs[4][64];
  __shared__ WaveSlab slabs[4];
  if constexpr (NRM) stage_normal_operands(nfi_dyn_lds, k.image, VD ? kVdW1F : kW1F, VD ? kVdW2 : kW2F);
  ClockProbe clock;
  clock.start(k);
  if (PREC == 1) {
    // fp16 fragments overlay the fp32 fragment area; biases stay where they are
    for (int i = threadIdx.x; i < kB1F; i += blockDim.x) lds[i] = k.image[kW1H + i];
    for (int i = kB1F + threadIdx.x; i < kLdsImageFloats; i += blockDim.x) lds[i] = k.image[i];
  } else {
    for (int i = threadIdx.x; i < kImg; i += blockDim.x) lds[i] = k.image[i];
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  WaveSlab& slab = slabs[wave];
  float* vf = vfs[wave];
  const int S = k.S;
  const bool valid = lane < S;
  const float fill_near = ordered_key_inv(~k.reduce[0]), fill_far = ordered_key_inv(k.reduce[1]);
  const float bg = k.white ? 1.0f : 0.0f;
  const size_t tb = TEX == 0 ? 128 : 64;
  const uint32_t n_rays = (uint32_t)k.n_scenes * (uint32_t)k.hw;
  // queue position -> ray id.  Tile order: consecutive positions walk 8x8 pixel tiles, so the few
  // thousand rays in flight at any time cover a compact image region (a compact part of the three
  // planes) instead of a band of scanlines - better L2/Infinity-Cache reuse of the gather stream.
  const uint32_t tiles_x = (uint32_t)k.width >> 3;
  auto ray_of = [&](uint32_t pos) -> uint32_t {
    if (k.xcd_blocks || !k.tile_order) return pos;
    const uint32_t scene = pos / (uint32_t)k.hw, p = pos - scene * (uint32_t)k.hw;
    const uint32_t tile = p >> 6, in = p & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    return scene * (uint32_t)k.hw + ((ty << 3) + (in >> 3)) * (uint32_t)k.width + (tx << 3) + (in & 7u);
  };

  FieldParams P = make_field_params(k.texels, k.res, TEX, k.A, k.use_sdf, k.beta, k.alpha, lds, kImg, k.layout);
  P.vf = vf;
  P.w1t = nfi_dyn_lds; P.w2r0 = nfi_dyn_lds + kW1TFloats;       // (kRenderNormals only)
  int cur_scene = -1;

  // kRenderExtra: this wave's semantics table [A][kSemPitch]: column = sample (coarse [0,64), fine [64,128)).  Zeroed once:
  // a tile the field skips (all 16 points outside the cube) leaves an earlier ray's probabilities behind, and those meet
  // weights that are exactly 0 - finite stale values are harmless, uninitialised LDS would not be.
  float* semT = nullptr;
  if constexpr (SEMP > 0) {
    if (k.semantics) {
      semT = nfi_dyn_lds + (NRM ? kNrmLdsFloats : 0) + wave * (k.A * kSemPitch);
      for (int i = lane; i < k.A * kSemPitch; i += 64) ((lds_float*)semT)[i] = 0.0f;
      wave_lds_fence();
    }
  }

  auto load_inputs = [&](uint32_t ray, RayInputs& in) {
    const size_t r3 = (size_t)ray * 3;
    in.hit = k.hit[ray];
    in.ox = k.ro[r3]; in.oy = k.ro[r3 + 1]; in.oz = k.ro[r3 + 2];
    in.dx = k.rd[r3]; in.dy = k.rd[r3 + 1]; in.dz = k.rd[r3 + 2];
    in.near = k.near_raw[ray]; in.far = k.far_raw[ray];
    in.noise = (k.noise_c && valid) ? k.noise_c[(size_t)ray * S + lane] : 0.0f;
    in.u = (k.fine && valid) ? k.noise_f[(size_t)ray * k.noise_f_stride + lane] : 0.0f;
  };

  // ray indices: cur (being marched), nxt (inputs being loaded), and one more in flight
  RayQueue queue(k, lane);
  auto fetch = [&]() -> uint32_t { return queue.fetch(); };
  uint32_t cur = fetch(), nxt = fetch(), fly = 0;
  RayInputs in, pre;
  // PROF: per-wave cycle accumulators [0..3] field tiles (see field_wave), [4] ray set-up, [5] coarse field,
  // [6] resample, [7] fine field, [8] merge, [9] composite + store, [10] rays, [11] total
  unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk0 = PROF ? __builtin_readcyclecounter() : 0;
  if (cur < n_rays) load_inputs(ray_of(cur), in);
  while (cur < n_rays) {
    if (nxt < n_rays) load_inputs(ray_of(nxt), pre);
    const uint32_t ray = ray_of(cur);
    const uint32_t hitb = in.hit;
    // the ray's five results: stored after the work fetch at the end of the iteration (see there)
    float out_r = bg, out_g = bg, out_b = bg, out_d = 0.0f, out_m = 0.0f;
    if (k.skip_missed && !(hitb & 2)) {
      // the ray's line stays outside the (inflated) scene cube: every sample has sigma == 0
      if constexpr (EXTRA) {
        if (k.coords && lane < 3) k.coords[(size_t)ray * 3 + lane] = 0.0f;
        if (k.semantics && lane < k.A) k.semantics[(size_t)ray * k.A + lane] = 0.0f;
        if (NRM && lane < 3) k.normals[(size_t)ray * 3 + lane] = bg;      // sum w n + (1 - mask) on a white background
      }
      if constexpr (TAPS) {
        if (k.stash && valid) {
          // an all-zero row: its composite backward is exactly zero and its points (the ray origin) carry no gradient
          const size_t zs = (size_t)ray * (size_t)k.tap_stride + lane;
          k.t_coarse[zs] = 0.0f; k.sigma_coarse[zs] = 0.0f;
          float* q = k.rgb_coarse + zs * 3; q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f;
          if (k.fine) {
            k.t_fine[zs] = 0.0f; k.sigma_fine[zs] = 0.0f;
            q = k.rgb_fine + zs * 3; q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f;
          }
        }
      }
    } else {
      unsigned long long t0 = PROF ? __builtin_readcyclecounter() : 0;
      const int scene = (int)fastdiv(ray, k.div_hw);
      if (scene != cur_scene) {
        cur_scene = scene;
        const char* tex_scene = reinterpret_cast<const char*>(k.texels) + (size_t)scene * 3 * k.res * k.res * tb;
        P.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tex_scene), 0, (int)P.scene_bytes, 0x00020000);
        wave_lds_fence();
        {
          int c = lane & 3, row = lane >> 2;
          float v = 0.0f;
          if (k.att && c < 3 && row >= 1 && row <= k.A) v = k.att[((size_t)scene * k.A + (row - 1)) * 3 + c];
          vf[lane] = v;
        }
        wave_lds_fence();
      }
      const float ox = in.ox, oy = in.oy, oz = in.oz, dx = in.dx, dy = in.dy, dz = in.dz;
      float near = in.near, far = in.far;
      finish_planes((hitb & 1) != 0, fill_near, fill_far, near, far);
      const float dnorm = norm3(dx, dy, dz);
      const size_t rs = (size_t)ray * (size_t)k.tap_stride;      // row of this ray in the per-sample tap / stash arrays

      // ---- coarse pass ----
      float tc = 0.0f;
      if (valid) tc = stratified_depth(near, far, lane, S, in.noise, k.noise_c != nullptr);
      MergeIn c;
      float ncx = 0.0f, ncy = 0.0f, ncz = 0.0f, nfx = 0.0f, nfy = 0.0f, nfz = 0.0f;     // NRM: the samples' unit normals
      unsigned long long t1 = PROF ? __builtin_readcyclecounter() : 0;
      {
        SampleOut q = field_wave<TEX, ATT, true, PREC, VD, SEMP, NRM>(P, k.scene_range, lane, ox + dx * tc, oy + dy * tc, oz + dz * tc,
                                                                      valid, semT, nullptr, &slab.srt[0][0], PROF ? pc : nullptr,
                                                                      k.xray, (int)ray);
        c.t = tc; c.sigma = q.sigma; c.r = q.r; c.g = q.g; c.b = q.b;
        if constexpr (NRM) { ncx = q.nx; ncy = q.ny; ncz = q.nz; }
      }
      int n = S;
      int rank_c = lane, rank_f = 0;
      unsigned long long t2 = PROF ? __builtin_readcyclecounter() : 0, t3 = t2, t4 = t2, t5 = t2;
      if (k.fine) {
        // ---- hierarchical resampling + fine pass ----
        ResampleTaps rt;
        float tf = resample_ray(slab, c.sigma, tc, S, dnorm, in.u, lane, TERM ? &rt : nullptr);
        MergeIn f;
        if (PROF) { asm volatile("" :: "v"(tf)); t3 = __builtin_readcyclecounter(); }
        if constexpr (TERM) {
          // depth of the first coarse sample in front of which the coarse transmittance is already below eps
          const uint64_t dm = __ballot(valid && rt.T < k.term_eps);
          float t_dead = INFINITY;
          if (dm) t_dead = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(tc), (int)__builtin_ctzll(dm)));
          // compaction: live fine samples to the low lanes, dropped ones behind them (they keep their depth)
          const bool live = valid && tf <= t_dead;
          const uint64_t lm = __ballot(live);
          const int nl = __builtin_popcountll(lm);
          const int nb = __builtin_popcountll(lm & ((1ull << lane) - 1ull));
          const int pos = live ? nb : nl + (lane - nb);
          wave_lds_fence();
          slab.cdf[pos] = tf;
          wave_lds_fence();
          tf = slab.cdf[lane];
          wave_lds_fence();
          const bool vl = lane < nl;
          SampleOut q = field_wave<TEX, ATT, true, PREC, VD>(P, k.scene_range, lane, ox + dx * tf, oy + dy * tf, oz + dz * tf, vl,
                                                             nullptr, nullptr, &slab.srt[0][0], nullptr, k.xray, (int)ray);
          f.t = tf; f.sigma = vl ? q.sigma : 0.0f; f.r = vl ? q.r : 0.0f; f.g = vl ? q.g : 0.0f; f.b = vl ? q.b : 0.0f;
        } else {
          SampleOut q = field_wave<TEX, ATT, true, PREC, VD, SEMP, NRM>(P, k.scene_range, lane, ox + dx * tf, oy + dy * tf, oz + dz * tf,
                                                                        valid, semT ? semT + 64 : nullptr, nullptr, &slab.srt[0][0],
                                                                        PROF ? pc : nullptr, k.xray, (int)ray);
          f.t = tf; f.sigma = q.sigma; f.r = q.r; f.g = q.g; f.b = q.b;
          if constexpr (NRM) { nfx = q.nx; nfy = q.ny; nfz = q.nz; }
        }
        if constexpr (TAPS) {
          if (valid) {
            if (k.t_fine) k.t_fine[rs + lane] = tf;
            if (k.sigma_fine) k.sigma_fine[rs + lane] = f.sigma;
            if (k.rgb_fine) { float* q = k.rgb_fine + (rs + lane) * 3; q[0] = f.r; q[1] = f.g; q[2] = f.b; }
          }
        }
        n = 2 * S;
        if (PROF) t4 = __builtin_readcyclecounter();
        merge_pair_scatter(slab, c, f, S, lane, rank_c, rank_f);
        if (PROF) t5 = __builtin_readcyclecounter();
      } else {
        if (valid) { slab.srt[0][lane] = c.t; slab.srt[1][lane] = c.sigma; slab.srt[2][lane] = c.r; slab.srt[3][lane] = c.g; slab.srt[4][lane] = c.b; }
        wave_lds_fence();
      }
      float w[2];
      CompositeOut o = composite_slab<2>(slab, n, dnorm, k.white, lane, w);
      out_r = o.r; out_g = o.g; out_b = o.b; out_d = o.depth; out_m = o.mask;
      if constexpr (EXTRA) {
        if (k.coords) composite_coords<2>(slab, w, n, lane, ox, oy, oz, dx, dy, dz, k.coords + (size_t)ray * 3);
        float wc = 0.0f, wf = 0.0f;
        if ((SEMP > 0 && semT) || NRM) {
          // the merged weights back in source order (lane = sample): the cdf row is free after the merge
          wave_lds_fence();
          slab.cdf[lane] = w[0]; slab.cdf[64 + lane] = w[1];
          wave_lds_fence();
          wc = valid ? slab.cdf[rank_c] : 0.0f;
          wf = (valid && k.fine) ? slab.cdf[rank_f] : 0.0f;
        }
        if constexpr (NRM) {
          // normal_map = sum_k w_k n_k (+ 1 - mask on a white background), lib/nerf_utils.py:149-151, 159
          const float bgn = k.white ? 1.0f - o.mask : 0.0f;
          const float mx = wave_sum(wc * ncx + wf * nfx) + bgn, my = wave_sum(wc * ncy + wf * nfy) + bgn,
                      mz = wave_sum(wc * ncz + wf * nfz) + bgn;
          if (lane == 0) { float* q = k.normals + (size_t)ray * 3; q[0] = mx; q[1] = my; q[2] = mz; }
        }
        if constexpr (SEMP > 0) {
          if (semT) {
            const lds_float* sl = (const lds_float*)semT;
            float mine = 0.0f;
            for (int a = 0; a < k.A; ++a) {
              const float sa = wave_sum(wc * sl[a * kSemPitch + lane] + wf * sl[a * kSemPitch + 64 + lane]);
              if (lane == a) mine = sa;
            }
            if (lane < k.A) k.semantics[(size_t)ray * k.A + lane] = mine;
          }
        }
      }
      if constexpr (TAPS) {
        if (lane == 0) {
          if (k.near_plane) k.near_plane[ray] = near;
          if (k.far_plane) k.far_plane[ray] = far;
        }
        if (valid) {
          if (k.t_coarse) k.t_coarse[rs + lane] = tc;
          if (k.sigma_coarse) k.sigma_coarse[rs + lane] = c.sigma;
          if (k.rgb_coarse) { float* q = k.rgb_coarse + (rs + lane) * 3; q[0] = c.r; q[1] = c.g; q[2] = c.b; }
          if (k.perm) {
            k.perm[(size_t)ray * n + rank_c] = lane;
            if (k.fine) k.perm[(size_t)ray * n + rank_f] = S + lane;
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          int e = j * 64 + lane;
          if (e < n) {
            if (k.weights) k.weights[(size_t)ray * n + e] = w[j];
            if (k.t_sorted) k.t_sorted[(size_t)ray * n + e] = slab.srt[0][e];
          }
        }
      }
      wave_lds_fence();  // slab is reused by the next ray
      if (PROF) {
        unsigned long long t6 = __builtin_readcyclecounter();
        pc[4] += t1 - t0; pc[5] += t2 - t1; pc[6] += t3 - t2; pc[7] += t4 - t3; pc[8] += t5 - t4; pc[9] += t6 - t5; pc[10] += 1;
      }
    }
    // The work fetch waits for its atomic's round trip - and, vmcnt being ONE in-order counter of loads and stores, for
    // every store the wave has in flight.  Here, after the ray's arithmetic and BEFORE its result stores, nothing of this
    // ray is in flight any more (the taps / extra maps of those variants excepted): the wait is the atomic's alone.
    fly = fetch();
    if (lane == 0) {
      k.rgb[(size_t)ray * 3] = out_r; k.rgb[(size_t)ray * 3 + 1] = out_g; k.rgb[(size_t)ray * 3 + 2] = out_b;
      k.depth[ray] = out_d; k.mask[ray] = out_m;
    }
    in = pre;
    cur = nxt;
    nxt = fly;
  }
  if (PROF && k.prof && lane == 0) {
    pc[11] = __builtin_readcyclecounter() - tk0;
    for (int i = 0; i < 12; ++i) atomicAdd(k.prof + i, pc[i]);
  }
  clock.stop(k);
}

#ifdef NFI_SINGLE_KERNEL
// register-budget work (tools/vgpr_liveness.py): compile ONLY the plain inference kernel of one texel storage type -
// hipcc -S --cuda-device-only -DNFI_SINGLE_KERNEL=<TEX> [-DNFI_SINGLE_MODE=<kRender...>] -DNFI_RENDER_OCC=<waves per SIMD> ... -
// seconds instead of minutes
#ifndef NFI_SINGLE_MODE
#define NFI_SINGLE_MODE 0     // kRenderPlain ... kRenderExtra
#endif
template __global__ void render_fwd_kernel<NFI_SINGLE_KERNEL, true, NFI_RENDER_OCC, NFI_SINGLE_MODE, 1>(RenderKernelParams);
#endif
#if !defined(NFI_SINGLE_KERNEL) || defined(NFI_SINGLE_WIDE)
// The same pipeline for 64 < S <= 128 samples per pass (BASELINE cfg5, ray_multiplier=2): every lane
// owns two coarse and two fine samples (element e = slot*64 + lane), the field is marched 64 points
// at a time, and the merge ranks all 2S keys against each other.
template <int TEX, bool ATT, int MODE, int PREC, bool VD = false, int OCC = NFI_RENDER_OCC>
__global__ __launch_bounds__(256, OCC) void render_fwd_wide_kernel(RenderKernelParams k) {
  constexpr bool TAPS = MODE == kRenderTaps, TERM = MODE == kRenderTerm;
  constexpr bool EXTRA = MODE == kRenderExtra || MODE == kRenderNormals, NRM = MODE == kRenderNormals;
  constexpr int SEMP = (EXTRA && ATT) ? -kSemPitchWide : 0;        // < 0: 16-bit table entries
  constexpr int kImg = VD ? kVdImageFloats : kLdsImageFloats;
  __shared__ __attribute__((aligned(16))) float lds[kImg];
  __shared__ __attribute__((aligned(16))) float vfs[4][64];
  __shared__ WaveSlabWide slabs[4];
  if constexpr (NRM) stage_normal_operands(nfi_dyn_lds, k.image, VD ? kVdW1F : kW1F, VD ? kVdW2 : kW2F);
  ClockProbe clock;
  clock.start(k);
  if (PREC == 1) {
    for (int i = threadIdx.x; i < kB1F; i += blockDim.x) lds[i] = k.image[kW1H + i];
    for (int i = kB1F + threadIdx.x; i < kLdsImageFloats; i += blockDim.x) lds[i] = k.image[i];
  } else {
    for (int i = threadIdx.x; i < kImg; i += blockDim.x) lds[i] = k.image[i];
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  WaveSlabWide& slab = slabs[wave];
  float* vf = vfs[wave];
  const int S = k.S;
  const float fill_near = ordered_key_inv(~k.reduce[0]), fill_far = ordered_key_inv(k.reduce[1]);
  const float bg = k.white ? 1.0f : 0.0f;
  const size_t tb = TEX == 0 ? 128 : 64;
  const uint32_t n_rays = (uint32_t)k.n_scenes * (uint32_t)k.hw;
  const uint32_t tiles_x = (uint32_t)k.width >> 3;
  auto ray_of = [&](uint32_t pos) -> uint32_t {
    if (k.xcd_blocks || !k.tile_order) return pos;
    const uint32_t scene = pos / (uint32_t)k.hw, p = pos - scene * (uint32_t)k.hw;
    const uint32_t tile = p >> 6, in = p & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    return scene * (uint32_t)k.hw + ((ty << 3) + (in >> 3)) * (uint32_t)k.width + (tx << 3) + (in & 7u);
  };
  FieldParams P = make_field_params(k.texels, k.res, TEX, k.A, k.use_sdf, k.beta, k.alpha, lds, kImg, k.layout);
  P.vf = vf;
  P.w1t = nfi_dyn_lds; P.w2r0 = nfi_dyn_lds + kW1TFloats;       // (kRenderNormals only)
  int cur_scene = -1;
  // kRenderExtra: this wave's semantics table [A][kSemPitchWide], column = sample: coarse [0,128), fine [128,256)
  typedef __attribute__((address_space(3))) unsigned short lds_u16;
  float* semT = nullptr;
  if constexpr (SEMP != 0) {
    if (k.semantics) {
      semT = reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(nfi_dyn_lds + (NRM ? kNrmLdsFloats : 0)) +
                                      wave * (k.A * kSemPitchWide));
      for (int i = lane; i < k.A * kSemPitchWide / 2; i += 64) ((lds_float*)semT)[i] = 0.0f;
      wave_lds_fence();
    }
  }
  // column `c` of the table (16-bit entries), as the pointer field_wave hands to tile_epilogue
  auto sem_col = [&](int c) -> float* {
    return semT ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(semT) + c) : nullptr;
  };
  RayQueue queue(k, lane);
  uint32_t cur = queue.fetch(), nxt = 0;
  while (cur < n_rays) {
    nxt = queue.fetch();
    const uint32_t ray = ray_of(cur);
    const uint32_t hitb = k.hit[ray];
    if (k.skip_missed && !(hitb & 2)) {
      if (lane == 0) {
        k.rgb[(size_t)ray * 3] = bg; k.rgb[(size_t)ray * 3 + 1] = bg; k.rgb[(size_t)ray * 3 + 2] = bg;
        k.depth[ray] = 0.0f; k.mask[ray] = 0.0f;
      }
      if constexpr (EXTRA) {
        if (k.coords && lane < 3) k.coords[(size_t)ray * 3 + lane] = 0.0f;
        if (k.semantics && lane < k.A) k.semantics[(size_t)ray * k.A + lane] = 0.0f;
        if (NRM && lane < 3) k.normals[(size_t)ray * 3 + lane] = bg;
      }
      if constexpr (TAPS) {
        if (k.stash) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (j * 64 + lane < S) {
              const size_t zs = (size_t)ray * (size_t)k.tap_stride + j * 64 + lane;
              k.t_coarse[zs] = 0.0f; k.sigma_coarse[zs] = 0.0f;
              float* q = k.rgb_coarse + zs * 3; q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f;
              if (k.fine) {
                k.t_fine[zs] = 0.0f; k.sigma_fine[zs] = 0.0f;
                q = k.rgb_fine + zs * 3; q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f;
              }
            }
          }
        }
      }
    } else {
      const int scene = (int)fastdiv(ray, k.div_hw);
      if (scene != cur_scene) {
        cur_scene = scene;
        const char* tex_scene = reinterpret_cast<const char*>(k.texels) + (size_t)scene * 3 * k.res * k.res * tb;
        P.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tex_scene), 0, (int)P.scene_bytes, 0x00020000);
        wave_lds_fence();
        {
          int c = lane & 3, row = lane >> 2;
          float v = 0.0f;
          if (k.att && c < 3 && row >= 1 && row <= k.A) v = k.att[((size_t)scene * k.A + (row - 1)) * 3 + c];
          vf[lane] = v;
        }
        wave_lds_fence();
      }
      const size_t r3 = (size_t)ray * 3;
      const float ox = k.ro[r3], oy = k.ro[r3 + 1], oz = k.ro[r3 + 2];
      const float dx = k.rd[r3], dy = k.rd[r3 + 1], dz = k.rd[r3 + 2];
      float near = k.near_raw[ray], far = k.far_raw[ray];
      finish_planes((hitb & 1) != 0, fill_near, fill_far, near, far);
      const float dnorm = norm3(dx, dy, dz);
      const size_t rs = (size_t)ray * S;
      const size_t ts = (size_t)ray * (size_t)k.tap_stride;      // row of this ray in the per-sample tap / stash arrays
      float* stage = &slab.srt[0][0];

      // ---- coarse pass ----
      float tc[2], sc[2], rc[2], gc[2], bc[2];
      bool val[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = j * 64 + lane;
        val[j] = e < S;
        const float nz = (k.noise_c && val[j]) ? k.noise_c[rs + e] : 0.0f;
        tc[j] = val[j] ? stratified_depth(near, far, e, S, nz, k.noise_c != nullptr) : 0.0f;
      }
      float nrm[4][3];                  // NRM: unit normals of the coarse (slots 0, 1) and fine (2, 3) samples
#pragma unroll
      for (int j = 0; j < 4; ++j) nrm[j][0] = nrm[j][1] = nrm[j][2] = 0.0f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        SampleOut q = field_wave<TEX, ATT, true, PREC, VD, SEMP, NRM>(P, k.scene_range, lane, ox + dx * tc[j], oy + dy * tc[j],
                                                                      oz + dz * tc[j], val[j], sem_col(j * 64), nullptr,
                                                                      stage, nullptr, k.xray, (int)ray);
        sc[j] = q.sigma; rc[j] = q.r; gc[j] = q.g; bc[j] = q.b;
        if constexpr (NRM) { nrm[j][0] = q.nx; nrm[j][1] = q.ny; nrm[j][2] = q.nz; }
      }
      int n = S;
      float dep[4], sig[4], cr[4], cg[4], cb[4];
      int eidx[4], rank[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        dep[j] = tc[j]; sig[j] = sc[j]; cr[j] = rc[j]; cg[j] = gc[j]; cb[j] = bc[j];
        eidx[j] = val[j] ? j * 64 + lane : 0x7fffffff;
        dep[2 + j] = sig[2 + j] = cr[2 + j] = cg[2 + j] = cb[2 + j] = 0.0f;
        eidx[2 + j] = 0x7fffffff;
        rank[j] = j * 64 + lane; rank[2 + j] = 0;
      }
      if (k.fine) {
        // ---- hierarchical resampling + fine pass ----
        float u[2], tf[2], wtap[2], smtap[2], Tc[2];
        int ind[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) u[j] = val[j] ? k.noise_f[(size_t)ray * k.noise_f_stride + j * 64 + lane] : 0.0f;
        wave_lds_fence();
        resample_ray_wide<2>(slab, sc, tc, S, dnorm, u, lane, tf, wtap, smtap, ind, Tc);
        bool vfine[2] = {val[0], val[1]};
        if constexpr (TERM) {
          // first coarse sample in front of which the coarse transmittance is below eps; fine samples behind it are dropped
          const uint64_t dm0 = __ballot(val[0] && Tc[0] < k.term_eps);
          const uint64_t dm1 = __ballot(val[1] && Tc[1] < k.term_eps);
          float t_dead = INFINITY;
          if (dm0) t_dead = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(tc[0]), (int)__builtin_ctzll(dm0)));
          else if (dm1) t_dead = bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(tc[1]), (int)__builtin_ctzll(dm1)));
          const bool l0 = val[0] && tf[0] <= t_dead, l1 = val[1] && tf[1] <= t_dead;
          const uint64_t lm0 = __ballot(l0), lm1 = __ballot(l1);
          const int n0 = __builtin_popcountll(lm0), nl = n0 + __builtin_popcountll(lm1);
          const uint64_t below = (1ull << lane) - 1ull;
          const int b0 = __builtin_popcountll(lm0 & below), b1 = __builtin_popcountll(lm1 & below);
          const int p0 = l0 ? b0 : nl + (lane - b0);
          const int p1 = l1 ? n0 + b1 : nl + (64 - n0) + (lane - b1);
          wave_lds_fence();
          slab.cdf[p0] = tf[0];
          slab.cdf[p1] = tf[1];
          wave_lds_fence();
          tf[0] = slab.cdf[lane];
          tf[1] = slab.cdf[64 + lane];
          wave_lds_fence();
          vfine[0] = lane < nl;
          vfine[1] = 64 + lane < nl;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          SampleOut q = field_wave<TEX, ATT, true, PREC, VD, SEMP, NRM>(P, k.scene_range, lane, ox + dx * tf[j], oy + dy * tf[j],
                                                                        oz + dz * tf[j], vfine[j], sem_col(128 + j * 64),
                                                                        nullptr, stage, nullptr, k.xray, (int)ray);
          if (TERM && !vfine[j]) { q.sigma = 0.0f; q.r = 0.0f; q.g = 0.0f; q.b = 0.0f; }
          if constexpr (NRM) { nrm[2 + j][0] = q.nx; nrm[2 + j][1] = q.ny; nrm[2 + j][2] = q.nz; }
          dep[2 + j] = tf[j]; sig[2 + j] = q.sigma; cr[2 + j] = q.r; cg[2 + j] = q.g; cb[2 + j] = q.b;
          eidx[2 + j] = val[j] ? S + j * 64 + lane : 0x7fffffff;
          if constexpr (TAPS) {
            if (val[j]) {
              const size_t i = ts + j * 64 + lane;
              if (k.t_fine) k.t_fine[i] = tf[j];
              if (k.sigma_fine) k.sigma_fine[i] = q.sigma;
              if (k.rgb_fine) { float* o3 = k.rgb_fine + i * 3; o3[0] = q.r; o3[1] = q.g; o3[2] = q.b; }
            }
          }
        }
        n = 2 * S;
        wave_lds_fence();
        if (!merge_pair_scatter_wide(slab, dep, sig, cr, cg, cb, S, lane, rank))
          merge_scatter<4>(slab, dep, sig, cr, cg, cb, eidx, n, rank);
      } else {
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int e = j * 64 + lane;
          if (val[j]) { slab.srt[0][e] = tc[j]; slab.srt[1][e] = sc[j]; slab.srt[2][e] = rc[j]; slab.srt[3][e] = gc[j]; slab.srt[4][e] = bc[j]; }
        }
        wave_lds_fence();
      }
      float w[4];
      CompositeOut o = composite_slab<4>(slab, n, dnorm, k.white, lane, w);
      if (lane == 0) {
        k.rgb[(size_t)ray * 3] = o.r; k.rgb[(size_t)ray * 3 + 1] = o.g; k.rgb[(size_t)ray * 3 + 2] = o.b;
        k.depth[ray] = o.depth; k.mask[ray] = o.mask;
      }
      if constexpr (EXTRA) {
        if (k.coords) composite_coords<4>(slab, w, n, lane, ox, oy, oz, dx, dy, dz, k.coords + (size_t)ray * 3);
        float ws[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if ((SEMP != 0 && semT) || NRM) {
          // the merged weights back in source order (the cdf row is free after the merge)
          wave_lds_fence();
#pragma unroll
          for (int j = 0; j < 4; ++j) slab.cdf[j * 64 + lane] = w[j];
          wave_lds_fence();
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            ws[j] = val[j] ? slab.cdf[rank[j]] : 0.0f;
            ws[2 + j] = (val[j] && k.fine) ? slab.cdf[rank[2 + j]] : 0.0f;
          }
        }
        if constexpr (NRM) {
          const float bgn = k.white ? 1.0f - o.mask : 0.0f;
          float m3[3];
#pragma unroll
          for (int c = 0; c < 3; ++c)
            m3[c] = wave_sum((ws[0] * nrm[0][c] + ws[1] * nrm[1][c]) + (ws[2] * nrm[2][c] + ws[3] * nrm[3][c])) + bgn;
          if (lane == 0) { float* q = k.normals + (size_t)ray * 3; q[0] = m3[0]; q[1] = m3[1]; q[2] = m3[2]; }
        }
        if constexpr (SEMP != 0) {
          if (semT) {
            const lds_u16* sl = (const lds_u16*)semT;
            float mine = 0.0f;
            for (int a = 0; a < k.A; ++a) {
              const lds_u16* row = sl + a * kSemPitchWide + lane;
              const float sa = wave_sum((ws[0] * (float)row[0] + ws[1] * (float)row[64]) + (ws[2] * (float)row[128] + ws[3] * (float)row[192]));
              if (lane == a) mine = sa * (1.0f / 65535.0f);
            }
            if (lane < k.A) k.semantics[(size_t)ray * k.A + lane] = mine;
          }
        }
      }
      if constexpr (TAPS) {
        if (lane == 0) {
          if (k.near_plane) k.near_plane[ray] = near;
          if (k.far_plane) k.far_plane[ray] = far;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (val[j]) {
            const size_t i = ts + j * 64 + lane;
            if (k.t_coarse) k.t_coarse[i] = tc[j];
            if (k.sigma_coarse) k.sigma_coarse[i] = sc[j];
            if (k.rgb_coarse) { float* o3 = k.rgb_coarse + i * 3; o3[0] = rc[j]; o3[1] = gc[j]; o3[2] = bc[j]; }
            if (k.perm) {
              k.perm[(size_t)ray * n + rank[j]] = j * 64 + lane;
              if (k.fine) k.perm[(size_t)ray * n + rank[2 + j]] = S + j * 64 + lane;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int e = j * 64 + lane;
          if (e < n) {
            if (k.weights) k.weights[(size_t)ray * n + e] = w[j];
            if (k.t_sorted) k.t_sorted[(size_t)ray * n + e] = slab.srt[0][e];
          }
        }
      }
      wave_lds_fence();  // slab is reused by the next ray
    }
    cur = nxt;
  }
  clock.stop(k);
}

#endif   // wide kernel
#ifdef NFI_SINGLE_WIDE
template __global__ void render_fwd_wide_kernel<NFI_SINGLE_KERNEL, true, NFI_SINGLE_MODE, 1, false, NFI_RENDER_OCC>(RenderKernelParams);
#endif
#ifndef NFI_SINGLE_KERNEL
// One pass of 128 < S <= 512 samples without hierarchical resampling (run.py:2271: the inversion of a model trained
// without --fine_sampling asks for depth_samples_per_ray * 4 = 512 samples in a single pass).  Same persistent
// one-wave-per-ray scheme; the per-sample results go to this wave's LDS rows (element e = slot*64 + lane, already in
// ascending depth order - render_volume_density does not sort) and are composited from there.  The stage taps and the
// training stash are run-time switches here (a handful of wave-uniform branches per 64 samples).
struct __attribute__((aligned(16))) WaveSlabLong {
  float srt[5][NFI_MAX_SAMPLES_SINGLE_PASS];   // depth, sigma, r, g, b
  float stage[16 * 36];                        // field_wave's feature-tile transpose
};

template <int TEX, bool ATT, int PREC, bool VD = false>
__global__ __launch_bounds__(256, NFI_RENDER_OCC) void render_fwd_long_kernel(RenderKernelParams k) {
  constexpr int kImg = VD ? kVdImageFloats : kLdsImageFloats;
  constexpr int NS = NFI_MAX_SAMPLES_SINGLE_PASS / 64;
  __shared__ __attribute__((aligned(16))) float lds[kImg];
  __shared__ __attribute__((aligned(16))) float vfs[4][64];
  __shared__ WaveSlabLong slabs[4];
  ClockProbe clock;
  clock.start(k);
  if (PREC == 1) {
    for (int i = threadIdx.x; i < kB1F; i += blockDim.x) lds[i] = k.image[kW1H + i];
    for (int i = kB1F + threadIdx.x; i < kLdsImageFloats; i += blockDim.x) lds[i] = k.image[i];
  } else {
    for (int i = threadIdx.x; i < kImg; i += blockDim.x) lds[i] = k.image[i];
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  WaveSlabLong& slab = slabs[wave];
  float* vf = vfs[wave];
  const int S = k.S;
  const int slots = (S + 63) >> 6;
  const float fill_near = ordered_key_inv(~k.reduce[0]), fill_far = ordered_key_inv(k.reduce[1]);
  const float bg = k.white ? 1.0f : 0.0f;
  const size_t tb = TEX == 0 ? 128 : 64;
  const uint32_t n_rays = (uint32_t)k.n_scenes * (uint32_t)k.hw;
  const uint32_t tiles_x = (uint32_t)k.width >> 3;
  auto ray_of = [&](uint32_t pos) -> uint32_t {
    if (k.xcd_blocks || !k.tile_order) return pos;
    const uint32_t scene = pos / (uint32_t)k.hw, p = pos - scene * (uint32_t)k.hw;
    const uint32_t tile = p >> 6, in = p & 63u;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    return scene * (uint32_t)k.hw + ((ty << 3) + (in >> 3)) * (uint32_t)k.width + (tx << 3) + (in & 7u);
  };
  FieldParams P = make_field_params(k.texels, k.res, TEX, k.A, k.use_sdf, k.beta, k.alpha, lds, kImg, k.layout);
  P.vf = vf;
  int cur_scene = -1;
  RayQueue queue(k, lane);
  uint32_t cur = queue.fetch(), nxt = 0;
  while (cur < n_rays) {
    nxt = queue.fetch();
    const uint32_t ray = ray_of(cur);
    const uint32_t hitb = k.hit[ray];
    const size_t ts = (size_t)ray * (size_t)k.tap_stride;
    if (k.skip_missed && !(hitb & 2)) {
      if (lane == 0) {
        k.rgb[(size_t)ray * 3] = bg; k.rgb[(size_t)ray * 3 + 1] = bg; k.rgb[(size_t)ray * 3 + 2] = bg;
        k.depth[ray] = 0.0f; k.mask[ray] = 0.0f;
      }
      if (k.stash) {
        for (int e = lane; e < S; e += 64) {
          k.t_coarse[ts + e] = 0.0f; k.sigma_coarse[ts + e] = 0.0f;
          float* q = k.rgb_coarse + (ts + e) * 3; q[0] = 0.0f; q[1] = 0.0f; q[2] = 0.0f;
        }
      }
    } else {
      const int scene = (int)fastdiv(ray, k.div_hw);
      if (scene != cur_scene) {
        cur_scene = scene;
        const char* tex_scene = reinterpret_cast<const char*>(k.texels) + (size_t)scene * 3 * k.res * k.res * tb;
        P.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tex_scene), 0, (int)P.scene_bytes, 0x00020000);
        wave_lds_fence();
        {
          int c = lane & 3, row = lane >> 2;
          float v = 0.0f;
          if (k.att && c < 3 && row >= 1 && row <= k.A) v = k.att[((size_t)scene * k.A + (row - 1)) * 3 + c];
          vf[lane] = v;
        }
        wave_lds_fence();
      }
      const size_t r3 = (size_t)ray * 3;
      const float ox = k.ro[r3], oy = k.ro[r3 + 1], oz = k.ro[r3 + 2];
      const float dx = k.rd[r3], dy = k.rd[r3 + 1], dz = k.rd[r3 + 2];
      float near = k.near_raw[ray], far = k.far_raw[ray];
      finish_planes((hitb & 1) != 0, fill_near, fill_far, near, far);
      const float dnorm = norm3(dx, dy, dz);
      const size_t rs = (size_t)ray * S;
#pragma unroll 1
      for (int j = 0; j < slots; ++j) {
        const int e = j * 64 + lane;
        const bool val = e < S;
        const float nz = (k.noise_c && val) ? k.noise_c[rs + e] : 0.0f;
        const float t = val ? stratified_depth(near, far, e, S, nz, k.noise_c != nullptr) : 0.0f;
        SampleOut q = field_wave<TEX, ATT, true, PREC, VD>(P, k.scene_range, lane, ox + dx * t, oy + dy * t, oz + dz * t, val,
                                                           nullptr, nullptr, slab.stage, nullptr, k.xray, (int)ray);
        if (val) {
          slab.srt[0][e] = t; slab.srt[1][e] = q.sigma; slab.srt[2][e] = q.r; slab.srt[3][e] = q.g; slab.srt[4][e] = q.b;
          if (k.t_coarse) k.t_coarse[ts + e] = t;
          if (k.sigma_coarse) k.sigma_coarse[ts + e] = q.sigma;
          if (k.rgb_coarse) { float* o3 = k.rgb_coarse + (ts + e) * 3; o3[0] = q.r; o3[1] = q.g; o3[2] = q.b; }
        }
      }
      wave_lds_fence();
      float w[NS];
      CompositeOut o = composite_slab<NS>(slab, S, dnorm, k.white, lane, w);
      if (lane == 0) {
        k.rgb[(size_t)ray * 3] = o.r; k.rgb[(size_t)ray * 3 + 1] = o.g; k.rgb[(size_t)ray * 3 + 2] = o.b;
        k.depth[ray] = o.depth; k.mask[ray] = o.mask;
        if (k.near_plane) k.near_plane[ray] = near;
        if (k.far_plane) k.far_plane[ray] = far;
      }
      if (k.weights || k.t_sorted || k.perm) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const int e = j * 64 + lane;
          if (e < S) {
            if (k.weights) k.weights[rs + e] = w[j];
            if (k.t_sorted) k.t_sorted[rs + e] = slab.srt[0][e];
            if (k.perm) k.perm[rs + e] = e;
          }
        }
      }
      wave_lds_fence();  // the rows are reused by the next ray
    }
    cur = nxt;
  }
  clock.stop(k);
}

extern "C" size_t nfi_render_workspace_bytes(int64_t n_rays) {
  // ro, rd, near_raw, far_raw (fp32) + hit (u8, padded) + reduce[4]
  size_t n = (size_t)n_rays;
  return n * 8 * sizeof(float) + ((n + 63) & ~(size_t)63) + 64 + 8 * 64;     // + 8 per-XCD work counters
}

// workspace carve + ray set-up shared by nfi_render_setup and nfi_render_fwd
struct RenderWorkspace { uint32_t* reduce; uint32_t* xcd_counter; float* ro; float* rd; float* near_raw; float* far_raw; uint8_t* hit; int64_t n; };

static int render_carve(const nfi_render_args* a, RenderWorkspace& w) {
  REQUIRE(a && a->cam2world && a->workspace, "render: null pointer");
  REQUIRE(a->n_scenes > 0 && a->height > 0 && a->width > 0, "render: bad image shape");
  const int64_t n = (int64_t)a->n_scenes * a->height * a->width;
  if (a->workspace_bytes < nfi_render_workspace_bytes(n)) return fail(NFI_ERR_WORKSPACE_TOO_SMALL, "render: workspace too small");
  // [reduce: 16 floats][8 per-XCD work counters, 64 B apart: 128 floats][ro rd near far: 8n floats][hit: n bytes, padded]
  float* ws = reinterpret_cast<float*>(a->workspace);
  constexpr int kRays0 = 16 + 128;
  w.n = n;
  w.reduce = reinterpret_cast<uint32_t*>(ws);
  w.xcd_counter = reinterpret_cast<uint32_t*>(ws + 16);
  w.ro = a->ray_origins ? a->ray_origins : ws + kRays0;
  w.rd = a->ray_directions ? a->ray_directions : ws + kRays0 + 3 * n;
  w.near_raw = ws + kRays0 + 6 * n;
  w.far_raw = ws + kRays0 + 7 * n;
  w.hit = a->hit ? a->hit : reinterpret_cast<uint8_t*>(ws + kRays0 + 8 * n);
  return NFI_OK;
}

static int render_setup(const nfi_render_args* a, const RenderWorkspace& w, hipStream_t s) {
  // (64 + 512 + 32n + pad(n) bytes = nfi_render_workspace_bytes)  ONE memset clears the reduction cells and the counters
  if (hipMemsetAsync(w.reduce, 0, 64 + 8 * 64, s) != hipSuccess) return fail(NFI_ERR_LAUNCH, "render: memset failed");
  const int full_h = a->full_height > 0 ? a->full_height : a->height;
  REQUIRE(a->row_offset >= 0 && a->row_offset + a->height <= full_h, "render: row window outside the image");
  CameraParams cam{a->cam2world, a->focal, a->bbox, a->focal ? a->center : nullptr, full_h, a->width, 1, a->height, a->row_offset};
  RaygenOut rout{w.ro, w.rd, w.near_raw, w.far_raw, w.hit, w.reduce, a->scene_range};
  hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)std::min<int64_t>((w.n + 255) / 256, kRayBlocks)), dim3(256), 0, s, cam, a->n_scenes, rout);
  return NFI_OK;
}

extern "C" int nfi_render_setup(const nfi_render_args* a, nfi_stream_t stream) {
  RenderWorkspace w;
  int rc = render_carve(a, w);
  if (rc) return rc;
  rc = render_setup(a, w, (hipStream_t)stream);
  if (rc) return rc;
  return check_launch("render_setup");
}

extern "C" int nfi_render_fwd(const nfi_render_args* a, nfi_stream_t stream) {
  REQUIRE(a && a->cam2world && a->rgb && a->depth && a->mask && a->workspace, "render: null pointer");
  REQUIRE(a->n_scenes > 0 && a->height > 0 && a->width > 0, "render: bad image shape");
  REQUIRE(a->n_samples >= 4 && a->n_samples <= (a->fine_sampling ? NFI_MAX_SAMPLES : NFI_MAX_SAMPLES_SINGLE_PASS),
          "render: n_samples must be in [4,128] per pass with fine sampling, [4,512] for a single pass");
  REQUIRE(a->n_samples <= 64 || !a->profile_cycles, "render: the cycle profile exists for n_samples <= 64 only");
  REQUIRE(a->n_samples <= NFI_MAX_SAMPLES || !(a->semantics || a->coords || a->normals),
          "render: semantics / coords / normals maps exist for n_samples <= 128");
  REQUIRE(!a->fine_sampling || a->noise_fine, "render: fine sampling needs u (noise_fine)");
  REQUIRE(!a->semantics || a->n_attention > 0, "render: composited semantics need attention values (A > 0)");
  int rc = check_field_common(a->texels, a->plane_res, a->texel_dtype, a->decoder_image, a->n_attention,
                               a->attention_values, a->use_sdf, a->beta, a->alpha, a->texel_layout);
  if (rc) return rc;
  RenderWorkspace w;
  rc = render_carve(a, w);
  if (rc) return rc;
  const int64_t n = w.n;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* reduce = w.reduce;
  uint32_t* xcd_counter = w.xcd_counter;
  float *ro = w.ro, *rd = w.rd, *near_raw = w.near_raw, *far_raw = w.far_raw;
  uint8_t* hit = w.hit;
  REQUIRE(!(a->rays_ready && (a->ray_origins || a->ray_directions || a->hit || a->stash_t)),
          "render: rays_ready needs the ray set-up in the workspace (no ray_origins / ray_directions / hit taps, no stash)");
  if (!a->rays_ready) {
    rc = render_setup(a, w, s);
    if (rc) return rc;
  }

  const bool stash = a->stash_t || a->stash_sigma || a->stash_rgb;
  REQUIRE(!stash || (a->stash_t && a->stash_sigma && a->stash_rgb), "render: the training stash needs stash_t, stash_sigma and stash_rgb");
  REQUIRE(!stash || !(a->t_coarse || a->sigma_coarse || a->rgb_coarse || a->t_fine || a->sigma_fine || a->rgb_fine),
          "render: the training stash and the per-sample debug taps are mutually exclusive");
  const bool debug_tap = a->t_coarse || a->sigma_coarse || a->rgb_coarse || a->t_fine || a->sigma_fine || a->rgb_fine ||
                         a->t_sorted || a->weights || a->perm || a->near_plane || a->far_plane;
  const bool any_tap = debug_tap || stash;
  RenderKernelParams k;
  memset(&k, 0, sizeof(k));
  k.n_scenes = a->n_scenes; k.hw = a->height * a->width; k.S = a->n_samples;
  k.fine = a->fine_sampling; k.white = a->white_background; k.scene_range = a->scene_range;
  k.ro = ro; k.rd = rd; k.near_raw = near_raw; k.far_raw = far_raw; k.hit = hit; k.reduce = reduce;
  k.texels = a->texels; k.res = a->plane_res; k.layout = a->texel_layout; k.image = a->decoder_image; k.A = a->n_attention; k.att = a->attention_values;
  k.use_sdf = a->use_sdf; k.beta = a->beta; k.alpha = a->alpha;
  k.noise_c = a->noise_coarse; k.noise_f = a->noise_fine; k.noise_f_stride = a->noise_fine_row_stride;
  k.rgb = a->rgb; k.depth = a->depth; k.mask = a->mask;
  k.near_plane = a->near_plane; k.far_plane = a->far_plane;
  k.t_coarse = a->t_coarse; k.sigma_coarse = a->sigma_coarse; k.rgb_coarse = a->rgb_coarse;
  k.t_fine = a->t_fine; k.sigma_fine = a->sigma_fine; k.rgb_fine = a->rgb_fine;
  k.tap_stride = a->n_samples;
  if (stash) {
    // the stash rows hold the coarse samples in [0,S) and - with fine sampling - the fine samples in [S,2S)
    k.tap_stride = (a->fine_sampling ? 2 : 1) * a->n_samples; k.stash = 1;
    k.t_coarse = a->stash_t; k.sigma_coarse = a->stash_sigma; k.rgb_coarse = a->stash_rgb;
    if (a->fine_sampling) {
      k.t_fine = a->stash_t + a->n_samples;
      k.sigma_fine = a->stash_sigma + a->n_samples;
      k.rgb_fine = a->stash_rgb + 3 * (size_t)a->n_samples;
    }
  }
  k.t_sorted = a->t_sorted; k.weights = a->weights; k.perm = a->perm;
  k.skip_missed = (a->skip_missed_rays && !debug_tap) ? 1 : 0;

  k.counter = reduce + 3;
  k.width = a->width;
  // per-XCD queues (tuning bit 4 switches them off): square pixel blocks, needs image sides that are multiples of 8
  {
    // the largest of 32 / 16 / 8-pixel blocks that divides both image sides, TWO positions per atomic.
    // MI355X, 8 x 128^2 x (64+64), ms per launch chairs-like / every ray hits / one image (round 3, build-time variants):
    // 32x32 + 2: 0.854 / 1.299 / 0.164; 8x8 + 2: 0.848 / 1.328 / 0.167; 16x16 + 2: 0.891 / 1.301 / 0.170; 16x16 + 1 (the
    // round-1 default): 0.933 / 1.311 / 0.185; 32x32 + 1: 0.882 / 1.317 / 0.174; 4 per atomic: 0.88-0.90 / 1.33-1.38; one
    // device-wide counter 1.82 / 2.10.  Halving the atomics matters on every workload (the wave waits for each one's
    // result); the block side mostly through the balance of the chairs-like images, whose rays that cross the cube are
    // clustered.
    k.fetch_batch = 2;
    const int both = a->width | a->height;
    k.xcd_block_shift = (both & 31) == 0 ? 5 : ((both & 15) == 0 ? 4 : 3);
    const int side = 1 << k.xcd_block_shift;
    k.xcd_blocks = (((a->tuning >> 4) & 1) == 0 && (a->width % side == 0) && (a->height % side == 0)) ? 1 : 0;
    const uint32_t bw = (uint32_t)a->width >> k.xcd_block_shift, bh = (uint32_t)a->height >> k.xcd_block_shift;
    k.div_hw = make_fastdiv((uint32_t)k.hw);
    k.div_bw = make_fastdiv(bw > 0 ? bw : 1u);
    k.div_bps = make_fastdiv(bw * bh > 0 ? bw * bh : 1u);
  }
  k.xcd_counter = xcd_counter;
  k.tile_order = (((a->tuning >> 2) & 1) == 0 && (a->width % 8 == 0) && (a->height % 8 == 0)) ? 1 : 0;
  k.prof = (unsigned long long*)a->profile_cycles;
  k.xray = a->ray_features;
  k.clock_probe = reinterpret_cast<unsigned long long*>(a->clock_probe);
  const bool strict = ((a->tuning >> 3) & 1) != 0;   // exact-fp32 MLP instead of the split-fp16 one
  const bool term = a->termination_eps > 0.0f;
  const bool extra = a->semantics || a->coords || a->normals;
  REQUIRE(!a->normals || a->use_sdf, "render: the normals map needs the SDF decoder (use_sdf)");
  REQUIRE(a->termination_eps >= 0.0f && a->termination_eps < 1.0f, "render: termination_eps must be in [0,1)");
  REQUIRE(!term || a->fine_sampling, "render: termination_eps acts on the fine pass (fine_sampling)");
  REQUIRE(!term || !(any_tap || extra || a->profile_cycles || a->ray_features || strict),
          "render: termination_eps cannot be combined with stage taps, extra maps, the cycle profile, the view-direction decoder or the exact-fp32 MLP");
  REQUIRE(!extra || !(any_tap || a->profile_cycles || strict),
          "render: semantics / coords / normals maps cannot be combined with stage taps, the cycle profile or the exact-fp32 MLP");
  REQUIRE(!(extra && a->ray_features) || a->texel_dtype == NFI_TEXEL_F32,
          "render: with the view-direction decoder the semantics / coords / normals maps exist for fp32 texels");
  // the exact-fp32 MLP (a diagnostic of the split-fp16 arithmetic) and the cycle profile are built for fp32 texels only:
  // with 16-bit texel storage the texels, not the MLP operands, set the precision
  REQUIRE(!(strict || a->profile_cycles) || a->texel_dtype == NFI_TEXEL_F32,
          "render: the exact-fp32 MLP (tuning bit 3) and the cycle profile exist for fp32 texels");
  k.term_eps = a->termination_eps;
  k.semantics = a->semantics; k.coords = a->coords; k.normals = a->normals;
  REQUIRE(!(a->ray_features && a->profile_cycles), "render: no cycle profile with the view-direction decoder");
  // persistent 1-D grid: OCC blocks of 4 waves per CU, never more blocks than rays need
  // 2 blocks (8 waves) per CU: with the whole 256-VGPR budget the field tile keeps more loads and MFMA chains in
  // flight than at 3 blocks/CU (MI355X, 8 x 128^2: 0.96 vs 1.01-1.08 ms; 4 blocks/CU spill: 1.52 ms)
  // (fp16 texel storage, plain inference: the texels stay packed - 168 registers - and THREE blocks per CU fit without a
  //  spill: 0.705 vs 0.756 ms at 8 x 128^2, every ray hits 1.040 vs 1.174 ms)
  const int occ = NFI_RENDER_OCC;
  int64_t blocks = (int64_t)256 * occ, blocks3 = (int64_t)256 * 3;
  if (blocks > (n + 3) / 4) blocks = (n + 3) / 4;
  if (blocks3 > (n + 3) / 4) blocks3 = (n + 3) / 4;
  dim3 grid((unsigned)blocks), grid3((unsigned)blocks3);
  bool att = a->n_attention > 0;
  // kRenderExtra with semantics: the per-wave tables [A][pitch] in dynamic LDS
  const bool wide = a->n_samples > 64;
  const bool lng = a->n_samples > NFI_MAX_SAMPLES;     // single pass of up to 512 samples (no fine sampling: checked above)
  // (fp16 texels, 128 + 128, at THREE workgroups per CU - 168 registers, ~40 scratch reloads per ray outside the field
  //  tiles - was measured in round 4 and is slower: 1.387 vs 1.300 ms chairs-like, 2.249 vs 2.156 ms every ray hits, images
  //  identical (profiles/r4/wide_fp16_three_workgroups.log); unlike the 64 + 64 kernel, whose fp16 form gains 8 % from the
  //  third workgroup, a 256-sample ray's texel footprint makes 50 % more rays in flight cost more in the L2 than they hide)
  const size_t sem_lds = (a->semantics ? (size_t)4 * a->n_attention * (wide ? kSemPitchWide * sizeof(unsigned short) : kSemPitch * sizeof(float)) : 0) +
                         (a->normals ? (size_t)kNrmLdsFloats * sizeof(float) : 0);
  constexpr size_t kSemLdsMax = ((size_t)4 * NFI_MAX_ATTENTION * kSemPitch + kNrmLdsFloats) * sizeof(float);
  constexpr size_t kSemLdsMaxWide = (size_t)4 * NFI_MAX_ATTENTION * kSemPitchWide * sizeof(unsigned short) + kNrmLdsFloats * sizeof(float);
  if (a->event_start) (void)hipEventRecord((hipEvent_t)a->event_start, s);
#define NFI_LAUNCH_RENDER(TEX, ATT)                                                                                   \
  do {                                                                                                                \
    if (a->normals) {                                                                                                 \
      NFI_ENSURE_DYNAMIC_LDS((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderNormals, 1>), kSemLdsMax, "render");  \
      hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderNormals, 1>), grid, dim3(256), sem_lds, s, k); \
    } else if (extra) {                                                                                               \
      NFI_ENSURE_DYNAMIC_LDS((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderExtra, 1>), kSemLdsMax, "render");    \
      hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderExtra, 1>), grid, dim3(256), sem_lds, s, k); \
    } else if (term) hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderTerm, 1>), grid, dim3(256), 0, s, k); \
    else if (k.prof) hipLaunchKernelGGL((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderProf, 1>), grid, dim3(256), 0, s, k);      \
    else if (any_tap && strict) hipLaunchKernelGGL((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderTaps, 0>), grid, dim3(256), 0, s, k);   \
    else if (any_tap) hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderTaps, 1>), grid, dim3(256), 0, s, k);          \
    else if (strict) hipLaunchKernelGGL((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderPlain, 0>), grid, dim3(256), 0, s, k);            \
    else if (TEX != 0) hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, 3, kRenderPlain, 1>), grid3, dim3(256), 0, s, k);      \
    else hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderPlain, 1>), grid, dim3(256), 0, s, k);                      \
  } while (0)
#define NFI_LAUNCH_RENDER_WIDE(TEX, ATT)                                                                             \
  do {                                                                                                                \
    if (a->normals) {                                                                                                 \
      NFI_ENSURE_DYNAMIC_LDS((render_fwd_wide_kernel<TEX, ATT, kRenderNormals, 1>), kSemLdsMaxWide, "render");         \
      hipLaunchKernelGGL((render_fwd_wide_kernel<TEX, ATT, kRenderNormals, 1>), grid, dim3(256), sem_lds, s, k);       \
    } else if (extra) {                                                                                               \
      NFI_ENSURE_DYNAMIC_LDS((render_fwd_wide_kernel<TEX, ATT, kRenderExtra, 1>), kSemLdsMaxWide, "render");           \
      hipLaunchKernelGGL((render_fwd_wide_kernel<TEX, ATT, kRenderExtra, 1>), grid, dim3(256), sem_lds, s, k);         \
    } else if (term) hipLaunchKernelGGL((render_fwd_wide_kernel<TEX, ATT, kRenderTerm, 1>), grid, dim3(256), 0, s, k);   \
    else if (any_tap && strict) hipLaunchKernelGGL((render_fwd_wide_kernel<0, ATT, kRenderTaps, 0>), grid, dim3(256), 0, s, k);   \
    else if (any_tap) hipLaunchKernelGGL((render_fwd_wide_kernel<TEX, ATT, kRenderTaps, 1>), grid, dim3(256), 0, s, k);        \
    else if (strict) hipLaunchKernelGGL((render_fwd_wide_kernel<0, ATT, kRenderPlain, 0>), grid, dim3(256), 0, s, k);          \
    else hipLaunchKernelGGL((render_fwd_wide_kernel<TEX, ATT, kRenderPlain, 1>), grid, dim3(256), 0, s, k);                    \
  } while (0)
#define NFI_LAUNCH_RENDER_VD(TEX, ATT)                                                                                        \
  do {                                                                                                                      \
    if (a->normals && TEX == 0) { /* + the normal map (round 6): the distance is row 0 of the second layer here too */       \
      if (wide) {                                                                                                           \
        NFI_ENSURE_DYNAMIC_LDS((render_fwd_wide_kernel<0, ATT, kRenderNormals, 0, true>), kSemLdsMaxWide, "render");         \
        hipLaunchKernelGGL((render_fwd_wide_kernel<0, ATT, kRenderNormals, 0, true>), grid, dim3(256), sem_lds, s, k);       \
      } else {                                                                                                              \
        NFI_ENSURE_DYNAMIC_LDS((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderNormals, 0, true>), kSemLdsMax, "render");  \
        hipLaunchKernelGGL((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderNormals, 0, true>), grid, dim3(256), sem_lds, s, k); \
      }                                                                                                                     \
    } else if (extra && TEX == 0) { /* semantics / coords maps with the view-direction decoder (fp32 texels: checked above) */ \
      if (wide) {                                                                                                           \
        NFI_ENSURE_DYNAMIC_LDS((render_fwd_wide_kernel<0, ATT, kRenderExtra, 0, true>), kSemLdsMaxWide, "render");           \
        hipLaunchKernelGGL((render_fwd_wide_kernel<0, ATT, kRenderExtra, 0, true>), grid, dim3(256), sem_lds, s, k);         \
      } else {                                                                                                              \
        NFI_ENSURE_DYNAMIC_LDS((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderExtra, 0, true>), kSemLdsMax, "render");    \
        hipLaunchKernelGGL((render_fwd_kernel<0, ATT, NFI_RENDER_OCC, kRenderExtra, 0, true>), grid, dim3(256), sem_lds, s, k); \
      }                                                                                                                     \
    } else if (wide) hipLaunchKernelGGL((render_fwd_wide_kernel<TEX, ATT, kRenderTaps, 0, true>), grid, dim3(256), 0, s, k);   \
    else hipLaunchKernelGGL((render_fwd_kernel<TEX, ATT, NFI_RENDER_OCC, kRenderTaps, 0, true>), grid, dim3(256), 0, s, k);                \
  } while (0)
#define NFI_LAUNCH_RENDER_LONG(TEX, ATT)                                                                              \
  do {                                                                                                                \
    if (a->ray_features) hipLaunchKernelGGL((render_fwd_long_kernel<TEX, ATT, 0, true>), grid, dim3(256), 0, s, k);    \
    else if (strict) hipLaunchKernelGGL((render_fwd_long_kernel<0, ATT, 0>), grid, dim3(256), 0, s, k);                \
    else hipLaunchKernelGGL((render_fwd_long_kernel<TEX, ATT, 1>), grid, dim3(256), 0, s, k);                          \
  } while (0)
  if (lng) {
    if (a->texel_dtype == NFI_TEXEL_F32) {
      if (att) NFI_LAUNCH_RENDER_LONG(0, true); else NFI_LAUNCH_RENDER_LONG(0, false);
    } else if (a->texel_dtype == NFI_TEXEL_BF16) {
      if (att) NFI_LAUNCH_RENDER_LONG(1, true); else NFI_LAUNCH_RENDER_LONG(1, false);
    } else {
      if (att) NFI_LAUNCH_RENDER_LONG(2, true); else NFI_LAUNCH_RENDER_LONG(2, false);
    }
  } else if (a->ray_features) {
    if (a->texel_dtype == NFI_TEXEL_F32) {
      if (att) NFI_LAUNCH_RENDER_VD(0, true); else NFI_LAUNCH_RENDER_VD(0, false);
    } else if (a->texel_dtype == NFI_TEXEL_BF16) {
      if (att) NFI_LAUNCH_RENDER_VD(1, true); else NFI_LAUNCH_RENDER_VD(1, false);
    } else {
      if (att) NFI_LAUNCH_RENDER_VD(2, true); else NFI_LAUNCH_RENDER_VD(2, false);
    }
  } else if (wide) {
    if (a->texel_dtype == NFI_TEXEL_F32) {
      if (att) NFI_LAUNCH_RENDER_WIDE(0, true); else NFI_LAUNCH_RENDER_WIDE(0, false);
    } else if (a->texel_dtype == NFI_TEXEL_BF16) {
      if (att) NFI_LAUNCH_RENDER_WIDE(1, true); else NFI_LAUNCH_RENDER_WIDE(1, false);
    } else {
      if (att) NFI_LAUNCH_RENDER_WIDE(2, true); else NFI_LAUNCH_RENDER_WIDE(2, false);
    }
  } else if (a->texel_dtype == NFI_TEXEL_F32) {
    if (att) NFI_LAUNCH_RENDER(0, true); else NFI_LAUNCH_RENDER(0, false);
  } else if (a->texel_dtype == NFI_TEXEL_BF16) {
    if (att) NFI_LAUNCH_RENDER(1, true); else NFI_LAUNCH_RENDER(1, false);
  } else {
    if (att) NFI_LAUNCH_RENDER(2, true); else NFI_LAUNCH_RENDER(2, false);
  }
#undef NFI_LAUNCH_RENDER_LONG
#undef NFI_LAUNCH_RENDER_VD
#undef NFI_LAUNCH_RENDER_WIDE
#undef NFI_LAUNCH_RENDER
  if (a->event_stop) (void)hipEventRecord((hipEvent_t)a->event_stop, s);
  return check_launch("render_fwd");
}
#endif  // NFI_SINGLE_KERNEL
