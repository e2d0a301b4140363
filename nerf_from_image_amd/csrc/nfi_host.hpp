// Host-side plumbing shared by the translation units of libnfi_hip.so (nfi_kernels.hip, nfi_backward_field.hip):
// error buffer behind nfi_last_error(), argument checks, and the device helpers that stage the decoder image in LDS.
#pragma once
#include "nfi_device.hpp"
#include "../../include/nfi_hip.h"

#include <atomic>
#include <cstdio>

extern thread_local char nfi_err_buf[256];          // defined in nfi_kernels.hip

static inline int fail(int code, const char* msg) {
  snprintf(nfi_err_buf, sizeof(nfi_err_buf), "%s", msg);
  return code;
}
static inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(nfi_err_buf, sizeof(nfi_err_buf), "%s: %s", what, hipGetErrorString(e));
    return NFI_ERR_LAUNCH;
  }
  return NFI_OK;
}

// Raises a kernel's dynamic-LDS limit to `bytes` - the LARGEST size any launch of that kernel uses, a constant - once per
// device.  The attribute is process-global per kernel and device; because the value never changes, host threads
// launching concurrently (one per GPU under nn.DataParallel) cannot lower it under each other, and a failure is
// reported instead of surfacing as a failed launch later.
#define NFI_ENSURE_DYNAMIC_LDS(kernel, bytes, what)                                                                  \
  do {                                                                                                               \
    static std::atomic<unsigned long long> nfi_done_{0};                                                             \
    int nfi_dev_ = 0;                                                                                                \
    if (hipGetDevice(&nfi_dev_) != hipSuccess) return fail(NFI_ERR_LAUNCH, what ": hipGetDevice failed");            \
    const unsigned long long nfi_bit_ = 1ull << (nfi_dev_ & 63);                                                     \
    if (!(nfi_done_.load(std::memory_order_acquire) & nfi_bit_)) {                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)(bytes)) != hipSuccess)                                                           \
        return fail(NFI_ERR_LAUNCH, what ": hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");                \
      nfi_done_.fetch_or(nfi_bit_, std::memory_order_release);                                                       \
    }                                                                                                                \
  } while (0)

#define REQUIRE(cond, msg) \
  do {                     \
    if (!(cond)) return fail(NFI_ERR_INVALID_ARGUMENT, msg); \
  } while (0)

static inline int texel_bytes(int dtype) { return dtype == NFI_TEXEL_F32 ? 128 : 64; }

static inline int check_field_common(const void* texels, int plane_res, int texel_dtype, const float* image, int A,
                              const float* att, int use_sdf, const float* beta, const float* alpha, int layout = 0) {
  REQUIRE(layout == NFI_TEXELS_PLANAR || layout == NFI_TEXELS_INTERLEAVED, "field: bad texel layout");
  REQUIRE(texels && image, "field: null texels / decoder image");
  REQUIRE(plane_res >= 2 && plane_res <= 1024, "field: plane_res must be in [2,1024]");
  REQUIRE(texel_dtype >= NFI_TEXEL_F32 && texel_dtype <= NFI_TEXEL_F16, "field: bad texel dtype");
  REQUIRE(A >= 0 && A <= NFI_MAX_ATTENTION, "field: attention_values must be in [0,14]");
  REQUIRE(A == 0 || att, "field: attention_values tensor missing");
  REQUIRE(!use_sdf || (beta && alpha), "field: use_sdf needs beta and alpha");
  return NFI_OK;
}


namespace nfi {

// stage the decoder image (+ this scene's attention values in accumulator layout) into LDS
__device__ __forceinline__ void stage_field_lds(float* lds, const float* image, const float* att_scene, int A,
                                                int n_image = kLdsImageFloats) {
  for (int i = threadIdx.x; i < n_image; i += blockDim.x) lds[i] = image[i];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    int c = i & 3, row = i >> 2;  // row = 4g + r; value for feature row-1
    float v = 0.0f;
    if (att_scene && c < 3 && row >= 1 && row <= A) v = att_scene[(row - 1) * 3 + c];
    lds[n_image + i] = v;
  }
}

__device__ __forceinline__ FieldParams make_field_params(const void* texels_scene, int res, int tex, int A, int use_sdf,
                                                         const float* beta, const float* alpha, const float* lds,
                                                         int n_image = kLdsImageFloats, int layout = 0) {
  FieldParams P;
  uint32_t tb = tex == 0 ? 128u : 64u;
  P.scene_bytes = 3u * (uint32_t)res * (uint32_t)res * tb;
  P.pix_bytes = layout ? 3u * tb : tb;
  P.plane_bytes = layout ? tb : (uint32_t)res * (uint32_t)res * tb;
  P.row_bytes = (uint32_t)res * P.pix_bytes;
  P.row_pix_bytes = P.row_bytes + P.pix_bytes;
  P.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(texels_scene), 0, (int)P.scene_bytes, 0x00020000);
  P.res = res;
  P.res_m1 = (float)(res - 1);
  P.n_attention = A;
  P.use_sdf = use_sdf;
  P.inv_alpha = use_sdf ? 1.0f / alpha[0] : 1.0f;
  P.beta = use_sdf ? beta[0] : 1.0f;
  P.neg_log2e_over_beta = -kLog2e / P.beta;
  P.lds = lds;
  P.vf = lds + n_image;
  return P;
}

}  // namespace nfi
