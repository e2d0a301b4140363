// probe: do v_mfma_f32_16x16x4_f32 (f32-input MFMA) and f32 VALU work from two waves of one SIMD overlap?
// and the same for v_mfma_f32_16x16x32_f16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: f32 mfma only, 1: valu only, 2: even waves f32 mfma / odd waves valu, 3: f16 mfma only, 4: even f16 mfma / odd valu
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (MODE == 0 || MODE == 3) || ((MODE == 2 || MODE == 4) && ((wave >> 2) & 1) == 0);
  const bool f16 = (MODE == 3 || MODE == 4);
  f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)0.5f; }
  if (do_mfma) {
    if (!f16) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[3], 0, 0, 0);
        }
      }
    } else {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[3], 0, 0, 0);
        }
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int MODE> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  int iters = 2000;
  printf("16 waves/CU (4 per SIMD; mixed = 2 mfma waves + 2 valu waves per SIMD), %d iters: 16 mfma per iter per wave | 256 v_fma per iter per wave\n", iters);
  printf("f32 mfma only (all 8 waves)      : %.3f ms\n", run<0>(d, iters));
  printf("valu only (all 8 waves)          : %.3f ms\n", run<1>(d, iters));
  printf("even waves f32 mfma, odd valu    : %.3f ms   (half the mfma + half the valu work)\n", run<2>(d, iters));
  printf("f16 mfma only (all 8 waves)      : %.3f ms\n", run<3>(d, iters));
  printf("even waves f16 mfma, odd valu    : %.3f ms\n", run<4>(d, iters));
  return 0;
}
