// Probe: fp32 atomic scatter-add throughput on gfx950 as a function of how many dwords of one
// 128-B texel (32 channels) a single wave instruction covers.  Mirrors the plane-gradient scatter
// of field_query_bwd_kernel: T texel updates per wave, each texel = 32 consecutive floats.
//   mode 0: quad layout   - 16 texels/instr, lane 4p+q adds channel 4q+s (8 instrs s=0..7 -> stride 4... see code)
//   mode 1: half layout   - 2 texels/instr, lane 32h+c adds channel c
//   mode 2: mode 1 with plain stores (upper bound of the write path)
//   mode 3: mode 0 with plain stores
// build: hipcc --offload-arch=gfx950 -O3 atomic_scatter.hip -o atomic_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void scatter(float* __restrict__ img, unsigned n_texels, int rounds, int local) {
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  for (int r = 0; r < rounds; ++r) {
    // 16 texels per round per wave; `local` keeps them within a 2x2-ish neighbourhood stream
    if (MODE == 0 || MODE == 3) {
      unsigned p = lane >> 2, q = lane & 3;
      unsigned t = local ? (hash32(wave * 977u + r / 4) + p * 3u + (r & 3)) % n_texels
                         : hash32((wave * 4096u + r) * 16u + p) % n_texels;
      float* dst = img + (size_t)t * 32 + q * 8;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (MODE == 0) unsafeAtomicAdd(dst + s, 1.0f); else dst[s] = 1.0f;
      }
    } else {
      unsigned h = lane >> 5, c = lane & 31;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        unsigned p = s * 2 + h;
        unsigned t = local ? (hash32(wave * 977u + r / 4) + p * 3u + (r & 3)) % n_texels
                           : hash32((wave * 4096u + r) * 16u + p) % n_texels;
        float* dst = img + (size_t)t * 32 + c;
        if (MODE == 1) unsafeAtomicAdd(dst, 1.0f); else *dst = 1.0f;
      }
    }
  }
}

int main(int argc, char** argv) {
  // default: B=4 scenes x 3 planes x 256^2 texels (100 MB, far beyond the 8 x 4 MB L2); pass a small
  // count (e.g. 4096 = 512 KB) to see the L2-resident atomic rate
  const unsigned n_texels = argc > 1 ? (unsigned)atoi(argv[1]) : 4u * 3u * 256u * 256u;
  float* img; hipMalloc(&img, (size_t)n_texels * 32 * 4);
  hipMemset(img, 0, (size_t)n_texels * 32 * 4);
  const int blocks = 256 * 8, rounds = 768;          // 8192 waves x 768 x 16 texel updates = 100.7M texel updates
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[4] = {"atomic quad(16 texels/instr)", "atomic half(2 texels/instr)", "store  half", "store  quad"};
  for (int local = 0; local < 2; ++local)
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        if (mode == 0) scatter<0><<<blocks, 256>>>(img, n_texels, rounds, local);
        if (mode == 1) scatter<1><<<blocks, 256>>>(img, n_texels, rounds, local);
        if (mode == 2) scatter<2><<<blocks, 256>>>(img, n_texels, rounds, local);
        if (mode == 3) scatter<3><<<blocks, 256>>>(img, n_texels, rounds, local);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      double updates = (double)blocks * 4 * rounds * 16;
      printf("local=%d %-30s %8.3f ms  %.2f G texel-updates/s  %.1f G dword/s\n", local, names[mode], best,
             updates / best * 1e-6, updates * 32 / best * 1e-6);
    }
  return 0;
}
