// probe: lane / element mapping of ds_read_b64_tr_b16 on gfx950.  LDS halves hold their own index; lane l supplies the
// address of halves [4l, 4l+4); printed: for every lane the four 16-bit values it receives.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  auto p = reinterpret_cast<__attribute__((address_space(3))) h4*>(
      (__attribute__((address_space(3))) uint16_t*)(lds + 4 * threadIdx.x));
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
  s4 s = __builtin_bit_cast(s4, v);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)s[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" %4d (lane %2d elem %d)", h[l * 4 + j], h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
