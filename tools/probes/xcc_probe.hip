// probe: does s_getreg(HW_REG_XCC_ID) identify the XCD, and how do block ids map to XCDs?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    unsigned v = __builtin_amdgcn_s_getreg(6164);   // hwreg(20, 0, 4)
    unsigned full = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[blockIdx.x * 2] = (int)v; out[blockIdx.x * 2 + 1] = (int)full;
  }
}
int main() {
  int n = 64; int* d; hipMalloc(&d, n * 2 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%d:%d(%x) ", i, h[2 * i], h[2 * i + 1]);
  printf("\n");
  return 0;
}
