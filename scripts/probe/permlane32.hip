// Probe: what v_permlane32_swap_b32 does to (vdst, vsrc) = (a, b), lane by lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned *d, h[128];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 = %u, r1 = %u\n", l, h[l], h[64 + l]);
  return 0;
}
