// Probe the operand / result lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950 with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
  int l = threadIdx.x;
  float a = (float)(1 + l);          // A value of lane l
  float b = (float)(100 * (1 + l));  // B value of lane l
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // hypothesis: D[lane l][reg r] = A[lane 4*(l/4) + r] * B[lane l]
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      float want = (float)(1 + 4 * (l / 4) + r) * (float)(100 * (1 + l));
      if (h[l * 4 + r] != want) { ok = 0; if (l < 8) printf("lane %d reg %d got %g want %g\n", l, r, h[l*4+r], want); }
    }
  printf("hypothesis D[l][r] = A[4*(l/4)+r] * B[l]: %s\n", ok ? "OK" : "MISMATCH");
  for (int l = 0; l < 6; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
