// Probe: v_mfma_f32_16x16x32_bf16 on gfx950 — operand lane maps and issue cost.
//   assumed: A[m = l & 15][k = 8 (l >> 4) + i], B[k = 8 (l >> 4) + i][n = l & 15], C reg i of lane l = C[4 (l >> 4) + i][l & 15]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline uint16_t f2bf(float f) {  // exact for the small integers used here
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)(u >> 16);
}

__global__ void k_map(const float *A, const float *B, float *C) {  // A [16][32], B [32][16], C [16][16]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  union { bf16x8 v; uint16_t h[8]; } a, b;
  for (int i = 0; i < 8; ++i) {
    a.h[i] = f2bf(A[r * 32 + 8 * g + i]);
    b.h[i] = f2bf(B[(8 * g + i) * 16 + r]);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = c[i];
}

template <int MODE>  // 0: MFMA only, 1: VALU only, 2: both in one wave (interleaved), 3: MFMA chain dependent
__global__ void __launch_bounds__(256) k_time(float *out, int iters) {
  union { bf16x8 v; uint32_t u[4]; } a, b;
  for (int k = 0; k < 4; ++k) { a.u[k] = 0x3F803F80u; b.u[k] = 0x3F803F80u; }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const float m = 1.0001f;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c3, 0, 0, 0);
    }
    if (MODE == 3) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c0, 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#define V(x) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "v"(m))
      V(x0); V(x1); V(x2); V(x3); V(x0); V(x1); V(x2); V(x3);
      V(x0); V(x1); V(x2); V(x3); V(x0); V(x1); V(x2); V(x3);
    }
  }
  float r = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3;
  if (r == 1234.5f) out[0] = r;
}

template <int MODE>
float timed(float *d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_time<MODE>, dim3(256), dim3(256), 0, 0, d, iters);  // one wave per SIMD
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  float hA[16 * 32], hB[32 * 16], hC[256], ref[256];
  for (int i = 0; i < 16 * 32; ++i) hA[i] = (float)((i * 7 + 3) % 11 - 5);
  for (int i = 0; i < 32 * 16; ++i) hB[i] = (float)((i * 5 + 1) % 13 - 6);
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      float s = 0;
      for (int k = 0; k < 32; ++k) s += hA[m * 32 + k] * hB[k * 16 + n];
      ref[m * 16 + n] = s;
    }
  float *dA, *dB, *dC;
  (void)hipMalloc(&dA, sizeof(hA));
  (void)hipMalloc(&dB, sizeof(hB));
  (void)hipMalloc(&dC, sizeof(hC));
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  (void)hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += hC[i] != ref[i];
  printf("lane map: %d of 256 elements differ from the host product\n", bad);
  const int iters = 200000;
  const double clk = 2.4e6;  // cycles per ms at 2.4 GHz (nominal)
  float t0 = timed<0>(dC, iters), t1 = timed<1>(dC, iters), t2 = timed<2>(dC, iters), t3 = timed<3>(dC, iters);
  printf("4 independent MFMAs per iteration  %.2f ms  = %.1f cycles per MFMA\n", t0, t0 * clk / iters / 4);
  printf("16 VALU mul per iteration          %.2f ms  = %.1f cycles per op\n", t1, t1 * clk / iters / 16);
  printf("both, one wave                     %.2f ms  (sum %.2f)\n", t2, t0 + t1);
  printf("4 dependent MFMAs per iteration    %.2f ms  = %.1f cycles per MFMA\n", t3, t3 * clk / iters / 4);
  return 0;
}
