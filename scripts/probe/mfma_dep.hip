// Probe (round 3): issue cost of DEPENDENT v_mfma_f32_32x32x16_bf16 (accumulator of one instruction = srcC of the
// next) against independent accumulators, with one and with two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma_dep.hip -o scripts/probe/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(512) k(float *out, int iters, int waves_active) {
  const int wave = threadIdx.x >> 6;
  if (wave >= waves_active) return;
  union { bf16x8 v; unsigned u[4]; } a, b;
  for (int i = 0; i < 4; ++i) { a.u[i] = 0x3F803F80u; b.u[i] = (threadIdx.x & 1) ? 0x3F803F80u : 0u; }
  f32x16 c[NACC];
  for (int n = 0; n < NACC; ++n) c[n] = (f32x16){0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < NACC; ++n) s += c[n][0] + c[n][7];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(float *d, int waves) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 100000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(512), 0, 0, d, iters, waves);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double per = ms * 1e6 / ((double)iters * 8 * (waves / 4));
  printf("%d accumulator chain(s), %d wave(s) per SIMD: %.2f ms, %.1f ns per MFMA and SIMD (32 cycles at 2.1 GHz = 15.2 ns)\n",
         NACC, waves / 4, ms, per);
}

int main() {
  float *d;
  (void)hipMalloc(&d, 1024);
  run<4>(d, 4);
  for (int w : {4, 8}) {
    run<1>(d, w);
    run<2>(d, w);
    run<4>(d, w);
  }
  return 0;
}
