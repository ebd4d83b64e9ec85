// Measure f32 VALU issue rates on gfx950: v_fma_f32 vs v_pk_fma_f32, at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 1.0001f, d = 0.5f;
  const f32x2 c2 = {c, c}, d2 = {d, d};
  for (int i = 0; i < iters; ++i) {
    if (PK) {
#define P(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c2), "v"(d2))
      P(p0); P(p1); P(p2); P(p3); P(p4); P(p5); P(p6); P(p7);
      P(p0); P(p1); P(p2); P(p3); P(p4); P(p5); P(p6); P(p7);
    } else {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
      F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
      F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = PK ? p0[0] + p1[0] + p2[0] + p3[0] + p4[0] + p5[0] + p6[0] + p7[0] + p0[1]
                                                  : a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  float *d;
  (void)hipMalloc(&d, 256 * 8 * 2048 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk)
    for (int wps : {1, 2, 4, 8}) {
      dim3 grid(256 * wps), block(256);  // 256 threads = 4 waves = 1 per SIMD per block
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters);
        else hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      double instr_per_simd = (double)iters * 16 * wps;
      double ns_per_instr = ms * 1e6 / instr_per_simd;
      double flops = (double)iters * 16 * (pk ? 4 : 2) * 64.0 * 4 * wps * 256;
      printf("%s waves/SIMD=%d: %.3f ms, %.3f ns per wave-instr per SIMD, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, ns_per_instr, flops / (ms * 1e-3) / 1e12);
    }
  return 0;
}
