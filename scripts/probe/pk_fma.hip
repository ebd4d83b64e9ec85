// pk_fma.hip — does v_pk_fma_f32 double the f32 fma rate of the vector ALU on gfx950?  Eight independent accumulator
// chains per lane, scalar (v_fma_f32) against packed (v_pk_fma_f32, two floats per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/pk_fma scripts/probe/pk_fma.hip && ./scripts/probe/pk_fma
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

__global__ void __launch_bounds__(256) k_scalar(float *out, float a, float b) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_packed(float *out, float a, float b) {
  f32x2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x2){(float)(threadIdx.x + 2 * i), (float)(threadIdx.x + 2 * i + 1)};
  const f32x2 a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a2), "v"(b2));
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *d;
  hipMalloc(&d, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_scalar, dim3(4096), dim3(256), 0, 0, d, 0.999f, 0.001f);
      else hipLaunchKernelGGL(k_packed, dim3(4096), dim3(256), 0, 0, d, 0.999f, 0.001f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double fma = 4096.0 * 256 * 16 * ITER;
      if (rep == 2) printf("%s: %.3f ms, %.1f T fma/s (%.1f TFLOP/s)\n", which == 0 ? "v_fma_f32   " : "v_pk_fma_f32", ms, fma / ms / 1e9, 2 * fma / ms / 1e9);
    }
  }
  return 0;
}
