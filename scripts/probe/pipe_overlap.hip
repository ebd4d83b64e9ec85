// Probe: do the VALU and the matrix pipe of one SIMD overlap when they are fed by DIFFERENT waves?  Two waves per SIMD
// (workgroups of 8 waves, one per CU): waves 0-3 run role X, waves 4-7 role Y; a role is a loop of independent
// instructions of one kind.  Compare each role alone with the pair.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { IDLE = 0, MFMA_F32 = 1, MFMA_BF16 = 2, VALU_FMA = 3, VALU_INT = 4, VALU_FMAC = 5, VALU_MULC = 6, VALU_CVT = 7, VALU_ADD = 8 };

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed) {
  if (ROLE == MFMA_F32) {
    f32x16 c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 2.0f, c1, 0, 0, 0);
    }
    return c0[0] + c1[5];
  } else if (ROLE == MFMA_BF16) {
    union { bf16x8 v; unsigned u[4]; } a, b;
    for (int k = 0; k < 4; ++k) { a.u[k] = 0x3F803F80u; b.u[k] = __builtin_bit_cast(unsigned, seed) & 0x3F803F80u; }
    f32x16 c0 = {0}, c1 = {0};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.v, a.v, c1, 0, 0, 0);
    }
    return c0[0] + c1[5];
  } else if (ROLE == VALU_FMA) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
      F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
      F(a0); F(a1); F(a2); F(a3); F(a4); F(a5); F(a6); F(a7);
    }
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  } else if (ROLE == VALU_INT) {
    int a0 = (int)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const int c = 3;
    for (int i = 0; i < iters; ++i) {
#define G(x) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(c))
      G(a0); G(a1); G(a2); G(a3); G(a4); G(a5); G(a6); G(a7);
      G(a0); G(a1); G(a2); G(a3); G(a4); G(a5); G(a6); G(a7);
    }
    return (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
  }
  else if (ROLE == VALU_FMAC || ROLE == VALU_MULC || ROLE == VALU_CVT || ROLE == VALU_ADD) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const float c = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#define H2(x)                                                                                     \
  if (ROLE == VALU_FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %0" : "+v"(x) : "v"(c));            \
  else if (ROLE == VALU_MULC) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(x) : "v"(c));  \
  else if (ROLE == VALU_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));     \
  else asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(c));
      H2(a0); H2(a1); H2(a2); H2(a3); H2(a4); H2(a5); H2(a6); H2(a7);
      H2(a0); H2(a1); H2(a2); H2(a3); H2(a4); H2(a5); H2(a6); H2(a7);
    }
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
  return 0.0f;
}

template <int X, int Y>
__global__ void __launch_bounds__(512) k(float *out, int ix, int iy) {
  const int wave = threadIdx.x >> 6;
  float r = wave < 4 ? run_role<X>(ix, (float)threadIdx.x) : run_role<Y>(iy, (float)threadIdx.x);
  if (r == 12345.678f) out[0] = r;
}

template <int X, int Y>
float timed(float *d, int ix, int iy) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<X, Y>), dim3(256), dim3(512), 0, 0, d, ix, iy);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  float *d;
  (void)hipMalloc(&d, 1024);
  // iteration counts: each role alone ~ the same number of pipe cycles (2 x 64-cycle f32 MFMA = 128 cycles per
  // iteration; 2 x 32-cycle bf16 MFMA = 64; 16 VALU x 4 = 64)
  const int base = 200000;
  const int i_f32 = base, i_bf16 = 2 * base, i_valu = 2 * base;
  timed<VALU_FMA, VALU_FMA>(d, i_valu, i_valu);  // warm the clocks
  printf("f32 MFMA alone            %.2f ms\n", timed<MFMA_F32, IDLE>(d, i_f32, 0));
  printf("bf16 MFMA alone           %.2f ms\n", timed<MFMA_BF16, IDLE>(d, i_bf16, 0));
  printf("VALU fma alone            %.2f ms\n", timed<VALU_FMA, IDLE>(d, i_valu, 0));
  printf("VALU int alone            %.2f ms\n", timed<VALU_INT, IDLE>(d, i_valu, 0));
  printf("f32 MFMA  + VALU fma      %.2f ms\n", timed<MFMA_F32, VALU_FMA>(d, i_f32, i_valu));
  printf("f32 MFMA  + VALU int      %.2f ms\n", timed<MFMA_F32, VALU_INT>(d, i_f32, i_valu));
  printf("bf16 MFMA + VALU fma      %.2f ms\n", timed<MFMA_BF16, VALU_FMA>(d, i_bf16, i_valu));
  printf("bf16 MFMA + VALU int      %.2f ms\n", timed<MFMA_BF16, VALU_INT>(d, i_bf16, i_valu));
  printf("VALU fmac(vop2) alone     %.2f ms\n", timed<VALU_FMAC, IDLE>(d, i_valu, 0));
  printf("VALU mul clamp alone      %.2f ms\n", timed<VALU_MULC, IDLE>(d, i_valu, 0));
  printf("VALU cvt_pk_bf16 alone    %.2f ms\n", timed<VALU_CVT, IDLE>(d, i_valu, 0));
  printf("VALU add alone            %.2f ms\n", timed<VALU_ADD, IDLE>(d, i_valu, 0));
  printf("bf16 MFMA + fmac(vop2)    %.2f ms\n", timed<MFMA_BF16, VALU_FMAC>(d, i_bf16, i_valu));
  printf("bf16 MFMA + mul clamp     %.2f ms\n", timed<MFMA_BF16, VALU_MULC>(d, i_bf16, i_valu));
  printf("bf16 MFMA + cvt_pk_bf16   %.2f ms\n", timed<MFMA_BF16, VALU_CVT>(d, i_bf16, i_valu));
  printf("bf16 MFMA + add           %.2f ms\n", timed<MFMA_BF16, VALU_ADD>(d, i_bf16, i_valu));
  printf("f32 MFMA  + fmac(vop2)    %.2f ms\n", timed<MFMA_F32, VALU_FMAC>(d, i_f32, i_valu));
  printf("f32 MFMA  + add           %.2f ms\n", timed<MFMA_F32, VALU_ADD>(d, i_f32, i_valu));
  printf("VALU fma  + VALU fma      %.2f ms\n", timed<VALU_FMA, VALU_FMA>(d, i_valu, i_valu));
  printf("f32 MFMA  + f32 MFMA      %.2f ms\n", timed<MFMA_F32, MFMA_F32>(d, i_f32, i_f32));
  printf("bf16 MFMA + bf16 MFMA     %.2f ms\n", timed<MFMA_BF16, MFMA_BF16>(d, i_bf16, i_bf16));
  printf("f32 MFMA  + bf16 MFMA     %.2f ms\n", timed<MFMA_F32, MFMA_BF16>(d, i_f32, i_bf16));
  return 0;
}
