// v_mfma_f32_16x16x4_f32: lane maps and the arithmetic of the four k-products of one instruction / of a chain.
// Hypothesis: A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15], C reg i of lane l = C[4 (l >> 4) + i][l & 15],
// c = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) — i.e. a sequential fma chain over k ascending.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KS = 8;  // K = 32
__global__ void k(const float *A, const float *B, const float *C0, float *out) {
  const int l = threadIdx.x, n = l & 15, g = l >> 4;
  f32x4 c;
  for (int i = 0; i < 4; ++i) c[i] = C0[(4 * g + i) * 16 + n];
  for (int ks = 0; ks < KS; ++ks) {
    const float a = A[n * (4 * KS) + 4 * ks + g];
    const float b = B[(4 * ks + g) * 16 + n];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) out[(4 * g + i) * 16 + n] = c[i];
}
static float frand() { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0) * (1.0f + (rand() % 7)); }
int main() {
  const int K = 4 * KS;
  static float A[16 * 4 * KS], B[4 * KS * 16], C0[256], got[256];
  srand(777);
  for (auto &v : A) v = frand();
  for (auto &v : B) v = frand();
  for (auto &v : C0) v = frand();
  float *dA, *dB, *dC, *dO;
  (void)hipMalloc(&dA, sizeof(A)); (void)hipMalloc(&dB, sizeof(B)); (void)hipMalloc(&dC, sizeof(C0)); (void)hipMalloc(&dO, sizeof(got));
  (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
  (void)hipMemcpy(dC, C0, sizeof(C0), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
  (void)hipMemcpy(got, dO, sizeof(got), hipMemcpyDeviceToHost);
  int bad_seq = 0, bad_rev = 0, bad_one = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      float cs = C0[m * 16 + n], cr = cs, co = cs;
      for (int ks = 0; ks < KS; ++ks) {
        for (int q = 0; q < 4; ++q) cs = fmaf(A[m * K + 4 * ks + q], B[(4 * ks + q) * 16 + n], cs);
        for (int q = 3; q >= 0; --q) cr = fmaf(A[m * K + 4 * ks + q], B[(4 * ks + q) * 16 + n], cr);
        double acc = co;
        for (int q = 0; q < 4; ++q) acc += (double)A[m * K + 4 * ks + q] * B[(4 * ks + q) * 16 + n];
        co = (float)acc;
      }
      const float gv = got[m * 16 + n];
      bad_seq += gv != cs; bad_rev += gv != cr; bad_one += gv != co;
    }
  printf("mismatches of 256: sequential fma (k ascending) %d, reversed %d, single rounding %d\n", bad_seq, bad_rev, bad_one);
  return 0;
}
