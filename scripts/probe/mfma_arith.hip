// Which arithmetic does v_mfma_f32_32x32x2_f32 implement for the two k-products of one instruction and for a
// chain of instructions?  Candidates, evaluated on the host for random data and compared bit for bit:
//   (a) sequential fma:  c = fma(a1, b1, fma(a0, b0, c))
//   (b) reversed fma:    c = fma(a0, b0, fma(a1, b1, c))
//   (c) one rounding:    c = (float)((double)c + (double)a0*b0 + (double)a1*b1)
//   (d) products rounded, then added: c = (c + a0*b0) + a1*b1 with every op rounded
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KS = 8;  // chained instructions (K = 16)
__global__ void k(const float *A, const float *B, const float *C0, float *out) {
  // A: [32 m][2*KS k], B: [2*KS k][32 n], C0: [32][32]
  int l = threadIdx.x, n = l & 31, hf = l >> 5;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C0[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + n];
  for (int ks = 0; ks < KS; ++ks) {
    float a = A[n * (2 * KS) + 2 * ks + hf];   // A operand: row m = lane & 31, k = lane >> 5
    float b = B[(2 * ks + hf) * 32 + n];       // B operand: k = lane >> 5, col n = lane & 31
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + n] = c[r];
}
static float frand() { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0) * (1.0f + (rand() % 7)); }
int main() {
  const int K = 2 * KS;
  static float A[32 * 2 * KS], B[2 * KS * 32], C0[1024], got[1024];
  srand(12345);
  for (auto &v : A) v = frand();
  for (auto &v : B) v = frand();
  for (auto &v : C0) v = frand();
  float *dA, *dB, *dC, *dO;
  hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dC, sizeof(C0)); hipMalloc(&dO, sizeof(got));
  hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
  hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
  hipMemcpy(dC, C0, sizeof(C0), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
  hipMemcpy(got, dO, sizeof(got), hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      float ca = C0[m * 32 + n], cb = ca, cc = ca, cd = ca;
      for (int ks = 0; ks < KS; ++ks) {
        float a0 = A[m * K + 2 * ks], a1 = A[m * K + 2 * ks + 1], b0 = B[(2 * ks) * 32 + n], b1 = B[(2 * ks + 1) * 32 + n];
        ca = fmaf(a1, b1, fmaf(a0, b0, ca));
        cb = fmaf(a0, b0, fmaf(a1, b1, cb));
        cc = (float)((double)cc + (double)a0 * b0 + (double)a1 * b1);
        volatile float p0 = a0 * b0, p1 = a1 * b1;
        volatile float s = cd + p0;
        cd = s + p1;
      }
      float g = got[m * 32 + n];
      bad[0] += g != ca; bad[1] += g != cb; bad[2] += g != cc; bad[3] += g != cd;
    }
  printf("mismatches of 1024: (a) sequential fma %d, (b) reversed fma %d, (c) single rounding %d, (d) rounded products %d\n",
         bad[0], bad[1], bad[2], bad[3]);
  return 0;
}
