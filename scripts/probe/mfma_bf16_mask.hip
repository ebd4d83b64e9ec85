// Probe for the masked-sum backward on the bf16 matrix pipe:  M[j][k] = sum_s g[s][j] * u[s][k]  with g in {0, 1}
// (relu') and u an arbitrary f32 split EXACTLY into three bf16 pieces u = hi + mid + lo.  Every product g * piece is
// exact, so the only roundings are the f32 accumulations inside v_mfma_f32_32x32x16_bf16.  Checks, on one wave:
//   (1) the operand maps (accumulator-tile-as-A idiom, guide §3): X's column (hidden unit) on the lane, rows (samples)
//       in the registers; B element j of lane half h = u[sample 16 s + 8 (j >> 2) + 4 h + (j & 3)][col];
//   (2) the split is exact;
//   (3) the accumulated error over T tiles against an f64 sum, next to a sequential f32 fma chain's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __host__ inline unsigned short bf16_rne(float f) {  // round to nearest even (finite inputs)
  unsigned u;
  __builtin_memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __host__ inline float bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}

// g: [tiles][32 samples][32 hidden] bytes (0/1); u: [tiles][32 samples][6] f32; out: [32 hidden][32 cols] f32
__global__ void __launch_bounds__(64) k_probe(const unsigned char *g, const float *u, int tiles, float *out) {
  __shared__ unsigned short ub[32][36];  // [col][sample], 72-byte rows
  const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
  f32x16 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = lane; c < 32 * 36; c += 64) (&ub[0][0])[c] = 0;
  for (int t = 0; t < tiles; ++t) {
    __syncthreads();
    // pieces of u for sample n: lane half h handles k = h, 2 + h, 4 + h
    for (int q = 0; q < 3; ++q) {
      const int k = 2 * q + h;
      const float v = u[((size_t)t * 32 + n) * 6 + k];
      const unsigned short hi = bf16_rne(v);
      const float r1 = v - bf16_to_f32(hi);
      const unsigned short mid = bf16_rne(r1);
      const float r2 = r1 - bf16_to_f32(mid);
      const unsigned short lo = bf16_rne(r2);
      ub[3 * k + 0][n] = hi;
      ub[3 * k + 1][n] = mid;
      ub[3 * k + 2][n] = lo;
    }
    __syncthreads();
    for (int s = 0; s < 2; ++s) {
      union { bf16x8 v; unsigned short e[8]; } a, b;
      for (int j = 0; j < 8; ++j) {
        const int sample = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        a.e[j] = g[((size_t)t * 32 + sample) * 32 + n] ? 0x3F80 : 0;  // A: mask of hidden unit n for that sample
        b.e[j] = ub[n][sample];                                        // B: piece column n of that sample
      }
      d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, d, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = d[r];  // [hidden][col]
}

int main() {
  for (int tiles : {1, 8, 32}) {
    const int S = tiles * 32;
    std::vector<unsigned char> g((size_t)S * 32);
    std::vector<float> u((size_t)S * 6);
    srand(7 + tiles);
    for (auto &x : g) x = (rand() & 3) != 0;
    for (size_t i = 0; i < u.size(); ++i) {
      double mag = std::pow(10.0, -7.0 + 3.0 * (rand() / (double)RAND_MAX));
      u[i] = (float)(mag * (2.0 * (rand() / (double)RAND_MAX) - 1.0));
    }
    // exactness of the split
    int inexact = 0;
    for (float v : u) {
      unsigned short hi = bf16_rne(v);
      float r1 = v - bf16_to_f32(hi);
      unsigned short mid = bf16_rne(r1);
      float r2 = r1 - bf16_to_f32(mid);
      unsigned short lo = bf16_rne(r2);
      if ((double)bf16_to_f32(hi) + (double)bf16_to_f32(mid) + (double)bf16_to_f32(lo) != (double)v) inexact++;
    }
    unsigned char *dg;
    float *du, *dout;
    (void)hipMalloc(&dg, g.size());
    (void)hipMalloc(&du, u.size() * 4);
    (void)hipMalloc(&dout, 32 * 32 * 4);
    (void)hipMemcpy(dg, g.data(), g.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(du, u.data(), u.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dg, du, tiles, dout);
    std::vector<float> out(32 * 32);
    (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    double worst_mfma = 0, worst_chain = 0, scale = 0;
    int wrong_layout = 0;
    for (int j = 0; j < 32; ++j)
      for (int k = 0; k < 6; ++k) {
        double ref = 0, absum = 0;
        float chain = 0.0f;
        for (int s = 0; s < S; ++s)
          if (g[(size_t)s * 32 + j]) {
            ref += (double)u[(size_t)s * 6 + k];
            absum += std::fabs((double)u[(size_t)s * 6 + k]);
            chain = fmaf(1.0f, u[(size_t)s * 6 + k], chain);
          }
        float got = (out[j * 32 + 3 * k] + out[j * 32 + 3 * k + 1]) + out[j * 32 + 3 * k + 2];
        double e1 = std::fabs((double)got - ref) / absum, e2 = std::fabs((double)chain - ref) / absum;
        if (e1 > 1e-4) wrong_layout++;
        if (e1 > worst_mfma) worst_mfma = e1;
        if (e2 > worst_chain) worst_chain = e2;
        scale = absum;
      }
    int nonzero_pad = 0;
    for (int j = 0; j < 32; ++j)
      for (int c = 18; c < 32; ++c) nonzero_pad += out[j * 32 + c] != 0.0f;
    printf("tiles %2d (%4d samples): inexact splits %d, layout errors %d, padded columns nonzero %d, max |err| / sum|u|: "
           "bf16x3 MFMA %.3g, f32 fma chain %.3g (f32 eps 6e-8)\n", tiles, S, inexact, wrong_layout, nonzero_pad, worst_mfma,
           worst_chain);
    (void)scale;
  }
  return 0;
}
