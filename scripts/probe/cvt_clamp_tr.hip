// Probe (round 3): two hardware facts the layer-2-on-the-matrix-pipe step of the fused update kernels rests on.
//  (1) v_cvt_pk_bf16_f32 with the VOP3 clamp bit: does it clamp each converted half to [0, 1]?  (Then relu'(pre) of a
//      pre-activation scaled by 2^96 is ONE instruction per two values: 1 for pre >= 2^-96, 0 for pre <= 0.)
//  (2) ds_read_b64_tr_b16 on the XOR-swizzled [hidden unit][sample] mask image (64-byte rows, 8-byte chunk c of row j at
//      chunk c ^ ((j >> 1) & 7)): does lane (sample n, half hh) receive mask[j = 16 ks + 8 hh + e][n] in element e?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probe/cvt_clamp_tr.hip -o scripts/probe/cvt_clamp_tr
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k_cvt(const float *in, uint32_t *out, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  uint32_t r, r2;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(in[i]), "v"(in[(i + 1) % n]));
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r2) : "v"(in[i]), "v"(in[(i + 1) % n]));
  out[i] = r;
  out[n + i] = r2;
}

// one wave: lane (j = lane & 31, hf = lane >> 5) owns the packed masks of hidden unit j for the samples
// 8 g + 4 hf + 0..3, g = 0..3 (the accumulator-tile ownership of v_mfma_f32_32x32x16_bf16); value = 100 j + sample
__global__ void k_tr(unsigned short *out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[32 * 32];
  const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
  for (int g = 0; g < 4; ++g) {
    const int ch = (2 * g + hf) ^ ((j >> 1) & 7);
    unsigned short *dst = &img[j * 32 + 4 * ch];
    for (int e = 0; e < 4; ++e) dst[e] = (unsigned short)(100 * j + 8 * g + 4 * hf + e);
  }
  __syncthreads();
  const int i = lane & 15, g4 = lane >> 4, hh = g4 >> 1, q = i >> 2, p = i & 3;
  for (int ks = 0; ks < 2; ++ks)
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 16 * ks + 8 * hh + 4 * rr + q;
      const int ch = (4 * (g4 & 1) + p) ^ ((row >> 1) & 7);
      const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (s16x4 __attribute__((address_space(3))) *)(img + row * 32 + 4 * ch));
      for (int e = 0; e < 4; ++e) out[((ks * 2 + rr) * 64 + lane) * 4 + e] = (unsigned short)v[e];
    }
}

int main() {
  const float vals[] = {-1.0f, 0.0f, -0.0f, 0.5f, 1.0f, 2.0f, 1e-30f, 1e30f, INFINITY, -INFINITY, NAN, 1e-40f, 0.99f,
                        3.0e-39f, 0x1p-96f, 0x1p32f};
  const int n = sizeof(vals) / sizeof(vals[0]);
  float *din;
  uint32_t *dout, hout[64];
  (void)hipMalloc(&din, sizeof(vals));
  (void)hipMalloc(&dout, sizeof(hout));
  (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dout, n);
  (void)hipMemcpy(hout, dout, sizeof(uint32_t) * 2 * n, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_bf16_f32 clamp (low half = first source):\n");
  for (int i = 0; i < n; ++i) {
    uint32_t lo = hout[i] & 0xffff, lo2 = hout[n + i] & 0xffff, hi = hout[i] >> 16;
    uint32_t bits = lo << 16, bits2 = lo2 << 16, bitsh = hi << 16;
    float f, f2, fh;
    memcpy(&f, &bits, 4);
    memcpy(&f2, &bits2, 4);
    memcpy(&fh, &bitsh, 4);
    printf("  in % .6g -> clamp %g (0x%04x)   plain %g   [high half of the clamped pair: %g]\n", vals[i], f, lo, f2, fh);
  }
  unsigned short *dtr, htr[4 * 64 * 4];
  (void)hipMalloc(&dtr, sizeof(htr));
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, dtr);
  (void)hipMemcpy(htr, dtr, sizeof(htr), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int ks = 0; ks < 2; ++ks)
    for (int rr = 0; rr < 2; ++rr)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
          const int nn = lane & 31, hh = lane >> 5;
          const int want = 100 * (16 * ks + 8 * hh + 4 * rr + e) + nn;
          const int got = htr[((ks * 2 + rr) * 64 + lane) * 4 + e];
          if (got != want) {
            if (bad < 8) printf("  tr mismatch ks %d rr %d lane %d e %d: got %d want %d\n", ks, rr, lane, e, got, want);
            ++bad;
          }
        }
  printf("ds_read_b64_tr_b16 on the swizzled mask image: %s (%d mismatches)\n", bad ? "WRONG" : "as expected", bad);
  return bad != 0;
}
