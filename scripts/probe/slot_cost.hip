// Probe (round 3): cycles per "slot" = one v_mfma_f32_32x32x16_bf16 followed by K independent vector instructions and
// N LDS instructions in ONE wave's in-order stream, with the accumulator in VGPRs or in AGPRs, one or two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probe/slot_cost.hip -o scripts/probe/slot_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool AGPR, int K, int NW, int NR, int KIND>
__global__ void __launch_bounds__(512) k(float *out, int iters, int waves_active) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8][8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves_active) return;
  f32x4 a = {1.0f, 1.0f, 1.0f, 1.0f}, b = {(float)lane, 0.0f, 1.0f, 0.0f};
  f32x16 c = {0};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float one = 1.0f;
  unsigned long long w0 = 0x3f803f803f803f80ull, r0 = 0, r1 = 0;
  const unsigned addr = (unsigned)(size_t)(&lds[wave][0]) + lane * 8;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (KIND == 0) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(v[j % 8]) : "v"(one));
        else if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j % 8]) : "v"(one));
        else asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(v[j % 8]) : "v"(one));
      }
#pragma unroll
      for (int j = 0; j < NW; ++j) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(w0), "n"(512 * j));
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if (j & 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r1) : "v"(addr), "n"(512 * j));
        else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r0) : "v"(addr), "n"(512 * j));
      }
    }
    if (NW + NR) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = c[0] + c[5];
  for (int j = 0; j < 8; ++j) s += v[j];
  s += (float)(r0 + r1);
  if (s == 12345.678f) out[0] = s;
}

template <bool AGPR, int K, int NW, int NR, int KIND>
void run(float *d, int waves, const char *what) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 50000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<AGPR, K, NW, NR, KIND>), dim3(256), dim3(512), 0, 0, d, iters, waves);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double ns = ms * 1e6 / ((double)iters * 4);
  printf("%-6s acc, %d %s + %d ds_write_b64 + %d ds_read_tr per MFMA, %d wave(s)/SIMD: %6.1f ns per slot and wave stream, %6.1f per MFMA of the SIMD\n",
         AGPR ? "AGPR" : "VGPR", K, what, NW, NR, waves / 4, ns, ns / (waves / 4));
}

int main() {
  float *d;
  (void)hipMalloc(&d, 1024);
  run<false, 0, 0, 0, 0>(d, 4, "v_add");  // warm
  run<false, 0, 0, 0, 0>(d, 4, "v_add");
  run<false, 2, 0, 0, 0>(d, 4, "v_add");
  run<false, 4, 0, 0, 0>(d, 4, "v_add");
  run<false, 6, 0, 0, 0>(d, 4, "v_add");
  run<false, 8, 0, 0, 0>(d, 4, "v_add");
  run<false, 12, 0, 0, 0>(d, 4, "v_add");
  run<true, 0, 0, 0, 0>(d, 4, "v_add");
  run<true, 4, 0, 0, 0>(d, 4, "v_add");
  run<true, 6, 0, 0, 0>(d, 4, "v_add");
  run<true, 8, 0, 0, 0>(d, 4, "v_add");
  run<true, 12, 0, 0, 0>(d, 4, "v_add");
  run<false, 6, 0, 0, 1>(d, 4, "v_cvt_pk");
  run<false, 6, 0, 0, 2>(d, 4, "v_lshl_or");
  run<false, 0, 2, 0, 0>(d, 4, "v_add");
  run<false, 0, 0, 2, 0>(d, 4, "v_add");
  run<false, 0, 2, 2, 0>(d, 4, "v_add");
  run<false, 4, 1, 1, 0>(d, 4, "v_add");
  run<true, 4, 1, 1, 0>(d, 4, "v_add");
  // two waves per SIMD, identical streams
  run<false, 0, 0, 0, 0>(d, 8, "v_add");
  run<false, 4, 0, 0, 0>(d, 8, "v_add");
  run<false, 8, 0, 0, 0>(d, 8, "v_add");
  run<false, 12, 0, 0, 0>(d, 8, "v_add");
  run<true, 8, 0, 0, 0>(d, 8, "v_add");
  run<false, 4, 1, 1, 0>(d, 8, "v_add");
  run<false, 8, 2, 2, 0>(d, 8, "v_add");
  run<true, 8, 2, 2, 0>(d, 8, "v_add");
  return 0;
}
