// Probe: does a buffer written by one kernel come back from the memory-side cache (Infinity Cache, 256 MB) when the next
// kernel reads it?  Write X MB, read X MB, time the read; X = 32 .. 1024 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(float4 *p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ void k_read(const float4 *p, size_t n, float *out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 1234.5f) out[0] = s;
}
int main() {
  float4 *buf;
  float *out;
  const size_t maxb = 1024ull << 20;
  (void)hipMalloc(&buf, maxb);
  (void)hipMalloc(&out, 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (size_t mb : {32, 64, 96, 128, 160, 192, 224, 256, 384, 512, 1024}) {
    const size_t n = (mb << 20) / 16;
    float best_r = 1e9f, best_w = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, buf, n, (float)rep);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float msw;
      (void)hipEventElapsedTime(&msw, e0, e1);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, n, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float msr;
      (void)hipEventElapsedTime(&msr, e0, e1);
      if (msr < best_r) best_r = msr;
      if (msw < best_w) best_w = msw;
    }
    printf("%5zu MB: write %.1f GB/s, read-after-write %.1f GB/s\n", mb, (double)(mb << 20) / best_w / 1e6,
           (double)(mb << 20) / best_r / 1e6);
  }
  return 0;
}
