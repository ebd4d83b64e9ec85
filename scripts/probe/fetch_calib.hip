// Calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for 4-byte-per-lane coalesced streams (the access shape
// of the update kernels): reads a known number of bytes from a buffer much larger than the 256 MiB Infinity Cache
// and writes a known number of bytes.  Run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read_dword(const float *__restrict__ in, float *__restrict__ out, size_t n) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_write_dword(float *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0f;
}
int main() {
  const size_t n = (size_t)1 << 28;  // 2^28 floats = 1 GiB
  float *a, *b;
  (void)hipMalloc(&a, n * sizeof(float));
  (void)hipMalloc(&b, n * sizeof(float));
  (void)hipMemset(a, 0, n * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_read_dword, dim3(2048), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_write_dword, dim3(2048), dim3(256), 0, 0, b, n);
  }
  (void)hipDeviceSynchronize();
  printf("bytes per launch: %zu\n", n * sizeof(float));
  return 0;
}
