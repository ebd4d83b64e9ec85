// What does one optimisation step cost inside a persistent kernel?  256 workgroups (one per CU), per iteration: every
// workgroup writes a 642-double slab row, grid barrier, every workgroup reduces 3 columns over the 256 rows and writes
// 3 parameters, grid barrier, every workgroup reads the 642 parameters.   build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool grid_sync(unsigned *counter, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_readcyclecounter() - t0 > 4000000000ull) { ok = false; break; }
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

template <int MODE>  // 0: barriers only; 1: + slab write / column reduce / param exchange
__global__ void __launch_bounds__(512) k_loop(unsigned *counter, double *slab, float *params, int iters, float *out) {
  const int P = 642, nwg = gridDim.x;
  __shared__ float ps[648];
  float acc = 0.0f;
  unsigned target = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
      double *row = slab + ((size_t)(it & 1) * nwg + blockIdx.x) * P;
      for (int p = threadIdx.x; p < P; p += 512) row[p] = (double)(ps[p] + it);
    }
    target += nwg;
    if (!grid_sync(counter, target)) return;
    if (MODE == 1) {
      // columns 3 b .. 3 b + 2: 16 chains x 16 rows each, as k_reduce_adam sums them
      const int c = threadIdx.x >> 4, w = threadIdx.x & 15, col = blockIdx.x * 3 + c;
      if (c < 3 && col < P) {
        const double *base = slab + (size_t)(it & 1) * nwg * P + col;
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (w + 16 * u) < nwg ? base[(size_t)(w + 16 * u) * P] : 0.0;
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) a = a + v[u];
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t = t + __shfl(a, (threadIdx.x & ~15) + k, 64);
        if (w == 0) params[(size_t)((it + 1) & 1) * 648 + col] = (float)t * 1e-9f;
      }
    }
    target += nwg;
    if (!grid_sync(counter, target)) return;
    if (MODE == 1) {
      for (int p = threadIdx.x; p < P; p += 512) ps[p] = params[(size_t)((it + 1) & 1) * 648 + p];
      __syncthreads();
      acc += ps[threadIdx.x % P];
    }
  }
  if (out) out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
  const int nwg = 256, iters = 200;
  unsigned *counter; double *slab; float *params, *out;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&slab, sizeof(double) * 2 * nwg * 642)); CK(hipMalloc(&params, 4 * 2 * 648));
  CK(hipMalloc(&out, 4 * nwg * 512));
  CK(hipMemset(params, 0, 4 * 2 * 648));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(counter, 0, 4));
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k_loop<0>, dim3(nwg), dim3(512), 0, 0, counter, slab, params, iters, out);
      else hipLaunchKernelGGL(k_loop<1>, dim3(nwg), dim3(512), 0, 0, counter, slab, params, iters, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      unsigned c; CK(hipMemcpy(&c, counter, 4, hipMemcpyDeviceToHost));
      printf("mode %d: %.2f us per iteration (two barriers%s), counter %u of %u\n", mode, ms * 1e3 / iters,
             mode ? " + slab write + 3-column reduce + parameter exchange" : "", c, 2u * nwg * iters);
    }
  return 0;
}
