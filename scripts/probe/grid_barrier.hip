// Probe: cost of a device-wide barrier inside a persistent kernel on gfx950 (256 workgroups, one per CU), against the
// gap between two dependent kernels in one stream.  Every spin loop is bounded, so the kernel always exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Bar {
  unsigned int count;
  unsigned int gen;
  unsigned int timeout;
};

__device__ __forceinline__ bool grid_barrier(Bar *b, unsigned int n_blocks, unsigned int &local_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned int target = local_gen + 1;
    __threadfence();
    const unsigned int prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == n_blocks - 1) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned int spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {
          ok = false;
          atomicAdd(&b->timeout, 1u);
          break;
        }
      }
    }
    __threadfence();
  }
  local_gen += 1;
  __syncthreads();
  return ok;
}

__global__ void __launch_bounds__(768) k_persistent(Bar *b, int n_barriers, float *data, float *out) {
  unsigned int gen = 0;
  float acc = 0.0f;
  for (int i = 0; i < n_barriers; ++i) {
    // a little memory traffic that must be visible across the barrier
    if (threadIdx.x == 0) data[blockIdx.x] = (float)(i + 1);
    if (!grid_barrier(b, gridDim.x, gen)) break;
    acc += __builtin_nontemporal_load(&data[(blockIdx.x + 1) % gridDim.x]);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(768) k_tiny(float *data, int i) {
  if (threadIdx.x == 0) data[blockIdx.x] = data[(blockIdx.x + 1) % gridDim.x] + (float)i;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int nb = prop.multiProcessorCount;
  Bar *bar;
  float *data, *out;
  (void)hipMalloc(&bar, sizeof(Bar));
  (void)hipMalloc(&data, nb * sizeof(float));
  (void)hipMalloc(&out, nb * sizeof(float));
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int threads : {256, 768}) {
    for (int nbar : {1, 101, 1001}) {
      (void)hipMemsetAsync(bar, 0, sizeof(Bar), s);
      (void)hipMemsetAsync(data, 0, nb * sizeof(float), s);
      void *args[] = {&bar, &nbar, &data, &out};
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemsetAsync(bar, 0, sizeof(Bar), s);
        (void)hipEventRecord(e0, s);
        hipError_t rc = hipLaunchCooperativeKernel((void *)k_persistent, dim3(nb), dim3(threads), args, 0, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        if (rc != hipSuccess) printf("cooperative launch failed: %s\n", hipGetErrorString(rc));
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      Bar h;
      std::vector<float> ho(nb);
      (void)hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost);
      (void)hipMemcpy(ho.data(), out, nb * sizeof(float), hipMemcpyDeviceToHost);
      // expected: sum_{i=1..nbar} i
      double want = 0.5 * nbar * (nbar + 1.0);
      int bad = 0;
      for (int i = 0; i < nb; ++i) bad += ho[i] != (float)want;
      printf("persistent %d WGs x %d thr, %4d barriers: %.3f ms total, timeouts %u, wrong %d\n", nb, threads, nbar, ms,
             h.timeout, bad);
    }
  }
  // dependent tiny kernels in one stream
  for (int n : {100, 1000}) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0, s);
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_tiny, dim3(nb), dim3(768), 0, s, data, i);
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%d dependent tiny launches: %.3f ms = %.2f us each\n", n, ms, 1e3 * ms / n);
  }
  return 0;
}
