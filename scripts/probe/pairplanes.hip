#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int GH = 128, TL = 32;
struct PairPlanes {
  float p[2][TL][GH / 2 + 4];
  __device__ __forceinline__ void put(int k, int m, float v) { p[k & 1][m][k >> 1] = v; }
  __device__ __forceinline__ float4 get4(int hf, int m, int q) const {
    return *reinterpret_cast<const float4 *>(&p[hf][m][4 * q]);
  }
};
struct Sh { PairPlanes hP[2]; float other[33]; };
__global__ void k(int *bad) {
  __shared__ Sh sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  for (int r = 0; r < 16; ++r) { int m = (r & 3) + 8 * (r >> 2) + 4 * hf; sh.hP[1].put(j, m, (float)(j * 100 + m)); }
  __syncthreads();
  int b = 0;
  for (int q = 0; q < 16; ++q) {
    float4 a4 = sh.hP[1].get4(hf, n, q);
    float av[4] = {a4.x, a4.y, a4.z, a4.w};
    for (int c = 0; c < 4; ++c) { int kk = 2 * (4 * q + c) + hf; if (av[c] != (float)(kk * 100 + n)) b++; }
  }
  if (b) atomicAdd(bad, b);
}
int main() { int *d, h = 0; hipMalloc(&d, 4); hipMemcpy(d, &h, 4, hipMemcpyHostToDevice); hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("bad = %d\n", h); return 0; }
