// device vs host bit-comparison of the deterministic gate functions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/rl_detmath.h"
__global__ void k(const float *x, float *e, float *s, float *t, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { e[i] = rl_exp_nonpos(-fabsf(x[i])); s[i] = rl_sigmoidf(x[i]); t[i] = rl_tanhf(x[i]); }
}
int main() {
  const int n = 1 << 20;
  float *hx = (float *)malloc(n * 4), *he = (float *)malloc(n * 4), *hs = (float *)malloc(n * 4), *ht = (float *)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ((float)rand() / RAND_MAX * 2 - 1) * (i % 3 == 0 ? 0.3f : (i % 3 == 1 ? 4.0f : 30.0f));
  float *dx, *de, *ds, *dt;
  hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dt, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, de, ds, dt, n);
  hipMemcpy(he, de, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ht, dt, n * 4, hipMemcpyDeviceToHost);
  int be = 0, bs = 0, bt = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    float e = rl_exp_nonpos(-fabsf(hx[i])), s = rl_sigmoidf(hx[i]), t = rl_tanhf(hx[i]);
    if (memcmp(&e, &he[i], 4)) { be++; if (shown++ < 5) printf("exp x=%a host %a dev %a\n", hx[i], e, he[i]); }
    if (memcmp(&s, &hs[i], 4)) bs++;
    if (memcmp(&t, &ht[i], 4)) bt++;
  }
  printf("mismatches of %d: exp %d sigmoid %d tanh %d\n", n, be, bs, bt);
  return 0;
}
