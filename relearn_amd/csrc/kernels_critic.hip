// kernels_critic.hip — the fused critic step: forward + MSE loss + backward of the 5-128-1 value MLP over all samples.
//
// Reference semantics: ValuesOpt::update (src/torch/agents/critics/opt.rs:100-126): loss = mse_loss(V(obs), targets,
// Mean); backward; the Adam step itself is k_reduce_adam / k_adam_step (kernels_update.hip).
//
// What the hardware dictates (measured, scripts/probe/pipe_overlap.hip): v_mfma_f32_32x32x2_f32 runs at the f32 vector
// rate AND occupies the vector ALU — its time adds to every other wave's VALU time on the SIMD — while the bf16 matrix
// pipe (v_mfma_f32_32x32x16_bf16, 36 cycles) runs beside VOP2 vector work at the price of 8 issue cycles.  So both
// GEMM-shaped parts of the step run on the bf16 pipe, with every product exact:
//   an f32 value splits EXACTLY into three bf16 pieces, v = p0 + p1 + p2 (8 + 8 + 8 significand bits; each residual is
//   representable), and a product of two bf16 numbers is exact in the f32 accumulator.  The only roundings left are the
//   f32 accumulations inside the instruction — measured below the error of a sequential f32 fma chain over the same
//   terms (scripts/probe/mfma_bf16_mask.hip).  Nothing is computed at reduced precision.
//
// One wavefront owns a tile of 32 samples at a time.
//   forward   pre[s][j] = sum_k x~[s][k] W~1[j][k]  (k = 5: bias, x~ = 1) as sum over the 9 piece pairs of every k:
//             48 contraction slots = 3 issues per 32-unit hidden tile, oriented with the HIDDEN UNIT on the lane
//             (col = lane & 31) and the SAMPLE in the accumulator registers (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
//             relu, the 128 -> 1 layer as per-lane partial sums, one per-wave LDS transpose, per-sample loss and dL/dy
//             on the sample-owning lanes.
//   backward  M[j][k] = sum_s [pre_sj > 0] . u_sk  with  u_sk = dL/dy_s . x~_sk — a masked sum: the mask is 0 or 1, u
//             splits into three pieces, M = G^T [p0 | p1 | p2].  The forward accumulator tile is already laid out as this
//             instruction's A operand (sum over its row index: "X^T . B", cdna_hip_programming.md §3), so the mask goes
//             from registers to the matrix pipe without lane movement: per (sample, hidden unit) the VALU spends one
//             multiply-clamp (relu') and half a convert instead of the seven operations of a 6-column fma backward.
//             The 18 piece columns occupy 18 of the 32 output columns; the three pieces of a column are added when the
//             f32 accumulators are flushed into the f64 level of the two-level accumulation (kernels_update.hip says why
//             TRPO / Adam want that level).
//   at the end  dL/dW1[j][k] = w2_j M[j][k],  dL/db1[j] = w2_j M[j][5],  dL/db2 = sum dy,
//               dL/dW2[j] = sum_s dy_s h_sj = sum_k W~1[j][k] M[j][k]   (h_sj = [pre_sj > 0] W~1[j] . x~_s).
// Algorithmic flops per sample: 3 x (2*5*128 + 2*128) = 4608 (forward + 2 x backward of the 5-128-1 MLP).
#include "device_fns.hpp"
#include "kernels.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wavefront execute in program order; what is needed is that the compiler keeps that
  // order across lanes it cannot see a dependence between.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two f32 -> packed bf16 (element 0 in the low half), round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 x = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __builtin_bit_cast(float, bits16 << 16); }
// v = p0 + p1 + p2 exactly, each piece a bf16 bit pattern in the low half of a register
__device__ __forceinline__ void split3(float v, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
  p0 = pack_bf16(v, 0.0f);
  const float r1 = v - bf16_bits_to_f32(p0);
  p1 = pack_bf16(r1, 0.0f);
  const float r2 = r1 - bf16_bits_to_f32(p1);
  p2 = pack_bf16(r2, 0.0f);
}
__device__ __forceinline__ uint32_t pk(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }

union Frag {
  bf16x8 v;
  uint32_t u[4];
  uint64_t q[2];
};

}  // namespace

#ifndef RL_CRITIC_WAVES
#define RL_CRITIC_WAVES 8
#endif
constexpr int CRITIC_WAVES = RL_CRITIC_WAVES;  // waves per workgroup, one workgroup per CU
constexpr int C_NT = 4;           // 32-unit hidden tiles (H = 128)
constexpr int C_FLUSH = 16;       // f32 -> f64 flush period in tiles (512 samples per accumulator)
constexpr int C_COLS = 18;        // piece columns of the backward: 3 k + p, k = input feature (5 = bias), p = piece

#ifndef RL_CRITIC_OCC
#define RL_CRITIC_OCC 2  // waves per SIMD the register budget is set for
#endif
// Contraction slots of the forward (48 = 3 issues x 16; lane half h of issue i holds slots 16 i + 8 h + 0..7, i.e. the
// half's own list u = 8 i + j, 24 entries).  An entry pairs piece a of an input with piece b of the matching weight:
//   u = 0..8    input 2h,     (a, b) = (0,0) (0,1) (0,2) (1,0) (1,1) (1,2) (2,0) (2,1) (2,2)
//   u = 9..17   input 2h + 1, the same nine pairs
//   u = 18..23  h = 0: input 4, (0,0) (0,1) (0,2) (1,0) (1,1) (1,2)
//               h = 1: input 4, (2,0) (2,1) (2,2); then the bias input 1.0 (one piece) with the bias's three pieces
// so every lane needs three observation features of its sample (2h, 2h + 1, 4), like the f32 form did.
__global__ void __launch_bounds__(CRITIC_WAVES * 64, RL_CRITIC_OCC)
    k_critic_step_mfma(TrajDev tr, const float *__restrict__ params, double *__restrict__ slabA,
                       double *__restrict__ slabB, float two_over_B, uint32_t P) {
  constexpr int D = 5, H = 128, NT = C_NT;
  constexpr int IMG = H * 7 + 2;  // per hidden unit: M[0..5] (slot 6 unused); then db2, loss
  __shared__ float Ysh[CRITIC_WAVES][32][33];
  __shared__ __attribute__((aligned(8))) unsigned short Ubf[CRITIC_WAVES][C_COLS][36];  // [piece column][sample], 72-B rows
  __shared__ double Acc[CRITIC_WAVES][IMG];  // f64 level of the two-level accumulation, one image per wave

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H;
  const float b2 = W2[H];
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  double *acc64 = Acc[wave];
  for (int p = lane; p < IMG; p += 64) acc64[p] = 0.0;

  // B operands of the forward: the weight pieces of hidden unit 32 t + n in this half's slot order
  Frag fw[NT][3];
  float w2v[NT];
  // relu(x) = (x + |x|) / 2, so  y = b2 + (sum_j w2_j pre_j + sum_j w2_j |pre_j|) / 2:  the first sum is linear in the
  // inputs, v . x~ with v_k = sum_j w2_j W~1[j][k] (this lane keeps the v of its half's inputs: 2 hf, 2 hf + 1, and
  // 4 or the bias), the second costs one fma with an |.| source per (sample, unit) — no separate relu
  float lv[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = t * 32 + n;
    lv[0] = __builtin_fmaf(W2[j], W1[j * D + 2 * hf], lv[0]);
    lv[1] = __builtin_fmaf(W2[j], W1[j * D + 2 * hf + 1], lv[1]);
    lv[2] = __builtin_fmaf(W2[j], hf == 0 ? W1[j * D + 4] : b1[j], lv[2]);
    uint32_t a0, a1, a2, c0, c1, c2, e0, e1, e2, g0, g1, g2;
    split3(W1[j * D + 2 * hf], a0, a1, a2);
    split3(W1[j * D + 2 * hf + 1], c0, c1, c2);
    split3(W1[j * D + 4], e0, e1, e2);
    split3(b1[j], g0, g1, g2);
    fw[t][0].u[0] = pk(a0, a1);
    fw[t][0].u[1] = pk(a2, a0);
    fw[t][0].u[2] = pk(a1, a2);
    fw[t][0].u[3] = pk(a0, a1);
    fw[t][1].u[0] = pk(a2, c0);
    fw[t][1].u[1] = pk(c1, c2);
    fw[t][1].u[2] = pk(c0, c1);
    fw[t][1].u[3] = pk(c2, c0);
    fw[t][2].u[0] = pk(c1, c2);
    fw[t][2].u[1] = pk(e0, e1);
    fw[t][2].u[2] = hf == 0 ? pk(e2, e0) : pk(e2, g0);
    fw[t][2].u[3] = hf == 0 ? pk(e1, e2) : pk(g1, g2);
    w2v[t] = W2[j];
  }
  // backward accumulators (matrix pipe): dm[t][r] = sum over samples for hidden unit 32 t + row(r, hf), piece column n
  f32x16 dm[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dm[t] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) lv[q] = lv[q] + __shfl_xor(lv[q], m, 64);  // over the 32 lanes of the half
  const float big = 0x1p126f;
  double loss64 = 0.0, db2_64 = 0.0;
  wave_lds_fence();

  // f32 -> f64 flush: the three pieces of an input column sit in neighbouring lanes; add them (p0 + p1) + p2, then
  // accumulate in the wave's f64 image
  auto flush = [&]() {
    const bool owner = n < C_COLS && (n % 3) == 0;
    const int k = n / 3;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = dm[t][r];
        const float v1 = __shfl_down(v, 1, 64), v2 = __shfl_down(v, 2, 64);
        const float tot = (v + v1) + v2;
        const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
        if (owner) acc64[j * 7 + k] += (double)tot;
        dm[t][r] = 0.0f;
      }
  };

  const size_t n_tiles = (B + 31) / 32;
  const size_t wave_id = (size_t)blockIdx.x * CRITIC_WAVES + wave, n_waves = (size_t)gridDim.x * CRITIC_WAVES;
  int since_flush = 0;
  // per lane: features 2 hf, 2 hf + 1 and 4 of sample n, and its target
  struct TileOp {
    float xa, xb, xc, tgt;
    bool valid;
  };
  auto load_tile = [&](size_t g) {
    TileOp o;
    const size_t sidx = g * 32 + n;
    o.xa = o.xb = o.xc = o.tgt = 0.0f;
    o.valid = g < n_tiles && sidx < B;
    if (o.valid) {
      o.xa = tr.obs[(size_t)(2 * hf) * plane + sidx];
      o.xb = tr.obs[(size_t)(2 * hf + 1) * plane + sidx];
      o.xc = tr.obs[(size_t)4 * plane + sidx];
      o.tgt = tr.tgt[sidx];
    }
    return o;
  };

  TileOp op = load_tile(wave_id);
  for (size_t g = wave_id; g < n_tiles; g += n_waves) {
    const TileOp next = load_tile(g + n_waves);  // global loads run one tile ahead
    // ---- A operands of the forward: the input pieces of sample n in this half's slot order
    Frag fa[3];
    {
      uint32_t a0, a1, a2, c0, c1, c2, e0, e1, e2;
      split3(op.xa, a0, a1, a2);
      split3(op.xb, c0, c1, c2);
      split3(op.xc, e0, e1, e2);
      const uint32_t one = op.valid ? 0x3F80u : 0u;  // the bias input
      const uint32_t x0 = hf == 0 ? e0 : e2, x1 = hf == 0 ? e1 : one;
      fa[0].u[0] = pk(a0, a0);
      fa[0].u[1] = pk(a0, a1);
      fa[0].u[2] = pk(a1, a1);
      fa[0].u[3] = pk(a2, a2);
      fa[1].u[0] = pk(a2, c0);
      fa[1].u[1] = pk(c0, c0);
      fa[1].u[2] = pk(c1, c1);
      fa[1].u[3] = pk(c1, c2);
      fa[2].u[0] = pk(c2, c2);
      fa[2].u[1] = pk(x0, x0);
      fa[2].u[2] = pk(x0, x1);
      fa[2].u[3] = pk(x1, x1);
    }
    auto layer1 = [&](int t) {
      f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 3; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fw[t][i].v, c, 0, 0, 0);
      return c;
    };
    // ---- forward, one hidden tile at a time, software-pipelined: the matrix pipe works on hidden tile t + 1 while the
    // VALU does relu, the partial y and the relu' mask of hidden tile t; the mask is packed as the backward's A operand
    Frag ga[NT][2];
    float yp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yp[r] = 0.0f;
    f32x16 c = layer1(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 cn = c;
      if (t + 1 < NT) cn = layer1(t + 1);
      float gm[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pre = c[r];
        yp[r] = __builtin_fmaf(__builtin_fabsf(pre), w2v[t], yp[r]);
        // relu'(pre) as one VALU op: clamp(pre * 2^126) is 1 for every normal pre > 0 and 0 for pre <= 0
        gm[r] = __builtin_amdgcn_fmed3f(pre * big, 0.0f, 1.0f);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[t][s].u[i] = pack_bf16(gm[8 * s + 2 * i], gm[8 * s + 2 * i + 1]);
      c = cn;
    }
    // ---- y: transpose the 16 partial sums per lane through LDS (row = sample, column = source lane)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ysh[wave][(r & 3) + 8 * (r >> 2) + 4 * hf][n] = yp[r];
    wave_lds_fence();
    float part = 0.0f;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) part = part + Ysh[wave][n][hf * 16 + cc];
    {  // the linear half of relu: this half's inputs of sample n
      float lin = lv[0] * op.xa;
      lin = __builtin_fmaf(lv[1], op.xb, lin);
      lin = __builtin_fmaf(lv[2], hf == 0 ? op.xc : 1.0f, lin);
      part = part + lin;
    }
    const float other = __shfl_xor(part, 32, 64);
    const float p0 = hf == 0 ? part : other, p1 = hf == 0 ? other : part;
    const float y = 0.5f * (p0 + p1) + b2;
    const float d = y - op.tgt;
    const float dy = op.valid ? d * two_over_B : 0.0f;
    if (hf == 0 && op.valid) {
      loss64 += (double)(d * d);
      db2_64 += (double)dy;
    }
    // ---- u[sample n][k] = dy * x~_k for this lane's three k (half 0: k = 0, 1, 4; half 1: k = 2, 3 and 5, where
    // x~_5 = 1), split into exact bf16 pieces
    {
      const float uv[3] = {dy * op.xa, dy * op.xb, hf == 0 ? dy * op.xc : dy};
      const int kk[3] = {2 * hf, 2 * hf + 1, 4 + hf};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        uint32_t q0, q1, q2;
        split3(uv[q], q0, q1, q2);
        Ubf[wave][3 * kk[q] + 0][n] = (unsigned short)q0;
        Ubf[wave][3 * kk[q] + 1][n] = (unsigned short)q1;
        Ubf[wave][3 * kk[q] + 2][n] = (unsigned short)q2;
      }
    }
    wave_lds_fence();
    // ---- backward on the bf16 matrix pipe: B element j of lane half hf = piece[sample 16 s + 8 (j >> 2) + 4 hf + (j & 3)]
    Frag ub[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ub[s].q[0] = 0;
      ub[s].q[1] = 0;
      if (n < C_COLS) {
        ub[s].q[0] = *reinterpret_cast<const uint64_t *>(&Ubf[wave][n][16 * s + 4 * hf]);
        ub[s].q[1] = *reinterpret_cast<const uint64_t *>(&Ubf[wave][n][16 * s + 8 + 4 * hf]);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) dm[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[t][s].v, ub[s].v, dm[t], 0, 0, 0);
    wave_lds_fence();  // Ysh / Ubf are rewritten by the next tile
    if (++since_flush == C_FLUSH) {
      since_flush = 0;
      flush();
    }
    op = next;
  }
  flush();
  // loss / db2: reduce over the 32 owner lanes of the wave (f64 moved as two 32-bit halves)
  auto xlane = [](double v, int mask) {
    uint64_t bits = rl_f64_bits(v);
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, mask, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), mask, 64);
    return rl_f64_from_bits(((uint64_t)hi << 32) | lo);
  };
  double l = hf == 0 ? loss64 : 0.0, bsum = hf == 0 ? db2_64 : 0.0;
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) {
    l = l + xlane(l, s);
    bsum = bsum + xlane(bsum, s);
  }
  if (lane == 0) {
    acc64[H * 7] = bsum;   // db2
    acc64[H * 7 + 1] = l;  // loss partial
  }
  __syncthreads();
  // sum the per-wave images in wave order, turn M into gradients and write the workgroup's slab row
  for (uint32_t p = threadIdx.x; p <= P; p += CRITIC_WAVES * 64) {
    auto tot = [&](int src) {
      double s = Acc[0][src];
#pragma unroll
      for (int w = 1; w < CRITIC_WAVES; ++w) s = s + Acc[w][src];
      return s;
    };
    double s;
    if (p < (uint32_t)(H * D)) {
      int j = p / D, k = p % D;
      s = tot(j * 7 + k) * (double)W2[j];
    } else if (p < (uint32_t)(H * D + H)) {
      int j = p - H * D;
      s = tot(j * 7 + 5) * (double)W2[j];
    } else if (p < (uint32_t)(H * D + 2 * H)) {
      int j = p - H * D - H;
      s = tot(j * 7 + 5) * (double)b1[j];
#pragma unroll
      for (int k = 0; k < D; ++k) s += tot(j * 7 + k) * (double)W1[j * D + k];
    } else if (p == (uint32_t)(H * D + 2 * H)) {
      s = tot(H * 7);
    } else {
      s = tot(H * 7 + 1);
    }
    if (p < P) slabA[(size_t)blockIdx.x * P + p] = s;
    else slabB[(size_t)blockIdx.x * 4 + 0] = s;
  }
  if (threadIdx.x < 3) slabB[(size_t)blockIdx.x * 4 + 1 + threadIdx.x] = 0.0;
}

// ---------------------------------------------------------------- launcher
bool launch_critic_step_v2(rl_traj *traj, const rl_mlp *critic, uint64_t B_total) {
  if (traj->d.D != 5 || critic->hidden != 128 || critic->out_dim != 1) return false;
  ProfScope ps(traj->eng, RL_K_CRITIC_FUSED);
  float two_over_B = 2.0f / (float)B_total;
  // persistent grid: one fat workgroup per CU (fewer, fatter workgroups = fewer slab rows for the reduction that follows
  // every launch), one 32-sample tile per wave and iteration
  const uint64_t n_tiles = (traj->B + 31) / 32, cus = (uint64_t)traj->eng->prop.multiProcessorCount;
  uint64_t nb = (n_tiles + CRITIC_WAVES - 1) / CRITIC_WAVES;
  if (nb > cus) nb = cus;
  traj->nbC = (uint32_t)nb;
  hipLaunchKernelGGL(k_critic_step_mfma, dim3(traj->nbC), dim3(CRITIC_WAVES * 64), 0, traj->eng->stream, traj->d,
                     critic->d_params, traj->slabA, traj->slabB, two_over_B, (uint32_t)critic->P);
  return true;
}
