// kernels_critic.hip — the fused critic step: forward + MSE loss + backward of the 5-128-1 value MLP over all samples.
//
// Reference semantics: ValuesOpt::update (src/torch/agents/critics/opt.rs:100-126): loss = mse_loss(V(obs), targets,
// Mean); backward; the Adam step itself is k_reduce_adam / k_adam_step (kernels_update.hip).
//
// Tile machinery (bf16_tile.hpp): layer 1, the output layer, the masked-sum backward AND the two lane transposes a tile
// needs all run on the bf16 matrix pipe with exact three-piece splits — no reduced precision, no LDS in the tile loop.
// Per 32-sample tile (38 matrix instructions, ~190 vector instructions):
//   F  forward   pre = x~ W~1^T scaled by 2^96 (12 issues); relu' as packed 0/1 masks, one v_cvt_pk_bf16_f32 with the
//                clamp bit per two values — the backward's A operand as it stands;
//   T  transpose mask^T = mask^T . I (8 issues): the masks with the sample on the lane;
//   L  output    y_s = b2 + sum_k x~_sk q_sk,  q_sk = sum_j relu'(pre_sj) (w2_j W~1[j][k])  — a masked sum over the hidden
//                units (8 issues), then three fmas per lane and one half exchange; loss and dL/dy on the sample lanes;
//   P  pieces    u_sk = dy_s x~_sk as exact bf16 pieces, routed from the sample lanes to the piece columns by a
//                selection product (2 issues);
//   B  backward  M[j][k] = sum_s relu'(pre_sj) u_sk (8 issues).
//   at the end  dL/dW1[j][k] = w2_j M[j][k],  dL/db1[j] = w2_j M[j][5],  dL/db2 = sum dy,
//               dL/dW2[j] = sum_s dy_s h_sj = sum_k W~1[j][k] M[j][k]   (h_sj = [pre_sj > 0] W~1[j] . x~_s).
// The VALU's share per (sample, hidden unit) is 3/4 of an instruction (round 2: 2.5; round 1: 9).
// Algorithmic flops per sample: 3 x (2*5*128 + 2*128) = 4608 (forward + 2 x backward of the 5-128-1 MLP).
#include <type_traits>

#include "bf16_tile.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"

using bt::f32x16;
using bt::Frag;

// ONE wave per SIMD (four per workgroup, one workgroup per CU), 512 registers per lane.  The waves of a SIMD share one
// issue port (vector instruction ~4 cycles, matrix instruction 8, LDS instruction 14-25; profiles/r03_slot_cost.txt), so
// a second wave adds no throughput, only registers taken away: a single in-order stream per SIMD, software-pipelined
// across sample tiles and written slot by slot — one matrix instruction plus the ~5 vector instructions that issue in
// its 32-cycle shadow — keeps the matrix pipe fed if every consumer sits two slots or more behind its producer.
// (Round 2 and the first builds of this round ran two waves per SIMD at 256 registers with the transposes through LDS:
// 195-215 us per launch of 8.4 M samples, matrix pipe 39-53 % busy, the issue port saturated by LDS instructions.)
#ifndef ABL_WAVES
#define ABL_WAVES 4
#endif
constexpr int CRITIC_WAVES = ABL_WAVES;
constexpr int C_FLUSH = 64;  // f32 -> f64 flush period in tiles (2048 samples per accumulator: the accumulated error stays
                             // below a 128-sample f32 fma chain's, scripts/probe/mfma_bf16_mask.hip)

__global__ void __launch_bounds__(CRITIC_WAVES * 64, 1)
    k_critic_step_mfma(TrajDev tr, const float *__restrict__ params, double *__restrict__ slabA,
                       double *__restrict__ slabB, float two_over_B, uint32_t P) {
  constexpr int D = 5, H = 128, NT = bt::NT;
  constexpr int IMG = H * 7 + 2;  // per hidden unit: M[0..5] (slot 6 unused); then db2, loss
  __shared__ uint4 L2f[bt::L2_KS][64];       // output-layer A operands (staging: each wave copies them into registers)
  __shared__ double Acc[CRITIC_WAVES][IMG];  // f64 level of the two-level accumulation, one image per wave

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H;
  const float b2 = W2[H];
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  double *acc64 = Acc[wave];
  for (int p = lane; p < IMG; p += 64) acc64[p] = 0.0;

  // forward weight pieces of hidden unit 32 t + n in this half's slot order, scaled by 2^96 (bf16_tile.hpp: relu')
  Frag fw[NT][3];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = t * 32 + n;
    const float sc = bt::FWD_SCALE;
    bt::weight_frags(sc * W1[j * D + 2 * hf], sc * W1[j * D + 2 * hf + 1], sc * W1[j * D + 4], sc * b1[j], hf, fw[t]);
  }
  // output layer: A operands of q^T = Wp^T mask^T with Wp[j][k] = w2_j W~1[j][k]
  Frag fq[bt::L2_KS];
  bt::l2_build(L2f, (int)threadIdx.x, CRITIC_WAVES * 64,
               [&](int j, int k) { return W2[j] * (k < D ? W1[j * D + k] : b1[j]); });
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < bt::L2_KS; ++ks) fq[ks] = bt::l2_frag(L2f, ks, lane);
  Frag idb[2], selb[2];  // identity / piece-column selections (B operands of the two transposes)
  bt::ident_frags(lane, idb);
  bt::sel_frags(lane, selb);
  // backward accumulators (matrix pipe): dm[t][r] = sum over samples for hidden unit 32 t + row(r, hf), piece column n
  f32x16 dm[NT];
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < NT; ++t) dm[t] = zero16;
  double loss64 = 0.0, db2_64 = 0.0;
  bt::wave_lds_fence();

  const size_t n_tiles = (B + 31) / 32;
  const size_t wave_id = (size_t)blockIdx.x * CRITIC_WAVES + wave, n_waves = (size_t)gridDim.x * CRITIC_WAVES;
  // per lane: features 2 hf, 2 hf + 1 and 4 of sample n, and its target
  struct TileOp {
    float xa, xb, xc, tgt;
    bool valid;
  };
  // branch-free (padding lanes read sample B - 1 and are zeroed): the loads of tile i + 1 stay in flight across tile i
  // (32-bit element offsets from the uniform base pointers: a launch covers < 2^30 samples, checked by the launcher)
  const uint32_t B32 = (uint32_t)B, plane32 = (uint32_t)plane;
  auto load_tile = [&](size_t g) {
    TileOp o;
    const uint32_t sidx = (uint32_t)g * 32u + (uint32_t)n;
    o.valid = g < n_tiles && sidx < B32;
    const uint32_t sc = o.valid ? sidx : B32 - 1;
    // raw values: a padding lane's are zeroed where they are first USED (zeroed here, the select would be the loads'
    // first use and wait for them one iteration early)
    o.xa = tr.obs[(uint32_t)(2 * hf) * plane32 + sc];
    o.xb = tr.obs[(uint32_t)(2 * hf + 1) * plane32 + sc];
    o.xc = tr.obs[4u * plane32 + sc];
    o.tgt = tr.tgt[sc];
    return o;
  };
  auto zero_invalid = [](TileOp &o) {
    o.xa = o.valid ? o.xa : 0.0f;
    o.xb = o.valid ? o.xb : 0.0f;
    o.xc = o.valid ? o.xc : 0.0f;
    o.tgt = o.valid ? o.tgt : 0.0f;
  };
  auto mfma = [](const Frag &a, const Frag &b, const f32x16 &c) {
#ifdef ABL_NOMFMA
    f32x16 r = c;
    r[0] += __builtin_bit_cast(float, a.u[0] ^ b.u[0]);
    return r;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
#endif
  };

  // Pipeline state.  Iteration i runs, interleaved in one instruction stream:
  //   tile i      F (12 issues), relu' masks into ga[i & 1], T (8), L (8) -> q, then y, loss terms, dy and the split of
  //               dy x~ into pieces (in the shadow of B)
  //   tile i - 1  loss accumulation, pieces -> P (2), B (8, masks ga[(i - 1) & 1])
  //   tile i + 1  input pieces fa;   tile i + 2  global loads
  // so the loop runs one iteration past the wave's last tile (an invalid tile contributes zeros everywhere).
  // Each SLOT below is one matrix instruction and the ~5 vector instructions that issue in its 32-cycle shadow, pinned
  // by scheduling barriers; a matrix result is read by the vector unit two slots or more after its instruction, and
  // the three-way splits run as three interleaved chains (a lone wave issues a DEPENDENT vector instruction only every
  // ~6.6 cycles).
  Frag ga[2][NT][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int t = 0; t < NT; ++t) ga[b][t][0].q[0] = ga[b][t][0].q[1] = ga[b][t][1].q[0] = ga[b][t][1].q[1] = 0;
  TileOp cur = load_tile(wave_id), nx1 = load_tile(wave_id + n_waves);
  zero_invalid(cur);
  Frag fa[3];
  bt::input_frags(cur.xa, cur.xb, cur.xc, cur.valid, hf, fa);
  bt::Split3x3 us, xs;  // pieces of dy x~ (tile i - 1 at the top of an iteration), of the next tile's inputs
#pragma unroll
  for (int i = 0; i < 3; ++i) us.v[i] = us.r[i] = 0.0f, us.p0[i] = us.p1[i] = us.p2[i] = 0u;
  float d_prev = 0.0f, dy_prev = 0.0f;
  bool own_prev = false;

#define SLOT() __builtin_amdgcn_sched_barrier(0)
  auto iteration = [&](auto parity, size_t g) {
    constexpr int PAR = decltype(parity)::value;
    f32x16 c[NT], gt[NT], ut, q;
    Frag mt[bt::L2_KS], ub[2], pa[2];
    float first[NT];
    auto F = [&](int t, int i) { c[t] = mfma(fa[i], fw[t][i], i == 0 ? zero16 : c[t]); };
    auto T = [&](int t, int s) { gt[t] = mfma(ga[PAR][t][s], idb[s], s == 0 ? zero16 : gt[t]); };
    auto TCVT = [&](int t, int s) {  // half s of the transposed mask tile t -> B operand of k-step 2 t + s
#pragma unroll
      for (int i = 0; i < 4; ++i) mt[2 * t + s].u[i] = bt::pack_bf16(gt[t][8 * s + 2 * i], gt[t][8 * s + 2 * i + 1]);
    };
    auto MASKA = [&](int t) {
      first[t] = bt::mask_first(c[t]);
      bt::mask_half(c[t], first[t], 0, ga[PAR][t][0]);
    };
    auto MASKB = [&](int t) { bt::mask_half(c[t], first[t], 1, ga[PAR][t][1]); };
    auto L = [&](int ks) { q = mfma(fq[ks], mt[ks], ks == 0 ? zero16 : q); };
    auto BW = [&](int t, int s) { dm[t] = mfma(ga[PAR ^ 1][t][s], ub[s], dm[t]); };
    TileOp nx2;
    const uint32_t sidx2 = (uint32_t)(g + 2 * n_waves) * 32u + (uint32_t)n;
    SLOT();
    F(0, 0);
    us.st3();
    SLOT();
    F(0, 1);
    us.st4();
    loss64 += (double)(own_prev ? d_prev * d_prev : 0.0f);
    SLOT();
    F(0, 2);
    bt::piece_operand(us.get(0), us.get(1), us.get(2), pa);
    SLOT();
    F(1, 0);
    db2_64 += (double)(own_prev ? dy_prev : 0.0f);
    nx2.valid = g + 2 * n_waves < n_tiles && sidx2 < B32;
    const uint32_t sc2 = nx2.valid ? sidx2 : B32 - 1;
    SLOT();
    F(1, 1);
    MASKA(0);
    SLOT();
    F(1, 2);
    MASKB(0);
#ifdef ABL_NOLOAD
    nx2.xa = __builtin_bit_cast(float, sc2 & 0x3f800000u);
#else
    nx2.xa = tr.obs[(uint32_t)(2 * hf) * plane32 + sc2];
#endif
    SLOT();
    F(2, 0);
#ifdef ABL_NOLOAD
    nx2.xb = __builtin_bit_cast(float, sc2 & 0x3f000000u);
    nx2.xc = __builtin_bit_cast(float, sc2 & 0x3e800000u);
#else
    nx2.xb = tr.obs[(uint32_t)(2 * hf + 1) * plane32 + sc2];
    nx2.xc = tr.obs[4u * plane32 + sc2];
#endif
    SLOT();
    F(2, 1);
    MASKA(1);
    SLOT();
    F(2, 2);
    MASKB(1);
#ifdef ABL_NOLOAD
    nx2.tgt = __builtin_bit_cast(float, sc2 & 0x3f400000u);
#else
    nx2.tgt = tr.tgt[sc2];
#endif
    SLOT();
    T(0, 0);
    zero_invalid(nx1);
    xs.v[0] = nx1.xa, xs.v[1] = nx1.xb, xs.v[2] = nx1.xc;
    xs.st0();
    SLOT();
    T(0, 1);
    MASKA(2);
    SLOT();
    F(3, 0);
    MASKB(2);
    SLOT();
    F(3, 1);
    TCVT(0, 0);
    SLOT();
    F(3, 2);
    TCVT(0, 1);
    SLOT();
    T(1, 0);
    xs.st1();
    SLOT();
    T(1, 1);
    MASKA(3);
    SLOT();
    L(0);
    MASKB(3);
    SLOT();
    L(1);
    TCVT(1, 0);
    xs.st2();
    SLOT();
    T(2, 0);
    TCVT(1, 1);
    SLOT();
    T(2, 1);
    xs.st3();
    SLOT();
    L(2);
    xs.st4();
    bt::input_frags_pack(xs.get(0), xs.get(1), xs.get(2), nx1.valid, hf, 0, fa[0]);
    SLOT();
    L(3);
    TCVT(2, 0);
    SLOT();
    T(3, 0);
    TCVT(2, 1);
    SLOT();
    T(3, 1);
    bt::input_frags_pack(xs.get(0), xs.get(1), xs.get(2), nx1.valid, hf, 1, fa[1]);
    SLOT();
    ut = mfma(pa[0], selb[0], zero16);
    bt::input_frags_pack(xs.get(0), xs.get(1), xs.get(2), nx1.valid, hf, 2, fa[2]);
    SLOT();
    ut = mfma(pa[1], selb[1], ut);
    TCVT(3, 0);
    SLOT();
    L(4);
    TCVT(3, 1);
    SLOT();
    L(5);
#pragma unroll
    for (int i = 0; i < 4; ++i) ub[0].u[i] = bt::pack_bf16(ut[2 * i], ut[2 * i + 1]);
    SLOT();
    L(6);
#pragma unroll
    for (int i = 0; i < 4; ++i) ub[1].u[i] = bt::pack_bf16(ut[8 + 2 * i], ut[8 + 2 * i + 1]);
    SLOT();
    L(7);
    SLOT();
    BW(0, 0);
    SLOT();
    BW(0, 1);
    SLOT();
    BW(1, 0);
    // ---- tile i: y of sample n (this half's three inputs, then the other half's), loss terms, dy, pieces of dy x~
    const float x3 = hf == 0 ? cur.xc : 1.0f;
    const float sa_ = (q[0] + q[1]) + q[2], sb_ = (q[3] + q[4]) + q[5];
    SLOT();
    BW(1, 1);
    const float sc_ = (q[6] + q[7]) + q[8];
    const float part = __builtin_fmaf(x3, sc_, __builtin_fmaf(cur.xb, sb_, cur.xa * sa_));
    float p0, p1;
    bt::both_halves(part, p0, p1);
    SLOT();
    BW(2, 0);
    const float y = (p0 + p1) + b2;
    const float d = y - cur.tgt;
    const float dy = cur.valid ? d * two_over_B : 0.0f;
    SLOT();
    BW(2, 1);
    us.v[0] = dy * cur.xa, us.v[1] = dy * cur.xb, us.v[2] = hf == 0 ? dy * cur.xc : dy;
    us.st0();
    SLOT();
    BW(3, 0);
    us.st1();
    SLOT();
    BW(3, 1);
    us.st2();
    SLOT();
    d_prev = d;
    dy_prev = dy;
    own_prev = hf == 0 && cur.valid;
    cur = nx1;
    nx1 = nx2;
  };
#undef SLOT

  // tiles of this wave: wave_id + k n_waves, k = 0 .. my - 1; my + 1 iterations, rounded up to a pair
  const size_t my = wave_id < n_tiles ? (n_tiles - wave_id + n_waves - 1) / n_waves : 0;
  size_t g = wave_id;
#ifdef RL_LOOP_CLOCK
  const uint64_t clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // (the f32 -> f64 flush sits between runs of C_FLUSH tiles, outside the inner loop: its body stays one basic block,
  // so nothing the next trip needs can be sunk behind the matrix instructions it was placed beside)
  for (size_t it = 0; it < my + 1;) {
    const size_t lim = it + C_FLUSH < my + 1 ? it + C_FLUSH : my + 1;
    for (; it < lim; it += 2, g += 2 * n_waves) {
      iteration(std::integral_constant<int, 0>{}, g);
      iteration(std::integral_constant<int, 1>{}, g + n_waves);
    }
    bt::flush(dm, acc64, 7, n, hf);
  }
#ifdef RL_LOOP_CLOCK
  {
    const uint64_t clk1 = __builtin_readcyclecounter(), rt1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
      printf("block %d wave 0: %llu tiles, %llu shader cycles, %llu ns (100 MHz counter) -> %.1f cycles per tile, %.3f GHz\n",
             (int)blockIdx.x, (unsigned long long)my, (unsigned long long)(clk1 - clk0), (unsigned long long)(rt1 - rt0) * 10,
             (double)(clk1 - clk0) / (double)(my + 1), (double)(clk1 - clk0) / ((double)(rt1 - rt0) * 10.0));
  }
#endif
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
  if ((uint64_t)(traj->d.T + 1) * traj->d.n * 5 >= (1ull << 30)) return false;  // 32-bit element offsets in the kernel
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
