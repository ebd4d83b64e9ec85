// kernels_critic.hip — the fused critic step: forward + MSE loss + backward of the 5-128-1 value MLP over all samples.
//
// Reference semantics: ValuesOpt::update (src/torch/agents/critics/opt.rs:100-126): loss = mse_loss(V(obs), targets,
// Mean); backward; the Adam step itself is k_reduce_adam / k_adam_step (kernels_update.hip).
//
// Tile machinery (layer 1 and the masked-sum backward on the bf16 matrix pipe with exact three-piece splits — no
// reduced precision): bf16_tile.hpp.  On top of it, per 32-sample tile and wave:
//   relu through |x|:  relu(x) = (x + |x|) / 2, so  y = b2 + (sum_j w2_j pre_j + sum_j w2_j |pre_j|) / 2 — the first sum
//   is linear in the inputs (v . x~ with v_k = sum_j w2_j W~1[j][k], six numbers per launch), the second costs one fma
//   with an |.| source per (sample, unit): no separate relu; the partial sums of a lane's four units go through one
//   per-wave LDS transpose; per-sample loss and dL/dy on the sample-owning lanes; the backward's mask is HALF an
//   instruction per (sample, unit): the forward is scaled by 2^96 and v_cvt_pk_bf16_f32 with the clamp bit packs two
//   relu' values at once (bf16_tile.hpp).
//   at the end  dL/dW1[j][k] = w2_j M[j][k],  dL/db1[j] = w2_j M[j][5],  dL/db2 = sum dy,
//               dL/dW2[j] = sum_s dy_s h_sj = sum_k W~1[j][k] M[j][k]   (h_sj = [pre_sj > 0] W~1[j] . x~_s).
// (The output layer as a masked sum on the matrix pipe — what the Fisher-vector pass gains 17 % from, kernels_mfma.hip —
// loses here: 0.259 against 0.219 ms per step; this kernel has one |pre| chain to replace, not a second layer-1 product
// with its LDS-resident operands, and 16 more matrix instructions per tile push it against the matrix pipe.)
// Algorithmic flops per sample: 3 x (2*5*128 + 2*128) = 4608 (forward + 2 x backward of the 5-128-1 MLP).
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "bf16_tile.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"

static_assert(bt::RANGE_WORDS == RL_RANGE_WORDS && bt::RANGE_ALLOC_WORDS == RL_RANGE_ALLOC_WORDS &&
                  bt::GUARD_POLICY == RL_GUARD_POLICY && bt::GUARD_CRITIC == RL_GUARD_CRITIC,
              "engine.hpp and bf16_tile.hpp describe the same range / veto words");

using bt::f32x16;
using bt::Frag;

constexpr int CRITIC_WAVES = 8;  // waves per workgroup, one workgroup per CU (two waves per SIMD: the
                                               // tile state — 64 accumulators of each pass, 48 weight-piece registers —
                                               // does not fit three)
#ifndef RL_C_FLUSH
#define RL_C_FLUSH 64
#endif
constexpr int C_FLUSH = RL_C_FLUSH;                     // f32 -> f64 flush period in tiles (2048 samples per accumulator: the accumulated error stays below a 128-sample f32 fma chain's, scripts/probe/mfma_bf16_mask.hip)

// -DRL_CRITIC_TIMESTAMPS (a timing build, scripts/build_variant.sh): wave 0 of every workgroup records the constant
// 100 MHz clock at seven points of the launch; the launcher prints the averages over the workgroups (round 6: where the
// ~7 us a launch costs beyond its tiles go)
#ifdef RL_CRITIC_TIMESTAMPS
__device__ uint64_t g_critic_ts[1024 * 8];
__device__ uint64_t g_critic_wave_end[1024 * 8];  // when each wave of a workgroup has finished its tiles
#define RL_TS(k)                                                                                   \
  do {                                                                                             \
    if (threadIdx.x == 0) g_critic_ts[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();    \
  } while (0)
#else
#define RL_TS(k)
#endif

// CH = 1: the critic step (above).  CH = 2 (round 6): the DQN gradient — mean((Q(s)[a] - target)^2) of the 5-128-2
// action-value MLP, dqn.rs:316-326 — as TWO critic steps side by side: the loss reaches the hidden layer through row a_s
// of the output weights only, so channel c (waves 0-3: c = 0, waves 4-7: c = 1) is the critic step of the one-output
// network (W1, b1, W2[c], b2[c]) over the samples with a_s = c, every other sample contributing zero.  Both channels walk
// the same tiles (the forward is computed twice) — and the two waves of a SIMD, one of each channel, overlap each other's
// matrix and vector work, which the one-wave-per-SIMD kernel of rounds 3-5 (two backward channels in one wave's whole
// register file: k_dqn_step_bf16, still the kernel of the in-kernel TD targets) could not: 19.0 -> 11 us per launch.
// The workgroup's row combines them: dW1[j][k] = W2[0][j] M_0[j][k] + W2[1][j] M_1[j][k] (db1 likewise),
// dW2[c][j] = sum_k W~1[j][k] M_c[j][k], db2[c] = sum of channel c's dL/dy, the loss the sum of both channels'.
template <int CH>
__global__ void __launch_bounds__(CRITIC_WAVES * 64, 2)
    k_critic_step_mfma(TrajDev tr, const float *__restrict__ params, const uint32_t *__restrict__ wimg,
                       double *__restrict__ slabA, double *__restrict__ slabB, float two_over_B, uint32_t P,
                       uint32_t share_old, uint32_t share_young) {
  constexpr int D = 5, H = 128, NT = bt::NT;
  constexpr int IMG = H * 7 + 2;  // per hidden unit: M[0..5] (slot 6 unused); then db2, loss
#ifndef RL_CRITIC_Y_IN_REGISTERS
  __shared__ __attribute__((aligned(16))) float Ysh[CRITIC_WAVES][32][bt::YROW];
#endif
  __shared__ double Acc[CRITIC_WAVES][IMG];  // f64 level of the two-level accumulation, one image per wave

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: tile indices stay scalar
  const int n = lane & 31, hf = lane >> 5;
  const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H;
  const int chan = CH == 2 ? (wave >= CRITIC_WAVES / 2 ? 1 : 0) : 0;  // (wave-uniform) the output this wave differentiates
  const float b2 = W2[CH * H + chan];
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  double *acc64 = Acc[wave];
  RL_TS(0);
  // (the wave's f64 image is not zeroed: its first flush stores — bt::flush — and every wave flushes at least once)
  bool flushed = false;

  // weight pieces of hidden unit 32 t + n in this half's slot order; w2; the linear half of relu (v of this half's
  // inputs: 2 hf, 2 hf + 1, and 4 or the bias)
  Frag fw[NT][3];
  float w2v[NT];
  float lv[3] = {0.0f, 0.0f, 0.0f};
  const bool guard = blockIdx.x == 0 && wave == 0 && tr.range != nullptr;  // the numeric range guard (bf16_tile.hpp)
  float gxmin = 0.0f, gxmax = 0.0f;
  if (guard) bt::range_bounds(tr.range, lane, gxmin, gxmax);
  // (the pieces come ready-made from the module's weight image, written by whoever wrote the parameters: bf16_tile.hpp)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bt::WRaw r;
    bt::wimg_load(wimg, t, lane, fw[t], r, CH);
    const float w2 = CH == 2 && chan == 1 ? r.w2[1] : r.w2[0];
    lv[0] = __builtin_fmaf(w2, r.wa, lv[0]);
    lv[1] = __builtin_fmaf(w2, r.wb, lv[1]);
    lv[2] = __builtin_fmaf(w2, r.wc, lv[2]);
    // the forward runs on weights scaled by 2^96 (relu' by conversion, bf16_tile.hpp); the |pre| chain takes the scale
    // back out through w2 (both exact)
    if (guard) {  // (one wave sees all 128 units; the two-channel form is the DQN gradient: the policy chain's words)
      constexpr int CHAIN = CH == 2 ? bt::GUARD_POLICY : bt::GUARD_CRITIC;
      bt::range_guard_img(r, hf, gxmin, gxmax, tr.range_err + CHAIN, bt::range_veto(tr.range, CHAIN));
    }
    w2v[t] = bt::FWD_UNSCALE * w2;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) lv[q] = lv[q] + __shfl_xor(lv[q], m, 64);  // over the 32 lanes of the half
  // backward accumulators (matrix pipe): dm[t][r] = sum over samples for hidden unit 32 t + row(r, hf), piece column n
  f32x16 dm[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dm[t] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double loss64 = 0.0, db2_64 = 0.0;
  float loss32 = 0.0f, db2_32 = 0.0f;  // per-lane f32 partials over one flush period (<= 64 tiles), then f64: the two levels
                                         // of every sum over samples here
  bt::wave_lds_fence();

  RL_TS(1);  // (after the weights have arrived: the cross-lane sums above wait for them)
  Frag selb[2];  // piece-column selection (B operand of the routing product)
  bt::sel_frags(lane, selb);
  // Tiles: the full ones in the loop, a ragged last one (B not a multiple of 32) after it on the wave whose turn it
  // is, through the same code with a per-lane `valid` — the loop itself carries no validity selects.  Tile indices are
  // wave-uniform (SGPRs); the operands come through buffer loads with a constant per-lane byte offset and the tile's
  // offset as the scalar operand: no vector address arithmetic per tile.
  const uint32_t B32 = (uint32_t)B, plane32 = (uint32_t)plane;
  const uint32_t n_full = B32 / 32u, tail = B32 & 31u;
  // The two waves of a SIMD do not progress alike: the older one (waves 0-3 of the workgroup, launched first) wins the
  // issue arbitration and used to finish its tiles at 0.71 of the launch, leaving the younger one alone — one wave per
  // SIMD, nothing to overlap its matrix instructions with — for the rest (profiles/r06_critic_step_timeline.txt).  So the
  // tiles are not dealt evenly: an older wave plays `share_old` virtual waves, a younger one `share_young`, and both
  // finish together.  (Virtual wave ids: workgroup-major, the older waves' first.)
  // (CH = 2: the two channels walk the same tiles — wave w and wave w + 4 are one virtual wave)
  if (CH == 2) share_old = share_young = 1u;
  const uint32_t per_wg = CH == 2 ? (uint32_t)(CRITIC_WAVES / 2) : (CRITIC_WAVES / 2) * (share_old + share_young);
  const uint32_t my_share = wave < CRITIC_WAVES / 2 ? share_old : share_young;
  const uint32_t my_first =
      blockIdx.x * per_wg + (CH == 2 ? (uint32_t)(wave & (CRITIC_WAVES / 2 - 1))
                                     : wave < CRITIC_WAVES / 2 ? (uint32_t)wave * share_old
                                                               : (CRITIC_WAVES / 2) * share_old +
                                                                     (uint32_t)(wave - CRITIC_WAVES / 2) * share_young);
  const uint32_t n_waves = gridDim.x * per_wg;  // virtual waves of the launch
  const bt::rsrc_t obs_r = bt::make_rsrc(tr.obs, (uint32_t)D * plane32 * 4u), tgt_r = bt::make_rsrc(tr.tgt, B32 * 4u);
  const bt::rsrc_t act_r = bt::make_rsrc(tr.action, B32);  // (CH = 2 only: the action taken selects the channel)
  const uint32_t off_a = ((uint32_t)(2 * hf) * plane32 + (uint32_t)n) * 4u, off_b = off_a + plane32 * 4u;
  const uint32_t off_c = (4u * plane32 + (uint32_t)n) * 4u, off_t = (uint32_t)n * 4u;
  int since_flush = 0;
#ifdef RL_CRITIC_Y_IN_REGISTERS
  const int ysrc = bt::row_sum_source(n);  // where fold_rows16 leaves the output-layer sum of sample n
#endif
  // per lane: features 2 hf, 2 hf + 1 and 4 of sample n, and its target
  struct TileOp {
    float xa, xb, xc, tgt;
    uint32_t act;
  };
  auto load_tile = [&](uint32_t g) {  // g: wave-uniform tile index (< 2^25: the launcher bounds the element count)
    TileOp o;
    const uint32_t soff = g * 128u;
    o.xa = bt::buf_f32(obs_r, off_a, soff);
    o.xb = bt::buf_f32(obs_r, off_b, soff);
    o.xc = bt::buf_f32(obs_r, off_c, soff);
    o.tgt = bt::buf_f32(tgt_r, off_t, soff);
    o.act = CH == 2 ? bt::buf_u8(act_r, (uint32_t)n, g * 32u) : 0u;
    return o;
  };

  auto tile = [&](auto ragged, TileOp op) {
    constexpr bool RAGGED = decltype(ragged)::value;
    const bool valid = RAGGED ? (uint32_t)n < tail : true;
    if (RAGGED) {  // (the observation planes extend past sample B - 1: what a padding lane read is not zero)
      op.xa = valid ? op.xa : 0.0f;
      op.xb = valid ? op.xb : 0.0f;
      op.xc = valid ? op.xc : 0.0f;
      op.tgt = valid ? op.tgt : 0.0f;
    }
    Frag fa[3];
    bt::input_frags(op.xa, op.xb, op.xc, valid, hf, fa);
    // ---- forward, one hidden tile at a time, software-pipelined: the matrix pipe works on hidden tile t + 1 while the
    // VALU does the partial y and the relu' mask of hidden tile t; the mask is packed as the backward's A operand
    Frag ga[NT][2];
    float yp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yp[r] = 0.0f;
    f32x16 c = bt::layer1(fa, fw[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x16 cn = c;
      if (t + 1 < NT) cn = bt::layer1(fa, fw[t + 1]);
#pragma unroll
      for (int r = 0; r < 16; ++r) yp[r] = __builtin_fmaf(__builtin_fabsf(c[r]), w2v[t], yp[r]);
      bt::mask_tile(c, ga[t]);  // relu'(pre): one conversion per two values
      c = cn;
    }
#ifndef RL_CRITIC_Y_IN_REGISTERS
    // ---- y: transpose the 16 partial sums per lane through LDS (row = sample, column = source lane)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ysh[wave][(r & 3) + 8 * (r >> 2) + 4 * hf][n] = yp[r];
    bt::wave_lds_fence();
    float part = bt::row_sum16v(&Ysh[wave][n][hf * 16]);
    {  // the linear half of relu: this half's inputs of sample n
      float lin = lv[0] * op.xa;
      lin = __builtin_fmaf(lv[1], op.xb, lin);
      lin = __builtin_fmaf(lv[2], hf == 0 ? op.xc : 1.0f, lin);
      part = part + lin;
    }
    float p0, p1;
    bt::both_halves(part, p0, p1);
    const float y = 0.5f * (p0 + p1) + b2;
#else
    // ---- y (A/B build, -DRL_CRITIC_Y_IN_REGISTERS): the 16 partial sums per lane folded over the half's 32 unit lanes in
    // registers (bf16_tile.hpp fold_rows16), then one permute brings sample n's sum to both lanes that own sample n.
    // Measured in round 5: 11 LDS instructions fewer, 18 vector instructions and 21 wait states more per tile — the same
    // 0.20 ms per step (scripts/critic_only.py, both builds on one box): the LDS round trip was not what the tile waits for.
    const float abs_sum = bt::row_sums_to_samples(bt::fold_rows16(yp, lane), ysrc);
    float lin = lv[0] * op.xa;  // the linear half of relu: this half's inputs of sample n, then the other half's
    lin = __builtin_fmaf(lv[1], op.xb, lin);
    lin = __builtin_fmaf(lv[2], hf == 0 ? op.xc : 1.0f, lin);
    float l0, l1;
    bt::both_halves(lin, l0, l1);
    const float y = 0.5f * (abs_sum + (l0 + l1)) + b2;
#endif
    const float d = y - op.tgt;
    // (CH = 2: only the samples whose action is this wave's channel count — loss, db2 and the backward alike)
    const bool mine = CH == 2 ? valid && op.act == (uint32_t)chan : valid;
    const float dy = mine ? d * two_over_B : 0.0f;
    if (mine) {  // (both halves hold sample n and count it; the reduction after the loop reads half 0 only)
      loss32 = __builtin_fmaf(d, d, loss32);
      db2_32 = db2_32 + dy;
    }
    // ---- backward: u[sample][k] = dy * x~_k as exact pieces (routed to the piece columns by a selection product),
    // masked sum over the samples on the matrix pipe
    Frag ub[2];
    bt::piece_frags_mfma(dy, op.xa, op.xb, op.xc, hf, selb, ub);
    bt::backward(ga, ub, dm);
#ifndef RL_CRITIC_Y_IN_REGISTERS
    bt::wave_lds_fence();  // Ysh is rewritten by the next tile
#endif
    if (++since_flush == C_FLUSH) {
      since_flush = 0;
      bt::flush(dm, acc64, 7, n, hf, !flushed);
      flushed = true;
      loss64 += (double)loss32;
      db2_64 += (double)db2_32;
      loss32 = db2_32 = 0.0f;
    }
  };

  for (uint32_t vw = 0; vw < my_share; ++vw) {
   const uint32_t wave_id = my_first + vw;
   if (wave_id < n_full) {
    // global loads run one tile ahead (past the wave's last tile: that tile again), into two named buffers that take
    // turns — no register moves.  (Two tiles ahead — by register moves or by rotating three named buffers through a
    // loop unrolled three times — is SLOWER, 0.237 against 0.220 ms per step, although a timing build without the loads
    // runs in 0.197: what the loads cost is issue slots, not exposed latency.)
    TileOp op_a = load_tile(wave_id), op_b = op_a;
#ifdef RL_CRITIC_TIMESTAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RL_TS(2);  // (the first tile's operands are here)
#endif
    for (uint32_t g = wave_id; g < n_full; g += 2 * n_waves) {
      const uint32_t g1 = g + n_waves, g2 = g1 + n_waves;
      op_b = load_tile(g1 < n_full ? g1 : g);
      tile(std::false_type{}, op_a);
      if (g1 >= n_full) break;
      op_a = load_tile(g2 < n_full ? g2 : g1);
      tile(std::false_type{}, op_b);
    }
   }
   if (tail != 0 && n_full % n_waves == wave_id) tile(std::true_type{}, load_tile(n_full));
  }
  RL_TS(3);
#ifdef RL_CRITIC_TIMESTAMPS
  if (lane == 0) g_critic_wave_end[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_memrealtime();
#endif
  if (since_flush != 0 || !flushed) {  // (a wave whose tile count is a multiple of the flush period has nothing left: at
                                       // the headline size every wave owns exactly 2 x C_FLUSH tiles, and this was a
                                       // third flush of zeros; a wave without tiles still defines its image)
    bt::flush(dm, acc64, 7, n, hf, !flushed);
    loss64 += (double)loss32;
    db2_64 += (double)db2_32;
  }
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
  RL_TS(4);
  __syncthreads();
  RL_TS(5);
  // sum the per-wave images in wave order, turn M into gradients and write the workgroup's slab row
  for (uint32_t p = threadIdx.x; p <= P; p += CRITIC_WAVES * 64) {
    auto tot = [&](int src) {
      double s = Acc[0][src];
#pragma unroll
      for (int w = 1; w < CRITIC_WAVES; ++w) s = s + Acc[w][src];
      return s;
    };
    auto totc = [&](int c, int src) {  // CH = 2: the four waves of channel c
      double s = Acc[4 * c][src];
#pragma unroll
      for (int w = 1; w < CRITIC_WAVES / 2; ++w) s = s + Acc[4 * c + w][src];
      return s;
    };
    double s;
    if (CH == 2) {
      if (p < (uint32_t)(H * D)) {
        const int j = p / D, k = p % D;
        s = totc(0, j * 7 + k) * (double)W2[j] + totc(1, j * 7 + k) * (double)W2[H + j];
      } else if (p < (uint32_t)(H * D + H)) {
        const int j = p - H * D;
        s = totc(0, j * 7 + 5) * (double)W2[j] + totc(1, j * 7 + 5) * (double)W2[H + j];
      } else if (p < (uint32_t)(H * D + H + 2 * H)) {
        const int q = p - H * D - H, c = q / H, j = q % H;
        s = totc(c, j * 7 + 5) * (double)b1[j];
#pragma unroll
        for (int k = 0; k < D; ++k) s += totc(c, j * 7 + k) * (double)W1[j * D + k];
      } else if (p < P) {
        s = totc((int)(p - (H * D + H + 2 * H)), H * 7);
      } else {
        s = tot(H * 7 + 1);
      }
    } else if (p < (uint32_t)(H * D)) {
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
#ifdef RL_CRITIC_TIMESTAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RL_TS(6);
#endif
}

// ---------------------------------------------------------------- launcher
bool launch_critic_step_v2(rl_traj *traj, const rl_mlp *critic, uint64_t B_total) {
  if (critic->general) return launch_gen_mfma(traj, critic, RL_GEN_CRITIC, nullptr, B_total, nullptr, 0.0f, 0.0f);
  if (traj->d.D != 5 || critic->hidden != 128 || critic->out_dim != 1) return false;
  if ((uint64_t)(traj->d.T + 1) * traj->d.n * 5 >= (1ull << 30)) return false;  // 32-bit element offsets in the kernel
  traj_ensure_range(traj);
  const uint32_t *wimg = wimg_ensure(critic);
  ProfScope ps(traj->eng, RL_K_CRITIC_FUSED);
  float two_over_B = 2.0f / (float)B_total;
  // persistent grid: one fat workgroup per CU (fewer, fatter workgroups = fewer slab rows for the reduction that follows
  // every launch), one 32-sample tile per wave and iteration
  const uint64_t n_tiles = (traj->B + 31) / 32, cus = (uint64_t)traj->eng->prop.multiProcessorCount;
  uint64_t nb = (n_tiles + CRITIC_WAVES - 1) / CRITIC_WAVES;
  if (nb > cus) nb = cus;
  traj->nbC = (uint32_t)nb;
  traj->last_rows = traj->nbC;
  // shares of the tiles for the older / the younger wave of a SIMD (the kernel says why); RL_CRITIC_SHARES=a:b overrides
  uint32_t share_old = bt::SHARE_OLD, share_young = bt::SHARE_YOUNG;
  if (const char *sh = std::getenv("RL_CRITIC_SHARES")) {
    unsigned a = 0, b = 0;
    if (std::sscanf(sh, "%u:%u", &a, &b) == 2 && a >= 1 && b >= 1 && a <= 64 && b <= 64) share_old = a, share_young = b;
  }
  TrajDev d = traj->d;
  if (!traj->guard_next_critic) d.range = nullptr;  // (the range guard: first critic launch of the call only, engine.hpp)
  traj->guard_next_critic = false;
  hipLaunchKernelGGL(k_critic_step_mfma<1>, dim3(traj->nbC), dim3(CRITIC_WAVES * 64), 0, traj->eng->stream, d,
                     critic->d_params, wimg, traj->slabA, traj->slabB, two_over_B, (uint32_t)critic->P, share_old,
                     share_young);
#ifdef RL_CRITIC_TIMESTAMPS
  if (std::getenv("RL_CRITIC_TS_PRINT")) {
    static int calls = 0;
    if (++calls % 16 == 0) {  // (every 16th launch: the host synchronises for it)
      std::vector<uint64_t> h(1024 * 8);
      (void)hipStreamSynchronize(traj->eng->stream);
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_critic_ts), h.size() * 8);
      double sum[7] = {0}, first = 1e300, last = 0;
      for (uint32_t b = 0; b < traj->nbC; ++b) {
        for (int k = 0; k < 7; ++k) sum[k] += (double)(h[b * 8 + k] - h[b * 8]) * 0.01;
        first = std::min(first, (double)h[b * 8] * 0.01);
        last = std::max(last, (double)h[b * 8 + 6] * 0.01);
      }
      {
        std::vector<uint64_t> we(1024 * 8);
        (void)hipMemcpyFromSymbol(we.data(), HIP_SYMBOL(g_critic_wave_end), we.size() * 8);
        double per[8] = {0};
        for (uint32_t b = 0; b < traj->nbC; ++b)
          for (int w = 0; w < 8; ++w) per[w] += (double)(we[b * 8 + w] - h[b * 8]) * 0.01;
        std::fprintf(stderr, "critic ts: tiles of wave 0..7 done at (us, mean over workgroups):");
        for (int w = 0; w < 8; ++w) std::fprintf(stderr, " %.1f", per[w] / traj->nbC);
        std::fprintf(stderr, "\n");
      }
      std::fprintf(stderr, "critic ts (us from a workgroup's start, mean of %u): weights %.2f  first tile %.2f  loop end %.2f  "
                   "flushed %.2f  barrier %.2f  end %.2f | first start to last end %.2f\n", traj->nbC,
                   sum[1] / traj->nbC, sum[2] / traj->nbC, sum[3] / traj->nbC, sum[4] / traj->nbC, sum[5] / traj->nbC,
                   sum[6] / traj->nbC, last - first);
    }
  }
#endif
  return true;
}

// The DQN gradient of a 5-128-2 action-value network over the minibatch workspace `mb` (targets in `adv`, actions in
// `action`): k_critic_step_mfma<2>.  False: shape not built (the caller has other kernels).
bool launch_dqn_step_pair(rl_traj *mb, const rl_mlp *qnet, uint64_t B_total) {
  if (mb->d.D != 5 || qnet->hidden != 128 || qnet->out_dim != 2 || qnet->general) return false;
  if ((uint64_t)(mb->d.T + 1) * mb->d.n * 5 >= (1ull << 30)) return false;  // 32-bit element offsets in the kernel
  const uint32_t *wimg = wimg_ensure(qnet);
  ProfScope ps(mb->eng, RL_K_POLICY_FUSED);
  const uint64_t n_tiles = (mb->B + 31) / 32, cus = (uint64_t)mb->eng->prop.multiProcessorCount;
  uint64_t nb = (n_tiles + CRITIC_WAVES / 2 - 1) / (CRITIC_WAVES / 2);  // four tile-walking wave pairs per workgroup
  if (nb > cus) nb = cus;
  mb->nbV2 = (uint32_t)nb;  // slab rows of this launch (the slabs are sized for any grid up to 8 x CUs)
  TrajDev d = mb->d;
  d.tgt = mb->d.adv;  // (the minibatch workspace keeps its targets where a trajectory keeps advantages)
  if (!mb->guard_next_policy) d.range = nullptr;  // (the range guard: first step of an update only, engine.hpp)
  mb->guard_next_policy = false;
  hipLaunchKernelGGL(k_critic_step_mfma<2>, dim3((uint32_t)nb), dim3(CRITIC_WAVES * 64), 0, mb->eng->stream, d,
                     qnet->d_params, wimg, mb->slabA, mb->slabB, 2.0f / (float)B_total, (uint32_t)qnet->P, 1u, 1u);
  RL_HIP_CHECK(hipGetLastError());
  return true;
}
