// kernels_mfma.hip — the fused full-batch policy passes (gradient, Fisher-vector product, evaluation) of the 5-128-2
// categorical policy on the bf16 matrix pipe.  The tile machinery is bf16_tile.hpp; the critic step (one output) is
// kernels_critic.hip, the DQN gradient (two outputs, two backward channels) is k_dqn_step_bf16 in kernels_dqn.hip.
// (Rounds 1-2 ran these passes on the f32 MFMA with the backward on the VALU; that kernel is gone — its last user was
// the DQN gradient.)
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "bf16_tile.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"
#include "policy_fast.hpp"

constexpr int V2_WAVES = 8;        // policy kernels: one workgroup of eight waves per CU (two per SIMD)

// ================================================================================================
// Policy passes on the bf16 matrix pipe (tile machinery: bf16_tile.hpp; every product exact, f32 accumulation).
//   PASS_INIT / PASS_PPO   gradient of the (clipped) surrogate: forward, per-sample math, masked-sum backward
//   PASS_JVP               Fisher-vector product J^T (diag p - p p^T) J v: forward for the relu' masks, tangent forward,
//                          per-sample metric, masked-sum backward
//   PASS_EVAL              surrogate loss and KL of candidate parameters: forward only
// Forward output layer (INIT / PPO / EVAL): relu through |x| as in the critic step — for each of the two logits
//   z_a = b2_a + (v_a . x~ + sum_j w2_aj |pre_j|) / 2,  v_ak = sum_j w2_aj W~1[j][k].
// Tangent forward (JVP): the metric only needs the DIFFERENCE of the two tangent logits,
//   dz_0 - dz_1 = sum_j relu'(pre_j) (w2d_j t_j + t2d_j pre_j) + (vb2_0 - vb2_1),   w2d = W2[0] - W2[1], t2d likewise,
// and  w2d_j t_j + t2d_j pre_j = x~ . Z_j  with  Z_jk = w2d_j V~1[j][k] + t2d_j W~1[j][k]  — linear in the inputs, so it
// is a second layer-1 product with Z as the weight matrix (its piece fragments are built once per launch and parked in
// LDS: 12 KB shared by the workgroup's waves), one multiply-clamp and one fma per (sample, unit).
// Backward: the logit gradient of a 2-way softmax is antisymmetric (dz_1 = -dz_0), so one channel g = (dz_0 - dz_1) / 2:
//   M_d[j][k] = sum_s [pre_sj > 0] g_s x~_sk,  dW1[j][k] = (W2[0][j] - W2[1][j]) M_d[j][k],
//   dW2[0][j] = -dW2[1][j] = sum_k W~1[j][k] M_d[j][k];  db2 keeps its two exact per-channel sums.
// Reference semantics: Trpo::update closure + HessianVectorProduct (src/torch/agents/policies/trpo.rs:97-146,
// src/torch/optimizers/conjugate_gradient.rs:262-339), Ppo::update (policies/ppo.rs:124-137), Categorical
// (src/torch/distributions/categorical.rs).
// ================================================================================================
constexpr int PB_FLUSH = 16;  // f32 -> f64 flush period in tiles (32 / 64: 2 % faster Fisher-vector products, measured; TRPO's
                              // CG wants the shorter f32 accumulation)

template <int MODE, int WAVES, bool FW_LDS = false>
__global__ void __launch_bounds__(WAVES * 64)
    k_policy_bf16(TrajDev tr, const float *__restrict__ params, const uint32_t *__restrict__ wimg,
                  const float *__restrict__ tangent, float *__restrict__ lp0, double *__restrict__ slabA,
                  double *__restrict__ slabB, float inv_B, uint32_t P, const int32_t *__restrict__ skip, float clip_lo,
                  float clip_hi, uint32_t share_old, uint32_t share_young) {
  using bt::Frag;
  constexpr int D = 5, H = 128, NT = bt::NT, A = 2;
  constexpr bool BWD = MODE != PASS_EVAL;
  constexpr bool JVP = MODE == PASS_JVP;
  constexpr int IW = 7;              // f64 image slots per hidden unit (six columns)
  constexpr int PIMG_M = H * IW + 5; // then db2[0], db2[1], sum0, sum1, sum2
  __shared__ __attribute__((aligned(16))) float Ysh[JVP ? 1 : WAVES][32][bt::YROW];
  __shared__ double Acc[WAVES][BWD ? PIMG_M : 4];
  __shared__ uint4 Fz[JVP ? bt::L2_KS : 1][64];  // JVP: A operands of the masked sum over the hidden units (pieces of Z)
  __shared__ uint4 Fw[FW_LDS ? NT * 3 : 1][64];  // FW_LDS: the weight fragments live in LDS, shared by the waves
  if (skip != nullptr && *skip != 0) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: tile indices stay scalar
  const int n = lane & 31, hf = lane >> 5;
  const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H,
                           *__restrict__ b2 = W2 + A * H;
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  double *acc64 = Acc[wave];
  bool flushed = false;  // (the wave's f64 image is not zeroed: its first flush stores, bt::flush)

  Frag fw[NT][3];
  // Only the DIFFERENCE of the two logits enters a two-way log-softmax (it is shift-invariant), so the output layer is
  // one chain with the differenced weights w2d = W2[0] - W2[1]; the passes agree with each other because all of them
  // (and log pi_0, stored by PASS_INIT) use this form.  The rollout's bit-exact forward is device_fns.hpp, not this.
  float w2d[NT];
  float lvd[3] = {0.0f, 0.0f, 0.0f};  // the linear half of relu (bf16_tile.hpp) for the differenced logit
  float tb2d = 0.0f;
  const bool guard = blockIdx.x == 0 && wave == 0 && tr.range != nullptr;  // the numeric range guard (bf16_tile.hpp)
  float gxmin = 0.0f, gxmax = 0.0f;
  if (guard) bt::range_bounds(tr.range, lane, gxmin, gxmax);
  // (the pieces come ready-made from the module's weight image, written by whoever wrote the parameters: bf16_tile.hpp)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bt::WRaw r;
    bt::wimg_load(wimg, t, lane, fw[t], r, 2);
    // the forward runs on weights scaled by 2^96 (relu' by conversion, bf16_tile.hpp); the |pre| chain of the gradient /
    // evaluation passes takes the scale back out through w2d (both exact); the Fisher-vector pass only needs the masks
    if (guard)  // (one wave sees all 128 units)
      bt::range_guard_img(r, hf, gxmin, gxmax, tr.range_err + bt::GUARD_POLICY, bt::range_veto(tr.range, bt::GUARD_POLICY));
    if (FW_LDS && wave == t) {
#pragma unroll
      for (int i = 0; i < 3; ++i) Fw[t * 3 + i][lane] = fw[t][i].x;
    }
    const float wd = r.w2[0] - r.w2[1];
    if (!JVP) {
      lvd[0] = __builtin_fmaf(wd, r.wa, lvd[0]);
      lvd[1] = __builtin_fmaf(wd, r.wb, lvd[1]);
      lvd[2] = __builtin_fmaf(wd, r.wc, lvd[2]);
    }
    w2d[t] = bt::FWD_UNSCALE * wd;
  }
  if (JVP) {
    // Z_jk = w2d_j V~1[j][k] + t2d_j W~1[j][k] (k = 5: the bias row): the tangent logit difference is
    // sum_j relu'(pre_j) (x~ . Z_j) + (vb2_0 - vb2_1) = sum_k x~_k q_k + ..., q = the masked sum of Z's rows
    const float *__restrict__ V1 = tangent, *__restrict__ vb1 = V1 + H * D, *__restrict__ V2 = vb1 + H;
    bt::l2_build(Fz, (int)threadIdx.x, WAVES * 64, [&](int j, int k) {
      const float wd = W2[j] - W2[H + j], t2d = V2[j] - V2[H + j];
      const float w = k < D ? W1[j * D + k] : b1[j], v = k < D ? V1[j * D + k] : vb1[j];
      return __builtin_fmaf(t2d, w, wd * v);
    });
  }
  if (JVP || FW_LDS) __syncthreads();
  if (JVP) {
    const float *__restrict__ vb2 = tangent + H * D + H + A * H;
    tb2d = vb2[0] - vb2[1];
  } else {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) lvd[q] = lvd[q] + __shfl_xor(lvd[q], m, 64);
  }
  const float b2d = b2[0] - b2[1];
  Frag selb[2], idb[2];  // piece-column selection, identity (B operands of the routing / transposing products)
  if (BWD) bt::sel_frags(lane, selb);
  if (JVP) bt::ident_frags(lane, idb);
  bt::f32x16 dm[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dm[t] = (bt::f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double sum0 = 0.0, sum1 = 0.0, sum2 = 0.0, db2_0 = 0.0, db2_1 = 0.0;  // owner-lane f64 sums ...
  // ... fed through per-lane f32 partials over one flush period (PB_FLUSH tiles): the two levels of every sum over
  // samples in this library, and a conversion + an f64 addition less per sum and tile
  float s0f = 0.0f, s1f = 0.0f, s2f = 0.0f, d0f = 0.0f, d1f = 0.0f;
  auto fold = [&]() {
    sum0 += (double)s0f;
    sum1 += (double)s1f;
    sum2 += (double)s2f;
    db2_0 += (double)d0f;
    db2_1 += (double)d1f;
    s0f = s1f = s2f = d0f = d1f = 0.0f;
  };
  bt::wave_lds_fence();

  // sum over the 32 source lanes of 16 per-lane partials (+ this half's linear part): LDS transpose, the result for
  // sample n in both halves
  auto lane_sum = [&](const float(&yp)[16], float lin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) Ysh[JVP ? 0 : wave][(r & 3) + 8 * (r >> 2) + 4 * hf][n] = yp[r];
    bt::wave_lds_fence();
    float part = bt::row_sum16v(&Ysh[JVP ? 0 : wave][n][hf * 16]);
    part = part + lin;
    float p0, p1;
    bt::both_halves(part, p0, p1);
    bt::wave_lds_fence();
    return p0 + p1;
  };

  // Tiles: the full ones in the loop, a ragged last one (B not a multiple of 32) after it on the wave whose turn it is,
  // through the same code with a per-lane `valid` — the loop carries no validity selects.  Tile indices are wave-uniform
  // (SGPRs); the operands come through buffer loads with a constant per-lane byte offset and the tile's offset as the
  // scalar operand: no vector address arithmetic per tile (kernels_critic.hip has the same loop).
  const uint32_t B32 = (uint32_t)B, plane32 = (uint32_t)plane;
  const uint32_t n_full = B32 / 32u, tail = B32 & 31u;
  // (the tiles are dealt 5 : 3 between the older and the younger wave of a SIMD: bf16_tile.hpp SHARE_OLD; the
  // sixteen-wave evaluation pass deals evenly — its launcher passes 1 : 1)
  if (WAVES != 8) share_old = share_young = 1u;  // (compile-time for the sixteen-wave form: its loop below folds away)
  const uint32_t per_wg = (WAVES / 2) * (share_old + share_young);
  const uint32_t my_share = wave < WAVES / 2 ? share_old : share_young;
  const uint32_t my_first = blockIdx.x * per_wg + (wave < WAVES / 2 ? (uint32_t)wave * share_old
                                                   : (WAVES / 2) * share_old + (uint32_t)(wave - WAVES / 2) * share_young);
  const uint32_t n_waves = gridDim.x * per_wg;  // virtual waves of the launch
  const bt::rsrc_t obs_r = bt::make_rsrc(tr.obs, (uint32_t)D * plane32 * 4u), lp0_r = bt::make_rsrc(lp0, 2u * B32 * 4u);
  const bt::rsrc_t adv_r = bt::make_rsrc(tr.adv, B32 * 4u), act_r = bt::make_rsrc(tr.action, B32);
  const uint32_t off_a = ((uint32_t)(2 * hf) * plane32 + (uint32_t)n) * 4u, off_b = off_a + plane32 * 4u;
  const uint32_t off_c = (4u * plane32 + (uint32_t)n) * 4u, off_s = (uint32_t)n * 4u, off_s1 = off_s + B32 * 4u;
  int since_flush = 0;
  // per lane: features 2 hf, 2 hf + 1 and 4 of sample n, and the per-sample scalars the pass needs (log pi_0 of both
  // actions; advantage and action) — all requested one tile ahead
  struct TileOp {
    float xa, xb, xc, l0, l1, adv;
    int act;
  };
  auto load_tile = [&](uint32_t g) {  // g: wave-uniform tile index
    TileOp o;
    const uint32_t soff = g * 128u;
    o.xa = bt::buf_f32(obs_r, off_a, soff);
    o.xb = bt::buf_f32(obs_r, off_b, soff);
    o.xc = bt::buf_f32(obs_r, off_c, soff);
    o.l0 = o.l1 = o.adv = 0.0f;
    o.act = 0;
    if (MODE != PASS_INIT) {  // log pi_0 (written by PASS_INIT, read by the others)
      o.l0 = bt::buf_f32(lp0_r, off_s, soff);
      o.l1 = bt::buf_f32(lp0_r, off_s1, soff);
    }
    if (!JVP) {
      o.adv = bt::buf_f32(adv_r, off_s, soff);
      o.act = (int)bt::buf_u8(act_r, (uint32_t)n, g * 32u);
    }
    return o;
  };
  auto tile = [&](auto ragged, TileOp op_in, uint32_t g) {
    constexpr bool RAGGED = decltype(ragged)::value;
    struct {
      float xa, xb, xc, l0, l1, adv;
      int act;
      bool valid;
    } op;
    op.valid = RAGGED ? (uint32_t)n < tail : true;
    // (the arrays extend past sample B - 1 or the loads return 0 there: a padding lane's operands are zeroed)
    op.xa = op.valid ? op_in.xa : 0.0f;
    op.xb = op.valid ? op_in.xb : 0.0f;
    op.xc = op.valid ? op_in.xc : 0.0f;
    op.l0 = op.valid ? op_in.l0 : 0.0f;
    op.l1 = op.valid ? op_in.l1 : 0.0f;
    op.adv = op.valid ? op_in.adv : 0.0f;
    op.act = op.valid ? op_in.act : 0;
    const size_t sidx = (size_t)g * 32 + n;
    Frag fa[3];
    bt::input_frags(op.xa, op.xb, op.xc, op.valid, hf, fa);
    Frag ga[NT][2];
    float y0[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) y0[r] = 0.0f;
    bt::f32x16 q = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto fwd = [&](int t) {
      if (FW_LDS) {
        Frag f[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) f[i].x = Fw[t * 3 + i][lane];
        return bt::layer1(fa, f);
      }
      return bt::layer1(fa, fw[t]);
    };
    bt::f32x16 c = fwd(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bt::f32x16 cn = c;
      if (t + 1 < NT) cn = fwd(t + 1);
      if (!JVP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) y0[r] = __builtin_fmaf(__builtin_fabsf(c[r]), w2d[t], y0[r]);
      }
      if (BWD) bt::mask_tile(c, ga[t]);  // relu'(pre): one conversion per two values
      if (JVP) q = bt::masked_sum_tile(ga[t], idb, Fz, t, lane, q);
      c = cn;
    }
    float s0;
    if (JVP) {
      float p0, p1;
      bt::both_halves(bt::l2_dot(q, op.xa, op.xb, hf == 0 ? op.xc : 1.0f), p0, p1);
      s0 = p0 + p1;
    } else {
      float lin = lvd[0] * op.xa;
      lin = __builtin_fmaf(lvd[1], op.xb, lin);
      lin = __builtin_fmaf(lvd[2], hf == 0 ? op.xc : 1.0f, lin);
      s0 = 0.5f * lane_sum(y0, lin);
    }
    // ---- per-sample math on the owner lanes (lane n and n + 32 both hold sample n)
    float dz0 = 0.0f, dz1 = 0.0f;
    if (JVP) {
      const float delta = s0 + tb2d;
      const SoftPair old = soft_pair(op.l0 - op.l1);  // pi_0 from its stored log-probabilities
      dz0 = op.valid ? (old.p[0] * old.p[1]) * delta * inv_B : 0.0f;
      dz1 = -dz0;
    } else {
      const float adv = op.adv;
      const int act = op.act;
      const SoftPair sp = soft_pair(s0 + b2d);
      const float(&lp)[2] = sp.lp;
      if (MODE == PASS_PPO) {
        // clipped surrogate (policies/ppo.rs:124-137); see k_policy_pass for the tie rules of minimum()/clamp()
        const float l0a = act == 0 ? op.l0 : op.l1;
        const float lpa = act == 0 ? lp[0] : lp[1];
        const float ratio = fast_expf(lpa - l0a);
        const float clipped = ratio < clip_lo ? clip_lo : (ratio > clip_hi ? clip_hi : ratio);
        const float u1 = ratio * adv, u2 = clipped * adv;
        const bool inside = ratio >= clip_lo && ratio <= clip_hi;
        const float gr = u1 < u2 ? adv : (u1 > u2 ? (inside ? adv : 0.0f) : (inside ? adv : 0.5f * adv));
        const float cc = -(gr * ratio) * inv_B;
        dz0 = op.valid ? cc * ((act == 0 ? 1.0f : 0.0f) - sp.p[0]) : 0.0f;
        dz1 = op.valid ? cc * ((act == 1 ? 1.0f : 0.0f) - sp.p[1]) : 0.0f;
        if (op.valid) s0f = s0f + (u1 < u2 ? u1 : u2);  // (both halves count sample n; the final reduction reads half 0)
      } else if (MODE == PASS_INIT) {
        if (op.valid && hf == 0) {
          lp0[sidx] = lp[0];
          lp0[B + sidx] = lp[1];
        }
        const float lpa = act == 0 ? lp[0] : lp[1];
        const float ratio = fast_expf(lpa - lpa);
        const float cc = -(ratio * adv) * inv_B;
        const float pa0 = sp.p[0], pa1 = sp.p[1];
        dz0 = op.valid ? cc * ((act == 0 ? 1.0f : 0.0f) - pa0) : 0.0f;
        dz1 = op.valid ? cc * ((act == 1 ? 1.0f : 0.0f) - pa1) : 0.0f;
        const float cl0 = lp[0] < -3.402823466e+38f ? -3.402823466e+38f : lp[0];
        const float cl1 = lp[1] < -3.402823466e+38f ? -3.402823466e+38f : lp[1];
        float ent = cl0 * pa0;
        ent += cl1 * pa1;
        if (op.valid) {
          s0f = __builtin_fmaf(ratio, adv, s0f);
          s1f = s1f - ent;
          s2f = __builtin_fmaf(lpa, adv, s2f);
        }
      } else {  // PASS_EVAL
        const float l00 = op.l0, l01 = op.l1;
        const float lpa = act == 0 ? lp[0] : lp[1];
        const float l0a = act == 0 ? l00 : l01;
        const float ratio = fast_expf(lpa - l0a);
        float rel0 = l00 - lp[0], rel1 = l01 - lp[1];
        if (rel0 < -3.402823466e+38f) rel0 = -3.402823466e+38f;
        if (rel1 < -3.402823466e+38f) rel1 = -3.402823466e+38f;
        const SoftPair old = soft_pair(l00 - l01);  // pi_0 from its stored log-probabilities
        float kl = rel0 * old.p[0];
        kl += rel1 * old.p[1];
        if (op.valid) {
          s0f = __builtin_fmaf(ratio, adv, s0f);
          s1f = s1f + kl;
        }
      }
    }
    if (BWD) {
      d0f = d0f + dz0;  // (dz is 0 on padding lanes)
      d1f = d1f + dz1;
      // one channel: u[sample][k] = g * x~_k with g = (dz_0 - dz_1) / 2, masked sum over the samples on the matrix pipe
      Frag ub[2];
      bt::piece_frags_mfma(0.5f * (dz0 - dz1), op.xa, op.xb, op.xc, hf, selb, ub);
      bt::backward(ga, ub, dm);
      if (++since_flush == PB_FLUSH) {
        since_flush = 0;
        bt::flush(dm, acc64, IW, n, hf, !flushed);
        flushed = true;
        fold();
      }
    } else if (++since_flush == PB_FLUSH) {
      since_flush = 0;
      fold();
    }
  };
  for (uint32_t vw = 0; vw < my_share; ++vw) {
   const uint32_t wave_id = my_first + vw;
   if (wave_id < n_full) {
    // loads run one tile ahead (past the wave's last tile: that tile again), into two named buffers that take turns
    // (the Fisher-vector pass has no registers for a second buffer: one buffer and a move per operand there)
    TileOp op_a = load_tile(wave_id), op_b = op_a;
    if (JVP) {
      for (uint32_t g = wave_id; g < n_full; g += n_waves) {
        const uint32_t g1 = g + n_waves;
        op_b = load_tile(g1 < n_full ? g1 : g);
        tile(std::false_type{}, op_a, g);
        op_a = op_b;
      }
    } else {
      for (uint32_t g = wave_id; g < n_full; g += 2 * n_waves) {
        const uint32_t g1 = g + n_waves, g2 = g1 + n_waves;
        op_b = load_tile(g1 < n_full ? g1 : g);
        tile(std::false_type{}, op_a, g);
        if (g1 >= n_full) break;
        op_a = load_tile(g2 < n_full ? g2 : g1);
        tile(std::false_type{}, op_b, g1);
      }
    }
   }
   if (tail != 0 && n_full % n_waves == wave_id) tile(std::true_type{}, load_tile(n_full), n_full);
  }
  if (BWD && (since_flush != 0 || !flushed)) bt::flush(dm, acc64, IW, n, hf, !flushed);  // (nothing left when the last
                                                                // tile ended a flush period; a wave without tiles
                                                                // still defines its image)
  fold();
  auto xlane = [](double v, int mask) {
    uint64_t bits = rl_f64_bits(v);
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, mask, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), mask, 64);
    return rl_f64_from_bits(((uint64_t)hi << 32) | lo);
  };
  double r0 = hf == 0 ? sum0 : 0.0, r1 = hf == 0 ? sum1 : 0.0, r2 = hf == 0 ? db2_0 : 0.0, r3 = hf == 0 ? db2_1 : 0.0;
  double r4 = hf == 0 ? sum2 : 0.0;
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) {
    r0 = r0 + xlane(r0, s);
    r1 = r1 + xlane(r1, s);
    r2 = r2 + xlane(r2, s);
    r3 = r3 + xlane(r3, s);
    if (MODE == PASS_INIT) r4 = r4 + xlane(r4, s);
  }
  constexpr int TAIL = BWD ? H * IW : 0;
  if (lane == 0) {
    if (BWD) {
      acc64[TAIL + 0] = r2;
      acc64[TAIL + 1] = r3;
      acc64[TAIL + 2] = r0;
      acc64[TAIL + 3] = r1;
      if (MODE == PASS_INIT) acc64[TAIL + 4] = r4;
    } else {
      acc64[0] = r0;
      acc64[1] = r1;
    }
  }
  __syncthreads();
  auto tot = [&](int src) {
    double s = Acc[0][src];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) s = s + Acc[w][src];
    return s;
  };
  if (BWD) {
    for (uint32_t p = threadIdx.x; p < P; p += WAVES * 64) {
      double s = 0.0;
      if (p < (uint32_t)(H * D)) {  // M_0 = M_d, M_1 = -M_d
        int j = p / D, k = p % D;
        s = tot(j * IW + k) * ((double)W2[j] - (double)W2[H + j]);
      } else if (p < (uint32_t)(H * D + H)) {
        int j = p - H * D;
        s = tot(j * IW + 5) * ((double)W2[j] - (double)W2[H + j]);
      } else if (p < (uint32_t)(H * D + H + A * H)) {
        int q = p - H * D - H, a = q / H, j = q % H;
        s = tot(j * IW + 5) * (double)b1[j];
#pragma unroll
        for (int k = 0; k < D; ++k) s += tot(j * IW + k) * (double)W1[j * D + k];
        if (a == 1) s = -s;
      } else {
        s = tot(TAIL + (int)(p - (H * D + H + A * H)));
      }
      slabA[(size_t)blockIdx.x * P + p] = s;
    }
  }
  if (threadIdx.x < 4) {
    double v = 0.0;
    if (BWD) {
      if (!JVP && threadIdx.x < 2) v = tot(TAIL + 2 + threadIdx.x);
      if (MODE == PASS_INIT && threadIdx.x == 2) v = tot(TAIL + 4);
    } else if (threadIdx.x < 2) {
      v = tot(threadIdx.x);
    }
    slabB[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
  }
}

// gradient (PASS_INIT), Fisher-vector product (PASS_JVP) or loss/KL evaluation (PASS_EVAL) in one launch
bool launch_policy_v2(rl_traj *traj, const rl_mlp *policy, int mode, const float *d_tangent, uint64_t B_total,
                      const int32_t *d_skip, float clip_lo, float clip_hi) {
  if (policy->general) return launch_gen_mfma(traj, policy, mode, d_tangent, B_total, d_skip, clip_lo, clip_hi);
  if (traj->d.D != 5 || policy->hidden != 128 || policy->out_dim != 2) return false;
  if (mode == PASS_DQN) return false;  // k_dqn_step_bf16 (kernels_dqn.hip)
  if ((uint64_t)(traj->d.T + 1) * traj->d.n * 5 >= (1ull << 30)) return false;  // 32-bit element offsets in the kernels
  traj_ensure_range(traj);
  const uint32_t *wimg = wimg_ensure(policy);
  ProfScope ps(traj->eng, mode == PASS_JVP ? RL_K_POLICY_FVP : RL_K_POLICY_FUSED);
  float inv_B = 1.0f / (float)B_total;
  traj->last_rows = traj->nbV2;
  dim3 g(traj->nbV2), b(V2_WAVES * 64);
  hipStream_t s = traj->eng->stream;
  uint32_t P = (uint32_t)policy->P;
  uint32_t share_old = bt::SHARE_OLD, share_young = bt::SHARE_YOUNG;  // (RL_CRITIC_SHARES=a:b: A/B runs)
  if (const char *sh = std::getenv("RL_CRITIC_SHARES")) {
    unsigned a = 0, b2 = 0;
    if (std::sscanf(sh, "%u:%u", &a, &b2) == 2 && a >= 1 && b2 >= 1 && a <= 64 && b2 <= 64) share_old = a, share_young = b2;
  }
  TrajDev d = traj->d;
  if (!traj->guard_next_policy) d.range = nullptr;  // (the range guard: first policy launch of the call only, engine.hpp)
  traj->guard_next_policy = false;
#define BLAUNCH(MM)                                                                                                  \
  hipLaunchKernelGGL((k_policy_bf16<MM, V2_WAVES>), g, b, 0, s, d, policy->d_params, wimg, d_tangent, traj->lp0,       \
                     traj->slabA, traj->slabB, inv_B, P, d_skip, clip_lo, clip_hi, share_old, share_young)
  if (mode == PASS_INIT) BLAUNCH(PASS_INIT);
  else if (mode == PASS_JVP) BLAUNCH(PASS_JVP);
  else if (mode == PASS_PPO) BLAUNCH(PASS_PPO);
  else {
    // the evaluation pass has no backward state: with the weight fragments in LDS it fits twelve waves per CU (152
    // VGPRs) — 0.155 ms per launch at 8.4 M samples against 0.206 ms at eight (sixteen waves: 17 spills, no gain)
    constexpr int EVAL_WAVES = 16;
    hipLaunchKernelGGL((k_policy_bf16<PASS_EVAL, EVAL_WAVES, true>), g, dim3(EVAL_WAVES * 64), 0, s, d,
                       policy->d_params, wimg, d_tangent, traj->lp0, traj->slabA, traj->slabB, inv_B, P, d_skip, clip_lo,
                       clip_hi, 1u, 1u);
  }
#undef BLAUNCH
  return true;
}
