// kernels_gen_mfma.hip — the update passes of MLPs with SEVERAL hidden layers (MlpConfig.hidden_sizes, ff/mlp.rs:13-34;
// any rl_activation) as one fused launch on the bf16 matrix pipe: forward, per-sample loss terms, backward and weight
// gradients of one 32-sample tile per wave and iteration, no activation ever written to HBM.  (kernels_general.hip runs
// the same passes as one launch per layer over [unit][sample] planes: 178 ms for a TRPO + critic period of two 64-unit
// layers at 16,384 lanes, against 3 ms for the fused single-hidden-layer kernels.)
//
// Shapes: 1..3 hidden layers of at most 64 units (two 32-unit tiles; narrower layers are zero-padded) or one hidden layer
// of at most 128 (four tiles: the same 144 accumulator registers), at most 7 inputs, at most 2 outputs.  Every matrix product is an exact-piece product: both factors are split into three bf16
// pieces and the six piece pairs that matter are multiplied on v_mfma_f32_32x32x16_bf16 with f32 accumulation
// (bf16_tile.hpp) — the weights by rounding (|p1| <= 2^-9 |w|, |p2| <= 2^-18 |w|), activations by truncation, deltas by
// rounding, so the dropped pairs stay below 2^-24 of the product.
//
// Orientation: a tile of layer values X is an accumulator tile with the UNIT in the registers (row (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5)) and the SAMPLE on the lane (column lane & 31).
//   chains   Z_{l+1}^T = W_{l+1} A_l^T  and  D_l^T = W_{l+1}^T D_{l+1}^T  contract over the row index of the tile that
//            feeds them: the tile's registers, split and packed in place, ARE the B operand (bf16_tile.hpp's acc_row
//            order); the weights are A operands, built once per launch in that order and parked in LDS.
//   weight gradients  dW_l = D_l A_{l-1}^T contract over the SAMPLE — the lane index of both tiles.  Each is transposed on
//            the matrix pipe first: X^T = X^T . I with the packed pieces of X as A operand and an identity selection as B
//            (every entry one exact piece times one), packed by conversions: the unit on the lane, the samples in the
//            registers, the operand layout of a product that sums over samples.  Bias gradients are one more column: a
//            product with a one-hot column selector, all layers into one accumulator tile.
// The weight-gradient accumulators live in registers (f32 over at most 64 tiles = 2,048 samples, the two-level scheme
// of this library) and are added, as f64, to the wave's own slab row; k_reduce sums the rows.  One wave per SIMD: the
// accumulators of a 64-64 network are 144 registers next to the activations and operand pieces.
//
// Reference semantics: the same as kernels_mfma.hip / kernels_critic.hip (Trpo / Ppo closures, ValuesOpt::update);
// activations: ff/activation.rs:85-92.
#include <mutex>
#include <set>

#include "abi_internal.hpp"
#include "bf16_tile.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"
#include "policy_fast.hpp"

namespace {

using bt::f32x16;
using bt::Frag;

constexpr int GWAVES = 4;      // waves per workgroup (one per SIMD, 512 registers each)
constexpr int GM_FLUSH = 64;   // f32 -> f64 flush period in tiles
constexpr int GM_MAX_IN = 7, GM_MAX_HIDDEN = 3;

struct GmArgs {
  const float *params, *tangent;  // tangent: PASS_JVP only, laid out like params
  int in_dim, out_dim, act, out_act;
  int width[GM_MAX_HIDDEN];
  uint32_t off[GM_MAX_HIDDEN + 1];  // parameter offset of layer l's weights [N][K]; its bias follows them
  uint32_t P;
};

// fragment groups of the LDS weight image (three piece fragments each): forward layer 0 [ot], forward hidden layer l
// [ot][ks], forward output [ks], backward hidden layer l [it][ks], backward output [it]
// (GW = 32-unit tiles per hidden layer: 2 for up to three layers of at most 64 units, 4 for one layer of at most 128)
constexpr int gm_fh(int GW, int l) { return GW + (l - 1) * 2 * GW * GW; }
constexpr int gm_fo(int GW, int NL) { return GW + (NL - 1) * 2 * GW * GW; }
constexpr int gm_bh(int GW, int NL, int l) { return gm_fo(GW, NL) + 2 * GW + (l - 1) * 2 * GW * GW; }
constexpr int gm_bo(int GW, int NL) { return gm_bh(GW, NL, NL); }
constexpr int gm_groups(int GW, int NL) { return gm_bo(GW, NL) + GW; }
// PASS_JVP: the forward fragments of the tangent parameters follow, in the same order (layer 0, hidden, output)
constexpr int gm_fwd_groups(int GW, int NL) { return gm_fo(GW, NL) + 2 * GW; }
constexpr int gm_all_groups(int GW, int NL, bool jvp) { return gm_groups(GW, NL) + (jvp ? gm_fwd_groups(GW, NL) : 0); }
constexpr size_t gm_lds_bytes(int GW, int NL, bool jvp) {
  return (size_t)gm_all_groups(GW, NL, jvp) * 3 * 64 * 16 + (size_t)NL * 32 * GW * 4 * (jvp ? 2 : 1);
}

__device__ __forceinline__ constexpr int urow(int r, int kb) { return (r & 3) + 8 * (r >> 2) + 4 * kb; }

__device__ __forceinline__ float gm_act(int act, float x) {
  if (act == RL_ACT_RELU) return __builtin_fmaxf(x, 0.0f);
  if (act == RL_ACT_SIGMOID) return fast_sigmoidf(x);
  if (act == RL_ACT_TANH) return fast_tanhf(x);
  return x;
}
__device__ __forceinline__ float gm_slope(int act, float y) {
  if (act == RL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (act == RL_ACT_SIGMOID) return y * (1.0f - y);
  if (act == RL_ACT_TANH) return __builtin_fmaf(-y, y, 1.0f);
  return 1.0f;
}
// a whole tile at a time: one (wave-uniform) branch per tile, straight-line code inside
__device__ __forceinline__ void gm_act_tile(int act, f32x16 &c) {
  if (act == RL_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = __builtin_fmaxf(c[r], 0.0f);
  } else if (act == RL_ACT_SIGMOID) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = fast_sigmoidf(c[r]);
  } else if (act == RL_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = fast_tanhf(c[r]);
  }
}
// c *= act'(.) given the layer's outputs y
__device__ __forceinline__ void gm_slope_tile(int act, f32x16 &c, const f32x16 &y) {
  if (act == RL_ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = y[r] > 0.0f ? c[r] : 0.0f;
  } else if (act == RL_ACT_SIGMOID) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = c[r] * (y[r] * (1.0f - y[r]));
  } else if (act == RL_ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = c[r] * __builtin_fmaf(-y[r], y[r], 1.0f);
  }
}

// the B operand pieces of k-step q of an accumulator tile (registers 8 q .. 8 q + 7), by truncation (activations) ...
__device__ __forceinline__ void pieces_trunc(const f32x16 &t, int q, Frag (&x)[3]) {
  uint32_t p[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bt::split3t(t[8 * q + e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) x[c].u[i] = bt::pkh(p[c][2 * i], p[c][2 * i + 1]);
}
// ... and by rounding (deltas)
__device__ __forceinline__ void pieces_round(const f32x16 &t, int q, Frag (&x)[3]) {
  uint32_t p[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bt::split3(t[8 * q + e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) x[c].u[i] = bt::pk(p[c][2 * i], p[c][2 * i + 1]);
}

// acc += W X for one k-step: the six piece pairs that matter (w0 x0, w0 x1, w1 x0, w1 x1, w0 x2, w2 x0)
__device__ __forceinline__ f32x16 prod6(f32x16 acc, const Frag (&w)[3], const Frag (&x)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0].v, x[0].v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0].v, x[1].v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1].v, x[0].v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1].v, x[1].v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0].v, x[2].v, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2].v, x[0].v, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 prod6_lds(f32x16 acc, const uint4 (*img)[64], int grp, int lane, const Frag (&x)[3]) {
  Frag w[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) w[c].x = img[grp * 3 + c][lane];
  return prod6(acc, w, x);
}

// X^T of a tile given as packed pieces xb[q][piece] (k-steps q < QS): the pieces of X^T as operands of a product that
// sums over the SAMPLE, xt[sample k-step][piece]
template <int QS>
__device__ __forceinline__ void transpose_pieces(const Frag (&xb)[QS][3], const Frag (&idb)[2], Frag (&xt)[2][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x16 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < QS; ++q) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[q][c].v, idb[q].v, d, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) xt[s][c].u[i] = bt::pack_bf16(d[8 * s + 2 * i], d[8 * s + 2 * i + 1]);
  }
}

// dW += D^T-pieces x A^T-pieces over both sample k-steps
__device__ __forceinline__ f32x16 wgrad_tile(f32x16 acc, const Frag (&dt)[2][3], const Frag (&at)[2][3]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) acc = prod6(acc, dt[s], at[s]);
  return acc;
}
// bias column: sum over the samples of D^T's three pieces into column `c` of the bias tile
__device__ __forceinline__ f32x16 bias_tile(f32x16 acc, const Frag (&dt)[2][3], int c, int m) {
  Frag e;
  e.u[0] = e.u[1] = e.u[2] = e.u[3] = m == c ? 0x3F803F80u : 0u;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 3; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dt[s][p].v, e.v, acc, 0, 0, 0);
  return acc;
}

// The pair kernel's image: the same groups without the output layer's forward fragments — of their 32 rows only the
// module's (at most two) outputs are non-zero, so they live apart, one 16-byte entry per (group, piece, output, lane half)
// instead of 64: 1.5 KB instead of 12 KB (24 KB with the tangent's), which is what lets four pairs of waves and the
// tangent image share a CU's LDS.
constexpr int cp_bh(int GW, int NL, int l) { return GW + (NL - 1) * 2 * GW * GW + (l - 1) * 2 * GW * GW; }
constexpr int cp_bo(int GW, int NL) { return cp_bh(GW, NL, NL); }
constexpr int cp_groups(int GW, int NL) { return cp_bo(GW, NL) + GW; }
constexpr int cp_tan_groups(int GW, int NL) { return GW + (NL - 1) * 2 * GW * GW; }

// ---- the weight image: every fragment in the acc_row order of the tile it meets.  `src`: the parameters (forward and
// backward fragments from group g0 = 0) or the tangent (forward fragments only, from group TG); every thread of the
// workgroup (`nthreads`) calls it, the caller synchronises.
// `outc` != NULL: the compact layout above (output-layer forward fragments into outc[(ks * 3 + piece) * 4 + 2 j + half]).
template <int NL, int GW>
__device__ __forceinline__ void gm_build_image(const GmArgs &g, const float *__restrict__ src, uint4 (*img)[64], int g0,
                                               bool with_backward, float *bias_out, int nthreads, uint4 *outc = nullptr) {
  const bool compact = outc != nullptr;
  auto Kof = [&](int l) { return l == 0 ? g.in_dim : g.width[l - 1]; };
  auto Nof = [&](int l) { return l == NL ? g.out_dim : g.width[l]; };
  auto put_group = [&](int grp, int ln, const float (&v)[8]) {
    uint32_t p[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bt::split3(v[e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      img[grp * 3 + c][ln] = make_uint4(bt::pk(p[c][0], p[c][1]), bt::pk(p[c][2], p[c][3]), bt::pk(p[c][4], p[c][5]),
                                        bt::pk(p[c][6], p[c][7]));
  };
  // forward, layer 0 [ot]: inputs 4 hh + e (e < 4), the bias as input `in_dim`
  for (int idx = threadIdx.x; idx < GW * 64; idx += nthreads) {
    const int ot = idx >> 6, ln = idx & 63, mm = ln & 31, hh = ln >> 5;
    const int unit = ot * 32 + mm, N = Nof(0);
    const float *W = src + g.off[0];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = bt::acc_row(0, hh, e);
      v[e] = unit < N ? (k < g.in_dim ? W[unit * g.in_dim + k] : (k == g.in_dim ? W[N * g.in_dim + unit] : 0.0f)) : 0.0f;
    }
    put_group(g0 + ot, ln, v);
  }
  // forward [ot][ks] and backward [it][ks] through hidden layer l
#pragma unroll
  for (int l = 1; l < NL; ++l) {
    const int K = Kof(l), N = Nof(l);
    const float *W = src + g.off[l];
    for (int idx = threadIdx.x; idx < (with_backward ? 2 : 1) * 2 * GW * GW * 64; idx += nthreads) {
      const int dir = idx / (2 * GW * GW * 64), r2 = (idx >> 6) % (2 * GW * GW), ln = idx & 63, mm = ln & 31, hh = ln >> 5;
      const int tile = r2 / (2 * GW), ks = r2 % (2 * GW);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int across = 32 * (ks >> 1) + bt::acc_row(ks & 1, hh, e), own = tile * 32 + mm;
        // forward: row = own output unit, column = input `across`; backward (W^T): row = own input, column = output
        const int j = dir == 0 ? own : across, k = dir == 0 ? across : own;
        v[e] = (j < N && k < K) ? W[j * K + k] : 0.0f;
      }
      put_group(g0 + (dir == 0 ? gm_fh(GW, l) : (compact ? cp_bh(GW, NL, l) : gm_bh(GW, NL, l))) + r2, ln, v);
    }
  }
  {  // output layer: forward [ks], backward [it]
    const int K = Kof(NL);
    const float *W = src + g.off[NL];
    for (int idx = threadIdx.x; idx < (with_backward ? 3 : 2) * GW * 64; idx += nthreads) {
      const int q = idx >> 6, ln = idx & 63, mm = ln & 31, hh = ln >> 5;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int j, k;
        if (q < 2 * GW) {
          j = mm;
          k = 32 * (q >> 1) + bt::acc_row(q & 1, hh, e);
        } else {
          j = bt::acc_row(0, hh, e);
          k = (q - 2 * GW) * 32 + mm;
        }
        v[e] = (j < g.out_dim && k < K) ? W[j * K + k] : 0.0f;
      }
      if (compact && q < 2 * GW) {
        if (mm < 2) {
          uint32_t p[3][8];
#pragma unroll
          for (int e = 0; e < 8; ++e) bt::split3(v[e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
          for (int c = 0; c < 3; ++c)
            outc[(q * 3 + c) * 4 + 2 * mm + hh] = make_uint4(bt::pk(p[c][0], p[c][1]), bt::pk(p[c][2], p[c][3]),
                                                             bt::pk(p[c][4], p[c][5]), bt::pk(p[c][6], p[c][7]));
        }
      } else {
        put_group(g0 + (q < 2 * GW ? gm_fo(GW, NL) + q : (compact ? cp_bo(GW, NL) : gm_bo(GW, NL)) + (q - 2 * GW)), ln, v);
      }
    }
  }
#pragma unroll
  for (int l = 1; l <= NL; ++l) {
    const int N = Nof(l), K = Kof(l);
    for (int u = threadIdx.x; u < 32 * GW; u += nthreads) bias_out[(l - 1) * 32 * GW + u] = u < N ? src[g.off[l] + N * K + u] : 0.0f;
  }
}

constexpr int GM_CRITIC = 100;  // mean((V - target)^2); the policy modes are PASS_INIT / PASS_PPO / PASS_EVAL (kernels.hpp)

// one lane's view of a tile: its sample's inputs 4 kb .. 4 kb + 3 (bias input included) and per-sample scalars
struct TileIn {
  float x[4], tgt, adv, l0, l1;
  int act;
  bool valid;
};

// The per-sample loss terms from the module's two (pre-activation) outputs z0, z1 (JVP: and their tangents):
// d0 / d1 = d loss / d (pre-activation of output 0 / 1) on `owner` lanes, the pass's scalar sums on lanes that `counts`
// (the arithmetic of k_policy_bf16's per-sample section, kernels_mfma.hip; critic: mean((V - target)^2), inv_B = 2 / B).
template <int MODE>
__device__ __forceinline__ void gm_sample_terms(const GmArgs &g, float z0, float z1, float tz0, float tz1, const TileIn &op,
                                                bool owner, bool counts, float inv_B, float clip_lo, float clip_hi,
                                                float *__restrict__ lp0, uint32_t sidx, uint32_t B32, float (&sum32)[3],
                                                float &d0, float &d1) {
  constexpr bool JVP = MODE == PASS_JVP;
  if (MODE == GM_CRITIC) {
    const float v = gm_act(g.out_act, z0);
    const float d = v - op.tgt;
    if (owner) {
      if (counts) sum32[0] = __builtin_fmaf(d, d, sum32[0]);
      d0 = d * inv_B * gm_slope(g.out_act, v);  // inv_B = 2 / B here
    }
  } else if (JVP) {
    // Fisher metric of the categorical head on the tangent outputs: dz_a = p_a (ty_a - sum_b p_b ty_b) / B; two
    // actions: dz_0 = -dz_1 = p_0 p_1 (ty_0 - ty_1) / B
    const float y0 = gm_act(g.out_act, z0), y1 = gm_act(g.out_act, z1);
    const float s0 = gm_slope(g.out_act, y0), s1 = gm_slope(g.out_act, y1);
    const SoftPair sp = soft_pair(y0 - y1);
    const float gz = (sp.p[0] * sp.p[1]) * (tz0 * s0 - tz1 * s1) * inv_B;
    if (owner) {
      d0 = gz * s0;
      d1 = -gz * s1;
    }
  } else {
    const float y0 = gm_act(g.out_act, z0), y1 = gm_act(g.out_act, z1);
    const SoftPair sp = soft_pair(y0 - y1);
    const float adv = op.adv;
    const int act = op.act;
    const float lpa = act == 0 ? sp.lp[0] : sp.lp[1];
    float g0 = 0.0f, g1 = 0.0f;  // d loss / d output
    if (MODE == PASS_INIT) {
      if (owner && counts) {
        lp0[sidx] = sp.lp[0];
        lp0[B32 + sidx] = sp.lp[1];
      }
      const float cc = -adv * inv_B;  // ratio = exp(lpa - lpa) = 1
      g0 = cc * ((act == 0 ? 1.0f : 0.0f) - sp.p[0]);
      g1 = cc * ((act == 1 ? 1.0f : 0.0f) - sp.p[1]);
      const float cl0 = sp.lp[0] < -3.402823466e+38f ? -3.402823466e+38f : sp.lp[0];
      const float cl1 = sp.lp[1] < -3.402823466e+38f ? -3.402823466e+38f : sp.lp[1];
      float ent = cl0 * sp.p[0];
      ent += cl1 * sp.p[1];
      if (owner && counts) {
        sum32[0] = sum32[0] + adv;  // ratio * adv
        sum32[1] = sum32[1] - ent;
        sum32[2] = __builtin_fmaf(lpa, adv, sum32[2]);
      }
    } else if (MODE == PASS_PPO) {
      // clipped surrogate (policies/ppo.rs:124-137); see k_policy_pass for the tie rules of minimum()/clamp()
      const float l0a = act == 0 ? op.l0 : op.l1;
      const float ratio = fast_expf(lpa - l0a);
      const float clipped = ratio < clip_lo ? clip_lo : (ratio > clip_hi ? clip_hi : ratio);
      const float u1 = ratio * adv, u2 = clipped * adv;
      const bool inside = ratio >= clip_lo && ratio <= clip_hi;
      const float gr = u1 < u2 ? adv : (u1 > u2 ? (inside ? adv : 0.0f) : (inside ? adv : 0.5f * adv));
      const float cc = -(gr * ratio) * inv_B;
      g0 = cc * ((act == 0 ? 1.0f : 0.0f) - sp.p[0]);
      g1 = cc * ((act == 1 ? 1.0f : 0.0f) - sp.p[1]);
      if (owner && counts) sum32[0] = sum32[0] + (u1 < u2 ? u1 : u2);
    } else {  // PASS_EVAL: surrogate and KL(pi_0 || pi) of candidate parameters
      const float l0a = act == 0 ? op.l0 : op.l1;
      const float ratio = fast_expf(lpa - l0a);
      float rel0 = op.l0 - sp.lp[0], rel1 = op.l1 - sp.lp[1];
      if (rel0 < -3.402823466e+38f) rel0 = -3.402823466e+38f;
      if (rel1 < -3.402823466e+38f) rel1 = -3.402823466e+38f;
      const SoftPair old = soft_pair(op.l0 - op.l1);
      float kl = rel0 * old.p[0];
      kl += rel1 * old.p[1];
      if (owner && counts) {
        sum32[0] = __builtin_fmaf(ratio, adv, sum32[0]);
        sum32[1] = sum32[1] + kl;
      }
    }
    if (owner) {
      d0 = g0 * gm_slope(g.out_act, y0);
      d1 = g1 * gm_slope(g.out_act, y1);
    }
  }
}

template <int MODE, int NL, int GW>
__global__ void __launch_bounds__(GWAVES * 64)
    k_gen_mfma(TrajDev tr, GmArgs g, float *__restrict__ lp0, double *__restrict__ slabA, double *__restrict__ slabB,
               float inv_B, const int32_t *__restrict__ skip, float clip_lo, float clip_hi) {
  extern __shared__ uint4 gm_lds[];
  uint4(*img)[64] = reinterpret_cast<uint4(*)[64]>(gm_lds);
  constexpr bool JVP = MODE == PASS_JVP;
  constexpr int TG = gm_groups(GW, NL);  // first fragment group of the tangent parameters (JVP)
  float *bias = reinterpret_cast<float *>(gm_lds + (size_t)gm_all_groups(GW, NL, JVP) * 3 * 64);  // [NL][32 GW]: layers 1 .. NL
  float *tbias = bias + NL * 32 * GW;                                                               // (JVP) of the tangent
  constexpr bool BWD = MODE != PASS_EVAL;
  if (skip != nullptr && *skip != 0) return;
#ifdef GM_FIXED_RELU
  g.act = RL_ACT_RELU;
  g.out_act = RL_ACT_IDENTITY;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, kb = lane >> 5;
  auto Kof = [&](int l) { return l == 0 ? g.in_dim : g.width[l - 1]; };
  auto Nof = [&](int l) { return l == NL ? g.out_dim : g.width[l]; };

  gm_build_image<NL, GW>(g, g.params, img, 0, true, bias, GWAVES * 64);
  if (JVP) gm_build_image<NL, GW>(g, g.tangent, img, TG, false, tbias, GWAVES * 64);
  __syncthreads();

  Frag idb[2];
  bt::ident_frags(lane, idb);
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // weight-gradient accumulators: layer 0 [ot] (columns: inputs, then the bias), hidden [l - 1][ot][it], output [it]
  // (rows: outputs), biases of layers >= 1 (column = (l - 1) GW + ot)
  f32x16 dW0[GW], dWh[NL > 1 ? NL - 1 : 1][GW][GW], dWo[GW], dbt = zero16;
#pragma unroll
  for (int a = 0; a < GW; ++a) {
    dW0[a] = zero16;
    dWo[a] = zero16;
#pragma unroll
    for (int l = 0; l < (NL > 1 ? NL - 1 : 1); ++l)
#pragma unroll
      for (int b = 0; b < GW; ++b) dWh[l][a][b] = zero16;
  }
  double sum64[3] = {0.0, 0.0, 0.0};
  float sum32[3] = {0.0f, 0.0f, 0.0f};

  const size_t B = (size_t)tr.T * tr.n;
  const uint32_t B32 = (uint32_t)B, plane32 = (uint32_t)((size_t)(tr.T + 1) * tr.n);
  const size_t n_tiles = (B + 31) / 32;
  const size_t wave_id = (size_t)blockIdx.x * GWAVES + wave, n_waves = (size_t)gridDim.x * GWAVES;
  if (wave_id >= n_tiles) return;  // (no barrier below; the launcher counts slab rows for the waves that have tiles)
  double *__restrict__ row = slabA + wave_id * g.P;
  bool first_flush = true;

  // slab entry += tile entry, for the valid (output j, input k) of a weight tile / the bias columns
  auto put = [&](uint32_t at, float v) {
    if (first_flush) row[at] = (double)v;
    else row[at] = row[at] + (double)v;
  };
  auto flush_w = [&](f32x16 &t, int l, int ot, int it) {
    // (the lane's column through a register the optimiser cannot see through: with a visible one, the ~100 slab addresses
    // and bounds of a flush are loop-invariant, get computed ahead of the tile loop and live in scratch until the flush —
    // ~1 KB per lane written and read back by every launch: 270 MB at 16,384 x 128 samples)
    int mo = m, kbo = kb;
    asm volatile("" : "+v"(mo), "+v"(kbo));
    const int K = Kof(l), N = Nof(l), k = it * 32 + mo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = ot * 32 + urow(r, kbo);
      if (j < N) {
        if (k < K) put(g.off[l] + (uint32_t)(j * K + k), t[r]);
        else if (l == 0 && k == K) put(g.off[0] + (uint32_t)(N * K + j), t[r]);
      }
      t[r] = 0.0f;
    }
    // (the slab updates of one tile at a time: hoisting every load of the flush above the first store needs ~400 more
    // registers than the kernel has)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto flush_all = [&]() {
#pragma unroll
    for (int ot = 0; ot < GW; ++ot) flush_w(dW0[ot], 0, ot, 0);
#pragma unroll
    for (int l = 1; l < NL; ++l)
#pragma unroll
      for (int ot = 0; ot < GW; ++ot)
#pragma unroll
        for (int it = 0; it < GW; ++it) flush_w(dWh[l - 1][ot][it], l, ot, it);
#pragma unroll
    for (int it = 0; it < GW; ++it) flush_w(dWo[it], NL, 0, it);
    // biases of layers 1 .. NL: column c = (l - 1) GW + ot of the bias tile
#pragma unroll
    for (int l = 1; l <= NL; ++l) {
      int mo = m, kbo = kb;
      asm volatile("" : "+v"(mo), "+v"(kbo));
      const int N = Nof(l), K = Kof(l), ot = mo - (l - 1) * GW;
      if (ot >= 0 && ot < (l < NL ? GW : 1)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = ot * 32 + urow(r, kbo);
          if (j < N) put(g.off[l] + (uint32_t)(N * K + j), dbt[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dbt[r] = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      sum64[q] += (double)sum32[q];
      sum32[q] = 0.0f;
    }
    first_flush = false;
  };

  auto load_tile = [&](size_t t) {  // (branch-free: padding lanes read sample B - 1 and are zeroed)
    TileIn o;
    const uint32_t sidx = (uint32_t)t * 32u + (uint32_t)m;
    o.valid = t < n_tiles && sidx < B32;
    const uint32_t sc = o.valid ? sidx : B32 - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * kb + e;
      float v = 0.0f;
      if (k < g.in_dim) v = tr.obs[(uint32_t)k * plane32 + sc];
      else if (k == g.in_dim) v = 1.0f;
      o.x[e] = o.valid ? v : 0.0f;
    }
    o.tgt = o.adv = o.l0 = o.l1 = 0.0f;
    o.act = 0;
    if (MODE == GM_CRITIC) {
      const float tg = tr.tgt[sc];
      o.tgt = o.valid ? tg : 0.0f;
    } else if (!JVP) {
      const float adv = tr.adv[sc];
      const int act = (int)tr.action[sc];
      o.adv = o.valid ? adv : 0.0f;
      o.act = o.valid ? act : 0;
      if (MODE != PASS_INIT) {
        const float l0 = lp0[sc], l1 = lp0[B32 + sc];
        o.l0 = o.valid ? l0 : 0.0f;
        o.l1 = o.valid ? l1 : 0.0f;
      }
    }
    return o;
  };

  int since_flush = 0;
  TileIn op = load_tile(wave_id);
  for (size_t t = wave_id; t < n_tiles; t += n_waves) {
    const TileIn next = load_tile(t + n_waves);
    // ---- forward
    Frag xb0[1][3];
    {
      uint32_t p[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bt::split3t(op.x[e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xb0[0][c].u[0] = bt::pkh(p[c][0], p[c][1]);
        xb0[0][c].u[1] = bt::pkh(p[c][2], p[c][3]);
        xb0[0][c].u[2] = xb0[0][c].u[3] = 0u;
      }
    }
    // (JVP: the tangent of every layer rides along, ta_l = act'(.) (W_l ta_{l-1} + V_l a_{l-1} + vb_l), ta_{-1} = 0 —
    // forward-mode differentiation along the tangent parameters V, conjugate_gradient.rs:262-339)
    f32x16 a[NL][GW], ta[GW];
#pragma unroll
    for (int ot = 0; ot < GW; ++ot) {
      a[0][ot] = prod6_lds(zero16, img, ot, lane, xb0[0]);
      gm_act_tile(g.act, a[0][ot]);
      if (JVP) {
        ta[ot] = prod6_lds(zero16, img, TG + ot, lane, xb0[0]);
        gm_slope_tile(g.act, ta[ot], a[0][ot]);
      }
    }
    auto bias_rows = [&](const float *table, int l, int ot) {  // accumulator initialised with the bias of its row's unit
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = table[(l - 1) * 32 * GW + ot * 32 + urow(r, kb)];
      return c;
    };
#pragma unroll
    for (int l = 1; l < NL; ++l) {
      Frag ab[GW][2][3], tb[JVP ? GW : 1][2][3];
#pragma unroll
      for (int it = 0; it < GW; ++it)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          pieces_trunc(a[l - 1][it], q, ab[it][q]);
          if (JVP) pieces_trunc(ta[it], q, tb[it][q]);
        }
#pragma unroll
      for (int ot = 0; ot < GW; ++ot) {
        f32x16 c = bias_rows(bias, l, ot);
#pragma unroll
        for (int ks = 0; ks < 2 * GW; ++ks) c = prod6_lds(c, img, gm_fh(GW, l) + ot * 2 * GW + ks, lane, ab[ks >> 1][ks & 1]);
        gm_act_tile(g.act, c);
        a[l][ot] = c;
        if (JVP) {
          f32x16 tc = bias_rows(tbias, l, ot);
#pragma unroll
          for (int ks = 0; ks < 2 * GW; ++ks) {
            tc = prod6_lds(tc, img, gm_fh(GW, l) + ot * 2 * GW + ks, lane, tb[JVP ? ks >> 1 : 0][ks & 1]);
            tc = prod6_lds(tc, img, TG + gm_fh(GW, l) + ot * 2 * GW + ks, lane, ab[ks >> 1][ks & 1]);
          }
          gm_slope_tile(g.act, tc, c);
          ta[ot] = tc;  // (the pieces of the previous layer's tangent are in tb already)
        }
      }
    }
    f32x16 zt = bias_rows(bias, NL, 0), tzt = zero16;
    {
      Frag ab[GW][2][3];
#pragma unroll
      for (int it = 0; it < GW; ++it)
#pragma unroll
        for (int q = 0; q < 2; ++q) pieces_trunc(a[NL - 1][it], q, ab[it][q]);
#pragma unroll
      for (int ks = 0; ks < 2 * GW; ++ks) zt = prod6_lds(zt, img, gm_fo(GW, NL) + ks, lane, ab[ks >> 1][ks & 1]);
      if (JVP) {
        tzt = bias_rows(tbias, NL, 0);
#pragma unroll
        for (int ks = 0; ks < 2 * GW; ++ks) {
          Frag tb[3];
          pieces_trunc(ta[ks >> 1], ks & 1, tb);
          tzt = prod6_lds(tzt, img, gm_fo(GW, NL) + ks, lane, tb);
          tzt = prod6_lds(tzt, img, TG + gm_fo(GW, NL) + ks, lane, ab[ks >> 1][ks & 1]);
        }
      }
    }
    // ---- per-sample terms on the lanes of half 0 (rows 0 and 1 of the output tile are the module's outputs)
    float d0 = 0.0f, d1 = 0.0f;  // d loss / d (pre-activation of output 0 / 1)
    gm_sample_terms<MODE>(g, zt[0], zt[1], tzt[0], tzt[1], op, kb == 0 && op.valid, true, inv_B, clip_lo, clip_hi, lp0,
                          (uint32_t)t * 32u + (uint32_t)m, B32, sum32, d0, d1);
    if (BWD) {
      // ---- output layer: delta pieces (units 0 and 1 = elements 0 and 1 of half 0's k-step 0)
      Frag dob[1][3];
      {
        uint32_t p0[3], p1[3];
        bt::split3(d0, p0[0], p0[1], p0[2]);
        bt::split3(d1, p1[0], p1[1], p1[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dob[0][c].u[0] = kb == 0 ? bt::pk(p0[c], p1[c]) : 0u;
          dob[0][c].u[1] = dob[0][c].u[2] = dob[0][c].u[3] = 0u;
        }
      }
      Frag dT[GW][2][3], aT[GW][2][3];  // transposed pieces: deltas of the layer in hand, activations below it
      transpose_pieces<1>(dob, idb, dT[0]);
      dbt = bias_tile(dbt, dT[0], (NL - 1) * GW, m);
      f32x16 dl[GW];  // deltas of the hidden layer in hand
#pragma unroll
      for (int it = 0; it < GW; ++it) {
        Frag ab[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) pieces_trunc(a[NL - 1][it], q, ab[q]);
        transpose_pieces<2>(ab, idb, aT[it]);
        dWo[it] = wgrad_tile(dWo[it], dT[0], aT[it]);
        f32x16 c = prod6_lds(zero16, img, gm_bo(GW, NL) + it, lane, dob[0]);
        gm_slope_tile(g.act, c, a[NL - 1][it]);
        dl[it] = c;
      }
      // ---- hidden layers NL - 1 .. 1: weights between layer l - 1 and l
#pragma unroll
      for (int l = NL - 1; l >= 1; --l) {
        Frag db[GW][2][3];
#pragma unroll
        for (int ot = 0; ot < GW; ++ot) {
#pragma unroll
          for (int q = 0; q < 2; ++q) pieces_round(dl[ot], q, db[ot][q]);
          transpose_pieces<2>(db[ot], idb, dT[ot]);
          dbt = bias_tile(dbt, dT[ot], (l - 1) * GW + ot, m);
        }
        f32x16 dn[GW];
#pragma unroll
        for (int it = 0; it < GW; ++it) {
          Frag ab[2][3];
#pragma unroll
          for (int q = 0; q < 2; ++q) pieces_trunc(a[l - 1][it], q, ab[q]);
          transpose_pieces<2>(ab, idb, aT[it]);
#pragma unroll
          for (int ot = 0; ot < GW; ++ot) dWh[l - 1][ot][it] = wgrad_tile(dWh[l - 1][ot][it], dT[ot], aT[it]);
          f32x16 c = zero16;
#pragma unroll
          for (int ks = 0; ks < 2 * GW; ++ks)
            c = prod6_lds(c, img, gm_bh(GW, NL, l) + it * 2 * GW + ks, lane, db[ks >> 1][ks & 1]);
          gm_slope_tile(g.act, c, a[l - 1][it]);
          dn[it] = c;
        }
#pragma unroll
        for (int it = 0; it < GW; ++it) dl[it] = dn[it];
      }
      // ---- layer 0: inputs (and the bias input) against the deltas of hidden layer 0
      transpose_pieces<1>(xb0, idb, aT[0]);
#pragma unroll
      for (int ot = 0; ot < GW; ++ot) {
        Frag db[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) pieces_round(dl[ot], q, db[q]);
        transpose_pieces<2>(db, idb, dT[ot]);
        dW0[ot] = wgrad_tile(dW0[ot], dT[ot], aT[0]);
      }
    }
    // one flush site (the accumulators stay in registers only if nothing out of line touches them): every GM_FLUSH
    // tiles and after the wave's last tile
    if (++since_flush == GM_FLUSH || t + n_waves >= n_tiles) {
      since_flush = 0;
      if (BWD) {
        flush_all();
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          sum64[q] += (double)sum32[q];
          sum32[q] = 0.0f;
        }
      }
    }
    op = next;
  }
  // the wave's scalar sums: over its 32 owner lanes
  auto xlane = [](double v, int mask) {
    uint64_t bits = rl_f64_bits(v);
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, mask, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), mask, 64);
    return rl_f64_from_bits(((uint64_t)hi << 32) | lo);
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double s = kb == 0 ? sum64[q] : 0.0;
#pragma unroll
    for (int sft = 16; sft > 0; sft >>= 1) s = s + xlane(s, sft);
    sum64[q] = s;
  }
  if (lane == 0) {
    double *sb = slabB + wave_id * 4;
    sb[0] = sum64[0];
    sb[1] = sum64[1];
    sb[2] = sum64[2];
    sb[3] = 0.0;
  }
}

// ================================================================================================
// The same passes with a tile shared by TWO waves (hidden layers of at most 64 units = two 32-unit tiles): wave h of a
// pair owns unit tile h of every hidden layer — its activations, its deltas, the weight-gradient rows of its units — and
// the two meet through LDS where a product needs the other half: the operand pieces of a_{l-1} (forward), of D_l
// (backward chain) and of A_{l-1}^T (weight gradients), 6 KB each way, plus the two partial sums of the output layer.
// A wave then carries 80 accumulator registers instead of 144 and ~230 in all: TWO waves per SIMD, each hiding the
// other's dependency chains (the one-wave kernel above waits 42 % of its cycles).  Every pair of a workgroup walks the
// same number of tiles (a pair without a tile runs on zeros): the exchanges are workgroup barriers.  (Hand-overs between
// the two waves of a pair only — arrival counters in LDS, the partner polling — were measured and are SLOWER: 13.5
// against 12.5 ms for 20 critic steps of [64, 64] at 16,384 x 128; a polling wave takes issue slots from the SIMD's other
// wave.  Nor does running half of the pairs half a tile behind the others behind the SAME barriers help — every barrier
// interval then lasts as long as the longer of two different program segments: 16.3 ms.)
// One exchange region per wave: write, barrier, the partner reads; the next barrier in program order (there is always
// one before the region's next write) guards the reuse.
// ================================================================================================
constexpr int GP_PAIRS = 4;

template <int MODE, int NL, int PAIRS>
__global__ void __launch_bounds__(PAIRS * 128)
    k_gen_pair(TrajDev tr, GmArgs g, float *__restrict__ lp0, double *__restrict__ slabA, double *__restrict__ slabB,
               float inv_B, const int32_t *__restrict__ skip, float clip_lo, float clip_hi) {
  constexpr int GW = 2;
  constexpr bool JVP = MODE == PASS_JVP, BWD = MODE != PASS_EVAL;
  constexpr int TG = cp_groups(GW, NL);                                  // first fragment group of the tangent (JVP)
  constexpr int NG = TG + (JVP ? cp_tan_groups(GW, NL) : 0);             // groups of the image
  constexpr int ZV = JVP ? 4 : 2;                                         // partial outputs a wave hands over per sample
  extern __shared__ uint4 gm_lds[];
  uint4(*img)[64] = reinterpret_cast<uint4(*)[64]>(gm_lds);
  uint4 *outc = gm_lds + (size_t)NG * 3 * 64;                             // compact output-layer forward fragments ...
  uint4 *toutc = outc + 2 * GW * 3 * 4;                                   // ... and the tangent's
  float *bias = reinterpret_cast<float *>(toutc + (JVP ? 2 * GW * 3 * 4 : 0));  // [NL][64]: layers 1 .. NL
  float *tbias = bias + NL * 64;                                                 // (JVP) of the tangent
  uint4(*xch)[64] = reinterpret_cast<uint4(*)[64]>(bias + NL * 64 * (JVP ? 2 : 1));  // [PAIRS][2][6][64]
  float *zbuf = reinterpret_cast<float *>(xch + PAIRS * 2 * 6);                         // [PAIRS][2][ZV][32]
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, kb = lane >> 5, pair = wave >> 1, h = wave & 1;
  auto Kof = [&](int l) { return l == 0 ? g.in_dim : g.width[l - 1]; };
  auto Nof = [&](int l) { return l == NL ? g.out_dim : g.width[l]; };
  gm_build_image<NL, GW>(g, g.params, img, 0, true, bias, PAIRS * 128, outc);
  if (JVP) gm_build_image<NL, GW>(g, g.tangent, img, TG, false, tbias, PAIRS * 128, toutc);
  __syncthreads();

  Frag idb[2];
  bt::ident_frags(lane, idb);
  const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // this wave's rows of the weight gradients: layer 0 (unit tile h x inputs + bias), hidden layer l (unit tile h x input
  // tile it), the output layer's columns of input tile h, the bias columns of its tiles (wave 0: and the output layer's)
  // (hidden layers: dWown = against this wave's own input tile h, dWoth = against the partner's tile 1 - h; two named
  // arrays: an index that depends on h would put the accumulators in memory)
  f32x16 dW0 = zero16, dWown[NL > 1 ? NL - 1 : 1], dWoth[NL > 1 ? NL - 1 : 1], dWo = zero16, dbt = zero16;
#pragma unroll
  for (int l = 0; l < (NL > 1 ? NL - 1 : 1); ++l) dWown[l] = dWoth[l] = zero16;
  double sum64[3] = {0.0, 0.0, 0.0};
  float sum32[3] = {0.0f, 0.0f, 0.0f};

  const size_t B = (size_t)tr.T * tr.n;
  const uint32_t B32 = (uint32_t)B, plane32 = (uint32_t)((size_t)(tr.T + 1) * tr.n);
  const size_t n_tiles = (B + 31) / 32;
  const size_t pair_id = (size_t)blockIdx.x * PAIRS + pair, n_pairs = (size_t)gridDim.x * PAIRS;
  const size_t iters = (n_tiles + n_pairs - 1) / n_pairs;
  double *__restrict__ row = slabA + pair_id * g.P;  // the pair's row: the two waves fill disjoint entries
  bool first_flush = true;
  auto put = [&](uint32_t at, float v) {
    if (first_flush) row[at] = (double)v;
    else row[at] = row[at] + (double)v;
  };
  auto flush_w = [&](f32x16 &t, int l, int ot, int it) {
    // (the lane's column through a register the optimiser cannot see through: with a visible one, the ~100 slab addresses
    // and bounds of a flush are loop-invariant, get computed ahead of the tile loop and live in scratch until the flush —
    // ~1 KB per lane written and read back by every launch: 270 MB at 16,384 x 128 samples)
    int mo = m, kbo = kb;
    asm volatile("" : "+v"(mo), "+v"(kbo));
    const int K = Kof(l), N = Nof(l), k = it * 32 + mo;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = ot * 32 + urow(r, kbo);
      if (j < N) {
        if (k < K) put(g.off[l] + (uint32_t)(j * K + k), t[r]);
        else if (l == 0 && k == K) put(g.off[0] + (uint32_t)(N * K + j), t[r]);
      }
      t[r] = 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto flush_all = [&]() {
    flush_w(dW0, 0, h, 0);
#pragma unroll
    for (int l = 1; l < NL; ++l) {
      flush_w(dWown[l - 1], l, h, h);
      flush_w(dWoth[l - 1], l, h, 1 - h);
    }
    flush_w(dWo, NL, 0, h);
#pragma unroll
    for (int l = 1; l <= NL; ++l) {  // bias column (l - 1) GW + ot: this wave's tile of a hidden layer; wave 0: the outputs
      int mo = m, kbo = kb;
      asm volatile("" : "+v"(mo), "+v"(kbo));
      const int N = Nof(l), K = Kof(l), ot = mo - (l - 1) * GW;
      if (l < NL ? ot == h : (ot == 0 && h == 0)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = ot * 32 + urow(r, kbo);
          if (j < N) put(g.off[l] + (uint32_t)(N * K + j), dbt[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dbt[r] = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      sum64[q] += (double)sum32[q];
      sum32[q] = 0.0f;
    }
    first_flush = false;
  };
  auto load_tile = [&](size_t t) {  // (branch-free: padding lanes read sample B - 1 and are zeroed)
    TileIn o;
    const uint32_t sidx = (uint32_t)t * 32u + (uint32_t)m;
    o.valid = t < n_tiles && sidx < B32;
    const uint32_t sc = o.valid ? sidx : B32 - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * kb + e;
      float v = 0.0f;
      if (k < g.in_dim) v = tr.obs[(uint32_t)k * plane32 + sc];
      else if (k == g.in_dim) v = 1.0f;
      o.x[e] = o.valid ? v : 0.0f;
    }
    o.tgt = o.adv = o.l0 = o.l1 = 0.0f;
    o.act = 0;
    if (MODE == GM_CRITIC) {
      const float tg = tr.tgt[sc];
      o.tgt = o.valid ? tg : 0.0f;
    } else if (!JVP) {
      const float adv = tr.adv[sc];
      const int act = (int)tr.action[sc];
      o.adv = o.valid ? adv : 0.0f;
      o.act = o.valid ? act : 0;
      if (MODE != PASS_INIT) {
        const float l0 = lp0[sc], l1 = lp0[B32 + sc];
        o.l0 = o.valid ? l0 : 0.0f;
        o.l1 = o.valid ? l1 : 0.0f;
      }
    }
    return o;
  };
  auto bias_rows = [&](const float *table, int l) {  // accumulator of this wave's tile initialised with its units' biases
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = table[(l - 1) * 64 + h * 32 + urow(r, kb)];
    return c;
  };
  // acc += (output-layer weights of k-step ks of this wave's tile) x pieces, from the compact fragments
  auto out_prod = [&](f32x16 acc, const uint4 *oc, int ks, const Frag(&x)[3]) {
    Frag w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) w[c].x = m < 2 ? oc[(ks * 3 + c) * 4 + 2 * m + kb] : make_uint4(0u, 0u, 0u, 0u);
    return prod6(acc, w, x);
  };
  // the exchange: this wave's two k-steps x three pieces to the partner, the partner's to this wave
  uint4(*mine)[64] = xch + (pair * 2 + h) * 6, (*theirs)[64] = xch + (pair * 2 + (1 - h)) * 6;
  auto xput = [&](const Frag(&P)[2][3]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < 3; ++c) mine[q * 3 + c][lane] = P[q][c].x;
  };
  auto xget = [&](Frag(&P)[2][3]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < 3; ++c) P[q][c].x = theirs[q * 3 + c][lane];
  };

  int since_flush = 0;
  TileIn op = load_tile(pair_id);
  for (size_t it_ = 0; it_ < iters; ++it_) {
    const size_t t = pair_id + it_ * n_pairs;
    const TileIn next = load_tile(t + n_pairs);
    // ---- forward: this wave's unit tile of every layer
    Frag xb0[1][3];
    {
      uint32_t p[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bt::split3t(op.x[e], p[0][e], p[1][e], p[2][e]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xb0[0][c].u[0] = bt::pkh(p[c][0], p[c][1]);
        xb0[0][c].u[1] = bt::pkh(p[c][2], p[c][3]);
        xb0[0][c].u[2] = xb0[0][c].u[3] = 0u;
      }
    }
    // (JVP: the tangent rides along, ta_l = act'(.) (W_l ta_{l-1} + V_l a_{l-1} + vb_l), as in the one-wave kernel)
    f32x16 a[NL], ta = zero16;
    a[0] = prod6_lds(zero16, img, h, lane, xb0[0]);
    gm_act_tile(g.act, a[0]);
    if (JVP) {
      ta = prod6_lds(zero16, img, TG + h, lane, xb0[0]);
      gm_slope_tile(g.act, ta, a[0]);
    }
#pragma unroll
    for (int l = 1; l < NL; ++l) {
      // (own products first, then the partner's: one set of operand pieces alive at a time)
      Frag pc[2][3];
#pragma unroll
      for (int q = 0; q < 2; ++q) pieces_trunc(a[l - 1], q, pc[q]);
      if (l > 1) __syncthreads();  // (the partner has read the previous layer's pieces)
      xput(pc);
      f32x16 c = bias_rows(bias, l), tc = zero16;
      if (JVP) tc = bias_rows(tbias, l);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        c = prod6_lds(c, img, gm_fh(GW, l) + h * 2 * GW + h * 2 + q, lane, pc[q]);
        if (JVP) tc = prod6_lds(tc, img, TG + gm_fh(GW, l) + h * 2 * GW + h * 2 + q, lane, pc[q]);
      }
      __syncthreads();
      xget(pc);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        c = prod6_lds(c, img, gm_fh(GW, l) + h * 2 * GW + (1 - h) * 2 + q, lane, pc[q]);
        if (JVP) tc = prod6_lds(tc, img, TG + gm_fh(GW, l) + h * 2 * GW + (1 - h) * 2 + q, lane, pc[q]);
      }
      if (JVP) {  // the tangent of the layer below, the same way
#pragma unroll
        for (int q = 0; q < 2; ++q) pieces_trunc(ta, q, pc[q]);
        __syncthreads();  // (the partner has read the activation pieces)
        xput(pc);
#pragma unroll
        for (int q = 0; q < 2; ++q) tc = prod6_lds(tc, img, gm_fh(GW, l) + h * 2 * GW + h * 2 + q, lane, pc[q]);
        __syncthreads();
        xget(pc);
#pragma unroll
        for (int q = 0; q < 2; ++q) tc = prod6_lds(tc, img, gm_fh(GW, l) + h * 2 * GW + (1 - h) * 2 + q, lane, pc[q]);
      }
      gm_act_tile(g.act, c);
      a[l] = c;
      if (JVP) {
        gm_slope_tile(g.act, tc, c);
        ta = tc;
      }
    }
    // output layer: each wave sums over its own tile's units, the two partial sums meet in LDS (the bias with half 0's)
    f32x16 zp = zero16, tzp = zero16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      Frag ab[3];
      pieces_trunc(a[NL - 1], q, ab);
      zp = out_prod(zp, outc, h * 2 + q, ab);
      if (JVP) {
        tzp = out_prod(tzp, toutc, h * 2 + q, ab);
        Frag tb[3];
        pieces_trunc(ta, q, tb);
        tzp = out_prod(tzp, outc, h * 2 + q, tb);
      }
    }
    float *zmine = zbuf + ((pair * 2 + h) * ZV) * 32, *ztheirs = zbuf + ((pair * 2 + (1 - h)) * ZV) * 32;
    if (kb == 0) {
      zmine[m] = zp[0];
      zmine[32 + m] = zp[1];
      if (JVP) {
        zmine[64 + m] = tzp[0];
        zmine[96 + m] = tzp[1];
      }
    }
    __syncthreads();
    float z0, z1, tz0 = 0.0f, tz1 = 0.0f;
    {
      // (half 0's part first in both waves: the two compute identical sums)
      auto meet = [&](float minev, float theirv, float b) { return ((h == 0 ? minev : theirv) + (h == 0 ? theirv : minev)) + b; };
      z0 = meet(zp[0], ztheirs[m], bias[(NL - 1) * 64 + 0]);
      z1 = meet(zp[1], ztheirs[32 + m], bias[(NL - 1) * 64 + 1]);
      if (JVP) {
        tz0 = meet(tzp[0], ztheirs[64 + m], tbias[(NL - 1) * 64 + 0]);
        tz1 = meet(tzp[1], ztheirs[96 + m], tbias[(NL - 1) * 64 + 1]);
      }
    }
    // ---- per-sample terms: both waves compute them (identically), wave 0 counts them
    float d0 = 0.0f, d1 = 0.0f;
    gm_sample_terms<MODE>(g, z0, z1, tz0, tz1, op, kb == 0 && op.valid, h == 0, inv_B, clip_lo, clip_hi, lp0,
                          (uint32_t)t * 32u + (uint32_t)m, B32, sum32, d0, d1);
    if (BWD) {
      Frag dob[1][3];
      {
        uint32_t p0[3], p1[3];
        bt::split3(d0, p0[0], p0[1], p0[2]);
        bt::split3(d1, p1[0], p1[1], p1[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dob[0][c].u[0] = kb == 0 ? bt::pk(p0[c], p1[c]) : 0u;
          dob[0][c].u[1] = dob[0][c].u[2] = dob[0][c].u[3] = 0u;
        }
      }
      Frag dT[2][3], aT[2][3];  // transposed pieces: the deltas in hand, the activations below them
      transpose_pieces<1>(dob, idb, dT);
      if (h == 0) dbt = bias_tile(dbt, dT, (NL - 1) * GW, m);
      f32x16 dl;  // this wave's tile of the deltas of the hidden layer in hand
      {
        Frag ab[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) pieces_trunc(a[NL - 1], q, ab[q]);
        transpose_pieces<2>(ab, idb, aT);
        dWo = wgrad_tile(dWo, dT, aT);
        dl = prod6_lds(zero16, img, cp_bo(GW, NL) + h, lane, dob[0]);
        gm_slope_tile(g.act, dl, a[NL - 1]);
      }
#pragma unroll
      for (int l = NL - 1; l >= 1; --l) {
        // deltas of layer l - 1, this wave's tile: own k-steps, then the partner's
        f32x16 c = zero16;
        {
          Frag pc[2][3];
#pragma unroll
          for (int q = 0; q < 2; ++q) pieces_round(dl, q, pc[q]);
          if (l < NL - 1) __syncthreads();  // (the partner has read the transposed activations of the layer above)
          xput(pc);  // (first use after the z exchange's barrier)
          transpose_pieces<2>(pc, idb, dT);
          dbt = bias_tile(dbt, dT, (l - 1) * GW + h, m);
#pragma unroll
          for (int q = 0; q < 2; ++q) c = prod6_lds(c, img, cp_bh(GW, NL, l) + h * 2 * GW + h * 2 + q, lane, pc[q]);
          __syncthreads();
          xget(pc);
#pragma unroll
          for (int q = 0; q < 2; ++q) c = prod6_lds(c, img, cp_bh(GW, NL, l) + h * 2 * GW + (1 - h) * 2 + q, lane, pc[q]);
        }
        gm_slope_tile(g.act, c, a[l - 1]);
        // weight gradients of layer l: this wave's output units against its own input tile, then the partner's
        {
          Frag ab[2][3];
#pragma unroll
          for (int q = 0; q < 2; ++q) pieces_trunc(a[l - 1], q, ab[q]);
          transpose_pieces<2>(ab, idb, aT);
        }
        __syncthreads();  // (the partner has read the delta pieces)
        xput(aT);
        dWown[l - 1] = wgrad_tile(dWown[l - 1], dT, aT);
        __syncthreads();
        xget(aT);
        dWoth[l - 1] = wgrad_tile(dWoth[l - 1], dT, aT);
        dl = c;
      }
      // ---- layer 0
      {
        Frag db[2][3], xT[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) pieces_round(dl, q, db[q]);
        transpose_pieces<2>(db, idb, dT);
        transpose_pieces<1>(xb0, idb, xT);
        dW0 = wgrad_tile(dW0, dT, xT);
      }
    }
    __syncthreads();  // the exchange regions are free for the next tile
    if (++since_flush == GM_FLUSH || it_ + 1 == iters) {
      since_flush = 0;
      if (BWD) {
        flush_all();
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          sum64[q] += (double)sum32[q];
          sum32[q] = 0.0f;
        }
      }
    }
    op = next;
  }
  if (h != 0) return;  // the scalar sums live on wave 0 of the pair
  auto xlane = [](double v, int mask) {
    uint64_t bits = rl_f64_bits(v);
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, mask, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), mask, 64);
    return rl_f64_from_bits(((uint64_t)hi << 32) | lo);
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double s = kb == 0 ? sum64[q] : 0.0;
#pragma unroll
    for (int sft = 16; sft > 0; sft >>= 1) s = s + xlane(s, sft);
    sum64[q] = s;
  }
  if (lane == 0) {
    double *sb = slabB + pair_id * 4;
    sb[0] = sum64[0];
    sb[1] = sum64[1];
    sb[2] = sum64[2];
    sb[3] = 0.0;
  }
}

constexpr size_t gp_lds_bytes(int NL, int pairs, bool jvp) {
  return (size_t)(cp_groups(2, NL) + (jvp ? cp_tan_groups(2, NL) : 0)) * 3 * 64 * 16 + (size_t)(jvp ? 2 : 1) * 4 * 3 * 4 * 16 +
         (size_t)NL * 64 * 4 * (jvp ? 2 : 1) + (size_t)pairs * 2 * 6 * 64 * 16 + (size_t)pairs * 2 * (jvp ? 4 : 2) * 32 * 4;
}

template <int MODE, int NL>
void gp_launch(rl_traj *t, const GmArgs &g, uint32_t nwg, float inv_B, const int32_t *d_skip, float clip_lo, float clip_hi) {
  const size_t lds = gp_lds_bytes(NL, GP_PAIRS, MODE == PASS_JVP);
  {
    static std::mutex mu;
    static std::set<int> raised;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.insert(t->eng->device).second)
      RL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gen_pair<MODE, NL, GP_PAIRS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_gen_pair<MODE, NL, GP_PAIRS>), dim3(nwg), dim3(GP_PAIRS * 128), lds, t->eng->stream, t->d, g, t->lp0,
                     t->slabA, t->slabB, inv_B, d_skip, clip_lo, clip_hi);
  RL_HIP_CHECK(hipGetLastError());
}
template <int MODE>
void gp_launch_nl(rl_traj *t, const GmArgs &g, int NL, uint32_t nwg, float inv_B, const int32_t *d_skip, float clip_lo,
                  float clip_hi) {
  if (NL == 1) gp_launch<MODE, 1>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
  else if (NL == 2) gp_launch<MODE, 2>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
  else gp_launch<MODE, 3>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
}

template <int MODE, int NL, int GW>
void gm_launch(rl_traj *t, const GmArgs &g, uint32_t nwg, float inv_B, const int32_t *d_skip, float clip_lo,
               float clip_hi) {
  const size_t lds = gm_lds_bytes(GW, NL, MODE == PASS_JVP);
  {
    static std::mutex mu;
    static std::set<int> raised;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.insert(t->eng->device).second)
      RL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gen_mfma<MODE, NL, GW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_gen_mfma<MODE, NL, GW>), dim3(nwg), dim3(GWAVES * 64), lds, t->eng->stream, t->d, g, t->lp0,
                     t->slabA, t->slabB, inv_B, d_skip, clip_lo, clip_hi);
  RL_HIP_CHECK(hipGetLastError());
}

// gw = 4: one hidden layer of up to 128 units; gw = 2: up to three of up to 64
template <int MODE>
void gm_launch_nl(rl_traj *t, const GmArgs &g, int NL, int gw, uint32_t nwg, float inv_B, const int32_t *d_skip,
                  float clip_lo, float clip_hi) {
  if (gw == 4) gm_launch<MODE, 1, 4>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
  else if (NL == 1) gm_launch<MODE, 1, 2>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
  else if (NL == 2) gm_launch<MODE, 2, 2>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
  else gm_launch<MODE, 3, 2>(t, g, nwg, inv_B, d_skip, clip_lo, clip_hi);
}

// tiles per hidden layer of the instantiation that takes the module, 0 if none does
int gm_width_tiles(const rl_mlp *m) {
  uint32_t wmax = 0;
  for (uint32_t l = 0; l < m->n_hidden; ++l) wmax = m->widths[l] > wmax ? m->widths[l] : wmax;
  if (wmax <= 64) return 2;
  if (wmax <= 128 && m->n_hidden == 1) return 4;
  return 0;
}

}  // namespace

// true when the module's passes run as fused matrix-pipe launches (kernel documentation above for the shapes)
bool gen_mfma_fits(const rl_traj *t, const rl_mlp *m) {
  if (!m->general || !m->has_bias || m->n_hidden < 1 || m->n_hidden > (uint32_t)GM_MAX_HIDDEN) return false;
  if (m->in_dim > (uint32_t)GM_MAX_IN || m->out_dim > 2 || m->in_dim != t->d.D) return false;
  if (gm_width_tiles(m) == 0) return false;
  if ((uint64_t)(t->d.T + 1) * t->d.n * m->in_dim >= (1ull << 30)) return false;  // 32-bit element offsets in the kernel
  return true;
}

// mode: RL_GEN_CRITIC, PASS_INIT, PASS_PPO, PASS_EVAL or PASS_JVP (with the tangent parameters).  Leaves one slab row per WAVE: t->last_rows rows of slabA
// (gradient modes) and slabB.
bool launch_gen_mfma(rl_traj *t, const rl_mlp *m, int mode, const float *d_tangent, uint64_t B_total,
                     const int32_t *d_skip, float clip_lo, float clip_hi) {
  if (!gen_mfma_fits(t, m)) return false;
  if (mode != RL_GEN_CRITIC && mode != PASS_INIT && mode != PASS_PPO && mode != PASS_EVAL && mode != PASS_JVP) return false;
  // the tangent's forward fragments next to the parameters': 114 KB of LDS for two hidden layers, too much for three
  if (mode == PASS_JVP && (d_tangent == nullptr || (gm_lds_bytes(gm_width_tiles(m), (int)m->n_hidden, true) > 160 * 1024 &&
                                                    !(gm_width_tiles(m) == 2 && gp_lds_bytes((int)m->n_hidden, GP_PAIRS, true) <= 160 * 1024))))
    return false;
  if (mode != RL_GEN_CRITIC && m->out_dim != 2) return false;
  if (mode == RL_GEN_CRITIC && m->out_dim != 1) return false;
  GmArgs g{};
  g.params = m->d_params;
  g.tangent = d_tangent;
  g.in_dim = (int)m->in_dim;
  g.out_dim = (int)m->out_dim;
  g.act = m->act;
  g.out_act = m->out_act;
  for (uint32_t l = 0; l < m->n_hidden; ++l) g.width[l] = (int)m->widths[l];
  for (uint32_t l = 0; l <= m->n_hidden; ++l) g.off[l] = (uint32_t)m->layer_offset(l);
  g.P = (uint32_t)m->P;
  const uint64_t n_tiles = (t->B + 31) / 32, cus = (uint64_t)t->eng->prop.multiProcessorCount;
  const int NL = (int)m->n_hidden, gw = gm_width_tiles(m);
  const float inv_B = (mode == RL_GEN_CRITIC ? 2.0f : 1.0f) / (float)B_total;
  gen_ensure(t, m, 0, false, true);  // the P-sized vectors of the update workspace follow the module
  static const bool one_wave = getenv("RELEARN_GEN_ONE_WAVE") != nullptr;  // measurement override: the one-wave kernel
  if (gw == 2 && mode != PASS_EVAL && gp_lds_bytes(NL, GP_PAIRS, mode == PASS_JVP) <= 160 * 1024 && !one_wave) {
    // a tile per pair of waves, two waves per SIMD
    uint64_t nwg = (n_tiles + GP_PAIRS - 1) / GP_PAIRS;
    if (nwg > cus) nwg = cus;
    t->last_rows = (uint32_t)(nwg * GP_PAIRS);  // one slab row per pair
    traj_ensure_slabs(t, t->last_rows, m->P, t->last_rows);
    if (mode == RL_GEN_CRITIC) gp_launch_nl<GM_CRITIC>(t, g, NL, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
    else if (mode == PASS_INIT) gp_launch_nl<PASS_INIT>(t, g, NL, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
    else if (mode == PASS_PPO) gp_launch_nl<PASS_PPO>(t, g, NL, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
    else gp_launch_nl<PASS_JVP>(t, g, NL, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
    return true;
  }
  // (only reachable for a pass the pair kernel was admitted for when RELEARN_GEN_ONE_WAVE forces this kernel: the caller
  // then takes the per-layer path)
  if (gm_lds_bytes(gw, NL, mode == PASS_JVP) > 160 * 1024) return false;
  uint64_t nwg = (n_tiles + GWAVES - 1) / GWAVES;
  if (nwg > cus) nwg = cus;
  t->last_rows = (uint32_t)(nwg * GWAVES < n_tiles ? nwg * GWAVES : n_tiles);  // one slab row per wave that has tiles
  traj_ensure_slabs(t, t->last_rows, m->P, t->last_rows);
  if (mode == RL_GEN_CRITIC) gm_launch_nl<GM_CRITIC>(t, g, NL, gw, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
  else if (mode == PASS_INIT) gm_launch_nl<PASS_INIT>(t, g, NL, gw, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
  else if (mode == PASS_PPO) gm_launch_nl<PASS_PPO>(t, g, NL, gw, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
  else if (mode == PASS_JVP) gm_launch_nl<PASS_JVP>(t, g, NL, gw, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
  else gm_launch_nl<PASS_EVAL>(t, g, NL, gw, (uint32_t)nwg, inv_B, d_skip, clip_lo, clip_hi);
  return true;
}
