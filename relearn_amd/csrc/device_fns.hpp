// device_fns.hpp — per-lane device functions shared by the kernels.
//
// Arithmetic contract: this library is compiled with -ffp-contract=off.  Every fused multiply-add is
// written explicitly (__builtin_fmaf / __builtin_fma); every other a*b+c is two roundings, exactly like
// the reference's Rust and like the oracle (gcc -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/rl_chacha.h"
#include "../../include/rl_detmath.h"
#include "engine.hpp"

// ---------------------------------------------------------------- CartPole physics
// InternalPhysicalConstants::{next_state, angular_acceleration, normal_force}
// (reference src/envs/cartpole.rs:306-446).  f64 throughout, operation order preserved.
__device__ __forceinline__ double cp_angular_acceleration(const CartPoleDev &c, double thdot, double applied_force,
                                                          double signed_cart_friction, double w2, double sin_a,
                                                          double cos_a) {
  double alpha =
      (-applied_force - c.mass_length_pole * w2 * (sin_a + signed_cart_friction * cos_a)) * c.inv_total_mass;
  double beta = c.friction_pole * thdot / c.mass_length_pole;
  double numerator = c.gravity * sin_a + cos_a * (alpha + c.gravity * signed_cart_friction) - beta;
  double denominator =
      c.length_half_pole * (4.0 / 3.0 - c.mass_pole * cos_a * c.inv_total_mass * (cos_a - signed_cart_friction));
  return numerator / denominator;
}

__device__ __forceinline__ double cp_normal_force(const CartPoleDev &c, double acc, double w2, double sin_a,
                                                  double cos_a) {
  return c.total_weight - c.mass_length_pole * (acc * sin_a + w2 * cos_a);
}

struct LaneState {
  double x, xdot, th, thdot;
  uint32_t nv_pos;
  uint32_t steps_remaining;
  uint32_t reset_count;
};

// CartPole::step (cartpole.rs:128-154) + Wrapped<_, StepLimit>::step tail (wrappers/step_limit.rs:216-222).
// Returns the successor code; on Continue/Interrupt `s` holds the next state.
__device__ __forceinline__ int cp_step(const CartPoleDev &c, LaneState &s, int action) {
  double applied_force = action == 0 ? -c.action_force : c.action_force;
  double signed_cart_friction = s.nv_pos ? c.friction_cart : -c.friction_cart;
  double sin_a, cos_a;
  rl_sincos(s.th, &sin_a, &cos_a);
  double w2 = s.thdot * s.thdot;
  double acc = cp_angular_acceleration(c, s.thdot, applied_force, signed_cart_friction, w2, sin_a, cos_a);
  double nf = cp_normal_force(c, acc, w2, sin_a, cos_a);
  uint32_t nv_pos = (rl_f64_bits(nf * s.xdot) >> 63) ? 0u : 1u;  // is_sign_positive
  if (nv_pos != s.nv_pos) {
    signed_cart_friction = -signed_cart_friction;
    acc = cp_angular_acceleration(c, s.thdot, applied_force, signed_cart_friction, w2, sin_a, cos_a);
    nf = cp_normal_force(c, acc, w2, sin_a, cos_a);
  }
  double force_pole = c.mass_length_pole * (w2 * sin_a + acc * cos_a);
  double force_friction = -signed_cart_friction * nf;
  double net_force = applied_force + force_pole + force_friction;
  double cart_acc = net_force * c.inv_total_mass;
  double xdot = s.xdot + c.time_step * cart_acc;
  double x = s.x + c.time_step * xdot;
  double thdot = s.thdot + c.time_step * acc;
  double th = s.th + c.time_step * s.thdot;
  bool terminal = __builtin_fabs(x) > c.max_pos || __builtin_fabs(th) > c.max_angle;
  if (terminal) return RL_SUCC_TERMINATE;
  s.x = x;
  s.xdot = xdot;
  s.th = th;
  s.thdot = thdot;
  s.nv_pos = nv_pos;
  if (c.limit_kind != RL_LIMIT_NONE) {
    s.steps_remaining -= 1;
    if (s.steps_remaining == 0) return RL_SUCC_INTERRUPT;
  }
  return RL_SUCC_CONTINUE;
}

// features_out of StepLimitObsSpace<CartPolePhysicalStateSpace> (spaces/interval.rs:108-116,
// wrappers/step_limit.rs:127-140,194-200): each field `as f32`, `remaining` last.
template <int D>
__device__ __forceinline__ void cp_features(const CartPoleDev &c, const LaneState &s, float (&f)[D]) {
  f[0] = (float)s.x;
  f[1] = (float)s.xdot;
  f[2] = (float)s.th;
  f[3] = (float)s.thdot;
  if (D == 5) f[4] = (float)((double)s.steps_remaining / (double)c.max_steps);
}

// CartPole::initial_state (cartpole.rs:103-115) from the lane's env stream: reset k reads words [8k, 8k+8).
__device__ __forceinline__ void cp_reset(const CartPoleDev &c, LaneState &s, uint64_t global_lane) {
  uint32_t w[16];
  uint32_t k = s.reset_count;
  rl_chacha_block(c.key_env, (uint64_t)(k >> 1), global_lane, 4, w);
  // the upper or the lower half of the block by a bit select per word (v_bfi_b32): an index that depends on the lane
  // would put the block into scratch memory
  const uint32_t hi = 0u - (k & 1u);
  uint32_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (w[8 + i] & hi) | (w[i] & ~hi);
  s.x = rl_uniform_f64_from_u64(((uint64_t)v[1] << 32) | v[0], c.init_low, c.init_scale);
  s.xdot = rl_uniform_f64_from_u64(((uint64_t)v[3] << 32) | v[2], c.init_low, c.init_scale);
  s.th = rl_uniform_f64_from_u64(((uint64_t)v[5] << 32) | v[4], c.init_low, c.init_scale);
  s.thdot = rl_uniform_f64_from_u64(((uint64_t)v[7] << 32) | v[6], c.init_low, c.init_scale);
  s.nv_pos = 1;
  s.steps_remaining = c.max_steps;
  s.reset_count = k + 1;
}

__device__ __forceinline__ void lane_load(const EnvStateDev &st, uint32_t i, LaneState &s) {
  s.x = st.x[i];
  s.xdot = st.xdot[i];
  s.th = st.th[i];
  s.thdot = st.thdot[i];
  s.nv_pos = st.nv_pos[i];
  s.steps_remaining = st.steps_remaining[i];
  s.reset_count = st.reset_count[i];
}

__device__ __forceinline__ void lane_store(const EnvStateDev &st, uint32_t i, const LaneState &s) {
  st.x[i] = s.x;
  st.xdot[i] = s.xdot;
  st.th[i] = s.th;
  st.thdot[i] = s.thdot;
  st.nv_pos[i] = (uint8_t)s.nv_pos;
  st.steps_remaining[i] = s.steps_remaining;
  st.reset_count[i] = s.reset_count;
}

// ---------------------------------------------------------------- MLP forward
// Mlp::forward (torch/modules/ff/mlp.rs:139-151): relu(x W1^T + b1) W2^T + b2.  Summation orders (shared with the oracle):
//   layer 1: acc = bias; acc = fma(x_k, w_k, acc) with k ascending;
//   output layer: MLP_GROUPS = 16 interleaved partial chains — unit j feeds chain j mod 16, every chain starts from 0 and
//   runs acc = fma(h_j, w_j, acc), j ascending — combined by the fixed binary tree ((c0 + c1) + (c2 + c3)) + ... over
//   neighbouring chains, bias added last.  One thread may own all sixteen chains of a row (mlp_forward_lane*: sixteen
//   independent fma chains instead of one 128-long one), or G = 2, 4, 8, 16 cooperating threads own 16 / G neighbouring
//   chains each and meet through xor-shuffles (mlp_forward_group_lds): the result does not depend on G.
constexpr int MLP_GROUPS = 16;

template <int A>
__device__ __forceinline__ void mlp_tree(float (&c)[A][MLP_GROUPS], const float *__restrict__ b2, float (&z)[A]) {
#pragma unroll
  for (int a = 0; a < A; ++a) {
#pragma unroll
    for (int width = 1; width < MLP_GROUPS; width *= 2)
#pragma unroll
      for (int g = 0; g < MLP_GROUPS; g += 2 * width) c[a][g] = c[a][g] + c[a][g + width];
    z[a] = c[a][0] + b2[a];
  }
}

// weights wave-uniform (scalar loads), one row per lane
template <int D, int A>
__device__ __forceinline__ void mlp_forward_lane(const float *__restrict__ params, int H, const float (&x)[D],
                                                 float (&z)[A]) {
  const float *__restrict__ W1 = params;
  const float *__restrict__ b1 = W1 + H * D;
  const float *__restrict__ W2 = b1 + H;
  const float *__restrict__ b2 = W2 + A * H;
  float c[A][MLP_GROUPS];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int g = 0; g < MLP_GROUPS; ++g) c[a][g] = 0.0f;
  int j0 = 0;
  for (; j0 + MLP_GROUPS <= H; j0 += MLP_GROUPS) {
#pragma unroll
    for (int g = 0; g < MLP_GROUPS; ++g) {
      const int j = j0 + g;
      float acc = b1[j];
#pragma unroll
      for (int k = 0; k < D; ++k) acc = __builtin_fmaf(x[k], W1[j * D + k], acc);
      const float h = acc > 0.0f ? acc : 0.0f;
#pragma unroll
      for (int a = 0; a < A; ++a) c[a][g] = __builtin_fmaf(h, W2[a * H + j], c[a][g]);
    }
  }
#pragma unroll
  for (int g = 0; g < MLP_GROUPS; ++g) {  // hidden sizes that are not a multiple of 16
    const int j = j0 + g;
    if (j < H) {
      float acc = b1[j];
#pragma unroll
      for (int k = 0; k < D; ++k) acc = __builtin_fmaf(x[k], W1[j * D + k], acc);
      const float h = acc > 0.0f ? acc : 0.0f;
#pragma unroll
      for (int a = 0; a < A; ++a) c[a][g] = __builtin_fmaf(h, W2[a * H + j], c[a][g]);
    }
  }
  mlp_tree<A>(c, b2, z);
}

// The same forward with the parameters parked in LDS as one 8-float record per hidden unit {W1[j][0..D), pad, b1[j],
// W2[0][j], W2[1][j]} (two 16-byte broadcast reads per unit instead of a stream of scalar loads); `pk` also holds b2
// behind the H records.  Four floats of padding follow every eight records: the G threads of a row read records
// 16 / G units apart at once, and 8 units = 256 bytes is the width of the LDS — without the padding threads g and
// g + G / 2 (G = 2: the two threads of every row) hit the same banks with different addresses on every read (the 64k-lane
// rollout measured SQ_LDS_BANK_CONFLICT at 0.49 of its LDS cycles, LDS busy 0.67 of the time).
constexpr int MLP_PK_FLOATS = 8 * 128 + 4 * 16 + 4;  // records + padding of 128 hidden units, b2
__host__ __device__ constexpr int mlp_pk_at(int j) { return 8 * j + 4 * (j >> 3); }  // first float of unit j's record
template <int D>
__device__ __forceinline__ void mlp_pack_lds(float *__restrict__ pk, const float *__restrict__ params, int H, int tid,
                                             int nthreads) {
  const float *__restrict__ W1 = params;
  const float *__restrict__ b1 = W1 + H * D;
  const float *__restrict__ W2 = b1 + H;
  const float *__restrict__ b2 = W2 + 2 * H;
  for (int q = tid; q < 8 * H + 2; q += nthreads) {
    float v = 0.0f;
    if (q >= 8 * H) {
      pk[mlp_pk_at(H) + (q - 8 * H)] = b2[q - 8 * H];
      continue;
    }
    const int j = q >> 3, k = q & 7;
    if (k < D) v = W1[j * D + k];
    else if (k == 5) v = b1[j];
    else if (k == 6) v = W2[j];
    else if (k == 7) v = W2[H + j];
    pk[mlp_pk_at(j) + k] = v;
  }
}

// NB hidden units j0, j0 + 1, ... from their LDS records, their layer-1 chains advancing side by side (NB independent
// fma chains in flight, all records requested before the first use): relu and the units' terms of the output chains
template <int D, int NB>
__device__ __forceinline__ void mlp_units_lds(const float *__restrict__ pk, int j0, const float (&x)[D], float *c0,
                                              float *c1) {
  static_assert(D <= 5, "the packed record holds at most five input weights");
  float4 lo[NB], hi[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    lo[u] = *reinterpret_cast<const float4 *>(pk + mlp_pk_at(j0 + u));
    hi[u] = *reinterpret_cast<const float4 *>(pk + mlp_pk_at(j0 + u) + 4);
  }
  float acc[NB];
#pragma unroll
  for (int u = 0; u < NB; ++u) acc[u] = hi[u].y;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const float w = k == 0 ? lo[u].x : k == 1 ? lo[u].y : k == 2 ? lo[u].z : k == 3 ? lo[u].w : hi[u].x;
      acc[u] = __builtin_fmaf(x[k], w, acc[u]);
    }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const float h = acc[u] > 0.0f ? acc[u] : 0.0f;
    c0[u] = __builtin_fmaf(h, hi[u].z, c0[u]);
    c1[u] = __builtin_fmaf(h, hi[u].w, c1[u]);
  }
}

// G cooperating threads per row (G = 1, 2, 4, 8, 16; the G threads of a row are G consecutive lanes, `g` = lane % G):
// thread g owns the chains [g * 16 / G, (g + 1) * 16 / G), adds them by the tree's lower levels and meets its
// neighbours through xor-shuffles for the upper ones; every thread of the row ends with the same z.
template <int D, int G>
__device__ __forceinline__ void mlp_forward_group_lds(const float *__restrict__ pk, int H, int g, const float (&x)[D],
                                                      float (&z)[2]) {
  constexpr int OWN = MLP_GROUPS / G;  // chains per thread
  float c[2][OWN];
#pragma unroll
  for (int q = 0; q < OWN; ++q) c[0][q] = c[1][q] = 0.0f;
  int j0 = 0;
  constexpr int NB = OWN < 4 ? OWN : 4;  // units whose chains run side by side (eight measured slower)
  for (; j0 + MLP_GROUPS <= H; j0 += MLP_GROUPS) {
#pragma unroll
    for (int q = 0; q < OWN; q += NB) mlp_units_lds<D, NB>(pk, j0 + g * OWN + q, x, &c[0][q], &c[1][q]);
  }
  if (j0 < H) {  // hidden sizes that are not a multiple of 16
#pragma unroll
    for (int q = 0; q < OWN; ++q) {
      const int j = j0 + g * OWN + q;
      if (j < H) mlp_units_lds<D, 1>(pk, j, x, &c[0][q], &c[1][q]);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int width = 1; width < OWN; width *= 2)
#pragma unroll
      for (int q = 0; q < OWN; q += 2 * width) c[a][q] = c[a][q] + c[a][q + width];
    float s = c[a][0];
#pragma unroll
    for (int m = 1; m < G; m *= 2) s = s + __shfl_xor(s, m, 64);  // a + b == b + a bit for bit: both partners agree
    z[a] = s + pk[mlp_pk_at(H) + a];
  }
}

template <int D>
__device__ __forceinline__ void mlp_forward_lane_lds(const float *__restrict__ pk, int H, const float (&x)[D],
                                                     float (&z)[2]) {
  mlp_forward_group_lds<D, 1>(pk, H, 0, x, z);
}

// Categorical::new: log_softmax (torch/distributions/categorical.rs:29-33)
template <int A>
__device__ __forceinline__ void log_softmax_lane(const float (&z)[A], float (&lp)[A]) {
  float m = z[0];
#pragma unroll
  for (int a = 1; a < A; ++a)
    if (z[a] > m) m = z[a];
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < A; ++a) s += rl_expf(z[a] - m);
  float ls = rl_logf(s);
#pragma unroll
  for (int a = 0; a < A; ++a) lp[a] = (z[a] - m) - ls;
}

// `log_probs.exp().multinomial(1, true)` (categorical.rs:53) with an explicit uniform draw
template <int A>
__device__ __forceinline__ int categorical_sample_lane(const float (&lp)[A], float u) {
  float cum = 0.0f;
  int act = A - 1;
  bool found = false;
#pragma unroll
  for (int a = 0; a + 1 < A; ++a) {
    cum += rl_expf(lp[a]);
    if (!found && u < cum) {
      act = a;
      found = true;
    }
  }
  return act;
}

// ---------------------------------------------------------------- block reductions (deterministic)
// Sum over a block with a fixed tree order; result valid in thread 0.
template <int BLOCK, typename T>
__device__ __forceinline__ T block_sum(T v, T *smem) {
  int tid = threadIdx.x;
  smem[tid] = v;
  __syncthreads();
#pragma unroll
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) smem[tid] = smem[tid] + smem[tid + s];
    __syncthreads();
  }
  T r = smem[0];
  __syncthreads();
  return r;
}
