// policy_fast.hpp — transcendentals of the UPDATE passes (shared by kernels_mfma.hip and kernels_gen_mfma.hip):
// v_exp_f32 / v_log_f32 / v_rcp_f32 (about one ulp) with the base change as a multiplication — 2-3 instructions instead
// of the ~25 of rl_expf / rl_logf (include/rl_detmath.h).  The deterministic versions stay where results are compared
// bit for bit with the oracle (rollouts, values, GAE, targets); the update passes are compared with the f64 oracle
// within f32 tolerances, and they agree with each other because log pi_0 (stored by PASS_INIT) and every later log pi
// come from the same code.
#pragma once

__device__ __forceinline__ float fast_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// The two-way softmax of the logits {zd, 0} from ONE exponential: t = exp(-|zd|) (the rounding of the argument's base
// change is carried through: hi + lo = -|zd| log2 e to ~2^-48), r = 1 / (1 + t) (reciprocal + one Newton step), the
// probabilities r and t r — they sum to one within an ulp, so the two logit gradients stay antisymmetric, which two
// independent exponentials are not (measured: a 1.4e-6 relative bias in db2) — and lse = max(zd, 0) + log(1 + t).
struct SoftPair {
  float lp[2], p[2];
};
__device__ __forceinline__ SoftPair soft_pair(float zd) {
  const float ax = -__builtin_fabsf(zd);
  const float hi = ax * 1.4426950408889634f;
  const float lo = __builtin_fmaf(ax, 1.4426950408889634f, -hi) + ax * 1.925963033500343e-08f;  // log2 e = hi part + 1.93e-8
  const float e = __builtin_amdgcn_exp2f(hi);
  const float t = __builtin_fmaf(e * lo, 0.6931471805599453f, e);
  const float s1 = 1.0f + t;
  float r = __builtin_amdgcn_rcpf(s1);
  r = __builtin_fmaf(__builtin_fmaf(-s1, r, 1.0f), r, r);
  const float big = r, small = t * r;
  const float lse = __builtin_fmaxf(zd, 0.0f) + 0.6931471805599453f * __builtin_amdgcn_logf(s1);
  SoftPair o;
  o.lp[0] = zd - lse;
  o.lp[1] = -lse;
  o.p[0] = zd >= 0.0f ? big : small;
  o.p[1] = zd >= 0.0f ? small : big;
  return o;
}

// sigmoid and tanh of the update passes: one exponential of -|x| (no overflow), a reciprocal with one Newton step
__device__ __forceinline__ float fast_sigmoidf(float x) {
  const float t = __builtin_amdgcn_exp2f(-__builtin_fabsf(x) * 1.4426950408889634f);
  const float s1 = 1.0f + t;
  float r = __builtin_amdgcn_rcpf(s1);
  r = __builtin_fmaf(__builtin_fmaf(-s1, r, 1.0f), r, r);
  return x >= 0.0f ? r : t * r;
}
__device__ __forceinline__ float fast_tanhf(float x) {
  // (1 - t) / (1 + t), t = exp(-2 |x|); below |x| = 2^-6 the cancellation in 1 - t would cost bits: odd series instead
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_exp2f(ax * -2.8853900817779268f);
  const float s1 = 1.0f + t;
  float r = __builtin_amdgcn_rcpf(s1);
  r = __builtin_fmaf(__builtin_fmaf(-s1, r, 1.0f), r, r);
  const float big = (1.0f - t) * r;
  const float x2 = ax * ax;
  const float small = ax * __builtin_fmaf(x2, __builtin_fmaf(x2, 0.13333333f, -0.33333334f), 1.0f);
  const float m = ax < 0.015625f ? small : big;
  return x < 0.0f ? -m : m;
}
