// policy_terms.hpp — the per-sample part of a policy pass once the logits are known: d loss / d logits and the loss
// terms of Trpo::update's closure and HessianVectorProduct (policies/trpo.rs:97-146, conjugate_gradient.rs:262-339),
// Ppo::update (policies/ppo.rs:124-137), REINFORCE (surrogate at ratio 1) and DQN's mse_loss (dqn.rs:316-326).  Shared
// by the sample-parallel pass of the single-hidden-layer kernels (k_policy_pass, kernels_update.hip) and by the
// per-layer path of general MLP shapes (kernels_general.hip).
#pragma once
#include "device_fns.hpp"
#include "kernels.hpp"

#include <cfloat>

// z: logits; dzt: tangent logits (PASS_JVP); act: the action taken; adv: advantage (DQN: the target).
// Writes dz[a * B + b] (and lp0 in PASS_INIT); adds to the f64 sums s0, s1, s2 of the pass.
template <int MODE>
__device__ __forceinline__ void policy_sample_terms(const float (&z)[2], const float (&dzt)[2], int act, float adv,
                                                size_t b, size_t B, float *__restrict__ lp0, float *__restrict__ dz,
                                                float inv_B, float clip_lo, float clip_hi, double &s0, double &s1,
                                                double &s2) {
  constexpr int A = 2;
  if (MODE == PASS_JVP) {
  float lp[A], p[A];
  log_softmax_lane<A>(z, lp);
  float pdz = 0.0f;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    p[a] = rl_expf(lp[a]);
    pdz = __builtin_fmaf(p[a], dzt[a], pdz);
  }
#pragma unroll
  for (int a = 0; a < A; ++a) dz[(size_t)a * B + b] = p[a] * (dzt[a] - pdz) * inv_B;
    return;
  }
  float lp[A];
  if (MODE == PASS_DQN) {
    // action_values.gather(-1, actions).mse_loss(targets, Mean) + backward (dqn.rs:316-326); inv_B = 2 / B
    float dq = (act == 0 ? z[0] : z[1]) - adv;
    float g = dq * inv_B;
    dz[b] = act == 0 ? g : 0.0f;
    dz[B + b] = act == 1 ? g : 0.0f;
    s0 += (double)(dq * dq);
    return;
  }
  log_softmax_lane<A>(z, lp);
  if (MODE == PASS_PPO) {
    // clipped surrogate of Ppo::update (policies/ppo.rs:124-137) and its torch-autograd gradient: minimum()
    // splits a tie between its arguments, clamp() passes the gradient inside [lo, hi] (bounds included)
    float l0a = lp0[(size_t)act * B + b];
    float lpa = act == 0 ? lp[0] : lp[1];
    float ratio = rl_expf(lpa - l0a);
    float clipped = ratio < clip_lo ? clip_lo : (ratio > clip_hi ? clip_hi : ratio);
    float u1 = ratio * adv, u2 = clipped * adv;
    bool inside = ratio >= clip_lo && ratio <= clip_hi;
    float gr = u1 < u2 ? adv : (u1 > u2 ? (inside ? adv : 0.0f) : (inside ? adv : 0.5f * adv));
    float c = -(gr * ratio) * inv_B;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      float ind = a == act ? 1.0f : 0.0f;
      dz[(size_t)a * B + b] = c * (ind - rl_expf(lp[a]));
    }
    s0 += (double)(u1 < u2 ? u1 : u2);
    return;
  }
  if (MODE == PASS_INIT) {
    float lpa = act == 0 ? lp[0] : lp[1];
    s2 += (double)(lpa * adv);
    float ratio = rl_expf(lpa - lpa);
    float c = -(ratio * adv) * inv_B;
    float ent = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      float pa = rl_expf(lp[a]);
      float ind = a == act ? 1.0f : 0.0f;
      lp0[(size_t)a * B + b] = lp[a];
      dz[(size_t)a * B + b] = c * (ind - pa);
      float cl = lp[a] < -FLT_MAX ? -FLT_MAX : lp[a];
      ent += cl * pa;
    }
    s0 += (double)(ratio * adv);
    s1 += (double)(-ent);
  } else {  // PASS_EVAL
    float l0[A];
#pragma unroll
    for (int a = 0; a < A; ++a) l0[a] = lp0[(size_t)a * B + b];
    float lpa = act == 0 ? lp[0] : lp[1];
    float l0a = act == 0 ? l0[0] : l0[1];
    float ratio = rl_expf(lpa - l0a);
    float kl = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      float rel = l0[a] - lp[a];
      if (rel < -FLT_MAX) rel = -FLT_MAX;
      kl += rel * rl_expf(l0[a]);
    }
    s0 += (double)(ratio * adv);
    s1 += (double)kl;
  }
}
