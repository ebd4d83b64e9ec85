// kernels_rollout.hip — env step, fused persistent rollout, value forward and GAE scan.
//
// Layout: one env per lane, struct-of-arrays state in HBM, time-major trajectory planes so that every
// global access of a wavefront is one contiguous 256-B (f32) / 64-B (u8) segment.
#include <cstdlib>

#include "bf16_tile.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"

// ---------------------------------------------------------------- reset / observe / standalone step
__global__ void k_env_reset(CartPoleDev c, EnvStateDev st, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  LaneState s;
  s.reset_count = st.reset_count[i];
  cp_reset(c, s, c.lane_offset + i);
  lane_store(st, i, s);
}

template <int D>
__global__ void k_env_observe(CartPoleDev c, EnvStateDev st, uint32_t n, float *__restrict__ obs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  LaneState s;
  lane_load(st, i, s);
  float f[D];
  cp_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) obs[(size_t)d * n + i] = f[d];
}

// Environment::step for every lane (reference src/envs/cartpole.rs:128-154 through the step-limit wrapper),
// with auto-reset.  Algorithmic traffic per env-step: read state 32 B + sign 1 B + remaining 4 B
// + action 1 B; write state 37 B + reward 4 B + flag 1 B + next-obs 4*D B  => 100 B at D = 5.
// (FILLED: the same code under a second name for launches of >= 2^20 lanes, so that a kernel trace keeps the workload's
// latency-bound launches and the chip-filling ones of bench.py's `roofline_env_step.filled` apart)
template <int D, bool FILLED>
__global__ void __launch_bounds__(256) k_env_step(CartPoleDev c, EnvStateDev st, uint32_t n,
                                                  const uint8_t *__restrict__ actions, float *__restrict__ reward,
                                                  uint8_t *__restrict__ flag, float *__restrict__ obs_next,
                                                  float *__restrict__ term_obs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  LaneState s;
  lane_load(st, i, s);
  int a = actions[i];
  int succ = cp_step(c, s, a);
  float f[D];
  if (succ == RL_SUCC_INTERRUPT) {
    cp_features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) term_obs[(size_t)d * n + i] = f[d];
  }
  if (succ != RL_SUCC_CONTINUE) cp_reset(c, s, c.lane_offset + i);
  cp_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) obs_next[(size_t)d * n + i] = f[d];
  reward[i] = 1.0f;  // Reward(1.0) as f32
  flag[i] = (uint8_t)succ;
  lane_store(st, i, s);
}

// ---------------------------------------------------------------- fused persistent rollout
// HOT LOOP A of the reference (src/simulation/steps.rs:113-167 + torch/agents/policies/actor.rs:42-55):
// T env-actor steps per lane in one launch; lane state lives in registers for the whole horizon; the only
// HBM traffic is the 26 B/step trajectory record (obs 20 + action 1 + reward 4 + flag 1) plus the sparse
// interrupt successor observations.  The actor's uniform draw for global step t is word t of the lane's
// ChaCha8 actor stream; a 16-word block is regenerated every 16 steps and parked in an LDS column that is
// private to the thread (no bank conflicts: threads are consecutive in the fastest dimension).
// A step is a chain of dependent work (features -> 128-unit MLP -> softmax -> sample -> f64 sincos + physics), so a
// launch lasts T x the latency of one step whatever the lane count.  G threads per lane (G consecutive lanes of a wave)
// share the MLP — each owns 16 / G of the output layer's sixteen partial chains (mlp_forward_group_lds) — and repeat the
// cheap rest redundantly (a wavefront pays per instruction, not per active lane); thread 0 of the group stores.  The
// host picks G so that the launch has up to two waves per SIMD: 16 at <= 8,192 lanes, 2 at 65,536.
template <int D, int BLOCK, int G>
__global__ void __launch_bounds__(BLOCK) k_rollout_cartpole(CartPoleDev c, EnvStateDev st, TrajDev tr,
                                                            const float *__restrict__ policy, int H,
                                                            uint64_t t_global) {
  __shared__ uint32_t actor_words[16 * BLOCK];
  __shared__ __attribute__((aligned(16))) float pk[MLP_PK_FLOATS];  // the policy, one 8-float record per hidden unit
  const uint32_t n = tr.n, T = tr.T;
  mlp_pack_lds<D>(pk, policy, H, threadIdx.x, BLOCK);
  // new observations: their magnitude range is measured by the value forward that follows (k_mlp_forward_rows<.., RANGE>),
  // which starts from these words
  if (blockIdx.x == 0) bt::range_reset(tr.range, (int)threadIdx.x, BLOCK);
  __syncthreads();
  // (Launch order = lane order.  Giving each XCD a contiguous range of lanes, so that the 32-byte pieces a wave of G = 2
  // writes into the byte-wide `action` / `flag` planes meet in one L2, was measured in round 4: WRITE_SIZE is 222 MB per
  // launch either way — the 302 MB of the round-3 counter pass was the FIRST launch over untouched memory — and the
  // remapped launch is 2 % slower.)
  const uint32_t i = (blockIdx.x * BLOCK + threadIdx.x) / G;
  const int g = threadIdx.x % G;
  // lanes past the end follow lane n - 1 without storing: the group shuffles below need every lane of a group
  const bool live = i < n;
  const uint32_t il = live ? i : n - 1;
  const bool writer = live && g == 0;
  const uint64_t lane = c.lane_offset + il;
  LaneState s;
  lane_load(st, il, s);
  const size_t plane = (size_t)(T + 1) * n;
  uint64_t cur_block = ~0ull;
  for (uint32_t t = 0; t < T; ++t) {
    float f[D];
    cp_features<D>(c, s, f);
    if (writer) {
#pragma unroll
      for (int d = 0; d < D; ++d) tr.obs[d * plane + (size_t)t * n + il] = f[d];
    }
    // actor draw
    uint64_t w = t_global + t;
    uint64_t blk = w >> 4;
    if (blk != cur_block) {
      uint32_t words[16];
      rl_chacha_block(c.key_actor, blk, lane, 4, words);
#pragma unroll
      for (int k = 0; k < 16; ++k) actor_words[k * BLOCK + threadIdx.x] = words[k];
      cur_block = blk;
    }
    float u = rl_u32_to_unit_f32(actor_words[(uint32_t)(w & 15) * BLOCK + threadIdx.x]);
    float z[2], lp[2];
    mlp_forward_group_lds<D, G>(pk, H, g, f, z);
    log_softmax_lane<2>(z, lp);
    int a = categorical_sample_lane<2>(lp, u);
    int succ = cp_step(c, s, a);
    size_t o = (size_t)t * n + il;
    if (writer) {
      tr.action[o] = (uint8_t)a;
      tr.reward[o] = 1.0f;
      tr.flag[o] = (uint8_t)succ;
    }
    if (succ == RL_SUCC_INTERRUPT) {
      cp_features<D>(c, s, f);
      if (writer) {
#pragma unroll
        for (int d = 0; d < D; ++d) tr.term_obs[(size_t)d * T * n + o] = f[d];
      }
    }
    if (succ != RL_SUCC_CONTINUE) cp_reset(c, s, lane);
  }
  float f[D];
  cp_features<D>(c, s, f);
  if (writer) {
#pragma unroll
    for (int d = 0; d < D; ++d) tr.obs[d * plane + (size_t)T * n + il] = f[d];
    lane_store(st, il, s);
  }
}

// ---------------------------------------------------------------- value forward over SoA rows
// eval_extended_state_values (torch/agents/critics/mod.rs:116-131) on the lane layout: V for every
// obs[.][t][lane], t = 0..T.  One row per lane, A outputs.
// RANGE: the rows are a trajectory's observation planes, every one of them — the pass also folds the magnitude range of
// what it reads (bits of the smallest non-zero |x| and of the largest |x|) into range[0..1] for the fused update kernels'
// guard (bf16_tile.hpp; the words were reset by the rollout that wrote the planes).  A workgroup touches the shared
// words only when it improves on what it reads there: a handful of the tens of thousands of workgroups do.
template <int D, int A, bool RANGE = false>
__global__ void __launch_bounds__(256) k_mlp_forward_rows(const float *__restrict__ params, int H,
                                                          const float *__restrict__ in, size_t in_plane,
                                                          size_t rows, float *__restrict__ out, size_t out_plane,
                                                          uint32_t *__restrict__ range = nullptr) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!RANGE && r >= rows) return;
  const bool live = r < rows;
  float x[D], z[A];
#pragma unroll
  for (int d = 0; d < D; ++d) x[d] = live ? in[d * in_plane + r] : 0.0f;
  if (RANGE) {
    uint32_t lo = 0x7F7FFFFFu, hi = 0u;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const uint32_t a = __builtin_bit_cast(uint32_t, x[d]) & 0x7FFFFFFFu;
      hi = a > hi ? a : hi;
      lo = a != 0u && a < lo ? a : lo;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
      const uint32_t ol = (uint32_t)__shfl_xor((int)lo, m, 64), oh = (uint32_t)__shfl_xor((int)hi, m, 64);
      lo = ol < lo ? ol : lo;
      hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) bt::range_fold(range, blockIdx.x * 4 + (threadIdx.x >> 6), lo, hi);
    if (!live) return;
  }
  mlp_forward_lane<D, A>(params, H, x, z);
#pragma unroll
  for (int a = 0; a < A; ++a) out[a * out_plane + r] = z[a];
}

// the critic's value of one observation (feature d at x[d * stride])
template <int D>
__device__ __forceinline__ float critic_value_of(const float *__restrict__ critic, int H, const float *__restrict__ x0,
                                              size_t stride) {
  float x[D], z[1];
#pragma unroll
  for (int d = 0; d < D; ++d) x[d] = x0[(size_t)d * stride];
  mlp_forward_lane<D, 1>(critic, H, x, z);
  return z[0];
}

// ---------------------------------------------------------------- GAE + reward-to-go scan
// temporal_differences + gae + reward_to_go (critics/mod.rs:101-105,158-199) with the arithmetic of
// inplace_discounted_cumsum_from_end (torch/packed.rs:312-342): delta = (r + gamma*V') - V, every op
// rounded; a = a + (b * discount).  Lane-major: each lane walks its own time axis backwards, so
// trim_end/trim_start (packed.rs:195-267) are index shifts.  Episode ends: Terminate -> V' = 0;
// Interrupt -> V' = V(term_obs); a lane cut by the horizon is an Interrupt with successor obs[T].
// Algorithmic traffic: r 4 + V 4 + flag 1 in, adv 4 + rtg 4 out = 17 B/sample.
template <int D>
__global__ void __launch_bounds__(256) k_gae_scan(TrajDev tr, const float *__restrict__ critic, int H, float gamma,
                                                  float lambda, int rtg_only) {
  const uint32_t n = tr.n, T = tr.T;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float disc = lambda * gamma;
  float adv_next = 0.0f, rtg_next = 0.0f;
  if (rtg_only) {
    // RewardToGo critic (critics/rtg.rs:28-33): the "advantage" is the empirical discounted return
    for (uint32_t t = T; t-- > 0;) {
      size_t o = (size_t)t * n + i;
      float r = tr.reward[o], g;
      if (tr.flag[o] != RL_SUCC_CONTINUE || t == T - 1) {
        g = r;
      } else {
        float pg = rtg_next * gamma;
        g = r + pg;
      }
      tr.adv[o] = g;
      tr.rtg[o] = g;
      rtg_next = g;
    }
    return;
  }
  // The scan is one dependent chain per lane, but its inputs are not: the loads of GAE_AHEAD time steps are issued
  // before the chain walks through them (one memory round trip per 16 steps instead of one per step: 68 -> 36 us at
  // 4,096 lanes, 121 -> 51 us at 65,536).  Same operations in the same order as before.  Measured alternatives: a second
  // register buffer filled one batch ahead (the unrolled body, with the Interrupt branch's MLP inlined 32 times, no
  // longer fits the instruction cache: 190 / 396 us); the Interrupt branch as an out-of-line call (45 / 82 us).
  constexpr int GAE_AHEAD = 16;
  struct Batch {
    uint8_t f[GAE_AHEAD];
    float r[GAE_AHEAD], v[GAE_AHEAD];
  };
  const uint8_t *__restrict__ flag_in = tr.flag;
  const float *__restrict__ reward_in = tr.reward, *__restrict__ values_in = tr.values;
  auto load = [&](uint32_t hi, Batch &b) {  // steps hi - 1, hi - 2, ... (clamped at 0: the surplus is never used)
#pragma unroll
    for (int u = 0; u < GAE_AHEAD; ++u) {
      const size_t o = (size_t)(hi > (uint32_t)u ? hi - 1 - (uint32_t)u : 0u) * n + i;
      b.f[u] = flag_in[o];
      b.r[u] = reward_in[o];
      b.v[u] = values_in[o];
    }
  };
  float v_next = values_in[(size_t)T * n + i];
  auto walk = [&](uint32_t hi, const Batch &b) {
#pragma unroll
    for (int u = 0; u < GAE_AHEAD; ++u) {
      if (hi > (uint32_t)u) {
        const uint32_t t = hi - 1 - (uint32_t)u;
        const size_t o = (size_t)t * n + i;
        const uint8_t f = b.f[u];
        const float r = b.r[u], v = b.v[u];
        float vn;
        bool ends;
        if (f == RL_SUCC_TERMINATE) {
          vn = 0.0f;
          ends = true;
        } else if (f == RL_SUCC_INTERRUPT) {
          vn = critic_value_of<D>(critic, H, tr.term_obs + o, (size_t)T * n);
          ends = true;
        } else {
          vn = v_next;
          ends = (t == T - 1);
        }
        float dn = gamma * vn;
        float tmp = r + dn;
        float delta = tmp - v;
        float a, g;
        if (ends) {
          a = delta;
          g = r;
        } else {
          float pa = adv_next * disc;
          a = delta + pa;
          float pg = rtg_next * gamma;
          g = r + pg;
        }
        tr.adv[o] = a;
        tr.rtg[o] = g;
        adv_next = a;
        rtg_next = g;
        v_next = v;
      }
    }
  };
  Batch b;
  for (uint32_t hi = T; hi > 0;) {
    load(hi, b);
    walk(hi, b);
    if (hi <= (uint32_t)GAE_AHEAD) break;
    hi -= GAE_AHEAD;
  }
}

// ---------------------------------------------------------------- value targets of the critic update
// StepValueTarget::targets (critics/mod.rs:203-229), evaluated once per update under no-grad (opt.rs:101-104).
// OneStepTd = one_step_values (critics/mod.rs:139-150): rewards + discount_factor * estimated_next_values, where the
// next value is the masked extended value of eval_extended_state_values (:116-131): 0 after Terminate, V(successor
// observation) after Interrupt, V(obs[t+1]) inside an episode (slot T of `values` at the horizon cut).  One thread
// per sample: no scan.  `scalar * tensor`, then `tensor + tensor`: two roundings.
template <int D>
__global__ void __launch_bounds__(256) k_value_targets_td(TrajDev tr, const float *__restrict__ critic, int H,
                                                          float gamma) {
  const uint32_t n = tr.n, T = tr.T;
  const size_t B = (size_t)T * n;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= B) return;
  const uint8_t f = tr.flag[o];
  float vn;
  if (f == RL_SUCC_TERMINATE) {
    vn = 0.0f;
  } else if (f == RL_SUCC_INTERRUPT) {
    float x[D], z[1];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = tr.term_obs[(size_t)d * B + o];
    mlp_forward_lane<D, 1>(critic, H, x, z);
    vn = z[0];
  } else {
    vn = tr.values[o + n];
  }
  const float dn = gamma * vn;
  tr.tgt[o] = tr.reward[o] + dn;
}

// RewardToGo = reward_to_go (critics/mod.rs:101-105): the lane's reverse scan of k_gae_scan, into the target plane
__global__ void __launch_bounds__(64) k_value_targets_rtg(TrajDev tr, float gamma) {
  const uint32_t n = tr.n, T = tr.T;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float next = 0.0f;
  for (uint32_t t = T; t-- > 0;) {
    const size_t o = (size_t)t * n + i;
    const float r = tr.reward[o];
    float g;
    if (tr.flag[o] != RL_SUCC_CONTINUE || t == T - 1) {
      g = r;
    } else {
      const float pg = next * gamma;
      g = r + pg;
    }
    tr.tgt[o] = g;
    next = g;
  }
}

// ---------------------------------------------------------------- host launchers
static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

// ---------------------------------------------------------------- magnitude range of the observation planes
// range[0] = bits of the smallest non-zero |x|, range[1] = bits of the largest |x| over `count` floats (|x| as an unsigned
// integer orders like the magnitude; NaN and Inf come out on top): what the fused update kernels' range guard reads
// (bf16_tile.hpp).  One pass over the planes per period, HBM-bound (0.17 GB at the metric's size: ~40 us).
__global__ void __launch_bounds__(256) k_obs_range(const float *__restrict__ x, size_t count, uint32_t *__restrict__ range) {
  __shared__ uint32_t los[4], his[4];
  uint32_t lo = 0x7F7FFFFFu, hi = 0u;
  auto take = [&](float v) {
    const uint32_t a = __builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu;
    hi = a > hi ? a : hi;
    lo = a != 0u && a < lo ? a : lo;
  };
  // 16-byte loads, a grid-stride walk (the planes come from hipMalloc: 256-byte aligned); the tail element by element
  const size_t quads = count / 4;
  const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(x);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) {
    const float4 v = x4[i];
    take(v.x);
    take(v.y);
    take(v.z);
    take(v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) take(x[quads * 4 + threadIdx.x]);
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const uint32_t ol = (uint32_t)__shfl_xor((int)lo, m, 64), oh = (uint32_t)__shfl_xor((int)hi, m, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  // one pair of atomics per workgroup (thousands of same-address atomics resolve one after the other in the L2)
  if ((threadIdx.x & 63) == 0) {
    los[threadIdx.x >> 6] = lo;
    his[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      lo = los[w] < lo ? los[w] : lo;
      hi = his[w] > hi ? his[w] : hi;
    }
    bt::range_fold(range, blockIdx.x, lo, hi);
  }
}
__global__ void k_obs_range_reset(uint32_t *range) { bt::range_reset(range, (int)threadIdx.x, 64); }

void launch_obs_range(rl_traj *traj) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  const uint64_t D = traj->d.D > 5 ? traj->d.D : 5;
  const size_t count = (size_t)D * (traj->d.T + 1) * traj->d.n;
  size_t blocks = (count / 4 + 256 * 4 - 1) / (256 * 4);  // >= four 16-byte loads per thread
  const size_t cap = 4 * (size_t)traj->eng->prop.multiProcessorCount;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_obs_range_reset, dim3(1), dim3(64), 0, traj->eng->stream, traj->d.range);
  hipLaunchKernelGGL(k_obs_range, dim3((unsigned)blocks), dim3(256), 0, traj->eng->stream, traj->d.obs, count,
                     traj->d.range);
}

void launch_env_reset(rl_env *env) {
  if (env->kind != RL_ENV_CARTPOLE) return launch_chain_reset(env);
  ProfScope ps(env->eng, RL_K_SMALL);
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  hipLaunchKernelGGL(k_env_reset, dim3(cdiv(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n);
}

void launch_env_observe(rl_env *env, float *d_obs) {
  if (env->kind != RL_ENV_CARTPOLE) return launch_chain_observe(env, d_obs);
  ProfScope ps(env->eng, RL_K_SMALL);
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  if (env->D == 5)
    hipLaunchKernelGGL(k_env_observe<5>, dim3(cdiv(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n,
                       d_obs);
  else
    hipLaunchKernelGGL(k_env_observe<4>, dim3(cdiv(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n,
                       d_obs);
}

void launch_env_step(rl_env *env) {
  if (env->kind != RL_ENV_CARTPOLE) return launch_chain_step(env);
  ProfScope ps(env->eng, RL_K_ENV_STEP);
  uint32_t n = (uint32_t)env->cfg.n_lanes;
#define ENV_STEP(DD, FF)                                                                                              \
  hipLaunchKernelGGL((k_env_step<DD, FF>), dim3(cdiv(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n, \
                     env->d_actions, env->d_reward, env->d_flag, env->d_obs, env->d_term_obs)
  const bool filled = n >= (1u << 20);
  if (env->D == 5) {
    if (filled) ENV_STEP(5, true);
    else ENV_STEP(5, false);
  } else {
    if (filled) ENV_STEP(4, true);
    else ENV_STEP(4, false);
  }
#undef ENV_STEP
}

template <int D, int G>
static void launch_rollout_g(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  constexpr int BLOCK = 64;  // one wave per workgroup: the workgroups spread over all CUs at every lane count
  const uint32_t n = (uint32_t)env->cfg.n_lanes;
  hipLaunchKernelGGL((k_rollout_cartpole<D, BLOCK, G>), dim3(cdiv((size_t)n * G, BLOCK)), dim3(BLOCK), 0,
                     env->eng->stream, env->dev, env->st, traj->d, policy->d_params, (int)policy->hidden, env->t_global);
  traj->range_reset = true;  // (the kernel has reset the range words for the value forward that follows)
}

void launch_rollout(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  // threads per lane: up to two waves per SIMD for groups of at most four threads, one wave per SIMD for wider groups
  // (the launch is bound by the latency of a step; the wider the group, the more of a step every thread repeats).
  // Measured, 128 steps, ms for G = 1 / 2 / 4 / 8 / 16 (profiles/r02_rollout_group_sweep.txt): 65,536 lanes 0.78 /
  // 0.71 / 0.84 / 1.25 / 2.21; 32,768: 0.82 / 0.56 / 0.54 / 0.69 / 1.06; 16,384: 0.76 / 0.57 / 0.43 / 0.44 / 0.60;
  // 8,192: 0.75 / 0.53 / 0.44 / 0.35 / 0.38; 4,096: 0.75 / 0.53 / 0.41 / 0.36 / 0.30
  const uint64_t simds = 4ull * (uint64_t)env->eng->prop.multiProcessorCount, n = env->cfg.n_lanes;
  int G = 1;
  while (G < 16 && n * (uint64_t)(2 * G) <= (2 * G >= 8 ? 1 : 2) * simds * 64) G *= 2;
  if (const char *o = std::getenv("RELEARN_ROLLOUT_G")) G = std::atoi(o);  // measurement override
#define ROLL(DD)                                                   \
  switch (G) {                                                     \
    case 16: launch_rollout_g<DD, 16>(env, policy, traj); break;   \
    case 8: launch_rollout_g<DD, 8>(env, policy, traj); break;     \
    case 4: launch_rollout_g<DD, 4>(env, policy, traj); break;     \
    case 2: launch_rollout_g<DD, 2>(env, policy, traj); break;     \
    default: launch_rollout_g<DD, 1>(env, policy, traj); break;    \
  }
  if (env->D == 5) {
    ROLL(5)
  } else {
    ROLL(4)
  }
#undef ROLL
}

void launch_values(rl_traj *traj, const rl_mlp *critic) {
  ProfScope ps(traj->eng, RL_K_VALUES);
  size_t rows = (size_t)(traj->d.T + 1) * traj->d.n;
  if (traj->d.D == 5 && traj->range_reset && !traj->range_valid && !traj->range_fixed) {
    // the planes' magnitude range as a by-product of the pass that reads every observation anyway (no pass of its own)
    hipLaunchKernelGGL((k_mlp_forward_rows<5, 1, true>), dim3(cdiv(rows, 256)), dim3(256), 0, traj->eng->stream,
                       critic->d_params, (int)critic->hidden, traj->d.obs, rows, rows, traj->d.values, rows, traj->d.range);
    traj->range_valid = true;
    traj->range_reset = false;
  } else if (traj->d.D == 5)
    hipLaunchKernelGGL((k_mlp_forward_rows<5, 1>), dim3(cdiv(rows, 256)), dim3(256), 0, traj->eng->stream,
                       critic->d_params, (int)critic->hidden, traj->d.obs, rows, rows, traj->d.values, rows, (uint32_t *)nullptr);
  else
    hipLaunchKernelGGL((k_mlp_forward_rows<4, 1>), dim3(cdiv(rows, 256)), dim3(256), 0, traj->eng->stream,
                       critic->d_params, (int)critic->hidden, traj->d.obs, rows, rows, traj->d.values, rows, (uint32_t *)nullptr);
}

void launch_gae(rl_traj *traj, const rl_mlp *critic, float gamma, float lambda) {
  ProfScope ps(traj->eng, RL_K_GAE);
  uint32_t n = traj->d.n;
  traj->rtg_scan_valid = true;  // d.rtg <- the lane scan's reward-to-go at `gamma` (engine.hpp)
  traj->rtg_gamma = gamma;
  if (traj->d.D == 5)
    hipLaunchKernelGGL(k_gae_scan<5>, dim3(cdiv(n, 64)), dim3(64), 0, traj->eng->stream, traj->d,
                       critic ? critic->d_params : (const float *)nullptr, critic ? (int)critic->hidden : 0, gamma, lambda,
                       critic ? 0 : 1);
  else
    hipLaunchKernelGGL(k_gae_scan<4>, dim3(cdiv(n, 64)), dim3(64), 0, traj->eng->stream, traj->d,
                       critic ? critic->d_params : (const float *)nullptr, critic ? (int)critic->hidden : 0, gamma, lambda,
                       critic ? 0 : 1);
}

void launch_value_targets(rl_traj *traj, const rl_mlp *critic, float gamma) {
  ProfScope ps(traj->eng, RL_K_GAE);
  if (critic == nullptr) {
    hipLaunchKernelGGL(k_value_targets_rtg, dim3(cdiv(traj->d.n, 64)), dim3(64), 0, traj->eng->stream, traj->d, gamma);
    return;
  }
  const size_t B = (size_t)traj->d.T * traj->d.n;
  if (traj->d.D == 5)
    hipLaunchKernelGGL(k_value_targets_td<5>, dim3(cdiv(B, 256)), dim3(256), 0, traj->eng->stream, traj->d,
                       critic->d_params, (int)critic->hidden, gamma);
  else
    hipLaunchKernelGGL(k_value_targets_td<4>, dim3(cdiv(B, 256)), dim3(256), 0, traj->eng->stream, traj->d,
                       critic->d_params, (int)critic->hidden, gamma);
}

void launch_mlp_forward_host_rows(rl_mlp *mlp, const float *d_in_soa, size_t rows, float *d_out_soa) {
  ProfScope ps(mlp->eng, RL_K_VALUES);
  dim3 g(cdiv(rows, 256)), b(256);
  hipStream_t s = mlp->eng->stream;
  int H = (int)mlp->hidden;
#define FWD(DD, AA)                                                                                              \
  hipLaunchKernelGGL((k_mlp_forward_rows<DD, AA, false>), g, b, 0, s, mlp->d_params, H, d_in_soa, rows, rows, d_out_soa, \
                     rows)
  if (mlp->in_dim == 5 && mlp->out_dim == 2) FWD(5, 2);
  else if (mlp->in_dim == 5 && mlp->out_dim == 1) FWD(5, 1);
  else if (mlp->in_dim == 4 && mlp->out_dim == 2) FWD(4, 2);
  else if (mlp->in_dim == 4 && mlp->out_dim == 1) FWD(4, 1);
  else throw RlError(RL_ERR_UNSUPPORTED, "mlp forward: unsupported (in_dim, out_dim)");
#undef FWD
}
