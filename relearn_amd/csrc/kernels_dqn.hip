// kernels_dqn.hip — DQN data path: epsilon-greedy collection straight into the HBM replay ring, and the minibatch
// builder (gather of the sampled episodes + value targets).
//
// Reference: DqnActor::act (src/torch/agents/dqn.rs:360-379), ReplayBuffer::write_step (src/agents/buffers/
// replay.rs:89-115), DqnAgent::batch_update_slice_refs sample_minibatch (dqn.rs:280-314), StepValueTarget
// (src/torch/agents/critics/mod.rs:203-229).
#include "device_fns.hpp"
#include "kernels.hpp"
#include "replay.hpp"

struct DevEpEnds {
  uint32_t *base;  // [E][N]
  uint32_t N, lane;
  __device__ uint32_t get(uint32_t i) const { return base[(size_t)i * N + lane]; }
  __device__ void set(uint32_t i, uint32_t v) { base[(size_t)i * N + lane] = v; }
};

// Sequential per-lane actor generator: ChaCha8(seed_actor), stream = global lane id, word position `pos` kept in
// HBM between launches (it is the `rng_actor: Prng` of Steps, src/simulation/steps.rs:15-28).  The current
// 16-word block is parked in a lane-private LDS column.
template <int BLOCK>
struct LaneActorRng {
  uint32_t *col;  // &lds[threadIdx.x], stride BLOCK
  const uint32_t *key;
  uint64_t lane, pos, cur_block;
  __device__ uint32_t next_u32() {
    const uint64_t blk = pos >> 4;
    if (blk != cur_block) {
      uint32_t w[16];
      rl_chacha_block(key, blk, lane, 4, w);
#pragma unroll
      for (int k = 0; k < 16; ++k) col[k * BLOCK] = w[k];
      cur_block = blk;
    }
    const uint32_t v = col[(uint32_t)(pos & 15) * BLOCK];
    pos += 1;
    return v;
  }
  // BlockRng::next_u64: two consecutive words, low first
  __device__ uint64_t next_u64() {
    const uint64_t lo = next_u32();
    const uint64_t hi = next_u32();
    return (hi << 32) | lo;
  }
};

// T env-actor steps per lane with the DQN actor:
//   if rng.gen_bool(eps) { action_space.sample(rng) = gen_range(0..2) } else { argmax_a Q(obs)[a] }
// every step is appended to the lane's replay ring.
template <int D, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_rollout_cartpole_dqn(CartPoleDev c, EnvStateDev st, ReplayDev rp,
                                                                const float *__restrict__ qnet, int H, uint32_t T,
                                                                uint64_t p_int, int always_explore,
                                                                uint8_t *__restrict__ flags_out) {
  __shared__ uint32_t words[16 * BLOCK];
  __shared__ __attribute__((aligned(16))) float pk[8 * 128 + 4];  // the Q-network, one 8-float record per hidden unit
  const uint32_t n = rp.N;
  mlp_pack_lds<D>(pk, qnet, H, threadIdx.x, BLOCK);
  __syncthreads();
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint64_t lane = c.lane_offset + i;
  LaneState s;
  lane_load(st, i, s);
  LaneRing ring{rp.head[i], rp.count[i], rp.ep_head[i], rp.ep_count[i], rp.total[i]};
  DevEpEnds eps{rp.ep_end, n, i};
  LaneActorRng<BLOCK> rng{&words[threadIdx.x], c.key_actor, lane, rp.actor_pos[i], ~0ull};
  const size_t plane = (size_t)rp.C * n;
  for (uint32_t t = 0; t < T; ++t) {
    float f[D];
    cp_features<D>(c, s, f);
    int a;
    bool explore = always_explore != 0;
    if (!explore) explore = rng.next_u64() < p_int;  // Bernoulli::sample: v < (p * 2^64) as u64
    if (explore) {
      // UniformInt::sample_single(0, 2): widening multiply by 2, zone = (2 << 62) - 1
      for (;;) {
        const uint64_t v = rng.next_u64();
        const uint64_t lo = v << 1, hi = v >> 63;
        if (lo <= 0x7fffffffffffffffull) {
          a = (int)hi;
          break;
        }
      }
    } else {
      float z[2];
      mlp_forward_lane_lds<D>(pk, H, f, z);
      a = z[1] > z[0] ? 1 : 0;  // argmax: first maximal index
    }
    int succ = cp_step(c, s, a);
    // the engine's horizon rule: a lane still mid-episode at the end of the launch closes its
    // episode as Interrupt(successor observation) and carries the env state on into the next collection
    const bool horizon_cut = succ == RL_SUCC_CONTINUE && t + 1 == T;
    const int succ_rec = horizon_cut ? RL_SUCC_INTERRUPT : succ;
    const uint32_t slot_abs = ring_write_step(ring, rp.C, rp.E, eps, succ_rec != RL_SUCC_CONTINUE);
    if (slot_abs == 0xffffffffu) {
      *rp.error = 1;  // WriteExperienceError::Full: a single episode longer than the lane's capacity
      break;
    }
    const size_t o = (size_t)(slot_abs % rp.C) * n + i;
#pragma unroll
    for (int d = 0; d < D; ++d) rp.obs[d * plane + o] = f[d];
    rp.action[o] = (uint8_t)a;
    rp.reward[o] = 1.0f;  // CartPole::step reward (cartpole.rs:140)
    rp.flag[o] = (uint8_t)succ_rec;
    if (succ_rec == RL_SUCC_INTERRUPT) {
      cp_features<D>(c, s, f);
#pragma unroll
      for (int d = 0; d < D; ++d) rp.next_obs[d * plane + o] = f[d];
    }
    flags_out[(size_t)t * n + i] = (uint8_t)succ_rec;
    if (succ != RL_SUCC_CONTINUE) cp_reset(c, s, lane);
  }
  rp.head[i] = ring.head;
  rp.count[i] = ring.count;
  rp.ep_head[i] = ring.ep_head;
  rp.ep_count[i] = ring.ep_count;
  rp.total[i] = ring.total;
  rp.actor_pos[i] = rng.pos;
  lane_store(st, i, s);
}

// Minibatch builder: one workgroup per sampled episode.  Gathers the episode's steps from the ring into the compact
// sample arrays (obs plane stride `out_plane`) and computes the value targets:
//   RewardToGo: G_t = r_t + gamma * G_{t+1} (f32 multiply, then add — packed.rs:312-342 arithmetic)
//   OneStepTd : r_t + gamma * max_a Q(s_{t+1}); 0 beyond a Terminate, Q(interrupt successor) after an Interrupt
template <int D>
__global__ void __launch_bounds__(64) k_dqn_build_minibatch(ReplayDev rp, const uint32_t *__restrict__ ep_lane,
                                                            const uint32_t *__restrict__ ep_start,
                                                            const uint32_t *__restrict__ ep_len,
                                                            const uint32_t *__restrict__ ep_offset,
                                                            float *__restrict__ out_obs, size_t out_plane,
                                                            uint8_t *__restrict__ out_action,
                                                            float *__restrict__ out_target, float gamma,
                                                            int one_step_td, const float *__restrict__ qnet, int H) {
  __shared__ float rew[1024];
  __shared__ float carry;
  const uint32_t e = blockIdx.x;
  const uint32_t lane = ep_lane[e], start = ep_start[e], len = ep_len[e], off = ep_offset[e];
  const uint32_t n = rp.N;
  const size_t plane = (size_t)rp.C * n;
  for (uint32_t i = threadIdx.x; i < len; i += 64) {
    const size_t o = (size_t)((start + i) % rp.C) * n + lane;
#pragma unroll
    for (int d = 0; d < D; ++d) out_obs[d * out_plane + off + i] = rp.obs[d * plane + o];
    out_action[off + i] = rp.action[o];
    if (one_step_td) {
      const uint8_t fl = rp.flag[o];
      float vnext = 0.0f;
      if (fl != RL_SUCC_TERMINATE) {
        float x[D], z[2];
        if (fl == RL_SUCC_INTERRUPT || i + 1 == len) {
#pragma unroll
          for (int d = 0; d < D; ++d) x[d] = rp.next_obs[d * plane + o];
        } else {
          const size_t o1 = (size_t)((start + i + 1) % rp.C) * n + lane;
#pragma unroll
          for (int d = 0; d < D; ++d) x[d] = rp.obs[d * plane + o1];
        }
        mlp_forward_lane<D, 2>(qnet, H, x, z);
        vnext = z[1] > z[0] ? z[1] : z[0];  // amax(-1)
      }
      const float dn = gamma * vnext;
      out_target[off + i] = rp.reward[o] + dn;
    }
  }
  if (one_step_td) return;
  // reward-to-go: backwards in chunks of 1024 steps staged through LDS, scanned by one lane
  if (threadIdx.x == 0) carry = 0.0f;
  __syncthreads();
  for (uint32_t hi = len; hi > 0;) {
    const uint32_t lo = hi > 1024 ? hi - 1024 : 0, cnt = hi - lo;
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) rew[i] = rp.reward[(size_t)((start + lo + i) % rp.C) * n + lane];
    __syncthreads();
    if (threadIdx.x == 0) {
      float g = carry;
      bool first = hi == len;
      for (uint32_t i = cnt; i-- > 0;) {
        if (first) {
          g = rew[i];
          first = false;
        } else {
          float p = g * gamma;
          g = rew[i] + p;
        }
        rew[i] = g;
      }
      carry = g;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) out_target[off + lo + i] = rew[i];
    __syncthreads();
    hi = lo;
  }
}

// ---------------------------------------------------------------- minibatch episode sampling
// sample_minibatch of DqnAgent::batch_update_slice_refs (dqn.rs:280-291):
//   iter::repeat(buffers).flatten().map(|buf| buf.episodes().get(Uniform::new(0, buf.num_episodes()).sample(rng)))
//       .take_while(|ep| { let take = total < minibatch_steps; total += ep.len(); take })
// The draws are sequential in the agent's Prng, but draw j is `next_u64` number j of the stream unless an earlier
// draw was rejected by UniformInt's widening-multiply test (probability < num_episodes / 2^64).  So candidate j is
// evaluated by thread j of a 1024-wide chunk from stream words [pos + 2j, pos + 2j + 2), episode lengths are
// prefix-summed, and the take_while cut is the first candidate whose running total reaches minibatch_steps.  If any
// candidate of a chunk is rejected, one thread replays that chunk onwards with the plain sequential algorithm.
// take_while evaluates (and consumes the draw of) the first episode it refuses: pos advances past it.
struct EpisodePick {
  uint32_t start, len;
  bool rejected, empty;
};

__device__ inline uint64_t agent_u64(const AgentKey &key, uint64_t word_pos) {
  uint32_t w[16];
  rl_chacha_block(key.w, word_pos >> 4, 0, 4, w);
  const uint32_t i = (uint32_t)(word_pos & 15);
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k == (int)i) lo = w[k];
    if (k == (int)i + 1) hi = w[k];
  }
  if (i == 15) {  // the pair straddles two blocks
    rl_chacha_block(key.w, (word_pos >> 4) + 1, 0, 4, w);
    hi = w[0];
  }
  return ((uint64_t)hi << 32) | lo;
}

// Uniform::new(0usize, n).sample(rng) for the draw `v`, then Episodes::get (replay.rs:154-165)
__device__ inline EpisodePick pick_episode(const ReplayDev &rp, uint32_t lane, uint64_t v) {
  EpisodePick p{0, 0, false, false};
  const uint64_t n = rp.ep_count[lane];
  if (n == 0) {
    p.empty = true;  // Uniform::new(0, 0) panics in the reference
    return p;
  }
  const uint64_t ints_to_reject = (0ull - n) % n;  // (u64::MAX - range + 1) % range
  const uint64_t zone = ~0ull - ints_to_reject;
  const uint64_t lo = v * n, hi = __umul64hi(v, n);
  if (lo > zone) {
    p.rejected = true;
    return p;
  }
  const uint32_t idx = (uint32_t)hi, eh = rp.ep_head[lane];
  const uint32_t end = rp.ep_end[(size_t)((eh + idx) % rp.E) * rp.N + lane];
  const uint32_t start = idx == 0 ? rp.head[lane] : rp.ep_end[(size_t)((eh + idx - 1) % rp.E) * rp.N + lane];
  p.start = start;
  p.len = end - start;
  if (p.len == 0 || p.len > rp.C) p.empty = true;  // corrupt bookkeeping: never loop on it
  return p;
}

constexpr int SAMPLE_BLOCK = 1024;

__global__ void __launch_bounds__(SAMPLE_BLOCK) k_dqn_sample(ReplayDev rp, AgentKey key, uint64_t *agent_pos,
                                                             uint32_t minibatch_steps, uint32_t max_eps,
                                                             uint32_t *__restrict__ ep_lane,
                                                             uint32_t *__restrict__ ep_start,
                                                             uint32_t *__restrict__ ep_len,
                                                             uint32_t *__restrict__ ep_off, DqnCountsDev *counts,
                                                             int sequential, uint32_t n_batches) {
  __shared__ uint32_t scan[SAMPLE_BLOCK];
  __shared__ int s_flag[2];  // [0] rejected draw in this chunk, [1] empty buffer
  const uint32_t tid = threadIdx.x;
  // `n_batches` consecutive minibatches in one launch (the draws of minibatch b + 1 continue where b stopped);
  // minibatch b writes its lists at offset b * max_eps and its counts at counts[b]
  for (uint32_t batch = 0; batch < n_batches; ++batch, ep_lane += max_eps, ep_start += max_eps, ep_len += max_eps,
                ep_off += max_eps, ++counts) {
  __syncthreads();  // the previous minibatch's final position is visible
  const uint64_t pos0 = *(volatile uint64_t *)agent_pos;
  uint32_t total = 0;   // steps taken so far
  uint32_t n_eps = 0;   // episodes taken so far
  uint64_t draws = 0;   // u64 draws consumed so far
  bool done = false, fallback = sequential != 0;
  int err = 0;
  while (!done && !fallback) {
    if (tid < 2) s_flag[tid] = 0;
    __syncthreads();
    const uint64_t j = draws + tid;
    const uint32_t lane = (uint32_t)(j % rp.N);
    const EpisodePick p = pick_episode(rp, lane, agent_u64(key, pos0 + 2 * j));
    if (p.rejected) s_flag[0] = 1;
    if (p.empty) s_flag[1] = 1;
    scan[tid] = p.len;
    __syncthreads();
    if (s_flag[1]) {
      err = 2;
      break;
    }
    if (s_flag[0]) {
      fallback = true;
      break;
    }
    // inclusive prefix sum of the chunk's episode lengths
    for (int d = 1; d < SAMPLE_BLOCK; d <<= 1) {
      const uint32_t add = tid >= (uint32_t)d ? scan[tid - d] : 0;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    const uint32_t before = total + scan[tid] - p.len;
    const bool take = before < minibatch_steps;
    if (take && n_eps + tid < max_eps) {
      ep_lane[n_eps + tid] = lane;
      ep_start[n_eps + tid] = p.start;
      ep_len[n_eps + tid] = p.len;
      ep_off[n_eps + tid] = before;
    }
    const uint32_t n_take = __syncthreads_count(take);
    if (n_take < SAMPLE_BLOCK) {
      const uint32_t taken_steps = n_take == 0 ? 0 : scan[n_take - 1];
      total += taken_steps;
      n_eps += n_take;
      draws += n_take + 1;  // plus the refused candidate
      done = true;
    } else {
      total += scan[SAMPLE_BLOCK - 1];
      n_eps += SAMPLE_BLOCK;
      draws += SAMPLE_BLOCK;
    }
    __syncthreads();
  }
  if (fallback && tid == 0) {
    // plain sequential restatement from candidate `draws` on (pos counts words: 2 per draw, more after rejections)
    uint64_t pos = pos0 + 2 * draws;
    uint64_t cand = draws;
    for (;;) {
      const uint32_t lane = (uint32_t)(cand % rp.N);
      EpisodePick p;
      for (;;) {
        p = pick_episode(rp, lane, agent_u64(key, pos));
        pos += 2;
        if (!p.rejected) break;
      }
      if (p.empty) {
        err = 2;
        break;
      }
      const bool take = total < minibatch_steps;
      if (!take) break;
      if (n_eps < max_eps) {
        ep_lane[n_eps] = lane;
        ep_start[n_eps] = p.start;
        ep_len[n_eps] = p.len;
        ep_off[n_eps] = total;
      }
      total += p.len;
      n_eps += 1;
      cand += 1;
    }
    *agent_pos = pos;
  } else if (tid == 0 && err == 0) {
    *agent_pos = pos0 + 2 * draws;
  }
  if (tid == 0) {
    counts->n_eps = n_eps;
    counts->n_steps = total;
    counts->error = err != 0 ? err : (*rp.error != 0 ? 1 : 0);
    counts->pad = 0;
    __threadfence_block();
  }
  }
}

void launch_dqn_sample(rl_engine *eng, const ReplayDev &rp, const AgentKey &key, uint64_t *d_agent_pos,
                       uint32_t minibatch_steps, uint32_t max_eps, uint32_t *d_lane, uint32_t *d_start,
                       uint32_t *d_len, uint32_t *d_off, DqnCountsDev *d_counts, int sequential, uint32_t n_batches) {
  ProfScope ps(eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_dqn_sample, dim3(1), dim3(SAMPLE_BLOCK), 0, eng->stream, rp, key, d_agent_pos,
                     minibatch_steps, max_eps, d_lane, d_start, d_len, d_off, d_counts, sequential, n_batches);
}

static inline uint32_t cdiv_d(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

void launch_rollout_dqn(rl_env *env, const rl_mlp *qnet, const ReplayDev &rp, uint32_t T, uint64_t p_int,
                        int always_explore, uint8_t *d_flags) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  constexpr int BLOCK = 64;
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  if (env->D == 5)
    hipLaunchKernelGGL((k_rollout_cartpole_dqn<5, BLOCK>), dim3(cdiv_d(n, BLOCK)), dim3(BLOCK), 0, env->eng->stream,
                       env->dev, env->st, rp, qnet->d_params, (int)qnet->hidden, T, p_int, always_explore, d_flags);
  else
    hipLaunchKernelGGL((k_rollout_cartpole_dqn<4, BLOCK>), dim3(cdiv_d(n, BLOCK)), dim3(BLOCK), 0, env->eng->stream,
                       env->dev, env->st, rp, qnet->d_params, (int)qnet->hidden, T, p_int, always_explore, d_flags);
}

void launch_dqn_build_minibatch(rl_engine *eng, const ReplayDev &rp, uint32_t n_eps, const uint32_t *d_lane,
                                const uint32_t *d_start, const uint32_t *d_len, const uint32_t *d_off,
                                float *d_obs, size_t out_plane, uint8_t *d_action, float *d_target, float gamma,
                                int one_step_td, const rl_mlp *qnet) {
  ProfScope ps(eng, RL_K_VALUES);
  if (n_eps == 0) return;
  if (rp.D == 5)
    hipLaunchKernelGGL(k_dqn_build_minibatch<5>, dim3(n_eps), dim3(64), 0, eng->stream, rp, d_lane, d_start, d_len,
                       d_off, d_obs, out_plane, d_action, d_target, gamma, one_step_td, qnet->d_params,
                       (int)qnet->hidden);
  else
    hipLaunchKernelGGL(k_dqn_build_minibatch<4>, dim3(n_eps), dim3(64), 0, eng->stream, rp, d_lane, d_start, d_len,
                       d_off, d_obs, out_plane, d_action, d_target, gamma, one_step_td, qnet->d_params,
                       (int)qnet->hidden);
}
