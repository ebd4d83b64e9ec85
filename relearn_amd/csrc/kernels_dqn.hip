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
  __shared__ __attribute__((aligned(16))) float pk[MLP_PK_FLOATS];  // the Q-network, one 8-float record per hidden unit
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
// the fields of a lane's ring the pick needs (constant during an update: a workgroup may cache them)
struct LaneMeta {
  uint32_t n_eps, eh, head;  // stored episodes; ring index of the oldest (reduced mod E); first stored step
};
__device__ inline LaneMeta lane_meta(const ReplayDev &rp, uint32_t lane) {
  return LaneMeta{rp.ep_count[lane], rp.ep_head[lane] % rp.E, rp.head[lane]};  // ep_head counts up without wrapping
}

__device__ inline EpisodePick pick_episode(const ReplayDev &rp, uint32_t lane, const LaneMeta &lm, uint64_t v) {
  EpisodePick p{0, 0, false, false};
  const uint64_t n = lm.n_eps;
  if (n == 0) {
    p.empty = true;  // Uniform::new(0, 0) panics in the reference
    return p;
  }
  const uint64_t lo = v * n, hi = __umul64hi(v, n);
  if (lo > ~0ull - n) {  // ints_to_reject < n: only then can the draw fall outside the zone (the 64-bit division
                         // stays off the common path)
    const uint64_t ints_to_reject = (0ull - n) % n;  // (u64::MAX - range + 1) % range
    const uint64_t zone = ~0ull - ints_to_reject;
    if (lo > zone) {
      p.rejected = true;
      return p;
    }
  }
  const uint32_t idx = (uint32_t)hi, eh = lm.eh;
  // ring index (eh + idx) mod E: eh < E and idx < n <= E, one conditional subtraction
  const uint32_t r1 = eh + idx, r0 = r1 - 1;
  const uint32_t s1 = r1 >= rp.E ? r1 - rp.E : r1, s0 = (idx != 0 && r0 >= rp.E) ? r0 - rp.E : r0;
  const uint32_t head = lm.head;
  const uint32_t end = rp.ep_end[(size_t)s1 * rp.N + lane];
  const uint32_t prev = rp.ep_end[(size_t)(idx != 0 ? s0 : s1) * rp.N + lane];  // unconditional: both loads in flight
  const uint32_t start = idx == 0 ? head : prev;
  p.start = start;
  p.len = end - start;
  if (p.len == 0 || p.len > rp.C) p.empty = true;  // corrupt bookkeeping: never loop on it
  return p;
}

constexpr int SAMPLE_BLOCK = 1024;
constexpr int SAMPLE_CPT = 5;                            // candidates per thread and pass (consecutive)
constexpr int SAMPLE_CHUNK = SAMPLE_BLOCK * SAMPLE_CPT;  // 5,120: a 100 k-step minibatch of CartPole episodes is one pass
constexpr int SAMPLE_BLOCKS = SAMPLE_CHUNK / 8 + 1;      // ChaCha blocks a pass can touch (8 draws of 2 words per block)

// inclusive prefix sum of one value per thread over the workgroup (wave shuffles, then the 16 wave totals)
__device__ inline uint32_t block_inclusive_scan(uint32_t v, uint32_t *wave_tot, uint32_t tid) {
  const uint32_t lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= (uint32_t)d) v += up;
  }
  if (lane == 63) wave_tot[wave] = v;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < SAMPLE_BLOCK / 64; ++w) base += w < (int)wave ? wave_tot[w] : 0u;
  __syncthreads();  // wave_tot is reused by the next scan
  return base + v;
}

constexpr int SAMPLE_META_LANES = 4096;  // up to this many lanes the ring fields are cached in LDS (48 KB)

template <bool META_LDS>
__global__ void __launch_bounds__(SAMPLE_BLOCK) k_dqn_sample(ReplayDev rp, AgentKey key, uint64_t *agent_pos,
                                                             uint32_t minibatch_steps, uint32_t max_eps,
                                                             uint32_t *__restrict__ ep_lane,
                                                             uint32_t *__restrict__ ep_start,
                                                             uint32_t *__restrict__ ep_len,
                                                             uint32_t *__restrict__ ep_off, DqnCountsDev *counts,
                                                             int sequential, uint32_t n_batches) {
  __shared__ uint32_t words[SAMPLE_BLOCKS * 16];  // the pass's stretch of the agent's stream, one ChaCha block per thread
  __shared__ uint32_t wave_tot[SAMPLE_BLOCK / 64];
  __shared__ int s_flag[2];  // [0] rejected draw in this pass, [1] empty buffer
  __shared__ uint32_t s_last[2];  // episodes and steps the pass takes
  __shared__ uint64_t s_pos;      // stream position after a sequential replay (known to thread 0 only)
  __shared__ uint32_t meta[META_LDS ? 3 * SAMPLE_META_LANES : 3];
  const uint32_t tid = threadIdx.x;
  if (META_LDS)
    for (uint32_t l = tid; l < rp.N; l += SAMPLE_BLOCK) {
      const LaneMeta lm = lane_meta(rp, l);
      meta[3 * l] = lm.n_eps;
      meta[3 * l + 1] = lm.eh;
      meta[3 * l + 2] = lm.head;
    }
  auto meta_of = [&](uint32_t l) {
    return META_LDS ? LaneMeta{meta[3 * l], meta[3 * l + 1], meta[3 * l + 2]} : lane_meta(rp, l);
  };
  const int rp_error = *rp.error != 0 ? 1 : 0;
  uint64_t pos_now = *agent_pos;  // carried in registers from minibatch to minibatch
  // `n_batches` consecutive minibatches in one launch (the draws of minibatch b + 1 continue where b stopped);
  // minibatch b writes its lists at offset b * max_eps and its counts at counts[b]
  for (uint32_t batch = 0; batch < n_batches; ++batch, ep_lane += max_eps, ep_start += max_eps, ep_len += max_eps,
                ep_off += max_eps, ++counts) {
  __syncthreads();  // (first minibatch: the cached ring fields are in place)
  const uint64_t pos0 = pos_now;
  uint32_t total = 0;   // steps taken so far
  uint32_t n_eps = 0;   // episodes taken so far
  uint64_t draws = 0;   // u64 draws consumed so far
  bool done = false, fallback = sequential != 0;
  int err = 0;
  while (!done && !fallback) {
    if (tid < 2) s_flag[tid] = 0;
    // the stream words of candidates [draws, draws + CHUNK): word positions [first, first + 2 CHUNK)
    const uint64_t first = pos0 + 2 * draws, block0 = first >> 4;
    const uint32_t skew = (uint32_t)(first & 15);
    const uint32_t lane0 = (uint32_t)(draws % rp.N);
    if (tid < (uint32_t)SAMPLE_BLOCKS) {
      uint32_t w[16];
      rl_chacha_block(key.w, block0 + tid, 0, 4, w);
#pragma unroll
      for (int k = 0; k < 16; ++k) words[tid * 16 + k] = w[k];
    }
    __syncthreads();
    EpisodePick p[SAMPLE_CPT];
    uint32_t lanes[SAMPLE_CPT], mine = 0;
    bool rej = false, empty = false;
#pragma unroll
    for (int c = 0; c < SAMPLE_CPT; ++c) {
      const uint32_t jl = tid * SAMPLE_CPT + c;
      lanes[c] = (lane0 + jl) % rp.N;  // = (draws + jl) mod N without a 64-bit division per candidate
      const uint32_t wi = skew + 2 * jl;
      p[c] = pick_episode(rp, lanes[c], meta_of(lanes[c]), ((uint64_t)words[wi + 1] << 32) | words[wi]);
      rej |= p[c].rejected;
      empty |= p[c].empty;
      mine += p[c].len;
    }
    if (rej) s_flag[0] = 1;
    if (empty) s_flag[1] = 1;
    const uint32_t incl = block_inclusive_scan(mine, wave_tot, tid);  // (its barriers publish s_flag)
    if (s_flag[1]) {
      err = 2;
      break;
    }
    if (s_flag[0]) {
      fallback = true;
      break;
    }
    uint32_t before = total + incl - mine, took = 0;
#pragma unroll
    for (int c = 0; c < SAMPLE_CPT; ++c) {
      const bool take = before < minibatch_steps;
      const uint32_t slot = n_eps + tid * SAMPLE_CPT + c;
      if (take && slot < max_eps) {
        ep_lane[slot] = lanes[c];
        ep_start[slot] = p[c].start;
        ep_len[slot] = p[c].len;
        ep_off[slot] = before;
      }
      took += take ? 1u : 0u;
      before += take ? p[c].len : 0u;
    }
    // lengths are >= 1, so the taken candidates are a prefix of the pass (candidate 0 always belongs to it: the loop
    // would have ended otherwise); the thread that holds the prefix's last candidate publishes its size and steps
    const bool next_takes = tid + 1 < (uint32_t)SAMPLE_BLOCK && total + incl < minibatch_steps;
    if (took > 0 && !(took == (uint32_t)SAMPLE_CPT && next_takes)) {
      s_last[0] = tid * SAMPLE_CPT + took;
      s_last[1] = before - total;
    }
    __syncthreads();
    const uint32_t taken = s_last[0];
    total += s_last[1];
    n_eps += taken;
    if (taken < (uint32_t)SAMPLE_CHUNK) {
      draws += taken + 1;  // plus the refused candidate
      done = true;
    } else {
      draws += SAMPLE_CHUNK;
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
        p = pick_episode(rp, lane, meta_of(lane), agent_u64(key, pos));
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
    s_pos = pos;
  }
  if (fallback) {  // (uniform)
    __syncthreads();
    pos_now = s_pos;  // thread 0's err / n_eps / total are the minibatch's; the others only need the position
  } else if (err == 0) {
    pos_now = pos0 + 2 * draws;
  }
  if (tid == 0) {
    counts->n_eps = n_eps;
    counts->n_steps = total;
    counts->error = err != 0 ? err : rp_error;
    counts->pad = 0;
  }
  }
  if (tid == 0) *agent_pos = pos_now;
}

void launch_dqn_sample(rl_engine *eng, const ReplayDev &rp, const AgentKey &key, uint64_t *d_agent_pos,
                       uint32_t minibatch_steps, uint32_t max_eps, uint32_t *d_lane, uint32_t *d_start,
                       uint32_t *d_len, uint32_t *d_off, DqnCountsDev *d_counts, int sequential, uint32_t n_batches) {
  ProfScope ps(eng, RL_K_SMALL);
  if (rp.N <= (uint32_t)SAMPLE_META_LANES)
    hipLaunchKernelGGL(k_dqn_sample<true>, dim3(1), dim3(SAMPLE_BLOCK), 0, eng->stream, rp, key, d_agent_pos,
                       minibatch_steps, max_eps, d_lane, d_start, d_len, d_off, d_counts, sequential, n_batches);
  else
    hipLaunchKernelGGL(k_dqn_sample<false>, dim3(1), dim3(SAMPLE_BLOCK), 0, eng->stream, rp, key, d_agent_pos,
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
