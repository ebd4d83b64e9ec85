// kernels_dqn.hip — DQN data path: epsilon-greedy collection straight into the HBM replay ring, and the minibatch
// builder (gather of the sampled episodes + value targets).
//
// Reference: DqnActor::act (src/torch/agents/dqn.rs:360-379), ReplayBuffer::write_step (src/agents/buffers/
// replay.rs:89-115), DqnAgent::batch_update_slice_refs sample_minibatch (dqn.rs:280-314), StepValueTarget
// (src/torch/agents/critics/mod.rs:203-229).
#include <memory>

#include "bf16_tile.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"
#include "replay.hpp"

// one step record: two 16-byte accesses
__device__ __forceinline__ ReplayRec rec_load(const ReplayRec *__restrict__ p) {
  const uint4 a = reinterpret_cast<const uint4 *>(p)[0], b = reinterpret_cast<const uint4 *>(p)[1];
  ReplayRec r;
  r.x[0] = __uint_as_float(a.x), r.x[1] = __uint_as_float(a.y), r.x[2] = __uint_as_float(a.z), r.x[3] = __uint_as_float(a.w);
  r.x[4] = __uint_as_float(b.x), r.reward = __uint_as_float(b.y), r.af = b.z, r.pad = b.w;
  return r;
}
__device__ __forceinline__ void rec_store(ReplayRec *__restrict__ p, const ReplayRec &r) {
  reinterpret_cast<uint4 *>(p)[0] = make_uint4(__float_as_uint(r.x[0]), __float_as_uint(r.x[1]), __float_as_uint(r.x[2]),
                                               __float_as_uint(r.x[3]));
  reinterpret_cast<uint4 *>(p)[1] = make_uint4(__float_as_uint(r.x[4]), __float_as_uint(r.reward), r.af, 0u);
}

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
// G threads per lane (G consecutive lanes of a wave) share the greedy branch's Q-network forward, as in the fused TRPO
// rollout (kernels_rollout.hip: a launch lasts T x the latency of one step, and most of a greedy step is the 128-unit
// forward); everything else — draws, physics, ring bookkeeping — is repeated by every thread of the group, thread 0
// stores.  The result does not depend on G (mlp_forward_group_lds).
template <int D, int BLOCK, int G>
__global__ void __launch_bounds__(BLOCK) k_rollout_cartpole_dqn(CartPoleDev c, EnvStateDev st, ReplayDev rp,
                                                                const float *__restrict__ qnet, int H, uint32_t T,
                                                                uint64_t p_int, int always_explore,
                                                                uint8_t *__restrict__ flags_out,
                                                                uint32_t *__restrict__ range) {
  __shared__ uint32_t words[16 * BLOCK];
  __shared__ __attribute__((aligned(16))) float pk[MLP_PK_FLOATS];  // the Q-network, one 8-float record per hidden unit
  // the magnitude range of every observation that enters the store (steps and Interrupt successors), for the fused
  // gradient kernel's range guard (bf16_tile.hpp): one fold per wave and launch into words that are never reset — the
  // range of everything ever collected bounds every minibatch drawn from the store
  uint32_t r_lo = 0x7F7FFFFFu, r_hi = 0u;
  const uint32_t n = rp.N;
  mlp_pack_lds<D>(pk, qnet, H, threadIdx.x, BLOCK);
  __syncthreads();
  const uint32_t i0 = (blockIdx.x * BLOCK + threadIdx.x) / G;
  const int g = threadIdx.x % G;
  // lanes past the end follow lane n - 1 without storing: the group shuffles need every thread of a group
  const bool live = i0 < n;
  const uint32_t i = live ? i0 : n - 1;
  const bool writer = live && g == 0;
  const uint64_t lane = c.lane_offset + i;
  LaneState s;
  lane_load(st, i, s);
  LaneRing ring{rp.head[i], rp.count[i], rp.ep_head[i], rp.ep_count[i], rp.total[i]};
  struct GroupEpEnds {  // every thread of the group reads the lane's episode table, thread 0 writes it
    uint32_t *base;
    uint32_t N, lane;
    bool writer;
    __device__ uint32_t get(uint32_t k) const { return base[(size_t)k * N + lane]; }
    __device__ void set(uint32_t k, uint32_t v) {
      if (writer) base[(size_t)k * N + lane] = v;
    }
  } eps{rp.ep_end, n, i, writer};
  LaneActorRng<BLOCK> rng{&words[threadIdx.x], c.key_actor, lane, rp.actor_pos[i], ~0ull};
  bool full = false;
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
      mlp_forward_group_lds<D, G>(pk, H, g, f, z);
      a = z[1] > z[0] ? 1 : 0;  // argmax: first maximal index
    }
    int succ = cp_step(c, s, a);
    // the engine's horizon rule: a lane still mid-episode at the end of the launch closes its
    // episode as Interrupt(successor observation) and carries the env state on into the next collection
    const bool horizon_cut = succ == RL_SUCC_CONTINUE && t + 1 == T;
    const int succ_rec = horizon_cut ? RL_SUCC_INTERRUPT : succ;
    const uint32_t slot_abs = ring_write_step(ring, rp.C, rp.E, eps, succ_rec != RL_SUCC_CONTINUE);
    if (slot_abs == 0xffffffffu) {
      full = true;  // WriteExperienceError::Full: a single episode longer than the lane's capacity
      break;
    }
    const size_t o = (size_t)i * rp.C + slot_abs % rp.C;
#pragma unroll
    for (int d = 0; d < D; ++d) bt::range_accumulate(f[d], r_lo, r_hi);
    if (writer) {
      ReplayRec rec;
#pragma unroll
      for (int d = 0; d < 5; ++d) rec.x[d] = d < D ? f[d < D ? d : 0] : 0.0f;
      rec.reward = 1.0f;  // CartPole::step reward (cartpole.rs:140)
      rec.af = (uint32_t)a | (uint32_t)succ_rec << 8;
      rec_store(rp.rec + o, rec);
    }
    if (succ_rec == RL_SUCC_INTERRUPT) {
      cp_features<D>(c, s, f);
#pragma unroll
      for (int d = 0; d < D; ++d) bt::range_accumulate(f[d], r_lo, r_hi);
      if (writer) {
#pragma unroll
        for (int d = 0; d < D; ++d) rp.next[o].x[d] = f[d];
      }
    }
    if (writer) flags_out[(size_t)t * n + i] = (uint8_t)succ_rec;
    if (succ != RL_SUCC_CONTINUE) cp_reset(c, s, lane);
  }
  if (range != nullptr) bt::range_fold_wave(range, blockIdx.x, r_lo, r_hi);  // (every thread of the wave is here)
  if (!writer) return;
  if (full) *rp.error = 1;
  rp.head[i] = ring.head;
  rp.count[i] = ring.count;
  rp.ep_head[i] = ring.ep_head;
  rp.ep_count[i] = ring.ep_count;
  rp.total[i] = ring.total;
  rp.actor_pos[i] = rng.pos;
  lane_store(st, i, s);
}

// ---------------------------------------------------------------- collection with an action-value module of any shape
// DqnConfig<MB> is generic over the module (src/torch/agents/dqn.rs:26-39).  A module the fused collection kernel is not
// built for (rl_mlp::general: several hidden layers, other activations, a wider layer) collects one launch sequence per
// step — observe, the module's layer kernels over all lanes (kernels_general.hip), then this kernel: DqnActor::act
// (dqn.rs:360-379) from the lane's actor stream with the greedy branch reading the module's outputs, the env step, the
// ring write — with the fused kernel's stream discipline, ring bookkeeping and horizon rule (above), one thread per lane.
template <int D, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_dqn_lane_step(CartPoleDev c, EnvStateDev st, ReplayDev rp,
                                                         const float *__restrict__ q_values /* [2][n] */, uint64_t p_int,
                                                         int always_explore, int last_step,
                                                         uint8_t *__restrict__ flags_row) {
  __shared__ uint32_t words[16 * BLOCK];
  const uint32_t n = rp.N;
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint64_t lane = c.lane_offset + i;
  LaneState s;
  lane_load(st, i, s);
  LaneRing ring{rp.head[i], rp.count[i], rp.ep_head[i], rp.ep_count[i], rp.total[i]};
  struct EpEnds {
    uint32_t *base;
    uint32_t N, lane;
    __device__ uint32_t get(uint32_t k) const { return base[(size_t)k * N + lane]; }
    __device__ void set(uint32_t k, uint32_t v) { base[(size_t)k * N + lane] = v; }
  } eps{rp.ep_end, n, i};
  LaneActorRng<BLOCK> rng{&words[threadIdx.x], c.key_actor, lane, rp.actor_pos[i], ~0ull};
  float f[D];
  cp_features<D>(c, s, f);
  int a;
  bool explore = always_explore != 0;
  if (!explore) explore = rng.next_u64() < p_int;  // Bernoulli::sample: v < (p * 2^64) as u64
  if (explore) {
    for (;;) {  // UniformInt::sample_single(0, 2)
      const uint64_t v = rng.next_u64();
      const uint64_t lo = v << 1, hi = v >> 63;
      if (lo <= 0x7fffffffffffffffull) {
        a = (int)hi;
        break;
      }
    }
  } else {
    a = q_values[n + i] > q_values[i] ? 1 : 0;  // argmax: first maximal index
  }
  int succ = cp_step(c, s, a);
  const bool horizon_cut = succ == RL_SUCC_CONTINUE && last_step != 0;
  const int succ_rec = horizon_cut ? RL_SUCC_INTERRUPT : succ;
  const uint32_t slot_abs = ring_write_step(ring, rp.C, rp.E, eps, succ_rec != RL_SUCC_CONTINUE);
  if (slot_abs == 0xffffffffu) {  // WriteExperienceError::Full (the lane stops here; the host raises the error)
    *rp.error = 1;
    return;
  }
  const size_t o = (size_t)i * rp.C + slot_abs % rp.C;
  ReplayRec rec;
#pragma unroll
  for (int d = 0; d < 5; ++d) rec.x[d] = d < D ? f[d < D ? d : 0] : 0.0f;
  rec.reward = 1.0f;
  rec.af = (uint32_t)a | (uint32_t)succ_rec << 8;
  rec_store(rp.rec + o, rec);
  if (succ_rec == RL_SUCC_INTERRUPT) {
    cp_features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) rp.next[o].x[d] = f[d];
  }
  flags_row[i] = (uint8_t)succ_rec;
  if (succ != RL_SUCC_CONTINUE) cp_reset(c, s, lane);
  rp.head[i] = ring.head;
  rp.count[i] = ring.count;
  rp.ep_head[i] = ring.ep_head;
  rp.ep_count[i] = ring.ep_count;
  rp.total[i] = ring.total;
  rp.actor_pos[i] = rng.pos;
  lane_store(st, i, s);
}

// one-step TD targets from the module's outputs at the successor observations (the gather left rewards where the
// targets go and the successor codes in `flag`): r + gamma * max_a Q(s'), 0 beyond a Terminate — the arithmetic of
// k_dqn_build_minibatch (amax, `scalar * tensor`, then `tensor + tensor`)
__global__ void k_dqn_td_targets(float *__restrict__ tgt, const uint8_t *__restrict__ flag,
                                 const float *__restrict__ q_next /* [2][n] */, uint32_t n, float gamma) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float vnext = 0.0f;
  if (flag[i] != RL_SUCC_TERMINATE) {
    const float z0 = q_next[i], z1 = q_next[n + i];
    vnext = z1 > z0 ? z1 : z0;
  }
  const float dn = gamma * vnext;
  tgt[i] = tgt[i] + dn;
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
  const ReplayRec *__restrict__ ring = rp.rec + (size_t)lane * rp.C;
  for (uint32_t i = threadIdx.x; i < len; i += 64) {
    const uint32_t slot = (start + i) % rp.C;
    const ReplayRec rec = rec_load(ring + slot);
#pragma unroll
    for (int d = 0; d < D; ++d) out_obs[d * out_plane + off + i] = rec.x[d];
    out_action[off + i] = (uint8_t)(rec.af & 0xffu);
    if (one_step_td) {
      const uint32_t fl = rec.af >> 8;
      float vnext = 0.0f;
      if (fl != RL_SUCC_TERMINATE) {
        float x[D], z[2];
        if (fl == RL_SUCC_INTERRUPT || i + 1 == len) {
          const ReplayNext *__restrict__ nx = rp.next + (size_t)lane * rp.C + slot;
#pragma unroll
          for (int d = 0; d < D; ++d) x[d] = nx->x[d];
        } else {
          const ReplayRec r1 = rec_load(ring + (start + i + 1) % rp.C);
#pragma unroll
          for (int d = 0; d < D; ++d) x[d] = r1.x[d];
        }
        mlp_forward_lane<D, 2>(qnet, H, x, z);
        vnext = z[1] > z[0] ? z[1] : z[0];  // amax(-1)
      }
      const float dn = gamma * vnext;
      out_target[off + i] = rec.reward + dn;
    }
  }
  if (one_step_td) return;
  // reward-to-go: backwards in chunks of 1024 steps staged through LDS, scanned by one lane
  if (threadIdx.x == 0) carry = 0.0f;
  __syncthreads();
  for (uint32_t hi = len; hi > 0;) {
    const uint32_t lo = hi > 1024 ? hi - 1024 : 0, cnt = hi - lo;
    for (uint32_t i = threadIdx.x; i < cnt; i += 64) rew[i] = ring[(start + lo + i) % rp.C].reward;
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

// All minibatches of an update in ONE launch (reward-to-go targets do not depend on the network, so nothing in an update
// has to wait for them): one wave per sampled episode, blockIdx.y = minibatch.  Same arithmetic as the builder above
// (G_t = r_t + gamma * G_{t+1}: multiply, then add), the scan runs on broadcast values of one 64-step chunk at a time,
// last chunk first.  Minibatch b lands at obs + b * obs_stride (plane stride 2 * its step count: the T = 1 trajectory
// layout the gradient kernels read), action / target + b * step_stride.
constexpr int BUILD_ALL_WAVES = 4;
template <int D>
__global__ void __launch_bounds__(BUILD_ALL_WAVES * 64)
    k_dqn_build_all(ReplayDev rp, const uint32_t *__restrict__ ep_lane, const uint32_t *__restrict__ ep_start,
                    const uint32_t *__restrict__ ep_len, const uint32_t *__restrict__ ep_offset, uint32_t max_eps,
                    const DqnCountsDev *__restrict__ counts, float *__restrict__ out_obs, size_t obs_stride,
                    uint8_t *__restrict__ out_action, float *__restrict__ out_target, size_t step_stride, float gamma,
                    uint8_t *__restrict__ out_flag) {
  // out_flag != NULL: one-step TD — the targets are left to the gradient kernel (they use the current network): the
  // reward goes where the target would, the successor code into out_flag, the successor observation (the next step's,
  // or the stored one after an Interrupt / at the episode's last step) into time slot 1 of the observation planes
  const bool td = out_flag != nullptr;
  const uint32_t b = blockIdx.y, lane = threadIdx.x & 63;
  const uint32_t e = blockIdx.x * BUILD_ALL_WAVES + (threadIdx.x >> 6);
  const uint32_t n_eps = counts[b].n_eps, n_steps = counts[b].n_steps;
  if (e >= n_eps) return;
  const size_t eo = (size_t)b * max_eps + e;
  const uint32_t ln = ep_lane[eo], start = ep_start[eo], len = ep_len[eo], off = ep_offset[eo];
  const size_t out_plane = (size_t)2 * n_steps;
  const ReplayRec *__restrict__ ring = rp.rec + (size_t)ln * rp.C;
  float *__restrict__ obs_b = out_obs + (size_t)b * obs_stride;
  uint8_t *__restrict__ act_b = out_action + (size_t)b * step_stride;
  float *__restrict__ tgt_b = out_target + (size_t)b * step_stride;
  float g = 0.0f;
  bool first = true;
  for (uint32_t hi = len; hi > 0;) {
    const uint32_t lo = hi > 64 ? hi - 64 : 0, cnt = hi - lo;
    const bool mine = lane < cnt;
    const uint32_t i = lo + (mine ? lane : 0);
    const ReplayRec rec = rec_load(ring + (start + i) % rp.C);
    const float rew = rec.reward;
    float out = 0.0f;
    for (uint32_t k = cnt; k-- > 0;) {  // (uniform trip count; every lane carries the same g)
      const float r = __shfl(rew, (int)k, 64);
      if (first) {
        g = r;
        first = false;
      } else {
        const float p = g * gamma;
        g = r + p;
      }
      if (lane == k) out = g;
    }
    if (mine) {
#pragma unroll
      for (int d = 0; d < D; ++d) obs_b[d * out_plane + off + i] = rec.x[d];
      act_b[off + i] = (uint8_t)(rec.af & 0xffu);
      tgt_b[off + i] = td ? rew : out;
      if (td) {
        const uint32_t fl = rec.af >> 8;
        out_flag[(size_t)b * step_stride + off + i] = (uint8_t)fl;
        float nx[D];
        if (fl == RL_SUCC_INTERRUPT || i + 1 == len) {
          const ReplayNext *__restrict__ nrec = rp.next + (size_t)ln * rp.C + (start + i) % rp.C;
#pragma unroll
          for (int d = 0; d < D; ++d) nx[d] = nrec->x[d];
        } else {
          const ReplayRec r1 = rec_load(ring + (start + i + 1) % rp.C);
#pragma unroll
          for (int d = 0; d < D; ++d) nx[d] = r1.x[d];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) obs_b[d * out_plane + n_steps + off + i] = nx[d];
      }
    }
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

// `stream`: the engine's stream, or the side stream an update draws its later minibatches on (not profiled there)
void launch_dqn_sample(rl_engine *eng, hipStream_t stream, const ReplayDev &rp, const AgentKey &key,
                       uint64_t *d_agent_pos, uint32_t minibatch_steps, uint32_t max_eps, uint32_t *d_lane,
                       uint32_t *d_start, uint32_t *d_len, uint32_t *d_off, DqnCountsDev *d_counts, int sequential,
                       uint32_t n_batches) {
  std::unique_ptr<ProfScope> ps;
  if (stream == eng->stream) ps.reset(new ProfScope(eng, RL_K_SMALL));
  if (rp.N <= (uint32_t)SAMPLE_META_LANES)
    hipLaunchKernelGGL(k_dqn_sample<true>, dim3(1), dim3(SAMPLE_BLOCK), 0, stream, rp, key, d_agent_pos,
                       minibatch_steps, max_eps, d_lane, d_start, d_len, d_off, d_counts, sequential, n_batches);
  else
    hipLaunchKernelGGL(k_dqn_sample<false>, dim3(1), dim3(SAMPLE_BLOCK), 0, stream, rp, key, d_agent_pos,
                       minibatch_steps, max_eps, d_lane, d_start, d_len, d_off, d_counts, sequential, n_batches);
  RL_HIP_CHECK(hipGetLastError());
}

static inline uint32_t cdiv_d(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

template <int G>
static void launch_rollout_dqn_g(rl_env *env, const rl_mlp *qnet, const ReplayDev &rp, uint32_t T, uint64_t p_int,
                                 int always_explore, uint8_t *d_flags, uint32_t *d_range) {
  constexpr int BLOCK = 64;
  const uint32_t n = (uint32_t)env->cfg.n_lanes;
  const dim3 grid(cdiv_d((size_t)n * G, BLOCK)), blk(BLOCK);
  if (env->D == 5)
    hipLaunchKernelGGL((k_rollout_cartpole_dqn<5, BLOCK, G>), grid, blk, 0, env->eng->stream, env->dev, env->st, rp,
                       qnet->d_params, (int)qnet->hidden, T, p_int, always_explore, d_flags, d_range);
  else
    hipLaunchKernelGGL((k_rollout_cartpole_dqn<4, BLOCK, G>), grid, blk, 0, env->eng->stream, env->dev, env->st, rp,
                       qnet->d_params, (int)qnet->hidden, T, p_int, always_explore, d_flags, d_range);
}

void launch_rollout_dqn(rl_env *env, const rl_mlp *qnet, const ReplayDev &rp, uint32_t T, uint64_t p_int,
                        int always_explore, uint8_t *d_flags, uint32_t *d_range) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  // threads per lane: as many as keep the launch within one wave per SIMD (launch_rollout, kernels_rollout.hip); a
  // collection that always explores never runs the forward and keeps one thread per lane
  const uint64_t simds = 4ull * (uint64_t)env->eng->prop.multiProcessorCount, n = env->cfg.n_lanes;
  int G = 1;
  while (!always_explore && G < 16 && n * (uint64_t)(2 * G) <= simds * 64) G *= 2;
  if (env->eng->kernel_variant == 1) G = 1;
  switch (G) {
    case 16: launch_rollout_dqn_g<16>(env, qnet, rp, T, p_int, always_explore, d_flags, d_range); break;
    case 8: launch_rollout_dqn_g<8>(env, qnet, rp, T, p_int, always_explore, d_flags, d_range); break;
    case 4: launch_rollout_dqn_g<4>(env, qnet, rp, T, p_int, always_explore, d_flags, d_range); break;
    case 2: launch_rollout_dqn_g<2>(env, qnet, rp, T, p_int, always_explore, d_flags, d_range); break;
    default: launch_rollout_dqn_g<1>(env, qnet, rp, T, p_int, always_explore, d_flags, d_range); break;
  }
  RL_HIP_CHECK(hipGetLastError());
}

void launch_rollout_dqn_general(rl_env *env, const rl_mlp *qnet, rl_traj *ws, float *d_q, const ReplayDev &rp, uint32_t T,
                                uint64_t p_int, int always_explore, uint8_t *d_flags) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  constexpr int BLOCK = 64;
  const uint32_t n = (uint32_t)env->cfg.n_lanes;
  for (uint32_t t = 0; t < T; ++t) {
    if (!always_explore) {  // (a collection that always explores never reads the module)
      launch_env_observe(env, env->d_obs);                          // [D][n]
      launch_gen_forward(ws, qnet, env->d_obs, (size_t)n, n, d_q);  // [2][n]
    }
    if (env->D == 5)
      hipLaunchKernelGGL((k_dqn_lane_step<5, BLOCK>), dim3(cdiv_d(n, BLOCK)), dim3(BLOCK), 0, env->eng->stream, env->dev,
                         env->st, rp, d_q, p_int, always_explore, t + 1 == T ? 1 : 0, d_flags + (size_t)t * n);
    else
      hipLaunchKernelGGL((k_dqn_lane_step<4, BLOCK>), dim3(cdiv_d(n, BLOCK)), dim3(BLOCK), 0, env->eng->stream, env->dev,
                         env->st, rp, d_q, p_int, always_explore, t + 1 == T ? 1 : 0, d_flags + (size_t)t * n);
  }
  RL_HIP_CHECK(hipGetLastError());
}

void launch_dqn_td_targets(rl_engine *eng, float *d_target, const uint8_t *d_flag, const float *d_q_next, uint32_t n,
                           float gamma) {
  ProfScope ps(eng, RL_K_VALUES);
  if (n == 0) return;
  hipLaunchKernelGGL(k_dqn_td_targets, dim3(cdiv_d(n, 256)), dim3(256), 0, eng->stream, d_target, d_flag, d_q_next, n,
                     gamma);
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

void launch_dqn_build_all(rl_engine *eng, const ReplayDev &rp, uint32_t n_batches, uint32_t widest_eps, uint32_t max_eps,
                          const uint32_t *d_lane, const uint32_t *d_start, const uint32_t *d_len, const uint32_t *d_off,
                          const DqnCountsDev *d_counts, float *d_obs, size_t obs_stride, uint8_t *d_action,
                          float *d_target, size_t step_stride, float gamma, uint8_t *d_flag) {
  ProfScope ps(eng, RL_K_VALUES);
  if (n_batches == 0 || widest_eps == 0) return;
  const dim3 grid(cdiv_d(widest_eps, BUILD_ALL_WAVES), n_batches), block(BUILD_ALL_WAVES * 64);
  if (rp.D == 5)
    hipLaunchKernelGGL(k_dqn_build_all<5>, grid, block, 0, eng->stream, rp, d_lane, d_start, d_len, d_off, max_eps,
                       d_counts, d_obs, obs_stride, d_action, d_target, step_stride, gamma, d_flag);
  else
    hipLaunchKernelGGL(k_dqn_build_all<4>, grid, block, 0, eng->stream, rp, d_lane, d_start, d_len, d_off, max_eps,
                       d_counts, d_obs, obs_stride, d_action, d_target, step_stride, gamma, d_flag);
  RL_HIP_CHECK(hipGetLastError());
}

// ================================================================================================
// The DQN gradient on the bf16 matrix pipe: forward + MSE loss + backward of the 5-128-2 action-value MLP over one
// minibatch — mean((Q(s)[a] - target)^2), dqn.rs:316-326 — with the tile machinery of bf16_tile.hpp (exact three-piece
// products, f32 accumulation; kernels_critic.hip is the one-output version of this kernel).  The loss gradient reaches
// the hidden layer through row a_s of the output weights, so the masked sums over the samples come in two channels,
//   M_c[j][k] = sum_{s : a_s = c} [pre_sj > 0] g_s x~_sk,  g_s = 2 (Q(s)[a_s] - target_s) / B,
// (36 piece columns: one more accumulator set than fits beside the forward at two waves per SIMD — the kernel runs four
// waves per workgroup with the whole register file; a minibatch is ~3 tiles per wave, so latency, not issue, binds) and
//   dW1[j][k] = W2[0][j] M_0[j][k] + W2[1][j] M_1[j][k],   db1[j] likewise with k = 5,
//   dW2[c][j] = sum_k W~1[j][k] M_c[j][k],                  db2[c] = sum_{a_s = c} g_s.
// The forward carries both output chains (relu through |x|, as in the critic step) and picks Q(s)[a_s] per sample.
// ================================================================================================
constexpr int DQN_WAVES = 4;
constexpr int DQN_FLUSH = 16;  // f32 -> f64 flush period in tiles

// TD: the workspace holds rewards instead of targets (`adv`), successor codes (`flag`) and the successor observation in
// time slot 1 of the T = 1 layout; the one-step TD target r + gamma max_a Q(s') (0 beyond a Terminate; critics/mod.rs:
// 139-150, 203-229) comes from a second forward with the same parameters (torch's no_grad target of dqn.rs:299-311).
template <bool TD>
__global__ void __launch_bounds__(DQN_WAVES * 64)
    k_dqn_step_bf16(TrajDev tr, const float *__restrict__ params, const uint32_t *__restrict__ wimg,
                    double *__restrict__ slabA,
                    double *__restrict__ slabB, float two_over_B, uint32_t P, float gamma) {
  using bt::f32x16;
  using bt::Frag;
  constexpr int D = 5, H = 128, NT = bt::NT, A = 2;
  constexpr int CH = H * 7;            // one channel's image: per hidden unit M[0..5] (slot 6 unused)
  constexpr int IMG = A * CH + A + 1;  // two channels, db2[0], db2[1], loss
  __shared__ float Ysh[DQN_WAVES][A][32][33];
  __shared__ double Acc[DQN_WAVES][IMG];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H,
                           *__restrict__ b2 = W2 + A * H;
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  double *acc64 = Acc[wave];
  bool flushed = false;  // (the wave's f64 image is not zeroed: its first flush stores, bt::flush)

  Frag fw[NT][3];
  float w2v[A][NT];
  float lv[A][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};  // the linear half of relu, per output
  const bool guard = blockIdx.x == 0 && wave == 0 && tr.range != nullptr;  // the numeric range guard (bf16_tile.hpp)
  float gxmin = 0.0f, gxmax = 0.0f;
  if (guard) bt::range_bounds(tr.range, lane, gxmin, gxmax);
  // (the 2^96-scaled pieces — relu' by conversion, bf16_tile.hpp; the |pre| chains take the scale back out — come
  // ready-made from the module's weight image, written by whoever wrote the parameters)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bt::WRaw r;
    bt::wimg_load(wimg, t, lane, fw[t], r, 2);
    if (guard)  // (one wave sees all 128 units)
      bt::range_guard_img(r, hf, gxmin, gxmax, tr.range_err + bt::GUARD_POLICY, bt::range_veto(tr.range, bt::GUARD_POLICY));
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const float w2 = r.w2[a];
      lv[a][0] = __builtin_fmaf(w2, r.wa, lv[a][0]);
      lv[a][1] = __builtin_fmaf(w2, r.wb, lv[a][1]);
      lv[a][2] = __builtin_fmaf(w2, r.wc, lv[a][2]);
      w2v[a][t] = bt::FWD_UNSCALE * w2;
    }
  }
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) lv[a][q] = lv[a][q] + __shfl_xor(lv[a][q], m, 64);
  const float b20 = b2[0], b21 = b2[1];
  f32x16 dm[A][NT];
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t) dm[a][t] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double loss64 = 0.0, db64[A] = {0.0, 0.0};
  float loss32 = 0.0f, db32[A] = {0.0f, 0.0f};
  bt::wave_lds_fence();

  Frag selb[2];
  bt::sel_frags(lane, selb);
  const size_t n_tiles = (B + 31) / 32;
  const size_t wave_id = (size_t)blockIdx.x * DQN_WAVES + wave, n_waves = (size_t)gridDim.x * DQN_WAVES;
  int since_flush = 0;
  struct TileOp {
    float xa, xb, xc, tgt;
    float na, nb, nc;  // (TD) the successor observation
    int act, succ;
    bool valid;
  };
  const uint32_t B32 = (uint32_t)B, plane32 = (uint32_t)plane;
  auto load_tile = [&](size_t g) {  // (branch-free: padding lanes read sample B - 1 and are zeroed)
    TileOp o;
    const uint32_t sidx = (uint32_t)g * 32u + (uint32_t)n;
    o.valid = g < n_tiles && sidx < B32;
    const uint32_t sc = o.valid ? sidx : B32 - 1;
    const float xa = tr.obs[(uint32_t)(2 * hf) * plane32 + sc], xb = tr.obs[(uint32_t)(2 * hf + 1) * plane32 + sc];
    const float xc = tr.obs[4u * plane32 + sc], tg = tr.adv[sc];  // (the minibatch workspace keeps its targets in `adv`)
    const int act = (int)tr.action[sc];
    o.xa = o.valid ? xa : 0.0f;
    o.xb = o.valid ? xb : 0.0f;
    o.xc = o.valid ? xc : 0.0f;
    o.tgt = o.valid ? tg : 0.0f;
    o.act = o.valid ? act : 0;
    o.na = o.nb = o.nc = 0.0f;
    o.succ = RL_SUCC_TERMINATE;
    if (TD) {
      const uint32_t s1 = B32 + sc;  // time slot 1
      const float na = tr.obs[(uint32_t)(2 * hf) * plane32 + s1], nb = tr.obs[(uint32_t)(2 * hf + 1) * plane32 + s1];
      const float nc = tr.obs[4u * plane32 + s1];
      const int succ = (int)tr.flag[sc];
      o.na = o.valid ? na : 0.0f;
      o.nb = o.valid ? nb : 0.0f;
      o.nc = o.valid ? nc : 0.0f;
      o.succ = o.valid ? succ : RL_SUCC_TERMINATE;
    }
    return o;
  };
  auto flush_all = [&]() {
    bt::flush(dm[0], acc64, 7, n, hf, !flushed);
    bt::flush(dm[1], acc64 + CH, 7, n, hf, !flushed);
    flushed = true;
    loss64 += (double)loss32;
    db64[0] += (double)db32[0];
    db64[1] += (double)db32[1];
    loss32 = db32[0] = db32[1] = 0.0f;
  };

  TileOp op = load_tile(wave_id);
  for (size_t g = wave_id; g < n_tiles; g += n_waves) {
    const TileOp next = load_tile(g + n_waves);
    Frag ga[NT][2];
    // both action values of one observation per sample lane; `masks`: keep relu' of this forward for the backward
    auto forward = [&](float xa, float xb, float xc, bool masks, float (&qv)[A]) {
      Frag fa[3];
      bt::input_frags(xa, xb, xc, op.valid, hf, fa);
      float yp[A][16];
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) yp[a][r] = 0.0f;
      f32x16 c = bt::layer1(fa, fw[0]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x16 cn = c;
        if (t + 1 < NT) cn = bt::layer1(fa, fw[t + 1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ab = __builtin_fabsf(c[r]);
          yp[0][r] = __builtin_fmaf(ab, w2v[0][t], yp[0][r]);
          yp[1][r] = __builtin_fmaf(ab, w2v[1][t], yp[1][r]);
        }
        if (masks) bt::mask_tile(c, ga[t]);
        c = cn;
      }
      // both outputs: transpose the 16 partial sums per lane through LDS (row = sample, column = source lane)
#pragma unroll
      for (int a = 0; a < A; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) Ysh[wave][a][(r & 3) + 8 * (r >> 2) + 4 * hf][n] = yp[a][r];
      bt::wave_lds_fence();
#pragma unroll
      for (int a = 0; a < A; ++a) {
        float part = bt::row_sum16(&Ysh[wave][a][n][hf * 16]);
        float lin = lv[a][0] * xa;
        lin = __builtin_fmaf(lv[a][1], xb, lin);
        lin = __builtin_fmaf(lv[a][2], hf == 0 ? xc : 1.0f, lin);
        part = part + lin;
        float p0, p1;
        bt::both_halves(part, p0, p1);
        qv[a] = 0.5f * (p0 + p1) + (a == 0 ? b20 : b21);
      }
      bt::wave_lds_fence();  // Ysh is rewritten by the next forward
    };
    float tgt = op.tgt;
    if (TD) {
      float qn[A];
      forward(op.na, op.nb, op.nc, false, qn);
      const float vnext = op.succ == RL_SUCC_TERMINATE ? 0.0f : (qn[1] > qn[0] ? qn[1] : qn[0]);  // amax(-1)
      const float dn = gamma * vnext;
      tgt = op.tgt + dn;  // (op.tgt holds the reward)
    }
    float qv[A];
    forward(op.xa, op.xb, op.xc, true, qv);
    const float d = (op.act == 0 ? qv[0] : qv[1]) - tgt;
    const float gq = op.valid ? d * two_over_B : 0.0f;
    const float g0 = op.act == 0 ? gq : 0.0f, g1 = op.act == 0 ? 0.0f : gq;
    if (hf == 0 && op.valid) {
      loss32 = __builtin_fmaf(d, d, loss32);
      db32[0] = db32[0] + g0;
      db32[1] = db32[1] + g1;
    }
    Frag ub[2];
    bt::piece_frags_mfma(g0, op.xa, op.xb, op.xc, hf, selb, ub);
    bt::backward(ga, ub, dm[0]);
    bt::piece_frags_mfma(g1, op.xa, op.xb, op.xc, hf, selb, ub);
    bt::backward(ga, ub, dm[1]);
    if (++since_flush == DQN_FLUSH) {
      since_flush = 0;
      flush_all();
    }
    op = next;
  }
  if (since_flush != 0 || !flushed) flush_all();  // (nothing left when the last tile ended a flush period; a wave
                                                  // without tiles still defines its image)
  auto xlane = [](double v, int mask) {
    uint64_t bits = rl_f64_bits(v);
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, mask, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), mask, 64);
    return rl_f64_from_bits(((uint64_t)hi << 32) | lo);
  };
  double l = hf == 0 ? loss64 : 0.0, s0 = hf == 0 ? db64[0] : 0.0, s1 = hf == 0 ? db64[1] : 0.0;
#pragma unroll
  for (int s = 16; s > 0; s >>= 1) {
    l = l + xlane(l, s);
    s0 = s0 + xlane(s0, s);
    s1 = s1 + xlane(s1, s);
  }
  if (lane == 0) {
    acc64[A * CH] = s0;
    acc64[A * CH + 1] = s1;
    acc64[A * CH + 2] = l;
  }
  __syncthreads();
  auto tot = [&](int src) {
    double s = Acc[0][src];
#pragma unroll
    for (int w = 1; w < DQN_WAVES; ++w) s = s + Acc[w][src];
    return s;
  };
  for (uint32_t p = threadIdx.x; p < P; p += DQN_WAVES * 64) {
    double s;
    if (p < (uint32_t)(H * D)) {
      const int j = p / D, k = p % D;
      s = tot(j * 7 + k) * (double)W2[j] + tot(CH + j * 7 + k) * (double)W2[H + j];
    } else if (p < (uint32_t)(H * D + H)) {
      const int j = p - H * D;
      s = tot(j * 7 + 5) * (double)W2[j] + tot(CH + j * 7 + 5) * (double)W2[H + j];
    } else if (p < (uint32_t)(H * D + H + A * H)) {
      const int q = p - H * D - H, a = q / H, j = q % H;
      s = tot(a * CH + j * 7 + 5) * (double)b1[j];
#pragma unroll
      for (int k = 0; k < D; ++k) s += tot(a * CH + j * 7 + k) * (double)W1[j * D + k];
    } else {
      s = tot(A * CH + (int)(p - (H * D + H + A * H)));
    }
    slabA[(size_t)blockIdx.x * P + p] = s;
  }
  if (threadIdx.x < 4) slabB[(size_t)blockIdx.x * 4 + threadIdx.x] = threadIdx.x == 0 ? tot(A * CH + 2) : 0.0;
}

// returns false when the kernel is not built for this shape (the caller falls back to the f32 passes)
bool launch_dqn_step_bf16(rl_traj *mb, const rl_mlp *qnet, uint64_t B_total, bool td_in_kernel, float gamma) {
  if (mb->d.D != 5 || qnet->hidden != 128 || qnet->out_dim != 2 || qnet->general) return false;
  if ((uint64_t)(mb->d.T + 1) * mb->d.n * 5 >= (1ull << 30)) return false;  // 32-bit element offsets in the kernel
  const uint32_t *wimg = wimg_ensure(qnet);
  ProfScope ps(mb->eng, RL_K_POLICY_FUSED);
  const uint64_t n_tiles = (mb->B + 31) / 32, cus = (uint64_t)mb->eng->prop.multiProcessorCount;
  uint64_t nb = (n_tiles + DQN_WAVES - 1) / DQN_WAVES;
  if (nb > cus) nb = cus;
  mb->nbV2 = (uint32_t)nb;  // slab rows of this launch (the slabs are sized for any grid up to 8 x CUs)
  TrajDev d = mb->d;
  if (!mb->guard_next_policy) d.range = nullptr;  // (the range guard: first step of an update only, engine.hpp)
  mb->guard_next_policy = false;
  if (td_in_kernel)
    hipLaunchKernelGGL(k_dqn_step_bf16<true>, dim3((uint32_t)nb), dim3(DQN_WAVES * 64), 0, mb->eng->stream, d,
                       qnet->d_params, wimg, mb->slabA, mb->slabB, 2.0f / (float)B_total, (uint32_t)qnet->P, gamma);
  else
    hipLaunchKernelGGL(k_dqn_step_bf16<false>, dim3((uint32_t)nb), dim3(DQN_WAVES * 64), 0, mb->eng->stream, d,
                       qnet->d_params, wimg, mb->slabA, mb->slabB, 2.0f / (float)B_total, (uint32_t)qnet->P, gamma);
  RL_HIP_CHECK(hipGetLastError());
  return true;
}

// rl_dqn_replay_read: the store's step data in the documented plane layouts ([D][C][N] features, [C][N] the rest; lane
// fastest) out of the records — a test / inspection path
__global__ void __launch_bounds__(256) k_replay_planes(ReplayDev rp, int field, void *__restrict__ out) {
  const size_t cn = (size_t)rp.C * rp.N;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= cn) return;
  const uint32_t lane = (uint32_t)(idx % rp.N), slot = (uint32_t)(idx / rp.N);
  const size_t o = (size_t)lane * rp.C + slot;
  if (field == RL_REPLAY_OBS) {
    for (uint32_t d = 0; d < rp.D; ++d) static_cast<float *>(out)[d * cn + idx] = rp.rec[o].x[d];
  } else if (field == RL_REPLAY_NEXT_OBS) {
    for (uint32_t d = 0; d < rp.D; ++d) static_cast<float *>(out)[d * cn + idx] = rp.next[o].x[d];
  } else if (field == RL_REPLAY_ACTION) {
    static_cast<uint8_t *>(out)[idx] = (uint8_t)(rp.rec[o].af & 0xffu);
  } else if (field == RL_REPLAY_FLAG) {
    static_cast<uint8_t *>(out)[idx] = (uint8_t)(rp.rec[o].af >> 8);
  } else {
    static_cast<float *>(out)[idx] = rp.rec[o].reward;
  }
}

void launch_replay_planes(rl_engine *eng, const ReplayDev &rp, int field, void *d_out) {
  const size_t cn = (size_t)rp.C * rp.N;
  hipLaunchKernelGGL(k_replay_planes, dim3(cdiv_d(cn, 256)), dim3(256), 0, eng->stream, rp, field, d_out);
  RL_HIP_CHECK(hipGetLastError());
}
