// kernels_seq.hip — the recurrent configuration (BASELINE.json configs[4]): vectorised Chain lanes and the
// GRU -> ReLU -> MLP module (`GruMlpConfig`, src/torch/modules/mod.rs:14) as persistent, tile-resident MFMA kernels.
//
// Reference: Chain<Gru, Mlp> (src/torch/modules/chain.rs:127-186), GruImpl::cell_batch_step -> gru_cell
// (src/torch/modules/seq/rnn/gru.rs:30-39), gru_data over packed episodes (gru.rs:76-98), Chain env
// (src/envs/chain.rs:69-106), LatentStepLimit (src/envs/wrappers/step_limit.rs:57-89).
//
// One workgroup (4 waves) owns a TILE of 32 lanes for the whole horizon.  Wave w owns hidden units [32w, 32w+32)
// of every gate and of the MLP layer: its slices of W_hh (3 x 64 MFMA B-operands) and W1 (64) stay in registers,
// the recurrent state h of the tile stays in LDS ([k][m], padded) and in the owner lanes' registers, so a step
// touches HBM only for the trajectory record (and, in training passes, the activation record).
//   gates  : C[m][j] = b_hh[j] + sum_k h[m][k] W_hh[j][k]  — v_mfma_f32_32x32x2_f32 is an exact sequential
//            fma chain over k (measured, scripts/probe/mfma_arith.hip), so this equals the oracle's
//            acc = bias; acc = fma(h_k, w_k, acc), k ascending, bit for bit;
//   input  : gi = b_ih + sum_d x_d W_ih[j][d] on the VALU (D <= 8);
//   cell   : r = sig(gh_r + gi_r), z = sig(gh_z + gi_z), n = tanh(gi_n + gh_n * r), h' = (h - n) * z + n
//            (libtorch gru_cell operation order), rl_sigmoidf / rl_tanhf of include/rl_detmath.h;
//   head   : u = relu(b1 + W1 relu(h')) by MFMA, out_a = b2_a + sum_j u_j W2[a][j] as a sequential chain on the
//            VALU (one lane per (sample, output)).
#include "device_fns.hpp"
#include "kernels.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GH = 128;      // GRU hidden width (ChainConfig::hidden_dim, chain.rs:28)
constexpr int MH = 128;      // MLP hidden width (MlpConfig::default)
constexpr int TL = 32;       // lanes per tile
constexpr int SEQ_ARR = 7;   // activation arrays per step: r, z, n, gh_n, h_prev, relu(h'), u
constexpr int DPRE_ARR = 5;  // backward arrays per step: d pre_r, d pre_z, d pre_n, d pre_n * r, d u_pre
enum { ACT_R = 0, ACT_Z = 1, ACT_N = 2, ACT_GHN = 3, ACT_HPREV = 4, ACT_A1 = 5, ACT_U = 6 };

// ---------------------------------------------------------------- Chain lanes
struct ChainLane {
  uint32_t state, steps_remaining, reset_count;
};

__device__ __forceinline__ void chain_load(const EnvStateDev &st, uint32_t i, ChainLane &s) {
  s.state = (uint32_t)st.x[i];
  s.steps_remaining = st.steps_remaining[i];
  s.reset_count = st.reset_count[i];
}

__device__ __forceinline__ void chain_store(const EnvStateDev &st, uint32_t i, const ChainLane &s) {
  st.x[i] = (double)s.state;
  st.steps_remaining[i] = s.steps_remaining;
  st.reset_count[i] = s.reset_count;
}

// features of StepLimit-wrapped IndexSpace observations: one-hot (spaces/index.rs:104-116) [+ remaining]
template <int D>
__device__ __forceinline__ void chain_features(const CartPoleDev &c, const ChainLane &s, float (&f)[D]) {
#pragma unroll
  for (int d = 0; d < D; ++d) f[d] = (uint32_t)d == s.state ? 1.0f : 0.0f;
  if (D == 6) f[5] = (float)((double)s.steps_remaining / (double)c.max_steps);
}

__device__ __forceinline__ void chain_reset(const CartPoleDev &c, ChainLane &s) {
  s.state = 0;  // Chain::initial_state (chain.rs:75-77), no random draw
  s.steps_remaining = c.max_steps;
  s.reset_count += 1;
}

// Chain::step (chain.rs:83-105) + the step-limit tail; `word` is the lane's env-stream word for this global step
__device__ __forceinline__ int chain_step(const CartPoleDev &c, ChainLane &s, int action, uint32_t word,
                                          float &reward) {
  if (rl_u32_to_unit_f32(word) < 0.2f) action = 1 - action;  // Move::invert
  if (action == 0) {  // Move::Left
    s.state = 0;
    reward = 2.0f;
  } else if (s.state == c.chain_size - 1) {
    reward = 10.0f;
  } else {
    s.state += 1;
    reward = 0.0f;
  }
  if (c.limit_kind != RL_LIMIT_NONE) {
    s.steps_remaining -= 1;
    if (s.steps_remaining == 0) return RL_SUCC_INTERRUPT;
  }
  return RL_SUCC_CONTINUE;
}

__device__ __forceinline__ uint32_t stream_word(const uint32_t *key, uint64_t stream, uint64_t word) {
  uint32_t w[16];
  rl_chacha_block(key, word >> 4, stream, 4, w);
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k == (int)(word & 15)) v = w[k];
  return v;
}

__global__ void k_chain_reset(CartPoleDev c, EnvStateDev st, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ChainLane s;
  s.reset_count = st.reset_count[i];
  chain_reset(c, s);
  chain_store(st, i, s);
}

template <int D>
__global__ void k_chain_observe(CartPoleDev c, EnvStateDev st, uint32_t n, float *__restrict__ obs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ChainLane s;
  chain_load(st, i, s);
  float f[D];
  chain_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) obs[(size_t)d * n + i] = f[d];
}

template <int D>
__global__ void __launch_bounds__(256) k_chain_step(CartPoleDev c, EnvStateDev st, uint32_t n, uint64_t t_global,
                                                    const uint8_t *__restrict__ actions, float *__restrict__ reward,
                                                    uint8_t *__restrict__ flag, float *__restrict__ obs_next,
                                                    float *__restrict__ term_obs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ChainLane s;
  chain_load(st, i, s);
  float r;
  int succ = chain_step(c, s, actions[i], stream_word(c.key_env, c.lane_offset + i, t_global), r);
  float f[D];
  if (succ == RL_SUCC_INTERRUPT) {
    chain_features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) term_obs[(size_t)d * n + i] = f[d];
  }
  if (succ != RL_SUCC_CONTINUE) chain_reset(c, s);
  chain_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) obs_next[(size_t)d * n + i] = f[d];
  reward[i] = r;
  flag[i] = (uint8_t)succ;
  chain_store(st, i, s);
}

// ---------------------------------------------------------------- GRU-MLP parameter views
struct GruParams {
  const float *Wih, *Whh, *bih, *bhh, *W1, *b1, *W2, *b2;
};

__host__ __device__ inline GruParams gru_params(const float *p, int D, int A) {
  GruParams g;
  g.Wih = p;
  g.Whh = g.Wih + 3 * GH * D;
  g.bih = g.Whh + 3 * GH * GH;
  g.bhh = g.bih + 3 * GH;
  g.W1 = g.bhh + 3 * GH;
  g.b1 = g.W1 + MH * GH;
  g.W2 = g.b1 + MH;
  g.b2 = g.W2 + A * MH;
  return g;
}

__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// LDS of the forward kernels
struct SeqFwdShared {
  float hT[2][GH][TL + 1];  // recurrent state, [k][m], double-buffered
  float uS[TL][MH + 1];     // MLP hidden activations, [m][j]
  float xS[TL][8];          // observation features of the current step
  float w2S[2][MH];
  float outS[2][TL];
  int endS[TL];             // != 0: the lane's episode ended at this step (recurrent state restarts)
  int peek;                 // != 0: some lane of the tile needs a successor evaluation at this step
};

// Register-resident weight slices of one wave (unit j = 32 * wave + (lane & 31), k parity = lane >> 5)
template <int D>
struct SeqFwdWeights {
  float whh[3][GH / 2];
  float w1[GH / 2];
  float wih[3][D];
  float bih[3], bhh[3], b1;
};

template <int D>
__device__ __forceinline__ void seq_load_weights(SeqFwdWeights<D> &w, const GruParams &g, int wave, int lane) {
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
#pragma unroll
  for (int gte = 0; gte < 3; ++gte) {
    const int row = gte * GH + j;
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) w.whh[gte][ks] = g.Whh[(size_t)row * GH + 2 * ks + hf];
#pragma unroll
    for (int d = 0; d < D; ++d) w.wih[gte][d] = g.Wih[(size_t)row * D + d];
    w.bih[gte] = g.bih[row];
    w.bhh[gte] = g.bhh[row];
  }
#pragma unroll
  for (int ks = 0; ks < GH / 2; ++ks) w.w1[ks] = g.W1[(size_t)j * GH + 2 * ks + hf];
  w.b1 = g.b1[j];
}

// One cell + head evaluation for the tile.  Reads the state from sh.hT[cur] and `hown`, writes the new state to
// sh.hT[cur ^ 1] and `hnew`; the head outputs land in sh.outS (valid after the function returns: it ends with a
// barrier).  `store` != nullptr: record the 7 activation arrays of this (t, tile) block.
template <int D, int A>
__device__ __forceinline__ void seq_cell(SeqFwdShared &sh, const SeqFwdWeights<D> &w, int cur, const float (&hown)[16],
                                         float (&hnew)[16], float b2_mine, float *__restrict__ store, int wave,
                                         int lane) {
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n, nxt = cur ^ 1;
  f32x16 acc[3];
#pragma unroll
  for (int gte = 0; gte < 3; ++gte)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[gte][r] = w.bhh[gte];
#pragma unroll
  for (int ks = 0; ks < GH / 2; ++ks) {
    const float a = sh.hT[cur][2 * ks + hf][n];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte) acc[gte] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.whh[gte][ks], acc[gte], 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = acc_row(r, hf);
    float gi[3];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte) {
      float v = w.bih[gte];
#pragma unroll
      for (int d = 0; d < D; ++d) v = __builtin_fmaf(sh.xS[m][d], w.wih[gte][d], v);
      gi[gte] = v;
    }
    const float rr = rl_sigmoidf(acc[0][r] + gi[0]);
    const float zz = rl_sigmoidf(acc[1][r] + gi[1]);
    const float rn = acc[2][r] * rr;
    const float nn = rl_tanhf(gi[2] + rn);
    const float dn = hown[r] - nn;
    const float hz = dn * zz;
    const float hv = hz + nn;
    hnew[r] = hv;
    sh.hT[nxt][j][m] = hv;
    if (store != nullptr) {
      // [arr][j][m]: this lane owns 4 runs of 4 consecutive m for its unit j
      store[(size_t)ACT_R * GH * TL + j * TL + m] = rr;
      store[(size_t)ACT_Z * GH * TL + j * TL + m] = zz;
      store[(size_t)ACT_N * GH * TL + j * TL + m] = nn;
      store[(size_t)ACT_GHN * GH * TL + j * TL + m] = acc[2][r];
      store[(size_t)ACT_HPREV * GH * TL + j * TL + m] = hown[r];
      store[(size_t)ACT_A1 * GH * TL + j * TL + m] = hv > 0.0f ? hv : 0.0f;
    }
  }
  __syncthreads();
  f32x16 acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = w.b1;
#pragma unroll
  for (int ks = 0; ks < GH / 2; ++ks) {
    float a = sh.hT[nxt][2 * ks + hf][n];
    a = a > 0.0f ? a : 0.0f;  // Chain activation between the modules (chain.rs:165)
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.w1[ks], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = acc_row(r, hf);
    const float u = acc1[r] > 0.0f ? acc1[r] : 0.0f;
    sh.uS[m][j] = u;
    if (store != nullptr) store[(size_t)ACT_U * GH * TL + j * TL + m] = u;
  }
  __syncthreads();
  if (wave == 0 && hf < A) {
    // out_a[m] = b2_a + sum_j u[m][j] W2[a][j], sequential chain (lane = (m = n, a = hf))
    float z = b2_mine;
#pragma unroll 8
    for (int q = 0; q < MH; ++q) z = __builtin_fmaf(sh.uS[n][q], sh.w2S[hf][q], z);
    sh.outS[hf][n] = z;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- rollout (Chain env, recurrent policy)
// PolicyActor::act over SeqIterative::step (policies/actor.rs:42-55; chain.rs:175-186) for T steps of every lane.
// The episode state starts at zero at the beginning of the launch and after every episode end.
template <int D>
__global__ void __launch_bounds__(256, 1) k_rollout_chain_gru(CartPoleDev c, EnvStateDev st, TrajDev tr,
                                                              const float *__restrict__ params, uint64_t t_global) {
  constexpr int A = 2;
  __shared__ SeqFwdShared sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t lane0 = blockIdx.x * TL;
  const GruParams g = gru_params(params, D, A);
  SeqFwdWeights<D> w;
  seq_load_weights<D>(w, g, wave, lane);
  for (int q = threadIdx.x; q < A * MH; q += 256) sh.w2S[q / MH][q % MH] = g.W2[q];
  for (int q = threadIdx.x; q < 2 * GH * (TL + 1); q += 256) (&sh.hT[0][0][0])[q] = 0.0f;
  const float b2_mine = hf < A ? g.b2[hf] : 0.0f;
  float hown[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hown[r] = 0.0f;
  // env state of the tile: lanes 0..31 of wave 0
  const bool env_lane = wave == 0 && lane < TL;
  const uint32_t i = lane0 + (uint32_t)lane;
  const uint64_t glane = c.lane_offset + i;
  ChainLane s{0, 0, 0};
  const size_t plane = (size_t)(T + 1) * N;
  if (env_lane) {
    chain_load(st, i, s);
    float f[D];
    chain_features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      sh.xS[lane][d] = f[d];
      tr.obs[d * plane + i] = f[d];
    }
  }
  __syncthreads();
  int cur = 0;
  for (uint32_t t = 0; t < T; ++t) {
    float hnew[16];
    seq_cell<D, A>(sh, w, cur, hown, hnew, b2_mine, nullptr, wave, lane);
    if (env_lane) {
      float z[2] = {sh.outS[0][lane], sh.outS[1][lane]}, lp[2];
      log_softmax_lane<2>(z, lp);
      const uint64_t word = t_global + t;
      const float u = rl_u32_to_unit_f32(stream_word(c.key_actor, glane, word));
      const int a = categorical_sample_lane<2>(lp, u);
      float rew;
      const int succ = chain_step(c, s, a, stream_word(c.key_env, glane, word), rew);
      const size_t o = (size_t)t * N + i;
      tr.action[o] = (uint8_t)a;
      tr.reward[o] = rew;
      tr.flag[o] = (uint8_t)succ;
      float f[D];
      if (succ == RL_SUCC_INTERRUPT) {
        chain_features<D>(c, s, f);
#pragma unroll
        for (int d = 0; d < D; ++d) tr.term_obs[(size_t)d * T * N + o] = f[d];
      }
      if (succ != RL_SUCC_CONTINUE) chain_reset(c, s);
      chain_features<D>(c, s, f);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        sh.xS[lane][d] = f[d];
        tr.obs[d * plane + (size_t)(t + 1) * N + i] = f[d];
      }
      sh.endS[lane] = succ != RL_SUCC_CONTINUE;
    }
    __syncthreads();
    const int nxt = cur ^ 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const bool ended = sh.endS[m] != 0;
      hown[r] = ended ? 0.0f : hnew[r];
      if (ended) sh.hT[nxt][j][m] = 0.0f;
    }
    __syncthreads();
    cur = nxt;
  }
  if (env_lane) chain_store(st, i, s);
}

// ---------------------------------------------------------------- teacher-forced forward over a trajectory
// SeqPacked::seq_packed of Chain<Gru, Mlp> (chain.rs:151-161; gru.rs:76-98) on the lane layout.  Writes the module
// outputs out[a][t][lane]; optionally the outputs at the successor observations of cut episodes (extended
// observation sequences, features.rs:132-178) and the activation record for the backward pass.
template <int D, int A>
__global__ void __launch_bounds__(256, 1) k_gru_seq_forward(TrajDev tr, const float *__restrict__ params,
                                                            float *__restrict__ out, float *__restrict__ succ_out,
                                                            float *__restrict__ act) {
  __shared__ SeqFwdShared sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = gru_params(params, D, A);
  SeqFwdWeights<D> w;
  seq_load_weights<D>(w, g, wave, lane);
  for (int q = threadIdx.x; q < A * MH; q += 256) sh.w2S[q / MH][q % MH] = g.W2[q];
  for (int q = threadIdx.x; q < 2 * GH * (TL + 1); q += 256) (&sh.hT[0][0][0])[q] = 0.0f;
  const float b2_mine = hf < A ? g.b2[hf] : 0.0f;
  float hown[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hown[r] = 0.0f;
  const bool io_lane = wave == 0 && lane < TL;
  const uint32_t i = lane0 + (uint32_t)lane;
  const size_t plane = (size_t)(T + 1) * N;
  int cur = 0, flag = 0;
  bool peeking = false, need = false;
  uint32_t t = 0;
  // One cell evaluation per iteration.  A step whose episode is cut in some lane of the tile is followed by a
  // "peek" iteration that evaluates the successor observation from the post-step state without advancing it.
  while (t < T) {
    if (!peeking) {
      if (io_lane) {
#pragma unroll
        for (int d = 0; d < D; ++d) sh.xS[lane][d] = tr.obs[d * plane + (size_t)t * N + i];
        flag = tr.flag[(size_t)t * N + i];
        sh.endS[lane] = flag != RL_SUCC_CONTINUE;
      }
      if (threadIdx.x == 0) sh.peek = 0;
    }
    __syncthreads();
    float hout[16];
    float *store = (act != nullptr && !peeking) ? act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL : nullptr;
    seq_cell<D, A>(sh, w, cur, hown, hout, b2_mine, store, wave, lane);
    if (io_lane) {
      if (!peeking) {
#pragma unroll
        for (int a = 0; a < A; ++a) out[((size_t)a * T + t) * N + i] = sh.outS[a][lane];
        need = succ_out != nullptr && (flag == RL_SUCC_INTERRUPT || (flag == RL_SUCC_CONTINUE && t == T - 1));
        if (succ_out != nullptr && !need)
#pragma unroll
          for (int a = 0; a < A; ++a) succ_out[((size_t)a * T + t) * N + i] = 0.0f;
        if (need) {
          sh.peek = 1;
#pragma unroll
          for (int d = 0; d < D; ++d)
            sh.xS[lane][d] = flag == RL_SUCC_INTERRUPT ? tr.term_obs[((size_t)d * T + t) * N + i]
                                                       : tr.obs[d * plane + (size_t)T * N + i];
        }
      } else if (need) {
#pragma unroll
        for (int a = 0; a < A; ++a) succ_out[((size_t)a * T + t) * N + i] = sh.outS[a][lane];
      }
    }
    __syncthreads();
    if (!peeking && sh.peek != 0) {
      // keep the post-step state as the input of the peek iteration; the reset of ended lanes waits
#pragma unroll
      for (int r = 0; r < 16; ++r) hown[r] = hout[r];
      cur ^= 1;
      peeking = true;
    } else {
      // commit: restart the state of lanes whose episode ended at step t
      const int buf = peeking ? cur : (cur ^ 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(r, hf);
        const bool ended = sh.endS[m] != 0;
        const float hv = peeking ? hown[r] : hout[r];
        hown[r] = ended ? 0.0f : hv;
        if (ended) sh.hT[buf][j][m] = 0.0f;
      }
      cur = buf;
      peeking = false;
      t += 1;
    }
    __syncthreads();
  }
}

static inline uint32_t cdiv_s(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

// ---------------------------------------------------------------- GAE with a recurrent critic
// gae / temporal_differences / reward_to_go (critics/mod.rs:101-199) with the value of every observation and of
// every cut episode's successor taken from the teacher-forced forward (values [T][n], succ [T][n]).  Same
// arithmetic as k_gae_scan.  Also mirrors the values into the trajectory's [T+1][n] plane for inspection.
__global__ void __launch_bounds__(64) k_seq_gae(TrajDev tr, const float *__restrict__ values,
                                                const float *__restrict__ succ, float gamma, float lambda) {
  const uint32_t n = tr.n, T = tr.T;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float disc = lambda * gamma;
  float adv_next = 0.0f, rtg_next = 0.0f, v_next = 0.0f;
  tr.values[(size_t)T * n + i] = succ[(size_t)(T - 1) * n + i];
  for (uint32_t t = T; t-- > 0;) {
    const size_t o = (size_t)t * n + i;
    const uint8_t f = tr.flag[o];
    const float r = tr.reward[o], v = values[o];
    float vn;
    bool ends;
    if (f == RL_SUCC_TERMINATE) {
      vn = 0.0f;
      ends = true;
    } else if (f == RL_SUCC_INTERRUPT || t == T - 1) {
      vn = succ[o];
      ends = true;
    } else {
      vn = v_next;
      ends = false;
    }
    const float dn = gamma * vn;
    const float tmp = r + dn;
    const float delta = tmp - v;
    float a, g;
    if (ends) {
      a = delta;
      g = r;
    } else {
      const float pa = adv_next * disc;
      a = delta + pa;
      const float pg = rtg_next * gamma;
      g = r + pg;
    }
    tr.adv[o] = a;
    tr.rtg[o] = g;
    tr.values[o] = v;
    adv_next = a;
    rtg_next = g;
    v_next = v;
  }
}

void launch_seq_gae(rl_traj *traj, float gamma, float lambda) {
  ProfScope ps(traj->eng, RL_K_GAE);
  uint32_t n = traj->d.n;
  hipLaunchKernelGGL(k_seq_gae, dim3(cdiv_s(n, 64)), dim3(64), 0, traj->eng->stream, traj->d, traj->seq.out,
                     traj->seq.succ, gamma, lambda);
}

void launch_chain_reset(rl_env *env) {
  ProfScope ps(env->eng, RL_K_SMALL);
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  hipLaunchKernelGGL(k_chain_reset, dim3(cdiv_s(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n);
}

void launch_chain_observe(rl_env *env, float *d_obs) {
  ProfScope ps(env->eng, RL_K_SMALL);
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  if (env->D == 5)
    hipLaunchKernelGGL(k_chain_observe<5>, dim3(cdiv_s(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n,
                       d_obs);
  else
    hipLaunchKernelGGL(k_chain_observe<6>, dim3(cdiv_s(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n,
                       d_obs);
}

void launch_chain_step(rl_env *env) {
  ProfScope ps(env->eng, RL_K_ENV_STEP);
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  if (env->D == 5)
    hipLaunchKernelGGL(k_chain_step<5>, dim3(cdiv_s(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n,
                       env->t_global, env->d_actions, env->d_reward, env->d_flag, env->d_obs, env->d_term_obs);
  else
    hipLaunchKernelGGL(k_chain_step<6>, dim3(cdiv_s(n, 256)), dim3(256), 0, env->eng->stream, env->dev, env->st, n,
                       env->t_global, env->d_actions, env->d_reward, env->d_flag, env->d_obs, env->d_term_obs);
}

void launch_rollout_chain_gru(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  RL_REQUIRE(env->D == 5, "recurrent rollout: built for 5 observation features (Chain under a latent step limit)");
  uint32_t tiles = traj->d.n / TL;
  hipLaunchKernelGGL(k_rollout_chain_gru<5>, dim3(tiles), dim3(256), 0, env->eng->stream, env->dev, env->st, traj->d,
                     policy->d_params, env->t_global);
}

void launch_gru_seq_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_succ, float *d_act) {
  ProfScope ps(traj->eng, RL_K_POLICY_FUSED);
  RL_REQUIRE(traj->d.D == 5, "recurrent forward: built for 5 observation features");
  uint32_t tiles = traj->d.n / TL;
  if (mod->out_dim == 2)
    hipLaunchKernelGGL((k_gru_seq_forward<5, 2>), dim3(tiles), dim3(256), 0, traj->eng->stream, traj->d,
                       mod->d_params, d_out, d_succ, d_act);
  else
    hipLaunchKernelGGL((k_gru_seq_forward<5, 1>), dim3(tiles), dim3(256), 0, traj->eng->stream, traj->d,
                       mod->d_params, d_out, d_succ, d_act);
}
