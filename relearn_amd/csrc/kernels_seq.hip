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
#include "seq_common.hpp"

// ---------------------------------------------------------------- lanes of the IndexSpace-observation envs
// Chain (chain.rs) and MemoryGame (memory.rs) share the lane code: `c.mem_actions` == 0 selects Chain (a launch-uniform
// branch).  MemoryGame keeps (current_state, initial_state) and the word position of the lane's env stream: its only
// random draw is `rng.gen_range(0..num_actions)` in initial_state, taken SEQUENTIALLY from the lane's stream like one
// worker's env Prng in the reference (a rejection loop, so the number of words per reset is not fixed).
struct ChainLane {
  uint32_t state, steps_remaining, reset_count;
  uint32_t initial;   // MemoryGame: the state the episode started in
  uint64_t env_pos;   // MemoryGame: next unread word of the lane's env stream (always even: u64 draws only)
};

__device__ __forceinline__ void chain_load(const EnvStateDev &st, uint32_t i, ChainLane &s) {
  s.state = (uint32_t)st.x[i];
  s.initial = (uint32_t)st.xdot[i];
  s.env_pos = (uint64_t)st.th[i];
  s.steps_remaining = st.steps_remaining[i];
  s.reset_count = st.reset_count[i];
}

__device__ __forceinline__ void chain_store(const EnvStateDev &st, uint32_t i, const ChainLane &s) {
  st.x[i] = (double)s.state;
  st.xdot[i] = (double)s.initial;
  st.th[i] = (double)s.env_pos;  // exact below 2^53 words
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

// rand 0.8.5 `gen_range(0..range)` for u64/usize (UniformInt::sample_single): widening multiply, accept when the low
// half is inside the zone `(range << leading_zeros(range)) - 1`; every attempt reads one u64 = stream words
// (pos, pos + 1), low word first.  The loop ends with probability 1; 64 attempts bound it (each fails w.p. <= 1/2).
__device__ __forceinline__ uint32_t lane_gen_range(const uint32_t *key, uint64_t glane, uint64_t &pos, uint64_t range) {
  const uint64_t zone = (range << __clzll((long long)range)) - 1;
  uint64_t hi = 0;
  for (int attempt = 0; attempt < 64; ++attempt) {
    uint32_t w[16];
    rl_chacha_block(key, pos >> 4, glane, 4, w);
    uint32_t lo32 = 0, hi32 = 0;
#pragma unroll
    for (int k = 0; k < 16; k += 2)
      if (k == (int)(pos & 15)) {
        lo32 = w[k];
        hi32 = w[k + 1];
      }
    pos += 2;
    const uint64_t v = ((uint64_t)hi32 << 32) | lo32;
    hi = __umul64hi(v, range);
    if (v * range <= zone) break;
  }
  return (uint32_t)hi;
}

__device__ __forceinline__ void chain_reset(const CartPoleDev &c, ChainLane &s, uint64_t glane) {
  if (c.mem_actions) {  // MemoryGame::initial_state (memory.rs:87-90)
    s.state = lane_gen_range(c.key_env, glane, s.env_pos, c.mem_actions);
    s.initial = s.state;
  } else {
    s.state = 0;  // Chain::initial_state (chain.rs:75-77), no random draw
  }
  s.steps_remaining = c.max_steps;
  s.reset_count += 1;
}

// Chain::step (chain.rs:83-105) / MemoryGame::step (memory.rs:96-114) + the step-limit tail; `word` is the lane's
// env-stream word for this global step (Chain's slip draw)
__device__ __forceinline__ int chain_step(const CartPoleDev &c, ChainLane &s, int action, uint32_t word,
                                          float &reward) {
  if (c.bandit) {  // Bandit::step (bandits.rs:66-77): Deterministic::sample draws nothing
    reward = c.bandit_r[action];
    return RL_SUCC_TERMINATE;
  }
  if (c.mem_actions) {
    if (s.state == c.chain_size - 1) {  // the last of num_actions + history_len states: the answer step
      reward = (uint32_t)action == s.initial ? 1.0f : -1.0f;
      return RL_SUCC_TERMINATE;  // passes through the step limit untouched (step_limit.rs:216-222)
    }
    s.state = s.state < c.mem_actions ? c.mem_actions : s.state + 1;
    reward = 0.0f;
  } else {
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

// The env side of the fused rollouts, for both env kinds (the policy side is chosen by the kernel)
struct ChainOps {
  using State = ChainLane;
  static __device__ __forceinline__ void load(const EnvStateDev &st, uint32_t i, State &s) { chain_load(st, i, s); }
  static __device__ __forceinline__ void store(const EnvStateDev &st, uint32_t i, const State &s) { chain_store(st, i, s); }
  template <int D>
  static __device__ __forceinline__ void features(const CartPoleDev &c, const State &s, float (&f)[D]) {
    chain_features<D>(c, s, f);
  }
  // Environment::step; the slip draw of global step `word` is word `word` of the lane's env stream
  static __device__ __forceinline__ int step(const CartPoleDev &c, State &s, int a, uint64_t glane, uint64_t word,
                                             float &reward) {
    return chain_step(c, s, a, stream_word(c.key_env, glane, word), reward);
  }
  static __device__ __forceinline__ void reset(const CartPoleDev &c, State &s, uint64_t glane) { chain_reset(c, s, glane); }
};

struct CartPoleOps {
  using State = LaneState;
  static __device__ __forceinline__ void load(const EnvStateDev &st, uint32_t i, State &s) { lane_load(st, i, s); }
  static __device__ __forceinline__ void store(const EnvStateDev &st, uint32_t i, const State &s) { lane_store(st, i, s); }
  template <int D>
  static __device__ __forceinline__ void features(const CartPoleDev &c, const State &s, float (&f)[D]) {
    cp_features<D>(c, s, f);
  }
  static __device__ __forceinline__ int step(const CartPoleDev &c, State &s, int a, uint64_t, uint64_t, float &reward) {
    reward = 1.0f;  // CartPole::step (cartpole.rs:140)
    return cp_step(c, s, a);
  }
  static __device__ __forceinline__ void reset(const CartPoleDev &c, State &s, uint64_t glane) { cp_reset(c, s, glane); }
};

__global__ void k_chain_reset(CartPoleDev c, EnvStateDev st, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ChainLane s;
  chain_load(st, i, s);
  chain_reset(c, s, c.lane_offset + i);
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
  if (succ != RL_SUCC_CONTINUE) chain_reset(c, s, c.lane_offset + i);
  chain_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) obs_next[(size_t)d * n + i] = f[d];
  reward[i] = r;
  flag[i] = (uint8_t)succ;
  chain_store(st, i, s);
}

// LDS of the forward kernels
constexpr int LSTM_KS_LDS = 15;  // k-steps (of GH / 4 = 32) of the LSTM's o gate whose W_hh operands wait in LDS
template <bool WITH_W1>
struct SeqFwdSharedT {
  float hT[2][GH][TLS];     // recurrent state, [k][m], double-buffered
  float uS[WITH_W1 ? 1 : TL][MH + 1];  // MLP hidden activations, [m][j] (LSTM: in the state buffer the step has finished reading)
  float xS[TL][8];          // observation features of the current step
  float w2S[2][MH];
  float outS[2][TL];
  int endS[TL];             // != 0: the lane's episode ended at this step (recurrent state restarts)
  int endS2[TL];            // the teacher-forced forward stages step t + 1 while step t's flags are still being read
  int peek;                 // != 0: some lane of the tile needs a successor evaluation at this step
  // LSTM only: the MLP's first layer as MFMA B operands, [wave][k-step][lane] (its four gate matrices fill the
  // register budget the GRU spends on three gates + this layer)
  float w1S[WITH_W1 ? 8 : 1][WITH_W1 ? GH / 4 : 1][WITH_W1 ? 64 : 1];
  // LSTM only: what else the register file does not hold next to four gates' W_hh operands — the first LSTM_KS_LDS
  // k-steps of the o gate's operands, [wave][k-step][lane], and the input projection with both biases, [gate][row][unit]
  float whS[WITH_W1 ? 8 : 1][WITH_W1 ? LSTM_KS_LDS : 1][WITH_W1 ? 64 : 1];
  float wiS[WITH_W1 ? 4 : 1][WITH_W1 ? 7 : 1][WITH_W1 ? GH : 1];  // rows 0..D-1: W_ih[.][d]; row 5: b_ih; row 6: b_hh
};
using SeqFwdShared = SeqFwdSharedT<false>;
using LstmFwdShared = SeqFwdSharedT<true>;

// ---------------------------------------------------------------- 16-unit ownership (v_mfma_f32_16x16x4_f32)
// The same cell with EIGHT waves per tile: wave w owns hidden units [16w, 16w+16) of every gate and of the MLP layer,
// ~180 registers, so two waves share a SIMD (workgroup of 512 threads, one per CU) and one wave's ds_reads / gate
// transcendentals run under the other's MFMAs.  16x16x4 lane maps (measured, scripts/probe/mfma16_arith.hip):
// A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15], C register i of lane l = C[4 (l >> 4) + i][l & 15]; the four
// k-products of an instruction and chained instructions form one sequential fma chain over k ascending, so the
// results are bit-identical to the 32x32x2 formulation and to the oracle.

template <int D>
struct SeqFwdWeights16 {
  float whh[3][GH / 4];
  float w1[GH / 4];
  float wih[3][D];
  float bih[3], bhh[3], b1;
};

template <int D>
__device__ __forceinline__ void seq_load_weights16(SeqFwdWeights16<D> &w, const GruParams &g, int wave, int lane) {
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
#pragma unroll
  for (int gte = 0; gte < 3; ++gte) {
    const int row = gte * GH + j;
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks) w.whh[gte][ks] = g.Whh[(size_t)row * GH + 4 * ks + g4];
#pragma unroll
    for (int d = 0; d < D; ++d) w.wih[gte][d] = g.Wih[(size_t)row * D + d];
    w.bih[gte] = g.bih[row];
    w.bhh[gte] = g.bhh[row];
  }
#pragma unroll
  for (int ks = 0; ks < GH / 4; ++ks) w.w1[ks] = g.W1[(size_t)j * GH + 4 * ks + g4];
  w.b1 = g.b1[j];
}

// One cell + head evaluation for the tile.  Reads the state from sh.hT[cur] and `hown`, writes the new state to
// sh.hT[cur ^ 1] and `hnew`; the head outputs land in sh.outS (valid after the function returns: it ends with a
// barrier).  `store` != nullptr: record the 7 activation arrays of this (t, tile) block ([half][unit][16] arrays, rec_at: a lane's
// four samples of an M-tile are contiguous, one 16-byte store per array and M-tile).
template <int D, int A, bool FINAL_BARRIER = true>
__device__ __forceinline__ void seq_cell16(SeqFwdShared &sh, const SeqFwdWeights16<D> &w, int cur,
                                           const float (&hown)[8], float (&hnew)[8], float b2_mine,
                                           float *__restrict__ store, int wave, int lane) {
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16, nxt = cur ^ 1;
  f32x4 acc[3][2];
#pragma unroll
  for (int gte = 0; gte < 3; ++gte)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[gte][mt] = (f32x4){w.bhh[gte], w.bhh[gte], w.bhh[gte], w.bhh[gte]};
#pragma unroll
  for (int ks = 0; ks < GH / 4; ++ks) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float a = sh.hT[cur][4 * ks + g4][16 * mt + n16];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte)
        acc[gte][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.whh[gte][ks], acc[gte][mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 rv, zv, nv, gv, pv, av;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = acc16_row(mt, i, g4), r = 4 * mt + i;
      float gi[3];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) {
        float v = w.bih[gte];
#pragma unroll
        for (int d = 0; d < D; ++d) v = __builtin_fmaf(sh.xS[m][d], w.wih[gte][d], v);
        gi[gte] = v;
      }
      const float rr = rl_sigmoidf(acc[0][mt][i] + gi[0]);
      const float zz = rl_sigmoidf(acc[1][mt][i] + gi[1]);
      const float rn = acc[2][mt][i] * rr;
      const float nn = rl_tanhf(gi[2] + rn);
      const float dn = hown[r] - nn;
      const float hz = dn * zz;
      const float hv = hz + nn;
      hnew[r] = hv;
      sh.hT[nxt][j][m] = hv;
      rv[i] = rr;
      zv[i] = zz;
      nv[i] = nn;
      gv[i] = acc[2][mt][i];
      pv[i] = hown[r];
      av[i] = hv > 0.0f ? hv : 0.0f;
    }
    if (store != nullptr) {
      float *__restrict__ row = store + rec_at(j, 16 * mt + 4 * g4);
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_R * GH * TL) = rv;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_Z * GH * TL) = zv;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_N * GH * TL) = nv;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_GHN * GH * TL) = gv;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_HPREV * GH * TL) = pv;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_A1 * GH * TL) = av;
    }
  }
  __syncthreads();
  f32x4 acc1[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) acc1[mt] = (f32x4){w.b1, w.b1, w.b1, w.b1};
#pragma unroll
  for (int ks = 0; ks < GH / 4; ++ks) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float a = sh.hT[nxt][4 * ks + g4][16 * mt + n16];
      a = a > 0.0f ? a : 0.0f;  // Chain activation between the modules (chain.rs:165)
      acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.w1[ks], acc1[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 uv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = acc16_row(mt, i, g4);
      const float u = acc1[mt][i] > 0.0f ? acc1[mt][i] : 0.0f;
      sh.uS[m][j] = u;
      uv[i] = u;
    }
    if (store != nullptr)
      *reinterpret_cast<f32x4 *>(store + (size_t)ACT_U * GH * TL + rec_at(j, 16 * mt + 4 * g4)) = uv;
  }
  __syncthreads();
  if (wave == 0 && (lane >> 5) < A) {
    const int n = lane & 31, hf = lane >> 5;
    float z = b2_mine;
#pragma unroll 8
    for (int q = 0; q < MH; ++q) z = __builtin_fmaf(sh.uS[n][q], sh.w2S[hf][q], z);
    sh.outS[hf][n] = z;
  }
  // the head's outputs are read by lanes of wave 0 only (the env / io lanes): a caller that syncs the workgroup later
  // anyway needs just the in-wave ordering of these LDS writes
  if (FINAL_BARRIER) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------- LSTM cell (seq/rnn/lstm.rs:17-51), same ownership
// Wave w owns units [16w, 16w+16) of the four gates [i; f; g; o]: 4 x 32 B operands of W_hh in registers; the MLP's
// first layer comes from LDS (w1S).  Per-lane state: h (registers 0..7) and the cell state c (registers 8..15), both
// zero at the start of an episode (LstmImpl::initial_cell_state, lstm.rs:22-31); only h is shared through LDS.
//   i = sigmoid(.), f = sigmoid(.), g = tanh(.), o = sigmoid(.);  c' = f * c + i * g;  h' = o * tanh(c')
// with every pre-activation = (b_hh + W_hh h) + (b_ih + W_ih x), the same fma chains as the GRU's r and z gates.
template <int D>
struct LstmFwdWeights16 {
  float whh[3][GH / 4];                 // gates i, f, g
  float whh_o[GH / 4 - LSTM_KS_LDS];    // gate o, k-steps LSTM_KS_LDS .. (the first ones: LstmFwdShared::whS)
  float b1;
};

template <int D>
__device__ __forceinline__ void lstm_load_weights16(LstmFwdWeights16<D> &w, const GruParams &g, LstmFwdShared &sh,
                                                    int wave, int lane) {
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
#pragma unroll
  for (int gte = 0; gte < 4; ++gte) {
    const int row = gte * GH + j;
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks) {
      const float v = g.Whh[(size_t)row * GH + 4 * ks + g4];
      if (gte < 3) w.whh[gte < 3 ? gte : 0][ks] = v;
      else if (ks >= LSTM_KS_LDS) w.whh_o[ks >= LSTM_KS_LDS ? ks - LSTM_KS_LDS : 0] = v;
      else sh.whS[wave][ks < LSTM_KS_LDS ? ks : 0][lane] = v;
    }
    if (g4 == 0) {  // one lane group per unit fills the shared input projection
#pragma unroll
      for (int d = 0; d < D; ++d) sh.wiS[gte][d][j] = g.Wih[(size_t)row * D + d];
      sh.wiS[gte][5][j] = g.bih[row];
      sh.wiS[gte][6][j] = g.bhh[row];
    }
  }
  for (int ks = 0; ks < GH / 4; ++ks) sh.w1S[wave][ks][lane] = g.W1[(size_t)j * GH + 4 * ks + g4];
  w.b1 = g.b1[j];
}

template <int D, int A, bool FINAL_BARRIER = true>
__device__ __forceinline__ void lstm_cell16(LstmFwdShared &sh, const LstmFwdWeights16<D> &w, int cur,
                                            const float (&sown)[16], float (&snew)[16], float b2_mine,
                                            float *__restrict__ store, int wave, int lane) {
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16, nxt = cur ^ 1;
  f32x4 acc[4][2];
#pragma unroll
  for (int gte = 0; gte < 4; ++gte)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float b = sh.wiS[gte][6][j];
      acc[gte][mt] = (f32x4){b, b, b, b};
    }
#pragma unroll
  for (int ks = 0; ks < GH / 4; ++ks) {
    const float wo = ks < LSTM_KS_LDS ? sh.whS[wave][ks < LSTM_KS_LDS ? ks : 0][lane]
                                      : w.whh_o[ks >= LSTM_KS_LDS ? ks - LSTM_KS_LDS : 0];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float a = sh.hT[cur][4 * ks + g4][16 * mt + n16];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte)
        acc[gte][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.whh[gte][ks], acc[gte][mt], 0, 0, 0);
      acc[3][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wo, acc[3][mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 iv, fv, gv, ov, pv, av, cv, tv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = acc16_row(mt, i, g4), r = 4 * mt + i;
      float pre[4];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) {
        float v = sh.wiS[gte][5][j];
#pragma unroll
        for (int d = 0; d < D; ++d) v = __builtin_fmaf(sh.xS[m][d], sh.wiS[gte][d][j], v);
        pre[gte] = acc[gte][mt][i] + v;
      }
      const float ig = rl_sigmoidf(pre[0]), fg = rl_sigmoidf(pre[1]), gg = rl_tanhf(pre[2]), og = rl_sigmoidf(pre[3]);
      const float cprev = sown[8 + r];
      const float fc = fg * cprev;
      const float iga = ig * gg;
      const float cn = fc + iga;
      const float tc = rl_tanhf(cn);
      const float hv = og * tc;
      snew[r] = hv;
      snew[8 + r] = cn;
      sh.hT[nxt][j][m] = hv;
      iv[i] = ig;
      fv[i] = fg;
      gv[i] = gg;
      ov[i] = og;
      pv[i] = sown[r];
      av[i] = hv > 0.0f ? hv : 0.0f;
      cv[i] = cprev;
      tv[i] = tc;
    }
    if (store != nullptr) {
      float *__restrict__ row = store + rec_at(j, 16 * mt + 4 * g4);
      *reinterpret_cast<f32x4 *>(row + (size_t)LACT_I * GH * TL) = iv;
      *reinterpret_cast<f32x4 *>(row + (size_t)LACT_F * GH * TL) = fv;
      *reinterpret_cast<f32x4 *>(row + (size_t)LACT_G * GH * TL) = gv;
      *reinterpret_cast<f32x4 *>(row + (size_t)LACT_O * GH * TL) = ov;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_HPREV * GH * TL) = pv;
      *reinterpret_cast<f32x4 *>(row + (size_t)ACT_A1 * GH * TL) = av;
      *reinterpret_cast<f32x4 *>(row + (size_t)LACT_CPREV * GH * TL) = cv;
      *reinterpret_cast<f32x4 *>(row + (size_t)LACT_TC * GH * TL) = tv;
    }
  }
  __syncthreads();
  // every wave has read h(t) by now: its buffer holds the MLP's hidden activations [m][j] until the next step's gates
  // write h(t+2) there (after the workgroup barrier that ends this step)
  float *__restrict__ uS = &sh.hT[cur][0][0];
  static_assert(TL * (MH + 1) <= GH * TLS, "the head's activations fit the state buffer");
  f32x4 acc1[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) acc1[mt] = (f32x4){w.b1, w.b1, w.b1, w.b1};
#pragma unroll 8
  for (int ks = 0; ks < GH / 4; ++ks) {
    const float b = sh.w1S[wave][ks][lane];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float a = sh.hT[nxt][4 * ks + g4][16 * mt + n16];
      a = a > 0.0f ? a : 0.0f;  // Chain activation between the modules (chain.rs:165)
      acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 uv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = acc16_row(mt, i, g4);
      const float u = acc1[mt][i] > 0.0f ? acc1[mt][i] : 0.0f;
      uS[m * (MH + 1) + j] = u;
      uv[i] = u;
    }
    if (store != nullptr)
      *reinterpret_cast<f32x4 *>(store + (size_t)ACT_U * GH * TL + rec_at(j, 16 * mt + 4 * g4)) = uv;
  }
  __syncthreads();
  if (wave == 0 && (lane >> 5) < A) {
    const int n = lane & 31, hf = lane >> 5;
    float z = b2_mine;
#pragma unroll 8
    for (int q = 0; q < MH; ++q) z = __builtin_fmaf(uS[n * (MH + 1) + q], sh.w2S[hf][q], z);
    sh.outS[hf][n] = z;
  }
  // the head's outputs are read by lanes of wave 0 only (the env / io lanes): a caller that syncs the workgroup later
  // anyway needs just the in-wave ordering of these LDS writes
  if (FINAL_BARRIER) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// The two cells behind one interface for the rollout / teacher-forced forward loops.  NS = per-lane state registers
// (the first eight are h, mirrored in LDS; the LSTM's second eight are c).
struct GruCell16 {
  static constexpr int NG = 3, NS = 8;
  using Shared = SeqFwdShared;
  template <int D>
  using Weights = SeqFwdWeights16<D>;
  template <int D>
  static __device__ __forceinline__ void load(Weights<D> &w, const GruParams &g, Shared &, int wave, int lane) {
    seq_load_weights16<D>(w, g, wave, lane);
  }
  template <int D, int A, bool FINAL_BARRIER = true>
  static __device__ __forceinline__ void cell(Shared &sh, const Weights<D> &w, int cur, const float (&sown)[NS],
                                              float (&snew)[NS], float b2_mine, float *__restrict__ store, int wave,
                                              int lane) {
    seq_cell16<D, A, FINAL_BARRIER>(sh, w, cur, sown, snew, b2_mine, store, wave, lane);
  }
};

struct LstmCell16 {
  static constexpr int NG = 4, NS = 16;
  using Shared = LstmFwdShared;
  template <int D>
  using Weights = LstmFwdWeights16<D>;
  template <int D>
  static __device__ __forceinline__ void load(Weights<D> &w, const GruParams &g, Shared &sh, int wave, int lane) {
    lstm_load_weights16<D>(w, g, sh, wave, lane);
  }
  template <int D, int A, bool FINAL_BARRIER = true>
  static __device__ __forceinline__ void cell(Shared &sh, const Weights<D> &w, int cur, const float (&sown)[NS],
                                              float (&snew)[NS], float b2_mine, float *__restrict__ store, int wave,
                                              int lane) {
    lstm_cell16<D, A, FINAL_BARRIER>(sh, w, cur, sown, snew, b2_mine, store, wave, lane);
  }
};

// ---------------------------------------------------------------- rollout (recurrent policy, either env kind)
// PolicyActor::act over SeqIterative::step (policies/actor.rs:42-55; chain.rs:175-186) for T steps of every lane.
// The episode state starts at zero at the beginning of the launch and after every episode end.
template <int D, class Env, class Cell>
__global__ void __launch_bounds__(W16 * 64, 2) k_rollout_gru(CartPoleDev c, EnvStateDev st, TrajDev tr,
                                                              const float *__restrict__ params, uint64_t t_global) {
  constexpr int A = 2, NS = Cell::NS;
  __shared__ typename Cell::Shared sh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, hf = lane >> 5, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t lane0 = blockIdx.x * TL;
  const GruParams g = seq_params(params, D, A, Cell::NG);
  typename Cell::template Weights<D> w;
  Cell::template load<D>(w, g, sh, wave, lane);
  for (int q = threadIdx.x; q < A * MH; q += W16 * 64) sh.w2S[q / MH][q % MH] = g.W2[q];
  for (int q = threadIdx.x; q < 2 * GH * TLS; q += W16 * 64) (&sh.hT[0][0][0])[q] = 0.0f;
  const float b2_mine = hf < A ? g.b2[hf] : 0.0f;
  float hown[NS];
#pragma unroll
  for (int r = 0; r < NS; ++r) hown[r] = 0.0f;
  // env state of the tile: lanes 0..31 of wave 0
  const bool env_lane = wave == 0 && lane < TL;
  const uint32_t i = lane0 + (uint32_t)lane;
  const uint64_t glane = c.lane_offset + i;
  typename Env::State s{};
  const size_t plane = (size_t)(T + 1) * N;
  if (env_lane) {
    Env::load(st, i, s);
    float f[D];
    Env::template features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      sh.xS[lane][d] = f[d];
      tr.obs[d * plane + i] = f[d];
    }
  }
  __syncthreads();
  int cur = 0;
  for (uint32_t t = 0; t < T; ++t) {
    float hnew[NS];
    Cell::template cell<D, A>(sh, w, cur, hown, hnew, b2_mine, nullptr, wave, lane);
    if (env_lane) {
      float z[2] = {sh.outS[0][lane], sh.outS[1][lane]}, lp[2];
      log_softmax_lane<2>(z, lp);
      const uint64_t word = t_global + t;
      const float u = rl_u32_to_unit_f32(stream_word(c.key_actor, glane, word));
      const int a = categorical_sample_lane<2>(lp, u);
      float rew;
      const int succ = Env::step(c, s, a, glane, word, rew);
      const size_t o = (size_t)t * N + i;
      tr.action[o] = (uint8_t)a;
      tr.reward[o] = rew;
      tr.flag[o] = (uint8_t)succ;
      float f[D];
      if (succ == RL_SUCC_INTERRUPT) {
        Env::template features<D>(c, s, f);
#pragma unroll
        for (int d = 0; d < D; ++d) tr.term_obs[(size_t)d * T * N + o] = f[d];
      }
      if (succ != RL_SUCC_CONTINUE) Env::reset(c, s, glane);
      Env::template features<D>(c, s, f);
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
    for (int r = 0; r < NS; ++r) {
      const int m = acc16_row((r & 7) >> 2, r & 3, g4);
      const bool ended = sh.endS[m] != 0;
      hown[r] = ended ? 0.0f : hnew[r];
      if (ended && r < 8) sh.hT[nxt][j][m] = 0.0f;
    }
    __syncthreads();
    cur = nxt;
  }
  if (env_lane) Env::store(st, i, s);
}

// ---------------------------------------------------------------- rollout (Chain lanes, feed-forward policy)
// k_rollout_cartpole's lane-per-thread loop (kernels_rollout.hip) over Chain: features -> in-lane MLP -> log-softmax
// -> inverse-CDF sample with the lane's actor word -> Chain::step with the lane's env word -> step limit -> reset.
template <int D, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_rollout_chain_mlp(CartPoleDev c, EnvStateDev st, TrajDev tr,
                                                             const float *__restrict__ policy, int H,
                                                             uint64_t t_global) {
  __shared__ __attribute__((aligned(16))) float pk[MLP_PK_FLOATS];  // the policy, one 8-float record per hidden unit
  const uint32_t n = tr.n, T = tr.T;
  mlp_pack_lds<D>(pk, policy, H, threadIdx.x, BLOCK);
  __syncthreads();
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint64_t glane = c.lane_offset + i;
  ChainLane s;
  chain_load(st, i, s);
  const size_t plane = (size_t)(T + 1) * n;
  for (uint32_t t = 0; t < T; ++t) {
    float f[D];
    chain_features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) tr.obs[d * plane + (size_t)t * n + i] = f[d];
    const uint64_t word = t_global + t;
    const float u = rl_u32_to_unit_f32(stream_word(c.key_actor, glane, word));
    float z[2], lp[2];
    mlp_forward_lane_lds<D>(pk, H, f, z);
    log_softmax_lane<2>(z, lp);
    const int a = categorical_sample_lane<2>(lp, u);
    float rew;
    const int succ = chain_step(c, s, a, stream_word(c.key_env, glane, word), rew);
    const size_t o = (size_t)t * n + i;
    tr.action[o] = (uint8_t)a;
    tr.reward[o] = rew;
    tr.flag[o] = (uint8_t)succ;
    if (succ == RL_SUCC_INTERRUPT) {
      chain_features<D>(c, s, f);
#pragma unroll
      for (int d = 0; d < D; ++d) tr.term_obs[(size_t)d * T * n + o] = f[d];
    }
    if (succ != RL_SUCC_CONTINUE) chain_reset(c, s, glane);
  }
  float f[D];
  chain_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) tr.obs[d * plane + (size_t)T * n + i] = f[d];
  chain_store(st, i, s);
}

// ---------------------------------------------------------------- teacher-forced forward over a trajectory
// SeqPacked::seq_packed of Chain<Gru, Mlp> (chain.rs:151-161; gru.rs:76-98) on the lane layout.  Writes the module
// outputs out[a][t][lane]; optionally the outputs at the successor observations of cut episodes (extended
// observation sequences, features.rs:132-178) and the activation record for the backward pass.
template <int D, int A, class Cell>
__global__ void __launch_bounds__(W16 * 64, 2) k_gru_seq_forward(TrajDev tr, const float *__restrict__ params,
                                                            float *__restrict__ out, float *__restrict__ succ_out,
                                                            float *__restrict__ act,
                                                            const int32_t *__restrict__ skip) {
  constexpr int NS = Cell::NS;
  __shared__ typename Cell::Shared sh;
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, hf = lane >> 5, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = seq_params(params, D, A, Cell::NG);
  typename Cell::template Weights<D> w;
  Cell::template load<D>(w, g, sh, wave, lane);
  for (int q = threadIdx.x; q < A * MH; q += W16 * 64) sh.w2S[q / MH][q % MH] = g.W2[q];
  for (int q = threadIdx.x; q < 2 * GH * TLS; q += W16 * 64) (&sh.hT[0][0][0])[q] = 0.0f;
  const float b2_mine = hf < A ? g.b2[hf] : 0.0f;
  float hown[NS];
#pragma unroll
  for (int r = 0; r < NS; ++r) hown[r] = 0.0f;
  const bool io_lane = wave == 0 && lane < TL;
  const uint32_t i = lane0 + (uint32_t)lane;
  const size_t plane = (size_t)(T + 1) * N;
  int cur = 0, flag = 0;
  bool peeking = false, need = false;
  uint32_t t = 0;
  // One cell evaluation per iteration.  A step whose episode is cut in some lane of the tile is followed by a
  // "peek" iteration that evaluates the successor observation from the post-step state without advancing it.
  // Two workgroup barriers per iteration: after the io lanes have seen the outputs (peek decision published), and
  // after the commit, which also stages the next step's observation and flags (flags double-buffered by step parity:
  // the commit of step t still reads its own while step t + 1's are written).
  auto endbuf = [&](uint32_t tt) -> int * { return (tt & 1) ? sh.endS2 : sh.endS; };
  auto stage = [&](uint32_t tt) {  // io lanes: observation and successor code of step tt
#pragma unroll
    for (int d = 0; d < D; ++d) sh.xS[lane][d] = tr.obs[d * plane + (size_t)tt * N + i];
    flag = tr.flag[(size_t)tt * N + i];
    endbuf(tt)[lane] = flag != RL_SUCC_CONTINUE;
  };
  if (io_lane) stage(0);
  if (threadIdx.x == 0) sh.peek = 0;
  __syncthreads();
  while (t < T) {
    float hout[NS];
    float *store = (act != nullptr && !peeking) ? act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL : nullptr;
    Cell::template cell<D, A, false>(sh, w, cur, hown, hout, b2_mine, store, wave, lane);
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
      for (int r = 0; r < NS; ++r) hown[r] = hout[r];
      cur ^= 1;
      peeking = true;
    } else {
      // commit: restart the state of lanes whose episode ended at step t
      const int buf = peeking ? cur : (cur ^ 1);
      const int *ends = endbuf(t);
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const int m = acc16_row((r & 7) >> 2, r & 3, g4);
        const bool ended = ends[m] != 0;
        const float hv = peeking ? hown[r] : hout[r];
        hown[r] = ended ? 0.0f : hv;
        // (after a peek the LDS image of the post-step state is written again from the owners' registers: the LSTM cell
        // parks the head's activations in the buffer it has just read — the peek's input, i.e. exactly that state; found
        // in round 4 with histories whose lanes are interrupted at different steps)
        if ((ended || peeking) && r < 8) sh.hT[buf][j][m] = hown[r];
      }
      cur = buf;
      // every thread has `peeking` in a register: the flag in LDS may be cleared while others still evaluate the branch
      if (peeking && threadIdx.x == 0) sh.peek = 0;
      peeking = false;
      t += 1;
      if (io_lane && t < T) stage(t);
    }
    __syncthreads();
  }
}


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
  // the inputs of 16 steps are requested before the chain walks through them (k_gae_scan, kernels_rollout.hip: one
  // memory round trip per 16 steps instead of one per step)
  constexpr int AHEAD = 16;
  const uint8_t *__restrict__ flag_in = tr.flag;
  const float *__restrict__ reward_in = tr.reward;
  for (uint32_t hi = T; hi > 0;) {
    uint8_t fb[AHEAD];
    float rb[AHEAD], vb[AHEAD], sb[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      const size_t o = (size_t)(hi > (uint32_t)u ? hi - 1 - (uint32_t)u : 0u) * n + i;
      fb[u] = flag_in[o];
      rb[u] = reward_in[o];
      vb[u] = values[o];
      sb[u] = succ[o];
    }
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) {
      if (hi > (uint32_t)u) {
        const uint32_t t = hi - 1 - (uint32_t)u;
        const size_t o = (size_t)t * n + i;
        const uint8_t f = fb[u];
        const float r = rb[u], v = vb[u];
        float vn;
        bool ends;
        if (f == RL_SUCC_TERMINATE) {
          vn = 0.0f;
          ends = true;
        } else if (f == RL_SUCC_INTERRUPT || t == T - 1) {
          vn = sb[u];
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
    hi = hi > (uint32_t)AHEAD ? hi - AHEAD : 0u;
  }
}


// one_step_values (critics/mod.rs:139-150) with a recurrent critic: next values from the teacher-forced forward
// (values [T][n]; succ [T][n] where an episode is cut), same selection rule as k_seq_gae; also mirrors the values
__global__ void __launch_bounds__(256) k_seq_value_targets_td(TrajDev tr, const float *__restrict__ values,
                                                              const float *__restrict__ succ, float gamma) {
  const uint32_t n = tr.n, T = tr.T;
  const size_t B = (size_t)T * n;
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= B) return;
  const uint32_t t = (uint32_t)(o / n);
  const uint8_t f = tr.flag[o];
  float vn;
  if (f == RL_SUCC_TERMINATE) vn = 0.0f;
  else if (f == RL_SUCC_INTERRUPT || t == T - 1) vn = succ[o];
  else vn = values[o + n];
  const float dn = gamma * vn;
  tr.tgt[o] = tr.reward[o] + dn;
  tr.values[o] = values[o];
  if (t == T - 1) tr.values[o + n] = succ[o];
}

void launch_seq_value_targets(rl_traj *traj, float gamma) {
  ProfScope ps(traj->eng, RL_K_GAE);
  const size_t B = (size_t)traj->d.T * traj->d.n;
  hipLaunchKernelGGL(k_seq_value_targets_td, dim3(cdiv_s(B, 256)), dim3(256), 0, traj->eng->stream, traj->d,
                     traj->seq.out, traj->seq.succ, gamma);
}

void launch_seq_gae(rl_traj *traj, float gamma, float lambda) {
  traj->rtg_scan_valid = false;  // (this scan writes the return plane too; only k_gae_scan's is vouched for, engine.hpp)
  ProfScope ps(traj->eng, RL_K_GAE);
  uint32_t n = traj->d.n;
  hipLaunchKernelGGL(k_seq_gae, dim3(cdiv_s(n, 64)), dim3(64), 0, traj->eng->stream, traj->d, traj->seq.out,
                     traj->seq.succ, gamma, lambda);
}

// ---------------------------------------------------------------- other widths embedded in the built shape
// index in the padded (5 -> GH -> MH) layout of element i of a chain's flat vector with logical widths D, H (recurrent),
// H2 (MLP hidden), A outputs and NG gates — RnnWeights order: w_ih [NG H, D], w_hh [NG H, H], b_ih, b_hh, then the MLP
__device__ __forceinline__ uint32_t seq_pad_index(uint32_t i, uint32_t D, uint32_t H, uint32_t H2, uint32_t A, uint32_t NG) {
  const uint32_t nWih = NG * H * D, nWhh = NG * H * H, nb = NG * H, nW1 = H2 * H, nW2 = A * H2;
  const uint32_t xWhh = NG * GH * 5, xbih = xWhh + NG * GH * GH, xbhh = xbih + NG * GH, xW1 = xbhh + NG * GH;
  const uint32_t xb1 = xW1 + MH * GH, xW2 = xb1 + MH, xb2 = xW2 + A * MH;
  if (i < nWih) {
    const uint32_t row = i / D, d = i % D;
    return ((row / H) * GH + row % H) * 5 + d;
  }
  i -= nWih;
  if (i < nWhh) {
    const uint32_t row = i / H, k = i % H;
    return xWhh + ((row / H) * GH + row % H) * GH + k;
  }
  i -= nWhh;
  if (i < nb) return xbih + (i / H) * GH + i % H;
  i -= nb;
  if (i < nb) return xbhh + (i / H) * GH + i % H;
  i -= nb;
  if (i < nW1) return xW1 + (i / H) * GH + i % H;
  i -= nW1;
  if (i < H2) return xb1 + i;
  i -= H2;
  if (i < nW2) return xW2 + (i / H2) * MH + i % H2;
  return xb2 + (i - nW2);
}
template <bool GATHER>
__global__ void k_seq_pad(const float *__restrict__ src, float *__restrict__ dst, uint32_t P, uint32_t D, uint32_t H,
                          uint32_t H2, uint32_t A, uint32_t NG) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const uint32_t x = seq_pad_index(i, D, H, H2, A, NG);
  if (GATHER) dst[i] = src[x];
  else dst[x] = src[i];
}
void launch_seq_pad(const rl_mlp *real, float *exec_dst, const float *real_src) {
  const uint32_t P = (uint32_t)real->P;
  hipLaunchKernelGGL(k_seq_pad<false>, dim3((P + 255) / 256), dim3(256), 0, real->eng->stream, real_src, exec_dst, P,
                     real->in_dim, real->gru_hidden, real->hidden, real->out_dim, (uint32_t)rl_module_gates(real->kind));
}
void launch_seq_unpad(const rl_mlp *real, const float *exec_src, float *real_dst) {
  const uint32_t P = (uint32_t)real->P;
  hipLaunchKernelGGL(k_seq_pad<true>, dim3((P + 255) / 256), dim3(256), 0, real->eng->stream, exec_src, real_dst, P,
                     real->in_dim, real->gru_hidden, real->hidden, real->out_dim, (uint32_t)rl_module_gates(real->kind));
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

void launch_rollout_gru(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  RL_REQUIRE(!policy->lane_kernels(), "this module rolls out through launch_stack_rollout");
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  RL_REQUIRE(env->D == 5, "recurrent rollout: built for 5 observation features");
  uint32_t tiles = traj->d.n / TL;
#define ROLL(ENV, CELL)                                                                                           \
  hipLaunchKernelGGL((k_rollout_gru<5, ENV, CELL>), dim3(tiles), dim3(W16 * 64), 0, env->eng->stream, env->dev, env->st, \
                     traj->d, policy->d_params, env->t_global)
  const bool lstm = policy->kind == RL_MODULE_LSTM_MLP;
  if (env->kind != RL_ENV_CARTPOLE) {
    if (lstm) ROLL(ChainOps, LstmCell16);
    else ROLL(ChainOps, GruCell16);
  } else {
    if (lstm) ROLL(CartPoleOps, LstmCell16);
    else ROLL(CartPoleOps, GruCell16);
  }
#undef ROLL
}

void launch_rollout_chain_mlp(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  RL_REQUIRE(env->D == 5, "Chain rollout: built for the 5 one-hot features (latent or no step limit)");
  uint32_t n = (uint32_t)env->cfg.n_lanes;
  hipLaunchKernelGGL((k_rollout_chain_mlp<5, 64>), dim3(cdiv_s(n, 64)), dim3(64), 0, env->eng->stream, env->dev, env->st,
                     traj->d, policy->d_params, (int)policy->hidden, env->t_global);
}

void launch_gru_seq_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_succ, float *d_act,
                            const int32_t *d_skip) {
  if (mod->lane_kernels()) return launch_stack_forward(traj, mod, d_out, d_succ, d_act != nullptr, d_skip);
  ProfScope ps(traj->eng, RL_K_POLICY_FUSED);
  RL_REQUIRE(traj->d.D == 5, "recurrent forward: built for 5 observation features");
  uint32_t tiles = traj->d.n / TL;
#define FWD(AA, CELL)                                                                                        \
  hipLaunchKernelGGL((k_gru_seq_forward<5, AA, CELL>), dim3(tiles), dim3(W16 * 64), 0, traj->eng->stream, traj->d, \
                     mod->d_params, d_out, d_succ, d_act, d_skip)
  const bool lstm = mod->kind == RL_MODULE_LSTM_MLP;
  // training forward of the GRU chain (activation record, no successor evaluations): the bf16-pipe kernels of
  // kernels_seq_train.hip; kernel variant 1 keeps the f32 cell for A/B runs
  if (!lstm && d_act != nullptr && d_succ == nullptr && traj->eng->kernel_variant != 1) {
    launch_gru_train_forward(traj, mod, d_out, d_act, d_skip);
    return;
  }
  // ... and of the LSTM chain: its recurrence on the bf16 pipe (four waves per tile), then the same head kernel
  if (lstm && d_act != nullptr && d_succ == nullptr && traj->eng->kernel_variant != 1) {
    launch_lstm_train_forward(traj, mod, d_out, d_act, d_skip);
    return;
  }
  if (mod->out_dim == 2) {
    if (lstm) FWD(2, LstmCell16);
    else FWD(2, GruCell16);
  } else {
    if (lstm) FWD(1, LstmCell16);
    else FWD(1, GruCell16);
  }
#undef FWD
}
