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
constexpr int TLS = TL + 16;  // LDS row stride of the [k][m] operand buffers read as 16x16x4 A operands: the four
                              // k-rows of an instruction start 48 floats apart = banks 0 / 48 / 32 / 16: no conflict
// Per (step, tile) block the training forward records SEQ_ARR [unit][lane] arrays and the backward DPRE_ARR (the strides
// are those of the LSTM, the larger of the two cells; kernels.hpp: RL_SEQ_ACT_ARRAYS / RL_SEQ_DPRE_ARRAYS).
//   GRU  record: r, z, n, gh_n, h_prev, relu(h'), u                        backward: d pre_r, d pre_z, d pre_n, d pre_n * r, d u_pre
//   LSTM record: i, f, g, o, h_prev, relu(h'), u, c_prev, tanh(c')         backward: d pre_i, d pre_f, d pre_g, d pre_o, d u_pre, d relu(h')
constexpr int SEQ_ARR = RL_SEQ_ACT_ARRAYS;
constexpr int DPRE_ARR = RL_SEQ_DPRE_ARRAYS;
enum { ACT_R = 0, ACT_Z = 1, ACT_N = 2, ACT_GHN = 3, ACT_HPREV = 4, ACT_A1 = 5, ACT_U = 6 };
enum { LACT_I = 0, LACT_F = 1, LACT_G = 2, LACT_O = 3, LACT_CPREV = 7, LACT_TC = 8 };  // 4, 5, 6 as above
enum { DPRE_DU = 4, DPRE_DA1 = 5 };

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

// ---------------------------------------------------------------- GRU-MLP parameter views
struct GruParams {
  const float *Wih, *Whh, *bih, *bhh, *W1, *b1, *W2, *b2;
};

// RnnWeights flat order (seq/rnn/mod.rs:223-257): w_ih [NG H, D], w_hh [NG H, H], b_ih, b_hh, then the MLP's two Linear
// layers; NG = RnnImpl::GATES_MULTIPLE: 3 for the GRU ([r; z; n]), 4 for the LSTM ([i; f; g; o])
__host__ __device__ inline GruParams seq_params(const float *p, int D, int A, int NG) {
  GruParams g;
  g.Wih = p;
  g.Whh = g.Wih + NG * GH * D;
  g.bih = g.Whh + NG * GH * GH;
  g.bhh = g.bih + NG * GH;
  g.W1 = g.bhh + NG * GH;
  g.b1 = g.W1 + MH * GH;
  g.W2 = g.b1 + MH;
  g.b2 = g.W2 + A * MH;
  return g;
}
__host__ __device__ inline GruParams gru_params(const float *p, int D, int A) { return seq_params(p, D, A, 3); }

__device__ __forceinline__ int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// LDS of the forward kernels
template <bool WITH_W1>
struct SeqFwdSharedT {
  float hT[2][GH][TLS];     // recurrent state, [k][m], double-buffered
  float uS[TL][MH + 1];     // MLP hidden activations, [m][j]
  float xS[TL][8];          // observation features of the current step
  float w2S[2][MH];
  float outS[2][TL];
  int endS[TL];             // != 0: the lane's episode ended at this step (recurrent state restarts)
  int peek;                 // != 0: some lane of the tile needs a successor evaluation at this step
  // LSTM only: the MLP's first layer as MFMA B operands, [wave][k-step][lane] (its four gate matrices fill the
  // register budget the GRU spends on three gates + this layer)
  float w1S[WITH_W1 ? 8 : 1][WITH_W1 ? GH / 4 : 1][WITH_W1 ? 64 : 1];
};
using SeqFwdShared = SeqFwdSharedT<false>;
using LstmFwdShared = SeqFwdSharedT<true>;

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

// ---------------------------------------------------------------- 16-unit ownership (v_mfma_f32_16x16x4_f32)
// The same cell with EIGHT waves per tile: wave w owns hidden units [16w, 16w+16) of every gate and of the MLP layer,
// ~180 registers, so two waves share a SIMD (workgroup of 512 threads, one per CU) and one wave's ds_reads / gate
// transcendentals run under the other's MFMAs.  16x16x4 lane maps (measured, scripts/probe/mfma16_arith.hip):
// A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15], C register i of lane l = C[4 (l >> 4) + i][l & 15]; the four
// k-products of an instruction and chained instructions form one sequential fma chain over k ascending, so the
// results are bit-identical to the 32x32x2 formulation and to the oracle.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int W16 = 8;  // waves per tile

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

// sample owned by accumulator register i of M-tile mt in lane group g4
__device__ __forceinline__ int acc16_row(int mt, int i, int g4) { return 16 * mt + 4 * g4 + i; }

// One cell + head evaluation for the tile.  Reads the state from sh.hT[cur] and `hown`, writes the new state to
// sh.hT[cur ^ 1] and `hnew`; the head outputs land in sh.outS (valid after the function returns: it ends with a
// barrier).  `store` != nullptr: record the 7 activation arrays of this (t, tile) block ([unit][lane] rows: a lane's
// four samples of an M-tile are contiguous, one 16-byte store per array and M-tile).
template <int D, int A>
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
      float *__restrict__ row = store + (size_t)j * TL + 16 * mt + 4 * g4;
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
      *reinterpret_cast<f32x4 *>(store + (size_t)ACT_U * GH * TL + (size_t)j * TL + 16 * mt + 4 * g4) = uv;
  }
  __syncthreads();
  if (wave == 0 && (lane >> 5) < A) {
    const int n = lane & 31, hf = lane >> 5;
    float z = b2_mine;
#pragma unroll 8
    for (int q = 0; q < MH; ++q) z = __builtin_fmaf(sh.uS[n][q], sh.w2S[hf][q], z);
    sh.outS[hf][n] = z;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- LSTM cell (seq/rnn/lstm.rs:17-51), same ownership
// Wave w owns units [16w, 16w+16) of the four gates [i; f; g; o]: 4 x 32 B operands of W_hh in registers; the MLP's
// first layer comes from LDS (w1S).  Per-lane state: h (registers 0..7) and the cell state c (registers 8..15), both
// zero at the start of an episode (LstmImpl::initial_cell_state, lstm.rs:22-31); only h is shared through LDS.
//   i = sigmoid(.), f = sigmoid(.), g = tanh(.), o = sigmoid(.);  c' = f * c + i * g;  h' = o * tanh(c')
// with every pre-activation = (b_hh + W_hh h) + (b_ih + W_ih x), the same fma chains as the GRU's r and z gates.
template <int D>
struct LstmFwdWeights16 {
  float whh[4][GH / 4];
  float wih[4][D];
  float bih[4], bhh[4], b1;
};

template <int D>
__device__ __forceinline__ void lstm_load_weights16(LstmFwdWeights16<D> &w, const GruParams &g, LstmFwdShared &sh,
                                                    int wave, int lane) {
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
#pragma unroll
  for (int gte = 0; gte < 4; ++gte) {
    const int row = gte * GH + j;
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks) w.whh[gte][ks] = g.Whh[(size_t)row * GH + 4 * ks + g4];
#pragma unroll
    for (int d = 0; d < D; ++d) w.wih[gte][d] = g.Wih[(size_t)row * D + d];
    w.bih[gte] = g.bih[row];
    w.bhh[gte] = g.bhh[row];
  }
  for (int ks = 0; ks < GH / 4; ++ks) sh.w1S[wave][ks][lane] = g.W1[(size_t)j * GH + 4 * ks + g4];
  w.b1 = g.b1[j];
}

template <int D, int A>
__device__ __forceinline__ void lstm_cell16(LstmFwdShared &sh, const LstmFwdWeights16<D> &w, int cur,
                                            const float (&sown)[16], float (&snew)[16], float b2_mine,
                                            float *__restrict__ store, int wave, int lane) {
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16, nxt = cur ^ 1;
  f32x4 acc[4][2];
#pragma unroll
  for (int gte = 0; gte < 4; ++gte)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[gte][mt] = (f32x4){w.bhh[gte], w.bhh[gte], w.bhh[gte], w.bhh[gte]};
#pragma unroll
  for (int ks = 0; ks < GH / 4; ++ks) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float a = sh.hT[cur][4 * ks + g4][16 * mt + n16];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte)
        acc[gte][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.whh[gte][ks], acc[gte][mt], 0, 0, 0);
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
        float v = w.bih[gte];
#pragma unroll
        for (int d = 0; d < D; ++d) v = __builtin_fmaf(sh.xS[m][d], w.wih[gte][d], v);
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
      float *__restrict__ row = store + (size_t)j * TL + 16 * mt + 4 * g4;
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
      sh.uS[m][j] = u;
      uv[i] = u;
    }
    if (store != nullptr)
      *reinterpret_cast<f32x4 *>(store + (size_t)ACT_U * GH * TL + (size_t)j * TL + 16 * mt + 4 * g4) = uv;
  }
  __syncthreads();
  if (wave == 0 && (lane >> 5) < A) {
    const int n = lane & 31, hf = lane >> 5;
    float z = b2_mine;
#pragma unroll 8
    for (int q = 0; q < MH; ++q) z = __builtin_fmaf(sh.uS[n][q], sh.w2S[hf][q], z);
    sh.outS[hf][n] = z;
  }
  __syncthreads();
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
  template <int D, int A>
  static __device__ __forceinline__ void cell(Shared &sh, const Weights<D> &w, int cur, const float (&sown)[NS],
                                              float (&snew)[NS], float b2_mine, float *__restrict__ store, int wave,
                                              int lane) {
    seq_cell16<D, A>(sh, w, cur, sown, snew, b2_mine, store, wave, lane);
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
  template <int D, int A>
  static __device__ __forceinline__ void cell(Shared &sh, const Weights<D> &w, int cur, const float (&sown)[NS],
                                              float (&snew)[NS], float b2_mine, float *__restrict__ store, int wave,
                                              int lane) {
    lstm_cell16<D, A>(sh, w, cur, sown, snew, b2_mine, store, wave, lane);
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
  __shared__ __attribute__((aligned(16))) float pk[8 * 128 + 4];  // the policy, one 8-float record per hidden unit
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
    float hout[NS];
    float *store = (act != nullptr && !peeking) ? act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL : nullptr;
    Cell::template cell<D, A>(sh, w, cur, hown, hout, b2_mine, store, wave, lane);
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
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const int m = acc16_row((r & 7) >> 2, r & 3, g4);
        const bool ended = sh.endS[m] != 0;
        const float hv = peeking ? hown[r] : hout[r];
        hown[r] = ended ? 0.0f : hv;
        if (ended && r < 8) sh.hT[buf][j][m] = 0.0f;
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

void launch_rollout_gru(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
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
  ProfScope ps(traj->eng, RL_K_POLICY_FUSED);
  RL_REQUIRE(traj->d.D == 5, "recurrent forward: built for 5 observation features");
  uint32_t tiles = traj->d.n / TL;
#define FWD(AA, CELL)                                                                                        \
  hipLaunchKernelGGL((k_gru_seq_forward<5, AA, CELL>), dim3(tiles), dim3(W16 * 64), 0, traj->eng->stream, traj->d, \
                     mod->d_params, d_out, d_succ, d_act, d_skip)
  const bool lstm = mod->kind == RL_MODULE_LSTM_MLP;
  if (mod->out_dim == 2) {
    if (lstm) FWD(2, LstmCell16);
    else FWD(2, GruCell16);
  } else {
    if (lstm) FWD(1, LstmCell16);
    else FWD(1, GruCell16);
  }
#undef FWD
}

// =====================================================================================================
// Training passes: per-sample output gradients, backward through time, weight-gradient GEMMs.
// Reference: what libtorch's autograd does for `loss.backward()` in COptimizer::backward_step
// (src/torch/optimizers/coptimizer.rs:13-26) on gru_data + linear layers over a packed batch.
// =====================================================================================================

// ---------------------------------------------------------------- d loss / d logits (policy)
// MODE_INIT: surrogate at ratio 1, loss = -mean(A): stores log pi_0, sums {ratio A, entropy, log pi(a) A}
// MODE_PPO : clipped surrogate against log pi_0 (policies/ppo.rs:124-137), sums {min(...)}
template <int MODE>
__global__ void __launch_bounds__(256) k_seq_policy_dlogits(TrajDev tr, const float *__restrict__ logits,
                                                            float *__restrict__ lp0, float *__restrict__ dz,
                                                            double *__restrict__ slabB, float inv_B, float clip_lo,
                                                            float clip_hi, const int32_t *__restrict__ skip) {
  __shared__ double red[256];
  if (skip != nullptr && *skip != 0) return;
  const size_t B = (size_t)tr.T * tr.n;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    float z[2] = {logits[b], logits[B + b]}, lp[2];
    log_softmax_lane<2>(z, lp);
    const int act = tr.action[b];
    const float adv = tr.adv[b];
    const float pa0 = rl_expf(lp[0]), pa1 = rl_expf(lp[1]);
    const float lpa = act == 0 ? lp[0] : lp[1];
    float c;
    if (MODE == PASS_EVAL) {
      // (loss, KL(pi_0 || pi)) of the current parameters (trpo.rs:124-140; categorical.rs:69-76)
      const float l00 = lp0[b], l01 = lp0[B + b];
      const float l0a = act == 0 ? l00 : l01;
      const float ratio = rl_expf(lpa - l0a);
      float rel0 = l00 - lp[0], rel1 = l01 - lp[1];
      if (rel0 < -3.402823466e+38f) rel0 = -3.402823466e+38f;
      if (rel1 < -3.402823466e+38f) rel1 = -3.402823466e+38f;
      float kl = rel0 * rl_expf(l00);
      kl += rel1 * rl_expf(l01);
      s0 += (double)(ratio * adv);
      s1 += (double)kl;
      continue;
    }
    if (MODE == PASS_INIT) {
      lp0[b] = lp[0];
      lp0[B + b] = lp[1];
      const float ratio = rl_expf(lpa - lpa);
      c = -(ratio * adv) * inv_B;
      const float cl0 = lp[0] < -3.402823466e+38f ? -3.402823466e+38f : lp[0];
      const float cl1 = lp[1] < -3.402823466e+38f ? -3.402823466e+38f : lp[1];
      float ent = cl0 * pa0;
      ent += cl1 * pa1;
      s0 += (double)(ratio * adv);
      s1 += (double)(-ent);
      s2 += (double)(lpa * adv);
    } else {
      const float l0a = lp0[(size_t)act * B + b];
      const float ratio = rl_expf(lpa - l0a);
      const float clipped = ratio < clip_lo ? clip_lo : (ratio > clip_hi ? clip_hi : ratio);
      const float u1 = ratio * adv, u2 = clipped * adv;
      const bool inside = ratio >= clip_lo && ratio <= clip_hi;
      const float gr = u1 < u2 ? adv : (u1 > u2 ? (inside ? adv : 0.0f) : (inside ? adv : 0.5f * adv));
      c = -(gr * ratio) * inv_B;
      s0 += (double)(u1 < u2 ? u1 : u2);
    }
    dz[b] = c * ((act == 0 ? 1.0f : 0.0f) - pa0);
    dz[B + b] = c * ((act == 1 ? 1.0f : 0.0f) - pa1);
  }
  const double t0 = block_sum<256>(s0, red), t1 = block_sum<256>(s1, red), t2 = block_sum<256>(s2, red);
  if (threadIdx.x == 0) {
    slabB[blockIdx.x * 4 + 0] = t0;
    slabB[blockIdx.x * 4 + 1] = t1;
    slabB[blockIdx.x * 4 + 2] = t2;
    slabB[blockIdx.x * 4 + 3] = 0.0;
  }
}

// critic: d mse_loss(V, target, Mean) / d V = 2 (V - target) / B (critics/opt.rs:109-115); sums {(V - target)^2}
__global__ void __launch_bounds__(256) k_seq_critic_dvalues(TrajDev tr, const float *__restrict__ values,
                                                            float *__restrict__ dz, double *__restrict__ slabB,
                                                            float two_over_B) {
  __shared__ double red[256];
  const size_t B = (size_t)tr.T * tr.n;
  double s0 = 0.0;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    const float d = values[b] - tr.rtg[b];
    dz[b] = d * two_over_B;
    s0 += (double)(d * d);
  }
  const double t0 = block_sum<256>(s0, red);
  if (threadIdx.x == 0) {
    slabB[blockIdx.x * 4 + 0] = t0;
    slabB[blockIdx.x * 4 + 1] = 0.0;
    slabB[blockIdx.x * 4 + 2] = 0.0;
    slabB[blockIdx.x * 4 + 3] = 0.0;
  }
}

// ---------------------------------------------------------------- backward through time
// One workgroup (eight waves) per tile, t = T-1 .. 0.  Wave w owns units k in [16w, 16w+16) of every back-propagated
// vector: its slices of W1^T (32 B-operands of 16x16x4 MFMAs) and W_hh^T (3 x 32) stay in registers; the vectors being
// multiplied pass through LDS ([128][33] per vector: d u_pre, then the three gate vectors side by side, so a step
// needs two workgroup barriers).  The sums over k run as independent MFMA chains (two half-chains per M-tile for
// W1^T, one chain per gate and M-tile for W_hh^T), added at the end: the matrix pipe stays busy instead of waiting
// for one accumulator.  A lane's four samples of an M-tile are contiguous in the record ([unit][lane] rows), so every
// record access is one 16-byte load / store; the record of step t-1 is requested as soon as step t has consumed its
// own, and lands under the gate products.  Writes the five per-step arrays the weight-gradient GEMMs read.
// (Gradients are compared with the oracle within fp32 tolerances, tests/test_gpu_gru.py: the order of these sums is
// free, unlike the forward's.)
template <int A>
__global__ void __launch_bounds__(W16 * 64, 2) k_gru_bptt(TrajDev tr, const float *__restrict__ params, int D,
                                                          const float *__restrict__ dz, const float *__restrict__ act,
                                                          float *__restrict__ dpre, const int32_t *__restrict__ skip) {
  __shared__ float bufU[GH][TLS];
  __shared__ float bufG[3][GH][TLS];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const size_t B = (size_t)T * N;
  const GruParams g = gru_params(params, D, A);
  float whhT[3][GH / 4], w1T[MH / 4], w2c[A];
#pragma unroll
  for (int gte = 0; gte < 3; ++gte)
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks) whhT[gte][ks] = g.Whh[(size_t)(gte * GH + 4 * ks + g4) * GH + j];
#pragma unroll
  for (int ks = 0; ks < MH / 4; ++ks) w1T[ks] = g.W1[(size_t)(4 * ks + g4) * GH + j];
#pragma unroll
  for (int a = 0; a < A; ++a) w2c[a] = g.W2[a * MH + j];

  // one step's inputs: the lane's 2 x 4 samples of its unit in six record arrays + u, the logit gradients and the
  // episode-end flags of those samples
  struct StepIn {
    f32x4 u[2], a1[2], r[2], z[2], n[2], ghn[2], hp[2], dzv[A][2];
    uint32_t end[2];  // four flag bytes
  };
  const size_t lo = (size_t)j * TL + 4 * g4;  // + 16 mt: first of the lane's four contiguous samples
  auto load_u = [&](StepIn &in, uint32_t t) {
    const float *__restrict__ ab = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      in.u[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_U * GH * TL + lo + 16 * mt);
#pragma unroll
      for (int a = 0; a < A; ++a)
        in.dzv[a][mt] = *reinterpret_cast<const f32x4 *>(dz + (size_t)a * B + (size_t)t * N + lane0 + 16 * mt + 4 * g4);
      in.end[mt] = *reinterpret_cast<const uint32_t *>(tr.flag + (size_t)t * N + lane0 + 16 * mt + 4 * g4);
    }
  };
  auto load_cell = [&](StepIn &in, uint32_t t) {
    const float *__restrict__ ab = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const size_t o = lo + 16 * mt;
      in.a1[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_A1 * GH * TL + o);
      in.r[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_R * GH * TL + o);
      in.z[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_Z * GH * TL + o);
      in.n[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_N * GH * TL + o);
      in.ghn[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_GHN * GH * TL + o);
      in.hp[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_HPREV * GH * TL + o);
    }
  };
  StepIn in;
  load_u(in, T - 1);
  load_cell(in, T - 1);
  f32x4 dhc[2];
  dhc[0] = dhc[1] = (f32x4){0, 0, 0, 0};
  for (uint32_t t = T; t-- > 0;) {
    float *__restrict__ db = dpre + ((size_t)t * tiles + tile) * DPRE_ARR * GH * TL;
    // head: d u_pre = [u > 0] W2^T dz
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 duv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float du = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) du = __builtin_fmaf(in.dzv[a][mt][i], w2c[a], du);
        du = in.u[mt][i] > 0.0f ? du : 0.0f;
        bufU[j][16 * mt + 4 * g4 + i] = du;
        duv[i] = du;
      }
      *reinterpret_cast<f32x4 *>(db + (size_t)4 * GH * TL + lo + 16 * mt) = duv;
    }
    uint32_t endw[2] = {in.end[0], in.end[1]};
    if (t > 0) load_u(in, t - 1);  // consumed at the top of the next step
    __syncthreads();
    // d relu(h') = W1^T d u_pre: two half-chains per M-tile
    f32x4 acc1[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc1[mt][0] = acc1[mt][1] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < MH / 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc1[mt][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bufU[4 * ks + g4][16 * mt + n16], w1T[ks],
                                                                acc1[mt][ks & 1], 0, 0, 0);
    // cell: h' = (h - n) z + n
    f32x4 dhdir[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 grv, gzv, dpnv, gnrv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * mt + 4 * g4 + i;
        const float rr = in.r[mt][i], zz = in.z[mt][i], nn = in.n[mt][i];
        const bool ended = ((endw[mt] >> (8 * i)) & 0xffu) != RL_SUCC_CONTINUE;
        float dh = ended ? 0.0f : dhc[mt][i];
        dh = dh + (in.a1[mt][i] > 0.0f ? acc1[mt][0][i] + acc1[mt][1][i] : 0.0f);
        const float dzg = dh * (in.hp[mt][i] - nn);
        const float dn = dh * (1.0f - zz);
        const float dpn = dn * (1.0f - nn * nn);
        const float dr = dpn * in.ghn[mt][i];
        grv[i] = dr * rr * (1.0f - rr);
        gzv[i] = dzg * zz * (1.0f - zz);
        gnrv[i] = dpn * rr;
        dpnv[i] = dpn;
        dhdir[mt][i] = dh * zz;
        bufG[0][j][m] = grv[i];
        bufG[1][j][m] = gzv[i];
        bufG[2][j][m] = gnrv[i];
      }
      const size_t o = lo + 16 * mt;
      *reinterpret_cast<f32x4 *>(db + (size_t)0 * GH * TL + o) = grv;
      *reinterpret_cast<f32x4 *>(db + (size_t)1 * GH * TL + o) = gzv;
      *reinterpret_cast<f32x4 *>(db + (size_t)2 * GH * TL + o) = dpnv;
      *reinterpret_cast<f32x4 *>(db + (size_t)3 * GH * TL + o) = gnrv;
    }
    if (t > 0) load_cell(in, t - 1);  // lands under the gate products below
    __syncthreads();
    // d h_prev = dh z + sum over gates of W_hh[g]^T d gh_g: one chain per gate and M-tile
    f32x4 accg[3][2];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte) accg[gte][0] = accg[gte][1] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int gte = 0; gte < 3; ++gte)
          accg[gte][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bufG[gte][4 * ks + g4][16 * mt + n16], whhT[gte][ks],
                                                               accg[gte][mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) dhc[mt][i] = dhdir[mt][i] + ((accg[0][mt][i] + accg[1][mt][i]) + accg[2][mt][i]);
  }
}

// ---------------------------------------------------------------- LSTM: backward through time
// The LSTM's four W_hh^T slices fill the registers the GRU kernel shares between W_hh^T and W1^T, so the head's backward
// (which has no recurrence) runs first as its own kernel over all (t, tile) blocks in parallel:
//   d u_pre = [u > 0] W2^T dz          -> dpre[DPRE_DU]
//   d relu(h') -> [relu(h') > 0] W1^T d u_pre   -> dpre[DPRE_DA1]
// and k_lstm_bptt walks t = T-1 .. 0 with (dh, dc) carried in registers:
//   dh = dh_next + d relu(h');  d o = dh tanh(c');  d c' = dc_next + dh o (1 - tanh(c')^2)
//   d f = d c' c;  d i = d c' g;  d g = d c' i;  dc = d c' f;  pre-activation gradients with the gates' derivatives;
//   dh_prev = sum over the four gates of W_hh[g]^T d pre_g     (one MFMA chain per gate and M-tile)
template <int A>
__global__ void __launch_bounds__(W16 * 64, 2) k_seq_head_backward(TrajDev tr, const float *__restrict__ params, int D,
                                                                    int NG, const float *__restrict__ dz,
                                                                    const float *__restrict__ act,
                                                                    float *__restrict__ dpre, uint32_t tiles,
                                                                    uint32_t blocks, const int32_t *__restrict__ skip) {
  __shared__ float bufU[GH][TLS];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const size_t B = (size_t)T * N;
  const GruParams g = seq_params(params, D, A, NG);
  float w1T[MH / 4], w2c[A];
#pragma unroll
  for (int ks = 0; ks < MH / 4; ++ks) w1T[ks] = g.W1[(size_t)(4 * ks + g4) * GH + j];
#pragma unroll
  for (int a = 0; a < A; ++a) w2c[a] = g.W2[a * MH + j];
  const size_t lo = (size_t)j * TL + 4 * g4;
  for (uint32_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    const uint32_t t = blk / tiles, lane0 = (blk % tiles) * TL;
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    float *__restrict__ db = dpre + (size_t)blk * DPRE_ARR * GH * TL;
    __syncthreads();  // the previous block's readers of bufU are done
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 uv = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_U * GH * TL + lo + 16 * mt);
      f32x4 dzv[A], duv;
#pragma unroll
      for (int a = 0; a < A; ++a)
        dzv[a] = *reinterpret_cast<const f32x4 *>(dz + (size_t)a * B + (size_t)t * N + lane0 + 16 * mt + 4 * g4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float du = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) du = __builtin_fmaf(dzv[a][i], w2c[a], du);
        du = uv[i] > 0.0f ? du : 0.0f;
        bufU[j][16 * mt + 4 * g4 + i] = du;
        duv[i] = du;
      }
      *reinterpret_cast<f32x4 *>(db + (size_t)DPRE_DU * GH * TL + lo + 16 * mt) = duv;
    }
    __syncthreads();
    f32x4 acc1[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc1[mt][0] = acc1[mt][1] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < MH / 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc1[mt][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bufU[4 * ks + g4][16 * mt + n16], w1T[ks],
                                                                acc1[mt][ks & 1], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_A1 * GH * TL + lo + 16 * mt);
      f32x4 dav;
#pragma unroll
      for (int i = 0; i < 4; ++i) dav[i] = a1[i] > 0.0f ? acc1[mt][0][i] + acc1[mt][1][i] : 0.0f;
      *reinterpret_cast<f32x4 *>(db + (size_t)DPRE_DA1 * GH * TL + lo + 16 * mt) = dav;
    }
  }
}

__global__ void __launch_bounds__(W16 * 64, 2) k_lstm_bptt(TrajDev tr, const float *__restrict__ params, int D, int A,
                                                           const float *__restrict__ act, float *__restrict__ dpre,
                                                           const int32_t *__restrict__ skip) {
  __shared__ float bufG[4][GH][TLS];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = seq_params(params, D, A, 4);
  float whhT[4][GH / 4];
#pragma unroll
  for (int gte = 0; gte < 4; ++gte)
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks) whhT[gte][ks] = g.Whh[(size_t)(gte * GH + 4 * ks + g4) * GH + j];
  const size_t lo = (size_t)j * TL + 4 * g4;
  f32x4 dhc[2], dcc[2];
  dhc[0] = dhc[1] = dcc[0] = dcc[1] = (f32x4){0, 0, 0, 0};
  for (uint32_t t = T; t-- > 0;) {
    const float *__restrict__ ab = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
    float *__restrict__ db = dpre + ((size_t)t * tiles + tile) * DPRE_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const size_t o = lo + 16 * mt;
      const f32x4 iv = *reinterpret_cast<const f32x4 *>(ab + (size_t)LACT_I * GH * TL + o);
      const f32x4 fv = *reinterpret_cast<const f32x4 *>(ab + (size_t)LACT_F * GH * TL + o);
      const f32x4 gv = *reinterpret_cast<const f32x4 *>(ab + (size_t)LACT_G * GH * TL + o);
      const f32x4 ov = *reinterpret_cast<const f32x4 *>(ab + (size_t)LACT_O * GH * TL + o);
      const f32x4 cp = *reinterpret_cast<const f32x4 *>(ab + (size_t)LACT_CPREV * GH * TL + o);
      const f32x4 tc = *reinterpret_cast<const f32x4 *>(ab + (size_t)LACT_TC * GH * TL + o);
      const f32x4 da1 = *reinterpret_cast<const f32x4 *>(db + (size_t)DPRE_DA1 * GH * TL + o);
      const uint32_t endw = *reinterpret_cast<const uint32_t *>(tr.flag + (size_t)t * N + lane0 + 16 * mt + 4 * g4);
      f32x4 div, dfv, dgv, dov;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * mt + 4 * g4 + i;
        const bool ended = ((endw >> (8 * i)) & 0xffu) != RL_SUCC_CONTINUE;
        const float dh = (ended ? 0.0f : dhc[mt][i]) + da1[i];
        const float dcn_in = ended ? 0.0f : dcc[mt][i];
        const float dO = dh * tc[i];
        const float dtc = dh * ov[i];
        const float dcn = dcn_in + dtc * (1.0f - tc[i] * tc[i]);
        const float dF = dcn * cp[i], dI = dcn * gv[i], dG = dcn * iv[i];
        dcc[mt][i] = dcn * fv[i];
        div[i] = dI * iv[i] * (1.0f - iv[i]);
        dfv[i] = dF * fv[i] * (1.0f - fv[i]);
        dgv[i] = dG * (1.0f - gv[i] * gv[i]);
        dov[i] = dO * ov[i] * (1.0f - ov[i]);
        bufG[0][j][m] = div[i];
        bufG[1][j][m] = dfv[i];
        bufG[2][j][m] = dgv[i];
        bufG[3][j][m] = dov[i];
      }
      *reinterpret_cast<f32x4 *>(db + (size_t)0 * GH * TL + o) = div;
      *reinterpret_cast<f32x4 *>(db + (size_t)1 * GH * TL + o) = dfv;
      *reinterpret_cast<f32x4 *>(db + (size_t)2 * GH * TL + o) = dgv;
      *reinterpret_cast<f32x4 *>(db + (size_t)3 * GH * TL + o) = dov;
    }
    __syncthreads();
    f32x4 accg[4][2];
#pragma unroll
    for (int gte = 0; gte < 4; ++gte) accg[gte][0] = accg[gte][1] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int gte = 0; gte < 4; ++gte)
          accg[gte][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bufG[gte][4 * ks + g4][16 * mt + n16], whhT[gte][ks],
                                                               accg[gte][mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dhc[mt][i] = ((accg[0][mt][i] + accg[1][mt][i]) + accg[2][mt][i]) + accg[3][mt][i];
    __syncthreads();  // bufG is rewritten by the next step
  }
}

// ---------------------------------------------------------------- weight-gradient GEMMs
// dW_hh = sum dgh (x) h_prev [384 x 128], dW1 = sum du (x) relu(h') [128 x 128] on the matrix cores with the
// sample index as the MFMA k dimension (k-pair (ks, hf) <-> sample m = 16 hf + ks, so every operand is 16
// consecutive floats of a [unit][32] row); dW_ih, the biases and dW2 on the VALU.  A workgroup accumulates a
// contiguous run of (t, tile) blocks in f32 and writes one row of partials; k_seq_reduce sums the rows in f64.
// The four A-operand arrays of a block (d gh_r, d gh_z, d gh_n, d u_pre: every wave needs all 512 rows) are fetched
// ONCE per workgroup with fully coalesced 16-byte loads into registers while the previous block's products run, then
// parked in LDS ([unit][36] rows: the 16-byte operand reads of 16 consecutive rows hit 64 different banks); a wave's B
// rows (h_prev, relu(h') of its own 32 units) and the rows only its VALU sums need come straight from HBM.
// NGT gate blocks in the module: 3 = GRU (staged: d gh_r, d gh_z, d gh_n = d pre_n * r), 4 = LSTM.  One launch covers the
// gates [G0, G0 + GN) and, WITH_W1, the head (W1, b1, W2, b2): the GRU takes everything in one launch (16 output tiles per
// wave = 256 accumulator registers), the LSTM's 20 tiles are split into gates {i, f} + head and gates {g, o}.
template <int D, int A, int NGT, int G0, int GN, bool WITH_W1>
__global__ void __launch_bounds__(256, 1) k_gru_wgrad(TrajDev tr, const float *__restrict__ dz,
                                                      const float *__restrict__ act, const float *__restrict__ dpre,
                                                      float *__restrict__ slab, uint32_t P, uint32_t tiles,
                                                      uint32_t blocks, uint32_t blocks_per_chunk,
                                                      const int32_t *__restrict__ skip) {
  constexpr int RS = TL + 4;
  constexpr int S_DU = GN, S_N = GN + (WITH_W1 ? 1 : 0);  // staged arrays: the hidden-side gate gradients [, d u_pre]
  __shared__ __attribute__((aligned(16))) float aS[S_N][GH][RS];
  __shared__ float xS[TL][8];
  __shared__ float dzS[2][TL];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const size_t B = (size_t)T * N, plane = (size_t)(T + 1) * N;
  f32x16 acc_hh[4 * GN], acc_w1[4];
#pragma unroll
  for (int q = 0; q < 4 * GN; ++q) acc_hh[q] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) acc_w1[q] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float dwih[GN][D], dbih[GN], dbhh[GN], db1 = 0.0f, dw2[A], db2 = 0.0f;
#pragma unroll
  for (int gte = 0; gte < GN; ++gte) {
    dbih[gte] = dbhh[gte] = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) dwih[gte][d] = 0.0f;
  }
#pragma unroll
  for (int a = 0; a < A; ++a) dw2[a] = 0.0f;
  const uint32_t b0 = blockIdx.x * blocks_per_chunk;
  const uint32_t b1 = b0 + blocks_per_chunk < blocks ? b0 + blocks_per_chunk : blocks;

  // staging registers: thread q holds the 16-byte pieces q, q + 256, q + 512, q + 768 of each [128][32] array
  f32x4 stg[S_N][4];
  float xn[D], dzn[A];  // wave 0, lanes < 32: the block's observation features and logit gradients
  auto stage_load = [&](uint32_t blk) {
    const float *__restrict__ db = dpre + (size_t)blk * DPRE_ARR * GH * TL;
    // GRU: dpre arrays 0, 1, 3 (hidden side of the n gate), 4; LSTM: 0, 1, 2, 3, 4
    const float *src[S_N];
#pragma unroll
    for (int a = 0; a < GN; ++a) src[a] = db + (size_t)((NGT == 3 && G0 + a == 2) ? 3 : G0 + a) * GH * TL;
    if constexpr (WITH_W1) src[S_DU] = db + (size_t)DPRE_DU * GH * TL;
#pragma unroll
    for (int a = 0; a < S_N; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) stg[a][i] = *reinterpret_cast<const f32x4 *>(src[a] + 4 * (threadIdx.x + 256 * i));
    if (wave == 0 && lane < TL) {
      const uint32_t t = blk / tiles, lane0 = (blk % tiles) * TL;
#pragma unroll
      for (int d = 0; d < D; ++d) xn[d] = tr.obs[d * plane + (size_t)t * N + lane0 + lane];
#pragma unroll
      for (int a = 0; a < A; ++a) dzn[a] = dz[(size_t)a * B + (size_t)t * N + lane0 + lane];
    }
  };
  if (b0 < b1) stage_load(b0);
  for (uint32_t blk = b0; blk < b1; ++blk) {
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    const float *__restrict__ db = dpre + (size_t)blk * DPRE_ARR * GH * TL;
    __syncthreads();  // the previous block's readers of the LDS operands are done
#pragma unroll
    for (int a = 0; a < S_N; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i;
        *reinterpret_cast<f32x4 *>(&aS[a][f >> 3][4 * (f & 7)]) = stg[a][i];
      }
    if (wave == 0 && lane < TL) {
#pragma unroll
      for (int d = 0; d < D; ++d) xS[lane][d] = xn[d];
#pragma unroll
      for (int a = 0; a < A; ++a) dzS[a][lane] = dzn[a];
    }
    // straight from HBM: the B operands of this wave's column tile (h_prev and relu(h') rows of unit k = j, samples
    // 16 hf .. 16 hf + 15) and the owner rows' operands that only the VALU sums use (d pre_n, u)
    f32x4 hp4[4], a14[4], dpnv[4], uv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t o = (size_t)j * TL + 16 * hf + 4 * q;
      hp4[q] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_HPREV * GH * TL + o);
      a14[q] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_A1 * GH * TL + o);
      dpnv[q] = *reinterpret_cast<const f32x4 *>(db + (size_t)2 * GH * TL + o);
      uv[q] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_U * GH * TL + o);
    }
    __syncthreads();
    if (blk + 1 < b1) stage_load(blk + 1);  // lands under this block's products
    float hpB[16], a1B[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hpB[4 * q + i] = hp4[q][i];
        a1B[4 * q + i] = a14[q][i];
      }
#pragma unroll
    for (int mt = 0; mt < 4 * GN; ++mt) {
      const int gte = mt >> 2, row = 32 * (mt & 3) + n;  // local gate index; unit of this lane's A row
      float av[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(&aS[gte][row][16 * hf + 4 * q]);
#pragma unroll
        for (int i = 0; i < 4; ++i) av[4 * q + i] = v4[i];
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        acc_hh[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], hpB[ks], acc_hh[mt], 0, 0, 0);
      if ((mt & 3) == wave) {
        // this wave owns rows 32 * wave + n of gate `gte` for the VALU-side sums
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int m = 16 * hf + ks;
          const float dgh = av[ks];
          const float dgi = (NGT == 3 && G0 + gte == 2) ? dpnv[ks >> 2][ks & 3] : dgh;  // GRU: input side of the n gate
          dbhh[gte] += dgh;
          dbih[gte] += dgi;
#pragma unroll
          for (int d = 0; d < D; ++d) dwih[gte][d] = __builtin_fmaf(dgi, xS[m][d], dwih[gte][d]);
        }
      }
    }
    if constexpr (WITH_W1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = 32 * mt + n;
      float av[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(&aS[S_DU][row][16 * hf + 4 * q]);
#pragma unroll
        for (int i = 0; i < 4; ++i) av[4 * q + i] = v4[i];
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        acc_w1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], a1B[ks], acc_w1[mt], 0, 0, 0);
      if (mt == wave) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int m = 16 * hf + ks;
          db1 += av[ks];
          const float u = uv[ks >> 2][ks & 3];
#pragma unroll
          for (int a = 0; a < A; ++a) dw2[a] = __builtin_fmaf(dzS[a][m], u, dw2[a]);
        }
      }
    }
    if (wave == 0 && lane < A * TL) db2 += dzS[lane >> 5][lane & 31];  // lane = (a = hf, m = n)
    }
  }
  // ---- write this workgroup's row of partials
  float *__restrict__ out = slab + (size_t)blockIdx.x * P;
  const size_t oWih = 0, oWhh = oWih + (size_t)NGT * GH * D, obih = oWhh + (size_t)NGT * GH * GH, obhh = obih + NGT * GH;
  const size_t oW1 = obhh + NGT * GH, ob1 = oW1 + (size_t)MH * GH, oW2 = ob1 + MH, ob2 = oW2 + (size_t)A * MH;
#pragma unroll
  for (int mt = 0; mt < 4 * GN; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[oWhh + (size_t)(G0 * GH + 32 * mt + acc_row(r, hf)) * GH + j] = acc_hh[mt][r];
  if constexpr (WITH_W1) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[oW1 + (size_t)(32 * mt + acc_row(r, hf)) * GH + j] = acc_w1[mt][r];
  }
  // VALU sums: the two halves of a wave hold samples 0..15 and 16..31 of the same rows
#pragma unroll
  for (int gte = 0; gte < GN; ++gte) {
    const int row = (G0 + gte) * GH + j;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float v = dwih[gte][d] + __shfl_xor(dwih[gte][d], 32, 64);
      if (hf == 0) out[oWih + (size_t)row * D + d] = v;
    }
    const float vi = dbih[gte] + __shfl_xor(dbih[gte], 32, 64), vh = dbhh[gte] + __shfl_xor(dbhh[gte], 32, 64);
    if (hf == 0) {
      out[obih + row] = vi;
      out[obhh + row] = vh;
    }
  }
  if constexpr (WITH_W1) {
    const float v = db1 + __shfl_xor(db1, 32, 64);
    if (hf == 0) out[ob1 + j] = v;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      const float v2 = dw2[a] + __shfl_xor(dw2[a], 32, 64);
      if (hf == 0) out[oW2 + (size_t)a * MH + j] = v2;
    }
  }
  if (WITH_W1 && wave == 0) {
    float v = db2;  // lanes of half `a` hold the per-sample-slot sums of output a
#pragma unroll
    for (int s = 16; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    if (n == 0 && hf < A) out[ob2 + hf] = v;
  }
}

// rows of f32 partials -> one f32 vector, accumulated in f64 in a fixed order: 64 columns per workgroup, the rows dealt
// round-robin to 16 thread groups (eight loads in flight each), the 16 partial sums added in group order
__global__ void __launch_bounds__(1024) k_seq_reduce(const float *__restrict__ slab, uint32_t rows, uint32_t P,
                                                     float *__restrict__ vec) {
  __shared__ double part[16][64];
  const uint32_t c = threadIdx.x & 63, grp = threadIdx.x >> 6, p = blockIdx.x * 64 + c;
  double s = 0.0;
  if (p < P) {
    uint32_t r = grp;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = slab[(size_t)(r + 16 * q) * P + p];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += (double)v[q];
    }
    for (; r < rows; r += 16) s += (double)slab[(size_t)r * P + p];
  }
  part[grp][c] = s;
  __syncthreads();
  if (grp == 0 && p < P) {
    double t = part[0][c];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += part[q][c];
    vec[p] = (float)t;
  }
}

void launch_seq_policy_dlogits(rl_traj *traj, int mode, uint64_t B_total, float clip_lo, float clip_hi,
                               const int32_t *d_skip) {
  ProfScope ps(traj->eng, RL_K_POLICY_PASS);
  float inv_B = 1.0f / (float)B_total;
  dim3 g(traj->nbB), b(256);
#define DL(MM)                                                                                                     \
  hipLaunchKernelGGL(k_seq_policy_dlogits<MM>, g, b, 0, traj->eng->stream, traj->d, traj->seq.out, traj->lp0, traj->dz, \
                     traj->slabB, inv_B, clip_lo, clip_hi, d_skip)
  if (mode == PASS_INIT) DL(PASS_INIT);
  else if (mode == PASS_EVAL) DL(PASS_EVAL);
  else DL(PASS_PPO);
#undef DL
}

void launch_seq_critic_dvalues(rl_traj *traj, uint64_t B_total) {
  ProfScope ps(traj->eng, RL_K_CRITIC_FWD);
  hipLaunchKernelGGL(k_seq_critic_dvalues, dim3(traj->nbB), dim3(256), 0, traj->eng->stream, traj->d, traj->seq.out,
                     traj->dz, traj->slabB, 2.0f / (float)B_total);
}

// backward through time + weight-gradient GEMMs + reduction: traj->vec[0..P) <- sum over this rank's samples
void launch_gru_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip) {
  rl_engine *e = traj->eng;
  const SeqDev &q = traj->seq;
  uint32_t P = (uint32_t)mod->P, blocks = traj->d.T * q.tiles;
  const bool lstm = mod->kind == RL_MODULE_LSTM_MLP;
  {
    ProfScope ps(e, RL_K_BACKWARD);
    if (lstm) {
      const uint32_t grid = blocks < 2048 ? blocks : 2048;
      if (mod->out_dim == 2)
        hipLaunchKernelGGL(k_seq_head_backward<2>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, 4,
                           traj->dz, q.act, q.dpre, q.tiles, blocks, d_skip);
      else
        hipLaunchKernelGGL(k_seq_head_backward<1>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, 4,
                           traj->dz, q.act, q.dpre, q.tiles, blocks, d_skip);
      hipLaunchKernelGGL(k_lstm_bptt, dim3(q.tiles), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5,
                         (int)mod->out_dim, q.act, q.dpre, d_skip);
    } else if (mod->out_dim == 2) {
      hipLaunchKernelGGL(k_gru_bptt<2>, dim3(q.tiles), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, traj->dz,
                         q.act, q.dpre, d_skip);
    } else {
      hipLaunchKernelGGL(k_gru_bptt<1>, dim3(q.tiles), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, traj->dz,
                         q.act, q.dpre, d_skip);
    }
  }
  {
    ProfScope ps(e, RL_K_CRITIC_FUSED);
#define WG(AA, NGT, G0, GN, W1)                                                                                   \
  hipLaunchKernelGGL((k_gru_wgrad<5, AA, NGT, G0, GN, W1>), dim3(q.chunks), dim3(256), 0, e->stream, traj->d, traj->dz, \
                     q.act, q.dpre, q.wg_slab, P, q.tiles, blocks, q.blocks_per_chunk, d_skip)
    if (mod->out_dim == 2) {
      if (lstm) {
        WG(2, 4, 0, 2, true);
        WG(2, 4, 2, 2, false);
      } else {
        WG(2, 3, 0, 3, true);
      }
    } else {
      if (lstm) {
        WG(1, 4, 0, 2, true);
        WG(1, 4, 2, 2, false);
      } else {
        WG(1, 3, 0, 3, true);
      }
    }
#undef WG
  }
  {
    ProfScope ps(e, RL_K_REDUCE);
    hipLaunchKernelGGL(k_seq_reduce, dim3(cdiv_s(P, 64)), dim3(1024), 0, e->stream, q.wg_slab, q.chunks, P, traj->vec);
  }
}

// =====================================================================================================
// Fisher-vector products through time (TRPO over the recurrent policy): J v by forward-mode differentiation.
// The tangent recurrence of a step needs W_hh h_dot (recurrent) and V_hh h (V = tangent parameters; h is known
// from the activation record, so this part is NOT recurrent).  It is split accordingly:
//   k_gru_tangent_pre : all (t, tile) blocks in parallel, tangent weights in registers:
//                       static terms V_hh h + v_bhh (+ V_ih x + v_bih), V1 relu(h') + v_b1, V2 u + v_b2
//   k_gru_tangent_rec : per tile, t ascending, the model's own W_hh / W1 slices in registers (as the forward):
//                       h_dot recurrence, u_dot, out_dot
// Reference: HessianVectorProduct::mat_vec_mul (src/torch/optimizers/conjugate_gradient.rs:312-338) — the double
// backward of the mean KL, which at theta_0 equals J^T (diag(p) - p p^T) J v / B.
// =====================================================================================================
template <int D, int A>
__global__ void __launch_bounds__(256, 1) k_gru_tangent_pre(TrajDev tr, const float *__restrict__ tangent,
                                                            const float *__restrict__ act, float *__restrict__ stat,
                                                            float *__restrict__ out_stat, uint32_t tiles,
                                                            uint32_t blocks, uint32_t blocks_per_chunk,
                                                            const int32_t *__restrict__ skip) {
  __shared__ float xS[TL][8];
  __shared__ float v2S[2][MH];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const size_t plane = (size_t)(T + 1) * N;
  const GruParams v = gru_params(tangent, D, A);
  SeqFwdWeights<D> w;
  seq_load_weights<D>(w, v, wave, lane);  // the same slicing as the forward, applied to the tangent parameters
  for (int q = threadIdx.x; q < A * MH; q += 256) v2S[q / MH][q % MH] = v.W2[q];
  const float vb2 = hf < A ? v.b2[hf] : 0.0f;
  const uint32_t b0 = blockIdx.x * blocks_per_chunk;
  const uint32_t b1 = b0 + blocks_per_chunk < blocks ? b0 + blocks_per_chunk : blocks;
  for (uint32_t blk = b0; blk < b1; ++blk) {
    const uint32_t t = blk / tiles, tile = blk % tiles, lane0 = tile * TL;
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    float *__restrict__ sb = stat + (size_t)blk * DPRE_ARR * GH * TL;
    __syncthreads();
    if (wave == 0 && lane < TL)
#pragma unroll
      for (int d = 0; d < D; ++d) xS[lane][d] = tr.obs[d * plane + (size_t)t * N + lane0 + lane];
    __syncthreads();
    f32x16 acc[3];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[gte][r] = w.bhh[gte];
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) {
      const float a = ab[(size_t)ACT_HPREV * GH * TL + (2 * ks + hf) * TL + n];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte)
        acc[gte] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.whh[gte][ks], acc[gte], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      float gi[3];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) {
        float q = w.bih[gte];
#pragma unroll
        for (int d = 0; d < D; ++d) q = __builtin_fmaf(xS[m][d], w.wih[gte][d], q);
        gi[gte] = q;
      }
      const size_t o = (size_t)j * TL + m;
      sb[(size_t)0 * GH * TL + o] = acc[0][r] + gi[0];  // static part of d(gh_r + gi_r)
      sb[(size_t)1 * GH * TL + o] = acc[1][r] + gi[1];
      sb[(size_t)2 * GH * TL + o] = gi[2];              // d gi_n
      sb[(size_t)3 * GH * TL + o] = acc[2][r];          // static part of d gh_n
    }
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = w.b1;
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[(size_t)ACT_A1 * GH * TL + (2 * ks + hf) * TL + n], w.w1[ks],
                                                  acc1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) sb[(size_t)4 * GH * TL + (size_t)j * TL + acc_row(r, hf)] = acc1[r];
    if (wave == 0 && hf < A) {
      float z = vb2;
#pragma unroll 8
      for (int q = 0; q < MH; ++q) z = __builtin_fmaf(ab[(size_t)ACT_U * GH * TL + q * TL + n], v2S[hf][q], z);
      out_stat[((size_t)hf * T + t) * N + lane0 + n] = z;
    }
  }
}

template <int A>
__global__ void __launch_bounds__(256, 1) k_gru_tangent_rec(TrajDev tr, const float *__restrict__ params, int D,
                                                            const float *__restrict__ act,
                                                            const float *__restrict__ stat,
                                                            const float *__restrict__ out_stat,
                                                            float *__restrict__ out_dot,
                                                            const int32_t *__restrict__ skip) {
  __shared__ float hdT[GH][TL + 1];
  __shared__ float a1dT[GH][TL + 1];
  __shared__ float udS[TL][MH + 1];
  __shared__ float w2S[2][MH];
  __shared__ int endS[TL];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = gru_params(params, D, A);
  float whh[3][GH / 2], w1[GH / 2];
#pragma unroll
  for (int gte = 0; gte < 3; ++gte)
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) whh[gte][ks] = g.Whh[(size_t)(gte * GH + j) * GH + 2 * ks + hf];
#pragma unroll
  for (int ks = 0; ks < GH / 2; ++ks) w1[ks] = g.W1[(size_t)j * GH + 2 * ks + hf];
  for (int q = threadIdx.x; q < A * MH; q += 256) w2S[q / MH][q % MH] = g.W2[q];
  for (int q = threadIdx.x; q < GH * (TL + 1); q += 256) (&hdT[0][0])[q] = 0.0f;
  float hd[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hd[r] = 0.0f;
  __syncthreads();
  for (uint32_t t = 0; t < T; ++t) {
    const size_t blk = (size_t)t * tiles + tile;
    const float *__restrict__ ab = act + blk * SEQ_ARR * GH * TL;
    const float *__restrict__ sb = stat + blk * DPRE_ARR * GH * TL;
    if (wave == 0 && lane < TL) endS[lane] = tr.flag[(size_t)t * N + lane0 + lane] != RL_SUCC_CONTINUE;
    f32x16 acc[3];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte) acc[gte] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) {
      const float a = hdT[2 * ks + hf][n];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) acc[gte] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, whh[gte][ks], acc[gte], 0, 0, 0);
    }
    __syncthreads();  // every wave has read the old h_dot (and endS is visible)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const size_t o = (size_t)j * TL + m;
      const float rr = ab[(size_t)ACT_R * GH * TL + o], zz = ab[(size_t)ACT_Z * GH * TL + o];
      const float nn = ab[(size_t)ACT_N * GH * TL + o], ghn = ab[(size_t)ACT_GHN * GH * TL + o];
      const float hp = ab[(size_t)ACT_HPREV * GH * TL + o], a1 = ab[(size_t)ACT_A1 * GH * TL + o];
      const float rd = rr * (1.0f - rr) * (acc[0][r] + sb[(size_t)0 * GH * TL + o]);
      const float zd = zz * (1.0f - zz) * (acc[1][r] + sb[(size_t)1 * GH * TL + o]);
      const float ghd = acc[2][r] + sb[(size_t)3 * GH * TL + o];
      const float nd = (1.0f - nn * nn) * (sb[(size_t)2 * GH * TL + o] + rd * ghn + rr * ghd);
      const float v = (hd[r] - nd) * zz + (hp - nn) * zd + nd;
      const float keep = endS[m] != 0 ? 0.0f : v;  // the next step of an ended episode starts from zero
      hd[r] = keep;
      hdT[j][m] = keep;
      a1dT[j][m] = a1 > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    f32x16 acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1dT[2 * ks + hf][n], w1[ks], acc1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const size_t o = (size_t)j * TL + m;
      const float u = ab[(size_t)ACT_U * GH * TL + o];
      udS[m][j] = u > 0.0f ? acc1[r] + sb[(size_t)4 * GH * TL + o] : 0.0f;
    }
    __syncthreads();
    if (wave == 0 && hf < A) {
      float z = out_stat[((size_t)hf * T + t) * N + lane0 + n];
#pragma unroll 8
      for (int q = 0; q < MH; ++q) z = __builtin_fmaf(udS[n][q], w2S[hf][q], z);
      out_dot[((size_t)hf * T + t) * N + lane0 + n] = z;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- the same two kernels for the LSTM chain
// p = pre-activation of a gate: p_dot = [V_hh h + v_bhh + V_ih x + v_bih] (static, k_lstm_tangent_pre -> stat[0..3])
//                                        + W_hh h_dot (recurrent, k_lstm_tangent_rec);
// i_dot = i (1 - i) p_dot_i, f_dot, o_dot alike, g_dot = (1 - g^2) p_dot_g;
// c'_dot = f_dot c + f c_dot + i_dot g + i g_dot;  h'_dot = o_dot tanh(c') + o (1 - tanh(c')^2) c'_dot.
template <int D, int A>
__global__ void __launch_bounds__(256, 1) k_lstm_tangent_pre(TrajDev tr, const float *__restrict__ tangent,
                                                             const float *__restrict__ act, float *__restrict__ stat,
                                                             float *__restrict__ out_stat, uint32_t tiles,
                                                             uint32_t blocks, uint32_t blocks_per_chunk,
                                                             const int32_t *__restrict__ skip) {
  __shared__ float xS[TL][8];
  __shared__ float v2S[2][MH];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const size_t plane = (size_t)(T + 1) * N;
  const GruParams v = seq_params(tangent, D, A, 4);
  float whh[4][GH / 2], w1[GH / 2], wih[4][D], bih[4], bhh[4];
#pragma unroll
  for (int gte = 0; gte < 4; ++gte) {
    const int row = gte * GH + j;
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) whh[gte][ks] = v.Whh[(size_t)row * GH + 2 * ks + hf];
#pragma unroll
    for (int d = 0; d < D; ++d) wih[gte][d] = v.Wih[(size_t)row * D + d];
    bih[gte] = v.bih[row];
    bhh[gte] = v.bhh[row];
  }
#pragma unroll
  for (int ks = 0; ks < GH / 2; ++ks) w1[ks] = v.W1[(size_t)j * GH + 2 * ks + hf];
  const float vb1 = v.b1[j];
  for (int q = threadIdx.x; q < A * MH; q += 256) v2S[q / MH][q % MH] = v.W2[q];
  const float vb2 = hf < A ? v.b2[hf] : 0.0f;
  const uint32_t b0 = blockIdx.x * blocks_per_chunk;
  const uint32_t b1 = b0 + blocks_per_chunk < blocks ? b0 + blocks_per_chunk : blocks;
  for (uint32_t blk = b0; blk < b1; ++blk) {
    const uint32_t t = blk / tiles, tile = blk % tiles, lane0 = tile * TL;
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    float *__restrict__ sb = stat + (size_t)blk * DPRE_ARR * GH * TL;
    __syncthreads();
    if (wave == 0 && lane < TL)
#pragma unroll
      for (int d = 0; d < D; ++d) xS[lane][d] = tr.obs[d * plane + (size_t)t * N + lane0 + lane];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int gte = 0; gte < 4; ++gte)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[gte][r] = bhh[gte];
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) {
      const float a = ab[(size_t)ACT_HPREV * GH * TL + (2 * ks + hf) * TL + n];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte)
        acc[gte] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, whh[gte][ks], acc[gte], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const size_t o = (size_t)j * TL + m;
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) {
        float q = bih[gte];
#pragma unroll
        for (int d = 0; d < D; ++d) q = __builtin_fmaf(xS[m][d], wih[gte][d], q);
        sb[(size_t)gte * GH * TL + o] = acc[gte][r] + q;
      }
    }
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = vb1;
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[(size_t)ACT_A1 * GH * TL + (2 * ks + hf) * TL + n], w1[ks], acc1,
                                                  0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) sb[(size_t)4 * GH * TL + (size_t)j * TL + acc_row(r, hf)] = acc1[r];
    if (wave == 0 && hf < A) {
      float z = vb2;
#pragma unroll 8
      for (int q = 0; q < MH; ++q) z = __builtin_fmaf(ab[(size_t)ACT_U * GH * TL + q * TL + n], v2S[hf][q], z);
      out_stat[((size_t)hf * T + t) * N + lane0 + n] = z;
    }
  }
}

template <int A>
__global__ void __launch_bounds__(256, 1) k_lstm_tangent_rec(TrajDev tr, const float *__restrict__ params, int D,
                                                             const float *__restrict__ act,
                                                             const float *__restrict__ stat,
                                                             const float *__restrict__ out_stat,
                                                             float *__restrict__ out_dot,
                                                             const int32_t *__restrict__ skip) {
  __shared__ float hdT[GH][TL + 1];
  __shared__ float a1dT[GH][TL + 1];
  __shared__ float udS[TL][MH + 1];
  __shared__ float w2S[2][MH];
  __shared__ int endS[TL];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = seq_params(params, D, A, 4);
  float whh[4][GH / 2], w1[GH / 2];
#pragma unroll
  for (int gte = 0; gte < 4; ++gte)
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) whh[gte][ks] = g.Whh[(size_t)(gte * GH + j) * GH + 2 * ks + hf];
#pragma unroll
  for (int ks = 0; ks < GH / 2; ++ks) w1[ks] = g.W1[(size_t)j * GH + 2 * ks + hf];
  for (int q = threadIdx.x; q < A * MH; q += 256) w2S[q / MH][q % MH] = g.W2[q];
  for (int q = threadIdx.x; q < GH * (TL + 1); q += 256) (&hdT[0][0])[q] = 0.0f;
  float hd[16], cd[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) hd[r] = cd[r] = 0.0f;
  __syncthreads();
  for (uint32_t t = 0; t < T; ++t) {
    const size_t blk = (size_t)t * tiles + tile;
    const float *__restrict__ ab = act + blk * SEQ_ARR * GH * TL;
    const float *__restrict__ sb = stat + blk * DPRE_ARR * GH * TL;
    if (wave == 0 && lane < TL) endS[lane] = tr.flag[(size_t)t * N + lane0 + lane] != RL_SUCC_CONTINUE;
    f32x16 acc[4];
#pragma unroll
    for (int gte = 0; gte < 4; ++gte) acc[gte] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks) {
      const float a = hdT[2 * ks + hf][n];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte) acc[gte] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, whh[gte][ks], acc[gte], 0, 0, 0);
    }
    __syncthreads();  // every wave has read the old h_dot (and endS is visible)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const size_t o = (size_t)j * TL + m;
      const float ig = ab[(size_t)LACT_I * GH * TL + o], fg = ab[(size_t)LACT_F * GH * TL + o];
      const float gg = ab[(size_t)LACT_G * GH * TL + o], og = ab[(size_t)LACT_O * GH * TL + o];
      const float cp = ab[(size_t)LACT_CPREV * GH * TL + o], tc = ab[(size_t)LACT_TC * GH * TL + o];
      const float a1 = ab[(size_t)ACT_A1 * GH * TL + o];
      const float id = ig * (1.0f - ig) * (acc[0][r] + sb[(size_t)0 * GH * TL + o]);
      const float fd = fg * (1.0f - fg) * (acc[1][r] + sb[(size_t)1 * GH * TL + o]);
      const float gd = (1.0f - gg * gg) * (acc[2][r] + sb[(size_t)2 * GH * TL + o]);
      const float od = og * (1.0f - og) * (acc[3][r] + sb[(size_t)3 * GH * TL + o]);
      const float cnd = fd * cp + fg * cd[r] + id * gg + ig * gd;
      const float tcd = (1.0f - tc * tc) * cnd;
      const float v = od * tc + og * tcd;
      const bool ended = endS[m] != 0;  // the next step of an ended episode starts from zero
      hd[r] = ended ? 0.0f : v;
      cd[r] = ended ? 0.0f : cnd;
      hdT[j][m] = hd[r];
      a1dT[j][m] = a1 > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    f32x16 acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < GH / 2; ++ks)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1dT[2 * ks + hf][n], w1[ks], acc1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const size_t o = (size_t)j * TL + m;
      const float u = ab[(size_t)ACT_U * GH * TL + o];
      udS[m][j] = u > 0.0f ? acc1[r] + sb[(size_t)4 * GH * TL + o] : 0.0f;
    }
    __syncthreads();
    if (wave == 0 && hf < A) {
      float z = out_stat[((size_t)hf * T + t) * N + lane0 + n];
#pragma unroll 8
      for (int q = 0; q < MH; ++q) z = __builtin_fmaf(udS[n][q], w2S[hf][q], z);
      out_dot[((size_t)hf * T + t) * N + lane0 + n] = z;
    }
    __syncthreads();
  }
}

// dz <- (diag(p) - p p^T) out_dot / B with p = exp(log pi_0)  (the metric of the KL's Gauss-Newton form)
__global__ void __launch_bounds__(256) k_seq_fvp_dlogits(TrajDev tr, const float *__restrict__ out_dot,
                                                         const float *__restrict__ lp0, float *__restrict__ dz,
                                                         float inv_B, const int32_t *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  const size_t B = (size_t)tr.T * tr.n;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    const float p0 = rl_expf(lp0[b]), p1 = rl_expf(lp0[B + b]);
    const float d0 = out_dot[b], d1 = out_dot[B + b];
    const float pdz = __builtin_fmaf(p1, d1, __builtin_fmaf(p0, d0, 0.0f));
    dz[b] = p0 * (d0 - pdz) * inv_B;
    dz[B + b] = p1 * (d1 - pdz) * inv_B;
  }
}

void launch_gru_tangent(rl_traj *traj, const rl_mlp *mod, const float *d_tangent, uint64_t B_total,
                        const int32_t *d_skip) {
  rl_engine *e = traj->eng;
  const SeqDev &q = traj->seq;
  RL_REQUIRE(mod->out_dim == 2, "Fisher-vector products are for 2-action policies");
  uint32_t blocks = traj->d.T * q.tiles;
  {
    ProfScope ps(e, RL_K_POLICY_FUSED);
    if (mod->kind == RL_MODULE_LSTM_MLP) {
      hipLaunchKernelGGL((k_lstm_tangent_pre<5, 2>), dim3(q.chunks), dim3(256), 0, e->stream, traj->d, d_tangent, q.act,
                         q.dpre, q.succ, q.tiles, blocks, q.blocks_per_chunk, d_skip);
      hipLaunchKernelGGL(k_lstm_tangent_rec<2>, dim3(q.tiles), dim3(256), 0, e->stream, traj->d, mod->d_params, 5, q.act,
                         q.dpre, q.succ, q.out, d_skip);
    } else {
      hipLaunchKernelGGL((k_gru_tangent_pre<5, 2>), dim3(q.chunks), dim3(256), 0, e->stream, traj->d, d_tangent, q.act,
                         q.dpre, q.succ, q.tiles, blocks, q.blocks_per_chunk, d_skip);
      hipLaunchKernelGGL(k_gru_tangent_rec<2>, dim3(q.tiles), dim3(256), 0, e->stream, traj->d, mod->d_params, 5, q.act,
                         q.dpre, q.succ, q.out, d_skip);
    }
  }
  {
    ProfScope ps(e, RL_K_POLICY_PASS);
    hipLaunchKernelGGL(k_seq_fvp_dlogits, dim3(traj->nbB), dim3(256), 0, e->stream, traj->d, q.out, traj->lp0,
                       traj->dz, 1.0f / (float)B_total, d_skip);
  }
}
