// seq_common.hpp — definitions shared by the recurrent-configuration kernels (kernels_seq.hip: lanes, cells, rollout,
// teacher-forced forward; kernels_seq_bwd.hip: backward through time and weight gradients; kernels_seq_fvp.hip: forward-
// mode tangents for Fisher-vector products): tile geometry, workspace array indices, the flat-parameter view.
#pragma once
#include "device_fns.hpp"
#include "kernels.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GH = 128;      // GRU hidden width (ChainConfig::hidden_dim, chain.rs:28)
constexpr int MH = 128;      // MLP hidden width (MlpConfig::default)
constexpr int TL = 32;       // lanes per tile
constexpr int TLS = TL + 16;  // LDS row stride of the [k][m] operand buffers read as 16x16x4 A operands: the four
                              // k-rows of an instruction start 48 floats apart = banks 0 / 48 / 32 / 16: no conflict
// Per (step, tile) block the training forward records SEQ_ARR arrays of GH x TL floats and the backward DPRE_ARR (the strides
// are those of the LSTM, the larger of the two cells; kernels.hpp: RL_SEQ_ACT_ARRAYS / RL_SEQ_DPRE_ARRAYS).
//   GRU  record: r, z, n, gh_n, h_prev, relu(h'), u                        backward: d pre_r, d pre_z, d pre_n, d pre_n * r, d u_pre
//   LSTM record: i, f, g, o, h_prev, relu(h'), u, c_prev, tanh(c')         backward: d pre_i, d pre_f, d pre_g, d pre_o, d u_pre, d relu(h')
constexpr int SEQ_ARR = RL_SEQ_ACT_ARRAYS;
constexpr int DPRE_ARR = RL_SEQ_DPRE_ARRAYS;
// Inside one [GH][TL] record array the two M-tiles (samples 0-15, 16-31) are separate halves, [half][unit][16]: the four
// samples of an M-tile a lane owns are 16 bytes, and the 16 units of a wave make 1 KB contiguous per store instruction —
// full 128-byte lines.  ([unit][32] rows made every store 16 segments of 64 bytes, the other half of each line written
// a phase later: the training forward then sat on its record stores at 3.3 TB/s — with the stores removed it ran in
// 1.0 ms instead of 1.5.)
constexpr int REC_HALF = GH * 16;  // floats per half of a record array
__host__ __device__ constexpr uint32_t rec_at(int j, int m) { return (uint32_t)((m >> 4) * REC_HALF + j * 16 + (m & 15)); }
enum { ACT_R = 0, ACT_Z = 1, ACT_N = 2, ACT_GHN = 3, ACT_HPREV = 4, ACT_A1 = 5, ACT_U = 6 };
enum { LACT_I = 0, LACT_F = 1, LACT_G = 2, LACT_O = 3, LACT_CPREV = 7, LACT_TC = 8 };  // 4, 5, 6 as above
enum { DPRE_DU = 4, DPRE_DA1 = 5 };

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

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int W16 = 8;  // waves per tile
// sample owned by accumulator register i of M-tile mt in lane group g4
__device__ __forceinline__ int acc16_row(int mt, int i, int g4) { return 16 * mt + 4 * g4 + i; }

static inline uint32_t cdiv_s(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }
