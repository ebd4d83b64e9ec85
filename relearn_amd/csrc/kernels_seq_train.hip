// kernels_seq_train.hip — the GRU chain's training passes (the 90 gradient evaluations of a period in BASELINE.json
// configs[4]) with the recurrence on the bf16 matrix pipe, products from three-piece splits at f32 accuracy.
//
// What the hardware dictates (scripts/probe/pipe_overlap.hip, mfma16x32_bf16.hip): the f32 MFMAs of the rollout's cell
// occupy the vector ALU, so the gates' products and their sigmoid / tanh arithmetic add up (38 k cycles per step and
// SIMD against 16 k of matrix work); v_mfma_f32_16x16x32_bf16 runs beside vector work.  An f32 value is the exact sum
// of three bf16 pieces (bf16_tile.hpp) and a product of two bf16 numbers is exact in the f32 accumulator, so
//   h W^T = sum over the piece pairs (h_p, W_q)
// — the six pairs that matter at f32 accuracy, PIECE_ORDER below — costs 6 bf16 issues of 16 cycles where the f32 form
// costs 8 issues of 32 cycles for the same 32 values of k, and leaves the VALU to the gate arithmetic.  The accumulation
// order differs from the rollout's sequential fma chain (which stays on the f32 kernels of kernels_seq.hip: rollouts, values and GAE are bit-exact with the oracle); the
// training passes are compared with the f64 oracle within f32 tolerances (tests/test_gpu_gru.py), like every
// gradient in this library.
//
// The recurrent kernels carry ONLY the recurrence.  The head (ReLU -> Linear -> ReLU -> Linear) has no dependence
// between steps, so its forward and backward run as block-parallel kernels over all (step, tile) blocks — two
// workgroup barriers and a 128-deep product less on the critical path of every step:
//   k_gru_recur_fwd      t = 0 .. T-1: gates by MFMA, cell on the VALU; records r, z, n, gh_n, h_prev, relu(h')
//   k_seq_head_forward   u = relu(W1 relu(h') + b1) (recorded), out = W2 u + b2          (all blocks in parallel)
//   k_gru_head_backward  d u_pre (on chip), d relu(h'), and the head's own weight gradients  (all blocks in parallel)
//   k_gru_recur_bwd      t = T-1 .. 0: d gates on the VALU, d h_prev = sum_g W_hh[g]^T d gh_g by MFMA; the n gate's
//                        input-side sums
//   k_gru_wgrad_bf16     dW_hh with the sample as contraction index; db_hh and the r / z input-side sums
// Ownership is the rollout cell's: eight waves per tile of 32 lanes, wave w owns units [16w, 16w+16); accumulator
// register i of lane l is (sample 16 mt + 4 (l >> 4) + i, unit 16 w + (l & 15)) for both MFMA shapes: a lane's four
// samples of an M-tile are 16 contiguous bytes of a record array ([half][unit][16], rec_at in seq_common.hpp).
// Register budget per wave: a gate's W_hh fragments are 4 k-blocks x 3 pieces x 4 = 48 registers; two gates (forward)
// or nine of the twelve k-blocks (backward) stay in registers, the rest waits in LDS — the 256-register budget of two
// waves per SIMD does not hold 144 fragment registers next to the accumulators, the prefetched record of the next
// step and the gate arithmetic.
// Reference: gru_cell (src/torch/modules/seq/rnn/gru.rs:30-39), Chain (modules/chain.rs:127-186), and what libtorch's
// autograd does for loss.backward() on them (src/torch/optimizers/coptimizer.rs:13-26).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bf16_tile.hpp"
#include "seq_common.hpp"

namespace {

using bt::Frag;

// An f32 value is the exact sum of three bf16 pieces (round to nearest: |p1| <= 2^-9 |v|, |p2| <= 2^-18 |v|).  Of the
// nine piece pairs of a product a b the six with p + q < 3 are multiplied; the other three together are below
// 2^-26 |a||b|, a quarter of the rounding of ONE f32 product, and under the f32 accumulator's own rounding they do not
// change a single bit of the gradients the tests look at (tests/test_gpu_gru.py::test_piece_products_keep_f32_accuracy:
// the same distance from the f64 oracle as with all nine, 6.7e-8 / 2.9e-8 of max |g|; the f32 fma kernels: 4.7e-8 /
// 1.8e-8).  PIECE_ORDER 5 multiplies all nine.
constexpr int PIECE_ORDER = 3;
constexpr int PIECE_PAIRS = PIECE_ORDER == 3 ? 6 : PIECE_ORDER == 4 ? 8 : 9;
constexpr int LSTM_VALU_PER_MFMA = 6;  // ... of the LSTM's (0, 4, 6, 8 measured: no difference; its schedule is decided by the order of its phases' jobs)
constexpr int VALU_PER_MFMA = 4;  // vector instructions the forward's schedule places after each matrix instruction  // vector instructions the schedule of the forward places between two matrix instructions
// Piece images [sample][unit] in LDS: rows of GH halfwords (256 bytes — the width of the LDS), the 16-byte chunk c of
// row m stored at chunk (c + img_rot(m)) & 15.  A ds_read_b128 is served in four groups of sixteen lanes that are NOT
// contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: MI355X_MICROARCH.md, LDS).  The operand read of lane
// (n16, g4) — row n16, chunk 4 kb + g4 — is conflict-free when the rows' rotations are even and differ within rows
// {0-3, 12-15} and within rows {4-11}.  The padded 272-byte rows this file had before cost every such read twice its
// cycles (0.23 - 0.54 of all LDS cycles were bank conflicts: profiles/r02_pmc_gru_config5_summary.json, with the head
// forward's LDS busy 0.7 of the time).  The rotation keeps the piece writes (lane = unit, lane group = sample)
// conflict-free as well.  scripts/lds_conflicts.py evaluates every pattern used here under the banking rules.
__device__ __forceinline__ int img_rot(int m) { return 4 * (m & 3) + 2 * ((m >> 2) & 1); }
// halfword index of (sample m, unit k) in a [TL][GH] image (k a multiple of 8: the start of an operand chunk)
__device__ __forceinline__ int img_at(int m, int k) { return m * GH + ((k + 8 * img_rot(m)) & (GH - 1)); }
// The backward recurrence has no registers for rotated offsets (one per k-block of a row): its [sample][3 GH] rows
// are padded by 32 bytes instead — row m starts 2 m chunks into the LDS width, which gives the operand reads the same
// property with linear addresses; its piece writes pay for it (lanes l and l + 16 write rows four apart, 25 x 128
// bytes: two cycles instead of one).
constexpr int GROW = 3 * GH + 16;
// Piece images [unit][sample] (the sample is the contraction index): 64-byte rows of TL halfwords, chunk c of row r
// (the eight samples 4 c .. 4 c + 3, 16 + 4 c .. 16 + 4 c + 3: the two accumulator quadruples of ONE lane, so a lane
// parks a row's chunk with one 16-byte write; both operands of the product use the same order) stored at chunk
// c ^ ((r >> 1) & 3): reads (lane = row, lane group = chunk) and writes (the same pattern) are conflict-free.
__device__ __forceinline__ int timg_at(int r, int c) { return r * TL + 8 * (c ^ ((r >> 1) & 3)); }

// sum over the 16 lanes of a DPP row (lanes 16 q .. 16 q + 15), in every one of them: vector-ALU lane moves, no LDS
template <int CTRL>
__device__ __forceinline__ float dpp_lanes(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum16(float v) {
  v += dpp_lanes<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_lanes<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_lanes<0x141>(v);  // row_half_mirror: the other quad of the eight
  v += dpp_lanes<0x140>(v);  // row_mirror: the other eight
  return v;
}

// eight consecutive values of k -> the three operand fragments of their exact bf16 pieces
__device__ __forceinline__ void frags_of8(const float (&v)[8], Frag (&f)[3]) {
  uint32_t p[8][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) bt::split3(v[i], p[i][0], p[i][1], p[i][2]);
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int h = 0; h < 4; ++h) f[q].u[h] = bt::pk(p[2 * h][q], p[2 * h + 1][q]);
}

// acc += sum over the piece pairs (p + q < PIECE_ORDER) of a (3 fragments of A) and b (3 fragments of B)
__device__ __forceinline__ f32x4 mfma_pieces(const Frag (&a)[3], const Frag (&b)[3], f32x4 acc) {
#pragma unroll
  for (int p = 2; p >= 0; --p)  // small terms first
#pragma unroll
    for (int q = 2; q >= 0; --q)
      if (p + q < PIECE_ORDER) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p].v, b[q].v, acc, 0, 0, 0);
  return acc;
}

// Gate functions of the training forward.  The vector ALU bounded this kernel (0.73 busy with rl_sigmoidf / rl_tanhf,
// 0.52 with these; profiles/r02_pmc_gru_config5_summary.json): the gate arithmetic is written for instruction count — e^y through v_exp_f32 (one quarter-rate instruction instead of a
// 14-instruction range reduction and polynomial), quotients through v_rcp_f32 and one Newton step (6 instructions
// instead of the 10 of an IEEE division) — and without branches (a branch ends the basic block, and the scheduler
// interleaves matrix and vector instructions only inside one).  Each result is within 2 ulp of rl_sigmoidf / rl_tanhf
// (include/rl_detmath.h), which stay in the rollout and evaluation kernels: those are compared bit for bit with the
// oracle, the training passes within f32 tolerances.
__device__ __forceinline__ float exp_nonpos_fast(float y) {  // e^y, y <= 0
  return __builtin_amdgcn_exp2f(y * 1.4426950216e+00f);      // (flushes to 0 below 2^-126: the callers add 1)
}
__device__ __forceinline__ float quot_fast(float n, float d) {  // n / d for d in [1, 2]
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  const float q = n * r;
  return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float sigmoid_sel(float x) {
  const float t = exp_nonpos_fast(x < 0.0f ? x : -x);
  return quot_fast(x < 0.0f ? t : 1.0f, 1.0f + t);
}
__device__ __forceinline__ float tanh_sel(float x) {
  const float ax = x < 0.0f ? -x : x;
  const float x2 = ax * ax;
  float p = -8.8632355e-03f;
  p = __builtin_fmaf(x2, p, 2.1869488e-02f);
  p = __builtin_fmaf(x2, p, -5.3968254e-02f);
  p = __builtin_fmaf(x2, p, 1.3333334e-01f);
  p = __builtin_fmaf(x2, p, -3.3333334e-01f);
  const float small = __builtin_fmaf(ax, x2 * p, ax);
  const float t = exp_nonpos_fast(-2.0f * ax);
  const float big = quot_fast(1.0f - t, 1.0f + t);
  const float y = ax < 0.25f ? small : big;
  return rl_f32_from_bits(rl_f32_bits(y) | (rl_f32_bits(x) & 0x80000000u));  // odd; NaN in, NaN out
}

// ---------------------------------------------------------------- forward recurrence
template <int D>
__global__ void __launch_bounds__(W16 * 64, 2)
    k_gru_recur_fwd(TrajDev tr, const float *__restrict__ params, int A, float *__restrict__ act,
                    const int32_t *__restrict__ skip) {
  __shared__ __attribute__((aligned(16))) unsigned short hP[3][TL * GH];  // h as pieces, [sample][unit] (img_at)
  __shared__ float xS[2][TL][9];  // (9: the four samples a wave reads at once are 4 rows apart — 144 bytes, not 128)
  __shared__ int endS[2][TL];
  __shared__ float wiS[3][D + 1][GH];
  __shared__ uint4 wnS[GH / 32][3][W16][64];  // the n gate's W_hh fragments (96 KB): 48 registers the budget does not have
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = gru_params(params, D, A);
  Frag wf[2][GH / 32][3];  // [gate r, z][k-block][piece]: W_hh[gate * GH + j][32 kb + 8 g4 + 0..7]
  float bhh[3];
#pragma unroll
  for (int gte = 0; gte < 3; ++gte) {
    const int row = gte * GH + j;
#pragma unroll
    for (int kb = 0; kb < GH / 32; ++kb) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = g.Whh[(size_t)row * GH + 32 * kb + 8 * g4 + i];
      if (gte < 2) {
        frags_of8(v, wf[gte][kb]);
      } else {
        Frag f[3];
        frags_of8(v, f);
#pragma unroll
        for (int q = 0; q < 3; ++q) wnS[kb][q][wave][lane] = f[q].x;
      }
    }
    bhh[gte] = g.bhh[row];
  }
  // the input projection's weights wait in LDS ([gate][feature | bias][unit]): 18 registers the products need more
  for (int q = threadIdx.x; q < 3 * (D + 1) * GH; q += W16 * 64) {
    const int gte = q / ((D + 1) * GH), d = (q / GH) % (D + 1), u = q % GH;
    wiS[gte][d][u] = d < D ? g.Wih[(size_t)(gte * GH + u) * D + d] : g.bih[gte * GH + u];
  }
  for (int q = threadIdx.x; q < (int)(sizeof(hP) / 4); q += W16 * 64) reinterpret_cast<uint32_t *>(&hP[0][0])[q] = 0u;
  float hown[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) hown[r] = 0.0f;
  const bool io_lane = wave == 0 && lane < TL;
  const uint32_t i_lane = lane0 + (uint32_t)(lane & (TL - 1));
  const size_t plane = (size_t)(T + 1) * N;
  float xin[D];
  int fin = 0;
  auto fetch = [&](uint32_t tt) {  // io lanes: observation and successor code of step tt, into registers
#pragma unroll
    for (int d = 0; d < D; ++d) xin[d] = tr.obs[d * plane + (size_t)tt * N + i_lane];
    fin = tr.flag[(size_t)tt * N + i_lane];
  };
  auto publish = [&](int buf) {
#pragma unroll
    for (int d = 0; d < D; ++d) xS[buf][lane][d] = xin[d];
    endS[buf][lane] = fin != RL_SUCC_CONTINUE;
  };
  if (io_lane) {
    fetch(0);
    publish(0);
  }
  __syncthreads();
  // A step is two phases with a barrier after each, and in both the matrix pipe and the vector ALU work side by side:
  //   phase 1: products of M-tile 1 (step t)       | gate arithmetic of M-tile 0 (step t) -> rows 0-15 of h(t+1)
  //   phase 2: products of M-tile 0 (step t + 1)   | gate arithmetic of M-tile 1 (step t) -> rows 16-31 of h(t+1)
  // (the products of an M-tile need its 16 samples of ALL units, i.e. that M-tile's gate arithmetic of every wave, and
  // nothing of the other M-tile).  The barrier keeps the two waves of a SIMD in the same phase, so the overlap has to
  // come from INSIDE a wave: each phase is one basic block, and the scheduler is asked to place a few vector
  // instructions after every matrix instruction.  (One phase per step — both M-tiles' products, then both M-tiles'
  // gates with M-tile 1's products between M-tile 0's — left the first products and the last gates uncovered: VALU
  // busy 0.52 + matrix pipe 0.48 added up to the step time.)  Rows of the image are rewritten only after the barrier
  // that follows their last read, so one image serves.
  f32x4 acc[3][2];
  auto start = [&](int mt) {
#pragma unroll
    for (int gte = 0; gte < 3; ++gte) acc[gte][mt] = (f32x4){bhh[gte], bhh[gte], bhh[gte], bhh[gte]};
  };
  // operand reads run one k-block ahead of the products
  auto products = [&](int mt) {
    auto frags = [&](int kb, Frag (&a)[3]) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[p].x = *reinterpret_cast<const uint4 *>(&hP[p][img_at(16 * mt + n16, 32 * kb + 8 * g4)]);
    };
#pragma unroll
    for (int kb = 0; kb < GH / 32; ++kb) {
      Frag fa[3], wn[3];
      frags(kb, fa);
#pragma unroll
      for (int q = 0; q < 3; ++q) wn[q].x = wnS[kb][q][wave][lane];
#pragma unroll
      for (int gte = 0; gte < 2; ++gte) acc[gte][mt] = mfma_pieces(fa, wf[gte][kb], acc[gte][mt]);
      acc[2][mt] = mfma_pieces(fa, wn, acc[2][mt]);
    }
  };
  auto gates = [&](int mt, uint32_t t) {
    const int cur = (int)(t & 1);
    float *__restrict__ store = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
    // (uniform block base + 32-bit lane offsets: the addresses stay out of the vector registers)
    const uint32_t row = rec_at(j, 16 * mt + 4 * g4);
    // -DRL_GRU_TIMING_FWD_ONE_PLANE (a TIMING build, results unusable; round 6): the forward of the partition the round-5
    // review asked for records relu(h') only — what that forward would cost
#ifndef RL_GRU_TIMING_FWD_ONE_PLANE
    *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_GHN * GH * TL) + row) = acc[2][mt];
    *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_HPREV * GH * TL) + row) =
        (f32x4){hown[4 * mt], hown[4 * mt + 1], hown[4 * mt + 2], hown[4 * mt + 3]};
#endif
    float wih[3][D], bih[3];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte) {
#pragma unroll
      for (int d = 0; d < D; ++d) wih[gte][d] = wiS[gte][d][j];
      bih[gte] = wiS[gte][D][j];
    }
    f32x4 rv, zv, nv, av;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = acc16_row(mt, i, g4), r = 4 * mt + i;
      float gi[3];
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) {
        float v = bih[gte];
#pragma unroll
        for (int d = 0; d < D; ++d) v = __builtin_fmaf(xS[cur][m][d], wih[gte][d], v);
        gi[gte] = v;
      }
      const float rr = sigmoid_sel(acc[0][mt][i] + gi[0]);
      const float zz = sigmoid_sel(acc[1][mt][i] + gi[1]);
      const float rn = acc[2][mt][i] * rr;
      const float nn = tanh_sel(gi[2] + rn);
      const float dn = hown[r] - nn;
      const float hz = dn * zz;
      const float hv = hz + nn;
      rv[i] = rr;
      zv[i] = zz;
      nv[i] = nn;
      av[i] = hv > 0.0f ? hv : 0.0f;
      // the state the next step starts from: zero after an episode end (SeqPacked restarts per episode)
      const float hn = endS[cur][m] != 0 ? 0.0f : hv;
      hown[r] = hn;
      uint32_t p0, p1, p2;
      bt::split3(hn, p0, p1, p2);
      hP[0][img_at(m, j)] = (unsigned short)p0;
      hP[1][img_at(m, j)] = (unsigned short)p1;
      hP[2][img_at(m, j)] = (unsigned short)p2;
    }
#ifndef RL_GRU_TIMING_FWD_ONE_PLANE
    *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_R * GH * TL) + row) = rv;
    *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_Z * GH * TL) + row) = zv;
    *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_N * GH * TL) + row) = nv;
#else
    asm volatile("" ::"v"(rv), "v"(zv), "v"(nv));  // (the values are still computed)
#endif
    *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_A1 * GH * TL) + row) = av;
  };
  // the order asked of the scheduler for a phase: one matrix instruction, a few vector ones (PHASE: the two phases share
  // a basic block, and groups of one sync id are matched against the whole block)
  auto interleave = [&](auto phase) {
    constexpr int PHASE = decltype(phase)::value;
#pragma unroll
    for (int k = 0; k < (GH / 32) * 3 * PIECE_PAIRS; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, PHASE);              // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, PHASE);  // VALU
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  start(0);
  products(0);  // (h = 0: the biases)
  for (uint32_t t = 0; t < T; ++t) {
    if (io_lane && t + 1 < T) fetch(t + 1);  // lands under the products
    __builtin_amdgcn_sched_barrier(0);
    start(1);
    gates(0, t);
    products(1);
    interleave(std::integral_constant<int, 0>{});
    __syncthreads();  // rows 0-15 of h(t+1) are complete; every wave has read rows 16-31 of h(t)
    // (the branch, always taken, gives phase 2 a basic block of its own: with both phases in one block the scheduler
    // honoured the requested order for one of them only)
    int taken;
    asm volatile("s_mov_b32 %0, 1" : "=s"(taken));
    if (taken != 0) {
      start(0);
      products(0);  // (after the last step: products nobody reads — a branch on t here would split the phase's block)
      gates(1, t);
      interleave(std::integral_constant<int, 1>{});
    }
    if (io_lane && t + 1 < T) publish((int)((t + 1) & 1));
    __syncthreads();  // rows 16-31 of h(t+1) and the inputs of step t + 1 are complete; rows 0-15 have been read
  }
}

// ---------------------------------------------------------------- forward recurrence of the LSTM chain
// The same two-phase step as k_gru_recur_fwd with four gates.  Their W_hh piece fragments are 4 x 98 KB per tile: the
// GRU's partition (eight waves of 256 registers, two gates in registers, one in LDS) cannot hold them — two gates in LDS
// are 196 KB — so a tile is FOUR waves, one per SIMD, with the 512-register budget: wave w owns the units [32 w, 32 w +
// 32) as two groups of 16 (u = 0, 1).  Of the twelve (gate, piece) fragment sets, eight stay in registers (256: what the
// accumulation-register half holds — the vector ALU addresses the other 256 only) and four wait in LDS (128 KB: the o
// gate and the g gate's third piece); the input projection's weights of a lane's two units are registers as well.  State per lane: c of its 2 x 8 (unit, sample) pairs and h (the record's h_prev); h as pieces in the
// LDS image every wave's products read.  Gate functions: the fast forms of the GRU's training forward; records: i, f, g,
// o, h_prev, relu(h'), c_prev, tanh(c') — the arrays lstm_cell16 (kernels_seq.hip) writes, read by k_lstm_bptt and the
// head / weight-gradient kernels.
template <int D>
__global__ void __launch_bounds__(256, 1)
    k_lstm_recur_fwd(TrajDev tr, const float *__restrict__ params, int A, float *__restrict__ act,
                     const int32_t *__restrict__ skip) {
  constexpr int LW = 4;  // waves per tile
  __shared__ __attribute__((aligned(16))) unsigned short hP[3][TL * GH];  // h as pieces, [sample][unit] (img_at)
  __shared__ float xS[2][TL][9];
  __shared__ int endS[2][TL];
  // W_hh fragments that do not fit the registers: the o gate's three pieces and the third piece of the g gate (128 KB)
  __shared__ uint4 wlS[GH / 32][4][LW][2][64];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = seq_params(params, D, A, 4);
  // in registers (256 of the accumulation-register half): the gates i, f with all pieces, the g gate's first two
  Frag wf[2][2][GH / 32][3];  // [gate i, f][unit group][k-block][piece]: W_hh[gate * GH + j][32 kb + 8 g4 + 0..7]
  Frag wg[2][GH / 32][2];     // [unit group][k-block][piece 0, 1] of the g gate
  float bhh[4][2], wih[4][2][D], bih[4][2];  // (the input projection's weights of this lane's two units: registers too)
#pragma unroll
  for (int gte = 0; gte < 4; ++gte)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = gte * GH + 32 * wave + 16 * u + n16;
#pragma unroll
      for (int kb = 0; kb < GH / 32; ++kb) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = g.Whh[(size_t)row * GH + 32 * kb + 8 * g4 + i];
        Frag f[3];
        frags_of8(v, f);
        if (gte < 2) {
#pragma unroll
          for (int q = 0; q < 3; ++q) wf[gte][u][kb][q] = f[q];
        } else if (gte == 2) {
          wg[u][kb][0] = f[0];
          wg[u][kb][1] = f[1];
          wlS[kb][3][wave][u][lane] = f[2].x;
        } else {
#pragma unroll
          for (int q = 0; q < 3; ++q) wlS[kb][q][wave][u][lane] = f[q].x;
        }
      }
      bhh[gte][u] = g.bhh[row];
      bih[gte][u] = g.bih[row];
#pragma unroll
      for (int d = 0; d < D; ++d) wih[gte][u][d] = g.Wih[(size_t)row * D + d];
    }
  for (int q = threadIdx.x; q < (int)(sizeof(hP) / 4); q += LW * 64) reinterpret_cast<uint32_t *>(&hP[0][0])[q] = 0u;
  // (h itself is not kept: a step records the h it leaves behind as the NEXT step's h_prev — 16 registers less)
  float cown[2][8];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) cown[u][r] = 0.0f;
  {  // h_prev of step 0
    float *__restrict__ store0 = act + (size_t)tile * SEQ_ARR * GH * TL;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        *reinterpret_cast<f32x4 *>(store0 + (uint32_t)(ACT_HPREV * GH * TL) + rec_at(32 * wave + 16 * u + n16, 16 * mt + 4 * g4)) =
            (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  }
  const bool io_lane = wave == 0 && lane < TL;
  const uint32_t i_lane = lane0 + (uint32_t)(lane & (TL - 1));
  const size_t plane = (size_t)(T + 1) * N;
  float xin[D];
  int fin = 0;
  auto fetch = [&](uint32_t tt) {
#pragma unroll
    for (int d = 0; d < D; ++d) xin[d] = tr.obs[d * plane + (size_t)tt * N + i_lane];
    fin = tr.flag[(size_t)tt * N + i_lane];
  };
  auto publish = [&](int buf) {
#pragma unroll
    for (int d = 0; d < D; ++d) xS[buf][lane][d] = xin[d];
    endS[buf][lane] = fin != RL_SUCC_CONTINUE;
  };
  if (io_lane) {
    fetch(0);
    publish(0);
  }
  __syncthreads();
  f32x4 acc[4][2][2];  // [gate][unit group][M-tile]
  auto start = [&](int mt) {
#pragma unroll
    for (int gte = 0; gte < 4; ++gte)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[gte][u][mt] = (f32x4){bhh[gte][u], bhh[gte][u], bhh[gte][u], bhh[gte][u]};
  };
  // (the register-resident fragments are asked into the accumulation-register half at every use: matrix instructions read
  // their B operand from either half, the vector ALU only from the other — left to itself the allocator keeps the 256
  // fragment registers in the vector half and spills the arithmetic around them)
  // (once per phase, ahead of its arithmetic: an asm statement inside the products would keep the scheduler from mixing
  // them with the gate arithmetic)
  auto pin = [](Frag &f) { asm volatile("" : "+a"(f.v)); };
  auto pin_all = [&]() {
#pragma unroll
    for (int kb = 0; kb < GH / 32; ++kb)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          pin(wf[0][u][kb][q]);
          pin(wf[1][u][kb][q]);
        }
        pin(wg[u][kb][0]);
        pin(wg[u][kb][1]);
      }
  };
  auto products = [&](int mt) {
#pragma unroll
    for (int kb = 0; kb < GH / 32; ++kb) {
      Frag fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fa[p].x = *reinterpret_cast<const uint4 *>(&hP[p][img_at(16 * mt + n16, 32 * kb + 8 * g4)]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        Frag wo[3], wgg[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) wo[q].x = wlS[kb][q][wave][u][lane];
        wgg[0] = wg[u][kb][0];
        wgg[1] = wg[u][kb][1];
        wgg[2].x = wlS[kb][3][wave][u][lane];
#pragma unroll
        for (int gte = 0; gte < 2; ++gte) acc[gte][u][mt] = mfma_pieces(fa, wf[gte][u][kb], acc[gte][u][mt]);
        acc[2][u][mt] = mfma_pieces(fa, wgg, acc[2][u][mt]);
        acc[3][u][mt] = mfma_pieces(fa, wo, acc[3][u][mt]);
      }
    }
  };
  auto gates = [&](int mt, uint32_t t) {
    const int cur = (int)(t & 1);
    float *__restrict__ store = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
    float *__restrict__ store_next = act + ((size_t)(t + 1 < T ? t + 1 : t) * tiles + tile) * SEQ_ARR * GH * TL;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 32 * wave + 16 * u + n16;
      const uint32_t row = rec_at(j, 16 * mt + 4 * g4);
      f32x4 iv, fv, gv, ov, pv, av, cv, tv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = acc16_row(mt, i, g4), r = 4 * mt + i;
        float pre[4];
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) {
          float v = bih[gte][u];
#pragma unroll
          for (int d = 0; d < D; ++d) v = __builtin_fmaf(xS[cur][m][d], wih[gte][u][d], v);
          pre[gte] = acc[gte][u][mt][i] + v;
        }
        const float ig = sigmoid_sel(pre[0]), fg = sigmoid_sel(pre[1]), gg = tanh_sel(pre[2]), og = sigmoid_sel(pre[3]);
        const float cprev = cown[u][r];
        const float cn = __builtin_fmaf(fg, cprev, ig * gg);
        const float tc = tanh_sel(cn);
        const float hv = og * tc;
        iv[i] = ig;
        fv[i] = fg;
        gv[i] = gg;
        ov[i] = og;
        av[i] = hv > 0.0f ? hv : 0.0f;
        cv[i] = cprev;
        tv[i] = tc;
        // the state the next step starts from: zero after an episode end
        const bool ended = endS[cur][m] != 0;
        const float hn = ended ? 0.0f : hv;
        pv[i] = hn;
        cown[u][r] = ended ? 0.0f : cn;
        uint32_t p0, p1, p2;
        bt::split3(hn, p0, p1, p2);
        hP[0][img_at(m, j)] = (unsigned short)p0;
        hP[1][img_at(m, j)] = (unsigned short)p1;
        hP[2][img_at(m, j)] = (unsigned short)p2;
      }
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(LACT_I * GH * TL) + row) = iv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(LACT_F * GH * TL) + row) = fv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(LACT_G * GH * TL) + row) = gv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(LACT_O * GH * TL) + row) = ov;
      if (t + 1 < T) *reinterpret_cast<f32x4 *>(store_next + (uint32_t)(ACT_HPREV * GH * TL) + row) = pv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_A1 * GH * TL) + row) = av;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(LACT_CPREV * GH * TL) + row) = cv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(LACT_TC * GH * TL) + row) = tv;
    }
  };
  auto interleave = [&](auto phase) {
    constexpr int PHASE = decltype(phase)::value;
#pragma unroll
    for (int k = 0; k < (GH / 32) * 8 * PIECE_PAIRS; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, PHASE);              // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, LSTM_VALU_PER_MFMA, PHASE);  // VALU
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  pin_all();
  start(0);
  products(0);  // (h = 0: the biases)
  for (uint32_t t = 0; t < T; ++t) {
    if (io_lane && t + 1 < T) fetch(t + 1);
    pin_all();
    __builtin_amdgcn_sched_barrier(0);
    // (the products first in program order, as in the second phase: their LDS reads then precede the gates' LDS writes —
    // other rows of the same image, which the compiler cannot tell apart — and the matrix instructions are free to sink
    // between the gate arithmetic; with the gates first they all waited behind the last write)
    start(1);
    products(1);
    gates(0, t);
    interleave(std::integral_constant<int, 0>{});
    __syncthreads();  // rows 0-15 of h(t+1) are complete; every wave has read rows 16-31 of h(t)
    int taken;
    asm volatile("s_mov_b32 %0, 1" : "=s"(taken));
    if (taken != 0) {
      pin_all();
      __builtin_amdgcn_sched_barrier(0);
      start(0);
      products(0);
      gates(1, t);
      interleave(std::integral_constant<int, 1>{});
    }
    if (io_lane && t + 1 < T) publish((int)((t + 1) & 1));
    __syncthreads();
  }
}

// ---------------------------------------------------------------- head forward, all (step, tile) blocks in parallel
// u = relu(b1 + W1 relu(h')) on the bf16 pipe (A operand: the recorded relu(h') block, split into its pieces and laid
// out [sample][unit] in LDS; B: this wave's 16 rows of W1 as register fragments), recorded for the backward;
// out_a = b2_a + sum_q u_q W2[a][q]: the 16 units of a wave summed across its lanes (DPP), the eight waves through LDS.
template <int A>
__global__ void __launch_bounds__(W16 * 64, 2)
    k_seq_head_forward(TrajDev tr, const float *__restrict__ params, int D, int NG, float *__restrict__ act,
                       float *__restrict__ out, uint32_t tiles, uint32_t blocks, const int32_t *__restrict__ skip) {
  __shared__ __attribute__((aligned(16))) unsigned short aK[3][TL * GH];  // relu(h') as pieces (img_at)
  __shared__ float vS[2][W16][TL];  // per output, wave and sample: the wave's 16 terms of W2 u
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const GruParams g = seq_params(params, D, A, NG);
  Frag w1f[GH / 32][3];  // W1[j][32 kb + 8 g4 + 0..7]
#pragma unroll
  for (int kb = 0; kb < GH / 32; ++kb) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = g.W1[(size_t)j * GH + 32 * kb + 8 * g4 + i];
    frags_of8(v, w1f[kb]);
  }
  const float b1 = g.b1[j];
  float w2c[A];
#pragma unroll
  for (int a = 0; a < A; ++a) w2c[a] = g.W2[a * MH + j];
  const int ha = (int)threadIdx.x / TL, hm = (int)threadIdx.x % TL;  // threads < A TL: output, sample
  const float b2v = ha < A ? g.b2[ha] : 0.0f;
  const uint32_t lo = rec_at(j, 4 * g4);
  f32x4 a1n[2];
  auto fetch = [&](uint32_t blk) {
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      a1n[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_A1 * GH * TL) + lo + REC_HALF * mt);
  };
  if (blockIdx.x < blocks) fetch(blockIdx.x);
  for (uint32_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    const uint32_t t = blk / tiles, lane0 = (blk % tiles) * TL;
    float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    __syncthreads();  // the previous block's readers of aK / vS are done
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t p0, p1, p2;
        bt::split3(a1n[mt][i], p0, p1, p2);
        const int m = 16 * mt + 4 * g4 + i;
        aK[0][img_at(m, j)] = (unsigned short)p0;
        aK[1][img_at(m, j)] = (unsigned short)p1;
        aK[2][img_at(m, j)] = (unsigned short)p2;
      }
    if (blk + gridDim.x < blocks) fetch(blk + gridDim.x);  // the next block's operand lands under this block's products
    __syncthreads();
    f32x4 acc1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc1[mt] = (f32x4){b1, b1, b1, b1};
#pragma unroll
    for (int kb = 0; kb < GH / 32; ++kb)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        Frag fa[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          fa[c].x = *reinterpret_cast<const uint4 *>(&aK[c][img_at(16 * mt + n16, 32 * kb + 8 * g4)]);
        acc1[mt] = mfma_pieces(fa, w1f[kb], acc1[mt]);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 uv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float u = acc1[mt][i] > 0.0f ? acc1[mt][i] : 0.0f;
        uv[i] = u;
#pragma unroll
        for (int a = 0; a < A; ++a) {
          const float part = sum16(u * w2c[a]);
          if (n16 == 0) vS[a][wave][acc16_row(mt, i, g4)] = part;
        }
      }
      *reinterpret_cast<f32x4 *>(ab + (uint32_t)(ACT_U * GH * TL) + lo + REC_HALF * mt) = uv;
    }
    __syncthreads();
    if (ha < A) {
      float v = b2v;
#pragma unroll
      for (int w = 0; w < W16; ++w) v += vS[ha][w][hm];
      out[((size_t)ha * T + t) * N + lane0 + hm] = v;
    }
  }
}

// ---------------------------------------------------------------- backward recurrence
// Reads the record of step t and d relu(h') (k_seq_head_backward), carries d h in registers:
//   dh = [episode continues] dh_next + d relu(h')
//   d n = dh (1 - z),  d pre_n = d n (1 - n^2),  d r = d pre_n gh_n,  d pre_r = d r r (1 - r),
//   d pre_z = dh (h_prev - n) z (1 - z),  d gh_n = d pre_n r
//   d h_prev = dh z + W_hh[r]^T d pre_r + W_hh[z]^T d pre_z + W_hh[n]^T d gh_n          (K = 384 on the matrix pipe)
// and writes the four per-step arrays the weight-gradient GEMMs read.
// -DRL_GRU_BWD_TIMESTAMPS (a timing build): waves 0 and 4 of every tile accumulate, over the steps, the 100 MHz clock
// spent in each phase's work and at each barrier; the launcher prints the means (round 6: where a step's 7.2 us go)
#ifdef RL_GRU_BWD_TIMESTAMPS
__device__ unsigned long long g_gru_bwd_ts[1024 * 2 * 6];
#endif
__global__ void __launch_bounds__(W16 * 64, 2)
    k_gru_recur_bwd(TrajDev tr, const float *__restrict__ params, int A, const float *__restrict__ act,
                    float *__restrict__ dpre, float *__restrict__ slab, uint32_t P,
                    const int32_t *__restrict__ skip) {
  constexpr int D = 5;
  constexpr int KB = 3 * GH / 32, KBL = 3;  // k-blocks of the K = 384 product; the last KBL keep their weights in LDS
  __shared__ __attribute__((aligned(16))) unsigned short gP[3][TL][GROW];  // gate gradients as pieces, [sample][gate unit]
  __shared__ uint4 wTS[KBL][3][W16][64];  // 72 KB: 36 registers the budget does not have
  __shared__ float xS[2][TL][9];          // observation features by step parity (the input side's weight gradients)
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const uint32_t tile = blockIdx.x, tiles = gridDim.x, lane0 = tile * TL;
  const GruParams g = gru_params(params, D, A);
  Frag wT[KB - KBL][3];  // [k-block][piece]: W_hh[32 kb + 8 g4 + 0..7][j]   (rows: gate units, r then z then n)
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = g.Whh[(size_t)(32 * kb + 8 * g4 + i) * GH + j];
    if (kb < KB - KBL) {
      frags_of8(v, wT[kb]);
    } else {
      Frag f[3];
      frags_of8(v, f);
#pragma unroll
      for (int q = 0; q < 3; ++q) wTS[kb - (KB - KBL)][q][wave][lane] = f[q].x;
    }
  }
  // The record of one M-tile (16 samples: this lane's four) of a step
  struct HalfIn {
    f32x4 r, z, n, ghn, hp, da1;
    uint32_t end;  // four flag bytes
  };
  // input side of the n gate (dW_ih[n], db_ih[n]) for this lane's unit: sums of d pre_n over its eight samples of every
  // step — d pre_n exists only here (the hidden side stores d pre_n r), so it never goes to HBM; the other gates' input
  // side equals their hidden side and is summed where those arrays are staged (k_gru_wgrad_bf16)
  float dwin[D], dbin = 0.0f;
#pragma unroll
  for (int d = 0; d < D; ++d) dwin[d] = 0.0f;
  const size_t plane = (size_t)(T + 1) * N;
  const uint32_t lo = rec_at(j, 4 * g4);  // + 16 mt: first of the lane's four contiguous samples (32-bit
                                                    // offsets from a uniform block base: no 64-bit address registers)
  auto load = [&](HalfIn &in, uint32_t t, int mt) {
    const size_t blk = (size_t)t * tiles + tile;
    const float *__restrict__ ab = act + blk * SEQ_ARR * GH * TL;
    const float *__restrict__ db = dpre + blk * DPRE_ARR * GH * TL;
    const uint32_t o = lo + REC_HALF * mt;
    // -DRL_GRU_TIMING_BWD_TWO_PLANES (a TIMING build, results unusable; round 6): the backward of the asked partition reads
    // h_prev and d relu(h') only (the gates it would recompute are stand-ins here: the recompute itself is NOT timed)
#ifndef RL_GRU_TIMING_BWD_TWO_PLANES
    in.r = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_R * GH * TL) + o);
    in.z = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_Z * GH * TL) + o);
    in.n = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_N * GH * TL) + o);
    in.ghn = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_GHN * GH * TL) + o);
    in.hp = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_HPREV * GH * TL) + o);
#else
    in.hp = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_HPREV * GH * TL) + o);
    in.r = in.z = in.n = in.ghn = in.hp * 0.25f + 0.5f;
#endif
    in.da1 = *reinterpret_cast<const f32x4 *>(db + (uint32_t)(DPRE_DA1 * GH * TL) + o);
    in.end = *reinterpret_cast<const uint32_t *>(tr.flag + ((size_t)t * N + lane0) + (uint32_t)(16 * mt + 4 * g4));
  };
  // (Round 6, measured and not kept: with every load of the loop issued unconditionally by every thread the compiler's
  // wait counts become exact — no s_waitcnt vmcnt(0) at the head of the phases any more — and the step takes the same
  // 7.0 us: the waits move, the time does not.  What a phase waits for is the LDS: every wave's product reads the same
  // 36 KB of gate-gradient pieces plus its 9 KB of weights, 368 KB per phase and CU = 1.4 us at 128 B per cycle, twice
  // per step (timing build -DRL_GRU_BWD_TIMESTAMPS: products 1.2 - 2.8 us, gate gradients 1.1 - 1.8 us per phase).)
  const bool x_thread = threadIdx.x < TL * D;
  auto load_x = [&](uint32_t t) {  // feature (tid / TL) of sample (tid % TL) of step t
    return tr.obs[(size_t)(threadIdx.x / TL) * plane + (size_t)t * N + lane0 + (threadIdx.x % TL)];
  };
  f32x4 dhc[2], acc[2];  // dhc: d h flowing in from step t + 1; acc: d h_prev of the M-tile — the direct term dh z, then
                          // the K = 384 product
  dhc[0] = dhc[1] = (f32x4){0, 0, 0, 0};
  // gate gradients of one M-tile of step t: pieces into the image rows of that M-tile, the three arrays the weight-gradient
  // kernel reads, the direct term of d h_prev, the n gate's input-side sums
  auto gates = [&](const HalfIn &in, uint32_t t, int mt) {
    float *__restrict__ db = dpre + ((size_t)t * tiles + tile) * DPRE_ARR * GH * TL;
    f32x4 grv, gzv, gnrv;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * mt + 4 * g4 + i;
      const float rr = in.r[i], zz = in.z[i], nn = in.n[i];
      const bool ended = ((in.end >> (8 * i)) & 0xffu) != RL_SUCC_CONTINUE;
      const float dh = (ended ? 0.0f : dhc[mt][i]) + in.da1[i];
      const float dzg = dh * (in.hp[i] - nn);
      const float dn = dh * (1.0f - zz);
      const float dpn = dn * (1.0f - nn * nn);
      const float dr = dpn * in.ghn[i];
      grv[i] = dr * rr * (1.0f - rr);
      gzv[i] = dzg * zz * (1.0f - zz);
      gnrv[i] = dpn * rr;
      acc[mt][i] = dh * zz;
      dbin += dpn;
#pragma unroll
      for (int d = 0; d < D; ++d) dwin[d] = __builtin_fmaf(dpn, xS[t & 1][m][d], dwin[d]);
      const float gv[3] = {grv[i], gzv[i], gnrv[i]};
#pragma unroll
      for (int gte = 0; gte < 3; ++gte) {
        uint32_t p0, p1, p2;
        bt::split3(gv[gte], p0, p1, p2);
        gP[0][m][gte * GH + j] = (unsigned short)p0;
        gP[1][m][gte * GH + j] = (unsigned short)p1;
        gP[2][m][gte * GH + j] = (unsigned short)p2;
      }
    }
    const uint32_t o = lo + REC_HALF * mt;
    *reinterpret_cast<f32x4 *>(db + (uint32_t)(0 * GH * TL) + o) = grv;
    *reinterpret_cast<f32x4 *>(db + (uint32_t)(1 * GH * TL) + o) = gzv;
    *reinterpret_cast<f32x4 *>(db + (uint32_t)(3 * GH * TL) + o) = gnrv;  // (array 2, d pre_n, stays on chip)
  };
  // acc[mt] += W_hh^T (gate gradients of M-tile mt): twelve k-blocks, the last three with their weights from LDS
  auto products = [&](int mt) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      Frag fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[p].x = *reinterpret_cast<const uint4 *>(&gP[p][16 * mt + n16][32 * kb + 8 * g4]);
      if (kb < KB - KBL) {
        acc[mt] = mfma_pieces(fa, wT[kb < KB - KBL ? kb : 0], acc[mt]);
      } else {
        Frag wl[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) wl[q].x = wTS[kb - (KB - KBL)][q][wave][lane];
        acc[mt] = mfma_pieces(fa, wl, acc[mt]);
      }
    }
    dhc[mt] = acc[mt];
  };
  // The two M-tiles of a step are half a step apart: the product of an M-tile needs the gate gradients of ITS 16 samples
  // of all units and nothing of the other M-tile's, so
  //   phase 1 of step t: products of M-tile 0 (step t)  | gate gradients of M-tile 1 (step t)
  //   phase 2 of step t: products of M-tile 1 (step t)  | gate gradients of M-tile 0 (step t - 1)
  // with a barrier after each — the matrix pipe and the vector ALU work side by side in both (one phase for the gradients
  // of both M-tiles and one for both products, as before, added the two up: 7.2 µs per step).  In program order the
  // products come first: their image reads then precede the gates' image writes (other rows, which the compiler cannot
  // tell apart) and the matrix instructions may sink between the gate arithmetic.
  HalfIn inA, inB;  // M-tile 0 of the step whose gates come next, M-tile 1 likewise
  float xin = 0.0f;
  if (x_thread) xS[(T - 1) & 1][threadIdx.x % TL][threadIdx.x / TL] = load_x(T - 1);
  load(inA, T - 1, 0);
  load(inB, T - 1, 1);
  if (x_thread && T > 1) xin = load_x(T - 2);
  __syncthreads();
  gates(inA, T - 1, 0);
  if (T > 1) load(inA, T - 2, 0);
  __syncthreads();
#ifdef RL_GRU_BWD_TIMESTAMPS
  unsigned long long ts_acc[6] = {0, 0, 0, 0, 0, 0}, ts_last = __builtin_amdgcn_s_memrealtime();
#define GRU_TS(k)                                                  \
  do {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    ts_acc[k] += now_ - ts_last;                                   \
    ts_last = now_;                                                \
  } while (0)
#else
#define GRU_TS(k)
#endif
  for (uint32_t t = T; t-- > 0;) {
    if (x_thread && t > 0) xS[(t - 1) & 1][threadIdx.x % TL][threadIdx.x / TL] = xin;  // the features of step t - 1
    if (x_thread && t > 1) xin = load_x(t - 2);
    // (the two waves of a SIMD — w and w + 4 — take the two jobs of a phase in opposite order, so that at any time one
    // feeds the matrix pipe and the other the vector ALU: inside a wave the scheduler keeps the two jobs apart)
    if (wave < W16 / 2) {
      products(0);
      __builtin_amdgcn_sched_barrier(0);
      GRU_TS(0);
      gates(inB, t, 1);
    } else {
      gates(inB, t, 1);
      __builtin_amdgcn_sched_barrier(0);
      GRU_TS(0);
      products(0);
    }
    if (t > 0) load(inB, t - 1, 1);
    GRU_TS(1);
    __syncthreads();  // rows 16-31 of the image of step t are complete; rows 0-15 have been read
    GRU_TS(2);
    if (wave < W16 / 2) {
      products(1);
      __builtin_amdgcn_sched_barrier(0);
      GRU_TS(3);
      if (t > 0) gates(inA, t - 1, 0);
    } else {
      if (t > 0) gates(inA, t - 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      GRU_TS(3);
      products(1);
    }
    if (t > 1) load(inA, t - 2, 0);
    GRU_TS(4);
    __syncthreads();  // rows 0-15 of the image of step t - 1 are complete; rows 16-31 have been read
    GRU_TS(5);
  }
#ifdef RL_GRU_BWD_TIMESTAMPS
  if (lane == 0 && (wave == 0 || wave == 4) && tile < 1024)
    for (int k = 0; k < 6; ++k) g_gru_bwd_ts[(tile * 2 + (wave >> 2)) * 6 + k] = ts_acc[k];
#endif
  // ---- this tile's row of partials: the n gate's rows of W_ih and b_ih (the four lane groups of a wave hold
  // different samples of the same unit)
  float *__restrict__ out = slab + (size_t)tile * P;
  const size_t obih = (size_t)3 * GH * D + (size_t)3 * GH * GH;
  auto over_groups = [](float v) {
    v = v + __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
  };
  const int row = 2 * GH + j;
  dbin = over_groups(dbin);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float v = over_groups(dwin[d]);
    if (g4 == 0) out[(size_t)row * D + d] = v;
  }
  if (g4 == 0) out[obih + row] = dbin;
}

// ---------------------------------------------------------------- head backward, all blocks in parallel, with the
// head's own weight gradients.  Per (step, tile) block:
//   d u_pre = [u > 0] W2^T dz                       (VALU; stays on chip)
//   d relu(h') = [relu(h') > 0] d u_pre W1          (contraction over the 128 units of u, times the piece pairs)
//                                                   -> dpre[DPRE_DA1], the backward recurrence's input
//   dW1 += d u_pre^T relu(h')                       (contraction over the block's 32 samples: one issue per tile and pair)
//   db1 += d u_pre,  dW2 += dz^T u,  db2 += dz      (VALU)
// Both products run on the bf16 pipe with exact three-piece operands (the f32 form kept this kernel at 0.68 of the f32
// matrix rate with the VALU blocked under it).  Three piece images per block in LDS:
//   uJ [sample][unit]  d u_pre with the unit contiguous  (A operand of the first product)
//   uM [unit][sample]  d u_pre with the sample contiguous (A operand of the second)
//   aM [unit][sample]  relu(h')                           (B operand of the second)
// so the weight-gradient kernel of the recurrence reads neither u, relu(h') nor d u_pre.  A workgroup walks blocks
// blk = blockIdx.x, + gridDim.x, ... and writes one row of f32 partials (head columns only) for the f64 reduction.
template <int A>
__global__ void __launch_bounds__(W16 * 64, 2)
    k_gru_head_backward(TrajDev tr, const float *__restrict__ params, int D, int NG, const float *__restrict__ dz,
                        const float *__restrict__ act, float *__restrict__ dpre, float *__restrict__ slab, uint32_t P,
                        uint32_t tiles, uint32_t blocks, const int32_t *__restrict__ skip) {
  __shared__ __attribute__((aligned(16))) unsigned short uJ[3][TL * GH];  // img_at
  __shared__ __attribute__((aligned(16))) unsigned short uM[3][MH * TL];  // timg_at
  __shared__ __attribute__((aligned(16))) unsigned short aM[3][GH * TL];  // timg_at
  __shared__ float dzS[2][TL];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const size_t B = (size_t)T * N;
  const GruParams g = seq_params(params, D, A, NG);  // (NG gate blocks in front of the head: 3 = GRU, 4 = LSTM)
  Frag w1f[MH / 32][3];  // B operand of d relu(h'): W1[32 kb + 8 g4 + 0..7][j] (rows: units of u; column: this lane's k)
#pragma unroll
  for (int kb = 0; kb < MH / 32; ++kb) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = g.W1[(size_t)(32 * kb + 8 * g4 + i) * GH + j];
    frags_of8(v, w1f[kb]);
  }
  float w2c[A];
#pragma unroll
  for (int a = 0; a < A; ++a) w2c[a] = g.W2[a * MH + j];
  f32x4 accw[GH / 16];  // dW1[16 wave + 4 g4 + i][16 nt + n16]
#pragma unroll
  for (int nt = 0; nt < GH / 16; ++nt) accw[nt] = (f32x4){0, 0, 0, 0};
  float db1 = 0.0f, dw2[A], db2 = 0.0f;
#pragma unroll
  for (int a = 0; a < A; ++a) dw2[a] = 0.0f;
  const uint32_t lo = rec_at(j, 4 * g4);
  f32x4 a1n[2], un[2];
  float dzn = 0.0f;
  auto fetch = [&](uint32_t blk) {
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      a1n[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_A1 * GH * TL) + lo + REC_HALF * mt);
      un[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_U * GH * TL) + lo + REC_HALF * mt);
    }
    if (wave == 0 && lane < A * TL) {  // lane = (output a = lane >> 5, sample lane & 31)
      const uint32_t t = blk / tiles, lane0 = (blk % tiles) * TL;
      dzn = dz[(size_t)(lane >> 5) * B + (size_t)t * N + lane0 + (lane & 31)];
    }
  };
  // this lane's eight samples of unit j (its two accumulator quadruples) -> chunk g4 of row j of each piece image
  auto park8 = [&](unsigned short (*img)[MH * TL], const f32x4 &v0, const f32x4 &v1) {
    uint32_t p[8][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bt::split3(v0[i], p[i][0], p[i][1], p[i][2]);
      bt::split3(v1[i], p[4 + i][0], p[4 + i][1], p[4 + i][2]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<uint4 *>(&img[c][timg_at(j, g4)]) =
          make_uint4(bt::pk(p[0][c], p[1][c]), bt::pk(p[2][c], p[3][c]), bt::pk(p[4][c], p[5][c]), bt::pk(p[6][c], p[7][c]));
  };
  if (blockIdx.x < blocks) fetch(blockIdx.x);
  for (uint32_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    float *__restrict__ db = dpre + (size_t)blk * DPRE_ARR * GH * TL;
    __syncthreads();  // the previous block's readers of the LDS images are done
    if (wave == 0 && lane < A * TL) {
      dzS[lane >> 5][lane & 31] = dzn;
      db2 += dzn;
    }
    f32x4 a1c[2], uc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      a1c[mt] = a1n[mt];
      uc[mt] = un[mt];
    }
    park8(aM, a1c[0], a1c[1]);
    __syncthreads();  // dz of the block is visible
    if (blk + gridDim.x < blocks) fetch(blk + gridDim.x);  // the next block's operands land under this block's products
    f32x4 duv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * mt + 4 * g4 + i;
        float du = 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
          du = __builtin_fmaf(dzS[a][m], w2c[a], du);
          dw2[a] = __builtin_fmaf(dzS[a][m], uc[mt][i], dw2[a]);
        }
        du = uc[mt][i] > 0.0f ? du : 0.0f;
        db1 += du;
        duv[mt][i] = du;
        uint32_t p0, p1, p2;
        bt::split3(du, p0, p1, p2);
        uJ[0][img_at(m, j)] = (unsigned short)p0;
        uJ[1][img_at(m, j)] = (unsigned short)p1;
        uJ[2][img_at(m, j)] = (unsigned short)p2;
      }
    }
    park8(uM, duv[0], duv[1]);
    __syncthreads();
    // d relu(h')[m][k = j] = sum over the units q of d u_pre[m][q] W1[q][k]
    f32x4 acc1[2];
    acc1[0] = acc1[1] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < MH / 32; ++kb)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        Frag fa[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          fa[c].x = *reinterpret_cast<const uint4 *>(&uJ[c][img_at(16 * mt + n16, 32 * kb + 8 * g4)]);
        acc1[mt] = mfma_pieces(fa, w1f[kb], acc1[mt]);
      }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 dav;
#pragma unroll
      for (int i = 0; i < 4; ++i) dav[i] = a1c[mt][i] > 0.0f ? acc1[mt][i] : 0.0f;
      *reinterpret_cast<f32x4 *>(db + (uint32_t)(DPRE_DA1 * GH * TL) + lo + REC_HALF * mt) = dav;
    }
    // dW1[row][col] += sum over the 32 samples of d u_pre[row][m] relu(h')[col][m]: A rows = this wave's 16 units
    {
      Frag fa[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) fa[c].x = *reinterpret_cast<const uint4 *>(&uM[c][timg_at(j, g4)]);
#pragma unroll
      for (int nt = 0; nt < GH / 16; ++nt) {
        Frag fb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) fb[c].x = *reinterpret_cast<const uint4 *>(&aM[c][timg_at(16 * nt + n16, g4)]);
        accw[nt] = mfma_pieces(fa, fb, accw[nt]);
      }
    }
  }
  // ---- this workgroup's row of partials (head columns)
  const size_t oW1 = (size_t)NG * GH * D + (size_t)NG * GH * GH + 2 * NG * GH, ob1 = oW1 + (size_t)MH * GH, oW2 = ob1 + MH,
               ob2 = oW2 + (size_t)A * MH;
  float *__restrict__ out = slab + (size_t)blockIdx.x * P;
#pragma unroll
  for (int nt = 0; nt < GH / 16; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[oW1 + (size_t)(16 * wave + 4 * g4 + i) * GH + 16 * nt + n16] = accw[nt][i];
  // the four lane groups of a wave hold different samples of the same units
  db1 = db1 + __shfl_xor(db1, 16, 64);
  db1 = db1 + __shfl_xor(db1, 32, 64);
  if (g4 == 0) out[ob1 + j] = db1;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    float v = dw2[a] + __shfl_xor(dw2[a], 16, 64);
    v = v + __shfl_xor(v, 32, 64);
    if (g4 == 0) out[oW2 + (size_t)a * MH + j] = v;
  }
  if (wave == 0) {
    float v = lane < A * TL ? db2 : 0.0f;  // half a of the wave holds the per-sample-slot sums of output a
#pragma unroll
    for (int s2 = 16; s2 > 0; s2 >>= 1) v += __shfl_xor(v, s2, 64);
    if ((lane & 31) == 0 && (lane >> 5) < A) out[ob2 + (lane >> 5)] = v;
  }
}

// ---------------------------------------------------------------- weight gradients of the recurrence
// dW_hh[g][k] = sum over (step, lane) of d gh_g . h_prev_k   (g over the 384 gate units: d pre_r, d pre_z, d pre_n r)
// with the SAMPLE as the contraction index: the records are [half][unit][16] arrays, so eight consecutive samples of a row
// are one operand fragment — no transposition.  A workgroup walks a contiguous run of (step, tile) blocks in HALF
// blocks of 16 samples.  Every element is split into its three bf16 pieces ONCE (each thread stages 1/512 of a half)
// and parked in LDS in operand layout, in one of two buffers: the pieces of half h + 1 are produced (VALU) while the
// matrix pipe works on half h — the two waves of a SIMD take the two jobs in opposite order (half_step below; with one
// buffer per block, or both waves in the same order, the jobs took turns: matrix pipe 0.55 busy, VALU 0.34).  Wave w
// accumulates output rows [192 (w >> 2), + 192) x columns [32 (w & 3), + 32): six 32x32 tiles, PIECE_PAIRS issues of
// v_mfma_f32_32x32x16_bf16 per tile and half, two tiles interleaved.  db_hh, and dW_ih / db_ih of the r and z gates
// (their input side equals their hidden side), are sums the staging threads keep for the rows they stage (thread q
// stages row q >> 2 of every gate in every half); the n gate's input side comes from the backward recurrence.  One row
// of f32 partials per workgroup (columns of the recurrent parameters only; the head's come from k_gru_head_backward).
// [row][16 samples] piece images: 32-byte rows, the two 16-byte chunks of rows 8 .. 15 (mod 16) swapped — operand reads
// (lane = row, upper half-wave = upper chunk) and the staging writes (four lanes per row) are conflict-free
// (scripts/lds_conflicts.py; the padded 48-byte rows before cost the writes twice their cycles)
__device__ __forceinline__ int wimg_at(int row, int hw) { return row * 16 + (hw ^ (((row >> 3) & 1) << 3)); }
// NG gate blocks: 3 = GRU (hidden-side deltas: d pre_r, d pre_z, d pre_n r = dpre arrays 0, 1, 3; the input side of r and z
// equals the hidden side), 4 = LSTM (d pre_i, f, g, o = arrays 0 .. 3; input side = hidden side for all four: this kernel
// then forms every recurrent column, eight 32x32 tiles per wave)
template <int D, int NG>
__global__ void __launch_bounds__(W16 * 64, 2)
    k_gru_wgrad_bf16(TrajDev tr, const float *__restrict__ act, const float *__restrict__ dpre,
                     float *__restrict__ slab, uint32_t P, uint32_t tiles, uint32_t blocks, uint32_t blocks_per_chunk,
                     const int32_t *__restrict__ skip) {
  constexpr int NIN = NG == 3 ? 2 : NG;  // gates whose input-side sums are formed here
  __shared__ __attribute__((aligned(16))) unsigned short AP[2][3][NG * GH * 16];  // d gh pieces, by half parity (wimg_at)
  __shared__ __attribute__((aligned(16))) unsigned short BP[2][3][GH * 16];      // h_prev pieces
  __shared__ float xS[2][TL][17];  // observations, by block parity: thread q holds column q / TL of sample q % TL (the
                                   // columns >= D are padding: every thread writes, no branch; 17: samples 4 rows apart
                                   // on different banks)
  if (skip != nullptr && *skip != 0) return;
  const int q = threadIdx.x, lane = q & 63, wave = q >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const int nt = wave & 3, mset = wave >> 2;
  bt::f32x16 acc[2 * NG];
#pragma unroll
  for (int i = 0; i < 2 * NG; ++i) acc[i] = (bt::f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // staging of a half: thread q takes the 16-byte piece (row q >> 2, samples 4 (q & 3) .. + 3 of the half) of each of
  // the three gate arrays and of h_prev
  const int srow = q >> 2, scol = 4 * (q & 3);
  float dwih[NIN][D], dbh[NG];
#pragma unroll
  for (int g3 = 0; g3 < NG; ++g3) dbh[g3] = 0.0f;
#pragma unroll
  for (int g2 = 0; g2 < NIN; ++g2)
#pragma unroll
    for (int d = 0; d < D; ++d) dwih[g2][d] = 0.0f;
  const uint32_t N = tr.n, T = tr.T;
  const size_t plane = (size_t)(T + 1) * N;
  const uint32_t b0 = blockIdx.x * blocks_per_chunk;
  const uint32_t b1 = b0 + blocks_per_chunk < blocks ? b0 + blocks_per_chunk : blocks;
  const uint32_t n_half = b1 > b0 ? 2 * (b1 - b0) : 0;  // (a chunk past the end writes a row of zeros)
  // register stages of the global loads: half h lands in slot h & 1, two halves before its pieces are produced
  struct Slot {
    f32x4 g[NG], hB;
    float xn;
  };
  Slot slot[2];
  slot[0].xn = slot[1].xn = 0.0f;
  // half h of the chunk: block b0 + h / 2, samples 16 (h & 1) .. + 15.  `steady`: the block after it exists as well (no
  // branch then: a branch would end the basic block the products and the staging arithmetic are interleaved in)
  auto fetch = [&](uint32_t h, Slot &sl, bool steady) {
    const uint32_t blk = b0 + (h >> 1);
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    const float *__restrict__ db = dpre + (size_t)blk * DPRE_ARR * GH * TL;
    const uint32_t o = rec_at(srow, 16 * (int)(h & 1) + scol);
#pragma unroll
    for (int g3 = 0; g3 < NG; ++g3)  // (GRU: array 3 is the hidden side of the n gate)
      sl.g[g3] = *reinterpret_cast<const f32x4 *>(db + (uint32_t)((NG == 3 && g3 == 2 ? 3 : g3) * GH * TL) + o);
    sl.hB = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_HPREV * GH * TL) + o);
    if ((h & 1) == 0 && (steady || blk + 1 < b1)) {  // feature q / TL of sample q % TL of the NEXT block (its sums run
      const uint32_t t = (blk + 1) / tiles, lane0 = ((blk + 1) % tiles) * TL;  // while its pieces are produced)
      const int xf = q / TL < D ? q / TL : D - 1;  // (threads beyond the D features load a value nobody reads)
      sl.xn = tr.obs[(size_t)xf * plane + (size_t)t * N + lane0 + (q % TL)];
    }
  };
  // four samples of one row -> their pieces, 8 bytes into each of the three piece images of a buffer
  auto park = [&](auto &img, int row, const f32x4 &v) {
    uint32_t p[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) bt::split3(v[i], p[i][0], p[i][1], p[i][2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint64_t w = (uint64_t)bt::pk(p[0][c], p[1][c]) | ((uint64_t)bt::pk(p[2][c], p[3][c]) << 32);
      *reinterpret_cast<uint64_t *>(&img[c][wimg_at(row, scol)]) = w;
    }
  };
  // park half h (in `sl`) into LDS buffer h & 1, and add it to the sums of the rows this thread stages (the block's x
  // was published two halves ago)
  auto stage = [&](uint32_t h, const Slot &sl) {
    const int buf = (int)(h & 1);
#pragma unroll
    for (int g3 = 0; g3 < NG; ++g3) park(AP[buf], g3 * GH + srow, sl.g[g3]);
    park(BP[buf], srow, sl.hB);
    if ((h & 1) == 0) xS[((h >> 1) + 1) & 1][q % TL][q / TL] = sl.xn;  // x of the next block
    const int xb = (int)((h >> 1) & 1);
#pragma unroll
    for (int g3 = 0; g3 < NG; ++g3)
#pragma unroll
      for (int i = 0; i < 4; ++i) dbh[g3] += sl.g[g3][i];
#pragma unroll
    for (int g2 = 0; g2 < NIN; ++g2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < D; ++d)
          dwih[g2][d] = __builtin_fmaf(sl.g[g2][i], xS[xb][16 * (h & 1) + scol + i][d], dwih[g2][d]);
  };
  auto products = [&](int buf) {  // contraction over the 16 samples of the half in buffer `buf`, two tiles interleaved
    Frag fb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) fb[c].x = *reinterpret_cast<const uint4 *>(&BP[buf][c][wimg_at(32 * nt + n, 8 * hf)]);
#pragma unroll
    for (int i = 0; i < 2 * NG; i += 2) {
      Frag fa[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          fa[u][c].x = *reinterpret_cast<const uint4 *>(&AP[buf][c][wimg_at(64 * NG * mset + 32 * (i + u) + n, 8 * hf)]);
#pragma unroll
      for (int pa = 2; pa >= 0; --pa)
#pragma unroll
        for (int pb = 2; pb >= 0; --pb)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            if (pa + pb < PIECE_ORDER) acc[i + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[u][pa].v, fb[pb].v, acc[i + u], 0, 0, 0);
    }
  };
  // one half: the pieces (and sums) of the NEXT half from slot `sl` into the other buffer, the fetch of the half after
  // that into the slot just freed, and the 54 products of this half — issued as one region, one matrix instruction then
  // a few vector instructions
  // The two waves of a SIMD (w and w + 4) take the two jobs of a half in opposite order, so that at any time one of them
  // feeds the matrix pipe and the other the vector ALU.  (Asking the scheduler to interleave the two jobs inside each wave
  // did not survive this loop's shape: the listing showed all matrix instructions in one run, and matrix pipe 0.55 +
  // VALU 0.34 added up to the time.)
  auto half_step = [&](uint32_t h, Slot &sl, int buf, bool steady) {
    if (wave < W16 / 2) {
      if (steady || h + 1 < n_half) stage(h + 1, sl);
      if (steady || h + 3 < n_half) fetch(h + 3, sl, steady);
      __builtin_amdgcn_sched_barrier(0);
      products(buf);
    } else {
      products(buf);
      __builtin_amdgcn_sched_barrier(0);
      if (steady || h + 1 < n_half) stage(h + 1, sl);
      if (steady || h + 3 < n_half) fetch(h + 3, sl, steady);
    }
    __syncthreads();  // buffer `buf` is free for half h + 2, the other buffer holds half h + 1
  };
  // n_half is even (whole blocks)
  if (n_half > 0) {
    {  // x of the first block
      const uint32_t t = b0 / tiles, lane0 = (b0 % tiles) * TL;
      const int xf = q / TL < D ? q / TL : D - 1;
      xS[0][q % TL][q / TL] = tr.obs[(size_t)xf * plane + (size_t)t * N + lane0 + (q % TL)];
    }
    fetch(0, slot[0], false);
    fetch(1, slot[1], false);
    __syncthreads();
    stage(0, slot[0]);
    if (n_half > 2) fetch(2, slot[0], false);
  }
  __syncthreads();
  uint32_t h = 0;
  for (; h + 6 < n_half; h += 2) {  // steady state: every half up to h + 4, and the block after it, exist
    half_step(h, slot[1], 0, true);
    half_step(h + 1, slot[0], 1, true);
  }
  for (; h < n_half; h += 2) {
    half_step(h, slot[1], 0, false);
    half_step(h + 1, slot[0], 1, false);
  }
  // ---- this workgroup's row of partials (recurrent columns)
  float *__restrict__ out = slab + (size_t)blockIdx.x * P;
  const size_t oWhh = (size_t)NG * GH * D;
#pragma unroll
  for (int i = 0; i < 2 * NG; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[oWhh + (size_t)(64 * NG * mset + 32 * i + acc_row(r, hf)) * GH + 32 * nt + n] = acc[i][r];
  // the four threads q & 3 of a row hold its partial sums over different samples
  const size_t obih = oWhh + (size_t)NG * GH * GH, obhh = obih + NG * GH;
  auto over4 = [](float v) {
    v = v + __shfl_xor(v, 1, 64);
    return v + __shfl_xor(v, 2, 64);
  };
#pragma unroll
  for (int g3 = 0; g3 < NG; ++g3) {
    const int row = g3 * GH + srow;
    const float vb = over4(dbh[g3]);
    if ((q & 3) == 0) {
      out[obhh + row] = vb;
      if (g3 < NIN) out[obih + row] = vb;  // (the GRU's n gate: its input side comes from the backward recurrence)
    }
    if (g3 < NIN) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float v = over4(dwih[g3][d]);
        if ((q & 3) == 0) out[(size_t)row * D + d] = v;
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------- launchers
// teacher-forced training forward of the GRU chain: seq.act (all seven arrays) and d_out [A][T][n]
void launch_gru_train_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip) {
  RL_REQUIRE(traj->d.D == 5, "recurrent forward: built for 5 observation features");
  rl_engine *e = traj->eng;
  const uint32_t tiles = traj->d.n / TL;
  const int A = (int)mod->out_dim;
  hipLaunchKernelGGL(k_gru_recur_fwd<5>, dim3(tiles), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, A, d_act,
                     d_skip);
  launch_seq_train_head_forward(traj, mod, d_out, d_act, d_skip);
}

// teacher-forced training forward of the LSTM chain: the recurrence on the bf16 pipe, then the head
void launch_lstm_train_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip) {
  RL_REQUIRE(traj->d.D == 5, "recurrent forward: built for 5 observation features");
  const uint32_t tiles = traj->d.n / TL;
  hipLaunchKernelGGL(k_lstm_recur_fwd<5>, dim3(tiles), dim3(256), 0, traj->eng->stream, traj->d, mod->d_params,
                     (int)mod->out_dim, d_act, d_skip);
  launch_seq_train_head_forward(traj, mod, d_out, d_act, d_skip);
}

// the head of either chain over all (step, tile) blocks: u (recorded) and the outputs from the recorded relu(h')
void launch_seq_train_head_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip) {
  rl_engine *e = traj->eng;
  const uint32_t tiles = traj->d.n / TL, blocks = traj->d.T * tiles;
  const uint32_t grid = blocks < 2048 ? blocks : 2048;
  const int NG = (int)rl_module_gates(mod->kind);
  if (mod->out_dim == 2)
    hipLaunchKernelGGL(k_seq_head_forward<2>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, NG,
                       d_act, d_out, tiles, blocks, d_skip);
  else
    hipLaunchKernelGGL(k_seq_head_forward<1>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, NG,
                       d_act, d_out, tiles, blocks, d_skip);
}

// head backward of the GRU chain with the head's weight gradients: rows [0, RL_SEQ_HEAD_ROWS) of `d_slab`
void launch_gru_train_head_backward(rl_traj *traj, const rl_mlp *mod, float *d_slab, const int32_t *d_skip) {
  const SeqDev &q = traj->seq;
  const uint32_t blocks = traj->d.T * q.tiles, grid = blocks < RL_SEQ_HEAD_ROWS ? blocks : RL_SEQ_HEAD_ROWS;
  if (grid < RL_SEQ_HEAD_ROWS)  // rows no workgroup writes
    RL_HIP_CHECK(hipMemsetAsync(d_slab + (size_t)grid * mod->P, 0, (size_t)(RL_SEQ_HEAD_ROWS - grid) * mod->P * sizeof(float),
                                traj->eng->stream));
  const int NG = (int)rl_module_gates(mod->kind);
  if (mod->out_dim == 2)
    hipLaunchKernelGGL(k_gru_head_backward<2>, dim3(grid), dim3(W16 * 64), 0, traj->eng->stream, traj->d, mod->d_params,
                       5, NG, traj->dz, q.act, q.dpre, d_slab, (uint32_t)mod->P, q.tiles, blocks, d_skip);
  else
    hipLaunchKernelGGL(k_gru_head_backward<1>, dim3(grid), dim3(W16 * 64), 0, traj->eng->stream, traj->d, mod->d_params,
                       5, NG, traj->dz, q.act, q.dpre, d_slab, (uint32_t)mod->P, q.tiles, blocks, d_skip);
}

// weight gradients of the recurrent parameters: rows [0, chunks) of seq.wg_slab, columns [0, W1)
void launch_gru_train_wgrad(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip) {
  const SeqDev &q = traj->seq;
  const uint32_t blocks = traj->d.T * q.tiles;
  if (mod->kind == RL_MODULE_LSTM_MLP)
    hipLaunchKernelGGL((k_gru_wgrad_bf16<5, 4>), dim3(q.chunks), dim3(W16 * 64), 0, traj->eng->stream, traj->d, q.act,
                       q.dpre, q.wg_slab, (uint32_t)mod->P, q.tiles, blocks, q.blocks_per_chunk, d_skip);
  else
    hipLaunchKernelGGL((k_gru_wgrad_bf16<5, 3>), dim3(q.chunks), dim3(W16 * 64), 0, traj->eng->stream, traj->d, q.act,
                       q.dpre, q.wg_slab, (uint32_t)mod->P, q.tiles, blocks, q.blocks_per_chunk, d_skip);
}

// backward recurrence of the GRU chain (after the head's backward has left d relu(h') in seq.dpre)
void launch_gru_train_recur_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip) {
  const SeqDev &q = traj->seq;
  // its W_ih / b_ih / b_hh partials: one row per tile behind the weight-gradient kernel's rows (the head kernel uses
  // the head columns of the same rows)
  hipLaunchKernelGGL(k_gru_recur_bwd, dim3(q.tiles), dim3(W16 * 64), 0, traj->eng->stream, traj->d, mod->d_params,
                     (int)mod->out_dim, q.act, q.dpre, q.wg_slab + (size_t)q.chunks * mod->P, (uint32_t)mod->P, d_skip);
#ifdef RL_GRU_BWD_TIMESTAMPS
  if (std::getenv("RL_GRU_TS_PRINT")) {
    static int calls = 0;
    if (++calls % 8 == 0) {
      const uint32_t tiles = q.tiles < 1024 ? q.tiles : 1024;
      std::vector<unsigned long long> h(1024 * 2 * 6);
      (void)hipStreamSynchronize(traj->eng->stream);
      (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gru_bwd_ts), h.size() * 8);
      for (int wv = 0; wv < 2; ++wv) {
        double sum[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t tl = 0; tl < tiles; ++tl)
          for (int k = 0; k < 6; ++k) sum[k] += (double)h[(tl * 2 + wv) * 6 + k] * 0.01 / traj->d.T;
        std::fprintf(stderr, "gru bwd ts, wave %d (%s first; us per step, mean over %u tiles): phase 1: %.2f + %.2f, barrier "
                     "%.2f; phase 2: %.2f + %.2f, barrier %.2f\n", 4 * wv, wv == 0 ? "products" : "gates", tiles,
                     sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles, sum[5] / tiles);
      }
    }
  }
#endif
}
