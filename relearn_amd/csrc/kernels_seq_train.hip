// kernels_seq_train.hip — the GRU chain's training passes (the 90 gradient evaluations of a period in BASELINE.json
// configs[4]) with the recurrence on the bf16 matrix pipe, every product exact.
//
// What the hardware dictates (scripts/probe/pipe_overlap.hip, mfma16x32_bf16.hip): the f32 MFMAs of the rollout's cell
// occupy the vector ALU, so the gates' products and their sigmoid / tanh arithmetic add up (38 k cycles per step and
// SIMD against 16 k of matrix work); v_mfma_f32_16x16x32_bf16 runs beside vector work.  An f32 value is the exact sum
// of three bf16 pieces (bf16_tile.hpp) and a product of two bf16 numbers is exact in the f32 accumulator, so
//   h W^T = sum over the nine piece pairs (h_p, W_q)
// costs 9 bf16 issues of 16 cycles where the f32 form costs 8 issues of 32 cycles for the same 32 values of k — and
// leaves the VALU to the gate arithmetic.  The accumulation order differs from the rollout's sequential fma chain
// (which stays on the f32 kernels of kernels_seq.hip: rollouts, values and GAE are bit-exact with the oracle); the
// training passes are compared with the f64 oracle within f32 tolerances (tests/test_gpu_gru.py), like every
// gradient in this library.
//
// The recurrent kernels carry ONLY the recurrence.  The head (ReLU -> Linear -> ReLU -> Linear) has no dependence
// between steps, so its forward and backward run as block-parallel kernels over all (step, tile) blocks — two
// workgroup barriers and a 128-deep product less on the critical path of every step:
//   k_gru_recur_fwd      t = 0 .. T-1: gates by MFMA, cell on the VALU; records r, z, n, gh_n, h_prev, relu(h')
//   k_seq_head_forward   u = relu(W1 relu(h') + b1) (recorded), out = W2 u + b2          (all blocks in parallel)
//   k_seq_head_backward  d u_pre, d relu(h')                  (kernels_seq_bwd.hip; all blocks in parallel)
//   k_gru_recur_bwd      t = T-1 .. 0: d gates on the VALU, d h_prev = sum_g W_hh[g]^T d gh_g by MFMA
// Ownership is the rollout cell's: eight waves per tile of 32 lanes, wave w owns units [16w, 16w+16); accumulator
// register i of lane l is (sample 16 mt + 4 (l >> 4) + i, unit 16 w + (l & 15)) for both MFMA shapes, so the records
// keep their layout ([unit][lane] rows, seq_common.hpp) and the weight-gradient GEMMs read them unchanged.
// Register budget per wave: 3 gates x 4 k-blocks x 3 pieces x 4 = 144 weight registers, 24 accumulators.
// Reference: gru_cell (src/torch/modules/seq/rnn/gru.rs:30-39), Chain (modules/chain.rs:127-186), and what libtorch's
// autograd does for loss.backward() on them (src/torch/optimizers/coptimizer.rs:13-26).
#include "bf16_tile.hpp"
#include "seq_common.hpp"

namespace {

using bt::Frag;

constexpr int HROW = GH + 8;      // halfwords per row of a [sample][unit] piece image: 272-byte rows, so the 16-byte
                                  // operand reads of 16 consecutive samples start 4 banks apart
constexpr int GROW = 3 * GH + 8;  // the backward's [sample][gate unit] rows (784 bytes: the same property)

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

// acc += sum over the nine piece pairs of a (3 fragments of A) and b (3 fragments of B)
__device__ __forceinline__ f32x4 mfma9(const Frag (&a)[3], const Frag (&b)[3], f32x4 acc) {
#pragma unroll
  for (int p = 2; p >= 0; --p)  // small terms first
#pragma unroll
    for (int q = 2; q >= 0; --q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p].v, b[q].v, acc, 0, 0, 0);
  return acc;
}

// ---------------------------------------------------------------- forward recurrence
template <int D>
__global__ void __launch_bounds__(W16 * 64, 2)
    k_gru_recur_fwd(TrajDev tr, const float *__restrict__ params, int A, float *__restrict__ act,
                    const int32_t *__restrict__ skip) {
  __shared__ __attribute__((aligned(16))) unsigned short hP[2][3][TL][HROW];  // h as pieces, [sample][unit], by step parity
  __shared__ float xS[2][TL][8];
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
  for (int q = threadIdx.x; q < (int)(sizeof(hP) / 4); q += W16 * 64) reinterpret_cast<uint32_t *>(hP)[q] = 0u;
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
  for (uint32_t t = 0; t < T; ++t) {
    const int cur = (int)(t & 1), nxt = cur ^ 1;
    if (io_lane && t + 1 < T) fetch(t + 1);  // lands under the products
    f32x4 acc[3][2];
#pragma unroll
    for (int gte = 0; gte < 3; ++gte)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[gte][mt] = (f32x4){bhh[gte], bhh[gte], bhh[gte], bhh[gte]};
    // operand reads one (k-block, M-tile) ahead of the products, and no further (the scheduler would otherwise
    // hoist more reads than the register budget holds)
    auto frags = [&](int it, Frag (&a)[3]) {
      const int kb = it >> 1, mt = it & 1;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[p].x = *reinterpret_cast<const uint4 *>(&hP[cur][p][16 * mt + n16][32 * kb + 8 * g4]);
    };
    Frag fa[2][3];
    frags(0, fa[0]);
#pragma unroll
    for (int it = 0; it < 2 * (GH / 32); ++it) {
      if (it + 1 < 2 * (GH / 32)) frags(it + 1, fa[(it + 1) & 1]);
      Frag wn[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) wn[q].x = wnS[it >> 1][q][wave][lane];
#pragma unroll
      for (int gte = 0; gte < 2; ++gte) acc[gte][it & 1] = mfma9(fa[it & 1], wf[gte][it >> 1], acc[gte][it & 1]);
      acc[2][it & 1] = mfma9(fa[it & 1], wn, acc[2][it & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    float *__restrict__ store = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      // (uniform block base + 32-bit lane offsets: the addresses stay out of the vector registers)
      const uint32_t row = (uint32_t)(j * TL + 16 * mt + 4 * g4);
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_GHN * GH * TL) + row) = acc[2][mt];
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_HPREV * GH * TL) + row) =
          (f32x4){hown[4 * mt], hown[4 * mt + 1], hown[4 * mt + 2], hown[4 * mt + 3]};
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
        const float rr = rl_sigmoidf(acc[0][mt][i] + gi[0]);
        const float zz = rl_sigmoidf(acc[1][mt][i] + gi[1]);
        const float rn = acc[2][mt][i] * rr;
        const float nn = rl_tanhf(gi[2] + rn);
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
        hP[nxt][0][m][j] = (unsigned short)p0;
        hP[nxt][1][m][j] = (unsigned short)p1;
        hP[nxt][2][m][j] = (unsigned short)p2;
      }
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_R * GH * TL) + row) = rv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_Z * GH * TL) + row) = zv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_N * GH * TL) + row) = nv;
      *reinterpret_cast<f32x4 *>(store + (uint32_t)(ACT_A1 * GH * TL) + row) = av;
    }
    if (io_lane && t + 1 < T) publish(nxt);
    __syncthreads();  // one barrier per step: the images of step t + 1 are complete, those of step t are free
  }
}

// ---------------------------------------------------------------- head forward, all (step, tile) blocks in parallel
// u = relu(b1 + W1 relu(h')) on the matrix cores (A operand: the recorded relu(h') block, [unit][lane] rows = the
// [k][m] layout of a 16x16x4 A operand), recorded for the backward; out_a = b2_a + sum_q u_q W2[a][q] as 8 partial
// sums of 16 terms per (sample, output), joined by a shuffle tree.
template <int A>
__global__ void __launch_bounds__(W16 * 64, 2)
    k_seq_head_forward(TrajDev tr, const float *__restrict__ params, int D, int NG, float *__restrict__ act,
                       float *__restrict__ out, uint32_t tiles, uint32_t blocks, const int32_t *__restrict__ skip) {
  __shared__ __attribute__((aligned(16))) float bufA[GH][TLS];
  __shared__ float uS[TL][MH + 1];
  __shared__ float w2S[2][MH];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4, j = 16 * wave + n16;
  const uint32_t N = tr.n, T = tr.T;
  const GruParams g = seq_params(params, D, A, NG);
  float w1[GH / 4];
#pragma unroll
  for (int ks = 0; ks < GH / 4; ++ks) w1[ks] = g.W1[(size_t)j * GH + 4 * ks + g4];
  const float b1 = g.b1[j];
  for (int q = threadIdx.x; q < A * MH; q += W16 * 64) w2S[q / MH][q % MH] = g.W2[q];
  const int hs = 4 * wave + g4, ha = (lane >> 3) & 1, hc = lane & 7;  // head: sample, output, 16-term chunk
  const float b2v = ha < A ? g.b2[ha] : 0.0f;
  const size_t lo = (size_t)j * TL + 4 * g4;
  f32x4 a1n[2];
  auto fetch = [&](uint32_t blk) {
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      a1n[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_A1 * GH * TL + lo + 16 * mt);
  };
  if (blockIdx.x < blocks) fetch(blockIdx.x);
  for (uint32_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    const uint32_t t = blk / tiles, lane0 = (blk % tiles) * TL;
    float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    __syncthreads();  // the previous block's readers of bufA / uS are done
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4 *>(&bufA[j][16 * mt + 4 * g4]) = a1n[mt];
    if (blk + gridDim.x < blocks) fetch(blk + gridDim.x);  // the next block's operand lands under this block's products
    __syncthreads();
    f32x4 acc1[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc1[mt] = (f32x4){b1, b1, b1, b1};
#pragma unroll
    for (int ks = 0; ks < GH / 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bufA[4 * ks + g4][16 * mt + n16], w1[ks], acc1[mt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 uv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float u = acc1[mt][i] > 0.0f ? acc1[mt][i] : 0.0f;
        uS[acc16_row(mt, i, g4)][j] = u;
        uv[i] = u;
      }
      *reinterpret_cast<f32x4 *>(ab + (size_t)ACT_U * GH * TL + lo + 16 * mt) = uv;
    }
    __syncthreads();
    float part = 0.0f;
    if (ha < A) {
#pragma unroll
      for (int q = 0; q < 16; ++q) part = __builtin_fmaf(uS[hs][16 * hc + q], w2S[ha][16 * hc + q], part);
    }
    part = part + __shfl_xor(part, 1, 64);
    part = part + __shfl_xor(part, 2, 64);
    part = part + __shfl_xor(part, 4, 64);
    if (hc == 0 && ha < A) out[((size_t)ha * T + t) * N + lane0 + hs] = part + b2v;
  }
}

// ---------------------------------------------------------------- backward recurrence
// Reads the record of step t and d relu(h') (k_seq_head_backward), carries d h in registers:
//   dh = [episode continues] dh_next + d relu(h')
//   d n = dh (1 - z),  d pre_n = d n (1 - n^2),  d r = d pre_n gh_n,  d pre_r = d r r (1 - r),
//   d pre_z = dh (h_prev - n) z (1 - z),  d gh_n = d pre_n r
//   d h_prev = dh z + W_hh[r]^T d pre_r + W_hh[z]^T d pre_z + W_hh[n]^T d gh_n          (K = 384 on the matrix pipe)
// and writes the four per-step arrays the weight-gradient GEMMs read.
__global__ void __launch_bounds__(W16 * 64, 2)
    k_gru_recur_bwd(TrajDev tr, const float *__restrict__ params, int D, int A, const float *__restrict__ act,
                    float *__restrict__ dpre, const int32_t *__restrict__ skip) {
  constexpr int KB = 3 * GH / 32, KBL = 3;  // k-blocks of the K = 384 product; the last KBL keep their weights in LDS
  __shared__ __attribute__((aligned(16))) unsigned short gP[3][TL][GROW];  // gate gradients as pieces, [sample][gate unit]
  __shared__ uint4 wTS[KBL][3][W16][64];  // 72 KB: 36 registers the budget does not have
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
  struct StepIn {
    f32x4 r[2], z[2], n[2], ghn[2], hp[2], da1[2];
    uint32_t end[2];  // four flag bytes
  };
  const uint32_t lo = (uint32_t)(j * TL + 4 * g4);  // + 16 mt: first of the lane's four contiguous samples (32-bit
                                                    // offsets from a uniform block base: no 64-bit address registers)
  auto load = [&](StepIn &in, uint32_t t) {
    const size_t blk = (size_t)t * tiles + tile;
    const float *__restrict__ ab = act + blk * SEQ_ARR * GH * TL;
    const float *__restrict__ db = dpre + blk * DPRE_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const uint32_t o = lo + 16 * mt;
      in.r[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_R * GH * TL) + o);
      in.z[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_Z * GH * TL) + o);
      in.n[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_N * GH * TL) + o);
      in.ghn[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_GHN * GH * TL) + o);
      in.hp[mt] = *reinterpret_cast<const f32x4 *>(ab + (uint32_t)(ACT_HPREV * GH * TL) + o);
      in.da1[mt] = *reinterpret_cast<const f32x4 *>(db + (uint32_t)(DPRE_DA1 * GH * TL) + o);
      in.end[mt] = *reinterpret_cast<const uint32_t *>(tr.flag + ((size_t)t * N + lane0) + (uint32_t)(16 * mt + 4 * g4));
    }
  };
  StepIn in;
  load(in, T - 1);
  f32x4 dhc[2];
  dhc[0] = dhc[1] = (f32x4){0, 0, 0, 0};
  for (uint32_t t = T; t-- > 0;) {
    float *__restrict__ db = dpre + ((size_t)t * tiles + tile) * DPRE_ARR * GH * TL;
    f32x4 acc[2];  // d h_prev: starts as the direct term dh z, then the K = 384 product is added (two interleaved
                   // chains keep the matrix pipe at full rate)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 grv, gzv, dpnv, gnrv;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * mt + 4 * g4 + i;
        const float rr = in.r[mt][i], zz = in.z[mt][i], nn = in.n[mt][i];
        const bool ended = ((in.end[mt] >> (8 * i)) & 0xffu) != RL_SUCC_CONTINUE;
        const float dh = (ended ? 0.0f : dhc[mt][i]) + in.da1[mt][i];
        const float dzg = dh * (in.hp[mt][i] - nn);
        const float dn = dh * (1.0f - zz);
        const float dpn = dn * (1.0f - nn * nn);
        const float dr = dpn * in.ghn[mt][i];
        grv[i] = dr * rr * (1.0f - rr);
        gzv[i] = dzg * zz * (1.0f - zz);
        gnrv[i] = dpn * rr;
        dpnv[i] = dpn;
        acc[mt][i] = dh * zz;
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
      const uint32_t o = lo + 16 * mt;
      *reinterpret_cast<f32x4 *>(db + (uint32_t)(0 * GH * TL) + o) = grv;
      *reinterpret_cast<f32x4 *>(db + (uint32_t)(1 * GH * TL) + o) = gzv;
      *reinterpret_cast<f32x4 *>(db + (uint32_t)(2 * GH * TL) + o) = dpnv;
      *reinterpret_cast<f32x4 *>(db + (uint32_t)(3 * GH * TL) + o) = gnrv;
    }
    if (t > 0) load(in, t - 1);  // lands under the products below
    __syncthreads();  // the image of step t is complete
    // operand reads one (k-block, M-tile) ahead of the products, and no further (the scheduler would otherwise
    // hoist more reads than the register budget holds)
    auto frags = [&](int it, Frag (&a)[3]) {
      const int kb = it >> 1, mt = it & 1;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[p].x = *reinterpret_cast<const uint4 *>(&gP[p][16 * mt + n16][32 * kb + 8 * g4]);
    };
    Frag fa[2][3];
    frags(0, fa[0]);
#pragma unroll
    for (int it = 0; it < 2 * KB; ++it) {
      if (it + 1 < 2 * KB) frags(it + 1, fa[(it + 1) & 1]);
      if ((it >> 1) < KB - KBL) {
        acc[it & 1] = mfma9(fa[it & 1], wT[(it >> 1) < KB - KBL ? (it >> 1) : 0], acc[it & 1]);
      } else {
        Frag wl[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) wl[q].x = wTS[(it >> 1) - (KB - KBL)][q][wave][lane];
        acc[it & 1] = mfma9(fa[it & 1], wl, acc[it & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dhc[0] = acc[0];
    dhc[1] = acc[1];
    __syncthreads();  // every wave has read the image: step t - 1 may overwrite it
  }
}

}  // namespace

// ---------------------------------------------------------------- launchers
// teacher-forced training forward of the GRU chain: seq.act (all seven arrays) and d_out [A][T][n]
void launch_gru_train_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip) {
  RL_REQUIRE(traj->d.D == 5, "recurrent forward: built for 5 observation features");
  rl_engine *e = traj->eng;
  const uint32_t tiles = traj->d.n / TL, blocks = traj->d.T * tiles;
  const int A = (int)mod->out_dim;
  hipLaunchKernelGGL(k_gru_recur_fwd<5>, dim3(tiles), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, A, d_act,
                     d_skip);
  const uint32_t grid = blocks < 2048 ? blocks : 2048;
  if (A == 2)
    hipLaunchKernelGGL(k_seq_head_forward<2>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, 3,
                       d_act, d_out, tiles, blocks, d_skip);
  else
    hipLaunchKernelGGL(k_seq_head_forward<1>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, 3,
                       d_act, d_out, tiles, blocks, d_skip);
}

// backward recurrence of the GRU chain (after k_seq_head_backward has left d u_pre and d relu(h') in seq.dpre)
void launch_gru_train_recur_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip) {
  const SeqDev &q = traj->seq;
  hipLaunchKernelGGL(k_gru_recur_bwd, dim3(q.tiles), dim3(W16 * 64), 0, traj->eng->stream, traj->d, mod->d_params, 5,
                     (int)mod->out_dim, q.act, q.dpre, d_skip);
}
