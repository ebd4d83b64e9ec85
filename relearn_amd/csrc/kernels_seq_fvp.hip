// kernels_seq_fvp.hip — Fisher-vector products through time for the trust-region update over the recurrent chains:
// forward-mode tangent kernels (static + recurrent parts) for the GRU and the LSTM cell.
#include "seq_common.hpp"

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
      const float a = ab[(size_t)ACT_HPREV * GH * TL + rec_at(2 * ks + hf, n)];
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
      const size_t o = rec_at(j, m);
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
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[(size_t)ACT_A1 * GH * TL + rec_at(2 * ks + hf, n)], w.w1[ks],
                                                  acc1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) sb[(size_t)4 * GH * TL + rec_at(j, acc_row(r, hf))] = acc1[r];
    if (wave == 0 && hf < A) {
      float z = vb2;
#pragma unroll 8
      for (int q = 0; q < MH; ++q) z = __builtin_fmaf(ab[(size_t)ACT_U * GH * TL + rec_at(q, n)], v2S[hf][q], z);
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
      const size_t o = rec_at(j, m);
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
      const size_t o = rec_at(j, m);
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
  // the MLP layer's tangent operands, [k-step][thread] (read back by the thread that wrote them): the four gates' 256
  // operand registers leave no room for these 64 next to the accumulators — the kernel spilt 45 registers with them
  __shared__ float w1S[GH / 2][256];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, hf = lane >> 5, j = 32 * wave + n;
  const uint32_t N = tr.n, T = tr.T;
  const size_t plane = (size_t)(T + 1) * N;
  const GruParams v = seq_params(tangent, D, A, 4);
  float whh[4][GH / 2], wih[4][D], bih[4], bhh[4];
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
  for (int ks = 0; ks < GH / 2; ++ks) w1S[ks][threadIdx.x] = v.W1[(size_t)j * GH + 2 * ks + hf];
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
      const float a = ab[(size_t)ACT_HPREV * GH * TL + rec_at(2 * ks + hf, n)];
#pragma unroll
      for (int gte = 0; gte < 4; ++gte)
        acc[gte] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, whh[gte][ks], acc[gte], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = acc_row(r, hf);
      const size_t o = rec_at(j, m);
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
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[(size_t)ACT_A1 * GH * TL + rec_at(2 * ks + hf, n)], w1S[ks][threadIdx.x], acc1,
                                                  0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) sb[(size_t)4 * GH * TL + rec_at(j, acc_row(r, hf))] = acc1[r];
    if (wave == 0 && hf < A) {
      float z = vb2;
#pragma unroll 8
      for (int q = 0; q < MH; ++q) z = __builtin_fmaf(ab[(size_t)ACT_U * GH * TL + rec_at(q, n)], v2S[hf][q], z);
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
      const size_t o = rec_at(j, m);
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
      const size_t o = rec_at(j, m);
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
  if (mod->lane_kernels()) launch_stack_tangent(traj, mod, d_tangent, d_skip);  // -> seq.out
  else {
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
