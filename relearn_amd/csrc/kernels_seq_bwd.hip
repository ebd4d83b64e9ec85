// kernels_seq_bwd.hip — training passes of the recurrent chains (GRU / LSTM -> ReLU -> MLP): per-sample output
// gradients, backward through time, weight-gradient GEMMs, f64 row reduction.  Workspace layout and tile geometry:
// seq_common.hpp; the forward that fills the activation record: kernels_seq.hip.
#include "seq_common.hpp"

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
    const float d = values[b] - tr.tgt[b];
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
// for one accumulator.  A lane's four samples of an M-tile are contiguous in the record ([half][unit][16] arrays, rec_at), so every
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
  const size_t lo = rec_at(j, 4 * g4);  // + 16 mt: first of the lane's four contiguous samples
  auto load_u = [&](StepIn &in, uint32_t t) {
    const float *__restrict__ ab = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      in.u[mt] = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_U * GH * TL + lo + REC_HALF * mt);
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
      const size_t o = lo + REC_HALF * mt;
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
      *reinterpret_cast<f32x4 *>(db + (size_t)4 * GH * TL + lo + REC_HALF * mt) = duv;
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
      const size_t o = lo + REC_HALF * mt;
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
  const size_t lo = rec_at(j, 4 * g4);
  for (uint32_t blk = blockIdx.x; blk < blocks; blk += gridDim.x) {
    const uint32_t t = blk / tiles, lane0 = (blk % tiles) * TL;
    const float *__restrict__ ab = act + (size_t)blk * SEQ_ARR * GH * TL;
    float *__restrict__ db = dpre + (size_t)blk * DPRE_ARR * GH * TL;
    __syncthreads();  // the previous block's readers of bufU are done
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 uv = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_U * GH * TL + lo + REC_HALF * mt);
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
      *reinterpret_cast<f32x4 *>(db + (size_t)DPRE_DU * GH * TL + lo + REC_HALF * mt) = duv;
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
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(ab + (size_t)ACT_A1 * GH * TL + lo + REC_HALF * mt);
      f32x4 dav;
#pragma unroll
      for (int i = 0; i < 4; ++i) dav[i] = a1[i] > 0.0f ? acc1[mt][0][i] + acc1[mt][1][i] : 0.0f;
      *reinterpret_cast<f32x4 *>(db + (size_t)DPRE_DA1 * GH * TL + lo + REC_HALF * mt) = dav;
    }
  }
}

__global__ void __launch_bounds__(W16 * 64, 2) k_lstm_bptt(TrajDev tr, const float *__restrict__ params, int D, int A,
                                                           const float *__restrict__ act, float *__restrict__ dpre,
                                                           const int32_t *__restrict__ skip) {
  // gate deltas [gate][sample][unit]: the matrix instruction ks of a lane group g4 takes the unit 32 g4 + ks, so four
  // consecutive ks are one 16-byte read.  Unit j sits at column 64 (j / 32) + j % 32 of a 228-float row: the operand reads
  // (lane = sample row, lane group = unit quarter) and the delta writes (lane = unit, lane group = sample) are both
  // conflict-free under the banking rules (scripts/lds_conflicts.py; rows of 132 floats cost the reads twice their cycles)
  constexpr int BG_ROW = 228, BG_Q = 64;
  __shared__ __attribute__((aligned(16))) float bufG[4][TL][BG_ROW];
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
    for (int ks = 0; ks < GH / 4; ++ks) whhT[gte][ks] = g.Whh[(size_t)(gte * GH + 32 * g4 + ks) * GH + j];
  const size_t lo = rec_at(j, 4 * g4);
  f32x4 dhc[2], dcc[2];
  dhc[0] = dhc[1] = dcc[0] = dcc[1] = (f32x4){0, 0, 0, 0};
  // (requesting a step's record one step ahead — 57 more registers across the matrix phase — was measured and is slower:
  // 3.45 against 3.07 ms per gradient at 16,384 x 100)
  for (uint32_t t = T; t-- > 0;) {
    const float *__restrict__ ab = act + ((size_t)t * tiles + tile) * SEQ_ARR * GH * TL;
    float *__restrict__ db = dpre + ((size_t)t * tiles + tile) * DPRE_ARR * GH * TL;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const size_t o = lo + REC_HALF * mt;
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
        const int col = BG_Q * (j >> 5) + (j & 31);
        bufG[0][m][col] = div[i];
        bufG[1][m][col] = dfv[i];
        bufG[2][m][col] = dgv[i];
        bufG[3][m][col] = dov[i];
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
    for (int k4 = 0; k4 < GH / 16; ++k4)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int gte = 0; gte < 4; ++gte) {
          const f32x4 av = *reinterpret_cast<const f32x4 *>(&bufG[gte][16 * mt + n16][BG_Q * g4 + 4 * k4]);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            accg[gte][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], whhT[gte][4 * k4 + q], accg[gte][mt], 0, 0, 0);
        }
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
// consecutive floats of a [unit][16] row of a record half); dW_ih, the biases and dW2 on the VALU.  A workgroup accumulates a
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
        // (record layout [half][unit][16], seq_common.hpp: float 4 f is unit (f & 511) >> 2, sample 16 (f >> 9) + 4 (f & 3))
        *reinterpret_cast<f32x4 *>(&aS[a][(f & 511) >> 2][16 * (f >> 9) + 4 * (f & 3)]) = stg[a][i];
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
      const size_t o = rec_at(j, 16 * hf + 4 * q);
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
// round-robin to 16 thread groups (eight loads in flight each), the 16 partial sums added in group order.  The columns
// may come from different producers: segment s covers columns [col0, col1) with `rows` rows of P floats at `base`.
struct SeqReduceSegs {
  static constexpr int MAX = 8;
  int n;
  uint32_t col0[MAX], col1[MAX], rows[MAX];
  const float *base[MAX];
};
__global__ void __launch_bounds__(1024) k_seq_reduce(SeqReduceSegs segs, uint32_t P, float *__restrict__ vec) {
  __shared__ double part[16][64];
  const uint32_t c = threadIdx.x & 63, grp = threadIdx.x >> 6, p = blockIdx.x * 64 + c;
  const float *__restrict__ slab = nullptr;
  uint32_t rows = 0;
#pragma unroll
  for (int k = 0; k < SeqReduceSegs::MAX; ++k)
    if (k < segs.n && p >= segs.col0[k] && p < segs.col1[k]) {
      slab = segs.base[k];
      rows = segs.rows[k];
    }
  double s = 0.0;
  const bool live = slab != nullptr;
  if (live) {
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
  if (grp == 0 && live) {
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
  if (mod->lane_kernels()) return launch_stack_backward(traj, mod, d_skip);
  rl_engine *e = traj->eng;
  const SeqDev &q = traj->seq;
  uint32_t P = (uint32_t)mod->P, blocks = traj->d.T * q.tiles;
  const bool lstm = mod->kind == RL_MODULE_LSTM_MLP;
  const bool gru_head = e->kernel_variant != 1;  // kernels_seq_train.hip: the head's backward (with its weight gradients)
                                                // and the recurrent weight gradients on the bf16 pipe, either cell
  {
    ProfScope ps(e, RL_K_BACKWARD);
    const bool split_head = lstm || e->kernel_variant != 1;  // the head's backward as its own block-parallel kernel
    if (gru_head) {
      launch_gru_train_head_backward(traj, mod, q.wg_slab + (size_t)q.chunks * P, d_skip);  // + the head's weight gradients
    } else if (split_head) {
      const uint32_t grid = blocks < 2048 ? blocks : 2048;
      const int NG = lstm ? 4 : 3;
      if (mod->out_dim == 2)
        hipLaunchKernelGGL(k_seq_head_backward<2>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, NG,
                           traj->dz, q.act, q.dpre, q.tiles, blocks, d_skip);
      else
        hipLaunchKernelGGL(k_seq_head_backward<1>, dim3(grid), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5, NG,
                           traj->dz, q.act, q.dpre, q.tiles, blocks, d_skip);
    }
    if (lstm) {
      hipLaunchKernelGGL(k_lstm_bptt, dim3(q.tiles), dim3(W16 * 64), 0, e->stream, traj->d, mod->d_params, 5,
                         (int)mod->out_dim, q.act, q.dpre, d_skip);
    } else if (split_head) {
      launch_gru_train_recur_backward(traj, mod, d_skip);  // kernels_seq_train.hip: the recurrence on the bf16 pipe
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
      if (gru_head) {
        launch_gru_train_wgrad(traj, mod, d_skip);
      } else if (lstm) {
        WG(2, 4, 0, 2, true);
        WG(2, 4, 2, 2, false);
      } else {
        WG(2, 3, 0, 3, true);
      }
    } else {
      if (gru_head) {
        launch_gru_train_wgrad(traj, mod, d_skip);
      } else if (lstm) {
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
    SeqReduceSegs segs{};
    auto seg = [&](uint32_t c0, uint32_t c1, const float *base, uint32_t rows) {
      segs.col0[segs.n] = c0;
      segs.col1[segs.n] = c1;
      segs.base[segs.n] = base;
      segs.rows[segs.n] = rows;
      segs.n += 1;
    };
    if (gru_head && lstm) {
      // every recurrent column from the weight-gradient kernel's rows, the head from the head kernel's
      const uint32_t oW1 = 4 * GH * 5 + 4 * GH * GH + 8 * GH;
      seg(0, oW1, q.wg_slab, q.chunks);
      seg(oW1, P, q.wg_slab + (size_t)q.chunks * P, RL_SEQ_HEAD_ROWS);
    } else if (gru_head) {
      // kernels_seq_train.hip: W_hh, the r / z rows of W_ih and b_ih, and b_hh from the weight-gradient kernel's rows;
      // the n rows of W_ih and b_ih from the backward recurrence's rows (one per tile); the head from the head kernel's
      const uint32_t oWhh = 3 * GH * 5, obih = oWhh + 3 * GH * GH, obhh = obih + 3 * GH, oW1 = obhh + 3 * GH;
      const float *extra = q.wg_slab + (size_t)q.chunks * P;
      seg(0, 2 * GH * 5, q.wg_slab, q.chunks);
      seg(2 * GH * 5, oWhh, extra, q.tiles);
      seg(oWhh, obih + 2 * GH, q.wg_slab, q.chunks);
      seg(obih + 2 * GH, obhh, extra, q.tiles);
      seg(obhh, oW1, q.wg_slab, q.chunks);
      seg(oW1, P, extra, RL_SEQ_HEAD_ROWS);
    } else {
      seg(0, P, q.wg_slab, q.chunks);
    }
    hipLaunchKernelGGL(k_seq_reduce, dim3(cdiv_s(P, 64)), dim3(1024), 0, e->stream, segs, P, traj->vec);
  }
}
