// kernels_general.hip — MLPs of any `MlpConfig.hidden_sizes` (src/torch/modules/ff/mlp.rs:13-34): no hidden layer,
// several, or one wider than 128 units.  The fused kernels of this library (kernels_critic.hip, kernels_mfma.hip,
// kernels_rollout.hip) are built for ONE hidden layer of at most 128 units — every BASELINE configuration; any other
// shape takes this path: per-layer kernels over [unit][sample] activation planes in a workspace attached to the
// trajectory.  It is the general path, not the fast one (f32 VALU products, one launch per layer).
//   k_gen_dense    Y = act(b + W X) for 64 samples per workgroup (X tile in LDS, weights through wave-uniform loads,
//                  four output units per pass); with a tangent: tY = act'(.) (W tX + V X + vb)   (forward-mode, for
//                  the Fisher-vector product of TRPO, conjugate_gradient.rs:262-339)
//   k_gen_delta    dX = act'(.) W^T dY                                   (backward through a hidden layer)
//   act is MlpConfig::activation for the hidden layers and ::output_activation for the last one (ff/mlp.rs:139-151,
//   ff/activation.rs:85-92: Identity, Relu, Sigmoid, Tanh); the derivative is taken from the stored OUTPUT of the
//   layer (relu: [y > 0], sigmoid: y (1 - y), tanh: 1 - y^2), so no pre-activation is kept
//   k_gen_wgrad    dW = dY X^T, db = sum dY over a chunk of samples, f64 partials into the slab rows k_reduce sums
//   k_gen_*_terms  the per-sample loss terms and d loss / d output from the stored outputs (the arithmetic of
//                  k_policy_pass / k_critic_fwd, kernels_update.hip)
// Every dot product is `acc = bias; acc = fma(x_k, w_k, acc)`, k ascending (the convention of this library's first
// layers); sums over samples are f32 within 64 samples and f64 above, like the fused kernels.
// The launchers at the end plug into the module-generic seams: launch_policy_pass / launch_critic_fwd /
// launch_mlp_backward (kernels_update.hip), values for GAE and TD targets, row-wise forward, and a step-by-step
// rollout over the standalone env kernels.
#include <mutex>
#include <set>

#include "abi_internal.hpp"
#include "policy_terms.hpp"

namespace {

constexpr int GT = 64;  // samples per workgroup tile

static inline uint32_t cdiv_g(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

// Activation::forward (ff/activation.rs:85-92) and its derivative in terms of the output y
__device__ __forceinline__ float act_apply(int act, float x) {
  if (act == RL_ACT_RELU) return x > 0.0f ? x : 0.0f;
  if (act == RL_ACT_SIGMOID) return rl_sigmoidf(x);
  if (act == RL_ACT_TANH) return rl_tanhf(x);
  return x;
}
__device__ __forceinline__ float act_slope(int act, float y) {
  if (act == RL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (act == RL_ACT_SIGMOID) return y * (1.0f - y);
  if (act == RL_ACT_TANH) return 1.0f - y * y;
  return 1.0f;
}

struct DenseArgs {
  const float *X, *tX;    // [K][.] inputs (rows `xs` / `txs` floats apart), tangent inputs (TANGENT)
  size_t xs, txs;
  const float *W, *b;     // [N][K], [N]
  const float *V, *vb;    // tangent parameters (TANGENT)
  float *Y, *tY;          // [N][.] outputs (rows `ys` floats apart; tY rows too)
  size_t ys;
  size_t S;               // samples
  int K, N;
  int act;                // rl_activation of this layer (wave-uniform)
  const int32_t *skip;
};

template <bool TANGENT>
__global__ void __launch_bounds__(256) k_gen_dense(DenseArgs a) {
  extern __shared__ float sm[];  // Xs[K][GT] (+ tXs[K][GT])
  if (a.skip != nullptr && *a.skip != 0) return;
  const int s = threadIdx.x & (GT - 1);
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave: its units share scalar weight loads
  const size_t s0 = (size_t)blockIdx.x * GT;
  float *Xs = sm, *tXs = sm + (size_t)a.K * GT;
  for (int idx = threadIdx.x; idx < a.K * GT; idx += 256) {
    const int k = idx >> 6, ss = idx & (GT - 1);
    const bool in = s0 + ss < a.S;
    Xs[idx] = in ? a.X[(size_t)k * a.xs + s0 + ss] : 0.0f;
    if (TANGENT) tXs[idx] = (in && a.tX != nullptr) ? a.tX[(size_t)k * a.txs + s0 + ss] : 0.0f;
  }
  __syncthreads();
  const bool live = s0 + s < a.S;
  for (int n0 = 4 * g; n0 < a.N; n0 += 16) {
    float acc[4], tacc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + u < a.N ? n0 + u : a.N - 1;
      acc[u] = a.b[n];
      tacc[u] = TANGENT ? a.vb[n] : 0.0f;
    }
    // eight consecutive k at a time: a unit's eight weights are contiguous (one wide wave-uniform load each), the sums
    // stay k-ascending fma chains
    int k = 0;
    for (; k + 8 <= a.K; k += 8) {
      float w[4][8], v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + u < a.N ? n0 + u : a.N - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          w[u][q] = a.W[(size_t)n * a.K + k + q];
          v[u][q] = TANGENT ? a.V[(size_t)n * a.K + k + q] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float xv = Xs[(k + q) * GT + s];
        const float tv = TANGENT ? tXs[(k + q) * GT + s] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u] = __builtin_fmaf(xv, w[u][q], acc[u]);
          if (TANGENT) {
            tacc[u] = __builtin_fmaf(tv, w[u][q], tacc[u]);
            tacc[u] = __builtin_fmaf(xv, v[u][q], tacc[u]);
          }
        }
      }
    }
    for (; k < a.K; ++k) {
      const float xv = Xs[k * GT + s];
      const float tv = TANGENT ? tXs[k * GT + s] : 0.0f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + u < a.N ? n0 + u : a.N - 1;
        acc[u] = __builtin_fmaf(xv, a.W[(size_t)n * a.K + k], acc[u]);
        if (TANGENT) {
          tacc[u] = __builtin_fmaf(tv, a.W[(size_t)n * a.K + k], tacc[u]);
          tacc[u] = __builtin_fmaf(xv, a.V[(size_t)n * a.K + k], tacc[u]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (n0 + u < a.N) {
          const float y = act_apply(a.act, acc[u]);
          a.Y[(size_t)(n0 + u) * a.ys + s0 + s] = y;
          if (TANGENT) a.tY[(size_t)(n0 + u) * a.ys + s0 + s] = act_slope(a.act, y) * tacc[u];
        }
    }
  }
}

// dX[k][s] = act'(X[k][s]) sum_n W[n][k] dY[n][s]   (X: the stored outputs of the layer below, `act` its activation)
__global__ void __launch_bounds__(256) k_gen_delta(const float *__restrict__ dY, size_t dys, int N,
                                                   const float *__restrict__ W, int K, const float *__restrict__ X,
                                                   float *__restrict__ dX, size_t S, int act,
                                                   const int32_t *__restrict__ skip) {
  extern __shared__ float sm[];  // dYs[N][GT]
  if (skip != nullptr && *skip != 0) return;
  const int s = threadIdx.x & (GT - 1);
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t s0 = (size_t)blockIdx.x * GT;
  for (int idx = threadIdx.x; idx < N * GT; idx += 256) {
    const int n = idx >> 6, ss = idx & (GT - 1);
    sm[idx] = s0 + ss < S ? dY[(size_t)n * dys + s0 + ss] : 0.0f;
  }
  __syncthreads();
  if (s0 + s >= S) return;
  for (int k0 = 4 * g; k0 < K; k0 += 16) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int n = 0;
    for (; n + 4 <= N; n += 4) {  // (W[n][k0 .. k0 + 3] are contiguous: one wide wave-uniform load per row)
      float w[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) w[q][u] = W[(size_t)(n + q) * K + (k0 + u < K ? k0 + u : K - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dv = sm[(n + q) * GT + s];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(dv, w[q][u], acc[u]);
      }
    }
    for (; n < N; ++n) {
      const float dv = sm[n * GT + s];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u < K ? k0 + u : K - 1;
        acc[u] = __builtin_fmaf(dv, W[(size_t)n * K + k], acc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u < K) {
        const size_t o = (size_t)(k0 + u) * S + s0 + s;
        dX[o] = act_slope(act, X[o]) * acc[u];
      }
  }
}

// dW[n][k] = sum_s dY[n][s] X[k][s], db[n] = sum_s dY[n][s] over the samples [chunk * c, chunk * (c + 1)) of slab row c:
// a workgroup owns a 64 x 64 tile of (n, k), a thread a 4 x 4 block of it (8 LDS reads per 16 fma); f32 within a
// 32-sample tile, f64 over the chunk
constexpr int WT = 32;  // samples per staged tile
__global__ void __launch_bounds__(256) k_gen_wgrad(const float *__restrict__ dY, size_t dys, int N,
                                                   const float *__restrict__ X, size_t xs, int K, size_t S,
                                                   uint32_t chunk, double *__restrict__ slab, uint32_t P,
                                                   uint32_t offW, uint32_t offB, const int32_t *__restrict__ skip) {
  __shared__ float dYs[WT][64 + 4], Xs[WT][64 + 4];  // [sample][row]: a thread's four rows are one 16-byte read
  if (skip != nullptr && *skip != 0) return;
  const int tiles_k = (K + 63) / 64;
  const int tn0 = 64 * (int)(blockIdx.x / tiles_k), tk0 = 64 * (int)(blockIdx.x % tiles_k);
  const int tn = 4 * (threadIdx.x >> 4), tk = 4 * (threadIdx.x & 15);
  const size_t c0 = (size_t)blockIdx.y * chunk, c1 = c0 + chunk < S ? c0 + chunk : S;
  double accw[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) accw[i][j] = 0.0;
  }
  for (size_t s0 = c0; s0 < c1; s0 += WT) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * WT; idx += 256) {  // coalesced over the samples of a row
      const int r = idx / WT, ss = idx % WT;
      const bool in = s0 + ss < c1;
      dYs[ss][r] = (in && tn0 + r < N) ? dY[(size_t)(tn0 + r) * dys + s0 + ss] : 0.0f;
      Xs[ss][r] = (in && tk0 + r < K) ? X[(size_t)(tk0 + r) * xs + s0 + ss] : 0.0f;
    }
    __syncthreads();
    float w[4][4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      b[i] = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[i][j] = 0.0f;
    }
#pragma unroll 4
    for (int ss = 0; ss < WT; ++ss) {
      float d[4], x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[i] = dYs[ss][tn + i];
        x[i] = Xs[ss][tk + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b[i] = b[i] + d[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[i][j] = __builtin_fmaf(d[i], x[j], w[i][j]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      accb[i] += (double)b[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) accw[i][j] += (double)w[i][j];
    }
  }
  double *__restrict__ row = slab + (size_t)blockIdx.y * P;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = tn0 + tn + i;
    if (n >= N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (tk0 + tk + j < K) row[offW + (size_t)n * K + tk0 + tk + j] = accw[i][j];
    if (offB != 0xFFFFFFFFu && tk0 == 0 && tk == 0) row[offB + n] = accb[i];  // (no column: a layer without a bias)
  }
}

// per-sample policy terms from stored logits (and tangent logits): the arithmetic of k_policy_pass
template <int MODE>
__global__ void __launch_bounds__(256) k_gen_policy_terms(TrajDev tr, const float *__restrict__ z,
                                                          const float *__restrict__ tz, float *__restrict__ lp0,
                                                          float *__restrict__ dz, double *__restrict__ slabB,
                                                          float inv_B, const int32_t *__restrict__ skip, float clip_lo,
                                                          float clip_hi) {
  __shared__ double red[256];
  if (skip != nullptr && *skip != 0) return;
  const size_t B = (size_t)tr.T * tr.n;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    const float zz[2] = {z[b], z[B + b]};
    const float tt[2] = {MODE == PASS_JVP ? tz[b] : 0.0f, MODE == PASS_JVP ? tz[B + b] : 0.0f};
    policy_sample_terms<MODE>(zz, tt, (int)tr.action[b], MODE == PASS_JVP ? 0.0f : tr.adv[b], b, B, lp0, dz, inv_B, clip_lo,
                              clip_hi, s0, s1, s2);
  }
  if (MODE != PASS_JVP) {
    double t0 = block_sum<256>(s0, red);
    double t1 = block_sum<256>(s1, red);
    double t2 = block_sum<256>(s2, red);
    if (threadIdx.x == 0) {
      slabB[blockIdx.x * 4 + 0] = t0;
      slabB[blockIdx.x * 4 + 1] = t1;
      slabB[blockIdx.x * 4 + 2] = t2;
      slabB[blockIdx.x * 4 + 3] = 0.0;
    }
  }
}

// critic: d = V - target, dz = 2 d / B, loss partial = sum d^2 (k_critic_fwd)
__global__ void __launch_bounds__(256) k_gen_critic_terms(TrajDev tr, const float *__restrict__ v,
                                                          float *__restrict__ dz, double *__restrict__ slabB,
                                                          float two_over_B) {
  __shared__ double red[256];
  const size_t B = (size_t)tr.T * tr.n;
  double s0 = 0.0;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    const float d = v[b] - tr.tgt[b];
    dz[b] = d * two_over_B;
    s0 += (double)(d * d);
  }
  double t0 = block_sum<256>(s0, red);
  if (threadIdx.x == 0) {
    slabB[blockIdx.x * 4 + 0] = t0;
    slabB[blockIdx.x * 4 + 1] = 0.0;
    slabB[blockIdx.x * 4 + 2] = 0.0;
    slabB[blockIdx.x * 4 + 3] = 0.0;
  }
}

// successor values of cut episodes for k_seq_gae / the TD targets: V(term_obs) after an Interrupt, V(obs[T]) for a lane
// the horizon cuts, 0 elsewhere (eval_extended_state_values, critics/mod.rs:116-131)
__global__ void __launch_bounds__(256) k_gen_successor_values(TrajDev tr, const float *__restrict__ v_term,
                                                              const float *__restrict__ v_last,
                                                              float *__restrict__ succ) {
  const size_t B = (size_t)tr.T * tr.n;
  const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const uint32_t t = (uint32_t)(b / tr.n), i = (uint32_t)(b % tr.n);
  const int f = tr.flag[b];
  float s = 0.0f;
  if (f == RL_SUCC_INTERRUPT) s = v_term[b];
  else if (f == RL_SUCC_CONTINUE && t == tr.T - 1) s = v_last[i];
  succ[b] = s;
}

// *flag = 0 when any step of the trajectory ended in an Interrupt (the host sets it non-zero before the launch)
__global__ void __launch_bounds__(256) k_gen_any_interrupt(TrajDev tr, int32_t *__restrict__ flag) {
  const size_t B = (size_t)tr.T * tr.n;
  bool any = false;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) any |= tr.flag[b] == RL_SUCC_INTERRUPT;
  if (__any(any) && (threadIdx.x & 63) == 0) *flag = 0;
}

// PolicyActor::act for one step of every lane from stored logits: Categorical::new + the inverse-CDF draw with word
// `word` of the lane's actor stream (the fused rollouts' arithmetic, kernels_rollout.hip)
__global__ void __launch_bounds__(256) k_gen_sample_actions(CartPoleDev c, const float *__restrict__ z, uint32_t n,
                                                            uint64_t word, uint8_t *__restrict__ actions) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
  rl_chacha_block(c.key_actor, word >> 4, c.lane_offset + i, 4, w);
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k == (int)(word & 15)) v = w[k];
  const float u = rl_u32_to_unit_f32(v);
  const float zz[2] = {z[i], z[n + i]};
  float lp[2];
  log_softmax_lane<2>(zz, lp);
  actions[i] = (uint8_t)categorical_sample_lane<2>(lp, u);
}

// one step's record: the observation the step started from, then what the env's step kernel left in its buffers
__global__ void __launch_bounds__(256) k_gen_record_obs(TrajDev tr, const float *__restrict__ obs, uint32_t t) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= tr.n) return;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  for (uint32_t d = 0; d < tr.D; ++d) tr.obs[d * plane + (size_t)t * tr.n + i] = obs[(size_t)d * tr.n + i];
}
__global__ void __launch_bounds__(256) k_gen_record_step(TrajDev tr, const uint8_t *__restrict__ actions,
                                                         const float *__restrict__ reward,
                                                         const uint8_t *__restrict__ flag,
                                                         const float *__restrict__ term_obs, uint32_t t) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= tr.n) return;
  const size_t o = (size_t)t * tr.n + i;
  tr.action[o] = actions[i];
  tr.reward[o] = reward[i];
  tr.flag[o] = flag[i];
  if (flag[i] == RL_SUCC_INTERRUPT)
    for (uint32_t d = 0; d < tr.D; ++d) tr.term_obs[(size_t)d * tr.T * tr.n + o] = term_obs[(size_t)d * tr.n + i];
}

// One CartPole step of every lane in ONE launch: PolicyActor::act from the stored logits (k_gen_sample_actions), the env
// step (k_env_step, kernels_rollout.hip), the step's record and the next observation — into the trajectory (slot t + 1)
// and into the env's observation buffer, where the next forward reads it.  The same values as the four launches it
// replaces (sample, env step, record step, record observation): a per-step sequence is launch-bound, 7 launches of ~7 us.
template <int D>
__global__ void __launch_bounds__(256) k_gen_step_cartpole(CartPoleDev c, EnvStateDev st, TrajDev tr,
                                                           const float *__restrict__ z, uint64_t word, uint32_t t,
                                                           float *__restrict__ obs_next) {
  const uint32_t n = tr.n, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
  rl_chacha_block(c.key_actor, word >> 4, c.lane_offset + i, 4, w);
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k == (int)(word & 15)) v = w[k];
  const float u = rl_u32_to_unit_f32(v);
  const float zz[2] = {z[i], z[n + i]};
  float lp[2];
  log_softmax_lane<2>(zz, lp);
  const int a = categorical_sample_lane<2>(lp, u);
  LaneState s;
  lane_load(st, i, s);
  const int succ = cp_step(c, s, a);
  const size_t o = (size_t)t * n + i, plane = (size_t)(tr.T + 1) * n;
  tr.action[o] = (uint8_t)a;
  tr.reward[o] = 1.0f;  // Reward(1.0) as f32
  tr.flag[o] = (uint8_t)succ;
  float f[D];
  if (succ == RL_SUCC_INTERRUPT) {
    cp_features<D>(c, s, f);
#pragma unroll
    for (int d = 0; d < D; ++d) tr.term_obs[(size_t)d * tr.T * n + o] = f[d];
  }
  if (succ != RL_SUCC_CONTINUE) cp_reset(c, s, c.lane_offset + i);
  cp_features<D>(c, s, f);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    obs_next[(size_t)d * n + i] = f[d];
    tr.obs[d * plane + (size_t)(t + 1) * n + i] = f[d];
  }
  lane_store(st, i, s);
}

// d loss / d (pre-activation of the output layer) = d loss / d output * act'(output)   (output_activation != Identity)
__global__ void __launch_bounds__(256) k_gen_output_slope(float *__restrict__ dz, const float *__restrict__ z, size_t count,
                                                          int act, const int32_t *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) dz[i] = dz[i] * act_slope(act, z[i]);
}

// ---------------------------------------------------------------- all layers of a forward in one workgroup
// The arithmetic of k_gen_dense (acc = bias; acc = fma(x_k, w_k, acc), k ascending; act), layer after layer on one
// 64-sample tile whose activations ping-pong between two LDS tiles: no activation plane in HBM and one launch for the
// whole network.  A workgroup's four waves split a layer's output units, four units per pass over the inputs, eight
// inputs per block of wave-uniform weight loads — k_gen_dense's inner loop.
struct ChainNet {
  const float *params;
  int n_layers, act, out_act;
  int K[RL_MLP_MAX_HIDDEN + 1], N[RL_MLP_MAX_HIDDEN + 1];
  uint32_t off[RL_MLP_MAX_HIDDEN + 1];
  uint32_t boff[RL_MLP_MAX_HIDDEN + 1];  // the layer's bias vector (a bias-less module: zeros behind its parameters)
  int wmax;  // widest input of any layer (rows of an LDS tile)
};

// X: [wmax][GT] input tile (rows 0 .. K[0]); returns the tile that holds the module's outputs (rows 0 .. out_dim).
// Every thread of the workgroup (`waves` waves, wave g takes the unit groups g, g + waves, ...) calls it; barriers inside.
__device__ __forceinline__ float *chain_forward_tile(const ChainNet &net, float *X, float *Y, int s, int g, int waves = 4) {
  float *in = X, *out = Y;
  for (int l = 0; l < net.n_layers; ++l) {
    const int K = net.K[l], N = net.N[l];
    const float *__restrict__ W = net.params + net.off[l];
    const float *__restrict__ b = net.params + net.boff[l];
    const int act = l + 1 == net.n_layers ? net.out_act : net.act;
    for (int n0 = 4 * g; n0 < N; n0 += 4 * waves) {
      float acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = b[n0 + u < N ? n0 + u : N - 1];
      int k = 0;
      for (; k + 8 <= K; k += 8) {
        float w[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + u < N ? n0 + u : N - 1;
#pragma unroll
          for (int q = 0; q < 8; ++q) w[u][q] = W[(size_t)n * K + k + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float xv = in[(k + q) * GT + s];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(xv, w[u][q], acc[u]);
        }
      }
      for (; k < K; ++k) {
        const float xv = in[k * GT + s];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(xv, W[(size_t)(n0 + u < N ? n0 + u : N - 1) * K + k], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (n0 + u < N) out[(n0 + u) * GT + s] = act_apply(act, acc[u]);
    }
    __syncthreads();
    float *t = in;
    in = out;
    out = t;
  }
  return in;
}

// rows of [in_dim][.] inputs (feature k of row r at X[k * xs + r]) -> outputs Y[o * ys + r]
__global__ void __launch_bounds__(256) k_gen_chain_rows(ChainNet net, const float *__restrict__ X, size_t xs, size_t rows,
                                                        float *__restrict__ Y, size_t ys, int out_dim,
                                                        const int32_t *__restrict__ skip) {
  extern __shared__ float chain_lds[];  // [2][wmax][GT]
  if (skip != nullptr && *skip != 0) return;
  const int s = threadIdx.x & (GT - 1);
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t s0 = (size_t)blockIdx.x * GT;
  float *t0 = chain_lds, *t1 = chain_lds + (size_t)net.wmax * GT;
  for (int idx = threadIdx.x; idx < net.K[0] * GT; idx += 256) {
    const int k = idx >> 6, ss = idx & (GT - 1);
    t0[idx] = s0 + ss < rows ? X[(size_t)k * xs + s0 + ss] : 0.0f;
  }
  __syncthreads();
  const float *z = chain_forward_tile(net, t0, t1, s, g);
  for (int idx = threadIdx.x; idx < out_dim * GT; idx += 256) {
    const int o = idx >> 6, ss = idx & (GT - 1);
    if (s0 + ss < rows) Y[(size_t)o * ys + s0 + ss] = z[o * GT + ss];
  }
}

// The fused rollout of kernels_rollout.hip (k_rollout_cartpole: T env-actor steps in one launch, lane state in registers
// for the whole horizon, the 26 B/step trajectory record the only HBM traffic) for policies with any hidden_sizes: a
// workgroup owns 64 lanes, its four waves share the policy forward of every step (chain_forward_tile), wave 0 keeps the
// lanes: features, the actor's draw, the env step, the records.  The same observations, draws, actions and records as the
// step-by-step launch sequence (tests/test_gpu_general_mlp.py replays them through the oracle's lanes bit for bit).
// (sixteen waves per workgroup: the launch is T x the latency of one step, and a step's latency is the number of
// weight blocks a wave walks through — one pass per layer at sixteen waves, four at four: 2.2 -> ? ms at 16,384 lanes)
constexpr int ROLL_WAVES = 16;
template <int D>
__global__ void __launch_bounds__(ROLL_WAVES * 64) k_gen_rollout_cartpole(CartPoleDev c, EnvStateDev st, TrajDev tr,
                                                                          ChainNet net, uint64_t t_global) {
  extern __shared__ float chain_lds[];  // [2][wmax][GT], then the actor words [16][GT]
  const int s = threadIdx.x & (GT - 1);
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool keeper = g == 0;
  float *t0 = chain_lds, *t1 = chain_lds + (size_t)net.wmax * GT;
  uint32_t *actor_words = reinterpret_cast<uint32_t *>(chain_lds + (size_t)2 * net.wmax * GT);
  const uint32_t n = tr.n, T = tr.T;
  const uint32_t i = blockIdx.x * GT + (uint32_t)s;
  const bool live = keeper && i < n;
  const uint32_t il = i < n ? i : n - 1;
  const uint64_t lane = c.lane_offset + il;
  LaneState ls;
  if (keeper) lane_load(st, il, ls);
  const size_t plane = (size_t)(T + 1) * n;
  uint64_t cur_block = ~0ull;
  for (uint32_t t = 0; t < T; ++t) {
    float f[D];
    if (keeper) {
      cp_features<D>(c, ls, f);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (live) tr.obs[d * plane + (size_t)t * n + il] = f[d];
        t0[d * GT + s] = f[d];
      }
      const uint64_t blk = (t_global + t) >> 4;
      if (blk != cur_block) {
        uint32_t words[16];
        rl_chacha_block(c.key_actor, blk, lane, 4, words);
#pragma unroll
        for (int k = 0; k < 16; ++k) actor_words[k * GT + s] = words[k];
        cur_block = blk;
      }
    }
    __syncthreads();
    const float *z = chain_forward_tile(net, t0, t1, s, g, ROLL_WAVES);  // (ends with a barrier)
    if (keeper) {
      const float u = rl_u32_to_unit_f32(actor_words[(uint32_t)((t_global + t) & 15) * GT + s]);
      const float zz[2] = {z[s], z[GT + s]};
      float lp[2];
      log_softmax_lane<2>(zz, lp);
      const int a = categorical_sample_lane<2>(lp, u);
      const int succ = cp_step(c, ls, a);
      const size_t o = (size_t)t * n + il;
      if (live) {
        tr.action[o] = (uint8_t)a;
        tr.reward[o] = 1.0f;
        tr.flag[o] = (uint8_t)succ;
      }
      if (succ == RL_SUCC_INTERRUPT) {
        cp_features<D>(c, ls, f);
        if (live) {
#pragma unroll
          for (int d = 0; d < D; ++d) tr.term_obs[(size_t)d * T * n + o] = f[d];
        }
      }
      if (succ != RL_SUCC_CONTINUE) cp_reset(c, ls, lane);
    }
    __syncthreads();  // the output tile is the next step's scratch
  }
  if (keeper) {
    float f[D];
    cp_features<D>(c, ls, f);
    if (live) {
#pragma unroll
      for (int d = 0; d < D; ++d) tr.obs[d * plane + (size_t)T * n + il] = f[d];
      lane_store(st, il, ls);
    }
  }
}

// the module as a kernel argument; false when a layer is too wide for two LDS tiles (+ the actor words) of a workgroup
bool chain_net(const rl_mlp *m, ChainNet *net) {
  net->params = m->d_params;
  net->n_layers = (int)m->n_layers();
  net->act = m->act;
  net->out_act = m->out_act;
  int wmax = (int)m->in_dim;
  for (uint32_t l = 0; l < m->n_layers(); ++l) {
    net->K[l] = (int)m->fan_in(l);
    net->N[l] = (int)m->fan_out(l);
    net->off[l] = (uint32_t)m->layer_offset(l);
    net->boff[l] = (uint32_t)m->bias_offset(l);
    if (net->N[l] > wmax) wmax = net->N[l];
  }
  net->wmax = wmax;
  return ((size_t)2 * wmax * GT + 16 * GT) * sizeof(float) <= 150 * 1024;
}

static void chain_lds_attr(rl_engine *e, const void *kern, size_t lds) {
  if (lds <= 48 * 1024) return;
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> raised;
  std::lock_guard<std::mutex> lock(mu);
  if (raised.insert({e->device, kern}).second)
    RL_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
}

// Y[o][r] = module(X[.][r]) for `rows` rows in one launch; false: a layer too wide (the caller runs the layer kernels)
bool launch_gen_chain_rows(rl_engine *e, const rl_mlp *m, const float *x, size_t xs, uint64_t rows, float *out, size_t ys,
                           const int32_t *skip) {
  ChainNet net;
  if (!chain_net(m, &net) || rows == 0 || e->kernel_variant == 1) return false;
  const size_t lds = (size_t)2 * net.wmax * GT * sizeof(float);
  chain_lds_attr(e, reinterpret_cast<const void *>(&k_gen_chain_rows), lds);
  hipLaunchKernelGGL(k_gen_chain_rows, dim3(cdiv_g(rows, GT)), dim3(256), lds, e->stream, net, x, xs, (size_t)rows, out, ys,
                     (int)m->out_dim, skip);
  RL_HIP_CHECK(hipGetLastError());
  return true;
}

template <bool TANGENT>
void dense(rl_engine *e, const DenseArgs &a) {
  const size_t lds = (size_t)a.K * GT * sizeof(float) * (TANGENT ? 2 : 1);
  if (lds > 64 * 1024) {  // the tangent tiles of a 256-wide layer need 128 KB: raised once per device and instantiation
    static std::mutex mu;
    static std::set<int> raised;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.insert(e->device).second)
      RL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gen_dense<TANGENT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  }
  hipLaunchKernelGGL((k_gen_dense<TANGENT>), dim3(cdiv_g(a.S, GT)), dim3(256), lds, e->stream, a);
  RL_HIP_CHECK(hipGetLastError());
}

}  // namespace

// ---------------------------------------------------------------- workspace
// activation planes [hidden unit][rows], tangent planes, two delta planes of the widest layer, outputs and tangent
// outputs [2][rows]; grown on demand
void gen_ensure(rl_traj *t, const rl_mlp *m, uint64_t rows, bool tangent, bool backward) {
  GenDev &g = t->gen;
  const uint64_t units = m->hidden_units() ? m->hidden_units() : 1;
  uint32_t wmax = 1;
  for (uint32_t l = 0; l < m->n_hidden; ++l) wmax = m->widths[l] > wmax ? m->widths[l] : wmax;
  auto grow = [&](float *&p, uint64_t &cap, uint64_t need) {
    if (cap >= need) return;
    dfree(p);
    p = nullptr;
    cap = 0;
    p = dalloc<float>(need);
    cap = need;
  };
  grow(g.act, g.cap_act, units * rows);
  grow(g.z, g.cap_z, 2 * rows);
  if (tangent) {
    grow(g.tact, g.cap_tact, units * rows);
    grow(g.tz, g.cap_tz, 2 * rows);
  }
  if (backward) grow(g.delta, g.cap_delta, 2ull * wmax * rows);
  // the P-sized vectors and the slab of the update workspace grow with the module (training passes only).  The slab is
  // tracked on its own: the recurrent path grows the vectors too, but keeps its partials elsewhere
  if (backward && t->Pmax < m->P) {
    for (float **p : {&t->vec, &t->cg_x, &t->cg_r, &t->cg_p, &t->prev_params, &t->descent}) {
      dfree(*p);
      *p = nullptr;
    }
    t->Pmax = (uint32_t)m->P;
    t->vec = dalloc<float>(t->Pmax + 4);
    t->cg_x = dalloc<float>(t->Pmax);
    t->cg_r = dalloc<float>(t->Pmax);
    t->cg_p = dalloc<float>(t->Pmax);
    t->prev_params = dalloc<float>(t->Pmax);
    t->descent = dalloc<float>(t->Pmax);
  }
  if (backward) {
    uint32_t rowsA = t->nbA;
    if (t->nbV2 > rowsA) rowsA = t->nbV2;
    if (t->nbC > rowsA) rowsA = t->nbC;
    traj_ensure_slabs(t, rowsA, m->P, t->nbB > rowsA ? t->nbB : rowsA);
  }
}

// room for `rowsA` slab rows of P partial sums and `rowsB` rows of four scalars (the partials of any pass are dead once
// its reduction has run, so growing drops them)
void traj_ensure_slabs(rl_traj *t, uint64_t rowsA, uint64_t P, uint64_t rowsB) {
  if (t->cap_slabA < rowsA * P) {
    dfree(t->slabA);
    t->slabA = nullptr;
    t->cap_slabA = 0;
    t->slabA = dalloc<double>(rowsA * P);
    t->cap_slabA = rowsA * P;
  }
  if (t->cap_slabB < rowsB * 4) {
    dfree(t->slabB);
    t->slabB = nullptr;
    t->cap_slabB = 0;
    t->slabB = dalloc<double>(rowsB * 4);
    t->cap_slabB = rowsB * 4;
  }
}

void gen_free(rl_traj *t) {
  GenDev &g = t->gen;
  for (float *p : {g.act, g.tact, g.delta, g.z, g.tz}) dfree(p);
  dfree(g.no_interrupt);
  g = GenDev{};
}

// ---------------------------------------------------------------- forward (+ tangent) over `rows` samples
// x: [in_dim][.] rows `xs` apart; activations into g.act (and g.tact); outputs [out][rows] into `out` (and `tout`)
static void gen_forward_impl(rl_traj *t, const rl_mlp *m, const float *x, size_t xs, uint64_t rows, float *out,
                             const float *tangent, float *tout, const int32_t *skip) {
  rl_engine *e = t->eng;
  GenDev &g = t->gen;
  const float *in = x, *tin = nullptr;
  size_t in_stride = xs;
  uint64_t unit0 = 0;
  for (uint32_t l = 0; l < m->n_layers(); ++l) {
    const bool last = l == m->n_hidden;
    const uint64_t off = m->layer_offset(l);
    DenseArgs a{};
    a.X = in;
    a.xs = in_stride;
    a.tX = tin;
    a.txs = rows;
    a.K = (int)m->fan_in(l);
    a.N = (int)m->fan_out(l);
    a.W = m->d_params + off;
    a.b = m->d_params + m->bias_offset(l);
    a.V = tangent ? tangent + off : nullptr;
    // (a bias-less module's tangent has no bias entries either: the zeros behind the module's parameters)
    a.vb = tangent ? (m->has_bias ? a.V + (size_t)a.N * a.K : m->d_params + m->P) : nullptr;
    a.Y = last ? out : g.act + unit0 * rows;
    a.tY = last ? tout : g.tact + unit0 * rows;
    a.ys = rows;
    a.S = rows;
    a.skip = skip;
    a.act = last ? m->out_act : m->act;
    if (tangent) dense<true>(e, a);
    else dense<false>(e, a);
    if (!last) {
      in = g.act + unit0 * rows;
      tin = tangent ? g.tact + unit0 * rows : nullptr;
      in_stride = rows;
      unit0 += m->widths[l];
    }
  }
}

void launch_gen_forward(rl_traj *t, const rl_mlp *m, const float *x, size_t xs, uint64_t rows, float *out) {
  if (launch_gen_chain_rows(t->eng, m, x, xs, rows, out, (size_t)rows, nullptr)) return;
  gen_ensure(t, m, rows, false, false);
  gen_forward_impl(t, m, x, xs, rows, out, nullptr, nullptr, nullptr);
}

// ---------------------------------------------------------------- the v1 seams (kernels_update.hip)
void launch_gen_policy_pass(rl_traj *t, const rl_mlp *m, int mode, const float *d_tangent, uint64_t B_total,
                            const int32_t *d_skip, float clip_lo, float clip_hi) {
  const uint64_t B = t->B;
  const bool jvp = mode == PASS_JVP;
  gen_ensure(t, m, B, jvp, true);
  GenDev &g = t->gen;
  const size_t plane = (size_t)(t->d.T + 1) * t->d.n;
  gen_forward_impl(t, m, t->d.obs, plane, B, g.z, jvp ? d_tangent : nullptr, jvp ? g.tz : nullptr, d_skip);
  float inv_B = 1.0f / (float)B_total;
  if (mode == PASS_DQN) inv_B = 2.0f / (float)B_total;
  dim3 grid(t->nbB), blk(256);
#define TERMS(MM)                                                                                                \
  hipLaunchKernelGGL((k_gen_policy_terms<MM>), grid, blk, 0, t->eng->stream, t->d, g.z, g.tz, t->lp0, t->dz, t->slabB, \
                     inv_B, d_skip, clip_lo, clip_hi)
  if (mode == PASS_INIT) TERMS(PASS_INIT);
  else if (mode == PASS_EVAL) TERMS(PASS_EVAL);
  else if (mode == PASS_DQN) TERMS(PASS_DQN);
  else if (mode == PASS_PPO) TERMS(PASS_PPO);
  else TERMS(PASS_JVP);
#undef TERMS
  RL_HIP_CHECK(hipGetLastError());
  if (m->out_act != RL_ACT_IDENTITY && mode != PASS_EVAL) {
    // the loss terms above are functions of the module's OUTPUTS: one more factor back to the output layer's
    // pre-activation (the forward-mode pass has applied the same slope to the tangent outputs already)
    const size_t count = (size_t)m->out_dim * B;
    hipLaunchKernelGGL(k_gen_output_slope, dim3(cdiv_g(count, 256)), dim3(256), 0, t->eng->stream, t->dz, g.z, count,
                       m->out_act, d_skip);
  }
}

void launch_gen_critic_fwd(rl_traj *t, const rl_mlp *m, uint64_t B_total) {
  const uint64_t B = t->B;
  gen_ensure(t, m, B, false, true);
  GenDev &g = t->gen;
  const size_t plane = (size_t)(t->d.T + 1) * t->d.n;
  gen_forward_impl(t, m, t->d.obs, plane, B, g.z, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_gen_critic_terms, dim3(t->nbB), dim3(256), 0, t->eng->stream, t->d, g.z, t->dz, t->slabB,
                     2.0f / (float)B_total);
  RL_HIP_CHECK(hipGetLastError());
  if (m->out_act != RL_ACT_IDENTITY)
    hipLaunchKernelGGL(k_gen_output_slope, dim3(cdiv_g(B, 256)), dim3(256), 0, t->eng->stream, t->dz, g.z, (size_t)B,
                       m->out_act, (const int32_t *)nullptr);
}

// backward from t->dz ([out][B]) through the activations of the last forward: slab rows [0, nbA) x P
void launch_gen_backward(rl_traj *t, const rl_mlp *m, const int32_t *d_skip) {
  rl_engine *e = t->eng;
  GenDev &g = t->gen;
  const uint64_t B = t->B;
  const size_t plane = (size_t)(t->d.T + 1) * t->d.n;
  const uint32_t P = (uint32_t)m->P;
  const float *dY = t->dz;
  uint64_t unit_hi = m->hidden_units();
  uint32_t wmax = 1;
  for (uint32_t l = 0; l < m->n_hidden; ++l) wmax = m->widths[l] > wmax ? m->widths[l] : wmax;
  for (uint32_t l = m->n_layers(); l-- > 0;) {
    const int K = (int)m->fan_in(l), N = (int)m->fan_out(l);
    const uint64_t off = m->layer_offset(l);
    const float *X;
    size_t xs;
    if (l == 0) {
      X = t->d.obs;
      xs = plane;
    } else {
      unit_hi -= m->widths[l - 1];
      X = g.act + unit_hi * B;
      xs = B;
    }
    const uint32_t tiles = (uint32_t)(((N + 63) / 64) * ((K + 63) / 64));
    hipLaunchKernelGGL(k_gen_wgrad, dim3(tiles, t->nbA), dim3(256), 0, e->stream, dY, (size_t)B, N, X, xs, K, (size_t)B,
                       t->bwd_chunk, t->slabA, P, (uint32_t)off,
                       m->has_bias ? (uint32_t)(off + (uint64_t)N * K) : 0xFFFFFFFFu, d_skip);
    if (l > 0) {
      float *dX = g.delta + (size_t)(l & 1) * wmax * B;
      hipLaunchKernelGGL(k_gen_delta, dim3(cdiv_g(B, GT)), dim3(256), (size_t)N * GT * sizeof(float), e->stream, dY,
                         (size_t)B, N, m->d_params + off, K, X, dX, (size_t)B, m->act, d_skip);
      dY = dX;
    }
  }
  RL_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------- values for GAE / TD targets
// seq.out[0] <- V(obs[t]) for t < T, seq.succ[0] <- successor values of cut episodes (the layout launch_seq_gae and
// launch_seq_value_targets read)
void launch_gen_values(rl_traj *t, const rl_mlp *critic) {
  const uint64_t n = t->d.n, T = t->d.T, B = T * n;
  SeqDev &q = t->seq;
  if (q.out == nullptr) {
    q.out = dalloc<float>(2 * B);
    q.succ = dalloc<float>(2 * B);
  }
  gen_ensure(t, critic, B, false, false);
  GenDev &g = t->gen;
  const size_t plane = (size_t)(T + 1) * n;
  // V(term_obs[t][lane]) for every (t, lane) — only the interrupted ones are used, and the whole forward is skipped
  // (device-side flag) when the trajectory holds none — V(obs[T][lane]), then V(obs[t])
  if (!g.no_interrupt) g.no_interrupt = dalloc<int32_t>(1);
  RL_HIP_CHECK(hipMemsetAsync(g.no_interrupt, 1, sizeof(int32_t), t->eng->stream));
  hipLaunchKernelGGL(k_gen_any_interrupt, dim3(256), dim3(256), 0, t->eng->stream, t->d, g.no_interrupt);
  auto fwd = [&](const float *x, size_t xs, uint64_t rows, float *out, const int32_t *skip) {
    if (!launch_gen_chain_rows(t->eng, critic, x, xs, rows, out, (size_t)rows, skip))
      gen_forward_impl(t, critic, x, xs, rows, out, nullptr, nullptr, skip);
  };
  fwd(t->d.term_obs, (size_t)B, B, g.z, g.no_interrupt);
  fwd(t->d.obs + (size_t)T * n, plane, n, g.z + B, nullptr);
  hipLaunchKernelGGL(k_gen_successor_values, dim3(cdiv_g(B, 256)), dim3(256), 0, t->eng->stream, t->d, g.z, g.z + B,
                     q.succ);
  fwd(t->d.obs, plane, B, q.out, nullptr);
}

// ---------------------------------------------------------------- rollout, one launch sequence per step
// observe -> policy forward -> sample -> env step -> record, over the standalone env kernels (either env family)
void launch_gen_rollout(rl_env *env, const rl_mlp *policy, rl_traj *t) {
  rl_engine *e = env->eng;
  const uint32_t n = t->d.n, T = t->d.T;
  gen_ensure(t, policy, n, false, false);
  GenDev &g = t->gen;
  ChainNet net;
  if (env->kind == RL_ENV_CARTPOLE && policy->out_dim == 2 && e->kernel_variant == 0 && chain_net(policy, &net)) {
    // CartPole lanes: all T steps in one launch
    const size_t lds = ((size_t)2 * net.wmax * GT + 16 * GT) * sizeof(float);
    if (env->D == 5) {
      chain_lds_attr(e, reinterpret_cast<const void *>(&k_gen_rollout_cartpole<5>), lds);
      hipLaunchKernelGGL(k_gen_rollout_cartpole<5>, dim3(cdiv_g(n, GT)), dim3(ROLL_WAVES * 64), lds, e->stream, env->dev, env->st, t->d,
                         net, env->t_global);
    } else {
      chain_lds_attr(e, reinterpret_cast<const void *>(&k_gen_rollout_cartpole<4>), lds);
      hipLaunchKernelGGL(k_gen_rollout_cartpole<4>, dim3(cdiv_g(n, GT)), dim3(ROLL_WAVES * 64), lds, e->stream, env->dev, env->st, t->d,
                         net, env->t_global);
    }
    RL_HIP_CHECK(hipGetLastError());
    env->t_global += T;
    return;
  }
  launch_rollout_stepwise(env, t, g.z, [&](uint32_t) {
    gen_forward_impl(t, policy, env->d_obs, (size_t)n, n, g.z, nullptr, nullptr, nullptr);
  });
}

// T steps of every lane as launch sequences: `forward(step)` leaves the policy's logits for env->d_obs ([D][n]) in z
// ([2][n]); the rest of a step is PolicyActor::act, the env's step kernel and the step's record.  Advances t_global.
void launch_rollout_stepwise(rl_env *env, rl_traj *t, const float *z, const std::function<void(uint32_t)> &forward) {
  rl_engine *e = env->eng;
  const uint32_t n = t->d.n, T = t->d.T;
  const dim3 grid(cdiv_g(n, 256)), blk(256);
  launch_env_observe(env, env->d_obs);
  if (env->kind == RL_ENV_CARTPOLE && env->A == 2 && e->kernel_variant != 1) {
    // CartPole lanes: a step is the policy's launches and one launch for everything else
    hipLaunchKernelGGL(k_gen_record_obs, grid, blk, 0, e->stream, t->d, env->d_obs, 0u);
    for (uint32_t step = 0; step < T; ++step) {
      forward(step);
      if (env->D == 5)
        hipLaunchKernelGGL(k_gen_step_cartpole<5>, grid, blk, 0, e->stream, env->dev, env->st, t->d, z, env->t_global,
                           step, env->d_obs);
      else
        hipLaunchKernelGGL(k_gen_step_cartpole<4>, grid, blk, 0, e->stream, env->dev, env->st, t->d, z, env->t_global,
                           step, env->d_obs);
      env->t_global += 1;
    }
    RL_HIP_CHECK(hipGetLastError());
    return;
  }
  for (uint32_t step = 0; step < T; ++step) {
    hipLaunchKernelGGL(k_gen_record_obs, grid, blk, 0, e->stream, t->d, env->d_obs, step);
    forward(step);
    hipLaunchKernelGGL(k_gen_sample_actions, grid, blk, 0, e->stream, env->dev, z, n, env->t_global, env->d_actions);
    launch_env_step(env);  // leaves reward, flag, the next observation and the interrupted successor in the env's buffers
    env->t_global += 1;
    hipLaunchKernelGGL(k_gen_record_step, grid, blk, 0, e->stream, t->d, env->d_actions, env->d_reward, env->d_flag,
                       env->d_term_obs, step);
  }
  hipLaunchKernelGGL(k_gen_record_obs, grid, blk, 0, e->stream, t->d, env->d_obs, T);
  RL_HIP_CHECK(hipGetLastError());
}

// dW = dY X^T (+ db) over S samples in `rows` slab rows of `chunk` samples: the weight-gradient kernel on caller-given
// planes (the stacked recurrent layers' gate deltas and inputs, kernels_seq_stack.hip)
void launch_gen_wgrad_planes(rl_traj *t, const float *dY, size_t dys, int N, const float *X, size_t xs, int K, size_t S,
                             uint32_t chunk, uint32_t rows, uint32_t P, uint32_t offW, uint32_t offB,
                             const int32_t *d_skip) {
  const uint32_t tiles = (uint32_t)(((N + 63) / 64) * ((K + 63) / 64));
  hipLaunchKernelGGL(k_gen_wgrad, dim3(tiles, rows), dim3(256), 0, t->eng->stream, dY, dys, N, X, xs, K, S, chunk,
                     t->slabA, P, offW, offB, d_skip);
}
