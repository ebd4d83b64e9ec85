// kernels_seq_stack.hip — recurrent chains with RnnBaseConfig::num_layers > 1 (src/torch/modules/seq/rnn/mod.rs:20-45,
// 223-257): stacked GRU / LSTM layers -> ReLU -> Mlp([h]) (modules/chain.rs:127-186).  Flat order per layer
// [W_ih (gates x layer input), W_hh (gates x hidden), b_ih, b_hh]; layer 0 reads the observation features, layer l > 0
// the hidden output of layer l - 1 of the same step (Tensor::gru / ::lstm with num_layers, seq/rnn/gru.rs:41-66); no
// dropout.  Every layer's state is zero at t = 0 and after a step whose flag ends the episode.
//
// The single-layer chains have fused tile kernels (kernels_seq*.hip, built for 5 -> 128 -> 128); this is the general
// path for the stacked ones, at the module's own widths: ONE THREAD PER LANE, 64 lanes per workgroup, and the unit loop
// of a layer dealt in quads to the workgroup's eight waves (a quad's gate rows share every loaded input value: 12-16 fma
// per load; weights come through wave-uniform loads).  States and the step's scratch live in [unit][lane] arrays in
// device memory (L2-resident: a workgroup's slice is 64 lanes wide), records of a training forward in [unit][sample]
// planes — the layout the per-layer weight-gradient kernel of kernels_general.hip reads, which forms every dW = dY X^T
// here as well.  Arithmetic contract of the forward: acc = bias; acc = fma(in_k, w_k, acc), k ascending, and the cells'
// operation order of seq_common.hpp — the tests' C restatement follows the same contract, so outputs compare bit for
// bit; backward and tangent passes are checked against an f64 restatement to f32 tolerance.
//   k_stack_forward   teacher-forced forward over T steps (+ successor outputs of cut episodes, + the record)
//   k_stack_step      one rollout step from the env's observation buffer (states persist between the launches)
//   k_stack_backward  reverse scan: d loss / d pre-activations of every layer and of the head into planes
//   k_stack_tangent   forward-mode derivative along a parameter tangent through the recorded activations
#include "abi_internal.hpp"

namespace {

constexpr int SL = 64;  // lanes per workgroup
constexpr int SW = 8;   // waves per workgroup
constexpr int UQ = 4;   // units per pass (forward-style products: the quad's gate rows share every loaded input)
constexpr int KB = 8;   // outputs per pass of the transposed products of the backward scan
constexpr int LSTM = RL_MODULE_LSTM_MLP, GRU = RL_MODULE_GRU_MLP;

// record planes of a layer: [RPN][H][B]
enum { RP_G0 = 0, RP_G1 = 1, RP_G2 = 2, RP_G3 = 3, RP_HPREV = 4, RP_HOUT = 5, RP_CPREV = 6, RP_TC = 7, RPN = 8 };
// state sets (h at slot 2 s, c at 2 s + 1)
enum { SET_A = 0, SET_B = 1, SET_PEEK = 2, SET_TA = 3, SET_TB = 4, N_SETS = 5 };

struct StackNet {
  const float *p;  // flat parameters (or a tangent in the same layout)
  const float *z;  // >= 4 H zeros: what a layer WITHOUT bias vectors starts its gate rows from (NULL: the layers have them)
  int D, H, H2, A, L;
  uint32_t off[RL_RNN_MAX_LAYERS + 1];  // W_ih of layer l; [L]: the head's W1
};
// b_ih / b_hh of a layer whose W_hh starts at Whh
template <int G>
__device__ __forceinline__ const float *stack_bih(const StackNet &net, const float *Whh) {
  return net.z != nullptr ? net.z : Whh + (size_t)G * net.H * net.H;
}
template <int G>
__device__ __forceinline__ const float *stack_bhh(const StackNet &net, const float *Whh) {
  return net.z != nullptr ? net.z : Whh + (size_t)G * net.H * net.H + G * net.H;
}

struct StackWs {
  float *st, *u, *din, *dst, *rec, *a1, *ur, *dg, *du;
  uint32_t n;
  uint64_t B;
};

template <int CELL>
__host__ __device__ constexpr int gates() { return CELL == LSTM ? 4 : 3; }

__device__ __forceinline__ float *slot(const StackWs &ws, const StackNet &net, int set, int hc, int l) {
  return ws.st + (((size_t)(2 * set + hc) * net.L + l) * net.H) * ws.n;
}
__device__ __forceinline__ float *rec_plane(const StackWs &ws, const StackNet &net, int l, int which) {
  return ws.rec + (((size_t)l * RPN + which) * net.H) * ws.B;
}

struct LaneCtx {
  uint32_t ii;  // lane (clamped into range: threads past n compute along and store nothing)
  bool live;
  int wave;
};

__device__ __forceinline__ LaneCtx lane_ctx(uint32_t n) {
  LaneCtx c;
  const uint32_t i = blockIdx.x * SL + (threadIdx.x & (SL - 1));
  c.live = i < n;
  c.ii = c.live ? i : n - 1;
  c.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  return c;
}

// acc[r] = fma(x_k, W[row[r]][k], acc[r]) for k ascending: R rows of a [.][K] matrix against one input vector.  Four k
// per trip: a row's four weights are contiguous (one wide wave-uniform load each), the four inputs are in flight
// together; every sum stays the k-ascending fma chain of the arithmetic contract.
template <int R, class XF>
__device__ __forceinline__ void dot_rows(float (&acc)[R], const float *__restrict__ W, const int (&row)[R], int K,
                                         XF x_at) {
  int k = 0;
  for (; k + 4 <= K; k += 4) {
    float x[4], w[R][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = x_at(k + q);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) w[r][q] = W[(size_t)row[r] * K + k + q];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(x[q], w[r][q], acc[r]);
  }
  for (; k < K; ++k) {
    const float x = x_at(k);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(x, W[(size_t)row[r] * K + k], acc[r]);
  }
}

// acc[u] += sum over r < R of d_r W[r][k0 + u]: N outputs k0 .. k0 + N - 1 of the transposed product W^T d (W rows `ld`
// floats apart).  FULL: all N outputs exist (contiguous weights, wide loads); else indices past kmax are clamped.
template <int N, bool FULL, class DF>
__device__ __forceinline__ void tdot_n(float (&acc)[N], const float *__restrict__ W, int ld, int k0, int kmax, int R,
                                       DF d_at) {
  int kk[N];
#pragma unroll
  for (int u = 0; u < N; ++u) kk[u] = FULL ? k0 + u : (k0 + u < kmax ? k0 + u : kmax);
  int r = 0;
  for (; r + 4 <= R; r += 4) {
    float d[4], w[4][N];
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = d_at(r + q);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < N; ++u) w[q][u] = W[(size_t)(r + q) * ld + kk[u]];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int u = 0; u < N; ++u) acc[u] = __builtin_fmaf(d[q], w[q][u], acc[u]);
  }
  for (; r < R; ++r) {
    const float d = d_at(r);
#pragma unroll
    for (int u = 0; u < N; ++u) acc[u] = __builtin_fmaf(d, W[(size_t)r * ld + kk[u]], acc[u]);
  }
}
template <int N, class DF>
__device__ __forceinline__ void tdot(float (&acc)[N], const float *__restrict__ W, int ld, int k0, int K, int R, DF d_at) {
  if (k0 + N <= K) tdot_n<N, true>(acc, W, ld, k0, K - 1, R, d_at);
  else tdot_n<N, false>(acc, W, ld, k0, K - 1, R, d_at);
}

// layer l of one step: in[k * in_stride + lane] (K inputs), states (h, c) -> (hn, cn); `fresh`: the lane's state is zero
template <int CELL, bool REC>
__device__ __forceinline__ void stack_cell(const StackNet &net, const StackWs &ws, int l, const float *__restrict__ in,
                                           size_t in_stride, const float *__restrict__ h, const float *__restrict__ c,
                                           bool fresh, float *__restrict__ hn, float *__restrict__ cn, const LaneCtx &lc,
                                           size_t b) {
  constexpr int G = gates<CELL>();
  const int H = net.H, K = l == 0 ? net.D : net.H;
  const size_t n = ws.n;
  const float *__restrict__ Wih = net.p + net.off[l], *__restrict__ Whh = Wih + (size_t)G * H * K;
  const float *__restrict__ bih = stack_bih<G>(net, Whh), *__restrict__ bhh = stack_bhh<G>(net, Whh);
  for (int j0 = UQ * lc.wave; j0 < H; j0 += UQ * SW) {
    float gif[G * UQ], ghf[G * UQ];  // [gate][unit of the quad]
    int row[G * UQ];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        const int j = j0 + u < H ? j0 + u : H - 1;
        row[g * UQ + u] = g * H + j;
        gif[g * UQ + u] = bih[g * H + j];
        ghf[g * UQ + u] = bhh[g * H + j];
      }
    dot_rows(gif, Wih, row, K, [&](int k) { return in[(size_t)k * in_stride + lc.ii]; });
    dot_rows(ghf, Whh, row, H, [&](int k) { return fresh ? 0.0f : h[(size_t)k * n + lc.ii]; });
    float gi[G][UQ], gh[G][UQ];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        gi[g][u] = gif[g * UQ + u];
        gh[g][u] = ghf[g * UQ + u];
      }
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const int j = j0 + u;
      if (j >= H) break;
      const float hp = fresh ? 0.0f : h[(size_t)j * n + lc.ii];
      float hnew;
      if (CELL == LSTM) {
        const float cp = fresh ? 0.0f : c[(size_t)j * n + lc.ii];
        const float ig = rl_sigmoidf(gh[0][u] + gi[0][u]), fg = rl_sigmoidf(gh[1][u] + gi[1][u]);
        const float gg = rl_tanhf(gh[2][u] + gi[2][u]), og = rl_sigmoidf(gh[G - 1][u] + gi[G - 1][u]);
        const float fc = fg * cp;
        const float iga = ig * gg;
        const float cc = fc + iga;
        const float tc = rl_tanhf(cc);
        hnew = og * tc;
        if (lc.live) {
          cn[(size_t)j * n + lc.ii] = cc;
          if (REC) {
            rec_plane(ws, net, l, RP_G0)[(size_t)j * ws.B + b] = ig;
            rec_plane(ws, net, l, RP_G1)[(size_t)j * ws.B + b] = fg;
            rec_plane(ws, net, l, RP_G2)[(size_t)j * ws.B + b] = gg;
            rec_plane(ws, net, l, RP_G3)[(size_t)j * ws.B + b] = og;
            rec_plane(ws, net, l, RP_CPREV)[(size_t)j * ws.B + b] = cp;
            rec_plane(ws, net, l, RP_TC)[(size_t)j * ws.B + b] = tc;
          }
        }
      } else {
        const float r = rl_sigmoidf(gh[0][u] + gi[0][u]);
        const float z = rl_sigmoidf(gh[1][u] + gi[1][u]);
        const float rn = gh[2][u] * r;
        const float nn = rl_tanhf(gi[2][u] + rn);
        const float dn = hp - nn;
        const float hz = dn * z;
        hnew = hz + nn;
        if (REC && lc.live) {
          rec_plane(ws, net, l, RP_G0)[(size_t)j * ws.B + b] = r;
          rec_plane(ws, net, l, RP_G1)[(size_t)j * ws.B + b] = z;
          rec_plane(ws, net, l, RP_G2)[(size_t)j * ws.B + b] = nn;
          rec_plane(ws, net, l, RP_G3)[(size_t)j * ws.B + b] = gh[2][u];
        }
      }
      if (lc.live) {
        hn[(size_t)j * n + lc.ii] = hnew;
        if (REC) {
          rec_plane(ws, net, l, RP_HPREV)[(size_t)j * ws.B + b] = hp;
          rec_plane(ws, net, l, RP_HOUT)[(size_t)j * ws.B + b] = hnew;
        }
      }
    }
  }
}

// Chain's ReLU and the Mlp on the top layer's output `top` ([H][n]); out[q] in every thread.  One barrier inside.
template <bool REC>
__device__ __forceinline__ void stack_head(const StackNet &net, const StackWs &ws, const float *__restrict__ top,
                                           const LaneCtx &lc, size_t b, float (&out)[2]) {
  const int H = net.H, H2 = net.H2, A = net.A;
  const size_t n = ws.n;
  const float *__restrict__ W1 = net.p + net.off[net.L], *__restrict__ b1 = W1 + (size_t)H2 * H;
  const float *__restrict__ W2 = b1 + H2, *__restrict__ b2 = W2 + (size_t)A * H2;
  for (int j0 = UQ * lc.wave; j0 < H2; j0 += UQ * SW) {
    float acc[UQ];
    int jj[UQ];
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      jj[u] = j0 + u < H2 ? j0 + u : H2 - 1;
      acc[u] = b1[jj[u]];
    }
    dot_rows(acc, W1, jj, H, [&](int k) {
      const float tv = top[(size_t)k * n + lc.ii];
      return tv > 0.0f ? tv : 0.0f;
    });
#pragma unroll
    for (int u = 0; u < UQ; ++u)
      if (j0 + u < H2 && lc.live) {
        const float uu = acc[u] > 0.0f ? acc[u] : 0.0f;
        ws.u[(size_t)(j0 + u) * n + lc.ii] = uu;
        if (REC) ws.ur[(size_t)(j0 + u) * ws.B + b] = uu;
      }
  }
  if (REC && lc.live)
    for (int k = lc.wave; k < H; k += SW) {
      const float tv = top[(size_t)k * n + lc.ii];
      ws.a1[(size_t)k * ws.B + b] = tv > 0.0f ? tv : 0.0f;
    }
  __syncthreads();
  const int qq[2] = {0, A > 1 ? 1 : 0};
  out[0] = b2[qq[0]];
  out[1] = b2[qq[1]];
  dot_rows(out, W2, qq, H2, [&](int j) { return ws.u[(size_t)j * n + lc.ii]; });
}

// all layers + head of one step: states of set `cur` (zero where `fresh`) -> set `nxt`
template <int CELL, bool REC>
__device__ __forceinline__ void stack_module_step(const StackNet &net, const StackWs &ws, const float *__restrict__ x,
                                                  size_t x_stride, int cur, int nxt, bool fresh, const LaneCtx &lc,
                                                  size_t b, float (&out)[2]) {
  for (int l = 0; l < net.L; ++l) {
    const float *in = l == 0 ? x : slot(ws, net, nxt, 0, l - 1);
    stack_cell<CELL, REC>(net, ws, l, in, l == 0 ? x_stride : (size_t)ws.n, slot(ws, net, cur, 0, l),
                          slot(ws, net, cur, 1, l), fresh, slot(ws, net, nxt, 0, l), slot(ws, net, nxt, 1, l), lc, b);
    __syncthreads();
  }
  stack_head<REC>(net, ws, slot(ws, net, nxt, 0, net.L - 1), lc, b, out);
}

// teacher-forced forward: out / succ [A][T][n] (succ may be NULL), the contract of launch_gru_seq_forward
template <int CELL, bool REC>
__global__ void __launch_bounds__(SL *SW) k_stack_forward(StackNet net, StackWs ws, TrajDev tr, float *__restrict__ out,
                                                          float *__restrict__ succ, const int32_t *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  const LaneCtx lc = lane_ctx(tr.n);
  const size_t n = tr.n, T = tr.T, B = T * n, plane = (T + 1) * n;
  bool fresh = true;
  int cur = SET_A;
  for (size_t t = 0; t < T; ++t) {
    const size_t b = t * n + lc.ii;
    float o[2];
    stack_module_step<CELL, REC>(net, ws, tr.obs + t * n, plane, cur, cur ^ 1, fresh, lc, b, o);
    const uint8_t f = tr.flag[b];
    if (lc.live && lc.wave == 0)
      for (int q = 0; q < net.A; ++q) out[(size_t)q * B + b] = o[q];
    if (succ != nullptr) {
      const bool need = f == RL_SUCC_INTERRUPT || (f == RL_SUCC_CONTINUE && t == T - 1);
      float o2[2] = {0.0f, 0.0f};
      if (__syncthreads_or(need && lc.live ? 1 : 0)) {
        // the successor observation (the interrupted step's, or obs[T] at the horizon) evaluated from the states after
        // step t, which are not advanced: the results go to the third state set.  Per-lane plane stride: term_obs planes
        // are T * n apart, obs planes (T + 1) * n.
        const bool intr = f == RL_SUCC_INTERRUPT;
        stack_module_step<CELL, false>(net, ws, intr ? tr.term_obs + t * n : tr.obs + T * n, intr ? B : plane, cur ^ 1,
                                       SET_PEEK, false, lc, b, o2);
      }
      if (lc.live && lc.wave == 0)
        for (int q = 0; q < net.A; ++q) succ[(size_t)q * B + b] = need ? o2[q] : 0.0f;
    }
    fresh = f != RL_SUCC_CONTINUE;
    cur ^= 1;
  }
}

// one rollout step: env observation buffer obs [D][n] -> logits z [A][n]; the lane's states restart where the step
// before ended its episode (flag_prev: that step's successor codes) or at the first step of a collection
template <int CELL>
__global__ void __launch_bounds__(SL *SW) k_stack_step(StackNet net, StackWs ws, const float *__restrict__ obs,
                                                       const uint8_t *__restrict__ flag_prev, int first, int parity,
                                                       float *__restrict__ z) {
  const LaneCtx lc = lane_ctx(ws.n);
  const bool fresh = first != 0 || flag_prev[lc.ii] != RL_SUCC_CONTINUE;
  float o[2];
  stack_module_step<CELL, false>(net, ws, obs, (size_t)ws.n, parity, parity ^ 1, fresh, lc, 0, o);
  if (lc.live && lc.wave == 0)
    for (int q = 0; q < net.A; ++q) z[(size_t)q * ws.n + lc.ii] = o[q];
}

// Reverse scan over the record of the last training forward: dz [A][B] -> d loss / d pre-activation planes
//   du [H2][B]   the head's hidden layer
//   dg [L][4H][B] the layers' gates — GRU rows [r; z; n (input side); n (hidden side, x r)], LSTM rows [i; f; g; o]
// from which the weight-gradient kernel forms every dW / db.  Per step and layer: (A) this wave's units turn the gradient
// into their output (carried from step t + 1, zero across an episode boundary, plus what the layer above — or the head —
// sends down at this step) into gate deltas; (B) this wave's input indices k gather W_hh^T / W_ih^T times those deltas:
// the gradient carried to step t - 1 and the one sent to the layer below.
template <int CELL>
__global__ void __launch_bounds__(SL *SW) k_stack_backward(StackNet net, StackWs ws, TrajDev tr,
                                                           const float *__restrict__ dz,
                                                           const int32_t *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  constexpr int G = gates<CELL>();
  const LaneCtx lc = lane_ctx(tr.n);
  const size_t n = tr.n, T = tr.T, B = T * n;
  const int H = net.H, H2 = net.H2, A = net.A, L = net.L;
  const float *__restrict__ W1 = net.p + net.off[L], *__restrict__ W2 = W1 + (size_t)H2 * H + H2;
  for (size_t t = T; t-- > 0;) {
    const size_t b = t * n + lc.ii;
    const bool ended = t == T - 1 || tr.flag[b] != RL_SUCC_CONTINUE;  // nothing flows in from step t + 1
    for (int j0 = KB * lc.wave; j0 < H2; j0 += KB * SW) {
      float acc[KB] = {};
      tdot(acc, W2, H2, j0, H2, A, [&](int q) { return dz[(size_t)q * B + b]; });
#pragma unroll
      for (int u = 0; u < KB; ++u)
        if (j0 + u < H2 && lc.live)
          ws.du[(size_t)(j0 + u) * B + b] = ws.ur[(size_t)(j0 + u) * B + b] > 0.0f ? acc[u] : 0.0f;
    }
    __syncthreads();
    {
      const float *__restrict__ top = rec_plane(ws, net, L - 1, RP_HOUT);
      for (int k0 = KB * lc.wave; k0 < H; k0 += KB * SW) {
        float acc[KB] = {};
        tdot(acc, W1, H, k0, H, H2, [&](int j) { return ws.du[(size_t)j * B + b]; });
#pragma unroll
        for (int u = 0; u < KB; ++u)
          if (k0 + u < H && lc.live)
            ws.din[(size_t)(k0 + u) * n + lc.ii] = top[(size_t)(k0 + u) * B + b] > 0.0f ? acc[u] : 0.0f;
      }
    }
    __syncthreads();
    for (int l = L - 1; l >= 0; --l) {
      const int K = l == 0 ? net.D : H;
      const float *__restrict__ Wih = net.p + net.off[l], *__restrict__ Whh = Wih + (size_t)G * H * K;
      float *__restrict__ dhc = ws.dst + ((size_t)l * H) * n, *__restrict__ dcc = ws.dst + ((size_t)(L + l) * H) * n;
      float *__restrict__ dg = ws.dg + ((size_t)l * 4 * H) * B;
      // (A)
      for (int j0 = UQ * lc.wave; j0 < H; j0 += UQ * SW)
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
          const int j = j0 + u;
          if (j >= H) break;
          const size_t jb = (size_t)j * B + b, jn = (size_t)j * n + lc.ii;
          const float dhl = (ended ? 0.0f : dhc[jn]) + ws.din[jn];
          if (CELL == LSTM) {
            const float ig = rec_plane(ws, net, l, RP_G0)[jb], fg = rec_plane(ws, net, l, RP_G1)[jb];
            const float gg = rec_plane(ws, net, l, RP_G2)[jb], og = rec_plane(ws, net, l, RP_G3)[jb];
            const float cp = rec_plane(ws, net, l, RP_CPREV)[jb], tc = rec_plane(ws, net, l, RP_TC)[jb];
            const float dO = dhl * tc;
            const float dcn = (ended ? 0.0f : dcc[jn]) + dhl * og * (1.0f - tc * tc);
            if (lc.live) {
              dg[jb] = dcn * gg * ig * (1.0f - ig);
              dg[(size_t)H * B + jb] = dcn * cp * fg * (1.0f - fg);
              dg[(size_t)2 * H * B + jb] = dcn * ig * (1.0f - gg * gg);
              dg[(size_t)3 * H * B + jb] = dO * og * (1.0f - og);
              dcc[jn] = dcn * fg;
              dhc[jn] = 0.0f;
            }
          } else {
            const float r = rec_plane(ws, net, l, RP_G0)[jb], z = rec_plane(ws, net, l, RP_G1)[jb];
            const float nn = rec_plane(ws, net, l, RP_G2)[jb], ghn = rec_plane(ws, net, l, RP_G3)[jb];
            const float hp = rec_plane(ws, net, l, RP_HPREV)[jb];
            const float dzg = dhl * (hp - nn);
            const float dn = dhl * (1.0f - z);
            const float dpn = dn * (1.0f - nn * nn);
            const float dr = dpn * ghn;
            if (lc.live) {
              dg[jb] = dr * r * (1.0f - r);
              dg[(size_t)H * B + jb] = dzg * z * (1.0f - z);
              dg[(size_t)2 * H * B + jb] = dpn;
              dg[(size_t)3 * H * B + jb] = dpn * r;
              dhc[jn] = dhl * z;  // h' = (h - n) z + n: the direct path
            }
          }
        }
      __syncthreads();
      // (B)
      for (int k0 = KB * lc.wave; k0 < H; k0 += KB * SW) {
        float acc[KB] = {}, acc2[KB] = {};
        // hidden-side delta of gate row `row`: the GRU's n rows take their fourth plane
        tdot(acc, Whh, H, k0, H, G * H, [&](int row) {
          const int hrow = (CELL == GRU && row >= 2 * H) ? row + H : row;
          return dg[(size_t)hrow * B + b];
        });
        if (l > 0) tdot(acc2, Wih, K, k0, H, G * H, [&](int row) { return dg[(size_t)row * B + b]; });
#pragma unroll
        for (int u = 0; u < KB; ++u)
          if (k0 + u < H && lc.live) {
            const size_t kn = (size_t)(k0 + u) * n + lc.ii;
            dhc[kn] = dhc[kn] + acc[u];
            if (l > 0) ws.din[kn] = acc2[u];
          }
      }
      __syncthreads();
    }
  }
}

// Forward-mode derivative along the tangent `tv` (same layout as the parameters) through the recorded activations:
// out_dot [A][T][n].  With p = a gate's pre-activation: p_dot = V_ih x + v_bih + W_ih x_dot + V_hh h + v_bhh + W_hh h_dot.
template <int CELL>
__global__ void __launch_bounds__(SL *SW) k_stack_tangent(StackNet net, StackNet tv, StackWs ws, TrajDev tr,
                                                          float *__restrict__ out_dot,
                                                          const int32_t *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  constexpr int G = gates<CELL>();
  const LaneCtx lc = lane_ctx(tr.n);
  const size_t n = tr.n, T = tr.T, B = T * n, plane = (T + 1) * n;
  const int H = net.H, H2 = net.H2, A = net.A, L = net.L;
  const float *__restrict__ W1 = net.p + net.off[L], *__restrict__ W2 = W1 + (size_t)H2 * H + H2;
  const float *__restrict__ V1 = tv.p + net.off[L], *__restrict__ vb1 = V1 + (size_t)H2 * H;
  const float *__restrict__ V2 = vb1 + H2, *__restrict__ vb2 = V2 + (size_t)A * H2;
  bool fresh = true;
  int cur = SET_TA;
  for (size_t t = 0; t < T; ++t) {
    const size_t b = t * n + lc.ii;
    const int nxt = cur == SET_TA ? SET_TB : SET_TA;
    for (int l = 0; l < L; ++l) {
      const int K = l == 0 ? net.D : H;
      const float *__restrict__ Wih = net.p + net.off[l], *__restrict__ Whh = Wih + (size_t)G * H * K;
      const float *__restrict__ Vih = tv.p + net.off[l], *__restrict__ Vhh = Vih + (size_t)G * H * K;
      const float *__restrict__ vbih = stack_bih<G>(tv, Vhh), *__restrict__ vbhh = stack_bhh<G>(tv, Vhh);
      const float *__restrict__ in = l == 0 ? tr.obs + t * n + lc.ii : rec_plane(ws, net, l - 1, RP_HOUT) + b;
      const size_t in_stride = l == 0 ? plane : B;
      const float *__restrict__ ind = l == 0 ? nullptr : slot(ws, net, nxt, 0, l - 1);
      const float *__restrict__ hprev = rec_plane(ws, net, l, RP_HPREV);
      const float *__restrict__ hd = slot(ws, net, cur, 0, l), *__restrict__ cd = slot(ws, net, cur, 1, l);
      float *__restrict__ hdn = slot(ws, net, nxt, 0, l), *__restrict__ cdn = slot(ws, net, nxt, 1, l);
      for (int j0 = UQ * lc.wave; j0 < H; j0 += UQ * SW) {
        float gidf[G * UQ], ghdf[G * UQ];
        int row[G * UQ];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
            const int j = j0 + u < H ? j0 + u : H - 1;
            row[g * UQ + u] = g * H + j;
            gidf[g * UQ + u] = vbih[g * H + j];
            ghdf[g * UQ + u] = vbhh[g * H + j];
          }
        dot_rows(gidf, Vih, row, K, [&](int k) { return in[(size_t)k * in_stride]; });
        if (l > 0) dot_rows(gidf, Wih, row, K, [&](int k) { return ind[(size_t)k * n + lc.ii]; });
        dot_rows(ghdf, Vhh, row, H, [&](int k) { return hprev[(size_t)k * B + b]; });
        dot_rows(ghdf, Whh, row, H, [&](int k) { return fresh ? 0.0f : hd[(size_t)k * n + lc.ii]; });
        float gid[G][UQ], ghd[G][UQ];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
            gid[g][u] = gidf[g * UQ + u];
            ghd[g][u] = ghdf[g * UQ + u];
          }
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
          const int j = j0 + u;
          if (j >= H) break;
          const size_t jb = (size_t)j * B + b, jn = (size_t)j * n + lc.ii;
          float hnd;
          if (CELL == LSTM) {
            const float ig = rec_plane(ws, net, l, RP_G0)[jb], fg = rec_plane(ws, net, l, RP_G1)[jb];
            const float gg = rec_plane(ws, net, l, RP_G2)[jb], og = rec_plane(ws, net, l, RP_G3)[jb];
            const float cp = rec_plane(ws, net, l, RP_CPREV)[jb], tc = rec_plane(ws, net, l, RP_TC)[jb];
            const float i_d = ig * (1.0f - ig) * (gid[0][u] + ghd[0][u]);
            const float f_d = fg * (1.0f - fg) * (gid[1][u] + ghd[1][u]);
            const float g_d = (1.0f - gg * gg) * (gid[2][u] + ghd[2][u]);
            const float o_d = og * (1.0f - og) * (gid[G - 1][u] + ghd[G - 1][u]);
            const float cdp = fresh ? 0.0f : cd[jn];
            const float cn_d = f_d * cp + fg * cdp + i_d * gg + ig * g_d;
            const float tc_d = (1.0f - tc * tc) * cn_d;
            hnd = o_d * tc + og * tc_d;
            if (lc.live) cdn[jn] = cn_d;
          } else {
            const float r = rec_plane(ws, net, l, RP_G0)[jb], z = rec_plane(ws, net, l, RP_G1)[jb];
            const float nn = rec_plane(ws, net, l, RP_G2)[jb], ghn = rec_plane(ws, net, l, RP_G3)[jb];
            const float hp = hprev[jb];
            const float r_d = r * (1.0f - r) * (gid[0][u] + ghd[0][u]);
            const float z_d = z * (1.0f - z) * (gid[1][u] + ghd[1][u]);
            const float n_d = (1.0f - nn * nn) * (gid[2][u] + r_d * ghn + r * ghd[2][u]);
            const float hdp = fresh ? 0.0f : hd[jn];
            hnd = (hdp - n_d) * z + (hp - nn) * z_d + n_d;
          }
          if (lc.live) hdn[jn] = hnd;
        }
      }
      __syncthreads();
    }
    {
      const float *__restrict__ top = rec_plane(ws, net, L - 1, RP_HOUT);
      const float *__restrict__ topd = slot(ws, net, nxt, 0, L - 1);
      for (int j0 = UQ * lc.wave; j0 < H2; j0 += UQ * SW) {
        float acc[UQ];
        int jj[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
          jj[u] = j0 + u < H2 ? j0 + u : H2 - 1;
          acc[u] = vb1[jj[u]];
        }
        dot_rows(acc, V1, jj, H, [&](int k) { return ws.a1[(size_t)k * B + b]; });
        dot_rows(acc, W1, jj, H,
                 [&](int k) { return top[(size_t)k * B + b] > 0.0f ? topd[(size_t)k * n + lc.ii] : 0.0f; });
#pragma unroll
        for (int u = 0; u < UQ; ++u)
          if (j0 + u < H2 && lc.live)
            ws.u[(size_t)(j0 + u) * n + lc.ii] = ws.ur[(size_t)(j0 + u) * B + b] > 0.0f ? acc[u] : 0.0f;
      }
    }
    __syncthreads();
    if (lc.wave == 0) {
      const int qq[2] = {0, A > 1 ? 1 : 0};
      float od[2] = {vb2[qq[0]], vb2[qq[1]]};
      dot_rows(od, V2, qq, H2, [&](int j) { return ws.ur[(size_t)j * B + b]; });
      dot_rows(od, W2, qq, H2, [&](int j) { return ws.u[(size_t)j * n + lc.ii]; });
      if (lc.live)
        for (int q = 0; q < A; ++q) out_dot[(size_t)q * B + b] = od[q];
    }
    fresh = tr.flag[b] != RL_SUCC_CONTINUE;
    cur = nxt;
  }
}

StackNet stack_net(const rl_mlp *m, const float *p) {
  StackNet s;
  s.p = p;
  s.z = m->has_bias ? nullptr : m->d_params + m->P;  // (the module's zeros also serve a tangent's absent bias entries)
  s.D = (int)m->in_dim;
  s.H = (int)m->gru_hidden;
  s.H2 = (int)m->hidden;
  s.A = (int)m->out_dim;
  s.L = (int)m->rnn_layers;
  for (uint32_t l = 0; l <= RL_RNN_MAX_LAYERS; ++l) s.off[l] = l <= m->rnn_layers ? (uint32_t)m->rnn_layer_offset(l) : 0;
  return s;
}

StackWs stack_ws(const rl_traj *t) {
  const SeqDev::Stack &k = t->seq.stack;
  StackWs w;
  w.st = k.st;
  w.u = k.u;
  w.din = k.din;
  w.dst = k.dst;
  w.rec = k.rec;
  w.a1 = k.a1;
  w.ur = k.ur;
  w.dg = k.dg;
  w.du = k.du;
  w.n = t->d.n;
  w.B = (uint64_t)t->d.T * t->d.n;
  return w;
}

void grow(float *&p, uint64_t &cap, uint64_t want) {
  if (cap >= want) return;
  dfree(p);
  p = nullptr;
  cap = 0;
  p = dalloc<float>(want);
  cap = want;
}

inline uint32_t cdiv_k(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace

void stack_ensure(rl_traj *t, const rl_mlp *mod, bool training) {
  RL_REQUIRE(rl_module_is_recurrent(mod->kind) && mod->lane_kernels(), "not a module of the lane-per-thread kernels");
  RL_REQUIRE(mod->in_dim == t->d.D, "module input width does not match the trajectory");
  SeqDev &q = t->seq;
  SeqDev::Stack &k = q.stack;
  const uint64_t n = t->d.n, T = t->d.T, B = T * n, L = mod->rnn_layers, H = mod->gru_hidden, H2 = mod->hidden;
  if (q.out == nullptr) {
    q.out = dalloc<float>(2 * B);
    q.succ = dalloc<float>(2 * B);
  }
  grow(k.st, k.cap_st, 2 * N_SETS * L * H * n);
  grow(k.u, k.cap_u, H2 * n);
  if (k.z == nullptr) k.z = dalloc<float>(2 * n);
  if (!training) return;
  grow(k.din, k.cap_din, H * n);
  grow(k.dst, k.cap_dst, 2 * L * H * n);
  grow(k.rec, k.cap_rec, L * RPN * H * B);
  grow(k.a1, k.cap_a1, H * B);
  grow(k.ur, k.cap_ur, H2 * B);
  grow(k.dg, k.cap_dg, L * 4 * H * B);
  grow(k.du, k.cap_du, H2 * B);
  // weight-gradient partials: at most 64 slab rows (the matrices are large: P doubles per row)
  uint64_t rows = t->nbA < 64 ? t->nbA : 64;
  uint64_t chunk = (B + rows - 1) / rows;
  chunk = ((chunk + 31) / 32) * 32;
  k.wg_chunk = (uint32_t)chunk;
  k.wg_rows = (uint32_t)((B + chunk - 1) / chunk);
  traj_ensure_slabs(t, k.wg_rows, mod->P, t->nbB);
  traj_ensure_pvec(t, mod->P);
}

void stack_free(rl_traj *t) {
  SeqDev::Stack &k = t->seq.stack;
  for (float *p : {k.st, k.u, k.z, k.din, k.dst, k.rec, k.a1, k.ur, k.dg, k.du}) dfree(p);
  k = SeqDev::Stack{};
}

void launch_stack_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_succ, bool record,
                          const int32_t *d_skip) {
  ProfScope ps(traj->eng, RL_K_POLICY_FUSED);
  const StackNet net = stack_net(mod, mod->d_params);
  const StackWs ws = stack_ws(traj);
  const dim3 grid(cdiv_k(traj->d.n, SL)), blk(SL * SW);
#define FWD(CELL, REC)                                                                                              \
  hipLaunchKernelGGL((k_stack_forward<CELL, REC>), grid, blk, 0, traj->eng->stream, net, ws, traj->d, d_out, d_succ, \
                     d_skip)
  const bool lstm = mod->kind == RL_MODULE_LSTM_MLP;
  if (record) {
    if (lstm) FWD(LSTM, true);
    else FWD(GRU, true);
  } else {
    if (lstm) FWD(LSTM, false);
    else FWD(GRU, false);
  }
#undef FWD
  RL_HIP_CHECK(hipGetLastError());
}

void launch_stack_rollout(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  ProfScope ps(env->eng, RL_K_ROLLOUT);
  RL_REQUIRE(policy->out_dim == 2, "stacked recurrent rollouts: two-action policies");
  const StackNet net = stack_net(policy, policy->d_params);
  const StackWs ws = stack_ws(traj);
  const dim3 grid(cdiv_k(traj->d.n, SL)), blk(SL * SW);
  float *z = traj->seq.stack.z;
  const bool lstm = policy->kind == RL_MODULE_LSTM_MLP;
  launch_rollout_stepwise(env, traj, z, [&](uint32_t step) {
    const int first = step == 0 ? 1 : 0, parity = (int)(step & 1);
    // (the step before has been recorded by now: its successor codes are in the trajectory)
    const uint8_t *flag_prev = traj->d.flag + (size_t)(step == 0 ? 0 : step - 1) * traj->d.n;
    if (lstm)
      hipLaunchKernelGGL(k_stack_step<LSTM>, grid, blk, 0, env->eng->stream, net, ws, env->d_obs, flag_prev, first,
                         parity, z);
    else
      hipLaunchKernelGGL(k_stack_step<GRU>, grid, blk, 0, env->eng->stream, net, ws, env->d_obs, flag_prev, first,
                         parity, z);
  });
}

// dz [A][B] and the record of the last training forward -> vec[0..P): the reverse scan, then one weight-gradient
// launch per matrix (partials in slab rows), then the reduction
void launch_stack_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip) {
  rl_engine *e = traj->eng;
  const StackNet net = stack_net(mod, mod->d_params);
  const StackWs ws = stack_ws(traj);
  const SeqDev::Stack &k = traj->seq.stack;
  const uint64_t n = traj->d.n, T = traj->d.T, B = T * n;
  const int H = net.H, H2 = net.H2, A = net.A, L = net.L, G = mod->kind == RL_MODULE_LSTM_MLP ? 4 : 3;
  const uint32_t P = (uint32_t)mod->P;
  {
    ProfScope ps(e, RL_K_POLICY_FUSED);
    const dim3 grid(cdiv_k(n, SL)), blk(SL * SW);
    if (mod->kind == RL_MODULE_LSTM_MLP)
      hipLaunchKernelGGL(k_stack_backward<LSTM>, grid, blk, 0, e->stream, net, ws, traj->d, traj->dz, d_skip);
    else
      hipLaunchKernelGGL(k_stack_backward<GRU>, grid, blk, 0, e->stream, net, ws, traj->d, traj->dz, d_skip);
  }
  {
    ProfScope ps(e, RL_K_BACKWARD);
    auto wgrad = [&](const float *dY, int N, const float *X, size_t xs, int K, uint64_t offW, uint64_t offB) {
      launch_gen_wgrad_planes(traj, dY, (size_t)B, N, X, xs, K, (size_t)B, k.wg_chunk, k.wg_rows, P, (uint32_t)offW,
                              (uint32_t)offB, d_skip);
    };
    const size_t plane = (size_t)(T + 1) * n;
    for (int l = 0; l < L; ++l) {
      const int K = l == 0 ? net.D : H;
      const uint64_t oWih = net.off[l], oWhh = oWih + (uint64_t)G * H * K;
      // (layers without bias vectors: no bias columns in the flat vector)
      const uint64_t obih = mod->has_bias ? oWhh + (uint64_t)G * H * H : 0xFFFFFFFFull;
      const uint64_t obhh = mod->has_bias ? obih + (uint64_t)G * H : 0xFFFFFFFFull;
      const float *dg = k.dg + (size_t)l * 4 * H * B;
      const float *X = l == 0 ? traj->d.obs : k.rec + ((size_t)(l - 1) * RPN + RP_HOUT) * H * B;
      const float *hp = k.rec + ((size_t)l * RPN + RP_HPREV) * H * B;
      wgrad(dg, G * H, X, l == 0 ? plane : (size_t)B, K, oWih, obih);
      if (G == 4) {
        wgrad(dg, 4 * H, hp, (size_t)B, H, oWhh, obhh);
      } else {  // hidden side: rows [r; z] as on the input side, the n rows from the fourth plane
        wgrad(dg, 2 * H, hp, (size_t)B, H, oWhh, obhh);
        wgrad(dg + (size_t)3 * H * B, H, hp, (size_t)B, H, oWhh + (uint64_t)2 * H * H,
              mod->has_bias ? obhh + (uint64_t)2 * H : 0xFFFFFFFFull);
      }
    }
    const uint64_t oW1 = net.off[L], ob1 = oW1 + (uint64_t)H2 * H, oW2 = ob1 + H2, ob2 = oW2 + (uint64_t)A * H2;
    wgrad(k.du, H2, k.a1, (size_t)B, H, oW1, ob1);
    wgrad(traj->dz, A, k.ur, (size_t)B, H2, oW2, ob2);
    RL_HIP_CHECK(hipGetLastError());
  }
  launch_reduce(traj, P, true, false, k.wg_rows, 0);
}

void launch_stack_tangent(rl_traj *traj, const rl_mlp *mod, const float *d_tangent, const int32_t *d_skip) {
  ProfScope ps(traj->eng, RL_K_POLICY_FUSED);
  const StackNet net = stack_net(mod, mod->d_params), tv = stack_net(mod, d_tangent);
  const StackWs ws = stack_ws(traj);
  const dim3 grid(cdiv_k(traj->d.n, SL)), blk(SL * SW);
  if (mod->kind == RL_MODULE_LSTM_MLP)
    hipLaunchKernelGGL(k_stack_tangent<LSTM>, grid, blk, 0, traj->eng->stream, net, tv, ws, traj->d, traj->seq.out, d_skip);
  else
    hipLaunchKernelGGL(k_stack_tangent<GRU>, grid, blk, 0, traj->eng->stream, net, tv, ws, traj->d, traj->seq.out, d_skip);
  RL_HIP_CHECK(hipGetLastError());
}
