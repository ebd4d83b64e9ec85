// kernels_update.hip — TRPO and critic update kernels (first correct HIP path, "v1").
//
// Structure of one full-batch pass over the B = T*n samples of a trajectory:
//   (1) a SAMPLE-PARALLEL pass (one sample per lane, weights wave-uniform through scalar loads) computes
//       everything that is local to a sample — logits, log-probs, loss/KL/entropy terms and dL/dz — and
//       leaves dz[a][b] in HBM (8 B/sample);
//   (2) a HIDDEN-UNIT-PARALLEL backward pass (one hidden unit per lane, samples wave-uniform through scalar
//       loads) turns dz into parameter gradients: every lane owns the D+1+A gradient entries of its hidden
//       unit in registers, so the sum over samples needs no cross-lane traffic at all;
//   (3) per-block partials land in a slab and a deterministic two-level reduction produces the <= 4 KiB
//       vector that is (optionally) all-reduced over xGMI and consumed by a single-workgroup kernel
//       (CG step / line-search bookkeeping / Adam), so a whole update needs no host round trip.
//
// Reference semantics followed: src/torch/agents/policies/trpo.rs:97-164,
// src/torch/optimizers/conjugate_gradient.rs:115-403, src/torch/distributions/categorical.rs:29-77,
// src/torch/agents/critics/opt.rs:100-126, src/torch/optimizers/coptimizer.rs:13-26.
#include "bf16_tile.hpp"
#include "comm_ipc.hpp"
#include "device_fns.hpp"
#include "kernels.hpp"
#include "policy_terms.hpp"

#include <cfloat>
#include <cmath>

static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

// ---------------------------------------------------------------- (1) sample-parallel policy pass
template <int D, int MODE>
__global__ void __launch_bounds__(256) k_policy_pass(TrajDev tr, const float *__restrict__ params,
                                                     const float *__restrict__ tangent, int H,
                                                     float *__restrict__ lp0, float *__restrict__ dz,
                                                     double *__restrict__ slabB, float inv_B,
                                                     const int32_t *__restrict__ skip, float clip_lo,
                                                     float clip_hi) {
  constexpr int A = 2;
  __shared__ double red[256];
  if (skip != nullptr && *skip != 0) return;
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  // per-sample terms are f32 (as the reference's Kind::Float tensors); their sum over samples is carried
  // in f64 so that the mean is rounded once (what a pairwise/blocked torch reduction approximates)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    float x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = tr.obs[d * plane + b];
    int act = tr.action[b];
    if (MODE == PASS_JVP) {
      const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H,
                               *__restrict__ b2 = W2 + A * H;
      const float *__restrict__ tW1 = tangent, *__restrict__ tb1 = tW1 + H * D, *__restrict__ tW2 = tb1 + H,
                               *__restrict__ tb2 = tW2 + A * H;
      float z[A], dzt[A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        z[a] = b2[a];
        dzt[a] = tb2[a];
      }
      for (int j = 0; j < H; ++j) {
        float pre = b1[j], tpre = tb1[j];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          pre = __builtin_fmaf(x[k], W1[j * D + k], pre);
          tpre = __builtin_fmaf(x[k], tW1[j * D + k], tpre);
        }
        float h = pre > 0.0f ? pre : 0.0f;
        float dh = pre > 0.0f ? tpre : 0.0f;
#pragma unroll
        for (int a = 0; a < A; ++a) {
          z[a] = __builtin_fmaf(h, W2[a * H + j], z[a]);
          dzt[a] = __builtin_fmaf(dh, W2[a * H + j], dzt[a]);
          dzt[a] = __builtin_fmaf(h, tW2[a * H + j], dzt[a]);
        }
      }
      policy_sample_terms<MODE>(z, dzt, act, 0.0f, b, B, lp0, dz, inv_B, clip_lo, clip_hi, s0, s1, s2);
    } else {
      float z[A];
      mlp_forward_lane<D, A>(params, H, x, z);
      const float none[A] = {0.0f, 0.0f};
      policy_sample_terms<MODE>(z, none, act, tr.adv[b], b, B, lp0, dz, inv_B, clip_lo, clip_hi, s0, s1, s2);
    }
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

// critic: y = V(s); d = y - target; dz = 2 d / B; loss partial = sum d^2
// (mse_loss(Mean) + backward, src/torch/agents/critics/opt.rs:109-115)
template <int D>
__global__ void __launch_bounds__(256) k_critic_fwd(TrajDev tr, const float *__restrict__ params, int H,
                                                    float *__restrict__ dz, double *__restrict__ slabB,
                                                    float two_over_B) {
  __shared__ double red[256];
  const size_t B = (size_t)tr.T * tr.n;
  const size_t plane = (size_t)(tr.T + 1) * tr.n;
  double s0 = 0.0;
  for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (size_t)gridDim.x * 256) {
    float x[D], z[1];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = tr.obs[d * plane + b];
    mlp_forward_lane<D, 1>(params, H, x, z);
    float d = z[0] - tr.tgt[b];
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

// ---------------------------------------------------------------- (2) hidden-unit-parallel backward
// Lane j owns hidden unit j: its W1 row, b1, W2 column live in VGPRs; the samples of the block's chunk are
// streamed through wave-uniform (scalar) loads, G at a time.  Per sample and lane:
//   pre = b1 + x.W1[j] (recomputed, D fma), dh = sum_a dz_a W2[a][j] masked by pre > 0,
//   gW1[j][:] += dh x, gb1[j] += dh, gW2[:][j] += dz relu(pre), gb2 += dz.
template <int D, int A>
__global__ void __launch_bounds__(128) k_mlp_backward(const float *__restrict__ params, int H,
                                                      const float *__restrict__ obs, size_t plane,
                                                      const float *__restrict__ dz, size_t B, uint32_t chunk,
                                                      double *__restrict__ slabA, uint32_t P,
                                                      const int32_t *__restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  constexpr int G = 8;
  const int j = threadIdx.x;
  const bool active = j < H;
  const int jj = active ? j : 0;
  const float *__restrict__ W1 = params, *__restrict__ b1 = W1 + H * D, *__restrict__ W2 = b1 + H;
  float w1[D], w2[A];
#pragma unroll
  for (int k = 0; k < D; ++k) w1[k] = W1[jj * D + k];
  const float bias = b1[jj];
#pragma unroll
  for (int a = 0; a < A; ++a) w2[a] = W2[a * H + jj];
  // Two-level accumulation: products and the sum over a group of G samples are f32 (the reference's tensors
  // are Kind::Float); groups are added into f64 accumulators so that the sum over the chunk — and, in the
  // reduction kernel, over all workgroups — is rounded to f32 only once.  TRPO's 10-iteration CG on the Fisher
  // matrix is ill-conditioned; f32 blocked sums over thousands of samples are not accurate enough for it.
  double Gw1[D], Gw2[A], Gb2[A], Gb1 = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) Gw1[k] = 0.0;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    Gw2[a] = 0.0;
    Gb2[a] = 0.0;
  }
  const size_t s_begin = (size_t)blockIdx.x * chunk;
  const size_t s_end = s_begin + chunk < B ? s_begin + chunk : B;
  size_t s = s_begin;
  for (; s + G <= s_end; s += G) {
    float xs[D][G], ds[A][G];
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
      for (int g = 0; g < G; ++g) xs[k][g] = obs[k * plane + s + g];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
      for (int g = 0; g < G; ++g) ds[a][g] = dz[(size_t)a * B + s + g];
    float gw1[D], gw2[A], gb2[A], gb1 = 0.0f;
#pragma unroll
    for (int k = 0; k < D; ++k) gw1[k] = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      gw2[a] = 0.0f;
      gb2[a] = 0.0f;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float pre = bias;
#pragma unroll
      for (int k = 0; k < D; ++k) pre = __builtin_fmaf(xs[k][g], w1[k], pre);
      float h = pre > 0.0f ? pre : 0.0f;
      float dh = 0.0f;
#pragma unroll
      for (int a = 0; a < A; ++a) dh = __builtin_fmaf(ds[a][g], w2[a], dh);
      float m = pre > 0.0f ? dh : 0.0f;
      gb1 = gb1 + m;
#pragma unroll
      for (int k = 0; k < D; ++k) gw1[k] = gw1[k] + m * xs[k][g];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        gw2[a] = gw2[a] + ds[a][g] * h;
        gb2[a] = gb2[a] + ds[a][g];
      }
    }
    Gb1 += (double)gb1;
#pragma unroll
    for (int k = 0; k < D; ++k) Gw1[k] += (double)gw1[k];
#pragma unroll
    for (int a = 0; a < A; ++a) {
      Gw2[a] += (double)gw2[a];
      Gb2[a] += (double)gb2[a];
    }
  }
  for (; s < s_end; ++s) {
    float pre = bias;
    float xv[D], dv[A];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      xv[k] = obs[k * plane + s];
      pre = __builtin_fmaf(xv[k], w1[k], pre);
    }
    float h = pre > 0.0f ? pre : 0.0f;
    float dh = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      dv[a] = dz[(size_t)a * B + s];
      dh = __builtin_fmaf(dv[a], w2[a], dh);
    }
    float m = pre > 0.0f ? dh : 0.0f;
    Gb1 += (double)m;
#pragma unroll
    for (int k = 0; k < D; ++k) Gw1[k] += (double)(m * xv[k]);
#pragma unroll
    for (int a = 0; a < A; ++a) {
      Gw2[a] += (double)(dv[a] * h);
      Gb2[a] += (double)dv[a];
    }
  }
  double *__restrict__ row = slabA + (size_t)blockIdx.x * P;
  if (active) {
#pragma unroll
    for (int k = 0; k < D; ++k) row[j * D + k] = Gw1[k];
    row[H * D + j] = Gb1;
#pragma unroll
    for (int a = 0; a < A; ++a) row[H * D + H + a * H + j] = Gw2[a];
    if (j == 0) {
#pragma unroll
      for (int a = 0; a < A; ++a) row[H * D + H + A * H + a] = Gb2[a];
    }
  }
}

// ---------------------------------------------------------------- (3) deterministic slab reduction
// Workgroup = 16 waves for 64 consecutive vector entries: wave w sums slab rows w, w+16, ... in order, the
// 16 partials are combined in wave order; f64 throughout, rounded to f32 once.  Rows are 512-B coalesced reads.
// sum of the slab rows r = w, w + 16, w + 32, ... of one column: the loads of up to 16 rows are issued before the first
// add (one memory round trip for <= 256 rows — the launch is latency-bound), the additions run in row order
__device__ __forceinline__ double column_partial(const double *__restrict__ slab, uint32_t nb, uint32_t stride,
                                                 uint32_t col, int w) {
  double acc = 0.0;
  for (uint32_t r0 = w; r0 < nb; r0 += 16 * 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const uint32_t r = r0 + 16 * u;
      v[u] = r < nb ? slab[(size_t)r * stride + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = acc + v[u];
  }
  return acc;
}

__global__ void __launch_bounds__(1024) k_reduce(const double *__restrict__ slabA, uint32_t nbA, uint32_t P,
                                                 const double *__restrict__ slabB, uint32_t nbB,
                                                 float *__restrict__ vec, int useA, int useB) {
  __shared__ double part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 64 + lane;
  double acc = 0.0;
  if (p < P) {
    if (useA) acc = column_partial(slabA, nbA, P, p, w);
  } else if (p < P + 4) {
    if (useB) acc = column_partial(slabB, nbB, 4, p - P, w);
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0 && p < P + 4) {
    bool write = p < P ? useA != 0 : useB != 0;
    if (write) {
      double t = part[0][lane];
#pragma unroll
      for (int k = 1; k < 16; ++k) t = t + part[k][lane];
      vec[p] = (float)t;
    }
  }
}

// k_reduce with narrower workgroups (round 6; k_reduce_adam_narrow below says why): CW columns per workgroup, the 64 / CW
// lane groups of a wave on different rows
template <int CW>
__global__ void __launch_bounds__(1024) k_reduce_narrow(const double *__restrict__ slabA, uint32_t nbA, uint32_t P,
                                                        const double *__restrict__ slabB, uint32_t nbB,
                                                        float *__restrict__ vec, int useA, int useB) {
  constexpr int GPW = 64 / CW, RG = 16 * GPW, U = 256 / RG;
  __shared__ double part[RG][CW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & (CW - 1), rg = w * GPW + lane / CW;
  const uint32_t p = blockIdx.x * CW + (uint32_t)c;
  auto partial = [&](const double *__restrict__ slab, uint32_t nb, uint32_t stride, uint32_t col) {
    double acc = 0.0;
    for (uint32_t r0 = (uint32_t)rg; r0 < nb; r0 += RG * U) {
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t r = r0 + RG * u;
        x[u] = r < nb ? slab[(size_t)r * stride + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = acc + x[u];
    }
    return acc;
  };
  double acc = 0.0;
  if (p < P) {
    if (useA) acc = partial(slabA, nbA, P, p);
  } else if (p < P + 4) {
    if (useB) acc = partial(slabB, nbB, 4, p - P);
  }
  part[rg][c] = acc;
  __syncthreads();
  if (threadIdx.x >= CW || p >= P + 4) return;
  if (p < P ? useA != 0 : useB != 0) {
    double t = part[0][c];
#pragma unroll
    for (int k = 1; k < RG; ++k) t = t + part[k][c];
    vec[p] = (float)t;
  }
}

// ---------------------------------------------------------------- single-workgroup bookkeeping kernels
constexpr int SB = 256;  // P <= 1026: four entries per thread, 8 barrier rounds per block-wide sum

// after the gradient reduction: x = 0; r = p = g; rr = g.g   (solve_conjugate_gradient prologue,
// conjugate_gradient.rs:377-384) + initial loss / entropy scalars
__global__ void __launch_bounds__(SB) k_trpo_begin(const float *__restrict__ vec, uint32_t P, double inv_B,
                                                   float *__restrict__ x, float *__restrict__ r,
                                                   float *__restrict__ p, TrpoStateDev *st) {
  __shared__ double red[SB];
  double part = 0.0;
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    float g = vec[i];
    x[i] = 0.0f;
    r[i] = g;
    p[i] = g;
    part += (double)(g * g);
  }
  float rr = (float)block_sum<SB>(part, red);  // Tensor::dot: f32 products, sum rounded once
  if (threadIdx.x == 0) {
    st->rr = rr;
    st->cg_done = 0;
    st->cg_iters = 0;
    st->loss0 = (float)(-((double)vec[P] * inv_B));
    st->entropy = (float)((double)vec[P + 1] * inv_B);
    st->ls_accepted = 0;
    st->ls_index = -1;
    st->ls_ratio = 0.0;
    st->status = RL_OPT_OK;
    st->prev_saved = 0;
  }
}

// one CG iteration body after z = A p is available (conjugate_gradient.rs:386-401)
__global__ void __launch_bounds__(SB) k_cg_step(float *__restrict__ vec, uint32_t P, float reg, float tol,
                                                float *__restrict__ x, float *__restrict__ r,
                                                float *__restrict__ p, TrpoStateDev *st) {
  __shared__ double red[SB];
  if (st->cg_done) return;
  double part = 0.0;
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    float z = vec[i] + reg * p[i];
    vec[i] = z;
    part += (double)(p[i] * z);
  }
  float pz = (float)block_sum<SB>(part, red);
  float rr = st->rr;
  float alpha = rr / pz;
  float nalpha = -alpha;
  part = 0.0;
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    x[i] = x[i] + alpha * p[i];
    float ri = r[i] + nalpha * vec[i];
    r[i] = ri;
    part += (double)(ri * ri);
  }
  float new_rr = (float)block_sum<SB>(part, red);
  bool done = (double)new_rr < (double)tol;
  if (!done) {
    float mu = new_rr / rr;
    for (uint32_t i = threadIdx.x; i < P; i += SB) {
      float pi = p[i] * mu;
      p[i] = pi + r[i];
    }
  }
  if (threadIdx.x == 0) {
    st->cg_iters += 1;
    if (done) st->cg_done = 1;
    else st->rr = new_rr;
  }
}

// step_dir.nan_to_num_(0.0, None, None) (conjugate_gradient.rs:152)
__global__ void __launch_bounds__(SB) k_cg_finish(float *__restrict__ x, uint32_t P) {
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    float v = x[i];
    if (v != v) v = 0.0f;
    else if (v > FLT_MAX) v = FLT_MAX;
    else if (v < -FLT_MAX) v = -FLT_MAX;
    x[i] = v;
  }
}

// step size + descent step + line-search prologue (conjugate_gradient.rs:155-168, 187-199)
__global__ void __launch_bounds__(SB) k_step_size(const float *__restrict__ vec, uint32_t P, float reg,
                                                  double max_kl, const float *__restrict__ x,
                                                  float *__restrict__ descent, const float *__restrict__ params,
                                                  float *__restrict__ prev, TrpoStateDev *st) {
  __shared__ double red[SB];
  __shared__ float ss_shared;
  double part = 0.0;
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    float hx = vec[i] + reg * x[i];
    part += (double)(x[i] * hx);
  }
  float xhx = (float)block_sum<SB>(part, red);
  if (threadIdx.x == 0) {
    double step_size = sqrt(1.0 / ((double)xhx + 1e-8) * max_kl * 2.0);
    if (step_size != step_size) step_size = 1.0;
    st->step_size = step_size;
    st->ls_loss = st->loss0;
    st->ls_kl = __builtin_inff();
    st->prev_saved = 1;
    ss_shared = (float)step_size;
  }
  __syncthreads();
  float ss = ss_shared;
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    descent[i] = ss * x[i];
    prev[i] = params[i];
  }
}

// A failed collective (the sticky error word of the peer-mailbox transport, comm_ipc.hpp; NULL without that transport)
// has left a vector of LOCAL sums behind: nothing computed from it may reach the parameters or the optimiser state.  The
// kernels that write them test the word — a candidate is not set, the line search rolls back to the saved parameters,
// an Adam step is skipped — and the host raises RL_ERR_COMM at its next synchronising call.
__device__ __forceinline__ bool comm_failed(const int32_t *comm_err) {
  return comm_err != nullptr && __hip_atomic_load(comm_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// param.copy_(prev_param - ratio * step) (conjugate_gradient.rs:204-213)
// `wimg` (may be NULL): the module's weight image, kept current beside the parameters (bf16_tile.hpp)
__global__ void __launch_bounds__(SB) k_ls_set_params(float *__restrict__ params, const float *__restrict__ prev,
                                                      const float *__restrict__ descent, uint32_t P, float ratio,
                                                      const TrpoStateDev *st, const int32_t *comm_err,
                                                      uint32_t *__restrict__ wimg, int A) {
  if (st->ls_accepted || comm_failed(comm_err)) return;
  for (uint32_t i = threadIdx.x; i < P; i += SB) {
    const float w = prev[i] - ratio * descent[i];
    params[i] = w;
    if (wimg) bt::wimg_store_param(wimg, i, w, A);
  }
}

// the weight image from scratch: one lane per parameter (the first fused launch of a C-ABI call, wimg_ensure)
__global__ void __launch_bounds__(SB) k_wimg_build(const float *__restrict__ params, uint32_t P,
                                                   uint32_t *__restrict__ wimg, int A) {
  const uint32_t i = blockIdx.x * SB + threadIdx.x;
  if (i < P) bt::wimg_store_param(wimg, i, params[i], A);
}

// acceptance test of one backtrack (conjugate_gradient.rs:215-223)
__global__ void k_ls_check(const float *__restrict__ vec, uint32_t P, double inv_B, int index, double ratio,
                           double max_kl, TrpoStateDev *st) {
  if (threadIdx.x != 0 || st->ls_accepted) return;
  float loss = (float)(-((double)vec[P] * inv_B));
  float kl = (float)((double)vec[P + 1] * inv_B);
  st->ls_loss = loss;
  st->ls_kl = kl;
  if ((double)loss < (double)st->loss0 && (double)kl <= max_kl) {
    st->ls_accepted = 1;
    st->ls_index = index;
    st->ls_ratio = ratio;
  }
}

// final classification + rollback (conjugate_gradient.rs:228-253)
__global__ void __launch_bounds__(SB) k_ls_finalize(float *__restrict__ params, const float *__restrict__ prev,
                                                    uint32_t P, double max_kl, int accept_violation,
                                                    TrpoStateDev *st, const int32_t *comm_err,
                                                    const uint32_t *__restrict__ veto) {
  __shared__ int status_shared;
  // (uniform) the update is void — a failed exchange, or the range guard's veto (bf16_tile.hpp range_guard: the passes this
  // search was built from are not to be used): back to the parameters it started from, if it got that far
  if (comm_failed(comm_err) || (veto != nullptr && *veto != 0u)) {
    if (st->prev_saved)
      for (uint32_t i = threadIdx.x; i < P; i += SB) params[i] = prev[i];
    return;
  }
  if (threadIdx.x == 0) {
    double loss = (double)st->ls_loss, cval = (double)st->ls_kl, initial = (double)st->loss0;
    int status;
    if (loss != loss) status = RL_OPT_NAN_LOSS;
    else if (cval != cval) status = RL_OPT_NAN_CONSTRAINT;
    else if (loss >= initial) status = RL_OPT_LOSS_NOT_IMPROVING;
    else if (cval >= max_kl && !accept_violation) status = RL_OPT_CONSTRAINT_VIOLATED;
    else status = RL_OPT_OK;
    st->status = status;
    status_shared = status;
  }
  __syncthreads();
  if (status_shared != RL_OPT_OK)
    for (uint32_t i = threadIdx.x; i < P; i += SB) params[i] = prev[i];
}

// torch::optim::Adam::step of libtorch 1.12 (third party; COptimizer::adam, coptimizer.rs:158-167).  The bias
// corrections come from the host, which tracks the step count (rl_adam.host_step): neg_step_size = -(float)(lr /
// (1 - beta1^step)), sqrt_bc2 = (float)sqrt(1 - beta2^step) — the same numbers for this kernel and for k_reduce_adam.
__global__ void __launch_bounds__(SB) k_adam_step(float *__restrict__ params, const float *__restrict__ grad,
                                                  float *__restrict__ m, float *__restrict__ v, uint64_t *step_ptr,
                                                  uint32_t P, uint64_t step, float neg_step_size, float sqrt_bc2,
                                                  double beta1, double beta2, double eps, double weight_decay,
                                                  const float *__restrict__ loss_sum, double inv_B,
                                                  float *__restrict__ loss_out, const int32_t *comm_err,
                                                  uint32_t *__restrict__ wimg, int A,
                                                  const uint32_t *__restrict__ veto) {
  if (comm_failed(comm_err)) return;  // `grad` holds local sums: no step, no step count
  if (veto != nullptr && *veto != 0u) return;  // the range guard refused the pass `grad` comes from: nothing is applied
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *step_ptr = step;
    if (loss_out) *loss_out = (float)((double)(*loss_sum) * inv_B);
  }
  float b1 = (float)beta1, b2 = (float)beta2;
  float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  float epsf = (float)eps;
  float wd = (float)weight_decay;
  for (uint32_t i = blockIdx.x * SB + threadIdx.x; i < P; i += gridDim.x * SB) {  // (elementwise: any grid)
    float g = grad[i];
    if (weight_decay != 0.0) g = g + wd * params[i];
    float mi = m[i] * b1 + omb1 * g;
    float vi = v[i] * b2 + omb2 * g * g;
    m[i] = mi;
    v[i] = vi;
    float denom = __fsqrt_rn(vi) / sqrt_bc2 + epsf;
    const float w = params[i] + (neg_step_size * mi) / denom;
    params[i] = w;
    if (wimg) bt::wimg_store_param(wimg, i, w, A);  // (the module's weight image stays current, bf16_tile.hpp)
  }
}

// k_reduce followed by the Adam update of the 64 entries each workgroup has just reduced (Adam is elementwise, so
// no second launch is needed when no all-reduce sits between the two: single-rank runs).  Same arithmetic as
// k_reduce + k_adam_step; `step` is the 1-based step index, tracked by the host.
// XCHG: with several ranks on the peer-mailbox transport, wave 0 exchanges the 64 reduced columns of the workgroup (one
// chunk of the collective, comm_ipc.hpp) between the reduction and the optimiser step — the multi-rank critic loop stays
// at two launches per step
template <bool XCHG>
__global__ void __launch_bounds__(1024) k_reduce_adam(const double *__restrict__ slabA, uint32_t nbA, uint32_t P,
                                                      const double *__restrict__ slabB, uint32_t nbB,
                                                      float *__restrict__ vec, float *__restrict__ params,
                                                      float *__restrict__ m, float *__restrict__ v,
                                                      uint64_t *step_ptr, uint64_t step, float neg_step_size,
                                                      float sqrt_bc2, double beta1, double beta2, double eps,
                                                      double weight_decay, double inv_B,
                                                      float *__restrict__ loss_out, IpcPeers peers,
                                                      uint32_t *__restrict__ wimg, int A,
                                                      const uint32_t *__restrict__ veto) {
  __shared__ double part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 64 + lane;
  // the optimiser state of this column is requested first, so that it arrives under the slab loads — and the range
  // guard's veto word with it (set by the fused launch this one reduces, or an earlier one of the call: the sums are then
  // not to be applied; bf16_tile.hpp range_guard)
  const uint32_t vetoed = veto != nullptr ? *veto : 0u;
  float p_old = 0.0f, m_old = 0.0f, v_old = 0.0f;
  if (w == 0 && p < P) {
    p_old = params[p];
    m_old = m[p];
    v_old = v[p];
  }
  double acc = 0.0;
  if (p < P) acc = column_partial(slabA, nbA, P, p, w);
  else if (p < P + 4) acc = column_partial(slabB, nbB, 4, p - P, w);
  part[w][lane] = acc;
  __syncthreads();
  if (w != 0) return;
  double t = part[0][lane];
#pragma unroll
  for (int k = 1; k < 16; ++k) t = t + part[k][lane];
  float gsum = (float)t;
  if (XCHG) {  // all 64 lanes; a failed exchange (sticky error word, comm_ipc.hpp) leaves vec, params and moments alone
    float total;
    if (!ipc_exchange_chunk(peers, blockIdx.x, (uint32_t)lane, p < P + 4 ? gsum : 0.0f, total)) return;
    gsum = total;
  }
  if (p >= P + 4) return;
  vec[p] = gsum;
  if (p >= P) {
    if (p == P && loss_out) *loss_out = (float)((double)gsum * inv_B);
    return;
  }
  if (vetoed != 0u) return;  // parameters, moments and step count stay as they are
  if (p == 0) *step_ptr = step;
  // neg_step_size = -(float)(lr / (1 - beta1^step)), sqrt_bc2 = (float)sqrt(1 - beta2^step): the host knows the step
  const float b1 = (float)beta1, b2 = (float)beta2;
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  const float epsf = (float)eps;
  float g = gsum;
  if (weight_decay != 0.0) g = g + (float)weight_decay * p_old;
  const float mi = m_old * b1 + omb1 * g;
  const float vi = v_old * b2 + omb2 * g * g;
  m[p] = mi;
  v[p] = vi;
  const float denom = __fsqrt_rn(vi) / sqrt_bc2 + epsf;
  const float w_new = p_old + (neg_step_size * mi) / denom;
  params[p] = w_new;
  // the next fused launch reads the weights as piece fragments: this lane owns parameter p, it also owns p's pieces in
  // the module's weight image (bf16_tile.hpp; NULL when this call has built none)
  if (wimg) bt::wimg_store_param(wimg, p, w_new, A);
}

// The same launch with NARROWER workgroups (round 6): CW = 16 or 32 columns per workgroup instead of 64, the 64 / CW lane
// groups of a wave taking different rows — four (two) times as many workgroups pull the slab, each a quarter (half) of
// the bytes, and a lane has 4 (8) loads in flight instead of 16.  A column's partial sums are per row group r, r + RG, ...
// (RG = 16 x 64 / CW groups) in row order, combined in group order: deterministic, another order than the wide form.
// (No mailbox exchange here: its chunks are 64 columns.)
template <int CW>
__global__ void __launch_bounds__(1024) k_reduce_adam_narrow(const double *__restrict__ slabA, uint32_t nbA, uint32_t P,
                                                             const double *__restrict__ slabB, uint32_t nbB,
                                                             float *__restrict__ vec, float *__restrict__ params,
                                                             float *__restrict__ m, float *__restrict__ v,
                                                             uint64_t *step_ptr, uint64_t step, float neg_step_size,
                                                             float sqrt_bc2, double beta1, double beta2, double eps,
                                                             double weight_decay, double inv_B,
                                                             float *__restrict__ loss_out, uint32_t *__restrict__ wimg,
                                                             int A, const uint32_t *__restrict__ veto) {
  constexpr int GPW = 64 / CW, RG = 16 * GPW;  // row groups per wave, per workgroup
  __shared__ double part[RG][CW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane & (CW - 1), rg = w * GPW + lane / CW;
  const uint32_t p = blockIdx.x * CW + (uint32_t)c;
  const uint32_t vetoed = veto != nullptr ? *veto : 0u;  // (the range guard's veto: k_reduce_adam)
  float p_old = 0.0f, m_old = 0.0f, v_old = 0.0f;
  if (threadIdx.x < CW && p < P) {  // the optimiser state of this column arrives under the slab loads
    p_old = params[p];
    m_old = m[p];
    v_old = v[p];
  }
  auto partial = [&](const double *__restrict__ slab, uint32_t nb, uint32_t stride, uint32_t col) {
    constexpr int U = 256 / RG;  // loads in flight per lane: one memory round trip for <= 256 rows
    double acc = 0.0;
    for (uint32_t r0 = (uint32_t)rg; r0 < nb; r0 += RG * U) {
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t r = r0 + RG * u;
        x[u] = r < nb ? slab[(size_t)r * stride + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = acc + x[u];
    }
    return acc;
  };
  double acc = 0.0;
  if (p < P) acc = partial(slabA, nbA, P, p);
  else if (p < P + 4) acc = partial(slabB, nbB, 4, p - P);
  part[rg][c] = acc;
  __syncthreads();
  if (threadIdx.x >= CW || p >= P + 4) return;
  double t = part[0][c];
#pragma unroll
  for (int k = 1; k < RG; ++k) t = t + part[k][c];
  const float gsum = (float)t;
  vec[p] = gsum;
  if (p >= P) {
    if (p == P && loss_out) *loss_out = (float)((double)gsum * inv_B);
    return;
  }
  if (vetoed != 0u) return;
  if (p == 0) *step_ptr = step;
  const float b1 = (float)beta1, b2 = (float)beta2;
  const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  const float epsf = (float)eps;
  float g = gsum;
  if (weight_decay != 0.0) g = g + (float)weight_decay * p_old;
  const float mi = m_old * b1 + omb1 * g;
  const float vi = v_old * b2 + omb2 * g * g;
  m[p] = mi;
  v[p] = vi;
  const float denom = __fsqrt_rn(vi) / sqrt_bc2 + epsf;
  const float w_new = p_old + (neg_step_size * mi) / denom;
  params[p] = w_new;
  if (wimg) bt::wimg_store_param(wimg, p, w_new, A);
}

// ---------------------------------------------------------------- launchers
// the word a failed mailbox exchange sets (NULL when that transport is not in use)
static const int32_t *comm_err_word(const rl_engine *e) { return e->ipc_active ? e->ipc_err : (const int32_t *)nullptr; }
// the range guard's veto word for updates of module `m` from passes over `traj` (bf16_tile.hpp range_guard: the fused
// critic step guards one-output modules — the critic chain — everything else is the policy chain: policy passes, the DQN
// gradient); only fused launches ever set it
static const uint32_t *veto_word(const rl_traj *traj, const rl_mlp *m) {
  if (traj->d.range == nullptr) return nullptr;
  return traj->d.range + RL_RANGE_WORDS + (m->out_dim == 1 ? RL_GUARD_CRITIC : RL_GUARD_POLICY);
}

void launch_policy_pass(rl_traj *traj, const rl_mlp *policy, int mode, const float *d_tangent, uint64_t B_total,
                        const int32_t *d_skip, float clip_lo, float clip_hi) {
  ProfScope ps(traj->eng, RL_K_POLICY_PASS);
  RL_REQUIRE(policy->out_dim == 2, "policy pass: only 2-action categorical policies are built");
  if (policy->general) return launch_gen_policy_pass(traj, policy, mode, d_tangent, B_total, d_skip, clip_lo, clip_hi);
  float inv_B = 1.0f / (float)B_total;
  dim3 g(traj->nbB), b(256);
  hipStream_t s = traj->eng->stream;
  int H = (int)policy->hidden;
#define PASS(DD, MM)                                                                                           \
  hipLaunchKernelGGL((k_policy_pass<DD, MM>), g, b, 0, s, traj->d, policy->d_params, d_tangent, H, traj->lp0, \
                     traj->dz, traj->slabB, inv_B, d_skip, clip_lo, clip_hi)
  if (mode == PASS_DQN) inv_B = 2.0f / (float)B_total;
  if (traj->d.D == 5) {
    if (mode == PASS_INIT) PASS(5, PASS_INIT);
    else if (mode == PASS_EVAL) PASS(5, PASS_EVAL);
    else if (mode == PASS_DQN) PASS(5, PASS_DQN);
    else if (mode == PASS_PPO) PASS(5, PASS_PPO);
    else PASS(5, PASS_JVP);
  } else {
    if (mode == PASS_INIT) PASS(4, PASS_INIT);
    else if (mode == PASS_EVAL) PASS(4, PASS_EVAL);
    else if (mode == PASS_DQN) PASS(4, PASS_DQN);
    else if (mode == PASS_PPO) PASS(4, PASS_PPO);
    else PASS(4, PASS_JVP);
  }
#undef PASS
}

void launch_critic_fwd(rl_traj *traj, const rl_mlp *critic, uint64_t B_total) {
  ProfScope ps(traj->eng, RL_K_CRITIC_FWD);
  if (critic->general) return launch_gen_critic_fwd(traj, critic, B_total);
  float two_over_B = 2.0f / (float)B_total;
  dim3 g(traj->nbB), b(256);
  if (traj->d.D == 5)
    hipLaunchKernelGGL(k_critic_fwd<5>, g, b, 0, traj->eng->stream, traj->d, critic->d_params, (int)critic->hidden,
                       traj->dz, traj->slabB, two_over_B);
  else
    hipLaunchKernelGGL(k_critic_fwd<4>, g, b, 0, traj->eng->stream, traj->d, critic->d_params, (int)critic->hidden,
                       traj->dz, traj->slabB, two_over_B);
}

void launch_mlp_backward(rl_traj *traj, const rl_mlp *mlp, const int32_t *d_skip) {
  ProfScope ps(traj->eng, RL_K_BACKWARD);
  if (mlp->general) return launch_gen_backward(traj, mlp, d_skip);
  RL_REQUIRE(mlp->hidden <= 128, "backward kernel v1 supports hidden <= 128");
  dim3 g(traj->nbA), b(128);
  size_t plane = (size_t)(traj->d.T + 1) * traj->d.n;
  hipStream_t s = traj->eng->stream;
  uint32_t P = (uint32_t)mlp->P;
#define BWD(DD, AA)                                                                                            \
  hipLaunchKernelGGL((k_mlp_backward<DD, AA>), g, b, 0, s, mlp->d_params, (int)mlp->hidden, traj->d.obs, plane, \
                     traj->dz, (size_t)traj->B, traj->bwd_chunk, traj->slabA, P, d_skip)
  if (traj->d.D == 5 && mlp->out_dim == 2) BWD(5, 2);
  else if (traj->d.D == 5 && mlp->out_dim == 1) BWD(5, 1);
  else if (traj->d.D == 4 && mlp->out_dim == 2) BWD(4, 2);
  else if (traj->d.D == 4 && mlp->out_dim == 1) BWD(4, 1);
  else throw RlError(RL_ERR_UNSUPPORTED, "backward: unsupported (obs_dim, out_dim)");
#undef BWD
}

// columns of the reduced vector per workgroup of the slab reductions: 16 (57 - 65 workgroups pull the slab, a lane has
// four loads in flight; round 6: 6.4 -> 4.5 us per reduce + Adam launch, scripts/critic_only.py) — or 64, the form of
// rounds 1-5, with RL_REDUCE_WIDTH=64 (A/B runs).  The mailbox exchange keeps the wide form: its chunks are 64 columns.
// (vectors of more than 2,048 entries — the recurrent chains, wide general MLPs — are bandwidth-, not latency-bound and
// keep the wide form)
static int reduce_width(uint32_t P) {
  static const int w = [] {
    const char *e = std::getenv("RL_REDUCE_WIDTH");
    return e != nullptr && std::atoi(e) == 64 ? 64 : 16;
  }();
  return P + 4 > 2048 ? 64 : w;
}

void launch_reduce(rl_traj *traj, uint32_t P, bool useA, bool useB, uint32_t rowsA, uint32_t rowsB) {
  ProfScope ps(traj->eng, RL_K_REDUCE);
  if (reduce_width(P) == 64)
    hipLaunchKernelGGL(k_reduce, dim3(cdiv(P + 4, 64)), dim3(1024), 0, traj->eng->stream, traj->slabA, rowsA, P,
                       traj->slabB, rowsB, traj->vec, useA ? 1 : 0, useB ? 1 : 0);
  else
    hipLaunchKernelGGL(k_reduce_narrow<16>, dim3(cdiv(P + 4, 16)), dim3(1024), 0, traj->eng->stream, traj->slabA, rowsA,
                       P, traj->slabB, rowsB, traj->vec, useA ? 1 : 0, useB ? 1 : 0);
}

void launch_trpo_begin(rl_traj *traj, rl_mlp *policy, uint64_t B_total) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_trpo_begin, dim3(1), dim3(SB), 0, traj->eng->stream, traj->vec, (uint32_t)policy->P,
                     1.0 / (double)B_total, traj->cg_x, traj->cg_r, traj->cg_p, traj->trpo);
}

void launch_cg_step(rl_traj *traj, uint32_t P, float reg, float tol) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_cg_step, dim3(1), dim3(SB), 0, traj->eng->stream, traj->vec, P, reg, tol, traj->cg_x,
                     traj->cg_r, traj->cg_p, traj->trpo);
}

void launch_cg_finish(rl_traj *traj, uint32_t P) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(SB), 0, traj->eng->stream, traj->cg_x, P);
}

void launch_step_size(rl_traj *traj, rl_mlp *policy, float reg, double max_kl) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_step_size, dim3(1), dim3(SB), 0, traj->eng->stream, traj->vec, (uint32_t)policy->P, reg,
                     max_kl, traj->cg_x, traj->descent, policy->d_params, traj->prev_params, traj->trpo);
}

void launch_ls_set_params(rl_traj *traj, rl_mlp *policy, double ratio) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_ls_set_params, dim3(1), dim3(SB), 0, traj->eng->stream, policy->d_params, traj->prev_params,
                     traj->descent, (uint32_t)policy->P, (float)ratio, traj->trpo, comm_err_word(traj->eng),
                     wimg_if_current(policy), (int)policy->out_dim);
}

// ---------------------------------------------------------------- the weight image (bf16_tile.hpp)
const uint32_t *wimg_ensure(const rl_mlp *m) {
  rl_engine *e = m->eng;
  if (m->d_wimg == nullptr) {
    RL_HIP_CHECK(hipSetDevice(e->device));
    void *p = nullptr;
    RL_HIP_CHECK(hipMalloc(&p, (size_t)bt::WIMG_WORDS * sizeof(uint32_t)));
    m->d_wimg = static_cast<uint32_t *>(p);
    m->wimg_epoch = 0;
  }
  if (m->wimg_epoch != e->call_epoch) {
    ProfScope ps(e, RL_K_SMALL);
    hipLaunchKernelGGL(k_wimg_build, dim3(((uint32_t)m->P + SB - 1) / SB), dim3(SB), 0, e->stream, m->d_params,
                       (uint32_t)m->P, m->d_wimg, (int)m->out_dim);
    m->wimg_epoch = e->call_epoch;
  }
  return m->d_wimg;
}

void launch_ls_check(rl_traj *traj, uint32_t P, uint64_t B_total, int index, double ratio, double max_kl) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_ls_check, dim3(1), dim3(64), 0, traj->eng->stream, traj->vec, P, 1.0 / (double)B_total, index,
                     ratio, max_kl, traj->trpo);
}

void launch_ls_finalize(rl_traj *traj, rl_mlp *policy, double max_kl, int accept_violation) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  hipLaunchKernelGGL(k_ls_finalize, dim3(1), dim3(SB), 0, traj->eng->stream, policy->d_params, traj->prev_params,
                     (uint32_t)policy->P, max_kl, accept_violation, traj->trpo, comm_err_word(traj->eng),
                     veto_word(traj, policy));
  wimg_invalidate(policy);  // (a rollback rewrites the parameters without the image)
}

// bias corrections of the optimiser's NEXT step (advances the host's step count)
static void adam_next_step(rl_adam *opt, float *neg_step_size, float *sqrt_bc2) {
  if (opt->error_epoch != opt->eng->error_epoch) {
    // an entry point failed on this engine since this optimiser last stepped: launches of it may have been vetoed on the
    // device (a failed exchange, the range guard) after the host had counted them — the device's count is the truth
    rl_engine *e = opt->eng;
    RL_HIP_CHECK(hipStreamSynchronize(e->main_stream));
    RL_HIP_CHECK(hipStreamSynchronize(e->aux_stream));
    uint64_t s = 0;
    RL_HIP_CHECK(hipMemcpyAsync(&s, opt->d_step, sizeof(s), hipMemcpyDeviceToHost, e->stream));
    RL_HIP_CHECK(hipStreamSynchronize(e->stream));
    opt->host_step = s;
    opt->error_epoch = e->error_epoch;
  }
  opt->host_step += 1;
  const double bc1 = 1.0 - std::pow(opt->cfg.beta1, (double)opt->host_step);
  const double bc2 = 1.0 - std::pow(opt->cfg.beta2, (double)opt->host_step);
  *neg_step_size = -(float)(opt->cfg.learning_rate / bc1);
  *sqrt_bc2 = (float)std::sqrt(bc2);
}

void launch_reduce_adam(rl_traj *traj, rl_adam *opt, uint32_t rowsA, uint32_t rowsB, int loss_slot,
                        uint64_t B_total) {
  ProfScope ps(traj->eng, RL_K_REDUCE);
  uint32_t P = (uint32_t)opt->mod->P;
  float neg_step_size, sqrt_bc2;
  adam_next_step(opt, &neg_step_size, &sqrt_bc2);
  rl_engine *e = traj->eng;
  float *loss_out = loss_slot >= 0 ? traj->losses + loss_slot : (float *)nullptr;
  if (e->ipc_active) {
    ProfScope pa(e, RL_K_ALLREDUCE);  // counted as a collective as well: the exchange runs inside this launch
    const IpcPeers peers = ipc_peers_next(e);
    hipLaunchKernelGGL(k_reduce_adam<true>, dim3(cdiv(P + 4, 64)), dim3(1024), 0, e->stream, traj->slabA, rowsA, P,
                       traj->slabB, rowsB, traj->vec, opt->mod->d_params, opt->d_m, opt->d_v, opt->d_step,
                       opt->host_step, neg_step_size, sqrt_bc2, opt->cfg.beta1, opt->cfg.beta2, opt->cfg.eps,
                       opt->cfg.weight_decay, 1.0 / (double)B_total, loss_out, peers, wimg_if_current(opt->mod),
                       (int)opt->mod->out_dim, veto_word(traj, opt->mod));
  } else if (reduce_width(P) != 64) {
    hipLaunchKernelGGL(k_reduce_adam_narrow<16>, dim3(cdiv(P + 4, 16)), dim3(1024), 0, e->stream, traj->slabA, rowsA, P,
                       traj->slabB, rowsB, traj->vec, opt->mod->d_params, opt->d_m, opt->d_v, opt->d_step,
                       opt->host_step, neg_step_size, sqrt_bc2, opt->cfg.beta1, opt->cfg.beta2, opt->cfg.eps,
                       opt->cfg.weight_decay, 1.0 / (double)B_total, loss_out, wimg_if_current(opt->mod),
                       (int)opt->mod->out_dim, veto_word(traj, opt->mod));
  } else {
    hipLaunchKernelGGL(k_reduce_adam<false>, dim3(cdiv(P + 4, 64)), dim3(1024), 0, e->stream, traj->slabA, rowsA, P,
                       traj->slabB, rowsB, traj->vec, opt->mod->d_params, opt->d_m, opt->d_v, opt->d_step,
                       opt->host_step, neg_step_size, sqrt_bc2, opt->cfg.beta1, opt->cfg.beta2, opt->cfg.eps,
                       opt->cfg.weight_decay, 1.0 / (double)B_total, loss_out, IpcPeers{}, wimg_if_current(opt->mod),
                       (int)opt->mod->out_dim, veto_word(traj, opt->mod));
  }
}

void launch_adam_step(rl_traj *traj, rl_adam *opt, int loss_slot, uint64_t B_total) {
  ProfScope ps(traj->eng, RL_K_SMALL);
  uint32_t P = (uint32_t)opt->mod->P;
  float neg_step_size, sqrt_bc2;
  adam_next_step(opt, &neg_step_size, &sqrt_bc2);
  hipLaunchKernelGGL(k_adam_step, dim3((P + SB - 1) / SB), dim3(SB), 0, traj->eng->stream, opt->mod->d_params, traj->vec, opt->d_m,
                     opt->d_v, opt->d_step, P, opt->host_step, neg_step_size, sqrt_bc2, opt->cfg.beta1, opt->cfg.beta2,
                     opt->cfg.eps, opt->cfg.weight_decay, traj->vec + P, 1.0 / (double)B_total,
                     loss_slot >= 0 ? traj->losses + loss_slot : (float *)nullptr, comm_err_word(traj->eng),
                     wimg_if_current(opt->mod), (int)opt->mod->out_dim, veto_word(traj, opt->mod));
}

void launch_adam_step_vec(rl_adam *opt, const float *d_grad) {
  ProfScope ps(opt->mod->eng, RL_K_SMALL);
  uint32_t P = (uint32_t)opt->mod->P;
  float neg_step_size, sqrt_bc2;
  adam_next_step(opt, &neg_step_size, &sqrt_bc2);
  hipLaunchKernelGGL(k_adam_step, dim3((P + SB - 1) / SB), dim3(SB), 0, opt->mod->eng->stream, opt->mod->d_params, d_grad, opt->d_m,
                     opt->d_v, opt->d_step, P, opt->host_step, neg_step_size, sqrt_bc2, opt->cfg.beta1, opt->cfg.beta2,
                     opt->cfg.eps, opt->cfg.weight_decay, (const float *)nullptr, 0.0, (float *)nullptr,
                     (const int32_t *)nullptr, wimg_if_current(opt->mod), (int)opt->mod->out_dim,
                     (const uint32_t *)nullptr);
}
