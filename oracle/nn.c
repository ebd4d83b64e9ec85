/* nn.c — oracle restatement of the MLP, Categorical, GAE, TRPO (conjugate-gradient trust region),
 * Adam and the ValuesOpt critic update.  TEST INFRASTRUCTURE (see oracle.h).
 */
#include "oracle.h"
#include "../include/rl_chacha.h"
#include "../include/rl_detmath.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

uint64_t oracle_mlp_num_params(oracle_mlp_shape s) {
  return (uint64_t)s.hidden * s.in_dim + s.hidden + (uint64_t)s.out_dim * s.hidden + s.out_dim;
}

/* ---- instantiate the shared update math for f32 (engine transcendental contract) and f64 (libm) */
#define REAL float
#define SUF _f32
#define RFMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define REXP(x) rl_expf(x)
#define RLOG(x) rl_logf(x)
#define RMIN (-FLT_MAX)
#define RMAXV FLT_MAX
#include "nn_impl.inc"
#undef REAL
#undef SUF
#undef RFMA
#undef REXP
#undef RLOG
#undef RMIN
#undef RMAXV

#define REAL double
#define SUF _f64
#define RFMA(a, b, c) __builtin_fma((a), (b), (c))
#define REXP(x) exp(x)
#define RLOG(x) log(x)
#define RMIN (-DBL_MAX)
#define RMAXV DBL_MAX
#include "nn_impl.inc"
#undef REAL
#undef SUF
#undef RFMA
#undef REXP
#undef RLOG
#undef RMIN
#undef RMAXV

/* ------------------------------------------------------------------ init
 * Linear::new (ff/linear.rs:54-68): fan_in = in_dim + 1 for kernel AND bias; Initializer default
 * Uniform(FanAvg) => lim = sqrt(3 * 2 / (fan_in + fan_out)) (initializers.rs:31-38, 78-108, 159-163);
 * kernel fan_out = out_dim (shape [out, in]); bias shape [out] => fan_out = out (calculate_fan_in_and_fan_out
 * with a 1-D shape: shape[0]).  libtorch's RNG is never seeded by the reference, so the draw
 * stream is engine-defined: ChaCha8(seed), stream 0, one `gen::<f32>()` per element in flat order,
 * value = (2u - 1) * lim computed in f32. */
void oracle_mlp_init(oracle_mlp_shape s, uint64_t seed, float *params) {
  oracle_prng r;
  oracle_prng_seed_from_u64(&r, seed);
  uint32_t dims[2][2] = {{s.in_dim, s.hidden}, {s.hidden, s.out_dim}};
  float *p = params;
  for (int l = 0; l < 2; ++l) {
    uint32_t in = dims[l][0], out = dims[l][1];
    float lim = (float)sqrt(3.0 * (2.0 / ((double)(in + 1) + (double)out)));
    uint64_t n = (uint64_t)in * out + out;
    for (uint64_t i = 0; i < n; ++i) {
      float u = oracle_prng_gen_f32(&r);
      *p++ = (2.0f * u - 1.0f) * lim;
    }
  }
}

void oracle_mlp_forward_f32(oracle_mlp_shape s, const float *params, const float *x, float *out) {
  mlp_view_f32 m = view_f32(s, params);
  float *pre = (float *)malloc(sizeof(float) * 2 * m.H);
  mlp_fwd_f32(&m, x, pre, pre + m.H, out);
  free(pre);
}

void oracle_mlp_forward_batch_f32(oracle_mlp_shape s, const float *params, const float *x, uint64_t n, float *out) {
  mlp_view_f32 m = view_f32(s, params);
  float *pre = (float *)malloc(sizeof(float) * 2 * m.H);
  for (uint64_t i = 0; i < n; ++i) mlp_fwd_f32(&m, x + i * m.D, pre, pre + m.H, out + i * m.A);
  free(pre);
}

/* ------------------------------------------------------------------ Categorical
 * src/torch/distributions/categorical.rs:29-77 */
static float expf_sel(float x, int use_libm) { return use_libm ? expf(x) : rl_expf(x); }
static float logf_sel(float x, int use_libm) { return use_libm ? logf(x) : rl_logf(x); }

void oracle_log_softmax_f32(const float *z, uint32_t n, float *lp, int use_libm) {
  float m = z[0];
  for (uint32_t a = 1; a < n; ++a)
    if (z[a] > m) m = z[a];
  float s = 0.0f;
  for (uint32_t a = 0; a < n; ++a) s += expf_sel(z[a] - m, use_libm);
  float ls = logf_sel(s, use_libm);
  for (uint32_t a = 0; a < n; ++a) lp[a] = (z[a] - m) - ls;
}

/* `log_probs.exp().multinomial(1, true)` (categorical.rs:53) with an explicit uniform: inverse CDF over
 * p_a = exp(lp_a) accumulated in index order; the last index absorbs rounding. */
int oracle_categorical_sample_u(const float *lp, uint32_t n, float u, int use_libm) {
  float cum = 0.0f;
  for (uint32_t a = 0; a + 1 < n; ++a) {
    cum += expf_sel(lp[a], use_libm);
    if (u < cum) return (int)a;
  }
  return (int)(n - 1);
}

float oracle_categorical_entropy_f32(const float *lp, uint32_t n, int use_libm) {
  float s = 0.0f;
  for (uint32_t a = 0; a < n; ++a) {
    float c = lp[a] < -FLT_MAX ? -FLT_MAX : lp[a];
    s += c * expf_sel(lp[a], use_libm);
  }
  return -s;
}

float oracle_categorical_kl_f32(const float *lp_self, const float *lp_other, uint32_t n, int use_libm) {
  float s = 0.0f;
  for (uint32_t a = 0; a < n; ++a) {
    float rel = lp_self[a] - lp_other[a];
    if (rel < -FLT_MAX) rel = -FLT_MAX;
    s += rel * expf_sel(lp_self[a], use_libm);
  }
  return s;
}

/* ------------------------------------------------------------------ GAE on packed features
 * eval_extended_state_values / temporal_differences / gae (critics/mod.rs:116-131, 158-199) */
void oracle_gae_packed(oracle_mlp_shape cs, const float *critic_params, const oracle_features *f, float gamma,
                       float lambda, float *adv_out, float *ext_values_out) {
  float *ext = (float *)malloc(sizeof(float) * (f->n_ext ? f->n_ext : 1));
  oracle_mlp_forward_batch_f32(cs, critic_params, f->ext_obs, f->n_ext, ext); /* out_dim == 1 => squeeze */
  for (uint64_t i = 0; i < f->n_ext; ++i)
    if (f->is_invalid[i]) ext[i] = 0.0f; /* masked_fill_ */
  if (ext_values_out) memcpy(ext_values_out, ext, sizeof(float) * f->n_ext);
  float *v = (float *)malloc(sizeof(float) * (f->n_steps ? f->n_steps : 1));
  oracle_packed_trim_end_f32(ext, f->ext_batch_sizes, f->n_ext_batches, 1, v);
  const float *vnext = ext + f->n_episodes; /* view_trim_start(1): skip the first time slice */
  for (uint64_t i = 0; i < f->n_steps; ++i) {
    float dn = gamma * vnext[i];
    float t = f->rewards[i] + dn;
    adv_out[i] = t - v[i];
  }
  float disc = lambda * gamma; /* f32 product (critics/mod.rs:198) */
  oracle_discounted_cumsum_from_end_f32(adv_out, f->n_steps, disc, f->batch_sizes, f->n_batches);
  free(v);
  free(ext);
}

/* one_step_values (critics/mod.rs:139-150), the OneStepTd arm of StepValueTarget::targets (:219-229):
 * `rewards + discount_factor * estimated_next_values` with estimated_next_values = view_trim_start(1) of the masked
 * extended values — Terminate (and a dropped dangling step) bootstraps from 0, Interrupt from V(successor).
 * Scalar times tensor, then tensor plus tensor: two f32 roundings. */
void oracle_one_step_values_packed(oracle_mlp_shape cs, const float *critic_params, const oracle_features *f,
                                   float gamma, float *out) {
  float *ext = (float *)malloc(sizeof(float) * (f->n_ext ? f->n_ext : 1));
  oracle_mlp_forward_batch_f32(cs, critic_params, f->ext_obs, f->n_ext, ext);
  for (uint64_t i = 0; i < f->n_ext; ++i)
    if (f->is_invalid[i]) ext[i] = 0.0f; /* masked_fill_ */
  const float *vnext = ext + f->n_episodes; /* view_trim_start(1) */
  for (uint64_t i = 0; i < f->n_steps; ++i) {
    float dn = gamma * vnext[i];
    out[i] = f->rewards[i] + dn;
  }
  free(ext);
}

/* reward_to_go (critics/mod.rs:101-105) */
void oracle_reward_to_go_packed(const oracle_features *f, float gamma, float *out) {
  memcpy(out, f->rewards, sizeof(float) * f->n_steps);
  oracle_discounted_cumsum_from_end_f32(out, f->n_steps, gamma, f->batch_sizes, f->n_batches);
}

/* ------------------------------------------------------------------ conjugate gradient + TRPO step */
void oracle_trpo_cfg_default(oracle_trpo_cfg *c) {
  /* ConjugateGradientOptimizerConfig::default (conjugate_gradient.rs:55-65); TrpoConfig (trpo.rs:38) */
  c->iterations = 10;
  c->max_backtracks = 15;
  c->backtrack_ratio = 0.8;
  c->hpv_reg_coeff = 1e-5;
  c->accept_violation = 0;
  c->max_kl = 0.01;
}

/* dot / solve_cg / oracle_cg_dense / oracle_trpo_update are instantiated from nn_impl.inc (f32 and f64) */

/* ------------------------------------------------------------------ Adam
 * COptimizer::adam (optimizers/coptimizer.rs:158-167) -> torch::optim::Adam of libtorch 1.12 (third party,
 * not under /root/reference): amsgrad off, eps 1e-8 default. */
void oracle_adam_cfg_default(oracle_adam_cfg *c) {
  c->lr = 1e-3;
  c->beta1 = 0.9;
  c->beta2 = 0.999;
  c->eps = 1e-8;
  c->weight_decay = 0.0;
}

oracle_adam_state *oracle_adam_new(uint64_t n) {
  oracle_adam_state *st = (oracle_adam_state *)calloc(1, sizeof(*st));
  st->n = n;
  st->m = (float *)calloc(n, sizeof(float));
  st->v = (float *)calloc(n, sizeof(float));
  return st;
}

void oracle_adam_free(oracle_adam_state *st) {
  if (!st) return;
  free(st->m);
  free(st->v);
  free(st);
}

void oracle_adam_step_f32(oracle_adam_state *st, const oracle_adam_cfg *cfg, float *params, const float *grad) {
  st->step += 1;
  double bc1 = 1.0 - pow(cfg->beta1, (double)st->step);
  double bc2 = 1.0 - pow(cfg->beta2, (double)st->step);
  float b1 = (float)cfg->beta1, b2 = (float)cfg->beta2;
  float omb1 = (float)(1.0 - cfg->beta1), omb2 = (float)(1.0 - cfg->beta2);
  float sqrt_bc2 = (float)sqrt(bc2);
  float step_size = (float)(cfg->lr / bc1);
  float eps = (float)cfg->eps;
  for (uint64_t i = 0; i < st->n; ++i) {
    float gi = grad[i];
    if (cfg->weight_decay != 0.0) gi = gi + (float)cfg->weight_decay * params[i];
    st->m[i] = st->m[i] * b1 + omb1 * gi;
    st->v[i] = st->v[i] * b2 + omb2 * gi * gi;
    float denom = sqrtf(st->v[i]) / sqrt_bc2 + eps;
    params[i] = params[i] + ((-step_size) * st->m[i]) / denom; /* addcdiv_(exp_avg, denom, -step_size) */
  }
}

/* ------------------------------------------------------------------ ValuesOpt critic
 * loss = mse_loss(critic(obs).squeeze(-1), targets, Mean) (critics/opt.rs:109-115) */
void oracle_critic_grad_f32(oracle_mlp_shape s, const float *params, const float *obs, const float *targets,
                            uint64_t n, float *grad_out, float *loss_out) {
  mlp_view_f32 m = view_f32(s, params);
  uint64_t P = oracle_mlp_num_params(s);
  double *g = (double *)calloc(P, sizeof(double));
  float *pre = (float *)malloc(sizeof(float) * (2 * m.H + 2));
  float *h = pre + m.H, *z = h + m.H, *dz = z + 1;
  double loss = 0.0;
  float two_over_n = 2.0f / (float)n;
  for (uint64_t i = 0; i < n; ++i) {
    const float *x = obs + i * m.D;
    mlp_fwd_f32(&m, x, pre, h, z);
    float d = z[0] - targets[i];
    loss += (double)(d * d);
    dz[0] = d * two_over_n;
    mlp_bwd_acc_f32(&m, x, pre, h, dz, g);
  }
  for (uint64_t i = 0; i < P; ++i) grad_out[i] = (float)g[i];
  if (loss_out) *loss_out = (float)(loss / (double)n);
  free(pre);
  free(g);
}

/* ---- f64 ground truth over MANY samples (full-size parity: 65,536 lanes x 128 steps = 8.4 M samples), OpenMP over
 * chunks of samples.  Inputs in the device's f32 storage (converted per chunk); every chunk runs the single-threaded
 * f64 functions above on its samples, and the chunk results — means over the chunk — are combined with their sample
 * weights in f64.  Chunk order does not matter at f64 resolution for the 1e-6 comparisons these serve. */
#define ORACLE_MT_CHUNK 4096
static void critic_grad_f64_chunk(oracle_mlp_shape s, const double *params, const double *obs, const double *targets,
                                  uint64_t n, double *grad_out, double *loss_out) {
  mlp_view_f64 m = view_f64(s, params);
  uint64_t P = oracle_mlp_num_params(s);
  double *pre = (double *)malloc(sizeof(double) * (2 * m.H + 2));
  double *h = pre + m.H, *z = h + m.H, *dz = z + 1;
  double loss = 0.0;
  for (uint64_t i = 0; i < P; ++i) grad_out[i] = 0.0;
  for (uint64_t i = 0; i < n; ++i) {
    const double *x = obs + i * m.D;
    mlp_fwd_f64(&m, x, pre, h, z);
    double d = z[0] - targets[i];
    loss += d * d;
    dz[0] = d * (2.0 / (double)n);
    mlp_bwd_acc_f64(&m, x, pre, h, dz, grad_out);
  }
  *loss_out = loss / (double)n;
  free(pre);
}

/* kind 0: policy surrogate gradient (aux = advantages; loss_out = -mean(A) at ratio 1), 1: Fisher-vector product with
 * tangent v (no reg term; loss_out untouched), 2: critic MSE gradient (aux = targets; loss_out = mean squared error) */
static void grad_mt(int kind, int f32_samples, oracle_mlp_shape s, const float *params, const float *obs,
                    const uint8_t *actions, const float *aux, const float *v, uint64_t n, double *grad_out,
                    double *loss_out);
void oracle_grad_f64_mt(int kind, oracle_mlp_shape s, const float *params, const float *obs, const uint8_t *actions,
                        const float *aux, const float *v, uint64_t n, double *grad_out, double *loss_out) {
  grad_mt(kind, 0, s, params, obs, actions, aux, v, n, grad_out, loss_out);
}
/* the same with the per-sample arithmetic of the f32 oracle functions (engine transcendentals, f32 fma chains; sums over a
 * chunk in f64, rounded to f32 per chunk, chunks combined in f64): what a correct f32 evaluation of the pass gives at
 * this sample count — the yardstick for the device's distance from the f64 truth */
void oracle_grad_f32_mt(int kind, oracle_mlp_shape s, const float *params, const float *obs, const uint8_t *actions,
                        const float *aux, const float *v, uint64_t n, double *grad_out, double *loss_out) {
  grad_mt(kind, 1, s, params, obs, actions, aux, v, n, grad_out, loss_out);
}
static void grad_mt(int kind, int f32_samples, oracle_mlp_shape s, const float *params, const float *obs,
                    const uint8_t *actions, const float *aux, const float *v, uint64_t n, double *grad_out,
                    double *loss_out) {
  const uint64_t P = oracle_mlp_num_params(s), D = s.in_dim;
  const uint64_t n_chunks = (n + ORACLE_MT_CHUNK - 1) / ORACLE_MT_CHUNK;
  double *pd = (double *)malloc(sizeof(double) * 2 * P);
  double *vd = pd + P;
  for (uint64_t i = 0; i < P; ++i) {
    pd[i] = (double)params[i];
    vd[i] = v ? (double)v[i] : 0.0;
  }
  for (uint64_t i = 0; i < P; ++i) grad_out[i] = 0.0;
  double loss_total = 0.0;
#pragma omp parallel
  {
    double *g = (double *)calloc(2 * P, sizeof(double));
    double *gc = g + P;
    double *x = (double *)malloc(sizeof(double) * ORACLE_MT_CHUNK * (D + 1));
    double *a = x + ORACLE_MT_CHUNK * D;
    int64_t *act = (int64_t *)malloc(sizeof(int64_t) * ORACLE_MT_CHUNK);
    double loss = 0.0;
#pragma omp for schedule(dynamic, 1)
    for (uint64_t c = 0; c < n_chunks; ++c) {
      const uint64_t lo = c * ORACLE_MT_CHUNK, m = (lo + ORACLE_MT_CHUNK <= n ? ORACLE_MT_CHUNK : n - lo);
      for (uint64_t i = 0; i < m * D; ++i) x[i] = (double)obs[lo * D + i];
      for (uint64_t i = 0; i < m; ++i) {
        a[i] = aux ? (double)aux[lo + i] : 0.0;
        act[i] = actions ? (int64_t)actions[lo + i] : 0;
      }
      double l = 0.0;
      if (f32_samples) {
        float *gf = (float *)malloc(sizeof(float) * P);
        float lf = 0.0f;
        static const float zero_adv = 0.0f;
        (void)zero_adv;
        if (kind == 0) oracle_policy_grad_f32(s, params, obs + lo * D, act, aux + lo, m, gf, &lf);
        else if (kind == 1) oracle_policy_fvp_f32(s, params, obs + lo * D, m, v, 0.0f, gf);
        else oracle_critic_grad_f32(s, params, obs + lo * D, aux + lo, m, gf, &lf);
        for (uint64_t i = 0; i < P; ++i) gc[i] = (double)gf[i];
        l = (double)lf;
        free(gf);
      } else if (kind == 0) oracle_policy_grad_f64(s, pd, x, act, a, m, gc, &l);
      else if (kind == 1) oracle_policy_fvp_f64(s, pd, x, m, vd, 0.0, gc);
      else critic_grad_f64_chunk(s, pd, x, a, m, gc, &l);
      const double w = (double)m / (double)n;  /* chunk mean -> its share of the full mean */
      for (uint64_t i = 0; i < P; ++i) g[i] += w * gc[i];
      loss += w * l;
    }
#pragma omp critical
    {
      for (uint64_t i = 0; i < P; ++i) grad_out[i] += g[i];
      loss_total += loss;
    }
    free(act);
    free(x);
    free(g);
  }
  if (loss_out) *loss_out = loss_total;
  free(pd);
}

/* ValuesOpt::update via n_backward_steps (critics/opt.rs:100-126; torch/agents/mod.rs:35-72;
 * COptimizer::backward_step coptimizer.rs:14-26) */
void oracle_critic_update_f32(oracle_mlp_shape s, float *params, oracle_adam_state *st, const oracle_adam_cfg *cfg,
                              const float *obs, const float *targets, uint64_t n, uint64_t n_steps,
                              float *losses_out) {
  uint64_t P = oracle_mlp_num_params(s);
  float *g = (float *)malloc(sizeof(float) * P);
  for (uint64_t k = 0; k < n_steps; ++k) {
    float loss;
    oracle_critic_grad_f32(s, params, obs, targets, n, g, &loss);
    if (losses_out) losses_out[k] = loss;
    oracle_adam_step_f32(st, cfg, params, g);
  }
  free(g);
}

/* ------------------------------------------------------------------ PPO / REINFORCE updates
 * Ppo::update (policies/ppo.rs:97-146): initial log-probs under no_grad, then n_backward_steps with Adam.
 * losses_out[k] = loss before step k; *entropy_out = mean entropy at the initial parameters. */
void oracle_ppo_update_f32(oracle_mlp_shape s, float *params, oracle_adam_state *st, const oracle_adam_cfg *cfg,
                           const float *obs, const int64_t *actions, const float *adv, uint64_t n, uint64_t n_steps,
                           double clip_distance, float *losses_out, float *entropy_out) {
  uint64_t P = oracle_mlp_num_params(s);
  float *g = (float *)malloc(sizeof(float) * P);
  float *lp0 = (float *)malloc(sizeof(float) * (n ? n : 1));
  oracle_policy_logp_f32(s, params, obs, actions, n, lp0, entropy_out);
  float lo = (float)(1.0 - clip_distance), hi = (float)(1.0 + clip_distance);
  for (uint64_t k = 0; k < n_steps; ++k) {
    float loss;
    oracle_ppo_grad_f32(s, params, obs, actions, adv, lp0, n, lo, hi, g, &loss);
    if (losses_out) losses_out[k] = loss;
    oracle_adam_step_f32(st, cfg, params, g);
  }
  free(lp0);
  free(g);
}

/* Reinforce::update (policies/reinforce.rs:64-88): one backward_step */
void oracle_reinforce_update_f32(oracle_mlp_shape s, float *params, oracle_adam_state *st, const oracle_adam_cfg *cfg,
                                 const float *obs, const int64_t *actions, const float *adv, uint64_t n,
                                 float *loss_out, float *entropy_out) {
  uint64_t P = oracle_mlp_num_params(s);
  float *g = (float *)malloc(sizeof(float) * P);
  float *lp = (float *)malloc(sizeof(float) * (n ? n : 1));
  oracle_policy_logp_f32(s, params, obs, actions, n, lp, entropy_out);
  if (loss_out) *loss_out = oracle_reinforce_loss_f32(s, params, obs, actions, adv, n);
  oracle_policy_grad_f32(s, params, obs, actions, adv, n, g, NULL);
  oracle_adam_step_f32(st, cfg, params, g);
  free(lp);
  free(g);
}

/* ------------------------------------------------------------------ MLPs of any hidden_sizes and activations
 * Mlp::forward (src/torch/modules/ff/mlp.rs:139-151): every hidden Linear followed by `activation`, the last Linear by
 * `output_activation`; Activation::forward (ff/activation.rs:85-92): identity / relu / sigmoid / tanh.  Parameters flat
 * as [kernel [out][in], bias [out]] per layer (ff/linear.rs:108-110).  Dot products in the engine's order, `acc = bias;
 * acc = fma(x_k, w_k, acc)`, k ascending; sigmoid / tanh are the engine's deterministic ones (include/rl_detmath.h) —
 * so the f32 version is what the device's per-layer kernels must reproduce bit for bit.  `act`, `out_act`: 0 Identity,
 * 1 Relu, 2 Sigmoid, 3 Tanh (the reference enum's declaration order).  x: [rows][in_dim]; out: [rows][out_dim]. */
static float layers_act_f32(int act, float x) {
  switch (act) {
    case 1: return x > 0.0f ? x : 0.0f;
    case 2: return rl_sigmoidf(x);
    case 3: return rl_tanhf(x);
    default: return x;
  }
}
void oracle_mlp_layers_forward_f32(uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden, uint32_t out_dim,
                                   int act, int out_act, const float *params, const float *x, uint64_t rows,
                                   float *out) {
  float a[256], b[256];
  for (uint64_t r = 0; r < rows; ++r) {
    const float *in = x + r * in_dim;
    uint32_t K = in_dim;
    const float *p = params;
    float *cur = a, *nxt = b;
    for (uint32_t l = 0; l <= n_hidden; ++l) {
      const uint32_t N = l == n_hidden ? out_dim : hidden_sizes[l];
      const float *W = p, *bias = p + (uint64_t)N * K;
      float *dst = l == n_hidden ? out + r * out_dim : cur;
      for (uint32_t n = 0; n < N; ++n) {
        float acc = bias[n];
        for (uint32_t k = 0; k < K; ++k) acc = __builtin_fmaf(in[k], W[(uint64_t)n * K + k], acc);
        dst[n] = layers_act_f32(l == n_hidden ? out_act : act, acc);
      }
      p = bias + N;
      in = dst;
      K = N;
      float *t = cur;
      cur = nxt;
      nxt = t;
    }
  }
}

/* ------------------------------------------------------------------ Linear::new with any Initializer
 * LinearConfig { kernel_init, bias_init } (src/torch/modules/ff/linear.rs:13-33,54-68) for every layer of an MLP;
 * Initializer / VarianceScale / TensorBuilder::build / init_orthogonal (src/torch/initializers.rs:8-64,67-83,152-176,
 * 328-364).  kind: 0 Zeros, 1 Constant(value), 2 Uniform(scale), 3 Normal(scale), 4 Orthogonal; scale: 0
 * Constant(value), 1 FanIn, 2 FanOut, 3 FanAvg; fan_in = in_dim + 1 for kernel and bias, fan_out = shape[0].
 * The draw stream is the engine's (libtorch's generator is never seeded by the reference): ChaCha8(seed), stream 0, one
 * gen::<f32>() per uniform element in flat order, Box-Muller on consecutive pairs for normal elements, orthogonal = QR by
 * modified Gram-Schmidt applied twice in f64 (positive diagonal of R: the sign fold is the identity). */
static void layers_normals(oracle_prng *r, size_t count, double *z) {
  const double two_pi = 6.283185307179586;
  for (size_t i = 0; i < count; i += 2) {
    const double u1 = (double)oracle_prng_gen_f32(r), u2 = (double)oracle_prng_gen_f32(r);
    double rho = sqrt(-2.0 * log(1.0 - u1)), sn, cs;
    rl_sincos(two_pi * u2, &sn, &cs);
    z[i] = rho * cs;
    if (i + 1 < count) z[i + 1] = rho * sn;
  }
}
static double layers_variance(int scale, double value, double fan_in, double fan_out) {
  switch (scale) {
    case 0: return value;
    case 1: return 1.0 / fan_in;
    case 2: return 1.0 / fan_out;
    default: return 2.0 / (fan_in + fan_out);
  }
}
static void layers_fill(oracle_prng *r, int kind, int scale, double value, float *dst, uint64_t rows, uint64_t cols,
                        double fan_in) {
  const size_t count = (size_t)rows * cols;
  const double fan_out = (double)rows;
  if (kind == 0) {
    for (size_t i = 0; i < count; ++i) dst[i] = 0.0f;
  } else if (kind == 1) {
    for (size_t i = 0; i < count; ++i) dst[i] = (float)value;
  } else if (kind == 2) {
    const float lim = (float)sqrt(3.0 * layers_variance(scale, value, fan_in, fan_out));
    for (size_t i = 0; i < count; ++i) {
      float t = 2.0f * oracle_prng_gen_f32(r);
      t = t - 1.0f;
      dst[i] = t * lim;
    }
  } else {
    double *z = (double *)malloc(sizeof(double) * (count + 1));
    layers_normals(r, count, z);
    if (kind == 3) {
      const double sd = sqrt(layers_variance(scale, value, fan_in, fan_out));
      for (size_t i = 0; i < count; ++i) dst[i] = (float)(sd * z[i]);
    } else {
      const int wide = rows < cols;
      const uint64_t R = wide ? cols : rows, Cn = wide ? rows : cols;
      double *a = (double *)malloc(sizeof(double) * (size_t)R * Cn);
      for (uint64_t rr = 0; rr < rows; ++rr)
        for (uint64_t c = 0; c < cols; ++c) {
          const double v = z[(size_t)rr * cols + c];
          if (wide) a[(size_t)rr * R + c] = v;
          else a[(size_t)c * R + rr] = v;
        }
      for (uint64_t c = 0; c < Cn; ++c) {
        double *v = a + (size_t)c * R;
        for (int pass = 0; pass < 2; ++pass)
          for (uint64_t q = 0; q < c; ++q) {
            const double *w = a + (size_t)q * R;
            double dot = 0.0;
            for (uint64_t rr = 0; rr < R; ++rr) dot += w[rr] * v[rr];
            for (uint64_t rr = 0; rr < R; ++rr) v[rr] -= dot * w[rr];
          }
        double nrm = 0.0;
        for (uint64_t rr = 0; rr < R; ++rr) nrm += v[rr] * v[rr];
        nrm = sqrt(nrm);
        for (uint64_t rr = 0; rr < R; ++rr) v[rr] /= nrm;
      }
      for (uint64_t rr = 0; rr < rows; ++rr)
        for (uint64_t c = 0; c < cols; ++c)
          dst[(size_t)rr * cols + c] = (float)(wide ? a[(size_t)rr * R + c] : a[(size_t)c * R + rr]);
      free(a);
    }
    free(z);
  }
}
void oracle_mlp_layers_init(uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden, uint32_t out_dim,
                            uint64_t seed, int k_kind, int k_scale, double k_value, int b_kind, int b_scale,
                            double b_value, float *params) {
  oracle_prng r;
  oracle_prng_seed_from_u64(&r, seed);
  uint32_t K = in_dim;
  float *p = params;
  for (uint32_t l = 0; l <= n_hidden; ++l) {
    const uint32_t N = l == n_hidden ? out_dim : hidden_sizes[l];
    layers_fill(&r, k_kind, k_scale, k_value, p, N, K, (double)K + 1.0);
    p += (size_t)N * K;
    layers_fill(&r, b_kind, b_scale, b_value, p, N, 1, (double)K + 1.0);
    p += N;
    K = N;
  }
}

/* RnnWeights::new with RnnBaseConfig { input_weights_init, hidden_weights_init, bias_init } (seq/rnn/mod.rs:20-45,223-257:
 * per layer W_ih [G H, layer input], W_hh [G H, H], b_ih, b_hh [G H] — 1-D tensors: fan_in 1, fan_out G H,
 * calculate_fan_in_and_fan_out, initializers.rs:90-103) and Linear::new with LinearConfig's pair (fan_in = in + 1) for
 * the chain's MLP; the engine's draw stream, layer after layer.  inits: input, hidden, bias, mlp kernel, mlp bias. */
void oracle_stack_init_with(oracle_gru_shape s, uint32_t num_layers, uint64_t seed, const oracle_init_spec *inits,
                            float *params) {
  const uint64_t H = s.hidden, D = s.in_dim, H2 = s.mlp_hidden, A = s.out_dim;
  const uint64_t R = (s.cell == ORACLE_CELL_LSTM ? 4 : 3) * H;
  oracle_prng r;
  oracle_prng_seed_from_u64(&r, seed);
  float *p = params;
  for (uint32_t l = 0; l < num_layers; ++l) {
    const uint64_t K = l == 0 ? D : H;
    layers_fill(&r, inits[0].kind, inits[0].scale, inits[0].value, p, R, K, (double)K);
    p += R * K;
    layers_fill(&r, inits[1].kind, inits[1].scale, inits[1].value, p, R, H, (double)H);
    p += R * H;
    for (int b = 0; b < 2; ++b) {
      layers_fill(&r, inits[2].kind, inits[2].scale, inits[2].value, p, R, 1, 1.0);
      p += R;
    }
  }
  const uint64_t dims[2][2] = {{H, H2}, {H2, A}};
  for (int l = 0; l < 2; ++l) {
    const uint64_t fin = dims[l][0], fout = dims[l][1];
    layers_fill(&r, inits[3].kind, inits[3].scale, inits[3].value, p, fout, fin, (double)fin + 1.0);
    p += fin * fout;
    layers_fill(&r, inits[4].kind, inits[4].scale, inits[4].value, p, fout, 1, (double)fin + 1.0);
    p += fout;
  }
}
