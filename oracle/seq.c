/* seq.c — oracle restatement of the recurrent configuration (BASELINE.json configs[4]): Chain wrapped in a step
 * limit, vectorised over lanes, with a GRU -> ReLU -> MLP policy / critic, backward through time, PPO and the
 * ValuesOpt critic update.  TEST INFRASTRUCTURE (see oracle.h).
 *
 * Reference: Chain (src/envs/chain.rs:69-106), LatentStepLimit (src/envs/wrappers/step_limit.rs:57-89),
 * IndexSpace features (src/spaces/index.rs:97-116), GruMlpConfig = ChainConfig<GruConfig, MlpConfig>
 * (src/torch/modules/mod.rs:14, chain.rs:12-56), RnnWeights::new (src/torch/modules/seq/rnn/mod.rs:223-257),
 * init_orthogonal (src/torch/initializers.rs:328-364), Steps::step (src/simulation/steps.rs:113-167).
 */
#include "oracle.h"
#include "../include/rl_chacha.h"
#include "../include/rl_detmath.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

uint64_t oracle_gru_num_params(oracle_gru_shape s) {
  uint64_t H = s.hidden, D = s.in_dim, H2 = s.mlp_hidden, A = s.out_dim;
  uint64_t G = s.cell == ORACLE_CELL_LSTM ? 4 : 3; /* RnnImpl::GATES_MULTIPLE (gru.rs / lstm.rs:20) */
  return G * H * D + G * H * H + G * H + G * H + H2 * H + H2 + A * H2 + A;
}

#define REAL float
#define SUF _f32
#define RFMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define RSIG(x) rl_sigmoidf(x)
#define RTANH(x) rl_tanhf(x)
#include "seq_impl.inc"
#include "stack_impl.inc"
#undef REAL
#undef SUF
#undef RFMA
#undef RSIG
#undef RTANH

#define REAL double
#define SUF _f64
#define RFMA(a, b, c) __builtin_fma((a), (b), (c))
#define RSIG(x) (1.0 / (1.0 + exp(-(x))))
#define RTANH(x) tanh(x)
#include "seq_impl.inc"
#include "stack_impl.inc"
#undef REAL
#undef SUF
#undef RFMA
#undef RSIG
#undef RTANH

/* ------------------------------------------------------------------ initialisation
 * RnnBaseConfig::default (seq/rnn/mod.rs:36-45): W_ih Glorot-uniform over the whole [3H, in] matrix
 * (lim = sqrt(6 / (in + 3H)), initializers.rs:96-108,159-163), W_hh orthogonal (QR of a normal [3H, H] matrix
 * with the signs of diag(R) folded into Q, initializers.rs:328-364), biases zero; then the MLP's two Linear
 * layers as oracle_mlp_init.  Engine-defined stream (the reference never seeds libtorch): ChaCha8(seed),
 * stream 0, one gen::<f32>() per uniform element; normals by Box-Muller on consecutive pairs of draws
 * (z0 = rho cos, z1 = rho sin, rho = sqrt(-2 ln(1 - u1)), angle 2 pi u2) in f64; QR by modified Gram-Schmidt
 * applied twice in f64 (R has a positive diagonal, so the sign fold is the identity). */
/* one recurrent layer's [W_ih (R x K), W_hh (R x H), b_ih, b_hh] drawn from r; returns the position behind it */
static float *init_rnn_layer(oracle_prng *r, float *p, uint64_t R, uint64_t K, uint64_t H) {
  float lim = (float)sqrt(3.0 * (2.0 / ((double)K + (double)R)));
  for (uint64_t i = 0; i < R * K; ++i) {
    float u = oracle_prng_gen_f32(r);
    *p++ = (2.0f * u - 1.0f) * lim;
  }
  double *a = (double *)malloc(sizeof(double) * R * H); /* column-major: a[c * R + row] */
  double *rowmajor = (double *)malloc(sizeof(double) * R * H);
  const double two_pi = 6.283185307179586;
  for (uint64_t i = 0; i < R * H; i += 2) {
    double u1 = (double)oracle_prng_gen_f32(r), u2 = (double)oracle_prng_gen_f32(r);
    double rho = sqrt(-2.0 * log(1.0 - u1)), sn, cs;
    rl_sincos(two_pi * u2, &sn, &cs);
    rowmajor[i] = rho * cs;
    if (i + 1 < R * H) rowmajor[i + 1] = rho * sn;
  }
  for (uint64_t row = 0; row < R; ++row)
    for (uint64_t c = 0; c < H; ++c) a[c * R + row] = rowmajor[row * H + c];
  for (uint64_t c = 0; c < H; ++c) {
    double *v = a + c * R;
    for (int pass = 0; pass < 2; ++pass)
      for (uint64_t q = 0; q < c; ++q) {
        const double *w = a + q * R;
        double dot = 0.0;
        for (uint64_t row = 0; row < R; ++row) dot += w[row] * v[row];
        for (uint64_t row = 0; row < R; ++row) v[row] -= dot * w[row];
      }
    double nrm = 0.0;
    for (uint64_t row = 0; row < R; ++row) nrm += v[row] * v[row];
    nrm = sqrt(nrm);
    for (uint64_t row = 0; row < R; ++row) v[row] /= nrm;
  }
  for (uint64_t row = 0; row < R; ++row)
    for (uint64_t c = 0; c < H; ++c) *p++ = (float)a[c * R + row];
  free(rowmajor);
  free(a);
  for (uint64_t i = 0; i < 2 * R; ++i) *p++ = 0.0f;
  return p;
}

/* RnnWeights::new's layer loop (seq/rnn/mod.rs:223-257): layer 0 reads in_dim features, the others the hidden size */
void oracle_stack_init(oracle_gru_shape s, uint32_t num_layers, uint64_t seed, float *params) {
  const uint64_t H = s.hidden, D = s.in_dim, H2 = s.mlp_hidden, A = s.out_dim;
  const uint64_t R = (s.cell == ORACLE_CELL_LSTM ? 4 : 3) * H; /* rows of the gate matrices */
  oracle_prng r;
  oracle_prng_seed_from_u64(&r, seed);
  float *p = params;
  for (uint32_t l = 0; l < num_layers; ++l) p = init_rnn_layer(&r, p, R, l == 0 ? D : H, H);
  uint64_t dims[2][2] = {{H, H2}, {H2, A}};
  for (int l = 0; l < 2; ++l) {
    uint64_t in = dims[l][0], out = dims[l][1];
    float lm = (float)sqrt(3.0 * (2.0 / ((double)(in + 1) + (double)out)));
    for (uint64_t i = 0; i < in * out + out; ++i) {
      float u = oracle_prng_gen_f32(&r);
      *p++ = (2.0f * u - 1.0f) * lm;
    }
  }
}

void oracle_gru_init(oracle_gru_shape s, uint64_t seed, float *params) { oracle_stack_init(s, 1, seed, params); }

uint64_t oracle_stack_num_params(oracle_gru_shape s, uint32_t num_layers) {
  uint64_t H = s.hidden, G = s.cell == ORACLE_CELL_LSTM ? 4 : 3;
  return oracle_gru_num_params(s) + (uint64_t)(num_layers - 1) * (G * H * 2 * H + 2 * G * H);
}

/* ------------------------------------------------------------------ vectorised Chain lanes */
oracle_chain_lanes *oracle_chain_lanes_new(uint64_t size, int limit_kind, uint64_t max_steps, uint64_t n_lanes,
                                           uint64_t lane_offset, uint64_t seed_env, uint64_t seed_actor) {
  oracle_chain_lanes *l = (oracle_chain_lanes *)calloc(1, sizeof(*l));
  oracle_chain_default(&l->env);
  l->env.size = size;
  l->limit_kind = limit_kind;
  l->max_steps = max_steps;
  l->seed_env = seed_env;
  l->seed_actor = seed_actor;
  l->n_lanes = n_lanes;
  l->lane_offset = lane_offset;
  l->state = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->steps_remaining = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->reset_count = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->initial = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->env_pos = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  oracle_chain_lanes_reset(l);
  return l;
}

oracle_chain_lanes *oracle_memory_lanes_new(uint64_t num_actions, uint64_t history_len, int limit_kind,
                                            uint64_t max_steps, uint64_t n_lanes, uint64_t lane_offset,
                                            uint64_t seed_env, uint64_t seed_actor) {
  oracle_chain_lanes *l = (oracle_chain_lanes *)calloc(1, sizeof(*l));
  oracle_memory_default(&l->memory);
  l->memory.num_actions = num_actions;
  l->memory.history_len = history_len;
  l->env.size = num_actions + history_len; /* IndexSpace::new(num_actions + history_len), memory.rs:62-64 */
  l->limit_kind = limit_kind;
  l->max_steps = max_steps;
  l->seed_env = seed_env;
  l->seed_actor = seed_actor;
  l->n_lanes = n_lanes;
  l->lane_offset = lane_offset;
  l->state = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->steps_remaining = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->reset_count = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->initial = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->env_pos = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  oracle_chain_lanes_reset(l);
  return l;
}

oracle_chain_lanes *oracle_bandit_lanes_new(const double *values, uint64_t n_lanes, uint64_t lane_offset,
                                            uint64_t seed_env, uint64_t seed_actor) {
  oracle_chain_lanes *l = oracle_chain_lanes_new(5, ORACLE_LIMIT_NONE, 0, n_lanes, lane_offset, seed_env, seed_actor);
  l->env.discount_factor = 1.0; /* bandits.rs:52-54 */
  l->bandit = 1;
  l->bandit_values[0] = values[0];
  l->bandit_values[1] = values[1];
  return l;
}

void oracle_chain_lanes_free(oracle_chain_lanes *l) {
  if (!l) return;
  free(l->state);
  free(l->steps_remaining);
  free(l->reset_count);
  free(l->initial);
  free(l->env_pos);
  free(l);
}

void oracle_memory_lanes_get_extra(const oracle_chain_lanes *l, uint64_t *initial, uint64_t *env_pos) {
  memcpy(initial, l->initial, sizeof(uint64_t) * l->n_lanes);
  memcpy(env_pos, l->env_pos, sizeof(uint64_t) * l->n_lanes);
}

/* a new episode in lane i */
static void index_lane_reset(oracle_chain_lanes *l, uint64_t i) {
  if (l->memory.num_actions) {
    oracle_prng r;
    oracle_prng_seed_from_u64(&r, l->seed_env);
    oracle_prng_set_stream(&r, l->lane_offset + i);
    oracle_prng_set_word_pos(&r, l->env_pos[i]);
    oracle_memory_initial_state(&l->memory, &r, &l->state[i], &l->initial[i]);
    l->env_pos[i] = oracle_prng_word_pos(&r);
  } else {
    l->state[i] = 0; /* Chain::initial_state (chain.rs:75-77) draws nothing */
  }
  l->steps_remaining[i] = l->max_steps;
  l->reset_count[i] += 1;
}

void oracle_chain_lanes_reset(oracle_chain_lanes *l) {
  for (uint64_t i = 0; i < l->n_lanes; ++i) index_lane_reset(l, i);
}

uint32_t oracle_chain_lanes_obs_dim(const oracle_chain_lanes *l) {
  return (uint32_t)l->env.size + (l->limit_kind == ORACLE_LIMIT_VISIBLE ? 1u : 0u);
}

static void chain_features(const oracle_chain_lanes *l, uint64_t i, float *f) {
  oracle_index_features(l->state[i], l->env.size, f);
  if (l->limit_kind == ORACLE_LIMIT_VISIBLE)
    f[l->env.size] = (float)oracle_step_limit_remaining(l->steps_remaining[i], l->max_steps);
}

void oracle_chain_lanes_observe(const oracle_chain_lanes *l, float *obs_soa) {
  uint32_t D = oracle_chain_lanes_obs_dim(l);
  float f[16];
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    chain_features(l, i, f);
    for (uint32_t d = 0; d < D; ++d) obs_soa[d * l->n_lanes + i] = f[d];
  }
}

void oracle_chain_lanes_get_state(const oracle_chain_lanes *l, uint64_t *state, uint64_t *steps_remaining,
                                  uint64_t *reset_count) {
  memcpy(state, l->state, sizeof(uint64_t) * l->n_lanes);
  memcpy(steps_remaining, l->steps_remaining, sizeof(uint64_t) * l->n_lanes);
  memcpy(reset_count, l->reset_count, sizeof(uint64_t) * l->n_lanes);
}

/* one env step of lane i with the slip draw taken from word `word` of the lane's env stream; auto-reset */
static int chain_lane_step(oracle_chain_lanes *l, uint64_t i, int action, oracle_prng *env_rng, uint64_t word,
                           float *reward, float *term_f) {
  double r;
  int succ;
  if (l->bandit) { /* Bandit::step (bandits.rs:66-77): Deterministic::sample draws nothing */
    r = l->bandit_values[action];
    succ = ORACLE_TERMINATE;
  } else if (l->memory.num_actions) {
    succ = oracle_memory_step(&l->memory, &l->state[i], l->initial[i], (uint64_t)action, &r);
  } else {
    oracle_prng_set_word_pos(env_rng, word);
    succ = oracle_chain_step(&l->env, &l->state[i], action, env_rng, &r);
  }
  if (l->limit_kind != ORACLE_LIMIT_NONE) succ = oracle_step_limit_apply(succ, &l->steps_remaining[i]);
  *reward = (float)r;
  if (succ == ORACLE_INTERRUPT && term_f) chain_features(l, i, term_f);
  if (succ != ORACLE_CONTINUE) index_lane_reset(l, i);
  return succ;
}

void oracle_chain_lanes_step(oracle_chain_lanes *l, const uint8_t *actions, float *reward, uint8_t *flag,
                             float *obs_next_soa, float *term_obs_soa) {
  uint32_t D = oracle_chain_lanes_obs_dim(l);
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    oracle_prng r;
    oracle_prng_seed_from_u64(&r, l->seed_env);
    oracle_prng_set_stream(&r, l->lane_offset + i);
    float tf[16], f[16];
    int succ = chain_lane_step(l, i, actions[i], &r, l->t_global, &reward[i], tf);
    flag[i] = (uint8_t)succ;
    if (succ == ORACLE_INTERRUPT && term_obs_soa)
      for (uint32_t d = 0; d < D; ++d) term_obs_soa[d * l->n_lanes + i] = tf[d];
    if (obs_next_soa) {
      chain_features(l, i, f);
      for (uint32_t d = 0; d < D; ++d) obs_next_soa[d * l->n_lanes + i] = f[d];
    }
  }
  l->t_global += 1;
}

/* T env-actor steps with the recurrent policy: PolicyActor::act (policies/actor.rs:42-55) over
 * Chain<Gru, Mlp>::step (chain.rs:175-186).  The actor's episode state starts at zero at the beginning of the
 * call and after every episode end (Steps::step calls actor.initial_state for each new episode,
 * steps.rs:126-131; the engine additionally restarts it at period boundaries so that the update's
 * teacher-forced forward reproduces the behaviour policy exactly). */
void oracle_chain_lanes_rollout_gru(oracle_chain_lanes *l, oracle_gru_shape ps, const float *params, uint64_t T,
                                    float *obs, uint8_t *action, float *reward, uint8_t *flag, float *term_obs,
                                    int n_threads) {
  uint32_t D = oracle_chain_lanes_obs_dim(l);
  uint64_t n = l->n_lanes;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
  for (uint64_t i = 0; i < n; ++i) {
    oracle_prng env_rng, act_rng;
    oracle_prng_seed_from_u64(&env_rng, l->seed_env);
    oracle_prng_set_stream(&env_rng, l->lane_offset + i);
    oracle_prng_seed_from_u64(&act_rng, l->seed_actor);
    oracle_prng_set_stream(&act_rng, l->lane_offset + i);
    oracle_prng_set_word_pos(&act_rng, l->t_global);
    float f[16], tf[16], z[16], lp[16];
    float *h = (float *)calloc((size_t)2 * ps.hidden, sizeof(float)); /* [h] or [h; c] */
    for (uint64_t t = 0; t < T; ++t) {
      chain_features(l, i, f);
      for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + t) * n + i] = f[d];
      float u = oracle_prng_gen_f32(&act_rng);
      oracle_gru_step_f32(ps, params, f, h, z);
      oracle_log_softmax_f32(z, ps.out_dim, lp, 0);
      int a = oracle_categorical_sample_u(lp, ps.out_dim, u, 0);
      float r;
      int succ = chain_lane_step(l, i, a, &env_rng, l->t_global + t, &r, tf);
      action[t * n + i] = (uint8_t)a;
      reward[t * n + i] = r;
      flag[t * n + i] = (uint8_t)succ;
      if (succ == ORACLE_INTERRUPT && term_obs)
        for (uint32_t d = 0; d < D; ++d) term_obs[(d * T + t) * n + i] = tf[d];
      if (succ != ORACLE_CONTINUE) memset(h, 0, sizeof(float) * 2 * ps.hidden);
    }
    chain_features(l, i, f);
    for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + T) * n + i] = f[d];
    free(h);
  }
  l->t_global += T;
}

/* T env-actor steps of Chain lanes with a feed-forward policy (PolicyActor::act over Mlp, policies/actor.rs:42-55) */
void oracle_chain_lanes_rollout_mlp(oracle_chain_lanes *l, oracle_mlp_shape ps, const float *params, uint64_t T,
                                    float *obs, uint8_t *action, float *reward, uint8_t *flag, float *term_obs) {
  uint32_t D = oracle_chain_lanes_obs_dim(l);
  uint64_t n = l->n_lanes;
  for (uint64_t i = 0; i < n; ++i) {
    oracle_prng env_rng, act_rng;
    oracle_prng_seed_from_u64(&env_rng, l->seed_env);
    oracle_prng_set_stream(&env_rng, l->lane_offset + i);
    oracle_prng_seed_from_u64(&act_rng, l->seed_actor);
    oracle_prng_set_stream(&act_rng, l->lane_offset + i);
    oracle_prng_set_word_pos(&act_rng, l->t_global);
    float f[16], tf[16], z[16], lp[16];
    for (uint64_t t = 0; t < T; ++t) {
      chain_features(l, i, f);
      for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + t) * n + i] = f[d];
      float u = oracle_prng_gen_f32(&act_rng);
      oracle_mlp_forward_f32(ps, params, f, z);
      oracle_log_softmax_f32(z, ps.out_dim, lp, 0);
      int a = oracle_categorical_sample_u(lp, ps.out_dim, u, 0);
      float r;
      int succ = chain_lane_step(l, i, a, &env_rng, l->t_global + t, &r, tf);
      action[t * n + i] = (uint8_t)a;
      reward[t * n + i] = r;
      flag[t * n + i] = (uint8_t)succ;
      if (succ == ORACLE_INTERRUPT && term_obs)
        for (uint32_t d = 0; d < D; ++d) term_obs[(d * T + t) * n + i] = tf[d];
    }
    chain_features(l, i, f);
    for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + T) * n + i] = f[d];
  }
  l->t_global += T;
}

/* ------------------------------------------------------------------ GAE with a recurrent critic
 * values[t][lane] = V at obs[t]; succ_values[t][lane] = V of the successor observation where the episode is cut
 * (Interrupt or horizon), both from oracle_gru_seq_forward.  Same arithmetic as oracle_lanes_gae. */
void oracle_seq_gae(uint64_t n, uint64_t T, const float *values, const float *succ_values, const float *reward,
                    const uint8_t *flag, float gamma, float lambda, float *adv_out, float *rtg_out) {
  float disc = lambda * gamma;
  for (uint64_t i = 0; i < n; ++i) {
    float adv_next = 0.0f, rtg_next = 0.0f;
    for (uint64_t t = T; t-- > 0;) {
      uint8_t f = flag[t * n + i];
      float vnext;
      int ends;
      if (f == ORACLE_TERMINATE) {
        vnext = 0.0f;
        ends = 1;
      } else if (f == ORACLE_INTERRUPT || t == T - 1) {
        vnext = succ_values[t * n + i];
        ends = 1;
      } else {
        vnext = values[(t + 1) * n + i];
        ends = 0;
      }
      float r = reward[t * n + i];
      float dn = gamma * vnext;
      float tmp = r + dn;
      float delta = tmp - values[t * n + i];
      float a, g;
      if (ends) {
        a = delta;
        g = r;
      } else {
        float pa = adv_next * disc;
        a = delta + pa;
        float pg = rtg_next * gamma;
        g = r + pg;
      }
      adv_out[t * n + i] = a;
      rtg_out[t * n + i] = g;
      adv_next = a;
      rtg_next = g;
    }
  }
}

/* one_step_values (critics/mod.rs:139-150) with a recurrent critic: the next value comes from the teacher-forced
 * forward (next step's output inside an episode, the successor output where it is cut) */
void oracle_seq_one_step_targets(uint64_t n, uint64_t T, const float *values, const float *succ_values,
                                 const float *reward, const uint8_t *flag, float gamma, float *out) {
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t t = 0; t < T; ++t) {
      uint8_t f = flag[t * n + i];
      float vnext;
      if (f == ORACLE_TERMINATE) vnext = 0.0f;
      else if (f == ORACLE_INTERRUPT || t == T - 1) vnext = succ_values[t * n + i];
      else vnext = values[(t + 1) * n + i];
      float dn = gamma * vnext;
      out[t * n + i] = reward[t * n + i] + dn;
    }
}

/* ------------------------------------------------------------------ per-sample output gradients
 * logits [2][T][n] -> d loss / d logits for the policy losses, and log-probs / entropy / loss sums.
 * mode 0: surrogate at ratio 1 (REINFORCE / first TRPO / PPO gradient), loss = -mean(A)
 * mode 1: PPO clipped surrogate against logp0 (policies/ppo.rs:124-137) */
void oracle_seq_policy_dlogits_f32(uint64_t B, const float *logits, const uint8_t *actions, const float *adv,
                                   const float *logp0, int mode, float clip_lo, float clip_hi, float *dlogits,
                                   float *logp_out, double *loss_sum_out, double *entropy_sum_out) {
  float inv_B = 1.0f / (float)B;
  double loss = 0.0, ent = 0.0;
  for (uint64_t b = 0; b < B; ++b) {
    float z[2] = {logits[b], logits[B + b]}, lp[2];
    oracle_log_softmax_f32(z, 2, lp, 0);
    int a = actions[b];
    float p0 = rl_expf(lp[0]), p1 = rl_expf(lp[1]);
    float c;
    if (mode == 0) {
      float ratio = rl_expf(lp[a] - lp[a]);
      c = -(ratio * adv[b]) * inv_B;
      loss += (double)(ratio * adv[b]);
      float cl0 = lp[0] < -FLT_MAX ? -FLT_MAX : lp[0], cl1 = lp[1] < -FLT_MAX ? -FLT_MAX : lp[1];
      float e = cl0 * p0;
      e += cl1 * p1;
      ent += (double)(-e);
    } else {
      float ratio = rl_expf(lp[a] - logp0[b]);
      float clipped = ratio < clip_lo ? clip_lo : (ratio > clip_hi ? clip_hi : ratio);
      float u1 = ratio * adv[b], u2 = clipped * adv[b];
      int inside = ratio >= clip_lo && ratio <= clip_hi;
      float gr = u1 < u2 ? adv[b] : (u1 > u2 ? (inside ? adv[b] : 0.0f) : (inside ? adv[b] : 0.5f * adv[b]));
      c = -(gr * ratio) * inv_B;
      loss += (double)(u1 < u2 ? u1 : u2);
    }
    dlogits[b] = c * ((a == 0 ? 1.0f : 0.0f) - p0);
    dlogits[B + b] = c * ((a == 1 ? 1.0f : 0.0f) - p1);
    if (logp_out) logp_out[b] = lp[a];
  }
  if (loss_sum_out) *loss_sum_out = loss;
  if (entropy_sum_out) *entropy_sum_out = ent;
}
