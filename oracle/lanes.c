/* lanes.c — the engine's vectorised rollout semantics restated with the scalar reference pieces,
 * and the train_parallel-structured CPU baseline.  TEST INFRASTRUCTURE (see oracle.h).
 *
 * Lane model (documented in DESIGN.md §Data layout): lane g (global id) is one
 * `Steps` iterator (src/simulation/steps.rs:113-167) over CartPole wrapped in a step limit, with
 *   env stream   = ChaCha8Rng::seed_from_u64(seed_env),  set_stream(g); the k-th reset of the lane
 *                  reads its 4 initial-state draws at word position 8k;
 *   actor stream = ChaCha8Rng::seed_from_u64(seed_actor), set_stream(g); the action of global step t
 *                  uses the single `gen::<f32>()` at word position t.
 * A lane that ends an episode starts the next one immediately (Steps::step does the same on its next call).
 */
#include "oracle.h"
#include "../include/rl_detmath.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

oracle_lanes *oracle_lanes_new(const oracle_cartpole *env, int limit_kind, uint64_t max_steps, uint64_t n_lanes,
                               uint64_t lane_offset, uint64_t seed_env, uint64_t seed_actor) {
  oracle_lanes *l = (oracle_lanes *)calloc(1, sizeof(*l));
  l->env = *env;
  l->limit_kind = limit_kind;
  l->max_steps = max_steps;
  l->seed_env = seed_env;
  l->seed_actor = seed_actor;
  l->n_lanes = n_lanes;
  l->lane_offset = lane_offset;
  l->state = (oracle_cartpole_state *)calloc(n_lanes, sizeof(oracle_cartpole_state));
  l->steps_remaining = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  l->reset_count = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  oracle_lanes_reset(l);
  return l;
}

void oracle_lanes_free(oracle_lanes *l) {
  if (!l) return;
  free(l->state);
  free(l->steps_remaining);
  free(l->reset_count);
  free(l);
}

static void lane_reset(oracle_lanes *l, uint64_t i, oracle_prng *env_rng) {
  oracle_prng_set_word_pos(env_rng, 8 * l->reset_count[i]);
  oracle_cartpole_initial_state(&l->env, env_rng, &l->state[i]);
  l->steps_remaining[i] = l->max_steps; /* Wrapped::initial_state (step_limit.rs:57-62, 187-192) */
  l->reset_count[i] += 1;
}

static void lane_env_rng(const oracle_lanes *l, uint64_t i, oracle_prng *r) {
  oracle_prng_seed_from_u64(r, l->seed_env);
  oracle_prng_set_stream(r, l->lane_offset + i);
}

void oracle_lanes_reset(oracle_lanes *l) {
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    oracle_prng r;
    lane_env_rng(l, i, &r);
    lane_reset(l, i, &r);
  }
}

void oracle_lanes_get_state(const oracle_lanes *l, double *state4, int32_t *nv_pos, uint64_t *steps_remaining,
                            uint64_t *reset_count) {
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    state4[0 * l->n_lanes + i] = l->state[i].x;
    state4[1 * l->n_lanes + i] = l->state[i].xdot;
    state4[2 * l->n_lanes + i] = l->state[i].th;
    state4[3 * l->n_lanes + i] = l->state[i].thdot;
    nv_pos[i] = l->state[i].nv_pos;
    steps_remaining[i] = l->steps_remaining[i];
    reset_count[i] = l->reset_count[i];
  }
}

void oracle_lanes_set_state(oracle_lanes *l, const double *state4, const int32_t *nv_pos,
                            const uint64_t *steps_remaining, const uint64_t *reset_count) {
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    l->state[i].x = state4[0 * l->n_lanes + i];
    l->state[i].xdot = state4[1 * l->n_lanes + i];
    l->state[i].th = state4[2 * l->n_lanes + i];
    l->state[i].thdot = state4[3 * l->n_lanes + i];
    l->state[i].nv_pos = nv_pos[i];
    l->steps_remaining[i] = steps_remaining[i];
    l->reset_count[i] = reset_count[i];
  }
}

static uint32_t lanes_obs_dim(const oracle_lanes *l) { return l->limit_kind == ORACLE_LIMIT_VISIBLE ? 5 : 4; }

void oracle_lanes_observe(const oracle_lanes *l, float *obs_soa) {
  uint32_t D = lanes_obs_dim(l);
  float f[5];
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
    for (uint32_t d = 0; d < D; ++d) obs_soa[d * l->n_lanes + i] = f[d];
  }
}

/* env.step of the wrapped env for one lane + auto-reset; returns the successor code */
static int lane_step(oracle_lanes *l, uint64_t i, int action, oracle_prng *env_rng, float *reward, float *term_f) {
  double r;
  int succ = oracle_cartpole_step(&l->env, &l->state[i], action, &r);
  if (l->limit_kind != ORACLE_LIMIT_NONE) succ = oracle_step_limit_apply(succ, &l->steps_remaining[i]);
  *reward = (float)r; /* f64 reward -> f32 tensor element (features.rs:202) */
  if (succ == ORACLE_INTERRUPT && term_f)
    oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, term_f);
  if (succ != ORACLE_CONTINUE) lane_reset(l, i, env_rng);
  return succ;
}

void oracle_lanes_step(oracle_lanes *l, const uint8_t *actions, float *reward, uint8_t *flag, float *obs_next_soa,
                       float *term_obs_soa) {
  uint32_t D = lanes_obs_dim(l);
  for (uint64_t i = 0; i < l->n_lanes; ++i) {
    oracle_prng r;
    lane_env_rng(l, i, &r);
    float tf[5], f[5];
    int succ = lane_step(l, i, actions[i], &r, &reward[i], tf);
    flag[i] = (uint8_t)succ;
    if (succ == ORACLE_INTERRUPT && term_obs_soa)
      for (uint32_t d = 0; d < D; ++d) term_obs_soa[d * l->n_lanes + i] = tf[d];
    if (obs_next_soa) {
      oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
      for (uint32_t d = 0; d < D; ++d) obs_next_soa[d * l->n_lanes + i] = f[d];
    }
  }
  l->t_global += 1;
}

/* PolicyActor::act (torch/agents/policies/actor.rs:42-55): features -> Mlp::step -> Categorical -> sample */
static int policy_act(oracle_mlp_shape ps, const float *params, const float *feat, float u) {
  float z[16], lp[16];
  oracle_mlp_forward_f32(ps, params, feat, z);
  oracle_log_softmax_f32(z, ps.out_dim, lp, 0);
  return oracle_categorical_sample_u(lp, ps.out_dim, u, 0);
}

void oracle_lanes_rollout(oracle_lanes *l, oracle_mlp_shape ps, const float *policy_params, uint64_t T, float *obs,
                          uint8_t *action, float *reward, uint8_t *flag, float *term_obs, int n_threads) {
  uint32_t D = lanes_obs_dim(l);
  uint64_t n = l->n_lanes;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
  for (uint64_t i = 0; i < n; ++i) {
    oracle_prng env_rng, act_rng;
    lane_env_rng(l, i, &env_rng);
    oracle_prng_seed_from_u64(&act_rng, l->seed_actor);
    oracle_prng_set_stream(&act_rng, l->lane_offset + i);
    oracle_prng_set_word_pos(&act_rng, l->t_global);
    float f[5], tf[5];
    for (uint64_t t = 0; t < T; ++t) {
      oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
      for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + t) * n + i] = f[d];
      float u = oracle_prng_gen_f32(&act_rng);
      int a = policy_act(ps, policy_params, f, u);
      float r;
      int succ = lane_step(l, i, a, &env_rng, &r, tf);
      action[t * n + i] = (uint8_t)a;
      reward[t * n + i] = r;
      flag[t * n + i] = (uint8_t)succ;
      if (succ == ORACLE_INTERRUPT && term_obs)
        for (uint32_t d = 0; d < D; ++d) term_obs[(d * T + t) * n + i] = tf[d];
    }
    oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
    for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + T) * n + i] = f[d];
  }
  l->t_global += T;
}

/* T env-actor steps of CartPole lanes with the recurrent policy (Chain<Gru, Mlp>::step, modules/chain.rs:175-186):
 * as oracle_lanes_rollout, the actor's episode state restarting at zero at the start of the call and after every
 * episode end. */
void oracle_lanes_rollout_gru(oracle_lanes *l, oracle_gru_shape ps, const float *params, uint64_t T, float *obs,
                              uint8_t *action, float *reward, uint8_t *flag, float *term_obs, int n_threads) {
  uint32_t D = lanes_obs_dim(l);
  uint64_t n = l->n_lanes;
  (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
  for (uint64_t i = 0; i < n; ++i) {
    oracle_prng env_rng, act_rng;
    lane_env_rng(l, i, &env_rng);
    oracle_prng_seed_from_u64(&act_rng, l->seed_actor);
    oracle_prng_set_stream(&act_rng, l->lane_offset + i);
    oracle_prng_set_word_pos(&act_rng, l->t_global);
    float f[5], tf[5], z[16], lp[16];
    float *h = (float *)calloc((size_t)2 * ps.hidden, sizeof(float)); /* [h] or [h; c] */
    for (uint64_t t = 0; t < T; ++t) {
      oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
      for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + t) * n + i] = f[d];
      float u = oracle_prng_gen_f32(&act_rng);
      oracle_gru_step_f32(ps, params, f, h, z);
      oracle_log_softmax_f32(z, ps.out_dim, lp, 0);
      int a = oracle_categorical_sample_u(lp, ps.out_dim, u, 0);
      float r;
      int succ = lane_step(l, i, a, &env_rng, &r, tf);
      action[t * n + i] = (uint8_t)a;
      reward[t * n + i] = r;
      flag[t * n + i] = (uint8_t)succ;
      if (succ == ORACLE_INTERRUPT && term_obs)
        for (uint32_t d = 0; d < D; ++d) term_obs[(d * T + t) * n + i] = tf[d];
      if (succ != ORACLE_CONTINUE) memset(h, 0, sizeof(float) * 2 * ps.hidden);
    }
    oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
    for (uint32_t d = 0; d < D; ++d) obs[(d * (T + 1) + T) * n + i] = f[d];
    free(h);
  }
  l->t_global += T;
}

/* Lane-major GAE and reward-to-go.  Same arithmetic as critics/mod.rs:158-199 + packed.rs:312-342
 * (delta = (r + gamma*V') - V with each op rounded; a = a + (b * discount)), applied along each lane;
 * an episode cut by the horizon is an Interrupt whose successor observation is obs[T]. */
void oracle_lanes_gae(oracle_mlp_shape cs, const float *critic_params, uint64_t n, uint64_t T, uint32_t D,
                      const float *obs, const float *reward, const uint8_t *flag, const float *term_obs, float gamma,
                      float lambda, float *values_out, float *adv_out, float *rtg_out) {
  float disc = lambda * gamma;
  for (uint64_t i = 0; i < n; ++i) {
    float x[8], v;
    for (uint64_t t = 0; t <= T; ++t) {
      for (uint32_t d = 0; d < D; ++d) x[d] = obs[(d * (T + 1) + t) * n + i];
      oracle_mlp_forward_f32(cs, critic_params, x, &v);
      values_out[t * n + i] = v;
    }
    float adv_next = 0.0f, rtg_next = 0.0f;
    for (uint64_t t = T; t-- > 0;) {
      uint8_t f = flag[t * n + i];
      float vnext;
      int ends;
      if (f == ORACLE_TERMINATE) {
        vnext = 0.0f;
        ends = 1;
      } else if (f == ORACLE_INTERRUPT) {
        for (uint32_t d = 0; d < D; ++d) x[d] = term_obs[(d * T + t) * n + i];
        oracle_mlp_forward_f32(cs, critic_params, x, &vnext);
        ends = 1;
      } else {
        vnext = values_out[(t + 1) * n + i];
        ends = (t == T - 1);
      }
      float r = reward[t * n + i];
      float dn = gamma * vnext;
      float tmp = r + dn;
      float delta = tmp - values_out[t * n + i];
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

/* one_step_values (critics/mod.rs:139-150) on the lane layout, the engine's horizon rule (cut = Interrupt(obs[T])):
 * target[t] = r[t] + gamma * V(successor of step t); Terminate -> 0, Interrupt -> V(term_obs) */
void oracle_lanes_one_step_targets(oracle_mlp_shape cs, const float *critic_params, uint64_t n, uint64_t T, uint32_t D,
                                   const float *obs, const float *reward, const uint8_t *flag, const float *term_obs,
                                   float gamma, float *out) {
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t t = 0; t < T; ++t) {
      float x[8], vnext;
      uint8_t f = flag[t * n + i];
      if (f == ORACLE_TERMINATE) {
        vnext = 0.0f;
      } else if (f == ORACLE_INTERRUPT) {
        for (uint32_t d = 0; d < D; ++d) x[d] = term_obs[(d * T + t) * n + i];
        oracle_mlp_forward_f32(cs, critic_params, x, &vnext);
      } else {
        for (uint32_t d = 0; d < D; ++d) x[d] = obs[(d * (T + 1) + t + 1) * n + i];
        oracle_mlp_forward_f32(cs, critic_params, x, &vnext);
      }
      float dn = gamma * vnext;
      out[t * n + i] = reward[t * n + i] + dn;
    }
}

oracle_vecbuffer *oracle_lanes_to_vecbuffer(uint64_t n, uint64_t T, uint32_t D, const float *obs,
                                            const uint8_t *action, const float *reward, const uint8_t *flag,
                                            const float *term_obs, int keep_last, uint64_t *lane_t_index_out) {
  oracle_vecbuffer *b = oracle_vecbuffer_new(D);
  float x[8], nx[8];
  for (uint64_t i = 0; i < n; ++i) {
    for (uint64_t t = 0; t < T; ++t) {
      for (uint32_t d = 0; d < D; ++d) x[d] = obs[(d * (T + 1) + t) * n + i];
      int f = flag[t * n + i];
      const float *np = NULL;
      if (f == ORACLE_INTERRUPT) {
        for (uint32_t d = 0; d < D; ++d) nx[d] = term_obs[(d * T + t) * n + i];
        np = nx;
      } else if (f == ORACLE_CONTINUE && t == T - 1 && keep_last) {
        for (uint32_t d = 0; d < D; ++d) nx[d] = obs[(d * (T + 1) + T) * n + i];
        np = nx;
        f = ORACLE_INTERRUPT;
      }
      if (lane_t_index_out) lane_t_index_out[b->len] = i * T + t;
      oracle_vecbuffer_write_step(b, x, action[t * n + i], (double)reward[t * n + i], f, np);
    }
    oracle_vecbuffer_end_experience(b); /* each lane is one thread of experience */
  }
  return b;
}

/* ------------------------------------------------------------------ CPU baseline */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* HOT LOOP A alone, long enough to time: every worker thread runs `steps_per_thread` scalar Steps::step calls
 * (steps.rs:113-167; PolicyActor::act as a batch-1 MLP forward) into its own VecBuffer, which is cleared every
 * `clear_every` steps (a period's worth: memory stays bounded).  Returns the wall time of the stepping (threads started
 * before the clock) and the total number of steps taken. */
double oracle_cartpole_rollout_only(uint64_t seed, uint32_t n_threads, uint64_t steps_per_thread, uint64_t clear_every,
                                    uint64_t max_steps, uint32_t hidden, const float *policy_params,
                                    uint64_t *steps_out) {
  oracle_cartpole env;
  oracle_cartpole_default(&env);
  oracle_mlp_shape ps = {5, hidden, 2};
  oracle_prng *t_env = (oracle_prng *)malloc(n_threads * sizeof(oracle_prng));
  oracle_prng *t_agent = (oracle_prng *)malloc(n_threads * sizeof(oracle_prng));
  uint64_t *counts = (uint64_t *)calloc(n_threads, sizeof(uint64_t));
  oracle_prng root, rng_env;
  oracle_prng_seed_from_u64(&root, seed);
  oracle_prng_from_rng(&rng_env, &root);
  for (uint32_t i = 0; i < n_threads; ++i) {
    oracle_prng_from_rng(&t_env[i], &rng_env);
    oracle_prng_from_rng(&t_agent[i], &root);
  }
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
  { (void)omp_get_thread_num(); }
#endif
  double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
#endif
  for (uint32_t i = 0; i < n_threads; ++i) {
    oracle_vecbuffer *buf = oracle_vecbuffer_new(5);
    int have_state = 0;
    oracle_cartpole_state s;
    uint64_t remaining = 0, count = 0;  /* (counted locally: a shared array of counters would bounce its cache line
                                           between the cores on every step) */
    oracle_prng my_env = t_env[i], my_agent = t_agent[i];  /* generators on the thread's own stack, for the same reason */
    float f[5], nf[5];
    for (uint64_t n = 0; n < steps_per_thread; ++n) {
      if (!have_state) {
        oracle_cartpole_initial_state(&env, &my_env, &s);
        remaining = max_steps;
        have_state = 1;
      }
      oracle_cartpole_features(&s, ORACLE_LIMIT_VISIBLE, remaining, max_steps, f);
      float u = oracle_prng_gen_f32(&my_agent);
      int a = policy_act(ps, policy_params, f, u);
      double r;
      int succ = oracle_cartpole_step(&env, &s, a, &r);
      succ = oracle_step_limit_apply(succ, &remaining);
      const float *np = NULL;
      if (succ == ORACLE_INTERRUPT) {
        oracle_cartpole_features(&s, ORACLE_LIMIT_VISIBLE, remaining, max_steps, nf);
        np = nf;
      }
      if (succ != ORACLE_CONTINUE) have_state = 0;
      oracle_vecbuffer_write_step(buf, f, a, r, succ, np);
      count += 1;
      if (clear_every && (n + 1) % clear_every == 0) oracle_vecbuffer_clear(buf);
    }
    counts[i] = count;
    oracle_vecbuffer_free(buf);
  }
  double secs = now_s() - t0;
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_threads; ++i) total += counts[i];
  if (steps_out) *steps_out = total;
  free(counts);
  free(t_env);
  free(t_agent);
  return secs;
}

/* The update's full-batch passes with every pass split over the cores (chunks of samples under `omp for`, chunk results
 * combined in f64 by sample weight): what libtorch's intra-op thread pool does to the reference's batched matmuls while
 * the agent itself runs on one Rust thread (torch/agents/actor_critic.rs:176-211).  Same pass COUNT as the
 * single-threaded update just made (n_evals line-search evaluations), same per-sample arithmetic; the parameters it
 * steps are private copies.  Returns seconds. */
#define INTRAOP_CHUNK 1024
static double update_intraop_seconds(oracle_mlp_shape ps, oracle_mlp_shape cs, const float *policy_params,
                                     const float *critic_params, const oracle_features *feat, const float *adv,
                                     const float *rtg, uint64_t cg_iterations, uint64_t n_evals, uint64_t critic_steps,
                                     uint32_t n_threads) {
  const uint64_t n = feat->n_steps, Pp = oracle_mlp_num_params(ps), Pc = oracle_mlp_num_params(cs);
  const uint64_t n_chunks = (n + INTRAOP_CHUNK - 1) / INTRAOP_CHUNK;
  float *pp = (float *)malloc(sizeof(float) * Pp), *cp = (float *)malloc(sizeof(float) * Pc);
  float *v = (float *)malloc(sizeof(float) * Pp);
  double *acc = (double *)malloc(sizeof(double) * (Pp > Pc ? Pp : Pc));
  float *g = (float *)malloc(sizeof(float) * (Pp > Pc ? Pp : Pc));
  memcpy(pp, policy_params, sizeof(float) * Pp);
  memcpy(cp, critic_params, sizeof(float) * Pc);
  for (uint64_t i = 0; i < Pp; ++i) v[i] = 1e-3f * (float)((i * 2654435761u) % 1000u) / 1000.0f;
  oracle_adam_state *opt = oracle_adam_new(Pc);
  oracle_adam_cfg acfg;
  oracle_adam_cfg_default(&acfg);
  /* kind 0: policy gradient, 1: Fisher-vector product, 2: (loss, KL) evaluation, 3: critic gradient */
  #define INTRAOP_PASS(kind, P)                                                                              \
    do {                                                                                                     \
      for (uint64_t i = 0; i < (P); ++i) acc[i] = 0.0;                                                       \
      _Pragma("omp parallel num_threads(n_threads)")                                                         \
      {                                                                                                      \
        double *mine = (double *)calloc((P), sizeof(double));                                                \
        float *gc = (float *)malloc(sizeof(float) * (P));                                                    \
        _Pragma("omp for schedule(dynamic, 1)")                                                              \
        for (uint64_t c = 0; c < n_chunks; ++c) {                                                            \
          const uint64_t lo = c * INTRAOP_CHUNK, m = lo + INTRAOP_CHUNK <= n ? INTRAOP_CHUNK : n - lo;       \
          float l0 = 0.0f, l1 = 0.0f;                                                                        \
          if ((kind) == 0) oracle_policy_grad_f32(ps, pp, feat->obs + lo * 5, feat->actions + lo, adv + lo, m, gc, &l0); \
          else if ((kind) == 1) oracle_policy_fvp_f32(ps, pp, feat->obs + lo * 5, m, v, 0.0f, gc);           \
          else if ((kind) == 2) {                                                                            \
            oracle_policy_loss_kl_f32(ps, pp, policy_params, feat->obs + lo * 5, feat->actions + lo, adv + lo, m, &l0, &l1); \
            gc[0] = l0;                                                                                      \
            gc[1] = l1;                                                                                      \
          } else oracle_critic_grad_f32(cs, cp, feat->obs + lo * 5, rtg + lo, m, gc, &l0);                   \
          const double w = (double)m / (double)n;                                                            \
          const uint64_t cnt = (kind) == 2 ? 2 : (P);                                                        \
          for (uint64_t i = 0; i < cnt; ++i) mine[i] += w * (double)gc[i];                                   \
        }                                                                                                    \
        _Pragma("omp critical")                                                                              \
        for (uint64_t i = 0; i < (P); ++i) acc[i] += mine[i];                                                \
        free(gc);                                                                                            \
        free(mine);                                                                                          \
      }                                                                                                      \
      for (uint64_t i = 0; i < (P); ++i) g[i] = (float)acc[i];                                               \
    } while (0)
  const double t0 = now_s();
  INTRAOP_PASS(0, Pp);
  for (uint64_t it = 0; it < cg_iterations + 1; ++it) {
    INTRAOP_PASS(1, Pp);
    for (uint64_t i = 0; i < Pp; ++i) v[i] = 0.5f * v[i] + 0.5f * g[i]; /* (a vector update per iteration, like CG's) */
  }
  for (uint64_t e = 0; e < n_evals; ++e) INTRAOP_PASS(2, Pp);
  for (uint64_t k = 0; k < critic_steps; ++k) {
    INTRAOP_PASS(3, Pc);
    oracle_adam_step_f32(opt, &acfg, cp, g);
  }
  const double secs = now_s() - t0;
  #undef INTRAOP_PASS
  oracle_adam_free(opt);
  free(g);
  free(acc);
  free(v);
  free(cp);
  free(pp);
  return secs;
}

void oracle_cartpole_trpo_period(uint64_t seed, uint64_t period_index, uint32_t n_threads, uint64_t steps_per_thread,
                                 uint64_t slack_steps, uint64_t max_steps, uint32_t hidden, float *policy_params,
                                 float *critic_params, oracle_adam_state *critic_opt, uint64_t critic_steps,
                                 oracle_period_stats *stats) {
  oracle_cartpole_trpo_period_ex(seed, period_index, n_threads, steps_per_thread, slack_steps, max_steps, hidden,
                                 policy_params, critic_params, critic_opt, critic_steps, 0, stats);
}

/* the collection half of a period: `n_threads` workers of TakeAlignedSteps over Steps::step, packed into one batch */
static oracle_features *collect_period_sample(uint64_t seed, uint64_t period_index, uint32_t n_threads,
                                              uint64_t steps_per_thread, uint64_t slack_steps, uint64_t max_steps,
                                              oracle_mlp_shape ps, const float *policy_params,
                                              double *rollout_seconds) {
  oracle_cartpole env;
  oracle_cartpole_default(&env);
  oracle_vecbuffer **buffers = (oracle_vecbuffer **)malloc(n_threads * sizeof(*buffers));
  oracle_prng *t_env = (oracle_prng *)malloc(n_threads * sizeof(oracle_prng));
  oracle_prng *t_agent = (oracle_prng *)malloc(n_threads * sizeof(oracle_prng));
  /* per-thread generators forked with from_rng (train.rs:99-106); re-derived per period from
   * (seed, period) because this harness keeps no state between calls */
  oracle_prng root, rng_env;
  oracle_prng_seed_from_u64(&root, seed + 0x9e3779b97f4a7c15ULL * period_index);
  oracle_prng_from_rng(&rng_env, &root);
  for (uint32_t i = 0; i < n_threads; ++i) {
    oracle_prng_from_rng(&t_env[i], &rng_env);
    oracle_prng_from_rng(&t_agent[i], &root);
    buffers[i] = oracle_vecbuffer_new(5);
  }
#ifdef _OPENMP
  /* the worker threads exist before the clock starts (train_parallel's threads are spawned per period, train.rs:124;
   * what is timed here is their stepping, not the OpenMP runtime's pool start-up) */
#pragma omp parallel num_threads(n_threads)
  { (void)omp_get_thread_num(); }
#endif
  double t0 = now_s();
  uint64_t n_take_max = steps_per_thread + slack_steps;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
#endif
  for (uint32_t i = 0; i < n_threads; ++i) {
    /* TakeAlignedSteps over Steps::step (take_steps.rs:80-93; steps.rs:113-167) */
    uint64_t n = steps_per_thread == 0 ? 0 : n_take_max;
    int have_state = 0;
    oracle_cartpole_state s;
    uint64_t remaining = 0;
    float f[5], nf[5];
    while (n != 0) {
      if (!have_state) {
        oracle_cartpole_initial_state(&env, &t_env[i], &s);
        remaining = max_steps;
        have_state = 1;
      }
      oracle_cartpole_features(&s, ORACLE_LIMIT_VISIBLE, remaining, max_steps, f);
      float u = oracle_prng_gen_f32(&t_agent[i]);
      int a = policy_act(ps, policy_params, f, u);
      double r;
      int succ = oracle_cartpole_step(&env, &s, a, &r);
      succ = oracle_step_limit_apply(succ, &remaining);
      const float *np = NULL;
      if (succ == ORACLE_INTERRUPT) {
        oracle_cartpole_features(&s, ORACLE_LIMIT_VISIBLE, remaining, max_steps, nf);
        np = nf;
      }
      if (succ != ORACLE_CONTINUE) have_state = 0;
      oracle_vecbuffer_write_step(buffers[i], f, a, r, succ, np);
      n -= 1;
      if (succ != ORACLE_CONTINUE && n <= slack_steps) n = 0;
    }
    oracle_vecbuffer_end_experience(buffers[i]);
  }
  *rollout_seconds = now_s() - t0;
  oracle_features *feat = oracle_features_from_buffers(buffers, n_threads);
  for (uint32_t i = 0; i < n_threads; ++i) oracle_vecbuffer_free(buffers[i]);
  free(buffers);
  free(t_env);
  free(t_agent);
  return feat;
}

void oracle_cartpole_trpo_period_ex(uint64_t seed, uint64_t period_index, uint32_t n_threads, uint64_t steps_per_thread,
                                    uint64_t slack_steps, uint64_t max_steps, uint32_t hidden, float *policy_params,
                                    float *critic_params, oracle_adam_state *critic_opt, uint64_t critic_steps,
                                    uint32_t intraop_threads, oracle_period_stats *stats) {
  oracle_cartpole env;
  oracle_cartpole_default(&env);
  oracle_mlp_shape ps = {5, hidden, 2}, cs = {5, hidden, 1};
  memset(stats, 0, sizeof(*stats));
  oracle_features *feat = collect_period_sample(seed, period_index, n_threads, steps_per_thread, slack_steps, max_steps,
                                                ps, policy_params, &stats->rollout_seconds);
  double t1 = now_s();
  /* ActorCriticAgent::batch_update_slice (torch/agents/actor_critic.rs:176-211), single thread */
  stats->steps = feat->n_steps;
  stats->episodes = feat->n_episodes;
  stats->mean_episode_length = feat->n_episodes ? (double)feat->n_steps / (double)feat->n_episodes : 0.0;
  float gamma = (float)fmin(0.99, env.discount_factor); /* critics/opt.rs:73 */
  float *adv = (float *)malloc(sizeof(float) * (feat->n_steps ? feat->n_steps : 1));
  float *rtg = (float *)malloc(sizeof(float) * (feat->n_steps ? feat->n_steps : 1));
  oracle_gae_packed(cs, critic_params, feat, gamma, 0.95f, adv, NULL);
  oracle_trpo_cfg cfg;
  oracle_trpo_cfg_default(&cfg);
  oracle_trpo_update_f32(ps, policy_params, feat->obs, feat->actions, adv, feat->n_steps, &cfg, &stats->trpo, NULL);
  oracle_reward_to_go_packed(feat, gamma, rtg);
  oracle_adam_cfg acfg;
  oracle_adam_cfg_default(&acfg);
  float *losses = (float *)malloc(sizeof(float) * (critic_steps ? critic_steps : 1));
  oracle_critic_update_f32(cs, critic_params, critic_opt, &acfg, feat->obs, rtg, feat->n_steps, critic_steps, losses);
  if (critic_steps) {
    stats->critic_loss_first = losses[0];
    stats->critic_loss_last = losses[critic_steps - 1];
  }
  stats->update_seconds = now_s() - t1;
  if (intraop_threads > 0 && feat->n_steps > 0) {
    const uint64_t n_evals = stats->trpo.num_backtracks >= 0 ? (uint64_t)stats->trpo.num_backtracks + 1 : cfg.max_backtracks;
    stats->update_intraop_seconds = update_intraop_seconds(ps, cs, policy_params, critic_params, feat, adv, rtg,
                                                           cfg.iterations, n_evals, critic_steps, intraop_threads);
    stats->update_intraop_threads = intraop_threads;
  }
  free(losses);
  free(adv);
  free(rtg);
  oracle_features_free(feat);
}

/* ---- the same period with its legs callable one at a time (bench.py's cpu_baseline: the update timed on one thread
 * over a PREFIX of the sample, and with its passes split over several thread counts, so that the count that is fastest
 * on the box at hand is measured rather than assumed).  The update is linear in the number of samples: every pass
 * visits every sample once.  Parameters are never changed: every leg works on private copies. */
struct oracle_cpu_sample {
  oracle_mlp_shape ps, cs;
  oracle_features *feat;
  float *adv, *rtg, *policy_params, *critic_params;
  uint64_t critic_steps, n_evals, cg_iterations;
};

oracle_cpu_sample *oracle_cpu_sample_collect(uint64_t seed, uint32_t n_threads, uint64_t steps_per_thread,
                                             uint64_t slack_steps, uint64_t max_steps, uint32_t hidden,
                                             const float *policy_params, const float *critic_params,
                                             oracle_period_stats *stats) {
  oracle_cartpole env;
  oracle_cartpole_default(&env);
  oracle_cpu_sample *s = (oracle_cpu_sample *)calloc(1, sizeof(*s));
  s->ps = (oracle_mlp_shape){5, hidden, 2};
  s->cs = (oracle_mlp_shape){5, hidden, 1};
  memset(stats, 0, sizeof(*stats));
  s->feat = collect_period_sample(seed, 0, n_threads, steps_per_thread, slack_steps, max_steps, s->ps, policy_params,
                                  &stats->rollout_seconds);
  stats->steps = s->feat->n_steps;
  stats->episodes = s->feat->n_episodes;
  stats->mean_episode_length = s->feat->n_episodes ? (double)s->feat->n_steps / (double)s->feat->n_episodes : 0.0;
  const uint64_t n = s->feat->n_steps ? s->feat->n_steps : 1, Pp = oracle_mlp_num_params(s->ps),
                 Pc = oracle_mlp_num_params(s->cs);
  s->adv = (float *)malloc(sizeof(float) * n);
  s->rtg = (float *)malloc(sizeof(float) * n);
  s->policy_params = (float *)malloc(sizeof(float) * Pp);
  s->critic_params = (float *)malloc(sizeof(float) * Pc);
  memcpy(s->policy_params, policy_params, sizeof(float) * Pp);
  memcpy(s->critic_params, critic_params, sizeof(float) * Pc);
  const float gamma = (float)fmin(0.99, env.discount_factor);
  oracle_gae_packed(s->cs, critic_params, s->feat, gamma, 0.95f, s->adv, NULL);
  oracle_reward_to_go_packed(s->feat, gamma, s->rtg);
  oracle_trpo_cfg cfg;
  oracle_trpo_cfg_default(&cfg);
  s->cg_iterations = cfg.iterations;
  s->n_evals = 1;
  return s;
}

/* GAE's value pass excluded (it ran at collection); TRPO + `critic_steps` Adam steps on the first `n_prefix` samples, one
 * thread, as ActorCriticAgent::batch_update_slice runs them.  Returns seconds; fills the TRPO statistics. */
double oracle_cpu_sample_update_one_thread(oracle_cpu_sample *s, uint64_t n_prefix, uint64_t critic_steps,
                                           oracle_period_stats *stats) {
  const uint64_t n = n_prefix && n_prefix < s->feat->n_steps ? n_prefix : s->feat->n_steps;
  const uint64_t Pp = oracle_mlp_num_params(s->ps), Pc = oracle_mlp_num_params(s->cs);
  float *pp = (float *)malloc(sizeof(float) * Pp), *cp = (float *)malloc(sizeof(float) * Pc);
  memcpy(pp, s->policy_params, sizeof(float) * Pp);
  memcpy(cp, s->critic_params, sizeof(float) * Pc);
  oracle_adam_state *opt = oracle_adam_new(Pc);
  oracle_adam_cfg acfg;
  oracle_adam_cfg_default(&acfg);
  oracle_trpo_cfg cfg;
  oracle_trpo_cfg_default(&cfg);
  float *losses = (float *)malloc(sizeof(float) * (critic_steps ? critic_steps : 1));
  const double t0 = now_s();
  oracle_trpo_update_f32(s->ps, pp, s->feat->obs, s->feat->actions, s->adv, n, &cfg, &stats->trpo, NULL);
  oracle_critic_update_f32(s->cs, cp, opt, &acfg, s->feat->obs, s->rtg, n, critic_steps, losses);
  const double secs = now_s() - t0;
  if (critic_steps) {
    stats->critic_loss_first = losses[0];
    stats->critic_loss_last = losses[critic_steps - 1];
  }
  s->n_evals = stats->trpo.num_backtracks >= 0 ? (uint64_t)stats->trpo.num_backtracks + 1 : cfg.max_backtracks;
  free(losses);
  oracle_adam_free(opt);
  free(cp);
  free(pp);
  return secs;
}

/* the same pass count over the first `n_prefix` samples with every pass split over `n_threads` threads */
double oracle_cpu_sample_update_intraop(oracle_cpu_sample *s, uint64_t n_prefix, uint64_t critic_steps,
                                        uint32_t n_threads) {
  oracle_features view = *s->feat;
  if (n_prefix && n_prefix < view.n_steps) view.n_steps = n_prefix;
  if (view.n_steps == 0) return 0.0;
  return update_intraop_seconds(s->ps, s->cs, s->policy_params, s->critic_params, &view, s->adv, s->rtg,
                                s->cg_iterations, s->n_evals, critic_steps, n_threads ? n_threads : 1);
}

void oracle_cpu_sample_free(oracle_cpu_sample *s) {
  if (!s) return;
  oracle_features_free(s->feat);
  free(s->adv);
  free(s->rtg);
  free(s->policy_params);
  free(s->critic_params);
  free(s);
}
