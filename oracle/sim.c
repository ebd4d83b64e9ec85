/* sim.c — oracle restatement of the simulation driver pieces: HistoryDataBound,
 * TakeAlignedSteps, VecBuffer/finalize_last_episode, ReplayBuffer eviction, tabular Q-learning and
 * the chain-tabular-q example.  TEST INFRASTRUCTURE (see oracle.h).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ HistoryDataBound
 * src/agents/buffers/mod.rs:25-113 */
static uint64_t div_ceil(uint64_t a, uint64_t b) { return a / b + (a % b ? 1 : 0); }

oracle_bound oracle_bound_divide(oracle_bound b, uint64_t n) {
  oracle_bound r = {div_ceil(b.min_steps, n), b.slack_steps};
  return r;
}

oracle_bound oracle_bound_max(oracle_bound a, oracle_bound b) {
  oracle_bound r = {a.min_steps > b.min_steps ? a.min_steps : b.min_steps,
                    a.slack_steps > b.slack_steps ? a.slack_steps : b.slack_steps};
  return r;
}

oracle_bound oracle_bound_with_default_slack(uint64_t min_steps) {
  uint64_t slack = min_steps / 100;
  if (slack < 5) slack = 5;
  if (slack > 1000) slack = 1000;
  oracle_bound r = {min_steps, slack};
  return r;
}

/* TakeAlignedSteps::next (src/simulation/take_steps.rs:18-33, 80-93) */
uint64_t oracle_take_aligned_count(const uint8_t *episode_done, uint64_t n_available, uint64_t min_steps,
                                   uint64_t slack_steps) {
  uint64_t n = min_steps == 0 ? 0 : min_steps + slack_steps;
  uint64_t taken = 0;
  while (n != 0 && taken < n_available) {
    int done = episode_done[taken];
    taken += 1;
    n -= 1;
    if (done && n <= slack_steps) n = 0;
  }
  return taken;
}

/* ------------------------------------------------------------------ VecBuffer
 * src/agents/buffers/vec.rs:15-143 */
oracle_vecbuffer *oracle_vecbuffer_new(uint32_t obs_dim) {
  oracle_vecbuffer *b = (oracle_vecbuffer *)calloc(1, sizeof(*b));
  b->obs_dim = obs_dim;
  return b;
}

void oracle_vecbuffer_free(oracle_vecbuffer *b) {
  if (!b) return;
  free(b->obs);
  free(b->next_obs);
  free(b->action);
  free(b->reward);
  free(b->next);
  free(b->episode_ends);
  free(b);
}

void oracle_vecbuffer_clear(oracle_vecbuffer *b) {
  b->len = 0;
  b->n_episode_ends = 0;
}

static void vb_reserve(oracle_vecbuffer *b, uint64_t need) {
  if (need <= b->cap) return;
  uint64_t cap = b->cap ? b->cap * 2 : 1024;
  while (cap < need) cap *= 2;
  b->obs = (float *)realloc(b->obs, cap * b->obs_dim * sizeof(float));
  b->next_obs = (float *)realloc(b->next_obs, cap * b->obs_dim * sizeof(float));
  b->action = (int32_t *)realloc(b->action, cap * sizeof(int32_t));
  b->reward = (double *)realloc(b->reward, cap * sizeof(double));
  b->next = (uint8_t *)realloc(b->next, cap);
  b->cap = cap;
}

static void vb_push_end(oracle_vecbuffer *b, uint64_t end) {
  if (b->n_episode_ends == b->cap_episode_ends) {
    b->cap_episode_ends = b->cap_episode_ends ? b->cap_episode_ends * 2 : 64;
    b->episode_ends = (uint64_t *)realloc(b->episode_ends, b->cap_episode_ends * sizeof(uint64_t));
  }
  b->episode_ends[b->n_episode_ends++] = end;
}

/* WriteExperienceIncremental::write_step (vec.rs:127-134) */
void oracle_vecbuffer_write_step(oracle_vecbuffer *b, const float *obs, int32_t action, double reward, int next,
                                 const float *next_obs) {
  vb_reserve(b, b->len + 1);
  uint64_t i = b->len;
  memcpy(b->obs + i * b->obs_dim, obs, b->obs_dim * sizeof(float));
  if (next == ORACLE_INTERRUPT && next_obs)
    memcpy(b->next_obs + i * b->obs_dim, next_obs, b->obs_dim * sizeof(float));
  else
    memset(b->next_obs + i * b->obs_dim, 0, b->obs_dim * sizeof(float));
  b->action[i] = action;
  b->reward[i] = reward;
  b->next[i] = (uint8_t)next;
  b->len += 1;
  if (next != ORACLE_CONTINUE) vb_push_end(b, b->len);
}

/* finalize_last_episode (src/agents/buffers/mod.rs:237-261) + end_experience (vec.rs:136-140) */
void oracle_vecbuffer_end_experience(oracle_vecbuffer *b) {
  if (b->len == 0 || b->next[b->len - 1] != ORACLE_CONTINUE) return;
  /* drop the dangling step; its observation becomes the interrupt successor of the one before */
  uint64_t last = b->len - 1;
  b->len -= 1;
  if (b->len > 0 && b->next[b->len - 1] == ORACLE_CONTINUE) {
    uint64_t prev = b->len - 1;
    b->next[prev] = ORACLE_INTERRUPT;
    memcpy(b->next_obs + prev * b->obs_dim, b->obs + last * b->obs_dim, b->obs_dim * sizeof(float));
    vb_push_end(b, b->len);
  }
}

uint64_t oracle_vecbuffer_num_steps(const oracle_vecbuffer *b) { return b->len; }
uint64_t oracle_vecbuffer_num_episodes(const oracle_vecbuffer *b) { return b->n_episode_ends; }

void oracle_vecbuffer_episode_ends(const oracle_vecbuffer *b, uint64_t *out) {
  memcpy(out, b->episode_ends, b->n_episode_ends * sizeof(uint64_t));
}

void oracle_vecbuffer_steps(const oracle_vecbuffer *b, float *obs, int32_t *action, double *reward, uint8_t *next,
                            float *next_obs) {
  if (obs) memcpy(obs, b->obs, b->len * b->obs_dim * sizeof(float));
  if (action) memcpy(action, b->action, b->len * sizeof(int32_t));
  if (reward) memcpy(reward, b->reward, b->len * sizeof(double));
  if (next) memcpy(next, b->next, b->len);
  if (next_obs) memcpy(next_obs, b->next_obs, b->len * b->obs_dim * sizeof(float));
}

/* ------------------------------------------------------------------ ReplayBuffer
 * src/agents/buffers/replay.rs:11-127 — a deque of steps + a deque of episode lengths; when a
 * write would exceed `capacity` the whole OLDEST episode is evicted first (replay.rs:89-115). */
struct oracle_replay {
  uint64_t capacity;
  int32_t *tags;     /* ring as a flat vector we compact on eviction (test sizes are tiny) */
  uint8_t *done;
  uint64_t len;
  uint64_t *ep_lens; /* completed episodes, oldest first */
  uint64_t n_eps, cap_eps;
  uint64_t total_step_count;
};

oracle_replay *oracle_replay_new(uint64_t capacity) {
  oracle_replay *r = (oracle_replay *)calloc(1, sizeof(*r));
  r->capacity = capacity;
  r->tags = (int32_t *)malloc((capacity + 1) * sizeof(int32_t));
  r->done = (uint8_t *)malloc(capacity + 1);
  return r;
}

void oracle_replay_free(oracle_replay *r) {
  if (!r) return;
  free(r->tags);
  free(r->done);
  free(r->ep_lens);
  free(r);
}

static void replay_push_ep(oracle_replay *r, uint64_t len) {
  if (r->n_eps == r->cap_eps) {
    r->cap_eps = r->cap_eps ? r->cap_eps * 2 : 16;
    r->ep_lens = (uint64_t *)realloc(r->ep_lens, r->cap_eps * sizeof(uint64_t));
  }
  r->ep_lens[r->n_eps++] = len;
}

static uint64_t replay_completed_steps(const oracle_replay *r) {
  uint64_t s = 0;
  for (uint64_t i = 0; i < r->n_eps; ++i) s += r->ep_lens[i];
  return s;
}

/* write_step (replay.rs:132-152): returns 0 ok, 1 = Full */
int oracle_replay_write_step(oracle_replay *r, int32_t tag, int episode_done) {
  if (r->len >= r->capacity) {
    /* drop the oldest complete episode; if there is none the buffer is Full */
    if (r->n_eps == 0) return 1;
    uint64_t drop = r->ep_lens[0];
    memmove(r->tags, r->tags + drop, (r->len - drop) * sizeof(int32_t));
    memmove(r->done, r->done + drop, r->len - drop);
    r->len -= drop;
    memmove(r->ep_lens, r->ep_lens + 1, (r->n_eps - 1) * sizeof(uint64_t));
    r->n_eps -= 1;
  }
  r->tags[r->len] = tag;
  r->done[r->len] = (uint8_t)episode_done;
  r->len += 1;
  r->total_step_count += 1;
  if (episode_done) replay_push_ep(r, r->len - replay_completed_steps(r));
  return 0;
}

/* end_experience (replay.rs:154-165) via finalize_last_episode */
void oracle_replay_end_experience(oracle_replay *r) {
  if (r->len == 0 || r->done[r->len - 1]) return;
  uint64_t completed = replay_completed_steps(r);
  r->len -= 1; /* pop dangling step */
  if (r->len > completed && !r->done[r->len - 1]) {
    r->done[r->len - 1] = 1; /* becomes Interrupt(final_observation) */
    r->total_step_count -= 1; /* only decremented when a new episode was created (replay.rs:155-157) */
    replay_push_ep(r, r->len - completed);
  }
}

uint64_t oracle_replay_num_steps(const oracle_replay *r) { return r->len; }
uint64_t oracle_replay_num_episodes(const oracle_replay *r) { return r->n_eps; }
uint64_t oracle_replay_total_step_count(const oracle_replay *r) { return r->total_step_count; }
void oracle_replay_dump(const oracle_replay *r, int32_t *tags, uint64_t *episode_lens) {
  memcpy(tags, r->tags, r->len * sizeof(int32_t));
  memcpy(episode_lens, r->ep_lens, r->n_eps * sizeof(uint64_t));
}

/* ------------------------------------------------------------------ tabular Q-learning
 * src/agents/tabular.rs:96-133, 157-233 */
oracle_tabular_q *oracle_tabular_q_new(uint64_t n_obs, uint64_t n_act, double gamma, double eps) {
  oracle_tabular_q *q = (oracle_tabular_q *)calloc(1, sizeof(*q));
  q->n_obs = n_obs;
  q->n_act = n_act;
  q->discount_factor = gamma;
  q->exploration_rate = eps;
  q->counts = (uint64_t *)calloc(n_obs * n_act, sizeof(uint64_t));
  q->values = (double *)calloc(n_obs * n_act, sizeof(double));
  return q;
}

void oracle_tabular_q_free(oracle_tabular_q *q) {
  if (!q) return;
  free(q->counts);
  free(q->values);
  free(q);
}

static uint64_t argmax_f64(const double *v, uint64_t n) {
  /* ndarray-stats argmax: first maximal element */
  uint64_t best = 0;
  for (uint64_t i = 1; i < n; ++i)
    if (v[i] > v[best]) best = i;
  return best;
}

/* BaseTabularQLearningActor::act (tabular.rs:222-232) */
int oracle_tabular_q_act(const oracle_tabular_q *q, uint64_t obs, int training, oracle_prng *rng) {
  if (training && oracle_prng_gen_f64(rng) < q->exploration_rate)
    return (int)oracle_prng_gen_range_u64(rng, 0, q->n_act);
  return (int)argmax_f64(q->values + obs * q->n_act, q->n_act);
}

/* BaseTabularQLearningAgent::step_update (tabular.rs:159-180) */
void oracle_tabular_q_step_update(oracle_tabular_q *q, uint64_t obs, uint64_t action, double reward, int next,
                                  uint64_t next_obs) {
  double discounted_next_value = 0.0;
  if (next != ORACLE_TERMINATE) {
    const double *row = q->values + next_obs * q->n_act;
    double m = row[0];
    for (uint64_t a = 1; a < q->n_act; ++a)
      if (row[a] > m) m = row[a];
    discounted_next_value = m * q->discount_factor;
  }
  uint64_t idx = obs * q->n_act + action;
  q->counts[idx] += 1;
  double value = reward + discounted_next_value;
  double weight = 1.0 / (double)q->counts[idx];
  q->values[idx] *= 1.0 - weight;
  q->values[idx] += weight * value;
}

void oracle_tabular_q_read(const oracle_tabular_q *q, double *values_out, uint64_t *counts_out) {
  if (values_out) memcpy(values_out, q->values, q->n_obs * q->n_act * sizeof(double));
  if (counts_out) memcpy(counts_out, q->counts, q->n_obs * q->n_act * sizeof(uint64_t));
}

/* examples/chain-tabular-q.rs:12-45 through train_parallel (simulation/train.rs:68-186).
 *   rng = seed_from_u64(seed); env build / agent build draw nothing;
 *   rng_env = from_rng(rng); rng_agent = rng;
 *   per thread i: (from_rng(rng_env), from_rng(rng_agent))               train.rs:99-106
 *   per period: bound = min_update_size{1,0}.divide(T).max({min_worker_steps,0}); each worker takes
 *   exactly that many steps of Steps::step (Chain never ends an episode), its last dangling step is
 *   dropped by finalize_last_episode and the one before becomes Interrupt(obs); then the buffers are
 *   drained in thread order through step_update (for_each_transient: next obs = following step's obs). */
void oracle_chain_tabular_q_train(uint64_t seed, uint64_t n_threads, uint64_t n_periods, uint64_t min_worker_steps,
                                  double *q_values_out, uint64_t *counts_out, uint64_t *total_steps_out) {
  oracle_chain env;
  oracle_chain_default(&env);
  oracle_tabular_q *q = oracle_tabular_q_new(env.size, 2, env.discount_factor, 0.2);
  oracle_prng rng, rng_env, *t_env, *t_agent;
  oracle_prng_seed_from_u64(&rng, seed);
  oracle_prng_from_rng(&rng_env, &rng);
  oracle_prng *rng_agent = &rng;
  t_env = (oracle_prng *)malloc(n_threads * sizeof(oracle_prng));
  t_agent = (oracle_prng *)malloc(n_threads * sizeof(oracle_prng));
  for (uint64_t i = 0; i < n_threads; ++i) {
    oracle_prng_from_rng(&t_env[i], &rng_env);
    oracle_prng_from_rng(&t_agent[i], rng_agent);
  }
  oracle_bound b = {1, 0};
  oracle_bound mw = {min_worker_steps, 0};
  b = oracle_bound_max(oracle_bound_divide(b, n_threads), mw);
  uint64_t n_take = b.min_steps + b.slack_steps;
  uint64_t *obs = (uint64_t *)malloc(n_threads * n_take * sizeof(uint64_t));
  uint8_t *act = (uint8_t *)malloc(n_threads * n_take);
  double *rew = (double *)malloc(n_threads * n_take * sizeof(double));
  uint64_t total = 0;
  for (uint64_t p = 0; p < n_periods; ++p) {
    /* collection: actors see the Q-table snapshot from the start of the period */
    for (uint64_t i = 0; i < n_threads; ++i) {
      uint64_t state = 0; /* Steps::new => fresh episode; Chain::initial_state = 0 */
      for (uint64_t t = 0; t < n_take; ++t) {
        int a = oracle_tabular_q_act(q, state, 1, &t_agent[i]);
        obs[i * n_take + t] = state;
        act[i * n_take + t] = (uint8_t)a;
        oracle_chain_step(&env, &state, a, &t_env[i], &rew[i * n_take + t]);
      }
    }
    /* update: the last step of each thread was dropped (dangling Continue); steps 0..n_take-2 remain,
     * each with next observation = obs of the following step (Continue or the final Interrupt) */
    for (uint64_t i = 0; i < n_threads; ++i) {
      for (uint64_t t = 0; t + 1 < n_take; ++t) {
        int next = (t + 2 == n_take) ? ORACLE_INTERRUPT : ORACLE_CONTINUE;
        oracle_tabular_q_step_update(q, obs[i * n_take + t], act[i * n_take + t], rew[i * n_take + t], next,
                                     obs[i * n_take + t + 1]);
        total += 1;
      }
    }
  }
  memcpy(q_values_out, q->values, env.size * 2 * sizeof(double));
  memcpy(counts_out, q->counts, env.size * 2 * sizeof(uint64_t));
  *total_steps_out = total;
  free(obs);
  free(act);
  free(rew);
  free(t_env);
  free(t_agent);
  oracle_tabular_q_free(q);
}

/* env.run(&actor(Evaluation), SimSeed::Root(seed), ()).take(n) (examples/chain-tabular-q.rs:47-50;
 * SimSeed::derive_rngs simulation/mod.rs:137-149) */
double oracle_chain_tabular_q_eval(const double *q_values, uint64_t seed, uint64_t n_steps, int32_t *actions_out) {
  oracle_chain env;
  oracle_chain_default(&env);
  oracle_tabular_q *q = oracle_tabular_q_new(env.size, 2, env.discount_factor, 0.2);
  memcpy(q->values, q_values, env.size * 2 * sizeof(double));
  oracle_prng env_rng, agent_rng;
  oracle_prng_seed_from_u64(&env_rng, seed);
  oracle_prng_seed_from_u64(&agent_rng, oracle_prng_next_u64(&env_rng));
  uint64_t state = 0;
  double total = 0.0;
  for (uint64_t t = 0; t < n_steps; ++t) {
    int a = oracle_tabular_q_act(q, state, 0, &agent_rng);
    if (actions_out) actions_out[t] = a;
    double r;
    oracle_chain_step(&env, &state, a, &env_rng, &r);
    total += r;
  }
  oracle_tabular_q_free(q);
  return total;
}
