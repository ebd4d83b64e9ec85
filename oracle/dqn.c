/* dqn.c — oracle restatement of the DQN path: epsilon-greedy collection into per-lane replay buffers, minibatch
 * episode sampling with the agent Prng, value targets and the n_backward_steps loop.
 * TEST INFRASTRUCTURE (see oracle.h).
 *
 * Reference: DqnAgent / DqnActor (src/torch/agents/dqn.rs:200-211, 263-337, 360-379), ReplayBuffer
 * (src/agents/buffers/replay.rs:89-127, 154-165), StepValueTarget (src/torch/agents/critics/mod.rs:101-148,
 * 203-229), n_backward_steps (src/torch/agents/mod.rs:35-72).
 *
 * Every lane owns one ReplayBuffer.  The eviction bookkeeping is `oracle_replay` (sim.c, pinned by the
 * reference's replay tests); its tags are the lane's absolute step numbers, which index an append-only store of
 * the step data.  The lane model's horizon rule writes the last step of a collection as
 * Interrupt(successor observation) when the episode is still running, so no step is ever left dangling.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

struct oracle_dqn_store {
  uint32_t obs_dim;
  uint64_t n_lanes, capacity;
  oracle_replay **rings;
  uint64_t *n_written, *cap_written;
  float **obs, **next_obs, **reward;
  uint8_t **action, **next;
  uint64_t *actor_pos;
};

oracle_dqn_store *oracle_dqn_store_new(uint64_t n_lanes, uint64_t capacity, uint32_t obs_dim) {
  oracle_dqn_store *s = (oracle_dqn_store *)calloc(1, sizeof(*s));
  s->obs_dim = obs_dim;
  s->n_lanes = n_lanes;
  s->capacity = capacity;
  s->rings = (oracle_replay **)calloc(n_lanes, sizeof(*s->rings));
  s->n_written = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  s->cap_written = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  s->obs = (float **)calloc(n_lanes, sizeof(float *));
  s->next_obs = (float **)calloc(n_lanes, sizeof(float *));
  s->reward = (float **)calloc(n_lanes, sizeof(float *));
  s->action = (uint8_t **)calloc(n_lanes, sizeof(uint8_t *));
  s->next = (uint8_t **)calloc(n_lanes, sizeof(uint8_t *));
  s->actor_pos = (uint64_t *)calloc(n_lanes, sizeof(uint64_t));
  for (uint64_t i = 0; i < n_lanes; ++i) s->rings[i] = oracle_replay_new(capacity);
  return s;
}

void oracle_dqn_store_free(oracle_dqn_store *s) {
  if (!s) return;
  for (uint64_t i = 0; i < s->n_lanes; ++i) {
    oracle_replay_free(s->rings[i]);
    free(s->obs[i]);
    free(s->next_obs[i]);
    free(s->reward[i]);
    free(s->action[i]);
    free(s->next[i]);
  }
  free(s->rings);
  free(s->n_written);
  free(s->cap_written);
  free(s->obs);
  free(s->next_obs);
  free(s->reward);
  free(s->action);
  free(s->next);
  free(s->actor_pos);
  free(s);
}

static int store_write(oracle_dqn_store *s, uint64_t lane, const float *obs, int action, float reward, int next,
                       const float *next_obs) {
  uint64_t k = s->n_written[lane];
  uint32_t D = s->obs_dim;
  if (k == s->cap_written[lane]) {
    uint64_t cap = s->cap_written[lane] ? 2 * s->cap_written[lane] : 256;
    s->obs[lane] = (float *)realloc(s->obs[lane], cap * D * sizeof(float));
    s->next_obs[lane] = (float *)realloc(s->next_obs[lane], cap * D * sizeof(float));
    s->reward[lane] = (float *)realloc(s->reward[lane], cap * sizeof(float));
    s->action[lane] = (uint8_t *)realloc(s->action[lane], cap);
    s->next[lane] = (uint8_t *)realloc(s->next[lane], cap);
    s->cap_written[lane] = cap;
  }
  /* ReplayBuffer::write_step (replay.rs:89-115) */
  if (oracle_replay_write_step(s->rings[lane], (int32_t)k, next != ORACLE_CONTINUE)) return 1;
  memcpy(s->obs[lane] + k * D, obs, D * sizeof(float));
  if (next_obs) memcpy(s->next_obs[lane] + k * D, next_obs, D * sizeof(float));
  else memset(s->next_obs[lane] + k * D, 0, D * sizeof(float));
  s->reward[lane][k] = reward;
  s->action[lane][k] = (uint8_t)action;
  s->next[lane][k] = (uint8_t)next;
  s->n_written[lane] = k + 1;
  return 0;
}

uint64_t oracle_dqn_store_actor_pos(const oracle_dqn_store *s, uint64_t lane) { return s->actor_pos[lane]; }

void oracle_dqn_store_lane_info(const oracle_dqn_store *s, uint64_t lane, uint64_t *num_steps, uint64_t *num_episodes,
                                uint64_t *total_step_count) {
  *num_steps = oracle_replay_num_steps(s->rings[lane]);
  *num_episodes = oracle_replay_num_episodes(s->rings[lane]);
  *total_step_count = oracle_replay_total_step_count(s->rings[lane]);
}

/* stored steps of a lane, oldest first (tags = absolute step numbers) and the lengths of its complete episodes */
void oracle_dqn_store_lane_dump(const oracle_dqn_store *s, uint64_t lane, int32_t *tags, uint64_t *episode_lens) {
  oracle_replay_dump(s->rings[lane], tags, episode_lens);
}

void oracle_dqn_store_step(const oracle_dqn_store *s, uint64_t lane, uint64_t abs_index, float *obs, uint8_t *action,
                           float *reward, uint8_t *next, float *next_obs) {
  uint32_t D = s->obs_dim;
  memcpy(obs, s->obs[lane] + abs_index * D, D * sizeof(float));
  memcpy(next_obs, s->next_obs[lane] + abs_index * D, D * sizeof(float));
  *action = s->action[lane][abs_index];
  *reward = s->reward[lane][abs_index];
  *next = s->next[lane][abs_index];
}

/* DqnActor::act (dqn.rs:360-379) */
static int dqn_act(oracle_mlp_shape qs, const float *qparams, const float *feat, double eps, oracle_prng *rng) {
  if (oracle_prng_gen_bool(rng, eps)) return (int)oracle_prng_gen_range_u64(rng, 0, qs.out_dim); /* IndexSpace::sample */
  float z[16];
  oracle_mlp_forward_f32(qs, qparams, feat, z);
  int best = 0; /* argmax: first maximal index */
  for (uint32_t a = 1; a < qs.out_dim; ++a)
    if (z[a] > z[best]) best = (int)a;
  return best;
}

/* T env-actor steps on every lane with the epsilon-greedy actor, written to the lanes' buffers.
 * flags_out: [T][n] successor codes as recorded (may be NULL).  Returns 1 when a buffer reported Full. */
int oracle_lanes_rollout_dqn(oracle_lanes *l, oracle_dqn_store *st, oracle_mlp_shape qs, const float *qparams,
                             uint64_t T, double eps, uint8_t *flags_out) {
  uint64_t n = l->n_lanes;
  int full = 0;
  for (uint64_t i = 0; i < n; ++i) {
    oracle_prng env_rng, act_rng;
    oracle_prng_seed_from_u64(&env_rng, l->seed_env);
    oracle_prng_set_stream(&env_rng, l->lane_offset + i);
    oracle_prng_seed_from_u64(&act_rng, l->seed_actor);
    oracle_prng_set_stream(&act_rng, l->lane_offset + i);
    oracle_prng_set_word_pos(&act_rng, st->actor_pos[i]);
    float f[8], nf[8];
    for (uint64_t t = 0; t < T; ++t) {
      oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, f);
      int a = dqn_act(qs, qparams, f, eps, &act_rng);
      double r;
      int succ = oracle_cartpole_step(&l->env, &l->state[i], a, &r);
      if (l->limit_kind != ORACLE_LIMIT_NONE) succ = oracle_step_limit_apply(succ, &l->steps_remaining[i]);
      int rec = (succ == ORACLE_CONTINUE && t + 1 == T) ? ORACLE_INTERRUPT : succ; /* horizon rule */
      const float *np = NULL;
      if (rec == ORACLE_INTERRUPT) {
        oracle_cartpole_features(&l->state[i], l->limit_kind, l->steps_remaining[i], l->max_steps, nf);
        np = nf;
      }
      if (store_write(st, i, f, a, (float)r, rec, np)) {
        full = 1;
        break;
      }
      if (flags_out) flags_out[t * n + i] = (uint8_t)rec;
      if (succ != ORACLE_CONTINUE) {
        oracle_prng_set_word_pos(&env_rng, 8 * l->reset_count[i]);
        oracle_cartpole_initial_state(&l->env, &env_rng, &l->state[i]);
        l->steps_remaining[i] = l->max_steps;
        l->reset_count[i] += 1;
      }
    }
    st->actor_pos[i] = oracle_prng_word_pos(&act_rng);
  }
  l->t_global += T;
  return full;
}

/* sample_minibatch's episode selection (dqn.rs:280-291).  Outputs per selected episode: lane, absolute index of
 * its first step, length.  Returns the number of episodes (or -1 when a lane holds no complete episode: the
 * reference panics in Uniform::new); *n_steps_out = total steps. */
int64_t oracle_dqn_sample(const oracle_dqn_store *st, oracle_prng *agent_rng, uint64_t minibatch_steps,
                          uint32_t *lane_out, uint32_t *start_out, uint32_t *len_out, uint64_t cap,
                          uint64_t *n_steps_out) {
  uint64_t total = 0, n_eps = 0;
  int32_t *tags = (int32_t *)malloc((st->capacity + 1) * sizeof(int32_t));
  uint64_t *lens = (uint64_t *)malloc((st->capacity + 1) * sizeof(uint64_t));
  int64_t rc = 0;
  for (uint64_t cand = 0;; ++cand) {
    uint64_t lane = cand % st->n_lanes; /* iter::repeat(&*buffers).flatten() */
    uint64_t n = oracle_replay_num_episodes(st->rings[lane]);
    if (n == 0) {
      rc = -1;
      break;
    }
    /* Uniform::new(0usize, n).sample(rng): UniformInt::sample with the precomputed zone (rand 0.8.5) */
    uint64_t ints_to_reject = (0ull - n) % n;
    uint64_t zone = ~0ull - ints_to_reject;
    uint64_t idx;
    for (;;) {
      uint64_t v = oracle_prng_next_u64(agent_rng);
      unsigned __int128 m = (unsigned __int128)v * (unsigned __int128)n;
      if ((uint64_t)m <= zone) {
        idx = (uint64_t)(m >> 64);
        break;
      }
    }
    /* Episodes::get(idx) (replay.rs:154-165) */
    oracle_replay_dump(st->rings[lane], tags, lens);
    uint64_t start_rel = 0;
    for (uint64_t e = 0; e < idx; ++e) start_rel += lens[e];
    /* take_while: the predicate sees the episode, then decides on the total BEFORE it */
    int take = total < minibatch_steps;
    total += lens[idx];
    if (!take) {
      total -= lens[idx];
      break;
    }
    if (n_eps < cap) {
      lane_out[n_eps] = (uint32_t)lane;
      start_out[n_eps] = (uint32_t)tags[start_rel];
      len_out[n_eps] = (uint32_t)lens[idx];
    }
    n_eps += 1;
  }
  free(tags);
  free(lens);
  *n_steps_out = total;
  return rc < 0 ? rc : (int64_t)n_eps;
}

/* observation / action / target arrays of the selected episodes, episode after episode ([n][D] rows).
 * RewardToGo: discounted_cumsum_from_end over each episode (packed.rs:312-342 arithmetic: a += b * discount);
 * OneStepTd: r + gamma * max_a Q(next), 0 after Terminate (critics/mod.rs:116-148, dqn.rs:300-309). */
void oracle_dqn_minibatch(const oracle_dqn_store *st, uint64_t n_eps, const uint32_t *lanes, const uint32_t *starts,
                          const uint32_t *lens, oracle_mlp_shape qs, const float *qparams, float gamma,
                          int one_step_td, float *obs_out, int64_t *actions_out, float *targets_out) {
  uint32_t D = st->obs_dim;
  uint64_t off = 0;
  for (uint64_t e = 0; e < n_eps; ++e) {
    uint64_t lane = lanes[e], s0 = starts[e], len = lens[e];
    for (uint64_t i = 0; i < len; ++i) {
      memcpy(obs_out + (off + i) * D, st->obs[lane] + (s0 + i) * D, D * sizeof(float));
      actions_out[off + i] = st->action[lane][s0 + i];
    }
    if (one_step_td) {
      for (uint64_t i = 0; i < len; ++i) {
        uint8_t nx = st->next[lane][s0 + i];
        float vnext = 0.0f; /* masked_fill_(is_invalid, 0) */
        if (nx != ORACLE_TERMINATE) {
          const float *x = nx == ORACLE_INTERRUPT ? st->next_obs[lane] + (s0 + i) * D : st->obs[lane] + (s0 + i + 1) * D;
          float z[16];
          oracle_mlp_forward_f32(qs, qparams, x, z);
          vnext = z[0];
          for (uint32_t a = 1; a < qs.out_dim; ++a)
            if (z[a] > vnext) vnext = z[a]; /* amax(-1) */
        }
        float dn = gamma * vnext;
        targets_out[off + i] = st->reward[lane][s0 + i] + dn;
      }
    } else {
      float g = 0.0f;
      for (uint64_t i = len; i-- > 0;) {
        if (i == len - 1) g = st->reward[lane][s0 + i];
        else {
          float p = g * gamma;
          g = st->reward[lane][s0 + i] + p;
        }
        targets_out[off + i] = g;
      }
    }
    off += len;
  }
}

/* DqnAgent::batch_update_slice_refs: opt_steps x {sample_minibatch, loss, backward, Adam step}.
 * losses_out[k] = loss before step k.  Returns 0, or -1 if sampling failed. */
int oracle_dqn_update_f32(const oracle_dqn_store *st, oracle_prng *agent_rng, oracle_mlp_shape qs, float *qparams,
                          oracle_adam_state *opt, const oracle_adam_cfg *acfg, uint64_t minibatch_steps,
                          uint64_t opt_steps, float gamma, int one_step_td, float *losses_out) {
  uint64_t cap_eps = minibatch_steps, cap_steps = minibatch_steps + st->capacity;
  uint32_t D = st->obs_dim;
  uint32_t *lanes = (uint32_t *)malloc(cap_eps * sizeof(uint32_t));
  uint32_t *starts = (uint32_t *)malloc(cap_eps * sizeof(uint32_t));
  uint32_t *lens = (uint32_t *)malloc(cap_eps * sizeof(uint32_t));
  float *obs = (float *)malloc(cap_steps * D * sizeof(float));
  int64_t *actions = (int64_t *)malloc(cap_steps * sizeof(int64_t));
  float *targets = (float *)malloc(cap_steps * sizeof(float));
  uint64_t P = oracle_mlp_num_params(qs);
  float *g = (float *)malloc(P * sizeof(float));
  int rc = 0;
  for (uint64_t k = 0; k < opt_steps; ++k) {
    uint64_t n_steps;
    int64_t n_eps = oracle_dqn_sample(st, agent_rng, minibatch_steps, lanes, starts, lens, cap_eps, &n_steps);
    if (n_eps < 0) {
      rc = -1;
      break;
    }
    oracle_dqn_minibatch(st, (uint64_t)n_eps, lanes, starts, lens, qs, qparams, gamma, one_step_td, obs, actions,
                         targets);
    float loss;
    oracle_dqn_grad_f32(qs, qparams, obs, actions, targets, n_steps, g, &loss);
    if (losses_out) losses_out[k] = loss;
    oracle_adam_step_f32(opt, acfg, qparams, g);
  }
  free(lanes);
  free(starts);
  free(lens);
  free(obs);
  free(actions);
  free(targets);
  free(g);
  return rc;
}

/* ExplorationRateSchedule::exploration_rate (schedules.rs:35-45); kind 0 = Constant(start), 1 = LinearAnnealed */
double oracle_exploration_rate(int kind, double start, double end, uint64_t period, uint64_t global_steps,
                               int training) {
  if (!training) return 0.0;
  if (kind == 0) return start;
  double frac = (double)global_steps / (double)period;
  if (!(frac < 1.0)) frac = 1.0;
  return frac * (end - start) + start;
}

/* DataCollectionSchedule::update_size (schedules.rs:58-68); kind 0 = Constant(first), 1 = FirstRest */
oracle_bound oracle_collection_update_size(int kind, uint64_t first, uint64_t rest, uint64_t global_steps) {
  uint64_t min_steps = kind == 0 ? first : (global_steps < first ? first : rest);
  return oracle_bound_with_default_slack(min_steps);
}
