// abi_dqn.hip — extern "C" entry points of include/relearn_hip.h, part: DQN with the replay store in HBM (host side only; kernels live in kernels_*.hip).
#include "abi_internal.hpp"

extern "C" {

// ---------------------------------------------------------------- DQN (src/torch/agents/dqn.rs)
int32_t rl_dqn_config_default(rl_dqn_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    std::memset(c, 0, sizeof(*c));
    c->target = RL_DQN_TARGET_REWARD_TO_GO;               // StepValueTarget::default (critics/mod.rs:211-215)
    c->exploration_kind = RL_SCHEDULE_LINEAR_ANNEALED;    // schedules.rs:23-31
    c->exploration_start = 1.0;
    c->exploration_end = 0.1;
    c->exploration_period = 10000000;
    c->minibatch_steps = 100000;                          // dqn.rs:63-70
    c->opt_steps_per_update = 50;
    c->buffer_capacity = 0;
    c->episode_capacity = 0;
    c->update_kind = RL_COLLECT_FIRST_REST;
    c->update_first = 1000000;
    c->update_rest = 100000;
    c->discount_factor = 0.99f;
  });
}

static double dqn_exploration_rate(const rl_dqn *q, bool training) {
  if (!training) return 0.0;  // schedules.rs:38
  if (q->cfg.exploration_kind == RL_SCHEDULE_CONSTANT) return q->cfg.exploration_start;
  double frac = (double)q->global_steps / (double)q->cfg.exploration_period;
  if (!(frac < 1.0)) frac = 1.0;  // f64::min(1.0)
  return frac * (q->cfg.exploration_end - q->cfg.exploration_start) + q->cfg.exploration_start;
}

static ReplayDev replay_alloc(rl_engine *e, uint32_t N, uint32_t C, uint32_t E, uint32_t D) {
  ReplayDev r{};
  r.N = N;
  r.C = C;
  r.E = E;
  r.D = D;
  size_t cn = (size_t)C * N;
  r.rec = dalloc<ReplayRec>(cn);
  r.next = dalloc<ReplayNext>(cn);
  r.head = dalloc<uint32_t>(N);
  r.count = dalloc<uint32_t>(N);
  r.ep_head = dalloc<uint32_t>(N);
  r.ep_count = dalloc<uint32_t>(N);
  r.total = dalloc<uint32_t>(N);
  r.ep_end = dalloc<uint32_t>((size_t)E * N);
  r.actor_pos = dalloc<uint64_t>(N);
  r.error = dalloc<int32_t>(1);
  uint32_t *zero_u32[] = {r.head, r.count, r.ep_head, r.ep_count, r.total};
  for (uint32_t *p : zero_u32) RL_HIP_CHECK(hipMemsetAsync(p, 0, (size_t)N * 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(r.actor_pos, 0, (size_t)N * 8, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(r.error, 0, 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(r.next, 0, cn * sizeof(ReplayNext), e->stream));
  return r;
}

static void replay_free(ReplayDev &r) {
  void *ptrs[] = {r.rec, r.next, r.head, r.count, r.ep_head, r.ep_count, r.total, r.ep_end, r.actor_pos, r.error};
  for (void *p : ptrs) dfree(p);
  r = ReplayDev{};
}

// point the minibatch workspace back at its own sample arrays (an all-at-once update leaves it looking at its last
// minibatch inside the big arrays)
static void dqn_own_arrays(rl_dqn *q) {
  if (!q->mb || !q->own_obs) return;
  q->mb->d.obs = q->own_obs;
  q->mb->d.adv = q->own_target;
  q->mb->d.action = q->own_action;
  q->mb->d.flag = q->own_flag;
  q->td_in_kernel = false;
}

// every device allocation of a DQN handle (also the clean-up of a failed rl_dqn_create)
static void dqn_release_device(rl_dqn *q) {
  replay_free(q->rp);
  void *ptrs[] = {q->d_agent_pos, q->d_ep_lane, q->d_ep_start, q->d_ep_len, q->d_ep_off, q->d_counts, q->d_flags,
                  q->all_obs,     q->all_target, q->all_action, q->all_flag};
  for (void *p : ptrs) dfree(p);
  q->all_obs = q->all_target = nullptr;
  q->all_action = q->all_flag = nullptr;
  if (q->draw_stream) {
    (void)hipStreamSynchronize(q->draw_stream);
    (void)hipStreamDestroy(q->draw_stream);
  }
  q->draw_stream = nullptr;
  for (hipEvent_t ev : q->draw_events) (void)hipEventDestroy(ev);
  q->draw_events.clear();
  if (q->main_event) (void)hipEventDestroy(q->main_event);
  q->main_event = nullptr;
  if (q->h_counts) (void)hipHostFree(q->h_counts);
  dfree(q->snap);
  dfree(q->d_q);
  dfree(q->d_q_next);
  q->h_counts = nullptr;
  q->d_agent_pos = nullptr;
  q->d_ep_lane = q->d_ep_start = q->d_ep_len = q->d_ep_off = nullptr;
  q->d_counts = nullptr;
  q->d_flags = nullptr;
  dqn_own_arrays(q);
  if (q->mb) rl_traj_destroy(q->mb);
  q->mb = nullptr;
}

// With several ranks a failure must be raised on ALL of them: a rank that throws before a collective leaves its peers
// blocked in it.  One float travels through the update workspace; true when any rank reports a failure.
static bool dqn_any_rank_failed(rl_dqn *q, bool local_failure) {
  rl_engine *e = q->eng;
  if (e->n_ranks <= 1) return local_failure;
  float flag = local_failure ? 1.0f : 0.0f;
  h2d(e, q->mb->vec, &flag, sizeof(flag));
  rl_allreduce_sum_f32(e, q->mb->vec, 1);
  d2h(e, &flag, q->mb->vec, sizeof(flag));
  return flag > 0.0f;
}

int32_t rl_dqn_create(rl_env *env, rl_mlp *qnet, rl_adam *opt, const rl_dqn_config *cfg, rl_dqn **out) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && qnet && opt && cfg && out, "NULL argument");
    *out = nullptr;
    rl_engine *e = env->eng;
    RL_REQUIRE(qnet->eng == e && opt->eng == e, "handles belong to different engines");
    RL_REQUIRE(opt->mod == qnet, "optimizer does not belong to the action-value module");
    RL_REQUIRE(qnet->in_dim == env->D && qnet->out_dim == env->A, "action-value module does not match the env");
    RL_REQUIRE(env->A == 2, "DQN kernels are built for 2-action envs");
    // DqnConfig<MB> is generic over the module (dqn.rs:26-39): feed-forward modules of any MlpConfig build (the fused
    // 5-128-2 shape on the fused kernels, others on the per-layer kernels); recurrent action-value modules do not
    if (rl_module_is_recurrent(qnet->kind))
      throw RlError(RL_ERR_BUILD_AGENT, "DQN is built for feed-forward action-value modules");
    RL_REQUIRE(cfg->target == RL_DQN_TARGET_REWARD_TO_GO || cfg->target == RL_DQN_TARGET_ONE_STEP_TD, "bad target");
    RL_REQUIRE(cfg->minibatch_steps > 0 && cfg->minibatch_steps < (1ull << 30), "bad minibatch_steps");
    RL_REQUIRE(cfg->buffer_capacity > 0 && cfg->buffer_capacity < (1ull << 31), "bad buffer_capacity");
    uint64_t E = cfg->episode_capacity ? cfg->episode_capacity : cfg->buffer_capacity;
    RL_REQUIRE(E <= cfg->buffer_capacity, "episode_capacity exceeds buffer_capacity");
    RL_REQUIRE(cfg->opt_steps_per_update <= 4096, "too many optimisation steps per update");
    if (cfg->exploration_kind == RL_SCHEDULE_LINEAR_ANNEALED)
      RL_REQUIRE(cfg->exploration_period > 0, "exploration_period must be positive");
    uint64_t N = env->cfg.n_lanes;
    RL_REQUIRE(cfg->buffer_capacity * N * (sizeof(ReplayRec) + sizeof(ReplayNext)) < (200ull << 30), "replay store would not fit in HBM");
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_dqn> q(new rl_dqn());
    q->eng = e;
    q->env = env;
    q->qnet = qnet;
    q->opt = opt;
    q->cfg = *cfg;
    q->cfg.episode_capacity = E;
    try {
    q->rp = replay_alloc(e, (uint32_t)N, (uint32_t)cfg->buffer_capacity, (uint32_t)E, env->D);
    q->d_agent_pos = dalloc<uint64_t>(1);
    RL_HIP_CHECK(hipMemsetAsync(q->d_agent_pos, 0, 8, e->stream));
    // take_while accepts episodes while total < minibatch_steps and every episode has >= 1 step
    q->max_eps = (uint32_t)cfg->minibatch_steps;
    q->max_steps_mb = cfg->minibatch_steps - 1 + cfg->buffer_capacity;
    // the episode lists of all opt_steps_per_update minibatches of an update are drawn in one launch
    const size_t nb = cfg->opt_steps_per_update ? cfg->opt_steps_per_update : 1;
    q->d_ep_lane = dalloc<uint32_t>(nb * q->max_eps);
    q->d_ep_start = dalloc<uint32_t>(nb * q->max_eps);
    q->d_ep_len = dalloc<uint32_t>(nb * q->max_eps);
    q->d_ep_off = dalloc<uint32_t>(nb * q->max_eps);
    q->d_counts = dalloc<DqnCountsDev>(nb);
    RL_HIP_CHECK(hipMemsetAsync(q->d_counts, 0, nb * sizeof(DqnCountsDev), e->stream));
    q->mb = traj_alloc(e, q->max_steps_mb, 1, env->D, true);
    {  // the fused step's range guard (bf16_tile.hpp): the words hold the magnitude range of every observation the
       // collection kernel has put into the store so far (k_rollout_cartpole_dqn folds what it writes, one fold per wave
       // and launch; never reset, so they bound every minibatch drawn from the store from outside).  Rounds 3-5 kept fixed bounds here, 2^-64 .. 2^16,
       // under which a Q-network with a zero bias could never pass the guard.
      std::vector<uint32_t> words(RL_RANGE_WORDS, 0u);  // every slot of the minima / the maxima (bf16_tile.hpp)
      for (int s = 0; s < 64; ++s) words[(size_t)s * 32] = 0x7F7FFFFFu;  // "nothing seen": largest finite minimum, maximum 0
      h2d(e, q->mb->d.range, words.data(), words.size() * sizeof(uint32_t));
      q->mb->range_fixed = true;  // (maintained by the builders: no measuring pass over the workspace)
    }
    q->own_obs = q->mb->d.obs;
    q->own_target = q->mb->d.adv;
    q->own_action = q->mb->d.action;
    q->own_flag = q->mb->d.flag;
    sync(e);
    } catch (...) {  // (unique_ptr frees the host struct only)
      dqn_release_device(q.get());
      throw;
    }
    e->live_handles += 1;
    *out = q.release();
  });
}

int32_t rl_dqn_destroy(rl_dqn *q) {
  if (!q) return RL_OK;
  (void)hipSetDevice(q->eng->device);
  (void)hipStreamSynchronize(q->eng->stream);
  (void)hipStreamSynchronize(q->eng->aux_stream);
  dqn_release_device(q);
  rl_engine *eng = q->eng;
  delete q;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_dqn_exploration_rate(const rl_dqn *q, int32_t training, double *rate_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && rate_out, "NULL argument");
    *rate_out = dqn_exploration_rate(q, training != 0);
  });
}

int32_t rl_dqn_min_update_size(const rl_dqn *q, uint64_t *min_steps_out, uint64_t *slack_steps_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && min_steps_out && slack_steps_out, "NULL argument");
    // DataCollectionSchedule::update_size (schedules.rs:58-68)
    uint64_t min_steps;
    if (q->cfg.update_kind == RL_COLLECT_CONSTANT) min_steps = q->cfg.update_first;
    else min_steps = q->global_steps < q->cfg.update_first ? q->cfg.update_first : q->cfg.update_rest;
    *min_steps_out = min_steps;
    // HistoryDataBound::with_default_slack (src/agents/buffers/mod.rs:54-63): 1 % of min_steps, between 5 and 1000
    uint64_t slack = min_steps / 100;
    slack = slack < 5 ? 5 : (slack > 1000 ? 1000 : slack);
    *slack_steps_out = slack;
  });
}

int32_t rl_dqn_collect(rl_dqn *q, uint64_t horizon, rl_dqn_collect_stats *stats) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q, "NULL argument");
    RL_REQUIRE(horizon > 0 && horizon < (1ull << 31), "bad horizon");
    rl_engine *e = q->eng;
    uint64_t N = q->rp.N;
    if (q->flags_cap < horizon * N) {
      dfree(q->d_flags);
      q->d_flags = nullptr;
      q->flags_cap = 0;
      q->d_flags = dalloc<uint8_t>(horizon * N);
      q->flags_cap = horizon * N;
    }
    // DqnAgent::actor(Training) (dqn.rs:200-211) + Bernoulli::new(p) of rand 0.8.5: p_int = (p * 2^64) as u64,
    // p == 1.0 always true without a draw
    double eps = dqn_exploration_rate(q, true);
    RL_REQUIRE(eps >= 0.0 && eps <= 1.0, "exploration rate outside [0, 1]");
    int always = eps == 1.0 ? 1 : 0;
    uint64_t p_int = always ? ~0ull : (uint64_t)(eps * 18446744073709551616.0);
    if (q->qnet->general) {
      if (!q->d_q) q->d_q = dalloc<float>(2 * (size_t)q->rp.N);
      launch_rollout_dqn_general(q->env, q->qnet, q->mb, q->d_q, q->rp, (uint32_t)horizon, p_int, always, q->d_flags);
    } else {
      launch_rollout_dqn(q->env, q->qnet, q->rp, (uint32_t)horizon, p_int, always, q->d_flags, q->mb->d.range);
    }
    q->env->t_global += horizon;
    q->steps_per_lane += horizon;
    q->last_horizon = horizon;
    int32_t err = 0;
    d2h(e, &err, q->rp.error, sizeof(err));
    if (dqn_any_rank_failed(q, err != 0))
      throw RlError(RL_ERR_BUFFER_FULL, err != 0 ? "replay buffer full: an episode outgrew the lane capacity"
                                                 : "replay buffer full on another rank");
    if (stats) {
      std::vector<uint8_t> fl(horizon * N);
      d2h(e, fl.data(), q->d_flags, fl.size());
      uint64_t ended = 0;
      for (uint8_t f : fl) ended += f != RL_SUCC_CONTINUE;
      stats->exploration_rate = eps;
      stats->steps = horizon * N;
      stats->episodes_ended = ended;
    }
  });
}

static AgentKey dqn_key(const rl_dqn *q) {
  AgentKey k;
  std::memcpy(k.w, q->cfg.agent_key, sizeof(k.w));
  return k;
}

// the sampler's verdict on `n` minibatches: RL_OK, or the first failure and what it was
static int32_t dqn_check_counts(const rl_dqn *q, const DqnCountsDev *counts, size_t n, const char **what) {
  for (size_t i = 0; i < n; ++i) {
    const DqnCountsDev &c = counts[i];
    if (c.error == 2) {
      *what = "minibatch sampling from a lane without a complete episode";
      return RL_ERR_INVALID_ARGUMENT;
    } else if (c.error != 0) {
      *what = "replay buffer full";
      return RL_ERR_BUFFER_FULL;
    } else if (c.n_eps > q->max_eps || c.n_steps > q->max_steps_mb) {
      *what = "minibatch exceeds its workspace";
      return RL_ERR_INVALID_ARGUMENT;
    } else if (c.n_steps == 0) {
      *what = "empty minibatch";
      return RL_ERR_INVALID_ARGUMENT;
    }
  }
  return RL_OK;
}

// one sample_minibatch (dqn.rs:279-314): draw episodes, gather them, compute targets
// draw the episode lists of `n_batches` consecutive minibatches (dqn.rs:280-291) in one launch and read back their
// sizes: the draws do not depend on the network, so the whole update needs this one host round trip
static void dqn_draw_minibatches(rl_dqn *q, int sequential, uint32_t n_batches, std::vector<DqnCountsDev> &counts,
                                 std::vector<uint64_t> &totals) {
  rl_engine *e = q->eng;
  launch_dqn_sample(e, e->stream, q->rp, dqn_key(q), q->d_agent_pos, (uint32_t)q->cfg.minibatch_steps, q->max_eps,
                    q->d_ep_lane, q->d_ep_start, q->d_ep_len, q->d_ep_off, q->d_counts, sequential, n_batches);
  counts.resize(n_batches);
  d2h(e, counts.data(), q->d_counts, n_batches * sizeof(DqnCountsDev));
  // local validation first, the verdict only after every rank has reported (a rank that throws here alone would leave
  // its peers blocked in the count all-reduce below)
  const char *what = "";
  const int32_t code = dqn_check_counts(q, counts.data(), counts.size(), &what);
  if (dqn_any_rank_failed(q, code != RL_OK))
    throw RlError(code != RL_OK ? code : RL_ERR_COMM, code != RL_OK ? what : "minibatch sampling failed on another rank");
  // the loss is a mean over all ranks' samples: sum the per-rank counts (two 16-bit halves each, exact in f32)
  totals.resize(n_batches);
  for (uint32_t k = 0; k < n_batches; ++k) totals[k] = counts[k].n_steps;
  if (e->n_ranks > 1) {
    std::vector<float> halves(2 * n_batches);
    for (uint32_t k = 0; k < n_batches; ++k) {
      halves[2 * k] = (float)(counts[k].n_steps & 0xffffu);
      halves[2 * k + 1] = (float)(counts[k].n_steps >> 16);
    }
    RL_REQUIRE(2 * n_batches <= q->mb->Pmax, "too many minibatches for the exchange buffer");
    h2d(e, q->mb->vec, halves.data(), halves.size() * sizeof(float));
    rl_allreduce_sum_f32(e, q->mb->vec, halves.size());
    d2h(e, halves.data(), q->mb->vec, halves.size() * sizeof(float));
    for (uint32_t k = 0; k < n_batches; ++k) totals[k] = (uint64_t)halves[2 * k] + ((uint64_t)halves[2 * k + 1] << 16);
  }
}

// gather minibatch `k` of the last draw and compute its targets (dqn.rs:293-314)
static void dqn_build_minibatch(rl_dqn *q, uint32_t k, const DqnCountsDev &c, uint64_t total) {
  dqn_own_arrays(q);
  q->last_n_eps = c.n_eps;
  q->last_n_steps = c.n_steps;
  q->last_total_steps = total;
  q->last_batch_index = k;
  rl_traj *mb = q->mb;
  mb->d.n = c.n_steps;
  mb->d.T = 1;
  traj_plan(mb, c.n_steps);
  const size_t o = (size_t)k * q->max_eps;
  const bool td = q->cfg.target == RL_DQN_TARGET_ONE_STEP_TD;
  if (td && q->qnet->general) {
    // the builder's in-kernel forward is the fused module's; any other module: gather with rewards, successor codes and
    // successor observations (time slot 1), the module's layer kernels over the successor observations, then the targets
    launch_dqn_build_all(q->eng, q->rp, 1, c.n_eps, q->max_eps, q->d_ep_lane + o, q->d_ep_start + o, q->d_ep_len + o,
                         q->d_ep_off + o, q->d_counts + k, mb->d.obs, 0, mb->d.action, mb->d.adv, 0,
                         q->cfg.discount_factor, mb->d.flag);
    if (q->cap_q_next < 2ull * c.n_steps) {
      dfree(q->d_q_next);
      q->d_q_next = nullptr;
      q->d_q_next = dalloc<float>(2ull * q->max_steps_mb);
      q->cap_q_next = 2ull * q->max_steps_mb;
    }
    launch_gen_forward(mb, q->qnet, mb->d.obs + c.n_steps, (size_t)2 * c.n_steps, c.n_steps, q->d_q_next);
    launch_dqn_td_targets(q->eng, mb->d.adv, mb->d.flag, q->d_q_next, c.n_steps, q->cfg.discount_factor);
    return;
  }
  launch_dqn_build_minibatch(q->eng, q->rp, c.n_eps, q->d_ep_lane + o, q->d_ep_start + o, q->d_ep_len + o,
                             q->d_ep_off + o, mb->d.obs, (size_t)2 * c.n_steps, mb->d.action, mb->d.adv,
                             q->cfg.discount_factor, td ? 1 : 0, q->qnet);
}

static void dqn_sample_minibatch(rl_dqn *q, int sequential) {
  std::vector<DqnCountsDev> counts;
  std::vector<uint64_t> totals;
  dqn_draw_minibatches(q, sequential, 1, counts, totals);
  dqn_build_minibatch(q, 0, counts[0], totals[0]);
}

// gradient of mean((Q(s)[a] - target)^2) over the current minibatch -> mb->vec[0..P), loss sum -> mb->vec[P]
// `step_opt` != nullptr: also take the optimiser step, recording the loss in slot `loss_slot`; without an all-reduce
// between them the reduction and the (elementwise) step are one launch
static void dqn_gradient(rl_dqn *q, rl_adam *step_opt = nullptr, int loss_slot = -1) {
  rl_traj *mb = q->mb;
  uint32_t P = (uint32_t)q->qnet->P;
  uint32_t rowsA, rowsB;
  // targets given (reward-to-go, or one-step TD built beforehand): two critic-step channels per SIMD
  // (kernels_critic.hip, k_critic_step_mfma<2>; RL_DQN_SINGLE_WAVE=1 keeps rounds 3-5's kernel for A/B runs); targets
  // formed inside the launch (one-step TD on the current network): k_dqn_step_bf16
  static const bool single_wave = std::getenv("RL_DQN_SINGLE_WAVE") != nullptr;
  if (q->eng->kernel_variant == 0 && !q->td_in_kernel && !single_wave &&
      launch_dqn_step_pair(mb, q->qnet, q->last_total_steps)) {
    rowsA = rowsB = mb->nbV2;
  } else if (q->eng->kernel_variant == 0 &&
             launch_dqn_step_bf16(mb, q->qnet, q->last_total_steps, q->td_in_kernel, q->cfg.discount_factor)) {
    rowsA = rowsB = mb->nbV2;
  } else {
    launch_policy_pass(mb, q->qnet, PASS_DQN, nullptr, q->last_total_steps, nullptr);
    launch_mlp_backward(mb, q->qnet, nullptr);
    rowsA = mb->nbA;
    rowsB = mb->nbB;
  }
  if (step_opt && !q->eng->has_collective()) {
    launch_reduce_adam(mb, step_opt, rowsA, rowsB, loss_slot, q->last_total_steps);
    return;
  }
  launch_reduce(mb, P, true, true, rowsA, rowsB);
  rl_allreduce_sum_f32(q->eng, mb->vec, P + 4);
  if (step_opt) launch_adam_step(mb, step_opt, loss_slot, q->last_total_steps);
}

int32_t rl_dqn_update(rl_dqn *q, rl_dqn_update_stats *stats, float *losses_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q, "NULL argument");
    rl_engine *e = q->eng;
    // self.global_steps = sum of total_step_count over the buffers (dqn.rs:276); every lane of every rank has
    // taken the same number of steps and the horizon rule drops none
    q->global_steps = q->steps_per_lane * (uint64_t)q->rp.N * (uint64_t)e->n_ranks;
    uint64_t K = q->cfg.opt_steps_per_update;
    // Reward-to-go targets do not depend on the network: whole chunks of minibatches are gathered, targets included, in
    // one launch each, and an optimisation step is a gradient launch and a reduction + Adam launch.  (One-step TD targets
    // use the current network: those minibatches are still built one step at a time.)
    const uint64_t D = q->rp.D, cap = q->max_steps_mb;
    // One-step TD targets use the current network: on the matrix-pipe kernel they are formed inside the gradient launch
    // (a second forward over the successor observations, which the gather leaves in time slot 1 of the workspace); the
    // other kernels still build those minibatches one step at a time.
    const bool td = q->cfg.target == RL_DQN_TARGET_ONE_STEP_TD;
    const bool td_fused = td && e->kernel_variant == 0 && D == 5 && q->qnet->hidden == 128 && !q->qnet->general;
    const bool all_at_once = K > 1 && (!td || td_fused) && K * cap * (8 * D + 6) <= (8ull << 30);
    if (all_at_once && !q->all_obs) {
      q->all_obs = dalloc<float>(K * D * 2 * cap);
      q->all_target = dalloc<float>(K * cap);
      q->all_action = dalloc<uint8_t>(K * cap);
    }
    if (all_at_once && td && !q->all_flag) q->all_flag = dalloc<uint8_t>(K * cap);
    // The draws do not depend on the network either, but they are one sequential chain through the agent's Prng (12 us
    // per minibatch on one CU).  One rank: the chain runs on a second stream in chunks of 2, 4, 8, ... minibatches
    // while the main stream trains on the chunks already drawn — the draw is faster than the training, so only the first
    // chunk is waited for.  (Several ranks agree on the counts through a collective first; per-kernel profiling keeps
    // everything on the profiled stream.)
    const bool pipelined = all_at_once && e->n_ranks == 1 && !e->profiling;
    std::vector<DqnCountsDev> counts;
    std::vector<uint64_t> totals;
    std::vector<uint32_t> chunk_end;  // minibatches [chunk_end[c - 1], chunk_end[c]) form chunk c
    if (pipelined) {
      for (uint64_t first = 0, size = 2; first < K; first += size, size *= 2) chunk_end.push_back((uint32_t)(first + size < K ? first + size : K));
      if (!q->draw_stream) RL_HIP_CHECK(hipStreamCreateWithFlags(&q->draw_stream, hipStreamNonBlocking));
      if (!q->main_event) RL_HIP_CHECK(hipEventCreateWithFlags(&q->main_event, hipEventDisableTiming));
      if (!q->h_counts) RL_HIP_CHECK(hipHostMalloc((void **)&q->h_counts, K * sizeof(DqnCountsDev), hipHostMallocDefault));
      while (q->draw_events.size() < chunk_end.size()) {
        hipEvent_t ev;
        RL_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        q->draw_events.push_back(ev);
      }
      // the store is as the main stream's collection left it
      RL_HIP_CHECK(hipEventRecord(q->main_event, e->stream));
      RL_HIP_CHECK(hipStreamWaitEvent(q->draw_stream, q->main_event, 0));
      for (size_t c = 0; c < chunk_end.size(); ++c) {
        const uint32_t first = c ? chunk_end[c - 1] : 0, size = chunk_end[c] - first;
        const size_t o = (size_t)first * q->max_eps;
        launch_dqn_sample(e, q->draw_stream, q->rp, dqn_key(q), q->d_agent_pos, (uint32_t)q->cfg.minibatch_steps,
                          q->max_eps, q->d_ep_lane + o, q->d_ep_start + o, q->d_ep_len + o, q->d_ep_off + o,
                          q->d_counts + first, 0, size);
        RL_HIP_CHECK(hipMemcpyAsync(q->h_counts + first, q->d_counts + first, size * sizeof(DqnCountsDev),
                                    hipMemcpyDeviceToHost, q->draw_stream));
        RL_HIP_CHECK(hipEventRecord(q->draw_events[c], q->draw_stream));
      }
      counts.resize(K);
      totals.resize(K);
    } else {
      if (K) dqn_draw_minibatches(q, 0, (uint32_t)K, counts, totals);
      chunk_end.push_back((uint32_t)K);
    }
    rl_traj *mb = q->mb;
    // With pipelined draws a sampler error can surface in a LATER chunk, after the earlier chunks' optimisation steps
    // have run: the update is all or nothing, so the network and the optimiser state are saved first and put back then
    // (four device copies of <= 4 KB; the one-launch draw validates every minibatch before the first step).
    const uint64_t Pq = q->qnet->P, host_step0 = q->opt->host_step;
    // (the fused gradient kernel's range guard reports after the steps it could not vouch for have run: the same
    // all-or-nothing rule — saved, and put back when the guard fires)
    const bool fused_shape = e->kernel_variant == 0 && D == 5 && q->qnet->hidden == 128 && !q->qnet->general;
    const bool snapshot = pipelined || fused_shape;
    if (snapshot) {
      if (!q->snap) q->snap = dalloc<float>(3 * Pq + 2);
      RL_HIP_CHECK(hipMemcpyAsync(q->snap, q->qnet->d_params, Pq * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
      RL_HIP_CHECK(hipMemcpyAsync(q->snap + Pq, q->opt->d_m, Pq * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
      RL_HIP_CHECK(hipMemcpyAsync(q->snap + 2 * Pq, q->opt->d_v, Pq * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
      RL_HIP_CHECK(hipMemcpyAsync(q->snap + 3 * Pq, q->opt->d_step, sizeof(uint64_t), hipMemcpyDeviceToDevice, e->stream));
    }
    try {
      for (size_t c = 0; c < chunk_end.size(); ++c) {
        const uint32_t first = c ? chunk_end[c - 1] : 0, size = chunk_end[c] - first;
        if (pipelined) {
          RL_HIP_CHECK(hipEventSynchronize(q->draw_events[c]));
          const char *what = "";
          const int32_t code = dqn_check_counts(q, q->h_counts + first, size, &what);
          if (code != RL_OK) throw RlError(code, what);
          // (test facility: RELEARN_DQN_FAIL_CHUNK=c fails chunk c as a sampler error would, tests/test_gpu_dqn.py)
          if (const char *inj = std::getenv("RELEARN_DQN_FAIL_CHUNK"))
            if ((size_t)std::atoi(inj) == c) throw RlError(RL_ERR_INVALID_ARGUMENT, "injected sampler failure");
          for (uint32_t k = first; k < first + size; ++k) {
            counts[k] = q->h_counts[k];
            totals[k] = counts[k].n_steps;
          }
        }
        if (all_at_once && size) {
          uint32_t widest = 0;
          for (uint32_t k = first; k < first + size; ++k) widest = counts[k].n_eps > widest ? counts[k].n_eps : widest;
          const size_t o = (size_t)first * q->max_eps;
          launch_dqn_build_all(e, q->rp, size, widest, q->max_eps, q->d_ep_lane + o, q->d_ep_start + o, q->d_ep_len + o,
                               q->d_ep_off + o, q->d_counts + first, q->all_obs + first * D * 2 * cap,
                               (size_t)(D * 2 * cap), q->all_action + first * cap, q->all_target + first * cap,
                               (size_t)cap, q->cfg.discount_factor, td ? q->all_flag + first * cap : (uint8_t *)nullptr);
        }
        for (uint64_t k = first; k < first + size; ++k) {
          if (all_at_once) {
            q->last_n_eps = counts[k].n_eps;
            q->last_n_steps = counts[k].n_steps;
            q->last_total_steps = totals[k];
            q->last_batch_index = (uint32_t)k;
            mb->d.n = counts[k].n_steps;
            mb->d.T = 1;
            traj_plan(mb, counts[k].n_steps);
            mb->d.obs = q->all_obs + k * D * 2 * cap;
            mb->d.action = q->all_action + k * cap;
            mb->d.adv = q->all_target + k * cap;
            if (td) mb->d.flag = q->all_flag + k * cap;
            q->td_in_kernel = td;
          } else {
            dqn_build_minibatch(q, (uint32_t)k, counts[k], totals[k]);
          }
          dqn_gradient(q, q->opt, (int)k);
        }
      }
      // the range guard's word is read here, inside the all-or-nothing scope (the stream is drained first: the word is
      // host memory the kernels write across the bus)
      if (fused_shape && K) {
        sync(e);
        range_check(q->mb, 1u << RL_GUARD_POLICY);
      }
    } catch (...) {
      if (snapshot) {
        if (pipelined) (void)hipStreamSynchronize(q->draw_stream);  // the later chunks' draws still advance the agent's Prng
        (void)hipMemcpyAsync(q->qnet->d_params, q->snap, Pq * sizeof(float), hipMemcpyDeviceToDevice, e->stream);
        wimg_invalidate(q->qnet);
        (void)hipMemcpyAsync(q->opt->d_m, q->snap + Pq, Pq * sizeof(float), hipMemcpyDeviceToDevice, e->stream);
        (void)hipMemcpyAsync(q->opt->d_v, q->snap + 2 * Pq, Pq * sizeof(float), hipMemcpyDeviceToDevice, e->stream);
        (void)hipMemcpyAsync(q->opt->d_step, q->snap + 3 * Pq, sizeof(uint64_t), hipMemcpyDeviceToDevice, e->stream);
        (void)hipStreamSynchronize(e->stream);
        q->opt->host_step = host_step0;
      }
      dqn_own_arrays(q);
      q->last_n_eps = q->last_n_steps = 0;  // no minibatch to read after a failed update
      throw;
    }
    // (after an all-at-once update `mb` keeps looking at the last minibatch: rl_dqn_minibatch_read / _gradient work;
    // with in-kernel TD targets the workspace holds rewards, not targets: the last minibatch is built once more by the
    // one-at-a-time builder, with the targets of the parameters the update ends on)
    if (all_at_once && td && K) dqn_build_minibatch(q, (uint32_t)(K - 1), counts[K - 1], totals[K - 1]);
    std::vector<float> h(K ? K : 1, 0.0f);
    if (K) d2h(e, h.data(), q->mb->losses, K * sizeof(float));
    range_check(q->mb, 1u << RL_GUARD_POLICY);
    if (losses_out && K) std::memcpy(losses_out, h.data(), K * sizeof(float));
    if (stats) {
      stats->opt_steps = K;
      stats->loss_first = K ? (double)h[0] : 0.0;
      stats->loss_last = K ? (double)h[K - 1] : 0.0;
      stats->global_steps = q->global_steps;
      stats->last_minibatch_steps = q->last_n_steps;
      stats->last_minibatch_episodes = q->last_n_eps;
    }
  });
}

static void replay_field(const rl_dqn *q, int32_t field, void **ptr, uint64_t *bytes) {
  const ReplayDev &r = q->rp;
  uint64_t N = r.N, C = r.C, E = r.E, D = r.D;
  *ptr = nullptr;
  switch (field) {
    case RL_REPLAY_HEAD: *ptr = r.head; *bytes = N * 4; break;
    case RL_REPLAY_COUNT: *ptr = r.count; *bytes = N * 4; break;
    case RL_REPLAY_EP_HEAD: *ptr = r.ep_head; *bytes = N * 4; break;
    case RL_REPLAY_EP_COUNT: *ptr = r.ep_count; *bytes = N * 4; break;
    case RL_REPLAY_TOTAL: *ptr = r.total; *bytes = N * 4; break;
    case RL_REPLAY_EP_END: *ptr = r.ep_end; *bytes = E * N * 4; break;
    // step data lives in records (engine.hpp): *ptr stays NULL, the read goes through launch_replay_planes
    case RL_REPLAY_OBS: *bytes = D * C * N * 4; break;
    case RL_REPLAY_NEXT_OBS: *bytes = D * C * N * 4; break;
    case RL_REPLAY_ACTION: *bytes = C * N; break;
    case RL_REPLAY_REWARD: *bytes = C * N * 4; break;
    case RL_REPLAY_FLAG: *bytes = C * N; break;
    case RL_REPLAY_ACTOR_POS: *ptr = r.actor_pos; *bytes = N * 8; break;
    case RL_REPLAY_LAST_FLAGS: *ptr = q->d_flags; *bytes = q->last_horizon * N; break;
    default: throw RlError(RL_ERR_INVALID_ARGUMENT, "unknown replay field");
  }
}

int32_t rl_dqn_replay_field_bytes(const rl_dqn *q, int32_t field, uint64_t *bytes) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && bytes, "NULL argument");
    void *p;
    replay_field(q, field, &p, bytes);
  });
}

int32_t rl_dqn_replay_read(rl_dqn *q, int32_t field, void *host, uint64_t bytes) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && host, "NULL argument");
    void *p;
    uint64_t need;
    replay_field(q, field, &p, &need);
    RL_REQUIRE(bytes == need, "byte count mismatch for replay field");
    if (!bytes) return;
    if (p) {
      d2h(q->eng, host, p, bytes);
      return;
    }
    RL_HIP_CHECK(hipSetDevice(q->eng->device));
    void *planes = dalloc<uint8_t>(bytes);
    try {
      launch_replay_planes(q->eng, q->rp, field, planes);
      d2h(q->eng, host, planes, bytes);
    } catch (...) {
      dfree(planes);
      throw;
    }
    dfree(planes);
  });
}

int32_t rl_dqn_minibatch_sample(rl_dqn *q, int32_t sequential, uint64_t *n_episodes_out, uint64_t *n_steps_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q, "NULL argument");
    dqn_sample_minibatch(q, sequential);
    if (n_episodes_out) *n_episodes_out = q->last_n_eps;
    if (n_steps_out) *n_steps_out = q->last_n_steps;
  });
}

int32_t rl_dqn_minibatch_read(rl_dqn *q, int32_t field, void *host, uint64_t bytes) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && host, "NULL argument");
    uint64_t ne = q->last_n_eps, ns = q->last_n_steps, D = q->rp.D;
    RL_REQUIRE(ns > 0, "no minibatch has been sampled");
    rl_engine *e = q->eng;
    switch (field) {
      case RL_MB_EP_LANE: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_lane + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_EP_START: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_start + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_EP_LEN: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_len + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_EP_OFFSET: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_off + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_OBS: {
        RL_REQUIRE(bytes == D * ns * 4, "byte count mismatch");
        // feature planes are 2 * n_steps apart in the workspace (T = 1 trajectory layout)
        for (uint64_t d = 0; d < D; ++d)
          d2h(e, (char *)host + d * ns * 4, q->mb->d.obs + d * 2 * ns, ns * 4);
        break;
      }
      case RL_MB_ACTION: RL_REQUIRE(bytes == ns, "byte count mismatch"); d2h(e, host, q->mb->d.action, bytes); break;
      case RL_MB_TARGET: RL_REQUIRE(bytes == ns * 4, "byte count mismatch"); d2h(e, host, q->mb->d.adv, bytes); break;
      default: throw RlError(RL_ERR_INVALID_ARGUMENT, "unknown minibatch field");
    }
  });
}

int32_t rl_dqn_minibatch_gradient(rl_dqn *q, float *grad_out, float *loss_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && grad_out, "NULL argument");
    RL_REQUIRE(q->last_n_steps > 0, "no minibatch has been sampled");
    uint32_t P = (uint32_t)q->qnet->P;
    dqn_gradient(q);
    std::vector<float> h(P + 4);
    d2h(q->eng, h.data(), q->mb->vec, (P + 4) * sizeof(float));
    range_check(q->mb, 1u << RL_GUARD_POLICY);
    std::memcpy(grad_out, h.data(), P * sizeof(float));
    if (loss_out) *loss_out = (float)((double)h[P] / (double)q->last_total_steps);
  });
}

int32_t rl_dqn_agent_rng_pos(rl_dqn *q, uint64_t *pos_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && pos_out, "NULL argument");
    d2h(q->eng, pos_out, q->d_agent_pos, sizeof(uint64_t));
  });
}

}  // extern "C"
