// abi_update.hip — extern "C" entry points of include/relearn_hip.h, part: policy / critic updates (TRPO, PPO, REINFORCE, value fitting) (host side only; kernels live in kernels_*.hip).
#include <chrono>

#include "abi_internal.hpp"

extern "C" {

// ---------------------------------------------------------------- recurrent gradient passes
// policy: teacher-forced forward (activation record) -> d loss / d logits -> [backward through time -> weight
// gradients -> reduce] -> traj->vec[0..P) and the per-sample sums in vec[P..P+4)
// (widths other than the kernels' run on the module's zero-padded twin: SeqScope; the gradient comes back in the twin's
// layout and is gathered into the module's flat order before anything else touches the vector)
static void seq_gradient_to_flat_order(const rl_mlp *mod, rl_traj *traj) {
  if (mod->exec == nullptr) return;
  launch_seq_unpad(mod, traj->vec, mod->x_tmp);
  RL_HIP_CHECK(hipMemcpyAsync(traj->vec, mod->x_tmp, mod->P * sizeof(float), hipMemcpyDeviceToDevice, traj->eng->stream));
}

// where a training forward of `mod` keeps its activation record (non-NULL asks the forward for one)
static float *seq_record(rl_traj *traj, const rl_mlp *mod) {
  return mod->lane_kernels() ? traj->seq.stack.rec : traj->seq.act;
}

static void seq_policy_pass(rl_mlp *policy, rl_traj *traj, int mode, bool backward, float lo, float hi) {
  SeqScope sc(traj, policy);
  seq_ensure(traj, sc.x, true);
  uint32_t P = (uint32_t)policy->P;
  launch_gru_seq_forward(traj, sc.x, traj->seq.out, nullptr, backward ? seq_record(traj, sc.x) : nullptr);
  launch_seq_policy_dlogits(traj, mode, b_total(traj), lo, hi);
  if (backward) {
    launch_gru_backward(traj, sc.x);
    seq_gradient_to_flat_order(policy, traj);
  }
  launch_reduce(traj, P, false, true, 0, traj->nbB);
  if (backward) rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
  else rl_allreduce_sum_f32(traj->eng, traj->vec + P, 4);
}

// (loss, KL) of the current parameters against log pi_0: forward without a record -> sums in vec[P..P+4)
static void seq_policy_eval(rl_mlp *policy, rl_traj *traj, const int32_t *d_skip) {
  SeqScope sc(traj, policy);
  seq_ensure(traj, sc.x, true);
  uint32_t P = (uint32_t)policy->P;
  launch_gru_seq_forward(traj, sc.x, traj->seq.out, nullptr, nullptr, d_skip);
  launch_seq_policy_dlogits(traj, PASS_EVAL, b_total(traj), 0.0f, 0.0f, d_skip);
  launch_reduce(traj, P, false, true, 0, traj->nbB);
  rl_allreduce_sum_f32(traj->eng, traj->vec + P, 4);
}

// Fisher-vector product with the tangent d_v at the parameters whose activation record is in place (the last
// seq_policy_pass with backward = true): vec[0..P) <- J^T (diag(p) - p p^T) J v / B
static void seq_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *d_v, const int32_t *d_skip) {
  SeqScope sc(traj, policy);
  seq_ensure(traj, sc.x, true);
  const float *tangent = d_v;
  if (policy->exec != nullptr) {  // the tangent in the twin's layout (its padding entries stay zero)
    launch_seq_pad(policy, policy->x_tan, d_v);
    tangent = policy->x_tan;
  }
  launch_gru_tangent(traj, sc.x, tangent, b_total(traj), d_skip);
  launch_gru_backward(traj, sc.x, d_skip);
  seq_gradient_to_flat_order(policy, traj);
  rl_allreduce_sum_f32(traj->eng, traj->vec, (uint32_t)policy->P);
}

static void seq_critic_pass(rl_mlp *critic, rl_traj *traj) {
  SeqScope sc(traj, critic);
  seq_ensure(traj, sc.x, true);
  uint32_t P = (uint32_t)critic->P;
  launch_gru_seq_forward(traj, sc.x, traj->seq.out, nullptr, seq_record(traj, sc.x));
  launch_seq_critic_dvalues(traj, b_total(traj));
  launch_gru_backward(traj, sc.x);
  seq_gradient_to_flat_order(critic, traj);
  launch_reduce(traj, P, false, true, 0, traj->nbB);
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

// ---------------------------------------------------------------- TRPO
int32_t rl_trpo_config_default(rl_trpo_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    // ConjugateGradientOptimizerConfig::default (conjugate_gradient.rs:55-65), TrpoConfig::default (trpo.rs:29-41)
    c->iterations = 10;
    c->max_backtracks = 15;
    c->backtrack_ratio = 0.8;
    c->hpv_reg_coeff = 1e-5;
    c->max_policy_step_kl = 0.01;
    c->accept_violation = 0;
  });
}

static void check_policy(const rl_mlp *policy, const rl_traj *traj) {
  RL_REQUIRE(policy && traj, "NULL argument");
  RL_REQUIRE(policy->eng == traj->eng, "handles belong to different engines");
  RL_REQUIRE(policy->in_dim == traj->d.D && policy->out_dim == 2, "policy shape does not match the trajectory");
}

// gradient pass: PASS_INIT -> backward -> reduce(A+B) -> allreduce
static void run_policy_gradient(rl_mlp *policy, rl_traj *traj) {
  if (rl_module_is_recurrent(policy->kind)) return seq_policy_pass(policy, traj, PASS_INIT, true, 0.0f, 0.0f);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 && launch_policy_v2(traj, policy, PASS_INIT, nullptr, b_total(traj), nullptr)) {
    launch_reduce(traj, P, true, true, traj->last_rows, traj->last_rows);
  } else {
    launch_policy_pass(traj, policy, PASS_INIT, nullptr, b_total(traj), nullptr);
    launch_mlp_backward(traj, policy, nullptr);
    launch_reduce(traj, P, true, true, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

// (loss, KL) of the current parameters against lp0: PASS_EVAL -> reduce(B) -> allreduce
static void run_policy_eval(rl_mlp *policy, rl_traj *traj, const int32_t *d_skip) {
  if (rl_module_is_recurrent(policy->kind)) return seq_policy_eval(policy, traj, d_skip);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 && launch_policy_v2(traj, policy, PASS_EVAL, nullptr, b_total(traj), d_skip)) {
    launch_reduce(traj, P, false, true, traj->last_rows, traj->last_rows);
  } else {
    launch_policy_pass(traj, policy, PASS_EVAL, nullptr, b_total(traj), d_skip);
    launch_reduce(traj, P, false, true, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec + P, 4);
}

// Fisher/Hessian-vector product pass with tangent d_v: PASS_JVP -> backward -> reduce(A) -> allreduce
static void run_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *d_v, const int32_t *d_skip) {
  if (rl_module_is_recurrent(policy->kind)) return seq_policy_fvp(policy, traj, d_v, d_skip);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 && launch_policy_v2(traj, policy, PASS_JVP, d_v, b_total(traj), d_skip)) {
    launch_reduce(traj, P, true, false, traj->last_rows, traj->last_rows);
  } else {
    launch_policy_pass(traj, policy, PASS_JVP, d_v, b_total(traj), d_skip);
    launch_mlp_backward(traj, policy, d_skip);
    launch_reduce(traj, P, true, false, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec, P);
}

// Trpo::update (trpo.rs:97-164) on the engine's current stream, in two halves.  The head enqueues everything up to and
// including the first two line-search candidates — about fifty launches without a host round trip; the tail reads the
// acceptance flag back every second candidate and collects the statistics.  (rl_actor_critic_update enqueues the other
// chain's launches between the two: the host is then never the reason one chain's kernels start late.)
static inline void trpo_candidate(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg, uint64_t i, double ratio) {
  launch_ls_set_params(traj, policy, ratio);
  run_policy_eval(policy, traj, &traj->trpo->ls_accepted);
  launch_ls_check(traj, (uint32_t)policy->P, b_total(traj), (int)i, ratio, cfg->max_policy_step_kl);
}

struct TrpoProgress {
  uint64_t next = 0;   // line-search candidates [0, next) are enqueued
  double ratio = 1.0;  // backtrack_ratio.powi(next - 1)
};

static TrpoProgress trpo_update_head(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg) {
  uint32_t P = (uint32_t)policy->P;
  uint64_t Bt = b_total(traj);
  float reg = (float)cfg->hpv_reg_coeff;
  // loss gradient at theta0 and CG prologue
  run_policy_gradient(policy, traj);
  launch_trpo_begin(traj, policy, Bt);
  // x = A^-1 g by `iterations` CG steps (early exit handled on the device)
  for (uint64_t it = 0; it < cfg->iterations; ++it) {
    run_policy_fvp(policy, traj, traj->cg_p, &traj->trpo->cg_done);
    launch_cg_step(traj, P, reg, 1e-10f);
  }
  launch_cg_finish(traj, P);
  // step size from x^T A x
  run_policy_fvp(policy, traj, traj->cg_x, nullptr);
  launch_step_size(traj, policy, reg, cfg->max_policy_step_kl);
  // backtracking line search: the first two candidates
  TrpoProgress pr;
  for (; pr.next < cfg->max_backtracks && pr.next < 2; ++pr.next) {
    if (pr.next > 0) pr.ratio *= cfg->backtrack_ratio;  // backtrack_ratio.powi(i)
    trpo_candidate(policy, traj, cfg, pr.next, pr.ratio);
  }
  return pr;
}

static void trpo_update_tail(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg, TrpoProgress pr,
                             rl_trpo_stats *stats) {
  rl_engine *e = traj->eng;
  double ratio = pr.ratio;
  for (uint64_t i = pr.next; i < cfg->max_backtracks; ++i) {
    // once a candidate is accepted the remaining iterations are no-ops on the device (every launch tests the flag);
    // reading the flag back every second candidate spares their launches — and, with several ranks, their
    // all-reduces.  Replicas are identical, so every rank leaves the loop at the same iteration.
    if ((i & 1) == 0) {
      int32_t accepted = 0;
      d2h(e, &accepted, &traj->trpo->ls_accepted, sizeof(accepted));
      if (accepted != 0) break;
    }
    ratio *= cfg->backtrack_ratio;
    trpo_candidate(policy, traj, cfg, i, ratio);
  }
  launch_ls_finalize(traj, policy, cfg->max_policy_step_kl, cfg->accept_violation);
  TrpoStateDev h;
  d2h(e, &h, traj->trpo, sizeof(h));
  ipc_check(e);
  range_check(traj, 1u << RL_GUARD_POLICY);
  stats->entropy = (double)h.entropy;
  stats->step_size = h.step_size;
  stats->loss_initial = (double)h.loss0;
  stats->loss_final = (double)h.ls_loss;
  stats->constraint_val_final = (double)h.ls_kl;
  stats->step_scale = h.ls_accepted ? h.ls_ratio : 0.0;
  stats->num_backtracks = h.ls_accepted ? (int64_t)h.ls_index : -1;
  stats->status = h.status;
  stats->cg_iterations = h.cg_iters;
}

static void trpo_update_impl(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg, rl_trpo_stats *stats) {
  trpo_update_tail(policy, traj, cfg, trpo_update_head(policy, traj, cfg), stats);
}

static void trpo_raise_nan(const rl_trpo_stats *stats) {
  if (stats->status == RL_OPT_NAN_LOSS || stats->status == RL_OPT_NAN_CONSTRAINT)
    throw RlError(RL_ERR_OPT_NAN, stats->status == RL_OPT_NAN_LOSS ? "NaN loss in policy optimization"
                                                                   : "NaN constraint in policy optimization");
}

int32_t rl_trpo_update(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg, rl_trpo_stats *stats) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(cfg && stats, "NULL argument");
    trpo_update_impl(policy, traj, cfg, stats);
    trpo_raise_nan(stats);
  });
}

int32_t rl_policy_gradient(rl_mlp *policy, rl_traj *traj, float *grad_out, float *loss_out, float *entropy_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(grad_out, "grad_out is NULL");
    uint32_t P = (uint32_t)policy->P;
    run_policy_gradient(policy, traj);
    std::vector<float> h(P + 4);
    d2h(traj->eng, h.data(), traj->vec, (P + 4) * sizeof(float));
    range_check(traj, 1u << RL_GUARD_POLICY);
    std::memcpy(grad_out, h.data(), P * sizeof(float));
    double inv_B = 1.0 / (double)b_total(traj);
    if (loss_out) *loss_out = (float)(-((double)h[P] * inv_B));
    if (entropy_out) *entropy_out = (float)((double)h[P + 1] * inv_B);
  });
}

int32_t rl_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *v, float reg, float *out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(v && out, "NULL argument");
    uint32_t P = (uint32_t)policy->P;
    run_policy_gradient(policy, traj);  // the product is taken at the current parameters: refresh log pi_0
    h2d(traj->eng, traj->cg_x, v, P * sizeof(float));  // (the recurrent pass above has grown the workspace)
    run_policy_fvp(policy, traj, traj->cg_x, nullptr);
    std::vector<float> h(P);
    d2h(traj->eng, h.data(), traj->vec, P * sizeof(float));
    range_check(traj, 1u << RL_GUARD_POLICY);
    for (uint32_t i = 0; i < P; ++i) out[i] = h[i] + reg * v[i];
  });
}

int32_t rl_policy_loss_kl(rl_mlp *policy, rl_traj *traj, const float *params0, float *loss_out, float *kl_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(params0 && loss_out && kl_out, "NULL argument");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj);
    // lp0 under params0, then evaluate the current parameters against it
    std::vector<float> cur(P);
    d2h(e, cur.data(), policy->d_params, P * sizeof(float));
    h2d(e, policy->d_params, params0, P * sizeof(float));
    wimg_invalidate(policy);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
    run_policy_gradient(policy, traj);  // fills lp0 under params0
    h2d(e, policy->d_params, cur.data(), P * sizeof(float));
    wimg_invalidate(policy);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
    run_policy_eval(policy, traj, nullptr);
    float h[4];
    d2h(e, h, traj->vec + P, sizeof(h));
    range_check(traj, 1u << RL_GUARD_POLICY);
    double inv_B = 1.0 / (double)Bt;
    *loss_out = (float)(-((double)h[0] * inv_B));
    *kl_out = (float)((double)h[1] * inv_B);
  });
}

// ---------------------------------------------------------------- critic
int32_t rl_adam_config_default(rl_adam_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    c->learning_rate = 1e-3;  // AdamConfig::default (coptimizer.rs:147-156)
    c->beta1 = 0.9;
    c->beta2 = 0.999;
    c->weight_decay = 0.0;
    c->eps = 1e-8;  // libtorch AdamOptions default
  });
}

int32_t rl_adam_create(rl_mlp *module, const rl_adam_config *cfg, rl_adam **out) {
  return guarded(module ? module->eng : nullptr, [&] {
    RL_REQUIRE(module && cfg && out, "NULL argument");
    *out = nullptr;
    rl_engine *e = module->eng;
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_adam> o(new rl_adam());
    o->eng = e;
    o->mod = module;
    o->cfg = *cfg;
    o->error_epoch = e->error_epoch;
    o->d_m = dalloc<float>(module->P);
    o->d_v = dalloc<float>(module->P);
    o->d_step = dalloc<uint64_t>(1);
    RL_HIP_CHECK(hipMemsetAsync(o->d_m, 0, module->P * sizeof(float), e->stream));
    RL_HIP_CHECK(hipMemsetAsync(o->d_v, 0, module->P * sizeof(float), e->stream));
    RL_HIP_CHECK(hipMemsetAsync(o->d_step, 0, sizeof(uint64_t), e->stream));
    sync(e);
    e->live_handles += 1;
    *out = o.release();
  });
}

int32_t rl_adam_destroy(rl_adam *o) {
  if (!o) return RL_OK;
  (void)hipSetDevice(o->eng->device);
  (void)hipStreamSynchronize(o->eng->stream);
  (void)hipStreamSynchronize(o->eng->aux_stream);  // (a critic chain left in flight by rl_actor_critic_update_begin)
  dfree(o->d_m);
  dfree(o->d_v);
  dfree(o->d_step);
  rl_engine *eng = o->eng;
  delete o;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_adam_step_host(rl_adam *o, const float *grad) {
  return guarded(o ? o->mod->eng : nullptr, [&] {
    RL_REQUIRE(o && grad, "NULL argument");
    rl_engine *e = o->mod->eng;
    float *d_g = dalloc<float>(o->mod->P);
    try {
      h2d(e, d_g, grad, o->mod->P * sizeof(float));
      launch_adam_step_vec(o, d_g);
      sync(e);
    } catch (...) {
      dfree(d_g);
      throw;
    }
    dfree(d_g);
  });
}

static void check_critic(const rl_mlp *critic, const rl_traj *traj) {
  RL_REQUIRE(critic && traj, "NULL argument");
  RL_REQUIRE(critic->eng == traj->eng, "handles belong to different engines");
  RL_REQUIRE(critic->in_dim == traj->d.D && critic->out_dim == 1, "critic shape does not match the trajectory");
}

// per-workgroup partial sums of the critic's MSE gradient and loss -> slabA / slabB (feed-forward modules)
static void critic_slabs(rl_mlp *critic, rl_traj *traj, uint32_t *rowsA, uint32_t *rowsB) {
  if (traj->eng->kernel_variant != 1 && launch_critic_step_v2(traj, critic, b_total(traj))) {
    *rowsA = *rowsB = traj->last_rows;
  } else {
    launch_critic_fwd(traj, critic, b_total(traj));
    launch_mlp_backward(traj, critic, nullptr);
    *rowsA = traj->nbA;
    *rowsB = traj->nbB;
  }
}

static void run_critic_gradient(rl_mlp *critic, rl_traj *traj) {
  if (rl_module_is_recurrent(critic->kind)) return seq_critic_pass(critic, traj);
  uint32_t P = (uint32_t)critic->P, rowsA, rowsB;
  critic_slabs(critic, traj, &rowsA, &rowsB);
  launch_reduce(traj, P, true, true, rowsA, rowsB);
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

// n_backward_steps (src/torch/agents/mod.rs:35-72) of full-batch MSE against traj->d.tgt with Adam: the launches, on the
// engine's current stream, without any host synchronisation (feed-forward modules on a device-side collective)
static void critic_enqueue_steps(rl_mlp *critic, rl_adam *opt, rl_traj *traj, uint64_t opt_steps, uint64_t first = 0) {
  RL_REQUIRE(opt_steps <= traj->max_losses, "too many optimisation steps per update");
  uint64_t Bt = b_total(traj);
  // no separate all-reduce between the reduction and the (elementwise) optimiser step: one rank, or the peer-mailbox
  // transport, whose exchange runs inside the reduction launch
  const rl_engine *eng = traj->eng;
  const bool mailbox = eng->ipc_active && !eng->comm && !eng->loopback && !eng->host_allreduce &&
                       ipc_allreduce_fits(eng, critic->P + 4);
  const bool fused = critic->kind == RL_MODULE_MLP && (!eng->has_collective() || mailbox);
  for (uint64_t k = first; k < opt_steps; ++k) {
    if (fused) {  // no all-reduce between the reduction and the (elementwise) optimiser step: one launch
      uint32_t rowsA, rowsB;
      critic_slabs(critic, traj, &rowsA, &rowsB);
      launch_reduce_adam(traj, opt, rowsA, rowsB, (int)k, Bt);
    } else {
      run_critic_gradient(critic, traj);
      launch_adam_step(traj, opt, (int)k, Bt);
    }
  }
}

// the losses of the steps just enqueued (synchronises the engine's current stream)
static void critic_collect(rl_traj *traj, uint64_t opt_steps, rl_critic_stats *stats, float *losses_out) {
  if (!(stats || losses_out)) {
    // nothing to read back — but the call's contract is the same: a failed exchange or a range-guard violation of THESE
    // launches is this call's error, not a later entry point's, and the guard is re-armed for the next call (ADVICE
    // round 5: both used to be skipped here)
    sync(traj->eng);
    ipc_check(traj->eng);
    range_check(traj, 1u << RL_GUARD_CRITIC);
    return;
  }
  {
    std::vector<float> h(opt_steps ? opt_steps : 1);
    if (opt_steps) d2h(traj->eng, h.data(), traj->losses, opt_steps * sizeof(float));
    else sync(traj->eng);
    ipc_check(traj->eng);
    range_check(traj, 1u << RL_GUARD_CRITIC);
    if (losses_out && opt_steps) std::memcpy(losses_out, h.data(), opt_steps * sizeof(float));
    if (stats) {
      stats->steps = opt_steps;
      stats->loss_first = opt_steps ? (double)h[0] : 0.0;
      stats->loss_last = opt_steps ? (double)h[opt_steps - 1] : 0.0;
    }
  }
}

static void critic_opt_steps(rl_mlp *critic, rl_adam *opt, rl_traj *traj, uint64_t opt_steps, rl_critic_stats *stats,
                             float *losses_out) {
  critic_enqueue_steps(critic, opt, traj, opt_steps);
  critic_collect(traj, opt_steps, stats, losses_out);
}

int32_t rl_critic_update(rl_mlp *critic, rl_adam *opt, rl_traj *traj, uint64_t opt_steps, rl_critic_stats *stats,
                         float *losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_critic(critic, traj);
    RL_REQUIRE(opt && opt->mod == critic, "optimizer does not belong to this module");
    traj->d.tgt = traj->d.rtg;
    critic_opt_steps(critic, opt, traj, opt_steps, stats, losses_out);
  });
}

int32_t rl_values_opt_config_default(rl_values_opt_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    c->opt_steps_per_update = 80;                // ValuesOptConfig::default (critics/opt.rs:40-51)
    c->target = RL_VALUE_TARGET_REWARD_TO_GO;    // StepValueTarget::default (critics/mod.rs:211-215)
    c->discount_factor = 0.99f;                  // max_discount_factor
  });
}

static void check_values_opt(const rl_mlp *critic, const rl_adam *opt, const rl_traj *traj,
                             const rl_values_opt_config *cfg) {
  check_critic(critic, traj);
  RL_REQUIRE(opt && opt->mod == critic, "optimizer does not belong to this module");
  RL_REQUIRE(cfg, "cfg is NULL");
  RL_REQUIRE(cfg->target == RL_VALUE_TARGET_REWARD_TO_GO || cfg->target == RL_VALUE_TARGET_ONE_STEP_TD,
             "unknown value target");
  RL_REQUIRE(cfg->discount_factor >= 0.0f && cfg->discount_factor <= 1.0f, "discount factor must be in [0, 1]");
}

// targets: once, from the critic as it stands now (tch::no_grad, opt.rs:101-104) -> traj->td, selected as traj->d.tgt
static void values_opt_targets(rl_mlp *critic, rl_traj *traj, const rl_values_opt_config *cfg) {
  if (traj->td == nullptr) {
    RL_HIP_CHECK(hipSetDevice(traj->eng->device));
    traj->td = dalloc<float>((size_t)traj->d.T * traj->d.n);
  }
  traj->d.tgt = traj->td;
  if (cfg->target == RL_VALUE_TARGET_REWARD_TO_GO && traj->rtg_scan_valid && traj->rtg_gamma == cfg->discount_factor) {
    // the advantage pass's lane scan has already left exactly these targets in the return plane (same recursion, same
    // operations: k_gae_scan / k_value_targets_rtg, kernels_rollout.hip) — no second scan over the rewards
    traj->d.tgt = traj->d.rtg;
  } else if (cfg->target == RL_VALUE_TARGET_REWARD_TO_GO) {
    launch_value_targets(traj, nullptr, cfg->discount_factor);
  } else if (rl_module_is_recurrent(critic->kind)) {
    SeqScope sc(traj, critic);
    seq_ensure(traj, sc.x, false);
    launch_gru_seq_forward(traj, sc.x, traj->seq.out, traj->seq.succ, nullptr);
    launch_seq_value_targets(traj, cfg->discount_factor);
  } else if (critic->general) {
    launch_gen_values(traj, critic);
    launch_seq_value_targets(traj, cfg->discount_factor);
  } else {
    launch_values(traj, critic);
    launch_value_targets(traj, critic, cfg->discount_factor);
  }
  traj->last_targets = traj->d.tgt;  // (what RL_TRAJ_TARGETS reads)
}

int32_t rl_values_opt_update(rl_mlp *critic, rl_adam *opt, rl_traj *traj, const rl_values_opt_config *cfg,
                             rl_critic_stats *stats, float *losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_values_opt(critic, opt, traj, cfg);
    values_opt_targets(critic, traj, cfg);
    critic_opt_steps(critic, opt, traj, cfg->opt_steps_per_update, stats, losses_out);
  });
}

// ---------------------------------------------------------------- policy and critic update side by side
// The critic chain of rl_actor_critic_update runs on the engine's auxiliary stream with a workspace of its own (slab
// rows, reduced vector) and the auxiliary collective channel: inside this scope every launcher — they all read
// traj->eng->stream and the trajectory's workspace pointers at enqueue time — serves that chain.
struct AuxChain {
  rl_engine *e;
  rl_traj *t;
  AuxChain(rl_traj *traj) : e(traj->eng), t(traj) {
    if (t->aux_vec == nullptr) {  // all three or none: a failed allocation must not leave a half-made workspace behind
      RL_HIP_CHECK(hipSetDevice(e->device));
      float *v = nullptr;
      double *a = nullptr, *b = nullptr;
      try {
        v = dalloc<float>(t->Pmax + 4);
        a = dalloc<double>(t->cap_slabA);
        b = dalloc<double>(t->cap_slabB);
      } catch (...) {
        dfree(v);
        dfree(a);
        dfree(b);
        throw;
      }
      t->aux_vec = v;
      t->aux_slabA = a;
      t->aux_slabB = b;
      t->aux_cap_slabA = t->cap_slabA;
      t->aux_cap_slabB = t->cap_slabB;
    }
    swap();
    e->stream = e->aux_stream;
    e->chan = 1;
  }
  ~AuxChain() {
    e->stream = e->main_stream;
    e->chan = 0;
    swap();
  }
  void swap() {
    std::swap(t->vec, t->aux_vec);
    std::swap(t->slabA, t->aux_slabA);
    std::swap(t->slabB, t->aux_slabB);
    std::swap(t->cap_slabA, t->aux_cap_slabA);
    std::swap(t->cap_slabB, t->aux_cap_slabB);
    std::swap(t->last_rows, t->aux_last_rows);
  }
};

// may the two chains be in flight together?  Both must run on kernels that keep everything but the slab rows and the
// reduced vector out of the trajectory's shared workspace (the fused single-launch passes), and a multi-rank job needs
// a collective that can serve two streams at once: the mailbox channels, the second RCCL communicator, or one of the
// host-blocking test transports (which serialise the chains on the host, correctly).
static bool chains_can_overlap(const rl_mlp *policy, const rl_mlp *critic, const rl_traj *traj,
                               const rl_values_opt_config *ccfg) {
  const rl_engine *e = traj->eng;
  // (with several ranks the environment switch is the one the ranks agreed on when the collective was installed)
  if (e->n_ranks > 1 ? e->agreed_serial_env : std::getenv("RELEARN_SERIAL_UPDATE") != nullptr) return false;
  if (e->serial_update || e->kernel_variant == 1) return false;
  (void)ccfg;
  // (general hidden_sizes keep P-sized vectors and activation planes in workspaces both chains would share)
  auto fused_passes = [&](const rl_mlp *m) {
    return m->kind == RL_MODULE_MLP && !m->general && traj->d.D == 5 && m->hidden == 128;
  };
  if (!fused_passes(policy) || !fused_passes(critic)) return false;
  if (policy->P > traj->Pmax || critic->P > traj->Pmax) return false;  // (the auxiliary vector is sized Pmax + 4)
  if ((uint64_t)(traj->d.T + 1) * traj->d.n * 5 >= (1ull << 30)) return false;  // (the fused kernels' own limit)
  if (e->comm != nullptr && e->comm_aux == nullptr) return false;
  return true;
}

// _begin: fork, the whole critic chain onto the auxiliary stream, the TRPO chain on the main stream, the join event
// recorded — and back to the caller with the critic chain possibly still in flight (engine->pending).
static void actor_critic_begin(rl_mlp *policy, rl_mlp *critic, rl_adam *critic_opt, rl_traj *traj,
                               const rl_trpo_config *pcfg, const rl_values_opt_config *ccfg, rl_trpo_stats *pstats) {
  check_policy(policy, traj);
  RL_REQUIRE(pcfg && pstats, "NULL argument");
  check_values_opt(critic, critic_opt, traj, ccfg);
  RL_REQUIRE(policy != critic, "policy and critic must be different modules");
  rl_engine *e = traj->eng;
  RL_REQUIRE(!e->pending.active, "an update is pending on this engine: call rl_actor_critic_update_finish first");
  const uint64_t K = ccfg->opt_steps_per_update;
  RL_REQUIRE(K <= traj->max_losses, "too many optimisation steps per update");
  rl_engine::PendingUpdate &pu = e->pending;
  pu.traj = traj;
  pu.critic = critic;
  pu.steps = K;
  pu.stats = rl_critic_stats{};
  pu.losses.assign(K ? K : 1, 0.0f);
  if (!chains_can_overlap(policy, critic, traj, ccfg)) {
    // policy.update, then critic.update (actor_critic.rs:196-208).  A NaN policy step is fatal BEFORE the critic moves,
    // as in the reference (Trpo::update panics inside policy.update)
    trpo_update_impl(policy, traj, pcfg, pstats);
    trpo_raise_nan(pstats);
    values_opt_targets(critic, traj, ccfg);
    critic_opt_steps(critic, critic_opt, traj, K, &pu.stats, pu.losses.data());
    pu.active = pu.joined = pu.collected = true;
    return;
  }
  traj_ensure_range(traj);  // (both chains' kernels read the observation range: measured once, before the fork)
  // fork: the auxiliary stream sees everything enqueued so far (rollout, values, advantages)
  RL_HIP_CHECK(hipEventRecord(e->ev_fork, e->main_stream));
  RL_HIP_CHECK(hipStreamWaitEvent(e->aux_stream, e->ev_fork, 0));
  try {
    // Neither chain has a host round trip before the line search's first read-back, and enqueuing a launch costs the
    // host a few microseconds: the critic chain's ~160 launches in one go would keep the TRPO chain's first kernel
    // waiting for the HOST (measured: profiles/r04_overlap_trace_8192.csv, the policy chain starts when the critic
    // chain is nearly done).  So: a few critic steps to give the device something to do, the TRPO chain's head, the
    // rest of the critic chain, and only then the TRPO chain's read-backs.
    uint64_t K0 = K < 6 ? K : 6;
    // (A/B on one rank only — the ranks of a job must enqueue their collectives in one order: K = the whole critic chain
    // first, round 4's order)
    if (const char *k0 = e->n_ranks == 1 ? std::getenv("RELEARN_CRITIC_HEAD_STEPS") : nullptr) {
      const long long v = std::atoll(k0);
      K0 = v < 0 ? 0 : ((uint64_t)v > K ? K : (uint64_t)v);
    }
    static const bool marks = std::getenv("RELEARN_HOST_MARKS") != nullptr;  // debugging aid: host time of each phase
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = marks ? now() : 0.0;
    {
      AuxChain aux(traj);
      values_opt_targets(critic, traj, ccfg);
      critic_enqueue_steps(critic, critic_opt, traj, K0);
    }
    const double t1 = marks ? now() : 0.0;
    const TrpoProgress pr = trpo_update_head(policy, traj, pcfg);
    const double t2 = marks ? now() : 0.0;
    {
      AuxChain aux(traj);
      critic_enqueue_steps(critic, critic_opt, traj, K, K0);
    }
    RL_HIP_CHECK(hipEventRecord(e->ev_join, e->aux_stream));
    const double t3 = marks ? now() : 0.0;
    trpo_update_tail(policy, traj, pcfg, pr, pstats);  // (its read-backs wait for the main stream only)
    if (marks)
      std::fprintf(stderr, "relearn_hip host marks (us): critic head %.0f, trpo head %.0f, critic rest %.0f, trpo tail %.0f\n",
                   t1 - t0, t2 - t1, t3 - t2, now() - t3);
  } catch (...) {
    (void)hipStreamSynchronize(e->aux_stream);  // nothing of this update may still be running when the error returns
    (void)hipStreamSynchronize(e->main_stream);
    range_discard(traj);  // (the other chain's guard words go with the call that set them)
    throw;
  }
  pu.active = true;
  pu.joined = pu.collected = false;
  if (pstats->status == RL_OPT_NAN_LOSS || pstats->status == RL_OPT_NAN_CONSTRAINT) {
    // fatal (the reference panics); the critic chain was already in flight beside the policy chain and has advanced the
    // critic and its optimiser state — the one ordering the side-by-side form cannot keep (include/relearn_hip.h)
    (void)hipStreamSynchronize(e->aux_stream);
    pu.active = false;
    trpo_raise_nan(pstats);
  }
}

// _finish: later work on the main stream follows the critic's last step; the chain's losses come back
static void actor_critic_finish(rl_traj *traj, rl_critic_stats *cstats, float *critic_losses_out) {
  rl_engine *e = traj->eng;
  rl_engine::PendingUpdate &pu = e->pending;
  RL_REQUIRE(pu.active, "no update is pending on this engine");
  RL_REQUIRE(pu.traj == traj, "the pending update belongs to another trajectory");
  if (!pu.collected) {
    engine_settle(e);
    try {
      AuxChain aux(traj);  // (the losses were written by the auxiliary chain: read them on its stream)
      critic_collect(traj, pu.steps, &pu.stats, pu.losses.data());
    } catch (...) {
      (void)hipStreamSynchronize(e->aux_stream);
      pu.active = false;
      throw;
    }
  }
  pu.active = false;
  if (cstats) *cstats = pu.stats;
  if (critic_losses_out && pu.steps) std::memcpy(critic_losses_out, pu.losses.data(), pu.steps * sizeof(float));
}

int32_t rl_actor_critic_update_begin(rl_mlp *policy, rl_mlp *critic, rl_adam *critic_opt, rl_traj *traj,
                                     const rl_trpo_config *pcfg, const rl_values_opt_config *ccfg,
                                     rl_trpo_stats *pstats) {
  return guarded(traj ? traj->eng : nullptr,
                 [&] { actor_critic_begin(policy, critic, critic_opt, traj, pcfg, ccfg, pstats); });
}

int32_t rl_actor_critic_update_finish(rl_traj *traj, rl_critic_stats *cstats, float *critic_losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(traj, "NULL argument");
    actor_critic_finish(traj, cstats, critic_losses_out);
  }, false);
}

int32_t rl_actor_critic_update(rl_mlp *policy, rl_mlp *critic, rl_adam *critic_opt, rl_traj *traj,
                               const rl_trpo_config *pcfg, const rl_values_opt_config *ccfg, rl_trpo_stats *pstats,
                               rl_critic_stats *cstats, float *critic_losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    actor_critic_begin(policy, critic, critic_opt, traj, pcfg, ccfg, pstats);
    actor_critic_finish(traj, cstats, critic_losses_out);
  });
}

int32_t rl_critic_gradient(rl_mlp *critic, rl_traj *traj, float *grad_out, float *loss_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_critic(critic, traj);
    RL_REQUIRE(grad_out, "grad_out is NULL");
    uint32_t P = (uint32_t)critic->P;
    traj->d.tgt = traj->d.rtg;  // the documented loss: MSE against RL_TRAJ_RETURNS, whatever targets an update left behind
    run_critic_gradient(critic, traj);
    std::vector<float> h(P + 4);
    d2h(traj->eng, h.data(), traj->vec, (P + 4) * sizeof(float));
    range_check(traj, 1u << RL_GUARD_CRITIC);
    std::memcpy(grad_out, h.data(), P * sizeof(float));
    if (loss_out) *loss_out = (float)((double)h[P] / (double)b_total(traj));
  });
}

// ---------------------------------------------------------------- PPO / REINFORCE / RewardToGo
int32_t rl_ppo_config_default(rl_ppo_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    c->opt_steps_per_update = 10;  // PpoConfig::default (ppo.rs:27-41)
    c->clip_distance = 0.2;
  });
}

// PASS_PPO gradient of the clipped surrogate against lp0 -> vec[0..P), sum of min(...) -> vec[P]
static void run_policy_ppo(rl_mlp *policy, rl_traj *traj, float lo, float hi) {
  if (rl_module_is_recurrent(policy->kind)) return seq_policy_pass(policy, traj, PASS_PPO, true, lo, hi);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 &&
      launch_policy_v2(traj, policy, PASS_PPO, nullptr, b_total(traj), nullptr, lo, hi)) {
    launch_reduce(traj, P, true, true, traj->last_rows, traj->last_rows);
  } else {
    launch_policy_pass(traj, policy, PASS_PPO, nullptr, b_total(traj), nullptr, lo, hi);
    launch_mlp_backward(traj, policy, nullptr);
    launch_reduce(traj, P, true, true, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

int32_t rl_ppo_update(rl_mlp *policy, rl_adam *opt, rl_traj *traj, const rl_ppo_config *cfg,
                      rl_policy_opt_stats *stats, float *losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(opt && opt->mod == policy, "optimizer does not belong to this module");
    RL_REQUIRE(cfg, "cfg is NULL");
    RL_REQUIRE(cfg->opt_steps_per_update <= traj->max_losses, "too many optimisation steps per update");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj), K = cfg->opt_steps_per_update;
    // initial_log_probs and the logged entropy (ppo.rs:107-118): the PASS_INIT pass stores log pi_0
    if (rl_module_is_recurrent(policy->kind)) seq_policy_pass(policy, traj, PASS_INIT, false, 0.0f, 0.0f);
    else run_policy_gradient(policy, traj);
    float h0[4];
    d2h(e, h0, traj->vec + P, sizeof(h0));
    // clip(1 - d, 1 + d): f64 scalars applied to a Float tensor
    float lo = (float)(1.0 - cfg->clip_distance), hi = (float)(1.0 + cfg->clip_distance);
    for (uint64_t k = 0; k < K; ++k) {
      run_policy_ppo(policy, traj, lo, hi);
      launch_adam_step(traj, opt, (int)k, Bt);
    }
    std::vector<float> h(K ? K : 1, 0.0f);
    if (K) d2h(e, h.data(), traj->losses, K * sizeof(float));
    range_check(traj, 1u << RL_GUARD_POLICY);
    for (auto &v : h) v = -v;  // loss = -mean(min(...))
    if (losses_out && K) std::memcpy(losses_out, h.data(), K * sizeof(float));
    if (stats) {
      stats->entropy = (double)h0[1] / (double)Bt;
      stats->steps = K;
      stats->loss_first = K ? (double)h[0] : 0.0;
      stats->loss_last = K ? (double)h[K - 1] : 0.0;
    }
  });
}

int32_t rl_reinforce_update(rl_mlp *policy, rl_adam *opt, rl_traj *traj, rl_policy_opt_stats *stats) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(opt && opt->mod == policy, "optimizer does not belong to this module");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj);
    // d(-mean(log pi(a) A))/d theta equals the surrogate gradient at ratio = 1 that PASS_INIT computes
    run_policy_gradient(policy, traj);
    float h0[4];
    d2h(e, h0, traj->vec + P, sizeof(h0));
    launch_adam_step(traj, opt, -1, Bt);
    range_check(traj, 1u << RL_GUARD_POLICY);
    if (stats) {
      stats->entropy = (double)h0[1] / (double)Bt;
      stats->steps = 1;
      stats->loss_first = stats->loss_last = -((double)h0[2] / (double)Bt);
    }
  });
}

int32_t rl_reward_to_go(rl_traj *traj, float gamma) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(traj, "NULL argument");
    launch_gae(traj, nullptr, gamma, 0.0f);
  });
}

}  // extern "C"
