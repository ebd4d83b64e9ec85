// kernels.hpp — host launchers implemented in the .hip translation units.
#pragma once
#include <cstdlib>
#include <functional>
// arrays of [128 units][32 lanes] floats per (step, tile) block in the recurrent training workspace (kernels_seq.hip)
#define RL_SEQ_ACT_ARRAYS 9
#define RL_SEQ_DPRE_ARRAYS 6
#include "engine.hpp"

// kernels_rollout.hip
void launch_env_reset(rl_env *env);
void launch_env_observe(rl_env *env, float *d_obs);
void launch_env_step(rl_env *env);
void launch_rollout(rl_env *env, const rl_mlp *policy, rl_traj *traj);
void launch_values(rl_traj *traj, const rl_mlp *critic);
void launch_gae(rl_traj *traj, const rl_mlp *critic, float gamma, float lambda);  // critic NULL: adv = rtg only
// critic regression targets into traj->d.tgt: one-step TD from traj->d.values (critic != NULL) or reward-to-go (NULL)
void launch_value_targets(rl_traj *traj, const rl_mlp *critic, float gamma);
void launch_mlp_forward_host_rows(rl_mlp *mlp, const float *d_in_soa, size_t rows, float *d_out_soa);
// magnitude range of the trajectory's observation planes -> traj->d.range[0..1] (the fused kernels' range guard)
void launch_obs_range(rl_traj *traj);
// ... once per content of the planes: whoever rewrites them (rollouts, rl_traj_write) clears `range_valid`
inline void traj_ensure_range(rl_traj *traj) {
  if (traj->range_valid || traj->range_fixed) return;
  launch_obs_range(traj);
  traj->range_valid = true;
}

// The weight image of a 5 -> 128 -> A module (bf16_tile.hpp): valid for one C-ABI call at a time.
//   wimg_ensure       a fused launcher about to read it: builds it (one small launch on the engine's current stream)
//                     unless this call already did, returns it
//   wimg_if_current   a launcher whose kernel writes the module's parameters one lane per parameter: the image to keep
//                     current beside them, or NULL when this call has not built one (nothing to keep)
//   wimg_invalidate   anything else that writes the parameters on the device inside a call
const uint32_t *wimg_ensure(const rl_mlp *m);
inline uint32_t *wimg_if_current(const rl_mlp *m) {
  return m->d_wimg != nullptr && m->wimg_epoch == m->eng->call_epoch ? m->d_wimg : nullptr;
}
inline void wimg_invalidate(const rl_mlp *m) { m->wimg_epoch = 0; }

// kernels_update.hip
enum PolicyPassMode { PASS_INIT = 0, PASS_EVAL = 1, PASS_JVP = 2, PASS_DQN = 3, PASS_PPO = 4 };
// PASS_INIT : lp0 <- log pi(.|s); dz <- d(-mean(ratio*A))/dz at theta0; slabB <- {sum A, sum entropy}
// PASS_EVAL : slabB <- {sum ratio*A, sum KL(pi0||pi)}                      (skipped when *skip_flag != 0)
// PASS_JVP  : dz <- (diag(p) - p p^T) J v / B_total                         (skipped when *skip_flag != 0)
// PASS_PPO  : dz <- d(-mean(min(ratio A, clip(ratio, lo, hi) A)))/dz against lp0; slabB <- {sum min(..)}  (ppo.rs:124-137)
// PASS_DQN  : outputs are action values; dz[a] <- [a == action] 2 (Q_a - target) / B_total with the target in
//             `adv`; slabB <- {sum (Q_a - target)^2}                        (dqn.rs:316-326)
void launch_policy_pass(rl_traj *traj, const rl_mlp *policy, int mode, const float *d_tangent, uint64_t B_total,
                        const int32_t *d_skip_flag, float clip_lo = 0.0f, float clip_hi = 0.0f);
void launch_critic_fwd(rl_traj *traj, const rl_mlp *critic, uint64_t B_total);
// J^T dz accumulated per block into slabA (lane = hidden unit, samples broadcast through scalar loads)
void launch_mlp_backward(rl_traj *traj, const rl_mlp *mlp, const int32_t *d_skip_flag);
// vec[0..P) <- sum_blocks slabA (if useA), vec[P..P+4) <- sum_blocks slabB; deterministic order
void launch_reduce(rl_traj *traj, uint32_t P, bool useA, bool useB, uint32_t rowsA, uint32_t rowsB);

void launch_trpo_begin(rl_traj *traj, rl_mlp *policy, uint64_t B_total);                 // after grad reduce
void launch_cg_step(rl_traj *traj, uint32_t P, float reg, float tol);                    // after HVP reduce
void launch_cg_finish(rl_traj *traj, uint32_t P);                                        // nan_to_num, tangent <- x
void launch_step_size(rl_traj *traj, rl_mlp *policy, float reg, double max_kl);          // after HVP(x) reduce
void launch_ls_set_params(rl_traj *traj, rl_mlp *policy, double ratio);
void launch_ls_check(rl_traj *traj, uint32_t P, uint64_t B_total, int index, double ratio, double max_kl);
void launch_ls_finalize(rl_traj *traj, rl_mlp *policy, double max_kl, int accept_violation);
void launch_adam_step(rl_traj *traj, rl_adam *opt, int loss_slot, uint64_t B_total);     // after critic reduce
void launch_adam_step_vec(rl_adam *opt, const float *d_grad);
// reduce(A + B) and Adam in one launch (single-rank runs: no all-reduce between the two)
void launch_reduce_adam(rl_traj *traj, rl_adam *opt, uint32_t rowsA, uint32_t rowsB, int loss_slot, uint64_t B_total);

// kernels_mfma.hip ("v2": f32-MFMA layer 1, lane = hidden unit backward; H = 128, D = 5 only)
// returns false when the shape is not supported (caller falls back to the v1 kernels)
bool launch_critic_step_v2(rl_traj *traj, const rl_mlp *critic, uint64_t B_total);
bool launch_policy_v2(rl_traj *traj, const rl_mlp *policy, int mode, const float *d_tangent, uint64_t B_total,
                      const int32_t *d_skip_flag, float clip_lo = 0.0f, float clip_hi = 0.0f);

// kernels_dqn.hip
// d_range (may be NULL): the collected observations' magnitude range is folded there (the fused gradient's range guard)
void launch_rollout_dqn(rl_env *env, const rl_mlp *qnet, const ReplayDev &rp, uint32_t T, uint64_t p_int,
                        int always_explore, uint8_t *d_flags, uint32_t *d_range = nullptr);
// an action-value module of any shape (rl_mlp::general): one launch sequence per step; `ws` lends the layer kernels their
// workspace, d_q [2][n] receives the module's outputs of every step
void launch_rollout_dqn_general(rl_env *env, const rl_mlp *qnet, rl_traj *ws, float *d_q, const ReplayDev &rp, uint32_t T,
                                uint64_t p_int, int always_explore, uint8_t *d_flags);
void launch_dqn_td_targets(rl_engine *eng, float *d_target, const uint8_t *d_flag, const float *d_q_next, uint32_t n,
                           float gamma);
struct AgentKey { uint32_t w[8]; };
void launch_dqn_sample(rl_engine *eng, hipStream_t stream, const ReplayDev &rp, const AgentKey &key, uint64_t *d_agent_pos,
                       uint32_t minibatch_steps, uint32_t max_eps, uint32_t *d_lane, uint32_t *d_start,
                       uint32_t *d_len, uint32_t *d_off, DqnCountsDev *d_counts, int sequential,
                       uint32_t n_batches = 1);
void launch_dqn_build_all(rl_engine *eng, const ReplayDev &rp, uint32_t n_batches, uint32_t widest_eps, uint32_t max_eps,
                          const uint32_t *d_lane, const uint32_t *d_start, const uint32_t *d_len, const uint32_t *d_off,
                          const DqnCountsDev *d_counts, float *d_obs, size_t obs_stride, uint8_t *d_action,
                          float *d_target, size_t step_stride, float gamma, uint8_t *d_flag);
void launch_replay_planes(rl_engine *eng, const ReplayDev &rp, int field, void *d_out);
bool launch_dqn_step_bf16(rl_traj *mb, const rl_mlp *qnet, uint64_t B_total, bool td_in_kernel, float gamma);
// kernels_critic.hip: the same gradient (targets given, not in-kernel TD) as two critic-step channels per SIMD
bool launch_dqn_step_pair(rl_traj *mb, const rl_mlp *qnet, uint64_t B_total);
void launch_dqn_build_minibatch(rl_engine *eng, const ReplayDev &rp, uint32_t n_eps, const uint32_t *d_lane,
                                const uint32_t *d_start, const uint32_t *d_len, const uint32_t *d_off,
                                float *d_obs, size_t out_plane, uint8_t *d_action, float *d_target, float gamma,
                                int one_step_td, const rl_mlp *qnet);

// kernels_seq.hip (recurrent configuration: Chain lanes, GRU -> ReLU -> MLP module; tiles of 32 lanes)
void launch_chain_reset(rl_env *env);
void launch_chain_observe(rl_env *env, float *d_obs);
void launch_chain_step(rl_env *env);
void launch_rollout_gru(rl_env *env, const rl_mlp *policy, rl_traj *traj);        // recurrent policy, either env kind
void launch_rollout_chain_mlp(rl_env *env, const rl_mlp *policy, rl_traj *traj);  // feed-forward policy on Chain lanes
// teacher-forced forward: d_out [A][T][n]; d_succ (may be NULL) [A][T][n]; d_act (may be NULL) activation record
void launch_gru_seq_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_succ, float *d_act,
                            const int32_t *d_skip = nullptr);
// kernels_general.hip: MLPs of any hidden_sizes (per-layer kernels; rl_mlp::general)
void gen_ensure(rl_traj *t, const rl_mlp *m, uint64_t rows, bool tangent, bool backward);
void gen_free(rl_traj *t);
void launch_gen_forward(rl_traj *t, const rl_mlp *m, const float *x, size_t x_stride, uint64_t rows, float *out);
void launch_gen_policy_pass(rl_traj *t, const rl_mlp *m, int mode, const float *d_tangent, uint64_t B_total,
                            const int32_t *d_skip, float clip_lo, float clip_hi);
void launch_gen_critic_fwd(rl_traj *t, const rl_mlp *m, uint64_t B_total);
void launch_gen_backward(rl_traj *t, const rl_mlp *m, const int32_t *d_skip);
void traj_ensure_slabs(rl_traj *t, uint64_t rowsA, uint64_t P, uint64_t rowsB);
// kernels_gen_mfma.hip: the passes of a several-hidden-layer MLP as one fused matrix-pipe launch (false: shape not built)
constexpr int RL_GEN_CRITIC = 100;  // mode: mean((V - target)^2); else PASS_INIT / PASS_PPO / PASS_EVAL / PASS_JVP
bool gen_mfma_fits(const rl_traj *t, const rl_mlp *m);
bool launch_gen_mfma(rl_traj *t, const rl_mlp *m, int mode, const float *d_tangent, uint64_t B_total,
                     const int32_t *d_skip, float clip_lo, float clip_hi);
void launch_gen_values(rl_traj *t, const rl_mlp *critic);  // -> seq.out / seq.succ (plane 0)
void launch_gen_rollout(rl_env *env, const rl_mlp *policy, rl_traj *t);
void launch_rollout_stepwise(rl_env *env, rl_traj *t, const float *z, const std::function<void(uint32_t)> &forward);
void launch_gen_wgrad_planes(rl_traj *t, const float *dY, size_t dys, int N, const float *X, size_t xs, int K, size_t S,
                             uint32_t chunk, uint32_t rows, uint32_t P, uint32_t offW, uint32_t offB,
                             const int32_t *d_skip);
// kernels_seq_stack.hip: recurrent chains with RnnBaseConfig::num_layers > 1 (one thread per lane, the unit loop of a
// layer dealt to a workgroup's waves; no lane-tile or width restriction)
void stack_ensure(rl_traj *t, const rl_mlp *mod, bool training);
void stack_free(rl_traj *t);
void launch_stack_rollout(rl_env *env, const rl_mlp *policy, rl_traj *traj);
void launch_stack_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_succ, bool record,
                          const int32_t *d_skip);
void launch_stack_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip);  // dz, record -> vec[0..P)
void launch_stack_tangent(rl_traj *traj, const rl_mlp *mod, const float *d_tangent, const int32_t *d_skip);  // -> seq.out
// a recurrent chain of other widths embedded in the kernels' shape (rl_mlp::exec): copy the flat vector `real_src`
// (layout of `real`) into the padded layout `exec_dst` / gather the padded vector `exec_src` into the flat layout
void launch_seq_pad(const rl_mlp *real, float *exec_dst, const float *real_src);
void launch_seq_unpad(const rl_mlp *real, const float *exec_src, float *real_dst);
// kernels_seq_train.hip: the GRU chain's training passes with the recurrence on the bf16 matrix pipe
void launch_gru_train_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip);
void launch_lstm_train_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip);
void launch_seq_train_head_forward(rl_traj *traj, const rl_mlp *mod, float *d_out, float *d_act, const int32_t *d_skip);
void launch_gru_train_head_backward(rl_traj *traj, const rl_mlp *mod, float *d_slab, const int32_t *d_skip);
void launch_gru_train_recur_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip);
void launch_gru_train_wgrad(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip);
constexpr uint32_t RL_SEQ_HEAD_ROWS = 512;  // rows of head-gradient partials behind the weight-gradient kernel's chunks
void launch_seq_gae(rl_traj *traj, float gamma, float lambda);
void launch_seq_value_targets(rl_traj *traj, float gamma);  // one-step TD targets from traj->seq.out / succ -> d.tgt  // reads traj->seq.out / succ (plane 0)
void launch_seq_policy_dlogits(rl_traj *traj, int mode, uint64_t B_total, float clip_lo, float clip_hi,
                               const int32_t *d_skip = nullptr);
void launch_seq_critic_dvalues(rl_traj *traj, uint64_t B_total);
void launch_gru_backward(rl_traj *traj, const rl_mlp *mod, const int32_t *d_skip = nullptr);  // dz, seq.act -> vec[0..P)
// Fisher-vector product pieces: static tangent terms (dpre, seq.succ) from the tangent parameters, then the recurrent
// tangent pass -> seq.out = d logits . tangent, then dz <- (diag(p) - p p^T) (.) / B
void launch_gru_tangent(rl_traj *traj, const rl_mlp *mod, const float *d_tangent, uint64_t B_total,
                        const int32_t *d_skip);
