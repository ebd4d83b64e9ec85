/* oracle.h — CPU restatement of the reference (edlanglois/relearn v0.3.1) hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under relearn_amd/ links, loads or calls this library.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and there
 * only as the checker / the reported CPU baseline — never as the thing measured or shipped.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - ChaCha block function: pinned by RFC 7539 (20 rounds) and eSTREAM ChaCha8 zero-key vectors.
 *   - packed / cumsum / features / buffers / take-aligned / step-limit / categorical / CG:
 *     pinned by the reference's own unit-test fixtures (tests/golden/reference_fixtures.json).
 *   - MLP backward, TRPO step, Adam, GAE numerics: pinned by vectors generated with PyTorch CPU
 *     autograd in the build container (tests/golden/make_torch_golden.py).
 *   - rand 0.8.5 sampling conventions (seed_from_u64, Uniform, gen_range, gen_bool) and
 *     CartPole numerics have no concrete values in the reference's tests: PARITY UNPINNED for
 *     "same seed as the Rust binary"; they are restated from the pinned crate versions' published
 *     algorithms.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- Prng (src/lib.rs:68) */
typedef struct {
  uint32_t key[8];
  uint64_t counter; /* block counter of the NEXT buffer refill */
  uint64_t stream;
  uint32_t buf[64]; /* rand_chacha generates 4 blocks per refill */
  uint32_t index;   /* next unread word; 64 = empty */
} oracle_prng;

void oracle_prng_seed_from_u64(oracle_prng *r, uint64_t seed);
void oracle_prng_from_seed(oracle_prng *r, const uint32_t key[8]);
void oracle_prng_from_rng(oracle_prng *out, oracle_prng *src);
void oracle_prng_set_stream(oracle_prng *r, uint64_t stream);
void oracle_prng_set_word_pos(oracle_prng *r, uint64_t word_pos);
uint64_t oracle_prng_word_pos(const oracle_prng *r);
uint32_t oracle_prng_next_u32(oracle_prng *r);
uint64_t oracle_prng_next_u64(oracle_prng *r);
float oracle_prng_gen_f32(oracle_prng *r);
double oracle_prng_gen_f64(oracle_prng *r);
uint64_t oracle_prng_gen_range_u64(oracle_prng *r, uint64_t low, uint64_t high);
int oracle_prng_gen_bool(oracle_prng *r, double p);
double oracle_prng_uniform_f64_inclusive(oracle_prng *r, double low, double high);

/* ---------------------------------------------------------------- Successor (src/envs/mod.rs:257) */
enum { ORACLE_CONTINUE = 0, ORACLE_TERMINATE = 1, ORACLE_INTERRUPT = 2 };

/* ---------------------------------------------------------------- CartPole (src/envs/cartpole.rs) */
typedef struct {
  double gravity, mass_cart, mass_pole, length_half_pole, friction_cart, friction_pole, time_step;
  double action_force, max_pos, max_angle, discount_factor;
  /* InternalPhysicalConstants (cartpole.rs:238-251) */
  double total_weight, inv_total_mass, mass_length_pole;
  int use_libm; /* 0: rl_detmath sincos (engine contract); 1: platform libm like the reference */
} oracle_cartpole;

typedef struct {
  double x, xdot, th, thdot;
  int32_t nv_pos; /* cached_normal_velocity_is_positive */
} oracle_cartpole_state;

void oracle_cartpole_default(oracle_cartpole *env);
void oracle_cartpole_finish(oracle_cartpole *env); /* recompute derived constants */
void oracle_cartpole_initial_state(const oracle_cartpole *env, oracle_prng *rng, oracle_cartpole_state *s);
void oracle_cartpole_next_state(const oracle_cartpole *env, const oracle_cartpole_state *s, double force,
                                oracle_cartpole_state *out);
/* returns successor code; *s is overwritten with the next state unless Terminate */
int oracle_cartpole_step(const oracle_cartpole *env, oracle_cartpole_state *s, int action, double *reward);

/* ---------------------------------------------------------------- Chain (src/envs/chain.rs) */
typedef struct { uint64_t size; double discount_factor; } oracle_chain;
void oracle_chain_default(oracle_chain *env);
int oracle_chain_step(const oracle_chain *env, uint64_t *state, int action, oracle_prng *rng, double *reward);
int oracle_chain_step_draw(const oracle_chain *env, uint64_t *state, int action, float draw, double *reward);

/* ---------------------------------------------------------------- MemoryGame (src/envs/memory.rs) */
typedef struct { uint64_t num_actions, history_len; double discount_factor; } oracle_memory;
void oracle_memory_default(oracle_memory *env);
/* state = (current_state, initial_state) */
void oracle_memory_initial_state(const oracle_memory *env, oracle_prng *rng, uint64_t *current, uint64_t *initial);
int oracle_memory_step(const oracle_memory *env, uint64_t *current, uint64_t initial, uint64_t action,
                       double *reward);

/* ---------------------------------------------------------------- step limits (src/envs/wrappers/step_limit.rs) */
enum { ORACLE_LIMIT_NONE = 0, ORACLE_LIMIT_LATENT = 1, ORACLE_LIMIT_VISIBLE = 2 };
/* apply the wrapper's post-step rule: decrement, Continue -> Interrupt at 0 (step_limit.rs:216-222) */
int oracle_step_limit_apply(int inner_successor, uint64_t *steps_remaining);
double oracle_step_limit_remaining(uint64_t steps_remaining, uint64_t max_steps);

/* observation features, f32 (spaces/interval.rs:108-116, index.rs:104-116, derive space.rs:504-513) */
void oracle_cartpole_features(const oracle_cartpole_state *s, int limit_kind, uint64_t steps_remaining,
                              uint64_t max_steps, float *out);
void oracle_index_features(uint64_t index, uint64_t size, float *out);

/* ---------------------------------------------------------------- HistoryDataBound / TakeAlignedSteps */
typedef struct { uint64_t min_steps, slack_steps; } oracle_bound;
oracle_bound oracle_bound_divide(oracle_bound b, uint64_t n);
oracle_bound oracle_bound_max(oracle_bound a, oracle_bound b);
oracle_bound oracle_bound_with_default_slack(uint64_t min_steps);
/* how many steps TakeAlignedSteps yields given the episode_done flag of each offered step */
uint64_t oracle_take_aligned_count(const uint8_t *episode_done, uint64_t n_available, uint64_t min_steps,
                                   uint64_t slack_steps);

/* ---------------------------------------------------------------- history buffers (generic steps) */
/* A step with an observation vector of obs_dim floats (features are what the torch path needs;
 * index envs use obs_dim = 1 with the index stored as a float for buffer tests). */
typedef struct {
  uint32_t obs_dim;
  uint64_t len, cap;
  float *obs;       /* [len][obs_dim] */
  float *next_obs;  /* [len][obs_dim], meaningful where next == INTERRUPT */
  int32_t *action;  /* [len] */
  double *reward;   /* [len] */
  uint8_t *next;    /* [len] successor code */
  uint64_t n_episode_ends, cap_episode_ends;
  uint64_t *episode_ends; /* one past the end index of each episode */
} oracle_vecbuffer;

oracle_vecbuffer *oracle_vecbuffer_new(uint32_t obs_dim);
void oracle_vecbuffer_free(oracle_vecbuffer *b);
void oracle_vecbuffer_clear(oracle_vecbuffer *b);
void oracle_vecbuffer_write_step(oracle_vecbuffer *b, const float *obs, int32_t action, double reward, int next,
                                 const float *next_obs);
void oracle_vecbuffer_end_experience(oracle_vecbuffer *b);
uint64_t oracle_vecbuffer_num_steps(const oracle_vecbuffer *b);
uint64_t oracle_vecbuffer_num_episodes(const oracle_vecbuffer *b);
/* copy-out helpers for tests */
void oracle_vecbuffer_episode_ends(const oracle_vecbuffer *b, uint64_t *out);
void oracle_vecbuffer_steps(const oracle_vecbuffer *b, float *obs, int32_t *action, double *reward, uint8_t *next,
                            float *next_obs);

/* replay buffer eviction rule (src/agents/buffers/replay.rs:89-115) on episode lengths only */
typedef struct oracle_replay oracle_replay;
oracle_replay *oracle_replay_new(uint64_t capacity);
void oracle_replay_free(oracle_replay *r);
int oracle_replay_write_step(oracle_replay *r, int32_t tag, int episode_done);
void oracle_replay_end_experience(oracle_replay *r);
uint64_t oracle_replay_num_steps(const oracle_replay *r);
uint64_t oracle_replay_num_episodes(const oracle_replay *r);
uint64_t oracle_replay_total_step_count(const oracle_replay *r);
void oracle_replay_dump(const oracle_replay *r, int32_t *tags, uint64_t *episode_lens);

/* ---------------------------------------------------------------- packed sequences (src/torch/packed.rs) */
/* batch sizes from monotonically non-increasing sequence lengths; returns count or -1 on error */
int64_t oracle_packed_batch_sizes(const uint64_t *sorted_lengths, uint64_t n_seq, uint64_t *batch_sizes_out,
                                  uint64_t cap);
/* packed index -> (sequence, offset) pairs in packing order */
void oracle_packed_order(const uint64_t *sorted_lengths, uint64_t n_seq, uint64_t *seq_out, uint64_t *off_out);
void oracle_discounted_cumsum_from_end_f32(float *data, uint64_t n, float discount, const uint64_t *batch_sizes,
                                           uint64_t n_batches);
/* trim helpers on batch-size vectors (packed.rs:195-267 Ragged semantics) */
uint64_t oracle_packed_trim_batch_sizes(const uint64_t *batch_sizes, uint64_t n, uint64_t trim, uint64_t *out);
void oracle_packed_trim_end_f32(const float *in, const uint64_t *batch_sizes, uint64_t n_batches, uint64_t trim,
                                float *out);

/* ---------------------------------------------------------------- history features (src/torch/agents/features.rs) */
typedef struct {
  uint32_t obs_dim;
  uint64_t n_steps, n_episodes, n_ext;
  uint64_t n_batches, n_ext_batches;
  uint64_t *batch_sizes;     /* [n_batches] structure of obs/actions/rewards */
  uint64_t *ext_batch_sizes; /* [n_ext_batches] */
  float *obs;                /* [n_steps][obs_dim] packed */
  float *ext_obs;            /* [n_ext][obs_dim] packed */
  uint8_t *is_invalid;       /* [n_ext] */
  int64_t *actions;          /* [n_steps] */
  float *rewards;            /* [n_steps] */
  uint64_t *src_index;       /* [n_steps] packed position -> index of the step in the source buffer order */
} oracle_features;

oracle_features *oracle_features_from_buffers(oracle_vecbuffer *const *buffers, uint64_t n_buffers);
void oracle_features_free(oracle_features *f);

/* ---------------------------------------------------------------- MLP (src/torch/modules/ff/{linear,mlp}.rs) */
typedef struct { uint32_t in_dim, hidden, out_dim; } oracle_mlp_shape;
uint64_t oracle_mlp_num_params(oracle_mlp_shape s);
/* Glorot-uniform with fan_in = in+1 for kernel and bias (ff/linear.rs:54-68, initializers.rs:78-108);
 * draws from the engine's documented init stream (ChaCha8 seed, stream 0, gen f32 per element). */
void oracle_mlp_init(oracle_mlp_shape s, uint64_t seed, float *params);
void oracle_mlp_forward_f32(oracle_mlp_shape s, const float *params, const float *x, float *out);
void oracle_mlp_forward_batch_f32(oracle_mlp_shape s, const float *params, const float *x, uint64_t n, float *out);
/* Linear::new with LinearConfig { kernel_init, bias_init } for every layer (ff/linear.rs:54-68, initializers.rs): kinds 0
 * Zeros, 1 Constant(value), 2 Uniform(scale), 3 Normal(scale), 4 Orthogonal; scales 0 Constant(value), 1 FanIn, 2 FanOut,
 * 3 FanAvg; draws from the engine's init stream */
void oracle_mlp_layers_init(uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden, uint32_t out_dim,
                            uint64_t seed, int k_kind, int k_scale, double k_value, int b_kind, int b_scale,
                            double b_value, float *params);
/* Mlp::forward for any hidden_sizes (<= 256 wide) and activations (ff/mlp.rs:139-151, ff/activation.rs:85-92); act codes
 * 0 Identity, 1 Relu, 2 Sigmoid, 3 Tanh; x [rows][in_dim] -> out [rows][out_dim]; the engine's fma order and detmath */
void oracle_mlp_layers_forward_f32(uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden, uint32_t out_dim,
                                   int act, int out_act, const float *params, const float *x, uint64_t rows,
                                   float *out);

/* ---------------------------------------------------------------- Categorical (src/torch/distributions/categorical.rs) */
void oracle_log_softmax_f32(const float *z, uint32_t n, float *lp, int use_libm);
int oracle_categorical_sample_u(const float *lp, uint32_t n, float u, int use_libm);
float oracle_categorical_entropy_f32(const float *lp, uint32_t n, int use_libm);
float oracle_categorical_kl_f32(const float *lp_self, const float *lp_other, uint32_t n, int use_libm);

/* ---------------------------------------------------------------- critic / GAE (src/torch/agents/critics/mod.rs) */
/* reference-structured: packed features -> packed advantages / reward-to-go */
void oracle_gae_packed(oracle_mlp_shape cs, const float *critic_params, const oracle_features *f, float gamma,
                       float lambda, float *adv_out, float *ext_values_out);
void oracle_reward_to_go_packed(const oracle_features *f, float gamma, float *out);
/* StepValueTarget::OneStepTd (critics/mod.rs:139-150, 219-229) */
void oracle_one_step_values_packed(oracle_mlp_shape cs, const float *critic_params, const oracle_features *f,
                                   float gamma, float *out);

/* ---------------------------------------------------------------- update math on flat sample arrays */
typedef struct {
  uint64_t iterations, max_backtracks;
  double backtrack_ratio, hpv_reg_coeff, max_kl;
  int accept_violation;
} oracle_trpo_cfg;
void oracle_trpo_cfg_default(oracle_trpo_cfg *c);

enum { ORACLE_OPT_OK = 0, ORACLE_OPT_LOSS_NOT_IMPROVING = 1, ORACLE_OPT_CONSTRAINT_VIOLATED = 2,
       ORACLE_OPT_NAN_LOSS = 3, ORACLE_OPT_NAN_CONSTRAINT = 4 };

typedef struct {
  double entropy, step_size, loss_initial, loss_final, constraint_val_final, step_scale;
  int64_t num_backtracks; /* -1 when the line search exhausted its budget */
  int32_t status;
  int32_t cg_iterations;
} oracle_trpo_stats;

/* f32 arithmetic ("as reference", Kind::Float) and f64 arithmetic (ground truth for tolerances) */
void oracle_policy_grad_f32(oracle_mlp_shape s, const float *params, const float *obs, const int64_t *actions,
                            const float *adv, uint64_t n, float *grad_out, float *loss_out);
void oracle_policy_fvp_f32(oracle_mlp_shape s, const float *params, const float *obs, uint64_t n, const float *v,
                           float reg, float *out);
void oracle_policy_loss_kl_f32(oracle_mlp_shape s, const float *params, const float *params0, const float *obs,
                               const int64_t *actions, const float *adv, uint64_t n, float *loss, float *kl);
void oracle_trpo_update_f32(oracle_mlp_shape s, float *params, const float *obs, const int64_t *actions,
                            const float *adv, uint64_t n, const oracle_trpo_cfg *cfg, oracle_trpo_stats *stats,
                            float *step_dir_out);
void oracle_policy_grad_f64(oracle_mlp_shape s, const double *params, const double *obs, const int64_t *actions,
                            const double *adv, uint64_t n, double *grad_out, double *loss_out);
void oracle_policy_fvp_f64(oracle_mlp_shape s, const double *params, const double *obs, uint64_t n, const double *v,
                           double reg, double *out);
void oracle_policy_loss_kl_f64(oracle_mlp_shape s, const double *params, const double *params0, const double *obs,
                               const int64_t *actions, const double *adv, uint64_t n, double *loss, double *kl);
void oracle_trpo_update_f64(oracle_mlp_shape s, double *params, const double *obs, const int64_t *actions,
                            const double *adv, uint64_t n, const oracle_trpo_cfg *cfg, oracle_trpo_stats *stats,
                            double *step_dir_out);
void oracle_cg_dense_f64(const double *A, const double *b, uint32_t n, uint64_t iterations, double tol, double *x);

/* solve_conjugate_gradient on an explicit dense matrix (conjugate_gradient.rs:371-403), f32 */
void oracle_cg_dense_f32(const float *A, const float *b, uint32_t n, uint64_t iterations, double tol, float *x);

typedef struct { double lr, beta1, beta2, eps, weight_decay; } oracle_adam_cfg;
void oracle_adam_cfg_default(oracle_adam_cfg *c);
typedef struct { uint64_t step; float *m, *v; uint64_t n; } oracle_adam_state;
oracle_adam_state *oracle_adam_new(uint64_t n);
void oracle_adam_free(oracle_adam_state *st);
void oracle_adam_step_f32(oracle_adam_state *st, const oracle_adam_cfg *cfg, float *params, const float *grad);

/* f64 ground truth over millions of samples (OpenMP over chunks): kind 0 policy gradient, 1 Fisher-vector product
 * (no regulariser), 2 critic MSE gradient; inputs in the device's f32 storage, obs [n][D], results in f64 */
void oracle_grad_f64_mt(int kind, oracle_mlp_shape s, const float *params, const float *obs, const uint8_t *actions,
                        const float *aux, const float *v, uint64_t n, double *grad_out, double *loss_out);
void oracle_grad_f32_mt(int kind, oracle_mlp_shape s, const float *params, const float *obs, const uint8_t *actions,
                        const float *aux, const float *v, uint64_t n, double *grad_out, double *loss_out);
void oracle_critic_grad_f32(oracle_mlp_shape s, const float *params, const float *obs, const float *targets,
                            uint64_t n, float *grad_out, float *loss_out);
/* n_steps x {mse; backward; adam}; losses_out[n_steps] = loss BEFORE each step (opt.rs:100-126) */
void oracle_critic_update_f32(oracle_mlp_shape s, float *params, oracle_adam_state *st, const oracle_adam_cfg *cfg,
                              const float *obs, const float *targets, uint64_t n, uint64_t n_steps,
                              float *losses_out);

/* PPO (policies/ppo.rs:97-146), REINFORCE (policies/reinforce.rs:64-88) */
void oracle_policy_logp_f32(oracle_mlp_shape s, const float *params, const float *obs, const int64_t *actions,
                            uint64_t n, float *logp_out, float *entropy_out);
void oracle_policy_logp_f64(oracle_mlp_shape s, const double *params, const double *obs, const int64_t *actions,
                            uint64_t n, double *logp_out, double *entropy_out);
void oracle_ppo_grad_f32(oracle_mlp_shape s, const float *params, const float *obs, const int64_t *actions,
                         const float *adv, const float *logp0, uint64_t n, float clip_lo, float clip_hi,
                         float *grad_out, float *loss_out);
void oracle_ppo_grad_f64(oracle_mlp_shape s, const double *params, const double *obs, const int64_t *actions,
                         const double *adv, const double *logp0, uint64_t n, double clip_lo, double clip_hi,
                         double *grad_out, double *loss_out);
float oracle_reinforce_loss_f32(oracle_mlp_shape s, const float *params, const float *obs, const int64_t *actions,
                                const float *adv, uint64_t n);
double oracle_reinforce_loss_f64(oracle_mlp_shape s, const double *params, const double *obs, const int64_t *actions,
                                 const double *adv, uint64_t n);
void oracle_ppo_update_f32(oracle_mlp_shape s, float *params, oracle_adam_state *st, const oracle_adam_cfg *cfg,
                           const float *obs, const int64_t *actions, const float *adv, uint64_t n, uint64_t n_steps,
                           double clip_distance, float *losses_out, float *entropy_out);
void oracle_reinforce_update_f32(oracle_mlp_shape s, float *params, oracle_adam_state *st, const oracle_adam_cfg *cfg,
                                 const float *obs, const int64_t *actions, const float *adv, uint64_t n,
                                 float *loss_out, float *entropy_out);

/* ---------------------------------------------------------------- tabular Q (src/agents/tabular.rs) */
typedef struct {
  uint64_t n_obs, n_act;
  double discount_factor, exploration_rate;
  uint64_t *counts; /* [n_obs][n_act] */
  double *values;   /* [n_obs][n_act] */
} oracle_tabular_q;
oracle_tabular_q *oracle_tabular_q_new(uint64_t n_obs, uint64_t n_act, double gamma, double eps);
void oracle_tabular_q_free(oracle_tabular_q *q);
int oracle_tabular_q_act(const oracle_tabular_q *q, uint64_t obs, int training, oracle_prng *rng);
void oracle_tabular_q_step_update(oracle_tabular_q *q, uint64_t obs, uint64_t action, double reward, int next,
                                  uint64_t next_obs);
void oracle_tabular_q_read(const oracle_tabular_q *q, double *values_out, uint64_t *counts_out);
/* examples/chain-tabular-q.rs with `n_threads` workers run sequentially (results are independent of
 * thread interleaving because workers only share the immutable actor snapshot). */
void oracle_chain_tabular_q_train(uint64_t seed, uint64_t n_threads, uint64_t n_periods, uint64_t min_worker_steps,
                                  double *q_values_out, uint64_t *counts_out, uint64_t *total_steps_out);
/* evaluation run: SimSeed::Root(seed), greedy actor, n steps; returns sum of rewards */
double oracle_chain_tabular_q_eval(const double *q_values, uint64_t seed, uint64_t n_steps, int32_t *actions_out);

/* ---------------------------------------------------------------- lane engine restatement */
/* The engine's vectorised semantics expressed with the scalar pieces above: lane `g` (global id)
 * runs Steps::step (simulation/steps.rs:113-167) against its own env/actor streams. */
typedef struct {
  oracle_cartpole env;
  int limit_kind;
  uint64_t max_steps;
  uint64_t seed_env, seed_actor;
  uint64_t n_lanes, lane_offset;
  /* per-lane persistent state */
  oracle_cartpole_state *state;
  uint64_t *steps_remaining;
  uint64_t *reset_count;
  uint64_t t_global; /* steps taken so far (same for every lane) */
} oracle_lanes;

oracle_lanes *oracle_lanes_new(const oracle_cartpole *env, int limit_kind, uint64_t max_steps, uint64_t n_lanes,
                               uint64_t lane_offset, uint64_t seed_env, uint64_t seed_actor);
void oracle_lanes_free(oracle_lanes *l);
void oracle_lanes_reset(oracle_lanes *l);
void oracle_lanes_get_state(const oracle_lanes *l, double *state4, int32_t *nv_pos, uint64_t *steps_remaining,
                            uint64_t *reset_count);
void oracle_lanes_set_state(oracle_lanes *l, const double *state4, const int32_t *nv_pos,
                            const uint64_t *steps_remaining, const uint64_t *reset_count);
/* one vectorised env step with given actions; outputs [n_lanes] (obs_next: [D][n_lanes] SoA) */
void oracle_lanes_step(oracle_lanes *l, const uint8_t *actions, float *reward, uint8_t *flag, float *obs_next_soa,
                       float *term_obs_soa);
void oracle_lanes_observe(const oracle_lanes *l, float *obs_soa);
/* T-step rollout with the MLP policy; time-major SoA outputs as the engine lays them out:
 * obs [D][T+1][n], action [T][n], reward [T][n], flag [T][n], term_obs [D][T][n] (written at Interrupt) */
void oracle_lanes_rollout(oracle_lanes *l, oracle_mlp_shape ps, const float *policy_params, uint64_t T, float *obs,
                          uint8_t *action, float *reward, uint8_t *flag, float *term_obs, int n_threads);
/* lane-major GAE / reward-to-go exactly as the engine defines it (horizon cut = Interrupt(obs[T])) */
void oracle_lanes_gae(oracle_mlp_shape cs, const float *critic_params, uint64_t n, uint64_t T, uint32_t D,
                      const float *obs, const float *reward, const uint8_t *flag, const float *term_obs, float gamma,
                      float lambda, float *values_out, float *adv_out, float *rtg_out);
void oracle_lanes_one_step_targets(oracle_mlp_shape cs, const float *critic_params, uint64_t n, uint64_t T, uint32_t D,
                                   const float *obs, const float *reward, const uint8_t *flag, const float *term_obs,
                                   float gamma, float *out);
/* convert a lane trajectory into reference-style episodes in a VecBuffer (one buffer, lanes in order);
 * horizon cut handled with keep_last (engine rule) or the reference's drop rule (buffers/mod.rs:237-261) */
oracle_vecbuffer *oracle_lanes_to_vecbuffer(uint64_t n, uint64_t T, uint32_t D, const float *obs,
                                            const uint8_t *action, const float *reward, const uint8_t *flag,
                                            const float *term_obs, int keep_last, uint64_t *lane_t_index_out);

/* ---------------------------------------------------------------- DQN (src/torch/agents/dqn.rs), dqn.c */
typedef struct oracle_dqn_store oracle_dqn_store; /* one ReplayBuffer (replay.rs:11-27) per lane */
oracle_dqn_store *oracle_dqn_store_new(uint64_t n_lanes, uint64_t capacity, uint32_t obs_dim);
void oracle_dqn_store_free(oracle_dqn_store *s);
uint64_t oracle_dqn_store_actor_pos(const oracle_dqn_store *s, uint64_t lane);
void oracle_dqn_store_lane_info(const oracle_dqn_store *s, uint64_t lane, uint64_t *num_steps, uint64_t *num_episodes,
                                uint64_t *total_step_count);
void oracle_dqn_store_lane_dump(const oracle_dqn_store *s, uint64_t lane, int32_t *tags, uint64_t *episode_lens);
void oracle_dqn_store_step(const oracle_dqn_store *s, uint64_t lane, uint64_t abs_index, float *obs, uint8_t *action,
                           float *reward, uint8_t *next, float *next_obs);
int oracle_lanes_rollout_dqn(oracle_lanes *l, oracle_dqn_store *st, oracle_mlp_shape qs, const float *qparams,
                             uint64_t T, double eps, uint8_t *flags_out);
int64_t oracle_dqn_sample(const oracle_dqn_store *st, oracle_prng *agent_rng, uint64_t minibatch_steps,
                          uint32_t *lane_out, uint32_t *start_out, uint32_t *len_out, uint64_t cap,
                          uint64_t *n_steps_out);
void oracle_dqn_minibatch(const oracle_dqn_store *st, uint64_t n_eps, const uint32_t *lanes, const uint32_t *starts,
                          const uint32_t *lens, oracle_mlp_shape qs, const float *qparams, float gamma,
                          int one_step_td, float *obs_out, int64_t *actions_out, float *targets_out);
void oracle_dqn_grad_f32(oracle_mlp_shape s, const float *params, const float *obs, const int64_t *actions,
                         const float *targets, uint64_t n, float *grad_out, float *loss_out);
void oracle_dqn_grad_f64(oracle_mlp_shape s, const double *params, const double *obs, const int64_t *actions,
                         const double *targets, uint64_t n, double *grad_out, double *loss_out);
int oracle_dqn_update_f32(const oracle_dqn_store *st, oracle_prng *agent_rng, oracle_mlp_shape qs, float *qparams,
                          oracle_adam_state *opt, const oracle_adam_cfg *acfg, uint64_t minibatch_steps,
                          uint64_t opt_steps, float gamma, int one_step_td, float *losses_out);
double oracle_exploration_rate(int kind, double start, double end, uint64_t period, uint64_t global_steps,
                               int training);
oracle_bound oracle_collection_update_size(int kind, uint64_t first, uint64_t rest, uint64_t global_steps);

/* ---------------------------------------------------------------- recurrent configuration (seq.c, seq_impl.inc) */
/* ChainConfig<GruConfig | LstmConfig, MlpConfig> (modules/mod.rs:14, chain.rs:12-56); `cell`: 0 = Gru (seq/rnn/gru.rs),
 * 1 = Lstm (seq/rnn/lstm.rs).  The recurrent state handed to the step function is [h] resp. [h; c]. */
enum { ORACLE_CELL_GRU = 0, ORACLE_CELL_LSTM = 1 };
typedef struct { uint32_t in_dim, hidden, mlp_hidden, out_dim, cell; } oracle_gru_shape;
/* flat parameter order = trainable_variables(): W_ih [3H,in], W_hh [3H,H], b_ih, b_hh, W1 [H2,H], b1, W2 [A,H2], b2 */
uint64_t oracle_gru_num_params(oracle_gru_shape s);
void oracle_gru_init(oracle_gru_shape s, uint64_t seed, float *params);
/* RnnBaseConfig::num_layers > 1 (seq/rnn/mod.rs:223-257): per layer [W_ih, W_hh, b_ih, b_hh] (layer 0 reads in_dim features,
 * layer l > 0 the hidden output of layer l - 1), then the head as above; stack_impl.inc */
uint64_t oracle_stack_num_params(oracle_gru_shape s, uint32_t num_layers);
void oracle_stack_init(oracle_gru_shape s, uint32_t num_layers, uint64_t seed, float *params);
/* ... with RnnBaseConfig's three initializers and the chain MLP's two (kind / scale codes of oracle_mlp_layers_init): nn.c */
typedef struct { int32_t kind, scale; double value; } oracle_init_spec;
void oracle_stack_init_with(oracle_gru_shape s, uint32_t num_layers, uint64_t seed, const oracle_init_spec *inits,
                            float *params);
void oracle_stack_seq_forward_f32(oracle_gru_shape s, uint32_t num_layers, const float *params, uint64_t n, uint64_t T,
                                  const float *obs, const uint8_t *flag, const float *term_obs, float *out, float *succ_out);
void oracle_stack_seq_forward_f64(oracle_gru_shape s, uint32_t num_layers, const double *params, uint64_t n, uint64_t T,
                                  const double *obs, const uint8_t *flag, const double *term_obs, double *out,
                                  double *succ_out);
void oracle_gru_step_f32(oracle_gru_shape s, const float *params, const float *x, float *h, float *out);
void oracle_gru_step_f64(oracle_gru_shape s, const double *params, const double *x, double *h, double *out);
void oracle_gru_seq_forward_f32(oracle_gru_shape s, const float *params, uint64_t n, uint64_t T, const float *obs,
                                const uint8_t *flag, const float *term_obs, float *out, float *succ_out);
void oracle_gru_seq_forward_f64(oracle_gru_shape s, const double *params, uint64_t n, uint64_t T, const double *obs,
                                const uint8_t *flag, const double *term_obs, double *out, double *succ_out);
void oracle_gru_seq_backward_f32(oracle_gru_shape s, const float *params, uint64_t n, uint64_t T, const float *obs,
                                 const uint8_t *flag, const float *dout, float *grad_out);
void oracle_gru_seq_backward_f64(oracle_gru_shape s, const double *params, uint64_t n, uint64_t T, const double *obs,
                                 const uint8_t *flag, const double *dout, double *grad_out);

void oracle_gru_seq_jvp_f32(oracle_gru_shape s, const float *params, const float *tangent, uint64_t n, uint64_t T,
                            const float *obs, const uint8_t *flag, float *out_dot);
void oracle_gru_seq_jvp_f64(oracle_gru_shape s, const double *params, const double *tangent, uint64_t n, uint64_t T,
                            const double *obs, const uint8_t *flag, double *out_dot);

/* Lanes of an env with IndexSpace observations: Chain (`memory.num_actions == 0`) or MemoryGame.  MemoryGame lanes
 * draw their initial states from the lane's env stream SEQUENTIALLY (`env_pos` = the stream's word position; the
 * step itself draws nothing), exactly like one worker thread's env Prng in the reference. */
typedef struct {
  oracle_chain env;
  oracle_memory memory;
  int limit_kind;
  uint64_t max_steps, seed_env, seed_actor, n_lanes, lane_offset;
  uint64_t *state, *steps_remaining, *reset_count;
  uint64_t *initial, *env_pos; /* MemoryGame only */
  uint64_t t_global;
  /* DeterministicBandit (src/envs/bandits.rs:109-116): one state, reward = the arm's value, every step terminates */
  int bandit;
  double bandit_values[2];
} oracle_chain_lanes;
/* Lanes of DeterministicBandit::from_values([v0, v1]) (bandits.rs:66-77 step, :109-116).  The singleton observation
 * is presented as one-hot(5) of state 0 — its one constant feature (NonEmptyFeatures) padded to the five inputs the
 * device kernels are built for; the discount factor is 1 (bandits.rs:52-54). */
oracle_chain_lanes *oracle_bandit_lanes_new(const double *values, uint64_t n_lanes, uint64_t lane_offset,
                                            uint64_t seed_env, uint64_t seed_actor);
oracle_chain_lanes *oracle_memory_lanes_new(uint64_t num_actions, uint64_t history_len, int limit_kind,
                                            uint64_t max_steps, uint64_t n_lanes, uint64_t lane_offset,
                                            uint64_t seed_env, uint64_t seed_actor);
void oracle_memory_lanes_get_extra(const oracle_chain_lanes *l, uint64_t *initial, uint64_t *env_pos);
oracle_chain_lanes *oracle_chain_lanes_new(uint64_t size, int limit_kind, uint64_t max_steps, uint64_t n_lanes,
                                           uint64_t lane_offset, uint64_t seed_env, uint64_t seed_actor);
void oracle_chain_lanes_free(oracle_chain_lanes *l);
void oracle_chain_lanes_reset(oracle_chain_lanes *l);
uint32_t oracle_chain_lanes_obs_dim(const oracle_chain_lanes *l);
void oracle_chain_lanes_observe(const oracle_chain_lanes *l, float *obs_soa);
void oracle_chain_lanes_get_state(const oracle_chain_lanes *l, uint64_t *state, uint64_t *steps_remaining,
                                  uint64_t *reset_count);
void oracle_chain_lanes_step(oracle_chain_lanes *l, const uint8_t *actions, float *reward, uint8_t *flag,
                             float *obs_next_soa, float *term_obs_soa);
void oracle_chain_lanes_rollout_gru(oracle_chain_lanes *l, oracle_gru_shape ps, const float *params, uint64_t T,
                                    float *obs, uint8_t *action, float *reward, uint8_t *flag, float *term_obs,
                                    int n_threads);
void oracle_chain_lanes_rollout_mlp(oracle_chain_lanes *l, oracle_mlp_shape ps, const float *params, uint64_t T,
                                    float *obs, uint8_t *action, float *reward, uint8_t *flag, float *term_obs);
void oracle_lanes_rollout_gru(oracle_lanes *l, oracle_gru_shape ps, const float *params, uint64_t T, float *obs,
                              uint8_t *action, float *reward, uint8_t *flag, float *term_obs, int n_threads);
void oracle_seq_gae(uint64_t n, uint64_t T, const float *values, const float *succ_values, const float *reward,
                    const uint8_t *flag, float gamma, float lambda, float *adv_out, float *rtg_out);
/* one-step TD targets from the teacher-forced outputs of a recurrent critic (values / successor values as in
 * oracle_seq_gae): r + gamma * V_next, Terminate -> 0, Interrupt / horizon cut -> succ */
void oracle_seq_one_step_targets(uint64_t n, uint64_t T, const float *values, const float *succ_values,
                                 const float *reward, const uint8_t *flag, float gamma, float *out);
void oracle_seq_policy_dlogits_f32(uint64_t B, const float *logits, const uint8_t *actions, const float *adv,
                                   const float *logp0, int mode, float clip_lo, float clip_hi, float *dlogits,
                                   float *logp_out, double *loss_sum_out, double *entropy_sum_out);

/* ---------------------------------------------------------------- CPU baseline (simulation/train.rs:68-186) */
typedef struct {
  double rollout_seconds, update_seconds;
  uint64_t steps, episodes;
  double mean_episode_length;
  oracle_trpo_stats trpo;
  float critic_loss_first, critic_loss_last;
  /* the same update's full-batch passes (1 policy gradient, iterations + 1 Fisher-vector products, the line search's
   * evaluations, critic_steps x {critic gradient, Adam}) with every pass split over `n_threads` cores the way libtorch's
   * intra-op pool splits the reference's matmuls; 0 when not requested (intraop_threads == 0) */
  double update_intraop_seconds;
  uint32_t update_intraop_threads;
} oracle_period_stats;

/* One train_parallel period of CartPole+VisibleStepLimit MLP-TRPO: `n_threads` OS threads each run the
 * scalar Steps::step loop for `steps_per_thread` steps (TakeAlignedSteps with slack), then the calling
 * thread does GAE -> TRPO -> critic update on the packed batch. */
double oracle_cartpole_rollout_only(uint64_t seed, uint32_t n_threads, uint64_t steps_per_thread, uint64_t clear_every,
                                    uint64_t max_steps, uint32_t hidden, const float *policy_params,
                                    uint64_t *steps_out);
void oracle_cartpole_trpo_period(uint64_t seed, uint64_t period_index, uint32_t n_threads, uint64_t steps_per_thread,
                                 uint64_t slack_steps, uint64_t max_steps, uint32_t hidden, float *policy_params,
                                 float *critic_params, oracle_adam_state *critic_opt, uint64_t critic_steps,
                                 oracle_period_stats *stats);
/* ... and, when intraop_threads > 0, the update's passes once more with intra-op parallelism (timing only: parameters
 * and optimiser state are left as the single-threaded update made them) */
void oracle_cartpole_trpo_period_ex(uint64_t seed, uint64_t period_index, uint32_t n_threads, uint64_t steps_per_thread,
                                    uint64_t slack_steps, uint64_t max_steps, uint32_t hidden, float *policy_params,
                                    float *critic_params, oracle_adam_state *critic_opt, uint64_t critic_steps,
                                    uint32_t intraop_threads, oracle_period_stats *stats);

/* The period's legs one at a time (bench.py's cpu_baseline): collect once, then time the update on one thread over a
 * prefix of the sample and with its passes split over any number of threads.  No leg changes the parameters. */
typedef struct oracle_cpu_sample oracle_cpu_sample;
oracle_cpu_sample *oracle_cpu_sample_collect(uint64_t seed, uint32_t n_threads, uint64_t steps_per_thread,
                                             uint64_t slack_steps, uint64_t max_steps, uint32_t hidden,
                                             const float *policy_params, const float *critic_params,
                                             oracle_period_stats *stats);
double oracle_cpu_sample_update_one_thread(oracle_cpu_sample *s, uint64_t n_prefix, uint64_t critic_steps,
                                           oracle_period_stats *stats);
double oracle_cpu_sample_update_intraop(oracle_cpu_sample *s, uint64_t n_prefix, uint64_t critic_steps,
                                        uint32_t n_threads);
void oracle_cpu_sample_free(oracle_cpu_sample *s);

#ifdef __cplusplus
}
#endif
#endif
