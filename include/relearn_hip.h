/* relearn_hip.h — C ABI of the MI355X-native batched rollout-and-update engine.
 *
 * This is the drop-in boundary for ONE hot path of edlanglois/relearn (v0.3.1):
 *   vectorised env step -> MLP policy/critic forward -> GAE/return scan -> TRPO + critic update.
 * The reference has no FFI of its own (it is 100 % Rust over tch/libtorch); the boundary it offers
 * is its trait surface.  Every entry point below names the reference interface it stands in for
 * (paths relative to the reference tree), and INTEGRATION.md shows the `extern "C"` block a Rust
 * maintainer adds to bind them behind `Environment` / `Actor` / `Agent` / `BatchUpdate`.
 *
 * Conventions
 *   - every function returns int32_t: 0 = RL_OK, otherwise an rl_status code; no exception or abort
 *     crosses the boundary; `rl_last_error(engine)` returns a static-lifetime-until-next-call message;
 *   - handles are opaque, created/destroyed by the library; device memory is library-owned;
 *   - host buffers are caller-owned and only borrowed for the duration of a call;
 *   - a handle is NOT thread-safe: one engine per GPU, driven by one host thread;
 *   - calls enqueue work on the engine's HIP stream; functions that return data to the host
 *     synchronise that stream, the others may return before the GPU has finished
 *     (`rl_engine_sync` waits);
 *   - there is NO CPU fallback: without a visible gfx950 device `rl_engine_create` fails with
 *     RL_ERR_NO_DEVICE.
 *
 * Data layout in HBM: one env per lane, struct-of-arrays, time-major trajectories
 *   obs[d][t][lane] f32, action[t][lane] u8, reward[t][lane] f32, flag[t][lane] u8 (successor code).
 * The flat sample index used by the update kernels is b = t * n_lanes + lane.
 */
#ifndef RELEARN_HIP_H
#define RELEARN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Incremented whenever an entry point is added or removed, an argument list or a struct layout changes, or a status code
 * is renumbered: a binding checks rl_abi_version() against the value it was generated from.  6 (round 6): covers the two
 * entry points round 5 added under the old number (rl_actor_critic_update_begin / _finish); nothing was added since. */
#define RL_ABI_VERSION 6

/* ---------------------------------------------------------------------------------------------
 * Status codes.  The non-generic ones mirror the reference's error enums:
 *   BuildAgentError              src/agents/mod.rs:219-226
 *   BuildEnvError                src/envs/builders.rs:62-66
 *   WriteExperienceError::Full   src/agents/buffers/mod.rs:225-228
 *   PackingError                 src/torch/packed.rs:18-23
 *   OptimizerStepError           src/torch/optimizers/mod.rs:80-94 (reported in *_stats.status;
 *                                NaN variants additionally make the call return RL_ERR_OPT_NAN because
 *                                the reference panics on them, src/torch/agents/policies/trpo.rs:154-162)
 */
typedef enum {
  RL_OK = 0,
  RL_ERR_INVALID_ARGUMENT = 1,
  RL_ERR_HIP = 2,
  RL_ERR_NO_DEVICE = 3,
  RL_ERR_BUILD_AGENT = 4,
  RL_ERR_BUILD_ENV = 5,
  RL_ERR_BUFFER_FULL = 6,
  RL_ERR_PACKING = 7,
  RL_ERR_COMM = 8,
  RL_ERR_OPT_NAN = 9,
  RL_ERR_UNSUPPORTED = 10
} rl_status;

/* Successor codes stored in `flag` (src/envs/mod.rs:257-269) */
enum { RL_SUCC_CONTINUE = 0, RL_SUCC_TERMINATE = 1, RL_SUCC_INTERRUPT = 2 };
/* OptimizerStepError as an integer (src/torch/optimizers/mod.rs:80-94) */
enum { RL_OPT_OK = 0, RL_OPT_LOSS_NOT_IMPROVING = 1, RL_OPT_CONSTRAINT_VIOLATED = 2, RL_OPT_NAN_LOSS = 3,
       RL_OPT_NAN_CONSTRAINT = 4 };
/* env kinds / step-limit wrappers (src/envs/cartpole.rs, chain.rs, wrappers/step_limit.rs:13,97) */
enum { RL_ENV_CARTPOLE = 0, RL_ENV_CHAIN = 1, RL_ENV_MEMORY = 2, RL_ENV_BANDIT = 3 };
enum { RL_LIMIT_NONE = 0, RL_LIMIT_LATENT = 1, RL_LIMIT_VISIBLE = 2 };

typedef struct rl_engine rl_engine;
typedef struct rl_env rl_env;
typedef struct rl_mlp rl_mlp;
typedef struct rl_traj rl_traj;
typedef struct rl_adam rl_adam;
typedef struct rl_dqn rl_dqn;

/* ---------------------------------------------------------------------------------------------
 * Engine (one per GPU).  Stands in for `Device::cuda_if_available()` + the libtorch runtime the
 * reference configures in ActorCriticConfig.device (src/torch/agents/actor_critic.rs:20-45). */
int32_t rl_abi_version(void);
int32_t rl_device_count(int32_t *count);
int32_t rl_engine_create(int32_t device_ordinal, rl_engine **out);
int32_t rl_engine_destroy(rl_engine *engine);
int32_t rl_engine_sync(rl_engine *engine);
const char *rl_last_error(const rl_engine *engine);
/* name of the device, gcnArchName (must start with "gfx950"), CU count */
int32_t rl_engine_info(const rl_engine *engine, char *name_out, size_t name_cap, char *arch_out, size_t arch_cap,
                       int32_t *compute_units);
/* Kernel selection: 0 (default) = best kernels for the shape (fused matrix-pipe update kernels at hidden 128,
 * obs_dim 5); 1 = the v1 kernels only (what other shapes fall back to; kept selectable as an in-library cross-check:
 * same results within the tolerances stated in tests/test_gpu_parity.py). */
int32_t rl_engine_set_kernel_variant(rl_engine *engine, int32_t variant);
/* Numeric range of the fused kernels (variant 0 at hidden 128, obs_dim 5: rl_trpo_update, rl_ppo_update,
 * rl_reinforce_update, rl_critic_update / rl_values_opt_update / rl_actor_critic_update, rl_policy_gradient / _fvp /
 * _loss_kl, rl_critic_gradient, rl_dqn_update).  The reference's forward is plain f32 (Mlp::forward,
 * src/torch/modules/ff/mlp.rs:139-151) and has no bound beyond f32's own.  The fused kernels compute every product of
 * layer 1 exactly on the bf16 matrix pipe with the weights scaled by 2^96 and read relu' off the scaled accumulator, which
 * is exact — bit-for-bit the mask of the f32 pre-activation's sign — when, for every hidden unit j,
 *     sum_k |W1[j][k]| * max|obs| + |b1[j]|  <  2^31      (the scaled accumulator stays finite), and
 *     max(max_k |W1[j][k]| * min{|obs| : obs != 0}, |b1[j]|)  >=  2^-46   unless the unit's weights and bias are all zero
 *                                                         (a non-zero pre-activation cannot fall below 2^-96),
 * and every observation is finite.  With Glorot-initialised 5-128 layers (|w| <= 0.22) that is |obs| < 1.9e9 whatever the
 * small end when the biases are non-zero, and |obs| >= 7e-14 when they are zero.  Observation pieces below 2^-126 (bf16
 * subnormals: |obs| < 2^-110) may be flushed by the matrix pipe; inside the range above their contribution is below the
 * f32 rounding of the terms that carry the pre-activation.
 * The library checks the condition in the first fused launch of each module of every call (the later launches of the
 * same call start from parameters that call produced itself, in steps bounded by the learning rate or the KL constraint),
 * from the weights the launch has just loaded and the magnitude range of the trajectory's observation planes (measured
 * once per rollout / rl_traj_write; for DQN the collecting rollout folds the observations it writes to the replay
 * store into the same range words).  Outside the range the call returns RL_ERR_UNSUPPORTED
 * — never a silently wrong mask; its outputs are then not valid, and an update is refused WHOLE: the launch that finds the
 * violation sets a veto word on the device, which every optimiser and line-search kernel behind it reads, so parameters,
 * Adam moments and step count are what they were at entry (rl_dqn_update: put back from a snapshot).  The policy chain
 * and the critic chain have a word each: under rl_actor_critic_update[_begin] a violation of one chain refuses that
 * chain, the error names it, and the other chain's step stands (like the NaN policy step above).
 * Kernel variant 1 is plain f32 and takes any magnitudes.  tests/test_gpu_numeric_range.py. */
/* HIP-event timing of everything enqueued between begin and end on the engine stream (milliseconds) */
int32_t rl_timer_begin(rl_engine *engine);
int32_t rl_timer_end(rl_engine *engine, float *elapsed_ms);
/* Per-kernel-class accumulated device time since the last reset, measured with HIP events around each
 * launch when profiling is on (off by default: events add host overhead).  classes: see rl_kernel_class. */
typedef enum {
  RL_K_ENV_STEP = 0, RL_K_ROLLOUT = 1, RL_K_VALUES = 2, RL_K_GAE = 3, RL_K_POLICY_PASS = 4, RL_K_BACKWARD = 5,
  RL_K_REDUCE = 6, RL_K_SMALL = 7, RL_K_CRITIC_FWD = 8, RL_K_ALLREDUCE = 9, RL_K_CRITIC_FUSED = 10,
  RL_K_POLICY_FUSED = 11, RL_K_POLICY_FVP = 12 /* Fisher-vector launches of the fused policy kernel */,
  RL_K_CLASS_COUNT = 13
} rl_kernel_class;
int32_t rl_profile_enable(rl_engine *engine, int32_t on);
int32_t rl_profile_read(rl_engine *engine, double *total_ms_out /*[RL_K_CLASS_COUNT]*/,
                        uint64_t *launches_out /*[RL_K_CLASS_COUNT]*/, int32_t reset);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: env lanes are sharded over ranks; the only exchange is an RCCL all-reduce (sum, f32)
 * of the <= 4 KiB reduced gradient / Hessian-vector / scalar vectors.  The reference has no
 * collective at all (threads + shared memory, src/simulation/train.rs:98-180); this replaces the
 * `Vec<buffer>` hand-off of train.rs:180.  `unique_id` is the 128-byte ncclUniqueId produced on
 * rank 0 and distributed by the caller (bench.py uses torch.distributed for that). */
/* RL_OK when the collective library can be bound in this process (dlopen of the librccl next to the HIP runtime in
 * use), RL_ERR_COMM otherwise.  Touches no GPU: every rank of a job calls it and the job agrees on the answer BEFORE
 * any rank enters the collective rl_comm_init (a rank that cannot load RCCL must not leave its peers blocked there). */
int32_t rl_comm_available(void);
/* Which shared objects the collective runs on: the librccl this library bound (empty before it is bound) and the
 * libamdhip64 this library is linked to — they must sit in the same directory (a host program may map a second copy of
 * the ROCm libraries; a communicator of the other copy's RCCL cannot use this runtime's streams). */
int32_t rl_comm_library_paths(char *rccl_out, size_t rccl_cap, char *hip_out, size_t hip_cap);
int32_t rl_comm_unique_id(uint8_t id_out[128]);
int32_t rl_comm_init(rl_engine *engine, int32_t rank, int32_t n_ranks, const uint8_t unique_id[128]);
int32_t rl_comm_destroy(rl_engine *engine);
/* Peer-mailbox transport for the same all-reduce (ranks = processes of ONE node): every rank owns a mailbox in its HBM,
 * exports it (rl_comm_ipc_handle: a 64-byte hipIpcMemHandle; n_ranks <= 16), the caller gathers the handles of all ranks
 * in rank order and hands them to rl_comm_init_ipc, which maps the peers' mailboxes.  An all-reduce is then ONE kernel
 * per rank: publish into every peer's mailbox, wait for every peer's sequence number (bounded), add the rows in rank
 * order — bit-identical replicas, no broadcast.  Vectors of <= 2048 floats (the feed-forward updates; recurrent modules
 * keep RCCL).  A peer that never arrives makes the next synchronising call return RL_ERR_COMM.  Undone by
 * rl_comm_destroy (all ranks, after a barrier of their own). */
int32_t rl_comm_ipc_handle(rl_engine *engine, int32_t n_ranks, uint8_t handle_out[64]);
int32_t rl_comm_init_ipc(rl_engine *engine, int32_t rank, int32_t n_ranks, const uint8_t *handles /* [n_ranks][64] */);
/* Three all-reduces of known vectors through whatever collective is installed; RL_ERR_COMM when a sum is wrong or a
 * peer does not arrive.  Collective: every rank calls it.  Lets a job agree that a transport works before relying on it. */
int32_t rl_comm_selftest(rl_engine *engine);
/* Host-staged collective for machines (or tests) without a usable RCCL communicator: for every all-reduce the library
 * copies the vector to the host, calls `fn(ctx, buf, count)` — which must sum `buf` element-wise over all ranks in place
 * (a gloo / MPI all-reduce, for instance) and return 0 — and copies the result back.  Same arithmetic contract as
 * rl_comm_init (sample-weighted means over all ranks, identical redundant updates); two PCIe hops and one host
 * collective per call, so it is a fallback, not the fast path.  Undone by rl_comm_destroy. */
typedef int32_t (*rl_host_allreduce_fn)(void *ctx, float *buf, uint64_t count);
int32_t rl_comm_init_host(rl_engine *engine, int32_t rank, int32_t n_ranks, rl_host_allreduce_fn fn, void *ctx);

/* ---------------------------------------------------------------------------------------------
 * Environments.  Mirrors `Environment::{initial_state, observe, step}` (src/envs/mod.rs:76-127) for N
 * lanes at once; state is explicit and owned by the handle. */
typedef struct {
  /* PhysicalConstants (src/envs/cartpole.rs:157-191) */
  double gravity, mass_cart, mass_pole, length_half_pole, friction_cart, friction_pole, time_step;
  /* EnvironmentParams (src/envs/cartpole.rs:194-216) */
  double action_force, max_pos, max_angle, discount_factor;
} rl_cartpole_params;
/* CartPole::default() */
int32_t rl_cartpole_params_default(rl_cartpole_params *p);

typedef struct {
  int32_t kind;          /* RL_ENV_CARTPOLE | RL_ENV_CHAIN (Chain::default, src/envs/chain.rs:38-45: obs = one-hot(5)) |
                          * RL_ENV_MEMORY (MemoryGame, src/envs/memory.rs:24-115: obs = one-hot(num_actions + history_len)) */
  int32_t limit_kind;    /* RL_LIMIT_* : `env.wrap(VisibleStepLimit::new(max_steps))` */
  uint64_t max_steps;    /* max_steps_per_episode (< 2^32) */
  uint64_t n_lanes;      /* lanes resident on THIS engine */
  uint64_t lane_offset;  /* global id of lane 0 (random streams are keyed by global lane id) */
  uint64_t seed_env;     /* env stream seed   (the `rng_env` of Steps, src/simulation/steps.rs:15-28) */
  uint64_t seed_actor;   /* actor stream seed (the `rng_actor` of Steps) */
  rl_cartpole_params cartpole;
  uint64_t chain_size;   /* RL_ENV_CHAIN: number of states (Chain::default: 5; the kernels are built for 5); 0 = 5 */
  /* RL_ENV_MEMORY: MemoryGame { num_actions, history_len } (memory.rs:24-40).  The device kernels are built for 2
   * actions and 5 observation features, i.e. MemoryGame::new(2, 3); other sizes -> RL_ERR_BUILD_ENV (the scalar host
   * env takes any).  0 / 0 = (2, 3).  Episodes last history_len + 1 steps; the only random draw is
   * `gen_range(0..num_actions)` in initial_state, taken sequentially from the lane's env stream. */
  uint64_t memory_num_actions, memory_history_len;
  /* RL_ENV_BANDIT: DeterministicBandit::from_values([v0, v1]) (src/envs/bandits.rs:109-116; Bandit::step :66-77): one
   * state, reward = the chosen arm's value, every step terminates, discount factor 1, no step limit.  The singleton
   * observation is presented as one-hot(5) of state 0 (its one constant feature padded to the kernels' five inputs). */
  double bandit_values[2];
} rl_env_config;

int32_t rl_env_create(rl_engine *engine, const rl_env_config *cfg, rl_env **out);
int32_t rl_env_destroy(rl_env *env);
/* number of observation features (CartPole 4, +1 `remaining` under VisibleStepLimit) and actions */
int32_t rl_env_dims(const rl_env *env, uint32_t *obs_dim, uint32_t *n_actions);
/* Environment::initial_state for every lane (a new episode in every lane) */
int32_t rl_env_reset(rl_env *env);
/* Environment::observe -> feature vectors, SoA [obs_dim][n_lanes] f32 (host buffer) */
int32_t rl_env_observe(rl_env *env, float *obs_out);
/* Environment::step for every lane with host-side actions (u8 indices); lanes whose episode ends start
 * a new one.  Outputs (host, may be NULL): reward[n] f32, flag[n] u8, next observation features
 * [obs_dim][n] (of the NEW episode when the lane was reset), interrupt successor features [obs_dim][n]
 * (valid where flag == RL_SUCC_INTERRUPT). */
int32_t rl_env_step(rl_env *env, const uint8_t *actions, float *reward_out, uint8_t *flag_out, float *obs_out,
                    float *term_obs_out);
/* Same step with everything device-resident (no PCIe): actions from / results to the env's own HBM
 * staging buffers.  `rl_env_upload_actions` fills the action buffer once; used by bench.py. */
int32_t rl_env_upload_actions(rl_env *env, const uint8_t *actions);
int32_t rl_env_step_resident(rl_env *env);
/* raw lane state for parity tests: state4 [4][n] f64 (x, xdot, theta, thetadot), nv_pos[n] i32 (cached sign),
 * steps_remaining[n] u64, reset_count[n] u64.  RL_ENV_CHAIN / RL_ENV_MEMORY: state4[0][lane] holds the state index (exact
 * in f64); RL_ENV_MEMORY: state4[1] the episode's initial state, state4[2] the word position of the lane's env stream;
 * the other state4 rows and nv_pos are unused */
int32_t rl_env_get_state(rl_env *env, double *state4, int32_t *nv_pos, uint64_t *steps_remaining,
                         uint64_t *reset_count);
int32_t rl_env_set_state(rl_env *env, const double *state4, const int32_t *nv_pos, const uint64_t *steps_remaining,
                         const uint64_t *reset_count);

/* Test hook for the random streams (Prng = ChaCha8Rng used as a counter-based generator, src/lib.rs:68):
 * `n_words` consecutive 32-bit words, starting at word `first_word`, of ChaCha8Rng::seed_from_u64(seed) with
 * set_stream(stream), computed ON THE DEVICE by the block function the rollouts, resets and env steps draw from
 * (include/rl_chacha.h as compiled for gfx950).  tests/test_gpu_chacha_independent.py compares them with a from-scratch
 * implementation (RFC 7539 quarter round, 8 rounds; rand_core's PCG32 seed expansion) that shares no code with the
 * library or the oracle. */
int32_t rl_debug_stream_words(rl_engine *engine, uint64_t seed, uint64_t stream, uint64_t first_word, uint32_t n_words,
                              uint32_t *words_out);

/* ---------------------------------------------------------------------------------------------
 * Modules.  `MlpConfig::build_module` with one hidden layer, ReLU, identity output
 * (src/torch/modules/ff/mlp.rs:25-69); parameters are exchanged in the reference's flat order
 * (kernel [out][in] then bias per layer: ff/linear.rs:108-110, torch/utils.rs:10-22). */
int32_t rl_mlp_create(rl_engine *engine, uint32_t in_dim, uint32_t hidden, uint32_t out_dim, rl_mlp **out);
/* Activation (src/torch/modules/ff/activation.rs:11-20,85-92): the unit variants of the reference's enum, in its
 * declaration order.  Sigmoid and tanh are the engine's deterministic ones (include/rl_detmath.h, within 2.5 ulp of
 * libm), shared with the oracle like every transcendental of the bit-exact paths. */
typedef enum {
  RL_ACT_IDENTITY = 0,
  RL_ACT_RELU = 1,
  RL_ACT_SIGMOID = 2,
  RL_ACT_TANH = 3
} rl_activation;
/* MlpConfig { hidden_sizes, activation, output_activation, .. } (src/torch/modules/ff/mlp.rs:13-34,139-151): in_dim ->
 * hidden_sizes[0] -> ... -> out_dim with `activation` after every hidden layer and `output_activation` on the output;
 * parameters flat as [W, b] per layer in layer order.  `n_hidden` in [0, 4], every width in [1, 256], in_dim in [1, 8]
 * (the envs of this library have 4 or 5 features: other widths work on host-made histories, rl_traj_write),
 * out_dim in {1, 2}, activations from rl_activation; anything else -> RL_ERR_BUILD_AGENT.  One hidden layer of at most
 * 128 units with the reference's defaults (Relu, Identity) is rl_mlp_create — the fused kernels every BASELINE
 * configuration runs on, which are built for exactly that; any other shape or activation runs per-layer kernels
 * (relearn_amd/csrc/kernels_general.hip: the general path, not the fast one) behind the same entry points — rollouts,
 * GAE, TRPO / PPO / REINFORCE and critic updates, DQN (collection one launch sequence per step), row-wise forward,
 * actor serialisation. */
int32_t rl_mlp_create_layers(rl_engine *engine, uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden,
                             uint32_t out_dim, int32_t activation, int32_t output_activation, rl_mlp **out);
/* The same with MlpConfig::linear_config's choice of bias vectors (LinearConfig { kernel_init, bias_init: Option<..> },
 * ff/linear.rs:13-33,54-68): bias = 0 builds every layer WITHOUT a bias vector (bias_init = None) — the flat parameter
 * vector then holds the kernels only (trainable_variables of such a Linear: the kernel, linear.rs:104-117), actor
 * documents carry `bias: null`, rl_mlp_init draws the kernels only and rl_mlp_init_with takes bias_init = NULL.  Such
 * modules run the per-layer kernels (also the 1 x <= 128 Relu shape the fused kernels would otherwise take). */
int32_t rl_mlp_create_config(rl_engine *engine, uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden,
                             uint32_t out_dim, int32_t activation, int32_t output_activation, int32_t bias, rl_mlp **out);
int32_t rl_mlp_destroy(rl_mlp *mlp);
int32_t rl_mlp_num_params(const rl_mlp *mlp, uint64_t *n);
/* Linear::new Glorot-uniform init (ff/linear.rs:54-68) from the engine-defined stream ChaCha8(seed) */
int32_t rl_mlp_init(rl_mlp *mlp, uint64_t seed);
/* LinearConfig { kernel_init, bias_init } (src/torch/modules/ff/linear.rs:13-33) with the reference's `Initializer`
 * (src/torch/initializers.rs:8-64,152-176,328-364): Zeros; Constant(value); Uniform(scale): U(+-sqrt(3 var));
 * Normal(scale): N(0, var); Orthogonal: QR of a normal [rows, cols] matrix (transposed when rows < cols), columns
 * multiplied by the sign of R's diagonal.  var = VarianceScale::variance: Constant(value) | 1 / fan_in | 1 / fan_out |
 * 2 / (fan_in + fan_out) with Linear::new's fan_in = in_dim + 1 for kernel AND bias and fan_out from the tensor's
 * shape.  Feed-forward modules only; every layer uses the same pair, as MlpConfig::linear_config does.  The draws come
 * from the engine's stream ChaCha8(seed), stream 0, in flat parameter order: one f32 per uniform element, Box-Muller
 * on consecutive pairs for normal ones (libtorch's generator is never seeded by the reference).
 * `bias_init` NULL <=> the module was built without bias vectors (rl_mlp_create_config(..., bias = 0):
 * LinearConfig::bias_init = None); a mismatch -> RL_ERR_INVALID_ARGUMENT.  rl_mlp_init(m, seed) = both Uniform(FanAvg),
 * the default. */
typedef enum { RL_INIT_ZEROS = 0, RL_INIT_CONSTANT = 1, RL_INIT_UNIFORM = 2, RL_INIT_NORMAL = 3, RL_INIT_ORTHOGONAL = 4 } rl_init_kind;
typedef enum { RL_SCALE_CONSTANT = 0, RL_SCALE_FAN_IN = 1, RL_SCALE_FAN_OUT = 2, RL_SCALE_FAN_AVG = 3 } rl_variance_scale;
typedef struct {
  int32_t kind;   /* rl_init_kind */
  int32_t scale;  /* rl_variance_scale, read by Uniform and Normal */
  double value;   /* Constant(value); VarianceScale::Constant(value) */
} rl_initializer;
int32_t rl_mlp_init_with(rl_mlp *mlp, uint64_t seed, const rl_initializer *kernel_init, const rl_initializer *bias_init);
/* The same for a recurrent chain: RnnBaseConfig { input_weights_init, hidden_weights_init, bias_init }
 * (seq/rnn/mod.rs:20-45; RnnWeights::new, :223-257: per layer W_ih from input_weights_init, W_hh from
 * hidden_weights_init, b_ih and b_hh from bias_init — 1-D tensors: fan_in 1, fan_out = gate rows) and the chain's
 * MlpConfig::linear_config { kernel_init, bias_init } (fan_in = in + 1), layer after layer from the same stream.
 * rl_mlp_init(m, seed) on a recurrent chain = (Uniform(FanAvg), Orthogonal, Zeros; Uniform(FanAvg) x 2), the defaults.
 * bias_init NULL <=> the module was built without recurrent bias vectors (rl_rnn_mlp_create_config, bias = 0:
 * RnnBaseConfig::bias_init = None); a mismatch -> RL_ERR_INVALID_ARGUMENT; mlp_bias_init NULL -> RL_ERR_UNSUPPORTED (the
 * chain's MLP is built with bias vectors); Orthogonal on a bias -> RL_ERR_INVALID_ARGUMENT (init_orthogonal asserts two
 * dimensions, initializers.rs:331-334). */
int32_t rl_rnn_mlp_init_with(rl_mlp *module, uint64_t seed, const rl_initializer *input_weights_init,
                             const rl_initializer *hidden_weights_init, const rl_initializer *bias_init,
                             const rl_initializer *mlp_kernel_init, const rl_initializer *mlp_bias_init);
int32_t rl_params_get(rl_mlp *mlp, float *host, uint64_t n);
int32_t rl_params_set(rl_mlp *mlp, const float *host, uint64_t n);
/* Forward::forward on host rows [n_rows][in_dim] -> [n_rows][out_dim] (test/utility path) */
int32_t rl_mlp_forward(rl_mlp *mlp, const float *rows, uint64_t n_rows, float *out);

/* Recurrent module `GruMlpConfig = ChainConfig<GruConfig, MlpConfig>` (src/torch/modules/mod.rs:14;
 * chain.rs:12-56; seq/rnn/mod.rs:20-45,223-257; seq/rnn/gru.rs:20-98): in_dim -> GRU(gru_hidden) -> ReLU ->
 * Linear(gru_hidden, mlp_hidden) -> ReLU -> Linear(mlp_hidden, out_dim).  The handle type is shared with the MLP:
 * every entry point that takes a module dispatches on its kind.  Flat parameter order = trainable_variables():
 * W_ih [3H, in] (gate rows r, z, n), W_hh [3H, H], b_ih [3H], b_hh [3H], then the MLP's kernel/bias pairs.
 * The kernels are built for gru_hidden = mlp_hidden = 128, in_dim = 5; narrower chains — in_dim 1..5, gru_hidden
 * 1..128, mlp_hidden 1..128 (RnnBaseConfig::hidden_size / ChainConfig::hidden_dim, MlpConfig::hidden_sizes = [h]; e.g.
 * the GRU(3 -> 4) of the reference's benches/rnn.rs) — run on the same kernels embedded with zero padding: padded
 * units stay exactly 0 and every real dot product only gains terms fma(0, 0, acc), so outputs, gradients and
 * Fisher-vector products are those of the narrow chain (forward bit-identical to the oracle's, tests/test_gpu_gru.py).
 * out_dim in {1, 2}; lanes in multiples of 32; trajectories of obs_dim = in_dim (rollouts need an env with five
 * observation features; narrower inputs come from rl_traj_write).  in_dim 6..8, widths 129..256 and stacked layers
 * (RnnBaseConfig::num_layers 2..4, rl_rnn_mlp_create) run the lane-per-thread kernels instead: see there.
 * rl_mlp_init: Glorot-uniform W_ih, orthogonal W_hh, zero biases (RnnBaseConfig::default), Linear::new for the MLP. */
int32_t rl_gru_mlp_create(rl_engine *engine, uint32_t in_dim, uint32_t gru_hidden, uint32_t mlp_hidden,
                          uint32_t out_dim, rl_mlp **out);
/* The same chain with the LSTM cell: `ChainConfig<LstmConfig, MlpConfig>` (Lstm = RnnBase<LstmImpl>,
 * src/torch/modules/seq/rnn/lstm.rs:12-51; note that the reference's alias `LstmMlpConfig` names the GRU chain,
 * modules/mod.rs:15).  Gate rows [i; f; g; o]: W_ih [4H, in], W_hh [4H, H], b_ih [4H], b_hh [4H]; the episode state
 * is (h, c), both zero at the start of an episode (lstm.rs:22-31); c' = f * c + i * g, h' = o * tanh(c').  Same
 * shapes, initialisation rule and entry points as the GRU chain (rollout, GAE, TRPO, PPO, REINFORCE, critic update,
 * serialisation). */
int32_t rl_lstm_mlp_create(rl_engine *engine, uint32_t in_dim, uint32_t lstm_hidden, uint32_t mlp_hidden,
                           uint32_t out_dim, rl_mlp **out);
/* Both chains with RnnBaseConfig's fields spelled out (seq/rnn/mod.rs:20-45: hidden_size, num_layers; the initializers
 * are RnnBaseConfig::default's): cell = RL_CELL_GRU | RL_CELL_LSTM; in_dim 1..8, hidden_size and mlp_hidden 1..256.
 * num_layers 1..4 (0 -> RL_ERR_BUILD_AGENT, > 4 -> RL_ERR_UNSUPPORTED).  Stacked layers (num_layers > 1): flat order
 * per layer [W_ih, W_hh, b_ih, b_hh] (RnnWeights::new, seq/rnn/mod.rs:223-257) — layer 0 reads the in_dim features,
 * layer l > 0 the hidden output of layer l - 1 of the same step (Tensor::gru / ::lstm with num_layers, gru.rs:41-66),
 * no dropout — then the MLP's kernel/bias pairs; rl_mlp_init draws layer after layer from one stream; the actor
 * document holds 4 tensors per layer.  Stacked chains, chains of in_dim 6..8 and widths above 128 run general lane-per-thread kernels
 * at the module's own widths (kernels_seq_stack.hip: any lane count, one launch sequence per rollout step; the path for
 * shapes the fused 5 -> 128 -> 128 tile kernels do not cover, not a fast one).  Every entry point of the single-layer
 * chains applies: rl_rollout (two-action policies on either lane family), rl_seq_forward, rl_gae, rl_trpo_update,
 * rl_ppo_update, rl_policy_gradient / _fvp / _loss_kl, rl_values_opt_update, rl_critic_update, serialisation.  Forward
 * bit-identical to the C restatement; gradients and Fisher-vector products to f32 tolerance against the f64
 * restatement, both pinned by PyTorch vectors (tests/test_gpu_stacked.py, tests/test_oracle_stacked.py). */
enum { RL_CELL_GRU = 0, RL_CELL_LSTM = 1 };
int32_t rl_rnn_mlp_create(rl_engine *engine, int32_t cell, uint32_t in_dim, uint32_t hidden_size, uint32_t num_layers,
                          uint32_t mlp_hidden, uint32_t out_dim, rl_mlp **out);
/* The same with RnnBaseConfig::bias_init's choice of bias vectors (Option<Initializer>, seq/rnn/mod.rs:20-45,246-251):
 * bias = 0 builds the recurrent layers WITHOUT b_ih / b_hh (RnnWeights::has_biases = false) — flat order per layer
 * [W_ih, W_hh], then the MLP's kernel/bias pairs (the chain's MLP keeps its own LinearConfig); actor documents carry
 * `has_biases: false` and two tensors per layer; rl_mlp_init draws the weights only; rl_rnn_mlp_init_with takes
 * bias_init = NULL.  Such a module runs the lane-per-thread kernels (the gate rows start from zeros kept behind the
 * parameters); outputs are bit-identical to those of the same module with zero bias vectors, gradients are that module's
 * without the bias entries. */
int32_t rl_rnn_mlp_create_config(rl_engine *engine, int32_t cell, uint32_t in_dim, uint32_t hidden_size,
                                 uint32_t num_layers, int32_t bias, uint32_t mlp_hidden, uint32_t out_dim, rl_mlp **out);

/* ---------------------------------------------------------------------------------------------
 * Trajectory store (replaces VecBuffer + LazyHistoryFeatures: src/agents/buffers/vec.rs:15-143,
 * src/torch/agents/features.rs:48-213 — trajectories are born in HBM, no H2D per update). */
typedef enum {
  RL_TRAJ_OBS = 0,       /* f32 [obs_dim][T+1][n] (slot T = observation after the last step) */
  RL_TRAJ_ACTION = 1,    /* u8  [T][n] */
  RL_TRAJ_REWARD = 2,    /* f32 [T][n] */
  RL_TRAJ_FLAG = 3,      /* u8  [T][n] successor code */
  RL_TRAJ_TERM_OBS = 4,  /* f32 [obs_dim][T][n], valid where flag == INTERRUPT */
  RL_TRAJ_VALUES = 5,    /* f32 [T+1][n] critic values of OBS (after rl_gae) */
  RL_TRAJ_ADVANTAGES = 6,/* f32 [T][n] */
  RL_TRAJ_RETURNS = 7,   /* f32 [T][n] discounted reward-to-go */
  RL_TRAJ_TARGETS = 8    /* f32 [T][n] value targets of the last rl_values_opt_update (StepValueTarget::targets) */
} rl_traj_field;

/* obs_dim in 1..8 (feed-forward modules take 4 or 5; the others serve recurrent chains of that in_dim, written through
 * rl_traj_write: the envs of this library have 4 or 5 features) */
int32_t rl_traj_create(rl_engine *engine, uint64_t n_lanes, uint64_t horizon, uint32_t obs_dim, rl_traj **out);
int32_t rl_traj_destroy(rl_traj *traj);
int32_t rl_traj_field_bytes(const rl_traj *traj, int32_t field, uint64_t *bytes);
int32_t rl_traj_read(rl_traj *traj, int32_t field, void *host, uint64_t bytes);
/* upload (tests: feed oracle-generated trajectories / advantages to the update kernels) */
int32_t rl_traj_write(rl_traj *traj, int32_t field, const void *host, uint64_t bytes);

/* HOT LOOP A: `horizon` calls of Steps::step per lane (src/simulation/steps.rs:113-167) with
 * PolicyActor::act (src/torch/agents/policies/actor.rs:42-55) fused in: one persistent kernel. */
int32_t rl_rollout(rl_env *env, const rl_mlp *policy, rl_traj *traj);

/* critic.advantages + reward_to_go (src/torch/agents/critics/mod.rs:101-199; opt.rs:95-104).
 * gamma = min(max_discount_factor, env discount) as f32 (opt.rs:73), lambda f32 (critics/mod.rs:78). */
int32_t rl_gae(rl_traj *traj, const rl_mlp *critic, float gamma, float lambda);
/* SeqPacked::seq_packed of a recurrent module on a trajectory (modules/chain.rs:151-161): outputs for every step,
 * host [out_dim][T][n]; succ_out (may be NULL): outputs at the successor observation of every cut episode
 * (Interrupt / horizon), 0 elsewhere.  The recurrent state restarts at t = 0 and after every episode end. */
int32_t rl_seq_forward(rl_mlp *module, rl_traj *traj, float *out, float *succ_out);

/* ---------------------------------------------------------------------------------------------
 * TRPO policy update: Trpo::update (src/torch/agents/policies/trpo.rs:97-164) =
 * ConjugateGradientOptimizer::trust_region_backward_step + backtracking_line_search
 * (src/torch/optimizers/conjugate_gradient.rs:115-255). */
typedef struct {
  uint64_t iterations;      /* 10 */
  uint64_t max_backtracks;  /* 15 */
  double backtrack_ratio;   /* 0.8 */
  double hpv_reg_coeff;     /* 1e-5 */
  double max_policy_step_kl;/* 0.01 (TrpoConfig, trpo.rs:38) */
  int32_t accept_violation; /* false */
} rl_trpo_config;
int32_t rl_trpo_config_default(rl_trpo_config *cfg);

/* logged scalars, named as the reference logs them under `policy/` (conjugate_gradient.rs:164,200,219-226;
 * trpo.rs:119) */
typedef struct {
  double entropy, step_size, loss_initial, loss_final, constraint_val_final, step_scale;
  int64_t num_backtracks; /* -1: line search exhausted */
  int32_t status;         /* RL_OPT_* */
  int32_t cg_iterations;
} rl_trpo_stats;

int32_t rl_trpo_update(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg, rl_trpo_stats *stats);
/* pieces of the update exposed for parity tests (host vectors in flat parameter order) */
int32_t rl_policy_gradient(rl_mlp *policy, rl_traj *traj, float *grad_out, float *loss_out, float *entropy_out);
int32_t rl_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *v, float reg, float *out);
int32_t rl_policy_loss_kl(rl_mlp *policy, rl_traj *traj, const float *params0, float *loss_out, float *kl_out);

/* ---------------------------------------------------------------------------------------------
 * Critic update: ValuesOpt::update (src/torch/agents/critics/opt.rs:100-126) = n_backward_steps
 * (src/torch/agents/mod.rs:35-72) of full-batch MSE with COptimizer Adam (optimizers/coptimizer.rs:13-26,
 * 136-167; libtorch default eps 1e-8, amsgrad off). */
typedef struct {
  double learning_rate, beta1, beta2, weight_decay; /* AdamConfig (coptimizer.rs:136-156) */
  double eps;                                       /* libtorch default 1e-8 (not a field in the reference) */
} rl_adam_config;
int32_t rl_adam_config_default(rl_adam_config *cfg);
int32_t rl_adam_create(rl_mlp *module, const rl_adam_config *cfg, rl_adam **out);
int32_t rl_adam_destroy(rl_adam *opt);
/* one optimizer step from a host gradient (parity tests) */
int32_t rl_adam_step_host(rl_adam *opt, const float *grad);

typedef struct {
  double loss_first, loss_last; /* loss before the first / last optimisation step */
  uint64_t steps;
} rl_critic_stats;
/* targets = RL_TRAJ_RETURNS (StepValueTarget::RewardToGo, critics/mod.rs:203-229); `losses_out` (host,
 * may be NULL) receives the loss before each of the opt_steps steps */
int32_t rl_critic_update(rl_mlp *critic, rl_adam *opt, rl_traj *traj, uint64_t opt_steps, rl_critic_stats *stats,
                         float *losses_out);
int32_t rl_critic_gradient(rl_mlp *critic, rl_traj *traj, float *grad_out, float *loss_out);

/* ValuesOpt::update with its configuration (src/torch/agents/critics/opt.rs:13-37,100-126): the targets are computed
 * once, under no-grad, from the critic as it stands when the call begins — StepValueTarget::targets
 * (critics/mod.rs:203-229):
 *   RewardToGo  discounted_cumsum_from_end of the rewards                         (reward_to_go, critics/mod.rs:101-105)
 *   OneStepTd   r_t + discount_factor * V(successor of step t)                    (one_step_values, critics/mod.rs:139-150)
 *               with the masked extended values of eval_extended_state_values (:116-131): 0 after Terminate,
 *               V(successor observation) after Interrupt (and at the horizon cut); `scalar * tensor` then
 *               `tensor + tensor`, two f32 roundings
 * — then opt_steps_per_update x {mse_loss(V(obs), targets, Mean); backward; Adam}.  The targets stay readable as
 * RL_TRAJ_TARGETS; OneStepTd also refreshes RL_TRAJ_VALUES.  rl_critic_update above is the RewardToGo case on the
 * returns rl_gae left in RL_TRAJ_RETURNS. */
enum { RL_VALUE_TARGET_REWARD_TO_GO = 0, RL_VALUE_TARGET_ONE_STEP_TD = 1 }; /* StepValueTarget, critics/mod.rs:203-214 */
typedef struct {
  uint64_t opt_steps_per_update; /* 80 (opt.rs:46) */
  int32_t target;                /* RL_VALUE_TARGET_* ; default RewardToGo (critics/mod.rs:211-215) */
  float discount_factor;         /* max_discount_factor.min(env discount) as f32 (opt.rs:73); default 0.99 */
} rl_values_opt_config;
int32_t rl_values_opt_config_default(rl_values_opt_config *cfg);
int32_t rl_values_opt_update(rl_mlp *critic, rl_adam *opt, rl_traj *traj, const rl_values_opt_config *cfg,
                             rl_critic_stats *stats, float *losses_out /* may be NULL */);

/* ---------------------------------------------------------------------------------------------
 * The two updates of ActorCriticAgent::batch_update_slice (src/torch/agents/actor_critic.rs:176-211) after the
 * advantages — `policy.update(&features, advantages, ..)` (Trpo::update) and `critic.update(&features, ..)`
 * (ValuesOpt::update) — as ONE call.  They share no data beyond the trajectory: the advantages come from the critic as
 * it stood BEFORE either update (rl_gae), the critic's targets are the returns (or one-step TD values of the critic as
 * it stands when the call begins), so the reference's sequential order is not a dependence.  The engine therefore
 * enqueues the critic chain (targets + opt_steps_per_update x {gradient, reduce + Adam}) on a second HIP stream, with its
 * own slab rows / reduced vector / collective channel, and runs the TRPO chain on the main stream beside it: each
 * chain's single-workgroup bookkeeping launches (and, with several ranks, its latency-bound all-reduces) hide under
 * the other chain's compute.  Arithmetic and order INSIDE each chain are those of rl_trpo_update and
 * rl_values_opt_update: results are bit-identical to calling the two one after the other (tests/test_gpu_parity.py).
 * The chains run one after the other instead when the modules are not the fused 5-128-{2,1} MLPs, when kernel variant 1
 * or rl_engine_set_serial_update(engine, 1) / RELEARN_SERIAL_UPDATE=1 is selected, or when an RCCL job could not build
 * its second communicator.  Later calls on the engine are ordered after both chains.  Errors: as the two calls. */
int32_t rl_actor_critic_update(rl_mlp *policy, rl_mlp *critic, rl_adam *critic_opt, rl_traj *traj,
                               const rl_trpo_config *policy_cfg, const rl_values_opt_config *critic_cfg,
                               rl_trpo_stats *policy_stats, rl_critic_stats *critic_stats /* may be NULL */,
                               float *critic_losses_out /* may be NULL */);
/* The same update in two halves, so that the NEXT collection period can start under the critic chain.  The next
 * period's actors depend on the updated policy only (`agent.actor(mode)` snapshots the policy, src/agents/mod.rs:48-59;
 * train_parallel hands those actors to its workers, src/simulation/train.rs:98-151); the critic is next read when the
 * following batch's advantages are formed (actor_critic.rs:190-194).
 *   _begin   enqueues the critic chain on the auxiliary stream, runs the TRPO chain to its end on the main stream and
 *            returns `policy_stats` — the critic chain may still be in flight.
 *   between  rl_rollout(env, policy, OTHER trajectory) runs beside the critic chain.  Every other entry point of the
 *            engine (and a rollout into the SAME trajectory, or with the critic as its policy) is ordered behind the
 *            chain by a device-side wait: nothing can observe a half-updated critic, the calls just do not overlap.
 *   _finish  waits for the chain and returns its statistics.  Exactly one update may be pending per engine
 *            (RL_ERR_INVALID_ARGUMENT otherwise; _finish without _begin likewise).
 * rl_actor_critic_update is _begin followed by _finish.  Every number is the one the sequential calls produce: the
 * halves reorder launches across streams, never arithmetic (tests/test_gpu_parity.py).  When the chains cannot run
 * side by side (see above) _begin runs both to the end and _finish only hands the statistics over.
 * A NaN policy step (RL_ERR_OPT_NAN) is raised by _begin and ends the update (nothing stays pending); in the
 * side-by-side form the critic chain has then already advanced the critic and its optimiser state — the reference,
 * panicking inside policy.update, never reaches critic.update; in the sequential form neither does this library. */
int32_t rl_actor_critic_update_begin(rl_mlp *policy, rl_mlp *critic, rl_adam *critic_opt, rl_traj *traj,
                                     const rl_trpo_config *policy_cfg, const rl_values_opt_config *critic_cfg,
                                     rl_trpo_stats *policy_stats);
int32_t rl_actor_critic_update_finish(rl_traj *traj, rl_critic_stats *critic_stats /* may be NULL */,
                                      float *critic_losses_out /* may be NULL */);
/* 1: rl_actor_critic_update runs its two chains one after the other on the main stream (A/B runs, per-kernel timing) */
int32_t rl_engine_set_serial_update(rl_engine *engine, int32_t serial);

/* ---------------------------------------------------------------------------------------------
 * First-order policy updates and the critic-free advantage (the rest of the ActorCriticConfig matrix,
 * src/torch/agents/actor_critic.rs:20-45).
 *   Ppo::update        src/torch/agents/policies/ppo.rs:97-146  (clipped surrogate, n_backward_steps with Adam)
 *   Reinforce::update  src/torch/agents/policies/reinforce.rs:64-88 (one Adam step on -mean(log pi(a) * A))
 *   RewardToGo critic  src/torch/agents/critics/rtg.rs:28-33 (advantages = discounted reward-to-go, no update) */
typedef struct {
  uint64_t opt_steps_per_update; /* 10 */
  double clip_distance;          /* 0.2 */
} rl_ppo_config;
int32_t rl_ppo_config_default(rl_ppo_config *cfg); /* PpoConfig::default, ppo.rs:27-41 */
typedef struct {
  double entropy;               /* mean entropy at the initial parameters ("entropy", ppo.rs:114; reinforce.rs:84) */
  double loss_first, loss_last; /* surrogate loss before the first / last optimisation step */
  uint64_t steps;
} rl_policy_opt_stats;
int32_t rl_ppo_update(rl_mlp *policy, rl_adam *opt, rl_traj *traj, const rl_ppo_config *cfg,
                      rl_policy_opt_stats *stats, float *losses_out /* may be NULL */);
int32_t rl_reinforce_update(rl_mlp *policy, rl_adam *opt, rl_traj *traj, rl_policy_opt_stats *stats);
/* RL_TRAJ_ADVANTAGES = RL_TRAJ_RETURNS = reward-to-go (reward_to_go, critics/mod.rs:101-105) */
int32_t rl_reward_to_go(rl_traj *traj, float gamma);

/* ---------------------------------------------------------------------------------------------
 * DQN with the replay buffer resident in HBM (BASELINE.json configs[2]).
 *   DqnConfig / DqnAgent / DqnActor     src/torch/agents/dqn.rs:26-72,110-380
 *   ExplorationRateSchedule, DataCollectionSchedule   src/torch/agents/schedules.rs:7-69
 *   ReplayBuffer                        src/agents/buffers/replay.rs:11-127
 * Every lane is one ReplayBuffer of `buffer_capacity` steps (the reference has one per worker thread,
 * dqn.rs:129-130): step data lives in a per-lane ring `[capacity][n_lanes]`, the eviction rule "drop the whole
 * oldest episode when full" runs inside the collecting kernel.  Minibatch episodes are drawn with the agent's own
 * Prng exactly like dqn.rs:280-291 (cycle over the buffers, `Uniform::new(0, num_episodes)`, stop after the
 * episode that reaches `minibatch_steps`) — by a device kernel that evaluates the draws in parallel and keeps the
 * sequential semantics.  Targets (dqn.rs:300-309), the MSE loss on the taken action's value (dqn.rs:316-326) and
 * Adam (n_backward_steps, src/torch/agents/mod.rs:35-72) follow. */
enum { RL_DQN_TARGET_REWARD_TO_GO = 0, RL_DQN_TARGET_ONE_STEP_TD = 1 }; /* StepValueTarget, critics/mod.rs:203-214 */
enum { RL_SCHEDULE_CONSTANT = 0, RL_SCHEDULE_LINEAR_ANNEALED = 1 };     /* schedules.rs:7-15 */
enum { RL_COLLECT_CONSTANT = 0, RL_COLLECT_FIRST_REST = 1 };            /* schedules.rs:50-56 */
typedef struct {
  int32_t target;
  int32_t exploration_kind;
  double exploration_start, exploration_end; /* Constant(rate): rate = exploration_start */
  uint64_t exploration_period;
  uint64_t minibatch_steps;      /* dqn.rs:63 */
  uint64_t opt_steps_per_update; /* dqn.rs:64 */
  uint64_t buffer_capacity;      /* steps PER LANE (dqn.rs:129-130 "capacity of each individual buffer") */
  uint64_t episode_capacity;     /* episode-end slots per lane; 0 = buffer_capacity (never binds, like the deque) */
  int32_t update_kind;
  uint64_t update_first, update_rest; /* Constant(value): value = update_first */
  float discount_factor;         /* env.discount_factor() as f32 (dqn.rs:179) */
  uint32_t agent_key[8];         /* the agent's Prng seed words: Prng::from_rng(rng) in build_agent (dqn.rs:94) */
} rl_dqn_config;
/* DqnConfig::default (dqn.rs:57-72) with buffer_capacity left at 0 (the caller divides its step budget over the
 * lanes) and discount 0.99 (CartPole, src/envs/cartpole.rs:203-213) */
int32_t rl_dqn_config_default(rl_dqn_config *cfg);
/* `qnet` maps obs_dim -> n_actions; `opt` must have been created for `qnet`.  DqnConfig<MB> is generic over the module
 * (dqn.rs:26-39): any feed-forward module builds — the fused 5-128-2 shape on the fused kernels, other MlpConfigs on the
 * per-layer kernels; recurrent modules -> RL_ERR_BUILD_AGENT. */
int32_t rl_dqn_create(rl_env *env, rl_mlp *qnet, rl_adam *opt, const rl_dqn_config *cfg, rl_dqn **out);
int32_t rl_dqn_destroy(rl_dqn *dqn);
/* ExplorationRateSchedule::exploration_rate(global_steps, mode) (schedules.rs:35-45); training = 0 -> 0.0 */
int32_t rl_dqn_exploration_rate(const rl_dqn *dqn, int32_t training, double *rate_out);
/* DqnAgent::min_update_size (dqn.rs:207-209): the step bound of the next collection, summed over all lanes */
int32_t rl_dqn_min_update_size(const rl_dqn *dqn, uint64_t *min_steps_out, uint64_t *slack_steps_out);
typedef struct {
  double exploration_rate;
  uint64_t steps;          /* steps written (all lanes of this rank) */
  uint64_t episodes_ended; /* Terminate + Interrupt flags among them (incl. the horizon cut) */
} rl_dqn_collect_stats;
/* `horizon` env-actor steps on every lane with the epsilon-greedy actor (DqnActor::act, dqn.rs:360-379), written
 * to the replay rings.  RL_ERR_BUFFER_FULL when an episode outgrows a lane's capacity (replay.rs:93-94). */
int32_t rl_dqn_collect(rl_dqn *dqn, uint64_t horizon, rl_dqn_collect_stats *stats /* may be NULL */);
typedef struct {
  double loss_first, loss_last;
  uint64_t opt_steps;
  uint64_t global_steps;         /* after the update (dqn.rs:276) */
  uint64_t last_minibatch_steps, last_minibatch_episodes;
} rl_dqn_update_stats;
/* DqnAgent::batch_update (dqn.rs:263-337): opt_steps_per_update x {sample minibatch, targets, MSE, Adam} */
int32_t rl_dqn_update(rl_dqn *dqn, rl_dqn_update_stats *stats, float *losses_out /* may be NULL */);

/* Inspection of the replay store and of single minibatches (what the reference's tests reach through
 * ReplayBuffer::{episodes, num_steps, total_step_count}, replay.rs:52-86). */
enum { RL_REPLAY_HEAD = 0, RL_REPLAY_COUNT = 1, RL_REPLAY_EP_HEAD = 2, RL_REPLAY_EP_COUNT = 3, RL_REPLAY_TOTAL = 4,
       RL_REPLAY_EP_END = 5 /* u32 [E][n] */, RL_REPLAY_OBS = 6 /* f32 [D][C][n] */, RL_REPLAY_NEXT_OBS = 7,
       RL_REPLAY_ACTION = 8 /* u8 [C][n] */, RL_REPLAY_REWARD = 9 /* f32 [C][n] */, RL_REPLAY_FLAG = 10 /* u8 */,
       RL_REPLAY_ACTOR_POS = 11 /* u64 [n] */, RL_REPLAY_LAST_FLAGS = 12 /* u8 [T][n] of the last collection */ };
int32_t rl_dqn_replay_field_bytes(const rl_dqn *dqn, int32_t field, uint64_t *bytes);
int32_t rl_dqn_replay_read(rl_dqn *dqn, int32_t field, void *host, uint64_t bytes);
/* draw one minibatch (advances the agent Prng) and build its observation / action / target arrays;
 * `sequential` != 0 forces the one-thread sampler that the parallel one falls back to when a draw is rejected */
int32_t rl_dqn_minibatch_sample(rl_dqn *dqn, int32_t sequential, uint64_t *n_episodes_out, uint64_t *n_steps_out);
enum { RL_MB_EP_LANE = 0, RL_MB_EP_START = 1, RL_MB_EP_LEN = 2, RL_MB_EP_OFFSET = 3 /* u32 [n_episodes] */,
       RL_MB_OBS = 4 /* f32 [D][n_steps] */, RL_MB_ACTION = 5 /* u8 [n_steps] */, RL_MB_TARGET = 6 /* f32 */ };
int32_t rl_dqn_minibatch_read(rl_dqn *dqn, int32_t field, void *host, uint64_t bytes);
/* gradient of the MSE loss on the current minibatch (no parameter change) */
int32_t rl_dqn_minibatch_gradient(rl_dqn *dqn, float *grad_out, float *loss_out);
/* word position of the agent Prng (stream 0 of agent_key) */
int32_t rl_dqn_agent_rng_pos(rl_dqn *dqn, uint64_t *pos_out);

/* ---------------------------------------------------------------------------------------------
 * Actor serialisation in the reference's on-disk format: the CBOR document that
 * `serde_cbor::to_writer(file, &agent.actor(ActorMode::Evaluation))` writes (examples/cartpole-trpo.rs:71-76,
 * examples/cartpole-dqn.rs) and `serde_cbor::from_reader` loads for evaluation (:82-93).
 *   PolicyActor { observation_space, action_space, policy_module }      src/torch/agents/policies/actor.rs:10-14
 *   DqnActor { observation_space, action_space, action_value_fn, exploration_rate }  src/torch/agents/dqn.rs:340-346
 *   Mlp { layers: [Linear { kernel, bias }], activation, output_activation }         src/torch/modules/ff/mlp.rs:45-50
 *   Chain { first: RnnBase { weights { flat_weights, has_biases }, hidden_size, dropout, type_ }, second: Mlp,
 *           activation }                                  src/torch/modules/chain.rs:58-63, seq/rnn/mod.rs:90-99,186-191
 *   TensorDef { kind, shape, requires_grad, byte_order, data }                       src/torch/serialize.rs:62-81
 * observation / action spaces are those of the env the actor was trained on (NonEmptyFeatures<...>). */
enum { RL_ACTOR_POLICY = 0, RL_ACTOR_DQN = 1 };
/* Writes the document into `buf` (capacity `cap`) and its length into *len_out; with buf == NULL only the length is
 * returned.  `exploration_rate` is used by RL_ACTOR_DQN only (evaluation actors carry 0.0, schedules.rs:38). */
int32_t rl_actor_to_cbor(rl_env *env, rl_mlp *module, int32_t actor_kind, double exploration_rate, uint8_t *buf,
                         uint64_t cap, uint64_t *len_out);
/* Loads the module parameters of such a document (either actor kind) into `module`; the document's module must
 * have the handle's structure and shapes (RL_ERR_INVALID_ARGUMENT otherwise). */
int32_t rl_module_from_cbor(rl_mlp *module, const uint8_t *buf, uint64_t len);

/* The serialisation leaves on their own (host-only, no engine needed) — what every tensor and the action space of an
 * actor document are made of, and what the reference's serde_test token fixtures pin (src/torch/serialize.rs:187-352,
 * src/spaces/indexed_type.rs:409-422):
 *   TensorDef { kind: KindDef, shape: [i64], requires_grad: bool, byte_order: ByteOrder, data: bytes }
 *                                                    src/torch/serialize.rs:62-81; `impl From<&Tensor>` :83-106
 *   KindDef (remote definition of tch::Kind), variants in declaration order              src/torch/serialize.rs:12-31
 *   IndexedTypeSpace<T>: its only field is #[serde(skip)] -> a struct of length 0       src/spaces/indexed_type.rs:57-64
 * `data` holds the elements in row-major order, native (little-endian) byte order, element size of the kind.
 * Writers: buf == NULL returns the length only.  Readers refuse documents with another field order, unknown variants,
 * a non-native byte order (the reference panics there, serialize.rs:110-114) or a data length that does not match
 * shape x element size. */
typedef enum {
  RL_KIND_UINT8 = 0, RL_KIND_INT8 = 1, RL_KIND_INT16 = 2, RL_KIND_INT = 3, RL_KIND_INT64 = 4, RL_KIND_HALF = 5,
  RL_KIND_FLOAT = 6, RL_KIND_DOUBLE = 7, RL_KIND_COMPLEX_HALF = 8, RL_KIND_COMPLEX_FLOAT = 9,
  RL_KIND_COMPLEX_DOUBLE = 10, RL_KIND_BOOL = 11, RL_KIND_QINT8 = 12, RL_KIND_QUINT8 = 13, RL_KIND_QINT32 = 14,
  RL_KIND_BFLOAT16 = 15
} rl_tensor_kind;
int32_t rl_tensor_def_to_cbor(int32_t kind, const int64_t *shape, uint32_t rank, int32_t requires_grad,
                              const void *data, uint64_t data_bytes, uint8_t *buf, uint64_t cap, uint64_t *len_out);
/* shape_out receives at most shape_cap extents, data_out at most data_cap bytes (either may be NULL to query sizes
 * through rank_out / data_bytes_out) */
int32_t rl_tensor_def_from_cbor(const uint8_t *buf, uint64_t len, int32_t *kind_out, int64_t *shape_out,
                                uint32_t shape_cap, uint32_t *rank_out, int32_t *requires_grad_out, void *data_out,
                                uint64_t data_cap, uint64_t *data_bytes_out);
int32_t rl_indexed_type_space_to_cbor(uint8_t *buf, uint64_t cap, uint64_t *len_out);

/* ---------------------------------------------------------------------------------------------
 * CPU-only plumbing configuration (BASELINE.json configs[0]): examples/chain-tabular-q.rs — Chain
 * (src/envs/chain.rs:20-106) + epsilon-greedy tabular Q-learning (src/agents/tabular.rs:88-233) driven by
 * train_parallel (src/simulation/train.rs:68-186) with `n_threads` worker threads.  Runs on the host, like the
 * reference; q_values_out / counts_out are [5][2] row-major. */
int32_t rl_chain_tabular_q_train(uint64_t seed, uint64_t n_threads, uint64_t n_periods, uint64_t min_worker_steps,
                                 double exploration_rate, double *q_values_out, uint64_t *counts_out,
                                 uint64_t *total_steps_out);
/* evaluation run of examples/chain-tabular-q.rs:47-50: greedy actor, SimSeed::Root(seed), n_steps steps */
int32_t rl_chain_tabular_q_eval(const double *q_values, uint64_t seed, uint64_t n_steps, uint8_t *actions_out,
                                double *total_reward_out);

#ifdef __cplusplus
}
#endif
#endif /* RELEARN_HIP_H */
