// engine.hpp — internal structures behind the opaque handles of include/relearn_hip.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/relearn_hip.h"

struct RlError : std::runtime_error {
  int32_t code;
  RlError(int32_t c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define RL_HIP_CHECK(expr)                                                                          \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      throw RlError(RL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" + \
                                    std::to_string(__LINE__) + ")");                                \
  } while (0)

#define RL_REQUIRE(cond, msg)                                        \
  do {                                                               \
    if (!(cond)) throw RlError(RL_ERR_INVALID_ARGUMENT, (msg));      \
  } while (0)

// ---- device-side plain structs passed by value to kernels ---------------------------------------
struct CartPoleDev {
  double gravity, mass_pole, length_half_pole, friction_cart, friction_pole, time_step;
  double action_force, max_pos, max_angle;
  double total_weight, inv_total_mass, mass_length_pole;
  double init_low, init_scale;  // Uniform::new_inclusive(-0.05, 0.05)
  uint32_t key_env[8], key_actor[8];
  uint64_t lane_offset;
  uint32_t max_steps;
  int32_t limit_kind;
  uint32_t chain_size;   // RL_ENV_CHAIN / RL_ENV_MEMORY: number of states = one-hot width
  uint32_t mem_actions;  // RL_ENV_MEMORY: num_actions (0 selects Chain in the shared lane code)
  uint32_t bandit;       // RL_ENV_BANDIT: != 0; reward = bandit_r[action], every step terminates
  float bandit_r[2];
};

struct EnvStateDev {
  double *x, *xdot, *th, *thdot;  // [n]; RL_ENV_CHAIN / RL_ENV_MEMORY keep their state index in x, MEMORY also its
                                  // initial state in xdot and its env-stream word position in th (exact in f64)
  uint8_t *nv_pos;                // [n] cached_normal_velocity_is_positive
  uint32_t *steps_remaining;      // [n]
  uint32_t *reset_count;          // [n]
};

constexpr int RL_RANGE_WORDS = 2 * 64 * 32;  // = bt::RANGE_WORDS (bf16_tile.hpp)
// after the range words, one more line: the guard's veto words, one per chain (= bt::RANGE_ALLOC_WORDS, bt::GUARD_*)
constexpr int RL_RANGE_ALLOC_WORDS = RL_RANGE_WORDS + 32;
constexpr int RL_GUARD_POLICY = 0, RL_GUARD_CRITIC = 1;
struct TrajDev {
  float *obs;       // [D][T+1][n]
  uint8_t *action;  // [T][n]
  float *reward;    // [T][n]
  uint8_t *flag;    // [T][n]
  float *term_obs;  // [D][T][n]
  float *values;    // [T+1][n]
  float *adv;       // [T][n]
  float *rtg;       // [T][n]
  float *tgt;       // [T][n] regression targets of the critic update: rtg, or the one-step TD targets (StepValueTarget)
  uint32_t n, T, D;
  // numeric range of the observation planes, for the fused kernels' guard (bf16_tile.hpp range_guard): 64 slots of the
  // bits of the smallest non-zero |obs|, 64 of the bits of the largest |obs|, one 128-byte line each (RL_RANGE_WORDS)
  uint32_t *range;
  // the guard's error words (one per chain: RL_GUARD_POLICY, RL_GUARD_CRITIC), in HOST memory mapped into the device
  // (hipHostMalloc): a violation is one store across the bus, and the host reads the word after a synchronisation it makes
  // anyway — no copy, no extra round trip per update.  The same violation sets the chain's veto word in DEVICE memory
  // (range[RL_RANGE_WORDS + chain]), which the optimiser kernels of the call read: nothing of a refused update is applied
  uint32_t *range_err;
};

// the DQN replay store: every lane is one ReplayBuffer (src/agents/buffers/replay.rs:11-27), see replay.hpp for the
// bookkeeping.  Step data: one 32-byte record per step, `[N][C]` with the ring slot fastest — the steps of an episode are
// consecutive records (two runs where it wraps), so gathering a sampled episode reads whole lines.  (Feature planes
// `[D][C][N]` cost one memory line per 4-byte element there: 8 lines per step, and the minibatch builder ran at 4 TB/s
// of line traffic for 40 MB of samples.)
struct alignas(32) ReplayRec {
  float x[5];       // observation features (x[4] unused when D = 4)
  float reward;
  uint32_t af;      // action | successor code << 8
  uint32_t pad;
};
struct alignas(32) ReplayNext {
  float x[8];       // successor observation, meaningful where the step's successor code is INTERRUPT
};
struct ReplayDev {
  ReplayRec *rec;    // [N][C]
  ReplayNext *next;  // [N][C]
  uint32_t *head, *count, *ep_head, *ep_count, *total;  // [N] LaneRing fields
  uint32_t *ep_end;     // [E][N] absolute one-past-the-end step index of each stored episode
  uint64_t *actor_pos;  // [N] word position of the lane's actor stream
  int32_t *error;       // != 0: a lane hit WriteExperienceError::Full / an empty buffer was sampled
  uint32_t N, C, E, D;
};

struct DqnCountsDev {
  uint32_t n_eps, n_steps;
  int32_t error;
  uint32_t pad;
};

// scalar state of one TRPO update, lives in HBM so that the whole update needs no host round trip
struct TrpoStateDev {
  float rr;            // CG residual norm squared
  int32_t cg_done;     // CG converged (residual < tol): later HVP launches become no-ops
  int32_t cg_iters;
  float loss0;         // initial loss
  float entropy;
  double step_size;
  int32_t ls_accepted;
  int32_t ls_index;    // backtrack index that was accepted
  double ls_ratio;
  float ls_loss, ls_kl;  // last evaluated
  int32_t status;
  int32_t prev_saved;  // k_step_size has saved the parameters: a rollback has something to restore
};

// ---- host-side handle structs ----------------------------------------------------------------------
struct RcclApi;  // dlopen'ed entry points (comm.cpp)
struct LoopbackGroup;

constexpr int RL_IPC_MAX_RANKS = 16;

struct rl_engine {
  int device = -1;
  // `stream` is the stream every launcher enqueues on: the engine's main stream, or — inside an AuxChain scope
  // (abi_update.hip: the critic chain of rl_actor_critic_update) — the auxiliary one.  `chan` names the collective
  // channel that goes with it (0 main, 1 auxiliary): two chains in flight must not share a communicator / mailbox.
  hipStream_t stream = nullptr;
  hipStream_t main_stream = nullptr, aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int chan = 0;
  hipDeviceProp_t prop{};
  std::string last_error;
  // timing
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  bool profiling = false;
  double prof_ms[RL_K_CLASS_COUNT] = {0};
  uint64_t prof_launches[RL_K_CLASS_COUNT] = {0};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
  std::vector<hipEvent_t> prof_event_pool;
  // comm
  void *comm = nullptr;  // ncclComm_t
  void *comm_aux = nullptr;  // a second communicator over the same ranks for the auxiliary chain (NULL: none — two chains
                             // then run one after the other)
  struct LoopbackGroup *loopback = nullptr;  // in-process test collective (RELEARN_LOOPBACK_COMM=1)
  rl_host_allreduce_fn host_allreduce = nullptr;  // host-staged collective (rl_comm_init_host)
  void *host_allreduce_ctx = nullptr;
  std::vector<float> host_allreduce_buf;
  // peer-mailbox collective (comm_ipc.hip): own mailbox (fine-grained HBM), the peers' mailboxes mapped through IPC
  // handles, the sequence number of the last collective, a device word a timed-out wait sets
  float *ipc_box = nullptr;
  float *ipc_peer[RL_IPC_MAX_RANKS] = {nullptr};
  bool ipc_peer_local[RL_IPC_MAX_RANKS] = {false};  // the peer is an engine of THIS process: its mailbox by address
  int32_t *ipc_err = nullptr;
  uint32_t ipc_seq[2] = {0, 0};  // per channel: each has its own half of every mailbox
  uint64_t ipc_timeout_ticks = 0;  // bound of one mailbox wait, 100 MHz ticks (RELEARN_IPC_TIMEOUT_MS)
  int ipc_box_ranks = 0, ipc_rank_of_box = -1;
  bool ipc_active = false;
  // any collective between a reduction and its consumer?  (false: the two may be fused into one launch)
  bool has_collective() const {
    return comm != nullptr || loopback != nullptr || host_allreduce != nullptr || ipc_active;
  }
  int rank = 0, n_ranks = 1;
  // settings every rank must take the same way, agreed through the collective when it is installed (comm_agree, abi.hip):
  // RELEARN_SERIAL_UPDATE set in ANY rank's environment
  bool agreed_serial_env = false;
  // host pinned scratch for small readbacks
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  // child handles keep the engine alive: rl_engine_destroy with live children defers the teardown until the
  // last child is destroyed (hosts with garbage collectors release handles in arbitrary order)
  int64_t live_handles = 0;
  bool zombie = false;
  // counts the C-ABI calls made on this engine (guarded(), abi_internal.hpp): what a module's weight image is valid for
  // (rl_mlp::wimg_epoch) — state derived from the parameters is never trusted across entry points
  uint64_t call_epoch = 0;
  // counts the C-ABI calls on this engine that returned an error: an optimiser whose launches may have been vetoed on the
  // device re-reads its step count before its next step (rl_adam::error_epoch)
  uint64_t error_epoch = 0;
  // 0: best available kernels (MFMA v2 where the shape allows); 1: v1 reference kernels only
  int kernel_variant = 0;
  // rl_actor_critic_update: run the policy chain and the critic chain one after the other on the main stream (what the
  // separate entry points do) instead of side by side on two streams — for A/B measurements and per-kernel profiling
  bool serial_update = false;
  // rl_actor_critic_update_begin .. _finish: the critic chain of the update on `traj` is (or may still be) in flight on the
  // auxiliary stream.  `joined`: the main stream already waits for its end (any entry point but a rollout into another
  // trajectory orders itself behind it: engine_settle).  `collected`: the chains ran in turn inside _begin and the
  // critic's statistics are already in `stats` / `losses`.
  struct PendingUpdate {
    bool active = false, joined = false, collected = false;
    struct rl_traj *traj = nullptr;
    const struct rl_mlp *critic = nullptr;
    uint64_t steps = 0;
    rl_critic_stats stats{};
    std::vector<float> losses;
  } pending;
};

struct rl_env {
  rl_engine *eng;
  rl_env_config cfg;
  int kind = RL_ENV_CARTPOLE;
  CartPoleDev dev;
  EnvStateDev st;
  uint32_t D, A;
  uint64_t t_global = 0;
  // staging buffers for the standalone step API
  uint8_t *d_actions = nullptr, *d_flag = nullptr;
  float *d_reward = nullptr, *d_obs = nullptr, *d_term_obs = nullptr;
};

enum { RL_MODULE_MLP = 0, RL_MODULE_GRU_MLP = 1, RL_MODULE_LSTM_MLP = 2 };
// a recurrent chain module (Chain<Gru | Lstm, Mlp>)?
inline bool rl_module_is_recurrent(int kind) { return kind == RL_MODULE_GRU_MLP || kind == RL_MODULE_LSTM_MLP; }
inline uint64_t rl_module_gates(int kind) { return kind == RL_MODULE_LSTM_MLP ? 4 : 3; }  // RnnImpl::GATES_MULTIPLE

constexpr uint32_t RL_MLP_MAX_HIDDEN = 4;   // hidden layers of a general MlpConfig
constexpr uint32_t RL_MLP_MAX_WIDTH = 256;  // widest hidden layer
constexpr uint32_t RL_TRAJ_MAX_OBS_DIM = 8;  // observation features of a trajectory (the envs here have 4 or 5)
constexpr uint32_t RL_RNN_MAX_LAYERS = 4;   // RnnBaseConfig::num_layers of a recurrent chain

struct rl_mlp {
  rl_engine *eng;
  uint32_t in_dim, hidden, out_dim;  // GRU_MLP: hidden = the MLP's hidden width
  uint64_t P;
  float *d_params = nullptr;
  // the weight image of a 5 -> 128 -> A module for the fused update kernels (bf16_tile.hpp "the weight image"): allocated
  // on first use, current while wimg_epoch == eng->call_epoch (wimg_ensure / wimg_if_current / wimg_invalidate, kernels.hpp)
  mutable uint32_t *d_wimg = nullptr;
  mutable uint64_t wimg_epoch = 0;
  int kind = RL_MODULE_MLP;
  uint32_t gru_hidden = 0;
  // RnnBaseConfig::num_layers (seq/rnn/mod.rs:20-45,223-257).  > 1: stacked layers — flat order [W_ih, W_hh, b_ih, b_hh]
  // per layer (layer l > 0 reads the hidden output of layer l - 1), then the head; such a module runs the lane-per-thread
  // kernels of kernels_seq_stack.hip at its own widths (no twin).
  uint32_t rnn_layers = 1;
  // the lane-per-thread kernels run this recurrent module (stacked layers, more input features than the fused tile
  // kernels' five, or widths above their 128)
  // (or recurrent weights without bias vectors, RnnBaseConfig::bias_init = None: `has_bias` false — the gate rows then
  // start from 4 x 256 zeros kept behind the P parameters, like the bias-less MLP layers; the chain's MLP keeps its own)
  bool lane_kernels() const { return rnn_layers > 1 || in_dim > 5 || gru_hidden > 128 || hidden > 128 || !has_bias; }
  uint64_t rnn_layer_offset(uint32_t l) const {  // W_ih of layer l; l == rnn_layers: the head's W1
    const uint64_t GH = (kind == RL_MODULE_LSTM_MLP ? 4 : 3) * (uint64_t)gru_hidden, nb = has_bias ? 2 * GH : 0;
    if (l == 0) return 0;
    return GH * (in_dim + gru_hidden) + nb + (uint64_t)(l - 1) * (GH * 2 * gru_hidden + nb);
  }
  // MlpConfig::hidden_sizes (ff/mlp.rs:13-34).  `general`: a shape the fused single-hidden-layer kernels do not cover
  // (no hidden layer, several, or one wider than 128) — it runs the per-layer kernels of kernels_general.hip; then
  // `hidden` is 0, which keeps every fused launcher away.
  uint32_t n_hidden = 1;
  uint32_t widths[RL_MLP_MAX_HIDDEN] = {0, 0, 0, 0};
  bool general = false;
  // MlpConfig::activation / output_activation (rl_activation); the fused kernels are built for Relu / Identity
  int act = 1, out_act = 0;
  // Recurrent chains of other widths than the kernels' 5 -> 128 -> 128 (RnnBaseConfig::hidden_size, ChainConfig::
  // hidden_dim, MlpConfig::hidden_sizes = [h]; in_dim <= 5, widths <= 128): `exec` is the module the kernels run — the
  // same network embedded in the built shape, every padding weight and bias zero, so padded units stay exactly 0 (GRU:
  // h' = h / 2 from h = 0; LSTM: c' = c / 2, h' = tanh(0) / 2) and every real dot product only gains terms fma(0, 0, acc).
  // Its parameter image is refreshed from this module's flat vector before a pass (seq_exec, abi.hip); gradients and
  // tangents are gathered / scattered between the two layouts (launch_seq_pad / _unpad).  NULL: the module is the built
  // shape itself.
  rl_mlp *exec = nullptr;
  float *x_tmp = nullptr, *x_tan = nullptr;  // [exec->P]: gather scratch, padded tangent (zero outside the real entries)
  // layer l (0 .. n_hidden; the last one is the output layer): fan-in, fan-out, offset of its kernel in the flat
  // parameter vector ([W, b] per layer, the reference's order)
  uint32_t n_layers() const { return n_hidden + 1; }
  uint32_t fan_in(uint32_t l) const { return l == 0 ? in_dim : widths[l - 1]; }
  uint32_t fan_out(uint32_t l) const { return l == n_hidden ? out_dim : widths[l]; }
  uint64_t layer_offset(uint32_t l) const {
    uint64_t o = 0;
    for (uint32_t i = 0; i < l; ++i) o += (uint64_t)fan_in(i) * fan_out(i) + (has_bias ? fan_out(i) : 0);
    return o;
  }
  // LinearConfig::bias_init = None (ff/linear.rs:13-33): layers without a bias vector — the flat vector holds the
  // kernels only.  Every layer kernel starts its dot products from `bias`; such a module's layers read it from
  // RL_MLP_MAX_WIDTH zeros kept BEHIND the P parameters in the same allocation (never written), and its gradient passes
  // skip the bias columns.  Always on the per-layer path (`general`).
  bool has_bias = true;
  uint64_t bias_offset(uint32_t l) const { return has_bias ? layer_offset(l) + (uint64_t)fan_in(l) * fan_out(l) : P; }
  uint32_t hidden_units() const {
    uint32_t s = 0;
    for (uint32_t i = 0; i < n_hidden; ++i) s += widths[i];
    return s;
  }
};

// workspace of the recurrent path, attached to a trajectory on first use (tiles of 32 lanes)
struct SeqDev {
  float *act = nullptr;     // [T][tiles][7][128][32]: r, z, n, gh_n, h_prev, relu(h'), u
  float *dpre = nullptr;    // [T][tiles][5][128][32]: d pre_r, d pre_z, d pre_n, d pre_n * r, d u_pre
  float *out = nullptr;     // [2][T][n] module outputs (logits / values)
  float *succ = nullptr;    // [2][T][n] outputs at successor observations of cut episodes
  float *wg_slab = nullptr; // [chunks][P] partial weight gradients (f32)
  uint32_t tiles = 0, chunks = 0, blocks_per_chunk = 0;
  uint64_t P = 0;
  // stacked layers (kernels_seq_stack.hip; grow-only arrays, capacities in floats)
  struct Stack {
    float *st = nullptr;    // [10][L][H][n] per-lane states: two (h, c) sets in turn, the successor evaluation's, two tangent sets
    float *u = nullptr;     // [H2][n] the head's hidden units of the step at hand
    float *z = nullptr;     // [2][n] logits of a rollout step
    float *din = nullptr;   // [H][n] backward: gradient into the layer below, same step
    float *dst = nullptr;   // [2][L][H][n] backward: gradients carried to the step before (h, c)
    float *rec = nullptr;   // [L][8][H][B] activation record of the last training forward
    float *a1 = nullptr;    // [H][B] relu(top layer's output)
    float *ur = nullptr;    // [H2][B] the head's hidden units
    float *dg = nullptr;    // [L][4H][B] d loss / d pre-activations
    float *du = nullptr;    // [H2][B]
    uint64_t cap_st = 0, cap_u = 0, cap_din = 0, cap_dst = 0, cap_rec = 0, cap_a1 = 0, cap_ur = 0, cap_dg = 0, cap_du = 0;
    uint32_t wg_rows = 0, wg_chunk = 0;  // weight-gradient partials: slab rows, samples per row
  } stack;
};

// workspace of the general-MLP path (kernels_general.hip), attached to a trajectory on first use
struct GenDev {
  float *act = nullptr, *tact = nullptr;  // [hidden units][rows]: activations of the last forward, their tangents
  float *delta = nullptr;                 // [2][widest layer][rows]: backward deltas (ping-pong)
  float *z = nullptr, *tz = nullptr;      // [2][rows]: outputs and tangent outputs
  int32_t *no_interrupt = nullptr;        // != 0: the trajectory holds no Interrupt (the successor-value forward is skipped)
  uint64_t cap_act = 0, cap_tact = 0, cap_delta = 0, cap_z = 0, cap_tz = 0;
};

struct rl_adam {
  rl_engine *eng;  // kept separately: the module may be destroyed before its optimizer
  rl_mlp *mod;
  rl_adam_config cfg;
  float *d_m = nullptr, *d_v = nullptr;
  uint64_t *d_step = nullptr;
  uint64_t host_step = 0;  // == *d_step once the stream has drained (every Adam launch increments both)
  // ... unless a launch was vetoed on the device (a failed exchange, the range guard): the entry point then returned an
  // error, and the first step after an error on this engine re-reads the count (rl_engine::error_epoch, adam_next_step)
  uint64_t error_epoch = 0;
};

struct rl_traj {
  rl_engine *eng;
  TrajDev d;
  uint64_t B;  // T * n
  // update workspace
  float *lp0 = nullptr;     // [2][B]
  float *dz = nullptr;      // [2][B]
  double *slabA = nullptr;  // [nbA][Pmax] per-workgroup partial sums (f64)
  double *slabB = nullptr;  // [nbB][4]
  float *vec = nullptr;     // reduced vector [Pmax + 4]
  float *cg_x = nullptr, *cg_r = nullptr, *cg_p = nullptr, *prev_params = nullptr, *descent = nullptr;
  float *td = nullptr;      // [T][n] value targets of the last rl_values_opt_update (allocated on first use)
  float *losses = nullptr;  // [max critic steps]
  TrpoStateDev *trpo = nullptr;
  uint32_t nbA = 0, nbB = 0, nbV2 = 0, nbC = 0, Pmax = 0, max_losses = 0;
  uint32_t last_rows = 0;   // slab rows the last fused pass (launch_policy_v2 / launch_critic_step_v2) wrote
  uint64_t cap_slabA = 0, cap_slabB = 0;  // doubles allocated (traj_ensure_slabs grows them)
  // the auxiliary chain's own copies (rl_actor_critic_update; swapped in by AuxChain, allocated on first use)
  double *aux_slabA = nullptr, *aux_slabB = nullptr;
  float *aux_vec = nullptr;
  uint64_t aux_cap_slabA = 0, aux_cap_slabB = 0;
  uint32_t aux_last_rows = 0;
  // d.range[0..1] describe the current observation planes (false after anything rewrote them: traj_ensure_range, abi.hip);
  // `range_fixed`: the words are constants of the producer (the DQN minibatch workspace: CartPole-generated observations)
  bool range_valid = false, range_fixed = false;
  // the rollout that wrote the planes has reset d.range[0..1]: the value forward that follows (launch_values) measures
  // the range on the way and no pass of its own is needed
  bool range_reset = false;
  uint32_t *h_range_err = nullptr;  // host view of d.range_err (two words)
  // the range guard runs in the FIRST fused launch of each kind after an entry point began (range_check at the end of
  // every entry point re-arms it): the later launches of the same call — 79 of a critic update's 80 steps, the
  // Fisher-vector products and line-search candidates of a TRPO update — start from parameters the same call produced in
  // steps bounded by the learning rate / the KL constraint, and the wave that runs the guard starts its tiles ~1 us late
  bool guard_next_policy = true, guard_next_critic = true;
  // d.rtg holds the reward-to-go of the CURRENT reward / flag planes at discount factor `rtg_gamma`, written by the
  // lane scan of k_gae_scan (launch_gae) — the same recursion, operation for operation, as the RewardToGo value targets
  // (k_value_targets_rtg): a critic update that asks for exactly those regresses on the plane instead of scanning the
  // rewards a second time.  Cleared by whatever rewrites the planes (rollouts, rl_traj_write, the array-fed scans).
  bool rtg_scan_valid = false;
  float rtg_gamma = 0.0f;
  const float *last_targets = nullptr;  // where the last rl_values_opt_update's targets are: `td`, or the return plane
  uint32_t bwd_chunk = 0;   // samples per backward block
  SeqDev seq;
  GenDev gen;
};

struct rl_dqn {
  rl_engine *eng;
  rl_env *env;
  rl_mlp *qnet;
  rl_adam *opt;
  rl_dqn_config cfg;
  ReplayDev rp;
  uint64_t *d_agent_pos = nullptr;   // word position of the agent's Prng (stream 0 of cfg.agent_key)
  uint32_t *d_ep_lane = nullptr, *d_ep_start = nullptr, *d_ep_len = nullptr, *d_ep_off = nullptr;  // [max_eps]
  DqnCountsDev *d_counts = nullptr;
  uint8_t *d_flags = nullptr;        // [T_cap][N] successor codes of the last collection
  uint64_t flags_cap = 0, last_horizon = 0;
  uint32_t max_eps = 0;
  uint64_t max_steps_mb = 0;         // sample capacity of the minibatch workspace
  rl_traj *mb = nullptr;             // minibatch workspace: T = 1, n = current minibatch size
  // reward-to-go updates: the compact samples of ALL minibatches of an update, built in one launch (first update)
  float *all_obs = nullptr, *all_target = nullptr;  // [K][D][2 * max_steps_mb], [K][max_steps_mb]
  uint8_t *all_action = nullptr, *all_flag = nullptr;  // [K][max_steps_mb]; successor codes (one-step TD only)
  bool td_in_kernel = false;  // the workspace holds rewards / successor codes / successor observations: the gradient
                              // kernel forms the one-step TD targets itself
  // ... drawn on a second stream in growing chunks while the main stream already trains on the earlier ones
  hipStream_t draw_stream = nullptr;
  std::vector<hipEvent_t> draw_events;
  hipEvent_t main_event = nullptr;
  DqnCountsDev *h_counts = nullptr;  // pinned, [K]
  float *d_q = nullptr, *d_q_next = nullptr;  // module outputs of a collection step [2][N] / at a minibatch's successor
                                              // observations (action-value modules on the per-layer kernels)
  uint64_t cap_q_next = 0;
  float *snap = nullptr;             // parameters + Adam moments + step count as of the start of a pipelined update
  // ... `mb` then points into them (the last minibatch stays readable); its own arrays, for the one-at-a-time builder:
  float *own_obs = nullptr, *own_target = nullptr;
  uint8_t *own_action = nullptr, *own_flag = nullptr;
  uint64_t global_steps = 0;         // as of the last update (dqn.rs:276)
  uint64_t steps_per_lane = 0;       // collected so far
  uint32_t last_n_eps = 0, last_n_steps = 0, last_batch_index = 0;
  uint64_t last_total_steps = 0;     // minibatch size summed over ranks
};

// ---- profiling helper: wraps a launch with events when enabled ---------------------------------------
struct ProfScope {
  rl_engine *e;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(rl_engine *eng, int c);
  ~ProfScope();
};

// abi.hip
void rl_allreduce_sum_f32(rl_engine *e, float *d_buf, size_t count);
void comm_agree(rl_engine *e);  // settings all ranks must share, agreed through the collective just installed
// comm_ipc.hip
bool ipc_allreduce_fits(const rl_engine *e, size_t count);
void ipc_allreduce(rl_engine *e, float *d_buf, size_t count);
void ipc_check(rl_engine *e);  // throws RL_ERR_COMM when a wait of an earlier collective timed out
void ipc_teardown(rl_engine *e);
