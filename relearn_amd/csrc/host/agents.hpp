// agents.hpp — C++ host side above the C ABI: the reference's trait surface for the accelerated path, same names,
// argument meaning and error behaviour (the reference is Rust; this image has no Rust toolchain, INTEGRATION.md shows
// the equivalent `extern "C"` binding a relearn maintainer adds).  Header-only; link against librelearn_hip.so.
//
//   EnvStructure / Environment     src/envs/mod.rs:76-127,165-193     -> CartPoleLanes, ChainLanes (vectorised)
//   BuildModule, MlpConfig, GruMlpConfig   src/torch/modules/mod.rs:14, ff/mlp.rs:13-34   -> MlpConfig, GruMlpConfig
//   BuildAgent / Agent / Actor / BatchUpdate / ActorMode   src/agents/mod.rs:48-59,101-114,144,167-215
//   Policy, Critic, Optimizer, TrustRegionOptimizer (abstract; Trpo / Ppo / Reinforce, ValuesOpt / RewardToGo,
//   AdamOptimizer, ConjugateGradientOptimizer implement them)   src/torch/agents/policies/mod.rs:21-53,
//   critics/mod.rs:20-99, src/torch/optimizers/mod.rs:25-92
//   ActorCriticConfig / ActorCriticAgent   src/torch/agents/actor_critic.rs:20-136,176-211
//   TrpoConfig / PpoConfig / ReinforceConfig   src/torch/agents/policies/{trpo,ppo,reinforce}.rs
//   ValuesOptConfig / RewardToGoConfig       src/torch/agents/critics/{opt,rtg}.rs
//   DqnConfig / DqnAgent                      src/torch/agents/dqn.rs:26-140,263-337
//   StatsLogger, ChunkLogger, ByCounter / ByTime, Display / TensorBoard back ends   src/logging/ -> logging.hpp
//   train_parallel's period loop and its metric names   src/simulation/train.rs:68-186
#pragma once
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <optional>
#include <vector>

#include "../../../include/relearn_hip.h"
#include "../../../include/rl_detmath.h"
#include "logging.hpp"
#include "prng.hpp"

namespace relearn {

// ---------------------------------------------------------------- errors (the reference's error enums)
struct Error : std::runtime_error {
  int32_t code;
  Error(int32_t c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct BuildAgentError : Error { using Error::Error; };        // src/agents/mod.rs:219-226
struct BuildEnvError : Error { using Error::Error; };          // src/envs/builders.rs:62-66
struct WriteExperienceFull : Error { using Error::Error; };    // WriteExperienceError::Full, buffers/mod.rs:225-228
struct OptimizerNanError : Error { using Error::Error; };      // panic in Trpo::update, policies/trpo.rs:154-162
struct NoDeviceError : Error { using Error::Error; };

inline void check(int32_t rc, const rl_engine *eng = nullptr) {
  if (rc == RL_OK) return;
  const char *m = rl_last_error(eng);
  std::string msg = m ? m : "";
  switch (rc) {
    case RL_ERR_BUILD_AGENT: throw BuildAgentError(rc, msg);
    case RL_ERR_BUILD_ENV: throw BuildEnvError(rc, msg);
    case RL_ERR_BUFFER_FULL: throw WriteExperienceFull(rc, msg);
    case RL_ERR_OPT_NAN: throw OptimizerNanError(rc, msg);
    case RL_ERR_NO_DEVICE: throw NoDeviceError(rc, msg);
    default: throw Error(rc, msg);
  }
}

// ---------------------------------------------------------------- engine + handles
class Engine {
 public:
  explicit Engine(int device = 0) { check(rl_engine_create(device, &h_)); }
  ~Engine() { rl_engine_destroy(h_); }
  Engine(const Engine &) = delete;
  Engine &operator=(const Engine &) = delete;
  rl_engine *handle() const { return h_; }
  void sync() { check(rl_engine_sync(h_), h_); }

 private:
  rl_engine *h_ = nullptr;
};

enum class ActorMode { Training, Evaluation };  // src/agents/mod.rs:144
enum class StepLimit { None = RL_LIMIT_NONE, Latent = RL_LIMIT_LATENT, Visible = RL_LIMIT_VISIBLE };

// EnvStructure of N vectorised lanes (the spaces are those of the wrapped reference env)
class EnvLanes {
 public:
  ~EnvLanes() { rl_env_destroy(h_); }
  EnvLanes(const EnvLanes &) = delete;
  EnvLanes &operator=(const EnvLanes &) = delete;
  rl_env *handle() const { return h_; }
  Engine &engine() const { return eng_; }
  uint64_t num_lanes() const { return cfg_.n_lanes; }
  uint32_t num_observation_features() const { return obs_dim_; }
  uint32_t num_actions() const { return n_actions_; }
  double discount_factor() const { return discount_; }
  // Environment::{initial_state, observe, step} for every lane (src/envs/mod.rs:76-127)
  void initial_state() { check(rl_env_reset(h_), eng_.handle()); }
  std::vector<float> observe() {
    std::vector<float> obs((size_t)obs_dim_ * cfg_.n_lanes);
    check(rl_env_observe(h_, obs.data()), eng_.handle());
    return obs;
  }
  struct StepResult {
    std::vector<float> reward, next_obs, interrupt_obs;
    std::vector<uint8_t> successor;  // RL_SUCC_CONTINUE / TERMINATE / INTERRUPT
  };
  StepResult step(const std::vector<uint8_t> &actions) {
    if (actions.size() != cfg_.n_lanes) throw Error(RL_ERR_INVALID_ARGUMENT, "one action per lane");
    StepResult r;
    r.reward.resize(cfg_.n_lanes);
    r.successor.resize(cfg_.n_lanes);
    r.next_obs.resize((size_t)obs_dim_ * cfg_.n_lanes);
    r.interrupt_obs.resize((size_t)obs_dim_ * cfg_.n_lanes);
    check(rl_env_step(h_, actions.data(), r.reward.data(), r.successor.data(), r.next_obs.data(),
                      r.interrupt_obs.data()), eng_.handle());
    return r;
  }

 protected:
  EnvLanes(Engine &eng, const rl_env_config &cfg, double discount) : eng_(eng), cfg_(cfg), discount_(discount) {
    check(rl_env_create(eng.handle(), &cfg_, &h_), eng.handle());
    check(rl_env_dims(h_, &obs_dim_, &n_actions_), eng.handle());
  }
  Engine &eng_;
  rl_env_config cfg_;
  rl_env *h_ = nullptr;
  uint32_t obs_dim_ = 0, n_actions_ = 0;
  double discount_;
};

// `CartPole::default().wrap(VisibleStepLimit::new(max_steps))` x n_lanes (src/envs/cartpole.rs, wrappers/step_limit.rs)
class CartPoleLanes : public EnvLanes {
 public:
  CartPoleLanes(Engine &eng, uint64_t n_lanes, uint64_t max_steps = 500, StepLimit limit = StepLimit::Visible,
                uint64_t seed_env = 0, uint64_t seed_actor = 1, uint64_t lane_offset = 0)
      : EnvLanes(eng, config(n_lanes, max_steps, limit, seed_env, seed_actor, lane_offset), 0.99) {}

 private:
  static rl_env_config config(uint64_t n, uint64_t max_steps, StepLimit limit, uint64_t se, uint64_t sa, uint64_t off) {
    rl_env_config c{};
    c.kind = RL_ENV_CARTPOLE;
    c.limit_kind = (int32_t)limit;
    c.max_steps = max_steps;
    c.n_lanes = n;
    c.lane_offset = off;
    c.seed_env = se;
    c.seed_actor = sa;
    check(rl_cartpole_params_default(&c.cartpole));
    return c;
  }
};

// `Chain::default().wrap(LatentStepLimit::new(max_steps))` x n_lanes (src/envs/chain.rs)
class ChainLanes : public EnvLanes {
 public:
  ChainLanes(Engine &eng, uint64_t n_lanes, uint64_t max_steps = 100, StepLimit limit = StepLimit::Latent,
             uint64_t seed_env = 0, uint64_t seed_actor = 1, uint64_t lane_offset = 0)
      : EnvLanes(eng, config(n_lanes, max_steps, limit, seed_env, seed_actor, lane_offset), 0.95) {}

 private:
  static rl_env_config config(uint64_t n, uint64_t max_steps, StepLimit limit, uint64_t se, uint64_t sa, uint64_t off) {
    rl_env_config c{};
    c.kind = RL_ENV_CHAIN;
    c.limit_kind = (int32_t)limit;
    c.max_steps = max_steps;
    c.n_lanes = n;
    c.lane_offset = off;
    c.seed_env = se;
    c.seed_actor = sa;
    check(rl_cartpole_params_default(&c.cartpole));
    c.chain_size = 5;
    return c;
  }
};

// `MemoryGame::new(2, 3)` x n_lanes, optionally under a step limit (src/envs/memory.rs); discount factor 1.0
class MemoryGameLanes : public EnvLanes {
 public:
  MemoryGameLanes(Engine &eng, uint64_t n_lanes, uint64_t num_actions = 2, uint64_t history_len = 3,
                  uint64_t max_steps = 0, StepLimit limit = StepLimit::None, uint64_t seed_env = 0,
                  uint64_t seed_actor = 1, uint64_t lane_offset = 0)
      : EnvLanes(eng, config(n_lanes, num_actions, history_len, max_steps, limit, seed_env, seed_actor, lane_offset),
                 1.0) {}

 private:
  static rl_env_config config(uint64_t n, uint64_t na, uint64_t hl, uint64_t max_steps, StepLimit limit, uint64_t se,
                              uint64_t sa, uint64_t off) {
    rl_env_config c{};
    c.kind = RL_ENV_MEMORY;
    c.limit_kind = (int32_t)limit;
    c.max_steps = max_steps;
    c.n_lanes = n;
    c.lane_offset = off;
    c.seed_env = se;
    c.seed_actor = sa;
    check(rl_cartpole_params_default(&c.cartpole));
    c.memory_num_actions = na;
    c.memory_history_len = hl;
    return c;
  }
};

// `DeterministicBandit::from_values([v0, v1])` x n_lanes (src/envs/bandits.rs:109-116): every step is an episode,
// reward = the chosen arm's value, discount factor 1.0 (bandits.rs:52-54).  The environment of the reference's universal
// agent test (`train_deterministic_bandit`, src/agents/testing.rs:14-64).
class DeterministicBanditLanes : public EnvLanes {
 public:
  DeterministicBanditLanes(Engine &eng, uint64_t n_lanes, double value0 = 0.0, double value1 = 1.0,
                           uint64_t seed_env = 0, uint64_t seed_actor = 1, uint64_t lane_offset = 0)
      : EnvLanes(eng, config(n_lanes, value0, value1, seed_env, seed_actor, lane_offset), 1.0) {}

 private:
  static rl_env_config config(uint64_t n, double v0, double v1, uint64_t se, uint64_t sa, uint64_t off) {
    rl_env_config c{};
    c.kind = RL_ENV_BANDIT;
    c.limit_kind = RL_LIMIT_NONE;
    c.n_lanes = n;
    c.lane_offset = off;
    c.seed_env = se;
    c.seed_actor = sa;
    check(rl_cartpole_params_default(&c.cartpole));
    c.bandit_values[0] = v0;
    c.bandit_values[1] = v1;
    return c;
  }
};

// ---------------------------------------------------------------- modules (BuildModule)
class Module {
 public:
  ~Module() { rl_mlp_destroy(h_); }
  Module(const Module &) = delete;
  Module &operator=(const Module &) = delete;
  rl_mlp *handle() const { return h_; }
  uint64_t num_parameters() const { return n_; }
  std::vector<float> parameters() const {  // trainable_variables() flattened, reference order
    std::vector<float> p(n_);
    check(rl_params_get(h_, p.data(), n_), eng_.handle());
    return p;
  }
  void set_parameters(const std::vector<float> &p) { check(rl_params_set(h_, p.data(), p.size()), eng_.handle()); }

 protected:
  Module(Engine &eng, rl_mlp *h) : eng_(eng), h_(h) { check(rl_mlp_num_params(h_, &n_), eng.handle()); }
  friend struct MlpConfig;
  friend struct GruMlpConfig;
  friend struct ChainLstmMlpConfig;
  Engine &eng_;
  rl_mlp *h_;
  uint64_t n_ = 0;
};

enum class Activation { Identity = RL_ACT_IDENTITY, Relu = RL_ACT_RELU, Sigmoid = RL_ACT_SIGMOID, Tanh = RL_ACT_TANH };

// Initializer / VarianceScale (src/torch/initializers.rs:8-64): the C ABI's rl_initializer with the reference's names
struct Initializer {
  rl_initializer raw;
  static Initializer Zeros() { return {{RL_INIT_ZEROS, RL_SCALE_FAN_AVG, 0.0}}; }
  static Initializer Constant(double v) { return {{RL_INIT_CONSTANT, RL_SCALE_FAN_AVG, v}}; }
  static Initializer Uniform(int32_t scale = RL_SCALE_FAN_AVG, double variance = 0.0) { return {{RL_INIT_UNIFORM, scale, variance}}; }
  static Initializer Normal(int32_t scale = RL_SCALE_FAN_AVG, double variance = 0.0) { return {{RL_INIT_NORMAL, scale, variance}}; }
  static Initializer Orthogonal() { return {{RL_INIT_ORTHOGONAL, RL_SCALE_FAN_AVG, 0.0}}; }
};

struct LinearConfig {  // LinearConfig { kernel_init, bias_init } (ff/linear.rs:13-33); default: Uniform(FanAvg) both
  Initializer kernel_init = Initializer::Uniform();
  std::optional<Initializer> bias_init = Initializer::Uniform();  // nullopt: layers without a bias vector
};

struct MlpConfig {  // MlpConfig { hidden_sizes, activation, output_activation, linear_config } (ff/mlp.rs:13-34)
  std::vector<uint32_t> hidden_sizes{128};  // MlpConfig::default; any list of up to four widths <= 256 builds
  Activation activation = Activation::Relu;             // between the hidden layers (ff/activation.rs:11-27)
  Activation output_activation = Activation::Identity;  // on the output
  LinearConfig linear_config;
  // width of the single hidden layer (the recurrent chains take exactly one)
  uint32_t single_hidden_size() const {
    if (hidden_sizes.size() != 1) throw BuildAgentError(RL_ERR_BUILD_AGENT, "the chain's MLP takes one hidden layer");
    return hidden_sizes[0];
  }
  std::unique_ptr<Module> build_module(Engine &eng, uint32_t in_dim, uint32_t out_dim, uint64_t seed) const {
    rl_mlp *h = nullptr;
    check(rl_mlp_create_config(eng.handle(), in_dim, hidden_sizes.data(), (uint32_t)hidden_sizes.size(), out_dim,
                               (int32_t)activation, (int32_t)output_activation, linear_config.bias_init ? 1 : 0, &h),
          eng.handle());
    std::unique_ptr<Module> m(new Module(eng, h));
    check(rl_mlp_init_with(h, seed, &linear_config.kernel_init.raw,
                           linear_config.bias_init ? &linear_config.bias_init->raw : nullptr),
          eng.handle());
    return m;
  }
};

// RnnBaseConfig (seq/rnn/mod.rs:20-45): the layer count and the three initializers
struct RnnBaseConfig {
  uint32_t num_layers = 1;  // stacked layers: 1..4
  Initializer input_weights_init = Initializer::Uniform();     // RnnBaseConfig::default: Uniform(FanAvg)
  Initializer hidden_weights_init = Initializer::Orthogonal();
  std::optional<Initializer> bias_init = Initializer::Zeros();  // nullopt: recurrent layers without bias vectors
};
using GruConfig = RnnBaseConfig;
using LstmConfig = RnnBaseConfig;

struct GruMlpConfig {  // ChainConfig<GruConfig, MlpConfig>::default (modules/mod.rs:14, chain.rs:19-32)
  GruConfig first_config;
  uint32_t hidden_dim = 128;
  MlpConfig second_config;
  std::unique_ptr<Module> build_module(Engine &eng, uint32_t in_dim, uint32_t out_dim, uint64_t seed) const {
    rl_mlp *h = nullptr;
    check(rl_rnn_mlp_create_config(eng.handle(), RL_CELL_GRU, in_dim, hidden_dim, first_config.num_layers,
                                   first_config.bias_init ? 1 : 0, second_config.single_hidden_size(), out_dim, &h),
          eng.handle());
    std::unique_ptr<Module> m(new Module(eng, h));
    const LinearConfig &lc = second_config.linear_config;
    check(rl_rnn_mlp_init_with(h, seed, &first_config.input_weights_init.raw, &first_config.hidden_weights_init.raw,
                               first_config.bias_init ? &first_config.bias_init->raw : nullptr, &lc.kernel_init.raw,
                               lc.bias_init ? &lc.bias_init->raw : nullptr),
          eng.handle());
    return m;
  }
};

// The reference defines `pub type LstmMlpConfig = ChainConfig<GruConfig, MlpConfig>` (modules/mod.rs:15): its
// "LSTM-MLP" configuration builds the GRU chain.  Kept as is, so that a configuration written against the reference
// builds the same network here.
using LstmMlpConfig = GruMlpConfig;

// ChainConfig<LstmConfig, MlpConfig>::default (chain.rs:19-32 with Lstm = RnnBase<LstmImpl>, seq/rnn/lstm.rs:12-51): what a
// user of the reference writes out to get an actual LSTM chain.
struct ChainLstmMlpConfig {
  LstmConfig first_config;
  uint32_t hidden_dim = 128;
  MlpConfig second_config;
  std::unique_ptr<Module> build_module(Engine &eng, uint32_t in_dim, uint32_t out_dim, uint64_t seed) const {
    rl_mlp *h = nullptr;
    check(rl_rnn_mlp_create_config(eng.handle(), RL_CELL_LSTM, in_dim, hidden_dim, first_config.num_layers,
                                   first_config.bias_init ? 1 : 0, second_config.single_hidden_size(), out_dim, &h),
          eng.handle());
    std::unique_ptr<Module> m(new Module(eng, h));
    const LinearConfig &lc = second_config.linear_config;
    check(rl_rnn_mlp_init_with(h, seed, &first_config.input_weights_init.raw, &first_config.hidden_weights_init.raw,
                               first_config.bias_init ? &first_config.bias_init->raw : nullptr, &lc.kernel_init.raw,
                               lc.bias_init ? &lc.bias_init->raw : nullptr),
          eng.handle());
    return m;
  }
};

// ---------------------------------------------------------------- optimizers (src/torch/optimizers/mod.rs:25-92)
// `Optimizer` (first-order: backward_step) and `TrustRegionOptimizer` (trust_region_backward_step) are the reference's
// two optimizer traits; the update loops that drive them live in the library, so an implementation here hands the
// library what it needs: the device optimizer state, or the trust-region solver's settings.
class Optimizer {
 public:
  virtual ~Optimizer() = default;
  virtual rl_adam *handle() const = 0;  // device state of the rule (moments, step count)
};
class TrustRegionOptimizer {
 public:
  virtual ~TrustRegionOptimizer() = default;
  // fill the solver's part of a trust-region step (the caller sets the constraint bound)
  virtual void configure(rl_trpo_config &c) const = 0;
};

class AdamOptimizer final : public Optimizer {  // COptimizer over torch's Adam (optimizers/coptimizer.rs:16-110)
 public:
  AdamOptimizer(Module &m, Engine &eng, const rl_adam_config &c) { check(rl_adam_create(m.handle(), &c, &h_), eng.handle()); }
  ~AdamOptimizer() override { rl_adam_destroy(h_); }
  AdamOptimizer(const AdamOptimizer &) = delete;
  AdamOptimizer &operator=(const AdamOptimizer &) = delete;
  rl_adam *handle() const override { return h_; }

 private:
  rl_adam *h_ = nullptr;
};

struct AdamConfig {  // optimizers/coptimizer.rs:136-156
  double learning_rate = 1e-3, beta1 = 0.9, beta2 = 0.999, weight_decay = 0.0;
  std::unique_ptr<Optimizer> build_optimizer(Module &m, Engine &eng) const {  // BuildOptimizer::build_optimizer
    rl_adam_config c;
    check(rl_adam_config_default(&c));
    c.learning_rate = learning_rate;
    c.beta1 = beta1;
    c.beta2 = beta2;
    c.weight_decay = weight_decay;
    return std::unique_ptr<Optimizer>(new AdamOptimizer(m, eng, c));
  }
};

// ---------------------------------------------------------------- device-resident history (VecBuffer's stand-in)
class DeviceHistory {
 public:
  DeviceHistory(Engine &eng, uint64_t n_lanes, uint64_t horizon, uint32_t obs_dim) : n_(n_lanes), T_(horizon) {
    check(rl_traj_create(eng.handle(), n_lanes, horizon, obs_dim, &h_), eng.handle());
  }
  ~DeviceHistory() { rl_traj_destroy(h_); }
  DeviceHistory(const DeviceHistory &) = delete;
  DeviceHistory &operator=(const DeviceHistory &) = delete;
  rl_traj *handle() const { return h_; }
  uint64_t num_steps() const { return n_ * T_; }
  uint64_t num_lanes() const { return n_; }
  uint64_t horizon() const { return T_; }
  std::vector<uint8_t> successors() const {
    std::vector<uint8_t> f(n_ * T_);
    check(rl_traj_read(h_, RL_TRAJ_FLAG, f.data(), f.size()));
    return f;
  }

 private:
  rl_traj *h_ = nullptr;
  uint64_t n_, T_;
};

// ---------------------------------------------------------------- policies and critics
struct ConjugateGradientOptimizerConfig;
class ConjugateGradientOptimizer final : public TrustRegionOptimizer {  // conjugate_gradient.rs:67-260
 public:
  ConjugateGradientOptimizer(uint64_t iterations, uint64_t max_backtracks, double backtrack_ratio, double hpv_reg_coeff,
                             bool accept_violation)
      : iterations_(iterations), max_backtracks_(max_backtracks), backtrack_ratio_(backtrack_ratio),
        hpv_reg_coeff_(hpv_reg_coeff), accept_violation_(accept_violation) {}
  void configure(rl_trpo_config &c) const override {
    c.iterations = iterations_;
    c.max_backtracks = max_backtracks_;
    c.backtrack_ratio = backtrack_ratio_;
    c.hpv_reg_coeff = hpv_reg_coeff_;
    c.accept_violation = accept_violation_ ? 1 : 0;
  }

 private:
  uint64_t iterations_, max_backtracks_;
  double backtrack_ratio_, hpv_reg_coeff_;
  bool accept_violation_;
};
struct ConjugateGradientOptimizerConfig {  // conjugate_gradient.rs:41-65
  uint64_t iterations = 10, max_backtracks = 15;
  double backtrack_ratio = 0.8, hpv_reg_coeff = 1e-5;
  bool accept_violation = false;
  std::unique_ptr<TrustRegionOptimizer> build_optimizer() const {
    return std::unique_ptr<TrustRegionOptimizer>(
        new ConjugateGradientOptimizer(iterations, max_backtracks, backtrack_ratio, hpv_reg_coeff, accept_violation));
  }
};

enum class PolicyKind { Trpo, Ppo, Reinforce };

// Policy (src/torch/agents/policies/mod.rs:21-53): a module and the rule that improves it from a history whose
// advantages are in place.  `update` logs what the reference's implementation logs, under the caller's scope.
class Policy {
 public:
  virtual ~Policy() = default;
  virtual PolicyKind kind() const = 0;
  virtual void update(DeviceHistory &history, StatsLogger &logger) = 0;
  Module &module() { return *module_; }  // AsModule::as_module
  int32_t last_status() const { return last_status_; }  // OptimizerStepError of the last update, RL_OPT_OK otherwise

 protected:
  Policy(Engine &eng, std::unique_ptr<Module> m) : eng_(eng), module_(std::move(m)) {}
  Engine &eng_;
  std::unique_ptr<Module> module_;
  int32_t last_status_ = RL_OPT_OK;
};

class Trpo final : public Policy {  // policies/trpo.rs:63-164
 public:
  Trpo(Engine &eng, std::unique_ptr<Module> m, std::unique_ptr<TrustRegionOptimizer> opt, double max_policy_step_kl)
      : Policy(eng, std::move(m)), optimizer_(std::move(opt)) {
    check(rl_trpo_config_default(&cfg_));
    optimizer_->configure(cfg_);
    cfg_.max_policy_step_kl = max_policy_step_kl;
  }
  PolicyKind kind() const override { return PolicyKind::Trpo; }
  void update(DeviceHistory &history, StatsLogger &logger) override {
    rl_trpo_stats st{};
    check(rl_trpo_update(module_->handle(), history.handle(), &cfg_, &st), eng_.handle());  // OptimizerNanError: the
                                                                                           // reference panics (trpo.rs:154-162)
    log_stats(st, logger);
  }
  const rl_trpo_config &config() const { return cfg_; }
  void log_stats(const rl_trpo_stats &st, StatsLogger &logger) {
    logger.log_scalar("entropy", st.entropy);                  // trpo.rs:119
    logger.log_scalar("step_size", st.step_size);              // conjugate_gradient.rs:164
    logger.log_scalar("loss_initial", st.loss_initial);        // :200
    if (st.num_backtracks >= 0) {
      logger.log_scalar("num_backtracks", (double)st.num_backtracks);  // :219
      logger.log_scalar("step_scale", st.step_scale);                  // :220
    }
    logger.log_scalar("loss_final", st.loss_final);            // :225
    logger.log_scalar("constraint_val_final", st.constraint_val_final);  // :226
    last_status_ = st.status;  // OptimizerStepError::{LossNotImproving, ConstraintViolated}: warn and continue
  }

 private:
  std::unique_ptr<TrustRegionOptimizer> optimizer_;
  rl_trpo_config cfg_{};
};

class Ppo final : public Policy {  // policies/ppo.rs:63-137
 public:
  Ppo(Engine &eng, std::unique_ptr<Module> m, std::unique_ptr<Optimizer> opt, uint64_t opt_steps_per_update,
      double clip_distance)
      : Policy(eng, std::move(m)), optimizer_(std::move(opt)) {
    cfg_.opt_steps_per_update = opt_steps_per_update;
    cfg_.clip_distance = clip_distance;
  }
  PolicyKind kind() const override { return PolicyKind::Ppo; }
  void update(DeviceHistory &history, StatsLogger &logger) override {
    rl_policy_opt_stats st{};
    check(rl_ppo_update(module_->handle(), optimizer_->handle(), history.handle(), &cfg_, &st, nullptr), eng_.handle());
    logger.log_scalar("entropy", st.entropy);  // ppo.rs:114 (ToLog::NoAbsLoss: no final loss)
  }

 private:
  std::unique_ptr<Optimizer> optimizer_;
  rl_ppo_config cfg_{};
};

class Reinforce final : public Policy {  // policies/reinforce.rs:40-90
 public:
  Reinforce(Engine &eng, std::unique_ptr<Module> m, std::unique_ptr<Optimizer> opt)
      : Policy(eng, std::move(m)), optimizer_(std::move(opt)) {}
  PolicyKind kind() const override { return PolicyKind::Reinforce; }
  void update(DeviceHistory &history, StatsLogger &logger) override {
    rl_policy_opt_stats st{};
    check(rl_reinforce_update(module_->handle(), optimizer_->handle(), history.handle(), &st), eng_.handle());
    logger.log_scalar("loss", st.loss_first);
    logger.log_scalar("entropy", st.entropy);  // reinforce.rs:84
  }

 private:
  std::unique_ptr<Optimizer> optimizer_;
};

template <typename MB = MlpConfig>
struct TrpoConfig {  // policies/trpo.rs:18-41
  MB policy_fn_config;
  ConjugateGradientOptimizerConfig optimizer_config;
  double max_policy_step_kl = 0.01;
  std::unique_ptr<Policy> build_policy(Engine &eng, uint32_t in_dim, uint32_t out_dim, uint64_t seed) const {
    return std::unique_ptr<Policy>(new Trpo(eng, policy_fn_config.build_module(eng, in_dim, out_dim, seed),
                                            optimizer_config.build_optimizer(), max_policy_step_kl));
  }
};
template <typename MB = MlpConfig>
struct PpoConfig {  // policies/ppo.rs:13-41
  MB policy_fn_config;
  AdamConfig optimizer_config;
  uint64_t opt_steps_per_update = 10;
  double clip_distance = 0.2;
  std::unique_ptr<Policy> build_policy(Engine &eng, uint32_t in_dim, uint32_t out_dim, uint64_t seed) const {
    std::unique_ptr<Module> m = policy_fn_config.build_module(eng, in_dim, out_dim, seed);
    std::unique_ptr<Optimizer> o = optimizer_config.build_optimizer(*m, eng);
    return std::unique_ptr<Policy>(new Ppo(eng, std::move(m), std::move(o), opt_steps_per_update, clip_distance));
  }
};
template <typename MB = MlpConfig>
struct ReinforceConfig {  // policies/reinforce.rs
  MB policy_fn_config;
  AdamConfig optimizer_config;
  std::unique_ptr<Policy> build_policy(Engine &eng, uint32_t in_dim, uint32_t out_dim, uint64_t seed) const {
    std::unique_ptr<Module> m = policy_fn_config.build_module(eng, in_dim, out_dim, seed);
    std::unique_ptr<Optimizer> o = optimizer_config.build_optimizer(*m, eng);
    return std::unique_ptr<Policy>(new Reinforce(eng, std::move(m), std::move(o)));
  }
};

// Critic (src/torch/agents/critics/mod.rs:20-99): turns a history into per-step advantages for the policy and
// improves itself from the same history afterwards.
class Critic {
 public:
  virtual ~Critic() = default;
  virtual void advantages(DeviceHistory &history) = 0;                    // Critic::advantages
  virtual void update(DeviceHistory &history, StatsLogger &logger) = 0;   // Critic::update
  virtual Module *module() { return nullptr; }                            // a critic may have no trainable state
  virtual double discount_factor() const = 0;
};

enum class StepValueTarget { RewardToGo, OneStepTd };  // critics/mod.rs:203-215 (default RewardToGo)

class ValuesOpt final : public Critic {  // critics/opt.rs:41-126
 public:
  ValuesOpt(Engine &eng, std::unique_ptr<Module> m, std::unique_ptr<Optimizer> opt, double discount_factor,
            double gae_lambda, StepValueTarget target, uint64_t opt_steps_per_update)
      : eng_(eng), module_(std::move(m)), optimizer_(std::move(opt)), gamma_((float)discount_factor),
        lambda_(gae_lambda), target_(target), steps_(opt_steps_per_update) {}
  void advantages(DeviceHistory &history) override {
    check(rl_gae(history.handle(), module_->handle(), gamma_, (float)lambda_), eng_.handle());
  }
  void update(DeviceHistory &history, StatsLogger &logger) override {
    // ValuesOpt::update (critics/opt.rs:100-126): targets once under no-grad, then n_backward_steps
    rl_critic_stats cs{};
    rl_values_opt_config vc{};
    vc.opt_steps_per_update = steps_;
    vc.target = target_ == StepValueTarget::OneStepTd ? RL_VALUE_TARGET_ONE_STEP_TD : RL_VALUE_TARGET_REWARD_TO_GO;
    vc.discount_factor = gamma_;
    check(rl_values_opt_update(module_->handle(), optimizer_->handle(), history.handle(), &vc, &cs, nullptr), eng_.handle());
    logger.log_scalar("loss", cs.loss_last);  // n_backward_steps, ToLog::All (torch/agents/mod.rs:68-70)
  }
  rl_values_opt_config config() const {
    rl_values_opt_config vc{};
    vc.opt_steps_per_update = steps_;
    vc.target = target_ == StepValueTarget::OneStepTd ? RL_VALUE_TARGET_ONE_STEP_TD : RL_VALUE_TARGET_REWARD_TO_GO;
    vc.discount_factor = gamma_;
    return vc;
  }
  Optimizer &optimizer() { return *optimizer_; }
  Module *module() override { return module_.get(); }
  double discount_factor() const override { return gamma_; }

 private:
  Engine &eng_;
  std::unique_ptr<Module> module_;
  std::unique_ptr<Optimizer> optimizer_;
  float gamma_;
  double lambda_;
  StepValueTarget target_;
  uint64_t steps_;
};

class RewardToGo final : public Critic {  // critics/rtg.rs:22-40
 public:
  RewardToGo(Engine &eng, double discount_factor) : eng_(eng), gamma_((float)discount_factor) {}
  void advantages(DeviceHistory &history) override { check(rl_reward_to_go(history.handle(), gamma_), eng_.handle()); }
  void update(DeviceHistory &, StatsLogger &) override {}  // RewardToGo::update does nothing
  double discount_factor() const override { return gamma_; }

 private:
  Engine &eng_;
  float gamma_;
};

template <typename MB = MlpConfig>
struct ValuesOptConfig {  // critics/opt.rs:13-37
  MB state_value_fn_config;
  AdamConfig optimizer_config;
  double gae_lambda = 0.95;          // AdvantageFn::Gae { lambda }
  StepValueTarget target = StepValueTarget::RewardToGo;
  uint64_t opt_steps_per_update = 80;
  double max_discount_factor = 0.99;
  std::unique_ptr<Critic> build_critic(Engine &eng, uint32_t in_dim, double env_discount_factor, uint64_t seed) const {
    std::unique_ptr<Module> m = state_value_fn_config.build_module(eng, in_dim, 1, seed);
    std::unique_ptr<Optimizer> o = optimizer_config.build_optimizer(*m, eng);
    const double g = env_discount_factor < max_discount_factor ? env_discount_factor : max_discount_factor;  // opt.rs:73
    return std::unique_ptr<Critic>(new ValuesOpt(eng, std::move(m), std::move(o), g, gae_lambda, target, opt_steps_per_update));
  }
};
struct RewardToGoConfig {  // critics/rtg.rs:14-20
  std::unique_ptr<Critic> build_critic(Engine &eng, uint32_t, double env_discount_factor, uint64_t) const {
    return std::unique_ptr<Critic>(new RewardToGo(eng, env_discount_factor));
  }
};

// ---------------------------------------------------------------- Actor (src/agents/mod.rs:101-114)
// `act(episode_state, observation, rng) -> action`, one observation at a time.  The batched rollouts are the fast
// path (`ActorCriticAgent::collect`); this is the reference's per-step interface for a caller that steps its own
// environment.  The random state belongs to the caller, as in the reference.
class Actor {
 public:
  virtual ~Actor() = default;
  virtual uint32_t act(const std::vector<float> &observation_features, Prng &rng) = 0;
};

// PolicyActor (policies/actor.rs:30-56) over a feed-forward policy module: logits from the device module,
// Categorical::new (log_softmax) and the inverse-CDF draw of the rollouts (`log_probs.exp().multinomial(1)` with an
// explicit uniform from `rng`; same arithmetic as device_fns.hpp, include/rl_detmath.h).  With `rng` at word t of
// stream `lane` of the env's actor seed this returns exactly the action the device rollout takes for that lane and
// step.  Recurrent modules carry their episode state on the device: they act through the rollouts only.
class PolicyActor final : public Actor {
 public:
  PolicyActor(Engine &eng, Module &policy_module, uint32_t in_dim, uint32_t n_actions)
      : eng_(eng), module_(policy_module), in_dim_(in_dim), n_actions_(n_actions) {}
  uint32_t act(const std::vector<float> &x, Prng &rng) override {
    if (x.size() != in_dim_) throw Error(RL_ERR_INVALID_ARGUMENT, "observation has the wrong number of features");
    std::vector<float> z(n_actions_), lp(n_actions_);
    check(rl_mlp_forward(module_.handle(), x.data(), 1, z.data()), eng_.handle());
    float m = z[0];
    for (uint32_t a = 1; a < n_actions_; ++a)
      if (z[a] > m) m = z[a];
    float s = 0.0f;
    for (uint32_t a = 0; a < n_actions_; ++a) s += rl_expf(z[a] - m);
    const float ls = rl_logf(s);
    for (uint32_t a = 0; a < n_actions_; ++a) lp[a] = (z[a] - m) - ls;
    const float u = rng.gen_f32();
    float cum = 0.0f;
    for (uint32_t a = 0; a + 1 < n_actions_; ++a) {
      cum += rl_expf(lp[a]);
      if (u < cum) return a;
    }
    return n_actions_ - 1;  // the last index absorbs rounding
  }

 private:
  Engine &eng_;
  Module &module_;
  uint32_t in_dim_, n_actions_;
};

template <typename P, typename C>
struct ActorCriticConfig;

// ActorCriticAgent (actor_critic.rs:72-136) over a Policy and a Critic; `batch_update` follows batch_update_slice
// (actor_critic.rs:176-211): advantages, policy update, critic update, with the reference's metric names.
class ActorCriticAgent {
 public:
  Policy &policy() { return *policy_; }
  Critic &critic() { return *critic_; }
  Module &policy_module() { return policy_->module(); }
  Module *critic_module() { return critic_->module(); }
  // Agent::actor(mode): the policy module is the actor's state (both modes sample from the policy's distribution,
  // policies/mod.rs:42-53); the returned actor acts one observation at a time, `collect` is the batched form
  std::unique_ptr<Actor> actor(ActorMode) {
    return std::unique_ptr<Actor>(new PolicyActor(eng_, policy_->module(), obs_dim_, n_actions_));
  }
  DeviceHistory buffer(uint64_t horizon) const { return DeviceHistory(eng_, n_lanes_, horizon, obs_dim_); }

  // BatchUpdate::batch_update
  void batch_update(DeviceHistory &history, StatsLogger &logger) {
    log_elapsed(logger, "adv_est_time", [&] {
      critic_->advantages(history);
      eng_.sync();
    });
    ScopedLogger pl(logger, "policy");
    ScopedLogger cl(logger, "critic");
    Trpo *trpo = dynamic_cast<Trpo *>(policy_.get());
    ValuesOpt *vopt = dynamic_cast<ValuesOpt *>(critic_.get());
    if (trpo != nullptr && vopt != nullptr) {
      // policy.update and critic.update share nothing but the history and its advantages: one call, the two launch chains
      // side by side on two streams (rl_actor_critic_update; the same numbers as the two updates in turn).  Both scopes
      // log the joint wall time as their update_time.
      rl_trpo_stats ps{};
      rl_critic_stats cs{};
      const rl_values_opt_config vc = vopt->config();
      const auto t0 = std::chrono::steady_clock::now();
      check(rl_actor_critic_update(trpo->module().handle(), vopt->module()->handle(), vopt->optimizer().handle(),
                                   history.handle(), &trpo->config(), &vc, &ps, &cs, nullptr),
            eng_.handle());
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      trpo->log_stats(ps, pl);
      pl.log_duration("update_time", secs);
      cl.log_scalar("loss", cs.loss_last);
      cl.log_duration("update_time", secs);
      return;
    }
    log_elapsed(pl, "update_time", [&] { policy_->update(history, pl); eng_.sync(); });
    log_elapsed(cl, "update_time", [&] { critic_->update(history, cl); });
  }

  // one data-collection pass of every lane with the current policy (train.rs:124-158's thread fan-out)
  void collect(EnvLanes &env, DeviceHistory &history) {
    check(rl_rollout(env.handle(), policy_->module().handle(), history.handle()), eng_.handle());
  }
  int32_t last_status() const { return policy_->last_status(); }

 private:
  template <typename P, typename C>
  friend struct ActorCriticConfig;
  ActorCriticAgent(Engine &eng) : eng_(eng) {}
  Engine &eng_;
  std::unique_ptr<Policy> policy_;
  std::unique_ptr<Critic> critic_;
  uint64_t n_lanes_ = 0;
  uint32_t obs_dim_ = 0, n_actions_ = 0;
};

template <typename P, typename C>
struct ActorCriticConfig {  // actor_critic.rs:20-45
  P policy_config;
  C critic_config;

  // BuildAgent::build_agent(env, rng): `seed` seeds the engine-defined initialisation streams
  std::unique_ptr<ActorCriticAgent> build_agent(EnvLanes &env, uint64_t seed) const {
    Engine &eng = env.engine();
    std::unique_ptr<ActorCriticAgent> a(new ActorCriticAgent(eng));
    a->n_lanes_ = env.num_lanes();
    a->obs_dim_ = env.num_observation_features();
    a->n_actions_ = env.num_actions();
    a->policy_ = policy_config.build_policy(eng, env.num_observation_features(), env.num_actions(), seed);
    a->critic_ = critic_config.build_critic(eng, env.num_observation_features(), env.discount_factor(), seed + 1);
    return a;
  }
};

// ---------------------------------------------------------------- DQN (src/torch/agents/dqn.rs)
template <typename VB = MlpConfig>
struct DqnConfig {  // dqn.rs:26-72
  VB action_value_fn_config;
  AdamConfig optimizer_config;
  bool one_step_td = false;  // StepValueTarget::{RewardToGo, OneStepTd}
  double exploration_start = 1.0, exploration_end = 0.1;
  uint64_t exploration_period = 10000000;
  uint64_t minibatch_steps = 100000, opt_steps_per_update = 50;
  uint64_t buffer_capacity = 10000000;  // TOTAL steps; divided over the lanes (each lane is one ReplayBuffer)
  uint64_t update_first = 1000000, update_rest = 100000;  // DataCollectionSchedule::FirstRest
};

class DqnAgent;
template <typename VB>
std::unique_ptr<DqnAgent> build_dqn_agent(const DqnConfig<VB> &c, EnvLanes &env, uint64_t seed, const uint32_t (&agent_key)[8]);

class DqnAgent {
 public:
  ~DqnAgent() { rl_dqn_destroy(dqn_); }  // before the optimizer and the module it refers to
  Module &action_value_fn() { return *q_; }
  struct Bound { uint64_t min_steps, slack_steps; };
  Bound min_update_size() const {  // dqn.rs:207-209
    Bound b{};
    check(rl_dqn_min_update_size(dqn_, &b.min_steps, &b.slack_steps), eng_.handle());
    return b;
  }
  double exploration_rate(ActorMode mode) const {
    double r = 0.0;
    check(rl_dqn_exploration_rate(dqn_, mode == ActorMode::Training ? 1 : 0, &r), eng_.handle());
    return r;
  }
  // every lane takes ceil(min_steps / lanes) steps with the epsilon-greedy actor (HistoryDataBound::divide)
  void collect(StatsLogger &logger) {
    const Bound b = min_update_size();
    const uint64_t horizon = (b.min_steps + n_lanes_ - 1) / n_lanes_;
    rl_dqn_collect_stats st{};
    check(rl_dqn_collect(dqn_, horizon, &st), eng_.handle());  // WriteExperienceFull when an episode outgrows a lane
    ScopedLogger sim(logger, "sim");
    sim.log_counter_increment("step/count", st.steps);           // train.rs:175
    sim.log_counter_increment("ep/count", st.episodes_ended);    // train.rs:171
  }
  void batch_update(StatsLogger &logger) {  // dqn.rs:263-337
    logger.log_scalar("exploration_rate", exploration_rate(ActorMode::Training));  // :268-272
    rl_dqn_update_stats st{};
    check(rl_dqn_update(dqn_, &st, nullptr), eng_.handle());
    logger.log_scalar("loss", st.loss_last);  // ToLog::All
    global_steps_ = st.global_steps;
  }
  uint64_t global_steps() const { return global_steps_; }

 private:
  template <typename VB>
  friend std::unique_ptr<DqnAgent> build_dqn_agent(const DqnConfig<VB> &, EnvLanes &, uint64_t, const uint32_t (&)[8]);
  explicit DqnAgent(Engine &eng) : eng_(eng) {}
  Engine &eng_;
  std::unique_ptr<Module> q_;
  std::unique_ptr<Optimizer> opt_;
  rl_dqn *dqn_ = nullptr;
  uint64_t n_lanes_ = 0, global_steps_ = 0;
};

// BuildAgent for DqnConfig (dqn.rs:74-96); `agent_key` is the 32-byte seed `Prng::from_rng(rng)` draws
template <typename VB>
std::unique_ptr<DqnAgent> build_dqn_agent(const DqnConfig<VB> &c, EnvLanes &env, uint64_t seed, const uint32_t (&agent_key)[8]) {
  Engine &eng = env.engine();
  std::unique_ptr<DqnAgent> a(new DqnAgent(eng));
  a->n_lanes_ = env.num_lanes();
  a->q_ = c.action_value_fn_config.build_module(eng, env.num_observation_features(), env.num_actions(), seed);
  a->opt_ = c.optimizer_config.build_optimizer(*a->q_, eng);
  rl_dqn_config d;
  check(rl_dqn_config_default(&d));
  d.target = c.one_step_td ? RL_DQN_TARGET_ONE_STEP_TD : RL_DQN_TARGET_REWARD_TO_GO;
  d.exploration_kind = RL_SCHEDULE_LINEAR_ANNEALED;
  d.exploration_start = c.exploration_start;
  d.exploration_end = c.exploration_end;
  d.exploration_period = c.exploration_period;
  d.minibatch_steps = c.minibatch_steps;
  d.opt_steps_per_update = c.opt_steps_per_update;
  d.buffer_capacity = (c.buffer_capacity + env.num_lanes() - 1) / env.num_lanes();
  d.update_kind = RL_COLLECT_FIRST_REST;
  d.update_first = c.update_first;
  d.update_rest = c.update_rest;
  d.discount_factor = (float)env.discount_factor();
  std::memcpy(d.agent_key, agent_key, sizeof(d.agent_key));
  check(rl_dqn_create(env.handle(), a->q_->handle(), a->opt_->handle(), &d, &a->dqn_), eng.handle());
  return a;
}

// ---------------------------------------------------------------- actor serialisation (examples/cartpole-trpo.rs:71-93)
// `serde_cbor::to_writer(file, &agent.actor(ActorMode::Evaluation))` / `serde_cbor::from_reader(file)`
inline std::vector<uint8_t> actor_to_cbor(EnvLanes &env, Module &module, int32_t actor_kind = RL_ACTOR_POLICY,
                                          double exploration_rate = 0.0) {
  uint64_t len = 0;
  check(rl_actor_to_cbor(env.handle(), module.handle(), actor_kind, exploration_rate, nullptr, 0, &len),
        env.engine().handle());
  std::vector<uint8_t> buf(len);
  check(rl_actor_to_cbor(env.handle(), module.handle(), actor_kind, exploration_rate, buf.data(), len, &len),
        env.engine().handle());
  return buf;
}
inline void module_from_cbor(Module &module, const std::vector<uint8_t> &doc, Engine &eng) {
  check(rl_module_from_cbor(module.handle(), doc.data(), doc.size()), eng.handle());
}

// ---------------------------------------------------------------- the period loop of train_parallel (train.rs:68-186)
inline void train_batched(ActorCriticAgent &agent, EnvLanes &env, DeviceHistory &history, uint64_t num_periods,
                          StatsLogger &logger) {
  for (uint64_t period = 0; period < num_periods; ++period) {
    const auto collect_start = std::chrono::steady_clock::now();
    agent.collect(env, history);
    env.engine().sync();
    {
      ScopedLogger sim(logger, "sim");
      uint64_t episodes = 0;
      for (uint8_t f : history.successors()) episodes += f != RL_SUCC_CONTINUE;
      // train.rs:160-175 logs the period's StepsSummary; with lanes that persist across periods the episode statistics
      // are per period: mean length = steps / episode ends (a moving estimate), CartPole's episode reward = its length
      if (episodes > 0) sim.log_scalar("ep/length_mean", (double)history.num_steps() / (double)episodes);
      sim.log_counter_increment("ep/count", episodes);                 // train.rs:171
      sim.log_counter_increment("step/count", history.num_steps());    // train.rs:175
      sim.log_duration("time", std::chrono::duration<double>(std::chrono::steady_clock::now() - collect_start).count());
    }
    const auto update_start = std::chrono::steady_clock::now();
    agent.batch_update(history, logger);
    ScopedLogger up(logger, "agent_update");
    up.log_duration("time", std::chrono::duration<double>(std::chrono::steady_clock::now() - update_start).count());
    up.log_counter_increment("count", 1);  // train.rs:182-184
  }
}

}  // namespace relearn
