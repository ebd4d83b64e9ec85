// host_api_demo.cpp — the C++ host API (relearn_amd/csrc/host/agents.hpp) used the way the reference's examples use
// its traits: examples/cartpole-trpo.rs (ActorCriticConfig<TrpoConfig<MlpConfig>, ValuesOptConfig<MlpConfig>>,
// train_parallel), a PPO agent with the recurrent module on Chain, and examples/cartpole-dqn.rs.
// Prints one JSON object: logged metric names/values and a checksum of every module's parameters, which
// tests/test_gpu_host_api.py compares with the same runs driven through the ctypes binding.
#include <cinttypes>
#include <cstdio>

#include "../../relearn_amd/csrc/host/agents.hpp"

using namespace relearn;

static double checksum(const std::vector<float> &p) {
  double s = 0.0;
  for (size_t i = 0; i < p.size(); ++i) s += (double)p[i] * (double)(1 + (i % 7));
  return s;
}

static void dump(const char *name, const RecordingLogger &log, double pol, double cri, bool last) {
  std::printf("\"%s\": {\"policy_checksum\": %.17g, \"critic_checksum\": %.17g, \"scalars\": {", name, pol, cri);
  bool first = true;
  for (auto &kv : log.scalars) {
    std::printf("%s\"%s\": %.17g", first ? "" : ", ", kv.first.c_str(), kv.second);
    first = false;
  }
  std::printf("}, \"counters\": {");
  first = true;
  for (auto &kv : log.counters) {
    std::printf("%s\"%s\": %" PRIu64, first ? "" : ", ", kv.first.c_str(), kv.second);
    first = false;
  }
  std::printf("}, \"durations\": [");
  first = true;
  for (auto &kv : log.durations) {
    std::printf("%s\"%s\"", first ? "" : ", ", kv.first.c_str());
    first = false;
  }
  std::printf("]}%s\n", last ? "" : ",");
}

int main() {
  try {
    Engine eng(0);
    std::printf("{\n");
    {  // examples/cartpole-trpo.rs
      CartPoleLanes env(eng, 256, 500, StepLimit::Visible, /*seed_env=*/0, /*seed_actor=*/1);
      ActorCriticConfig<TrpoConfig<MlpConfig>, ValuesOptConfig<MlpConfig>> cfg;
      cfg.critic_config.opt_steps_per_update = 5;
      auto agent = cfg.build_agent(env, /*seed=*/2);
      DeviceHistory history = agent->buffer(32);
      RecordingLogger log;
      train_batched(*agent, env, history, 2, log);
      dump("trpo", log, checksum(agent->policy_module().parameters()), checksum(agent->critic_module()->parameters()), false);
    }
    {  // PPO over the recurrent module on the partially observed Chain
      ChainLanes env(eng, 64, 100, StepLimit::Latent, 3, 4);
      ActorCriticConfig<PpoConfig<GruMlpConfig>, ValuesOptConfig<GruMlpConfig>> cfg;
      cfg.policy_config.opt_steps_per_update = 2;
      cfg.critic_config.opt_steps_per_update = 2;
      auto agent = cfg.build_agent(env, 11);
      DeviceHistory history = agent->buffer(20);
      RecordingLogger log;
      train_batched(*agent, env, history, 1, log);
      dump("ppo_gru", log, checksum(agent->policy_module().parameters()), checksum(agent->critic_module()->parameters()), false);
    }
    {  // a stacked recurrent chain with RnnBaseConfig's fields spelled out: two GRU layers of 24 units, Normal(FanIn) input
       // weights, an MLP head of 16 units with constant biases — through the same agent configuration
      ChainLanes env(eng, 64, 100, StepLimit::Latent, 3, 4);
      ActorCriticConfig<PpoConfig<GruMlpConfig>, ValuesOptConfig<GruMlpConfig>> cfg;
      for (GruMlpConfig *c : {&cfg.policy_config.policy_fn_config, &cfg.critic_config.state_value_fn_config}) {
        c->first_config.num_layers = 2;
        c->first_config.input_weights_init = Initializer::Normal(RL_SCALE_FAN_IN);
        c->hidden_dim = 24;
        c->second_config.hidden_sizes = {16};
        c->second_config.linear_config.bias_init = Initializer::Constant(0.125);
      }
      cfg.policy_config.opt_steps_per_update = 2;
      cfg.critic_config.opt_steps_per_update = 2;
      auto agent = cfg.build_agent(env, 15);
      const uint64_t G = 3 * 24, want = G * 5 + G * 24 + 2 * G + G * 24 + G * 24 + 2 * G + 16 * 24 + 16 + 2 * 16 + 2;
      if (agent->policy_module().num_parameters() != want) return 4;
      DeviceHistory history = agent->buffer(20);
      RecordingLogger log;
      train_batched(*agent, env, history, 1, log);
      dump("ppo_gru_stacked", log, checksum(agent->policy_module().parameters()), checksum(agent->critic_module()->parameters()), false);
    }
    {  // PPO with the critic-free reward-to-go advantage on MemoryGame lanes; `LstmMlpConfig` as the reference defines it
      MemoryGameLanes env(eng, 64, 2, 3, 0, StepLimit::None, 5, 6);
      ActorCriticConfig<PpoConfig<LstmMlpConfig>, RewardToGoConfig> cfg;
      cfg.policy_config.opt_steps_per_update = 2;
      auto agent = cfg.build_agent(env, 13);
      DeviceHistory history = agent->buffer(24);
      RecordingLogger log;
      train_batched(*agent, env, history, 1, log);
      dump("ppo_memory", log, checksum(agent->policy_module().parameters()), 0.0, false);
    }
    {  // examples/cartpole-dqn.rs, shrunk
      CartPoleLanes env(eng, 128, 500, StepLimit::Visible, 0, 1);
      DqnConfig<MlpConfig> cfg;
      cfg.minibatch_steps = 1000;
      cfg.opt_steps_per_update = 3;
      cfg.buffer_capacity = 128 * 256;
      cfg.update_first = 128 * 40;
      cfg.update_rest = 128 * 10;
      cfg.exploration_period = 100000;
      const uint32_t key[8] = {1, 2, 3, 4, 5, 6, 7, 8};
      auto agent = build_dqn_agent(cfg, env, 7, key);
      RecordingLogger log;
      for (int period = 0; period < 2; ++period) {
        agent->collect(log);
        agent->batch_update(log);
      }
      log.scalars["global_steps"] = (double)agent->global_steps();
      dump("dqn", log, checksum(agent->action_value_fn().parameters()), 0.0, false);
    }
    {  // MlpConfig { hidden_sizes: [64, 64] }: the general per-layer path behind the same agent configuration
      CartPoleLanes env(eng, 128, 500, StepLimit::Visible, 7, 8);
      ActorCriticConfig<TrpoConfig<MlpConfig>, ValuesOptConfig<MlpConfig>> cfg;
      cfg.policy_config.policy_fn_config.hidden_sizes = {64, 64};
      cfg.critic_config.state_value_fn_config.hidden_sizes = {64, 64};
      cfg.critic_config.opt_steps_per_update = 3;
      auto agent = cfg.build_agent(env, 9);
      if (agent->policy_module().num_parameters() != 5 * 64 + 64 + 64 * 64 + 64 + 64 * 2 + 2) return 3;
      DeviceHistory history = agent->buffer(16);
      RecordingLogger log;
      train_batched(*agent, env, history, 1, log);
      dump("trpo_two_layers", log, checksum(agent->policy_module().parameters()), checksum(agent->critic_module()->parameters()), false);
    }
    {  // Actor::act, one observation at a time: the scalar actor repeats the device rollout's choices when its Prng
       // stands at word t of stream `lane` of the env's actor seed
      const uint64_t n = 64, T = 6, seed_actor = 21;
      CartPoleLanes env(eng, n, 500, StepLimit::Visible, /*seed_env=*/20, seed_actor);
      ActorCriticConfig<TrpoConfig<MlpConfig>, RewardToGoConfig> cfg;
      auto agent = cfg.build_agent(env, 22);
      DeviceHistory history = agent->buffer(T);
      agent->collect(env, history);
      eng.sync();
      std::vector<float> obs(5 * (T + 1) * n);
      std::vector<uint8_t> act(T * n);
      check(rl_traj_read(history.handle(), RL_TRAJ_OBS, obs.data(), obs.size() * sizeof(float)), eng.handle());
      check(rl_traj_read(history.handle(), RL_TRAJ_ACTION, act.data(), act.size()), eng.handle());
      std::unique_ptr<Actor> actor = agent->actor(ActorMode::Training);
      uint64_t checked = 0, mismatches = 0;
      for (uint64_t t = 0; t < T; ++t)
        for (uint64_t lane = 0; lane < n; lane += 7) {
          Prng rng = Prng::seed_from_u64(seed_actor);
          rng.set_stream(lane);
          rng.set_word_pos(t);
          std::vector<float> x(5);
          for (int d = 0; d < 5; ++d) x[d] = obs[(d * (T + 1) + t) * n + lane];
          mismatches += actor->act(x, rng) != act[t * n + lane];
          ++checked;
        }
      std::printf("\"actor\": {\"checked\": %llu, \"mismatches\": %llu}\n", (unsigned long long)checked,
                  (unsigned long long)mismatches);
    }
    std::printf("}\n");
    // error behaviour: an unsupported module shape is a BuildAgentError
    try {
      MlpConfig bad;
      bad.hidden_sizes = {4096};
      bad.build_module(eng, 5, 2, 0);
      return 2;
    } catch (const BuildAgentError &) {
    }
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
