// host_abi.cpp — C entry points of the CPU-only plumbing configuration (BASELINE.json configs[0]:
// examples/chain-tabular-q.rs).  The reference runs this path on the CPU too; nothing here touches the GPU.
#include <cstring>
#include <exception>

#include "../../include/relearn_hip.h"
#include "host/tabular.hpp"

using namespace relearn;

extern "C" {

// examples/chain-tabular-q.rs:12-45: rng = seed_from_u64(seed); env/agent builders draw nothing;
// train_parallel(agent, env, cfg, from_rng(rng), rng)
int32_t rl_chain_tabular_q_train(uint64_t seed, uint64_t n_threads, uint64_t n_periods, uint64_t min_worker_steps,
                                 double exploration_rate, double *q_values_out, uint64_t *counts_out,
                                 uint64_t *total_steps_out) {
  try {
    if (!q_values_out || !counts_out || n_threads == 0) return RL_ERR_INVALID_ARGUMENT;
    Chain env;
    TabularQLearningAgent agent(env.num_observations(), env.num_actions(), env.discount_factor, exploration_rate);
    Prng rng = Prng::seed_from_u64(seed);
    Prng rng_env = Prng::from_rng(rng);
    TrainParallelConfig cfg{n_periods, n_threads, min_worker_steps};
    train_parallel(agent, env, cfg, rng_env, rng);
    std::memcpy(q_values_out, agent.values().data(), agent.values().size() * sizeof(double));
    std::memcpy(counts_out, agent.counts().data(), agent.counts().size() * sizeof(uint64_t));
    if (total_steps_out) {
      uint64_t total = 0;
      for (auto c : agent.counts()) total += c;
      *total_steps_out = total;
    }
    return RL_OK;
  } catch (const std::exception &) {
    return RL_ERR_INVALID_ARGUMENT;
  }
}

// env.run(&agent.actor(ActorMode::Evaluation), SimSeed::Root(seed), ()).take(n_steps)
// (examples/chain-tabular-q.rs:47-50; SimSeed::derive_rngs src/simulation/mod.rs:137-149)
int32_t rl_chain_tabular_q_eval(const double *q_values, uint64_t seed, uint64_t n_steps, uint8_t *actions_out,
                                double *total_reward_out) {
  try {
    if (!q_values || !total_reward_out) return RL_ERR_INVALID_ARGUMENT;
    Chain env;
    TabularQLearningAgent agent(env.num_observations(), env.num_actions(), env.discount_factor, 0.2);
    agent.set_values(q_values);
    Prng rng_env = Prng::seed_from_u64(seed);
    Prng rng_agent = Prng::seed_from_u64(rng_env.next_u64());
    auto actor = agent.actor(/*training=*/false);
    Steps<Chain, TabularQLearningActor> steps(env, actor, rng_env, rng_agent);
    double total = 0.0;
    for (uint64_t t = 0; t < n_steps; ++t) {
      auto s = steps.next();
      if (actions_out) actions_out[t] = s.action;
      total += s.feedback;
    }
    *total_reward_out = total;
    return RL_OK;
  } catch (const std::exception &) {
    return RL_ERR_INVALID_ARGUMENT;
  }
}

}  // extern "C"
