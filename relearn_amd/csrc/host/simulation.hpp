// simulation.hpp — host-side mirrors of the reference's simulation driver: Step / Steps (src/simulation/
// steps.rs:113-167), HistoryDataBound (src/agents/buffers/mod.rs:25-113), TakeAlignedSteps (src/simulation/
// take_steps.rs:18-93), VecBuffer + finalize_last_episode (src/agents/buffers/vec.rs, mod.rs:237-261) and
// train_parallel (src/simulation/train.rs:68-186).
#pragma once
#include <cstdint>
#include <optional>
#include <thread>
#include <vector>

#include "envs.hpp"

namespace relearn {

// PartialStep<O, A> (src/simulation/mod.rs:35-91): the continuing successor observation is omitted
template <typename O, typename A>
struct PartialStep {
  O observation;
  A action;
  double feedback;
  SuccessorKind next;
  std::optional<O> interrupt_observation;  // owned when next == Interrupt
  bool episode_done() const { return next != SuccessorKind::Continue; }
};

struct HistoryDataBound {
  uint64_t min_steps = 0, slack_steps = 0;
  HistoryDataBound divide(uint64_t n) const { return {min_steps / n + (min_steps % n ? 1 : 0), slack_steps}; }
  HistoryDataBound max(HistoryDataBound o) const {
    return {min_steps > o.min_steps ? min_steps : o.min_steps, slack_steps > o.slack_steps ? slack_steps : o.slack_steps};
  }
  static HistoryDataBound with_default_slack(uint64_t min_steps) {
    uint64_t slack = min_steps / 100;
    slack = slack < 5 ? 5 : (slack > 1000 ? 1000 : slack);
    return {min_steps, slack};
  }
};

// Steps: the environment-actor loop with separate env / actor generators
template <typename E, typename Actor>
class Steps {
 public:
  using Step = PartialStep<typename E::Observation, typename E::Action>;
  Steps(const E &env, const Actor &actor, Prng &rng_env, Prng &rng_actor)
      : env_(env), actor_(actor), rng_env_(rng_env), rng_actor_(rng_actor) {}

  Step next() {
    if (!state_) {  // start a new episode
      auto s = env_.initial_state(rng_env_);
      auto o = env_.observe(s, rng_env_);
      state_.emplace(Episode{std::move(s), std::move(o)});
    }
    Episode ep = std::move(*state_);
    state_.reset();
    const auto action = actor_.act(ep.observation, rng_actor_);
    auto [succ, reward] = env_.step(std::move(ep.env), action, rng_env_);
    Step out{ep.observation, action, reward, succ.kind, std::nullopt};
    if (succ.kind == SuccessorKind::Continue) {
      auto o = env_.observe(*succ.state, rng_env_);
      state_.emplace(Episode{std::move(*succ.state), std::move(o)});
    } else if (succ.kind == SuccessorKind::Interrupt) {
      out.interrupt_observation = env_.observe(*succ.state, rng_env_);
    }
    return out;
  }

 private:
  struct Episode {
    typename E::State env;
    typename E::Observation observation;
  };
  const E &env_;
  const Actor &actor_;
  Prng &rng_env_, &rng_actor_;
  std::optional<Episode> state_;
};

// VecBuffer: steps of successive episodes + one-past-the-end index of each episode
template <typename O, typename A>
class VecBuffer {
 public:
  using Step = PartialStep<O, A>;
  // WriteExperience::write_experience with TakeAlignedSteps applied to the source
  template <typename Source>
  void write_experience(Source &source, HistoryDataBound bound) {
    uint64_t n = bound.min_steps == 0 ? 0 : bound.min_steps + bound.slack_steps;
    while (n != 0) {
      Step s = source.next();
      n -= 1;
      if (s.episode_done() && n <= bound.slack_steps) n = 0;  // ended inside the slack interval
      const bool done = s.episode_done();
      steps_.push_back(std::move(s));
      if (done) episode_ends_.push_back(steps_.size());
    }
    end_experience();
  }
  // finalize_last_episode: a trailing Continue step is dropped; its observation becomes the Interrupt successor
  // of the step before it, unless that one already ended an episode
  void end_experience() {
    if (steps_.empty() || steps_.back().episode_done()) return;
    O final_observation = std::move(steps_.back().observation);
    steps_.pop_back();
    if (!steps_.empty() && !steps_.back().episode_done()) {
      steps_.back().next = SuccessorKind::Interrupt;
      steps_.back().interrupt_observation = std::move(final_observation);
      episode_ends_.push_back(steps_.size());
    }
  }
  void clear() {
    steps_.clear();
    episode_ends_.clear();
  }
  uint64_t num_steps() const { return steps_.size(); }
  uint64_t num_episodes() const { return episode_ends_.size(); }
  const std::vector<Step> &steps() const { return steps_; }
  const std::vector<uint64_t> &episode_ends() const { return episode_ends_; }
  std::vector<Step> drain_steps() {
    std::vector<Step> out;
    out.swap(steps_);
    episode_ends_.clear();
    return out;
  }

 private:
  std::vector<Step> steps_;
  std::vector<uint64_t> episode_ends_;
};

struct TrainParallelConfig {
  uint64_t num_periods = 1, num_threads = 1, min_worker_steps = 0;
};

// train_parallel: per-thread generators forked once with from_rng (env stream first, then agent stream, per
// thread); each period every worker fills its own buffer with an immutable actor snapshot, then the agent is
// updated from all buffers in thread order.
template <typename Agent, typename E>
void train_parallel(Agent &agent, const E &env, const TrainParallelConfig &cfg, Prng &rng_env, Prng &rng_agent) {
  using Buffer = typename Agent::HistoryBuffer;
  std::vector<Buffer> buffers(cfg.num_threads);
  std::vector<Prng> thread_env, thread_agent;
  for (uint64_t i = 0; i < cfg.num_threads; ++i) {
    thread_env.push_back(Prng::from_rng(rng_env));
    thread_agent.push_back(Prng::from_rng(rng_agent));
  }
  for (uint64_t period = 0; period < cfg.num_periods; ++period) {
    const HistoryDataBound bound =
        agent.min_update_size().divide(cfg.num_threads).max(HistoryDataBound{cfg.min_worker_steps, 0});
    std::vector<std::thread> workers;
    for (uint64_t i = 0; i < cfg.num_threads; ++i) {
      workers.emplace_back([&, i] {
        auto actor = agent.actor(/*training=*/true);
        Steps<E, decltype(actor)> steps(env, actor, thread_env[i], thread_agent[i]);
        buffers[i].write_experience(steps, bound);
      });
    }
    for (auto &w : workers) w.join();
    agent.batch_update(buffers);
  }
}

}  // namespace relearn
