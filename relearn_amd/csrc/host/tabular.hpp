// tabular.hpp — epsilon-greedy tabular Q-learning (reference src/agents/tabular.rs:88-233).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "simulation.hpp"

namespace relearn {

class TabularQLearningActor {
 public:
  TabularQLearningActor(std::shared_ptr<const std::vector<double>> q, uint64_t n_actions, double eps, bool training)
      : q_(std::move(q)), n_actions_(n_actions), eps_(eps), training_(training) {}
  // BaseTabularQLearningActor::act: explore with probability eps in training mode, else the first maximal action
  uint8_t act(uint64_t observation, Prng &rng) const {
    if (training_ && rng.gen_f64() < eps_) return static_cast<uint8_t>(rng.gen_range(0, n_actions_));
    const double *row = q_->data() + observation * n_actions_;
    uint64_t best = 0;
    for (uint64_t a = 1; a < n_actions_; ++a)
      if (row[a] > row[best]) best = a;
    return static_cast<uint8_t>(best);
  }

 private:
  std::shared_ptr<const std::vector<double>> q_;
  uint64_t n_actions_;
  double eps_;
  bool training_;
};

class TabularQLearningAgent {
 public:
  using HistoryBuffer = VecBuffer<uint64_t, uint8_t>;
  TabularQLearningAgent(uint64_t n_obs, uint64_t n_actions, double discount_factor, double exploration_rate)
      : n_obs_(n_obs), n_actions_(n_actions), gamma_(discount_factor), eps_(exploration_rate),
        counts_(n_obs * n_actions, 0), values_(std::make_shared<std::vector<double>>(n_obs * n_actions, 0.0)) {}

  TabularQLearningActor actor(bool training) const { return {values_, n_actions_, eps_, training}; }
  HistoryDataBound min_update_size() const { return {1, 0}; }

  // batch_update: drain every buffer in order; each step sees the observation of the step that follows it
  // (for_each_transient), the final Interrupt carries its own successor observation
  void batch_update(std::vector<HistoryBuffer> &buffers) {
    for (auto &buffer : buffers) {
      auto steps = buffer.drain_steps();
      for (size_t i = 0; i < steps.size(); ++i) {
        const auto &s = steps[i];
        std::optional<uint64_t> next_obs;
        if (s.next == SuccessorKind::Continue) {
          if (i + 1 >= steps.size()) continue;  // dangling step without a successor: skipped (map_transient)
          next_obs = steps[i + 1].observation;
        } else if (s.next == SuccessorKind::Interrupt) {
          next_obs = s.interrupt_observation;
        }
        step_update(s.observation, s.action, s.feedback, next_obs);
      }
    }
  }
  // step_update: 1/n step size towards r + gamma * max_a' Q(s', a')
  void step_update(uint64_t obs, uint8_t action, double reward, std::optional<uint64_t> next_obs) {
    auto &q = mutable_values();
    double discounted_next = 0.0;
    if (next_obs) {
      const double *row = q.data() + *next_obs * n_actions_;
      double best = row[0];
      for (uint64_t a = 1; a < n_actions_; ++a)
        if (row[a] > best) best = row[a];
      discounted_next = best * gamma_;
    }
    const size_t idx = obs * n_actions_ + action;
    counts_[idx] += 1;
    const double value = reward + discounted_next;
    const double weight = 1.0 / static_cast<double>(counts_[idx]);
    q[idx] *= 1.0 - weight;
    q[idx] += weight * value;
  }
  const std::vector<double> &values() const { return *values_; }
  const std::vector<uint64_t> &counts() const { return counts_; }
  void set_values(const double *v) { mutable_values().assign(v, v + n_obs_ * n_actions_); }

 private:
  // the reference panics when the table is updated while actors still share it (Arc::get_mut); here a new
  // table is made in that case so outstanding snapshots stay immutable
  std::vector<double> &mutable_values() {
    if (values_.use_count() > 1) values_ = std::make_shared<std::vector<double>>(*values_);
    return *values_;
  }
  uint64_t n_obs_, n_actions_;
  double gamma_, eps_;
  std::vector<uint64_t> counts_;
  std::shared_ptr<std::vector<double>> values_;
};

}  // namespace relearn
