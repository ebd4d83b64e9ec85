// envs.hpp — scalar host-side mirrors of the reference's environment interface (src/envs/mod.rs:76-127,
// 257-269): `initial_state(rng)`, `observe(state, rng)`, `step(state, action, rng) -> (Successor<State>, reward)`.
// Used by the CPU-only plumbing configuration (Chain + tabular Q) and as the scalar view of what the lanes do.
#pragma once
#include <cstdint>
#include <optional>
#include <utility>

#include "prng.hpp"

namespace relearn {

// Successor<T> (src/envs/mod.rs:257-269)
enum class SuccessorKind : uint8_t { Continue = 0, Terminate = 1, Interrupt = 2 };

template <typename T>
struct Successor {
  SuccessorKind kind = SuccessorKind::Terminate;
  std::optional<T> state;  // present for Continue and Interrupt

  static Successor Continue(T s) { return {SuccessorKind::Continue, std::move(s)}; }
  static Successor Terminate() { return {SuccessorKind::Terminate, std::nullopt}; }
  static Successor Interrupt(T s) { return {SuccessorKind::Interrupt, std::move(s)}; }
  bool episode_done() const { return kind != SuccessorKind::Continue; }
  // then_interrupt_if (src/envs/mod.rs:348-362)
  template <typename F>
  Successor then_interrupt_if(F &&f) && {
    if (kind == SuccessorKind::Continue && f(*state)) kind = SuccessorKind::Interrupt;
    return std::move(*this);
  }
  template <typename U, typename F>
  Successor<U> map(F &&f) && {
    if (kind == SuccessorKind::Terminate) return Successor<U>::Terminate();
    Successor<U> out;
    out.kind = kind;
    out.state = f(std::move(*state));
    return out;
  }
};

// Chain (src/envs/chain.rs:20-106): n states in a line, 2 actions, 0.2 slip probability
struct Chain {
  using State = uint64_t;
  using Observation = uint64_t;
  using Action = uint8_t;  // Move::Left = 0, Move::Right = 1
  uint64_t size = 5;
  double discount_factor = 0.95;

  uint64_t num_observations() const { return size; }
  uint64_t num_actions() const { return 2; }
  State initial_state(Prng &) const { return 0; }
  Observation observe(const State &s, Prng &) const { return s; }
  std::pair<Successor<State>, double> step(State s, Action a, Prng &rng) const {
    if (rng.gen_f32() < 0.2f) a = a ? 0 : 1;  // Move::invert
    if (a == 0) return {Successor<State>::Continue(0), 2.0};
    if (s == size - 1) return {Successor<State>::Continue(s), 10.0};
    return {Successor<State>::Continue(s + 1), 0.0};
  }
};

// MemoryGame (src/envs/memory.rs:24-115): start in one of `num_actions` states at random, walk through `history_len`
// further states whatever the action, then answer: +1 iff the action equals the initial state, else -1 (terminal)
struct MemoryGame {
  using State = std::pair<uint64_t, uint64_t>;  // (current_state, initial_state)
  using Observation = uint64_t;
  using Action = uint64_t;
  uint64_t num_actions_ = 2, history_len = 1;  // MemoryGame::default
  double discount_factor = 1.0;

  MemoryGame() = default;
  MemoryGame(uint64_t num_actions, uint64_t history) : num_actions_(num_actions), history_len(history) {}
  uint64_t num_observations() const { return num_actions_ + history_len; }
  uint64_t num_actions() const { return num_actions_; }
  State initial_state(Prng &rng) const {
    const uint64_t s = rng.gen_range(0, num_actions_);
    return {s, s};
  }
  Observation observe(const State &s, Prng &) const { return s.first; }
  std::pair<Successor<State>, double> step(State s, Action a, Prng &) const {
    if (s.first == num_actions_ + history_len - 1) return {Successor<State>::Terminate(), a == s.second ? 1.0 : -1.0};
    const uint64_t next = s.first < num_actions_ ? num_actions_ : s.first + 1;
    return {Successor<State>::Continue({next, s.second}), 0.0};
  }
};

// Wrapped<E, LatentStepLimit> (src/envs/wrappers/step_limit.rs:13-89): the limit is not observable
template <typename E>
struct WithLatentStepLimit {
  struct State {
    typename E::State inner;
    uint64_t steps_remaining;
  };
  using Observation = typename E::Observation;
  using Action = typename E::Action;
  E inner;
  uint64_t max_steps_per_episode = 100;

  uint64_t num_observations() const { return inner.num_observations(); }
  uint64_t num_actions() const { return inner.num_actions(); }
  State initial_state(Prng &rng) const { return {inner.initial_state(rng), max_steps_per_episode}; }
  Observation observe(const State &s, Prng &rng) const { return inner.observe(s.inner, rng); }
  std::pair<Successor<State>, double> step(State s, Action a, Prng &rng) const {
    auto [succ, reward] = inner.step(std::move(s.inner), a, rng);
    const uint64_t left = s.steps_remaining - 1;
    auto wrapped = std::move(succ).template map<State>([&](typename E::State in) { return State{std::move(in), left}; });
    return {std::move(wrapped).then_interrupt_if([](const State &n) { return n.steps_remaining == 0; }), reward};
  }
};

}  // namespace relearn
