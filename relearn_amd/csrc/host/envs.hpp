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
