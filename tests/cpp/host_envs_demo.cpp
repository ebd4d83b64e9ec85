// The scalar host-side MemoryGame (relearn_amd/csrc/host/envs.hpp, src/envs/memory.rs) driven like Steps::step drives an
// environment: prints one line per step; tests/test_oracle_memory.py compares the trace with the oracle's lane 0.
#include <cstdio>
#include <cstdlib>

#include "relearn_amd/csrc/host/envs.hpp"

using namespace relearn;

int main(int argc, char **argv) {
  const uint64_t num_actions = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2;
  const uint64_t history_len = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 3;
  const uint64_t max_steps = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 0;
  const uint64_t seed = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0;
  const int steps = argc > 5 ? std::atoi(argv[5]) : 40;
  Prng rng = Prng::seed_from_u64(seed);
  auto run = [&](auto env) {
    auto state = env.initial_state(rng);
    for (int t = 0; t < steps; ++t) {
      const uint64_t obs = env.observe(state, rng);
      const uint64_t action = (uint64_t)(t * 7 + 3) % num_actions;
      auto [succ, reward] = env.step(state, action, rng);
      std::printf("%llu %llu %.1f %d\n", (unsigned long long)obs, (unsigned long long)action, reward, (int)succ.kind);
      state = succ.kind == SuccessorKind::Continue ? *succ.state : env.initial_state(rng);
    }
  };
  if (max_steps) run(WithLatentStepLimit<MemoryGame>{MemoryGame(num_actions, history_len), max_steps});
  else run(MemoryGame(num_actions, history_len));
  return 0;
}
