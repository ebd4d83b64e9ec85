// Host-side code of the C ABI under AddressSanitizer / UndefinedBehaviorSanitizer (tests/test_sanitizers_cpu.py):
// the CPU-only configuration 1 — train_parallel over Chain with the tabular Q agent, then the greedy evaluation run
// (relearn_amd/csrc/host_abi.cpp: rl_chain_tabular_q_train / _eval; reference examples/chain-tabular-q.rs) — at several
// thread counts, and the scalar environments of host/envs.hpp.  Prints a checksum so the work cannot be optimised away.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "include/relearn_hip.h"
#include "relearn_amd/csrc/host/envs.hpp"

int main() {
  double checksum = 0.0;
  for (uint64_t threads : {1ull, 3ull, 4ull}) {
    double q[10];
    uint64_t counts[10], total = 0;
    if (rl_chain_tabular_q_train(0, threads, 3, 2000, 0.2, q, counts, &total) != RL_OK) return 1;
    std::vector<uint8_t> actions(2000);
    double reward = 0.0;
    if (rl_chain_tabular_q_eval(q, 0, actions.size(), actions.data(), &reward) != RL_OK) return 2;
    for (double v : q) checksum += v;
    checksum += reward + (double)total;
  }
  // argument checks take the error path, not a wild pointer
  if (rl_chain_tabular_q_train(0, 0, 1, 10, 0.2, nullptr, nullptr, nullptr) == RL_OK) return 3;
  {
    using namespace relearn;
    Prng rng = Prng::seed_from_u64(7);
    WithLatentStepLimit<MemoryGame> env{MemoryGame(3, 4), 9};
    auto state = env.initial_state(rng);
    for (int t = 0; t < 200; ++t) {
      const uint64_t obs = env.observe(state, rng);
      auto [succ, reward] = env.step(state, (uint64_t)(t % 3), rng);
      checksum += (double)obs + reward;
      state = succ.kind == SuccessorKind::Continue ? *succ.state : env.initial_state(rng);
    }
  }
  std::printf("host sanitize ok %.6f\n", checksum);
  return 0;
}
