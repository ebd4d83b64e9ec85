// Cart-Pole TRPO example — the C++ counterpart of the reference's examples/cartpole-trpo.rs on the MI355X engine:
// the same agent configuration (ActorCriticConfig<TrpoConfig<MlpConfig>, ValuesOptConfig<MlpConfig>>::default), the same
// environment (CartPole::default().wrap(VisibleStepLimit::new(500))), the same loggers (display every second update,
// TensorBoard every update), the actor saved as CBOR and reloaded for evaluation.  What differs is where the work runs:
// `lanes` environments step in lock-step on the GPU instead of one per host thread.
//
//   build:  g++ -std=c++17 -O2 -I. examples/cartpole_trpo.cpp -o cartpole_trpo -Lrelearn_amd -lrelearn_hip
//           (one line, plus -Wl,-rpath,$PWD/relearn_amd)
//   train:  ./cartpole_trpo [--lanes N] [--periods P] [--out DIR]
//   eval :  ./cartpole_trpo DIR/actor.cbor
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iostream>
#include <string>

#include "relearn_amd/csrc/host/agents.hpp"

using namespace relearn;
using AgentConfig = ActorCriticConfig<TrpoConfig<MlpConfig>, ValuesOptConfig<MlpConfig>>;

static int evaluate(Engine &eng, const std::string &actor_path) {
  std::printf("Loading actor from \"%s\"\n", actor_path.c_str());
  std::ifstream in(actor_path, std::ios::binary);
  if (!in) {
    std::fprintf(stderr, "cannot open %s\n", actor_path.c_str());
    return 1;
  }
  std::vector<uint8_t> doc((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  // env.run(&actor, SimSeed::Root(0), ()).take(10_000).summarize(): here 64 lanes x 160 steps = 10,240 steps
  CartPoleLanes env(eng, 64, 500, StepLimit::Visible, /*seed_env=*/0, /*seed_actor=*/1);
  auto policy = MlpConfig().build_module(eng, 5, 2, /*seed=*/0);
  module_from_cbor(*policy, doc, eng);
  DeviceHistory history(eng, 64, 160, 5);
  check(rl_rollout(env.handle(), policy->handle(), history.handle()), eng.handle());
  uint64_t episodes = 0;
  for (uint8_t f : history.successors()) episodes += f != RL_SUCC_CONTINUE;
  std::printf("\nEvaluation Stats\nsteps %llu  episodes %llu  mean episode length (= reward) %.3f\n",
              (unsigned long long)history.num_steps(), (unsigned long long)episodes,
              (double)history.num_steps() / (double)(episodes ? episodes : 1));
  return 0;
}

int main(int argc, char **argv) {
  uint64_t lanes = 4096, periods = 50;
  std::string out_dir, actor_path;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--lanes" && i + 1 < argc) lanes = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--periods" && i + 1 < argc) periods = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
    else if (a.rfind("--", 0) == 0) {
      std::fprintf(stderr, "Usage: %s [--lanes N] [--periods P] [--out DIR] | %s saved_actor.cbor\n", argv[0], argv[0]);
      return 2;
    } else actor_path = a;
  }
  try {
    Engine eng(0);
    if (!actor_path.empty()) return evaluate(eng, actor_path);

    if (out_dir.empty()) {
      char stamp[64];
      const std::time_t now = std::time(nullptr);
      std::strftime(stamp, sizeof stamp, "%Y-%m-%d_%H-%M-%S", std::localtime(&now));
      ::mkdir("data", 0777);
      ::mkdir("data/cartpole-trpo", 0777);
      out_dir = std::string("data/cartpole-trpo/") + stamp;
    }
    ::mkdir(out_dir.c_str(), 0777);

    CartPoleLanes env(eng, lanes, 500, StepLimit::Visible, /*seed_env=*/0, /*seed_actor=*/1);
    AgentConfig agent_config;  // TRPO: 10 CG iterations, <= 15 backtracks, max KL 0.01; critic: 80 Adam steps, lr 1e-3
    std::printf("Env: CartPole::default().wrap(VisibleStepLimit::new(500)) x %llu lanes\n", (unsigned long long)lanes);
    std::printf("Training Config: num_periods %llu, steps per lane and period 128\n", (unsigned long long)periods);
    auto agent = agent_config.build_agent(env, /*seed=*/0);
    DeviceHistory history = agent->buffer(128);

    std::printf("Logging to \"%s\"\n", out_dir.c_str());
    {
      DisplayLogger<ByCounter> display(ByCounter("agent_update/count", 2));
      TensorBoardLogger<ByCounter> board(ByCounter("agent_update/count", 1), out_dir);
      TeeLogger logger(display, board);
      train_batched(*agent, env, history, periods, logger);
    }  // the loggers flush when they go out of scope

    const std::string path = out_dir + "/actor.cbor";
    std::printf("Saving actor to \"%s\"\n", path.c_str());
    const std::vector<uint8_t> doc = actor_to_cbor(env, agent->policy_module());  // the evaluation actor's state is the policy module
    std::ofstream(path, std::ios::binary).write((const char *)doc.data(), (std::streamsize)doc.size());
    std::printf("To evaluate the actor run\n%s %s\n", argv[0], path.c_str());
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
