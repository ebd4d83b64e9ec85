// Exercises relearn_amd/csrc/host/logging.hpp (the reference's src/logging/ semantics) and prints what each chunk
// wrote as JSON lines; tests/test_host_logging.py checks them against a Python restatement of chunk.rs.
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "relearn_amd/csrc/host/logging.hpp"

using namespace relearn;

// a writer that prints every flushed chunk as one JSON object
class JsonWriter : public SummaryWriter {
 public:
  void write_summaries(const SummaryItems &items, double) override {
    std::printf("{\"chunk\": %d, \"items\": [", chunk_++);
    bool first = true;
    for (auto &it : items) {
      const ChunkSummary &s = *it.second;
      std::printf("%s{\"id\": \"%s\", \"kind\": \"%s\"", first ? "" : ", ", it.first->c_str(), s.variant_name());
      first = false;
      switch (s.kind) {
        case LogValue::CounterIncrement:
          std::printf(", \"increment\": %llu, \"initial_value\": %llu", (unsigned long long)s.increment,
                      (unsigned long long)s.initial_value);
          break;
        case LogValue::Duration:
        case LogValue::Scalar:
          std::printf(", \"count\": %llu, \"mean\": %.17g, \"stddev\": %.17g", (unsigned long long)s.stats.count,
                      s.stats.mean(), s.stats.stddev());
          break;
        case LogValue::Index: {
          std::printf(", \"counts\": [");
          for (size_t i = 0; i < s.counts.size(); ++i) std::printf("%s%zu", i ? ", " : "", s.counts[i]);
          std::printf("]");
          break;
        }
        default: break;
      }
      std::printf("}");
    }
    std::printf("]}\n");
  }

 private:
  int chunk_ = 0;
};

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "chunks";
  if (mode == "chunks") {
    // the shape of train_parallel's logging: per-step values, counter last, chunked on agent_update/count
    ChunkLogger<ByCounter, JsonWriter> logger(ByCounter("agent_update/count", 2), JsonWriter());
    double x = 0.5;
    for (int period = 0; period < 5; ++period) {
      {
        LogGroup group(logger);  // no flush inside a group even when the counter hits the multiple first
        ScopedLogger scoped(group, "agent_update");
        scoped.log_counter_increment("count", 1);
        scoped.log_duration("time", 0.001 * (period + 1));
      }
      ScopedLogger policy(logger, "policy");
      for (int k = 0; k <= period; ++k) {
        x = x * 1.7 - 0.3 * k;
        policy.log_scalar("entropy", x);
      }
      logger.log_index("worker0/step/action", (size_t)(period % 3), 3);
      if (period == 3) logger.log_counter_increment("sim/step/count", 7);
    }
    // an incompatible value is an error and leaves the summary unchanged
    try {
      logger.log_scalar("agent_update/count", 1.0);
      std::printf("{\"error\": null}\n");
    } catch (const LogError &e) {
      std::printf("{\"error\": \"%s\"}\n", e.what());
    }
    try {
      logger.log_index("worker0/step/action", 0, 4);
      std::printf("{\"error\": null}\n");
    } catch (const LogError &e) {
      std::printf("{\"error\": \"%s\"}\n", e.what());
    }
    // the destructor flushes what is left
  } else if (mode == "display") {
    std::ostringstream os;
    {
      ChunkLogger<ByCounter, DisplayBackend> logger(ByCounter("n", 1000), DisplayBackend(os));
      logger.log_scalar("policy/entropy", 0.6931);
      logger.log_scalar("policy/entropy", 0.6);
      logger.log_scalar("tiny", 2.5e-5);
      logger.log_scalar("huge", 1.25e7);
      logger.log_index("a/action", 1, 2);
      logger.log_index("a/action", 1, 2);
      logger.log_index("a/action", 0, 2);
      logger.log_counter_increment("b", 3);
      logger.log_duration("z/time", 0.25);
      logger.log_scalar("a", 1.0);  // sorts before "a/action": a shorter id is smaller
    }
    std::fputs(os.str().c_str(), stdout);
  } else if (mode == "tensorboard" && argc > 2) {
    TensorBoardLogger<ByCounter> logger(ByCounter("n", 1), argv[2]);
    std::printf("%s\n", logger.writer().path().c_str());
    for (int i = 0; i < 3; ++i) {
      LogGroup g(logger);
      g.log_scalar("policy/entropy", 0.5 + i);
      g.log_duration("time", 0.125 * (i + 1));
      g.log_index("action", (size_t)(i % 2), 2);
      g.log_index("action", 1, 2);
      g.log_counter_increment("n", 1);
    }
  } else if (mode == "bytime") {
    ChunkLogger<ByTime, JsonWriter> logger(ByTime(0.0), JsonWriter());  // every group start after > 0 s flushes
    logger.log_scalar("x", 1.0);
    logger.log_scalar("x", 3.0);  // flushes the chunk holding 1.0 first, then logs
  }
  return 0;
}
