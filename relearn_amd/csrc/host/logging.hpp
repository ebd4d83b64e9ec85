// logging.hpp — the reference's statistics-logging surface (src/logging/) for the C++ host API.
//
//   StatsLogger            logging/mod.rs:25-134      log / group / with_scope / flush + the log_* conveniences
//   LogValue               logging/mod.rs:180-186     Nothing | CounterIncrement | Duration | Scalar | Index{value,size}
//   ScopedLogger           logging/mod.rs:388-447     prepends "scope/" to every id
//   ChunkLogger<C, W>      logging/chunk.rs:40-118    per-id summaries of one chunk of the time series, written out
//                                                     and reset whenever the Chunker says so (and when destroyed)
//   ChunkSummary           logging/chunk.rs:162-266   Counter{increment, initial_value} | Duration / Scalar{online
//                                                     mean, variance} | Index{counts}
//   ByCounter, ByTime      chunk_by_counter.rs:10-82, chunk_by_time.rs:7-44
//   DisplayBackend         logging/display.rs:41-166  one line per dirty id on a text stream
// Host-only code (no device work); the engine's agents log through `StatsLogger &`.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace relearn {

struct LogValue {
  enum Kind { Nothing, CounterIncrement, Duration, Scalar, Index } kind = Nothing;
  uint64_t increment = 0;  // CounterIncrement
  double value = 0.0;      // Scalar, or Duration in seconds
  size_t index = 0, size = 0;
  static LogValue nothing() { return LogValue{}; }
  static LogValue counter(uint64_t n) { LogValue v; v.kind = CounterIncrement; v.increment = n; return v; }
  static LogValue duration(double seconds) { LogValue v; v.kind = Duration; v.value = seconds; return v; }
  static LogValue scalar(double x) { LogValue v; v.kind = Scalar; v.value = x; return v; }
  static LogValue idx(size_t value, size_t size) { LogValue v; v.kind = Index; v.index = value; v.size = size; return v; }
  const char *variant_name() const {
    static const char *names[] = {"Nothing", "CounterIncrement", "Duration", "Scalar", "Index"};
    return names[kind];
  }
};

// LogError (logging/mod.rs): the convenience functions of the reference unwrap() it, i.e. it is fatal there
struct LogError : std::runtime_error { using std::runtime_error::runtime_error; };

class StatsLogger {
 public:
  virtual ~StatsLogger() = default;
  // the three "internal helpers" every logger implements (mod.rs:47-52) + flush (:55)
  virtual void group_start() {}
  virtual void group_log(const std::string &id, const LogValue &value) = 0;
  virtual void group_end() {}
  virtual void flush() {}

  // StatsLogger::log (mod.rs:41-46): a group of one
  void log(const std::string &id, const LogValue &value) {
    group_start();
    try {
      group_log(id, value);
    } catch (...) {
      group_end();
      throw;
    }
    group_end();
  }
  void log_counter_increment(const std::string &id, uint64_t increment) { log(id, LogValue::counter(increment)); }
  void log_duration(const std::string &id, double seconds) { log(id, LogValue::duration(seconds)); }
  void log_scalar(const std::string &id, double value) { log(id, LogValue::scalar(value)); }
  void log_index(const std::string &id, size_t value, size_t size) { log(id, LogValue::idx(value, size)); }
};

// `logger.group()` (mod.rs:63-69, 290-345): no flush can happen between the first and the last value of the group
class LogGroup : public StatsLogger {
 public:
  explicit LogGroup(StatsLogger &inner) : inner_(inner) { inner_.group_start(); }
  ~LogGroup() override { inner_.group_end(); }
  LogGroup(const LogGroup &) = delete;
  LogGroup &operator=(const LogGroup &) = delete;
  void group_log(const std::string &id, const LogValue &v) override { inner_.group_log(id, v); }
  // group_start / group_end of a nested `log` are no-ops: the group is already open; flush waits for its end
 private:
  StatsLogger &inner_;
};

// `logger.with_scope("policy")`: prepends "policy/" to every id
class ScopedLogger : public StatsLogger {
 public:
  ScopedLogger(StatsLogger &inner, std::string scope) : inner_(inner), prefix_(std::move(scope) + "/") {}
  void group_start() override { inner_.group_start(); }
  void group_log(const std::string &id, const LogValue &v) override { inner_.group_log(prefix_ + id, v); }
  void group_end() override { inner_.group_end(); }
  void flush() override { inner_.flush(); }

 private:
  StatsLogger &inner_;
  std::string prefix_;
};

// keeps the last value of every scalar / duration and the running counters (tests, simple front ends)
class RecordingLogger : public StatsLogger {
 public:
  std::map<std::string, double> scalars, durations;
  std::map<std::string, uint64_t> counters;
  std::map<std::string, std::vector<size_t>> indices;
  void group_log(const std::string &id, const LogValue &v) override {
    switch (v.kind) {
      case LogValue::Scalar: scalars[id] = v.value; break;
      case LogValue::Duration: durations[id] = v.value; break;
      case LogValue::CounterIncrement: counters[id] += v.increment; break;
      case LogValue::Index: {
        auto &c = indices[id];
        if (c.size() < v.size) c.resize(v.size, 0);
        c[v.index] += 1;
        break;
      }
      case LogValue::Nothing: break;
    }
  }
};

// two loggers at once: the reference implements StatsLogger for pairs `(A, B)` (mod.rs:360-386), which is how the
// examples display and write TensorBoard files at the same time
class TeeLogger : public StatsLogger {
 public:
  TeeLogger(StatsLogger &a, StatsLogger &b) : a_(a), b_(b) {}
  void group_start() override {
    a_.group_start();
    b_.group_start();
  }
  void group_log(const std::string &id, const LogValue &v) override {
    a_.group_log(id, v);
    b_.group_log(id, v);
  }
  void group_end() override {
    a_.group_end();
    b_.group_end();
  }
  void flush() override {
    a_.flush();
    b_.flush();
  }

 private:
  StatsLogger &a_, &b_;
};

// the no-op logger: `()` implements StatsLogger in the reference (mod.rs:347-358)
class NullLogger : public StatsLogger {
 public:
  void group_log(const std::string &, const LogValue &) override {}
};

template <typename F>
void log_elapsed(StatsLogger &logger, const std::string &id, F &&f) {  // StatsLogger::log_elapsed (mod.rs:103-113)
  const auto t0 = std::chrono::steady_clock::now();
  f();
  logger.log_duration(id, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

// ---------------------------------------------------------------- chunk summaries
// utils/stats.rs:119-127 (Welford): population variance = squared_residual_sum / count
struct OnlineMeanVariance {
  double mean_ = 0.0, squared_residual_sum = 0.0;
  uint64_t count = 0;
  void push(double value) {
    const double residual_pre = value - mean_;
    count += 1;
    mean_ = mean_ + residual_pre / (double)count;
    const double residual_post = value - mean_;
    squared_residual_sum = squared_residual_sum + residual_pre * residual_post;
  }
  bool has_value() const { return count > 0; }
  double mean() const { return mean_; }
  double variance() const { return squared_residual_sum / (double)count; }
  double stddev() const { return std::sqrt(variance()); }
};

struct ChunkSummary {  // chunk.rs:162-266
  LogValue::Kind kind = LogValue::Nothing;
  uint64_t increment = 0, initial_value = 0;  // Counter
  OnlineMeanVariance stats;                   // Duration (seconds), Scalar
  std::vector<size_t> counts;                 // Index

  static ChunkSummary from(const LogValue &v) {
    ChunkSummary s;
    s.kind = v.kind;
    switch (v.kind) {
      case LogValue::CounterIncrement: s.increment = v.increment; break;
      case LogValue::Duration:
      case LogValue::Scalar: s.stats.push(v.value); break;
      case LogValue::Index:
        if (v.index >= v.size) throw LogError("index value out of range");
        s.counts.assign(v.size, 0);
        s.counts[v.index] += 1;
        break;
      case LogValue::Nothing: break;
    }
    return s;
  }
  // an incompatible value is an error and is not inserted (chunk.rs:203-236)
  void push(const LogValue &v) {
    if (v.kind != kind)
      throw LogError(std::string("incompatible value type; previously ") + variant_name() + ", now " + v.variant_name());
    switch (kind) {
      case LogValue::CounterIncrement: increment += v.increment; break;
      case LogValue::Duration:
      case LogValue::Scalar: stats.push(v.value); break;
      case LogValue::Index:
        if (counts.size() != v.size)
          throw LogError("incompatible index size; previously " + std::to_string(counts.size()) + ", now " +
                         std::to_string(v.size));
        if (v.index >= v.size) throw LogError("index value out of range");
        counts[v.index] += 1;
        break;
      case LogValue::Nothing: break;
    }
  }
  // start of the next chunk (chunk.rs:239-251): a counter keeps its running total as `initial_value`
  void reset() {
    switch (kind) {
      case LogValue::CounterIncrement:
        initial_value += increment;
        increment = 0;
        break;
      case LogValue::Duration:
      case LogValue::Scalar: stats = OnlineMeanVariance(); break;
      case LogValue::Index: counts.assign(counts.size(), 0); break;
      case LogValue::Nothing: break;
    }
  }
  const char *variant_name() const { return LogValue{kind}.variant_name(); }
};

// chunk.rs:8-28
class Chunker {
 public:
  virtual ~Chunker() = default;
  virtual bool flush_group_start() { return false; }
  virtual void note_log(const std::string &, const LogValue &) {}
  virtual void note_log_summary(const ChunkSummary &) {}
  virtual bool flush_group_end() { return false; }
  virtual void note_flush() = 0;
};

using SummaryItems = std::vector<std::pair<const std::string *, const ChunkSummary *>>;
class SummaryWriter {  // chunk.rs:31-35
 public:
  virtual ~SummaryWriter() = default;
  virtual void write_summaries(const SummaryItems &summaries, double elapsed_seconds) = 0;
};

// Flush at fixed multiples of a counter, after the group that brought the counter to the multiple
// (chunk_by_counter.rs): log the counter last in its group.
class ByCounter : public Chunker {
 public:
  ByCounter(std::string counter, uint64_t interval) : counter_(std::move(counter)), interval_(interval) {
    if (interval_ == 0) throw std::invalid_argument("ByCounter: interval must be positive");
  }
  void note_log(const std::string &id, const LogValue &) override {
    if (state_ == NoFlush && id == counter_) state_ = IdMatch;
  }
  void note_log_summary(const ChunkSummary &s) override {
    if (state_ != IdMatch) return;
    if (s.kind != LogValue::CounterIncrement) throw LogError("Target ID " + counter_ + " is not a counter");
    state_ = (s.increment + s.initial_value) % interval_ == 0 ? Flush : NoFlush;
  }
  bool flush_group_end() override { return state_ == Flush; }
  void note_flush() override { state_ = NoFlush; }

 private:
  enum State { NoFlush, IdMatch, Flush } state_ = NoFlush;
  std::string counter_;
  uint64_t interval_;
};

// Flush when a group STARTS after the chunk duration has elapsed (chunk_by_time.rs:32-43; default 5 s)
class ByTime : public Chunker {
 public:
  using Clock = std::chrono::steady_clock;
  explicit ByTime(double chunk_seconds = 5.0) : chunk_seconds_(chunk_seconds), start_(Clock::now()) {}
  bool flush_group_start() override {
    return std::chrono::duration<double>(Clock::now() - start_).count() > chunk_seconds_;
  }
  void note_flush() override { start_ = Clock::now(); }

 private:
  double chunk_seconds_;
  Clock::time_point start_;
};

// chunk.rs:38-118.  Ids are kept in sorted order (the reference's BTreeMap orders Ids by their name components; here
// the '/'-joined string is compared component-wise).
struct IdLess {
  bool operator()(const std::string &a, const std::string &b) const {
    size_t i = 0, j = 0;
    while (i <= a.size() && j <= b.size()) {
      const size_t ea = std::min(a.find('/', i), a.size()), eb = std::min(b.find('/', j), b.size());
      const int c = a.compare(i, ea - i, b, j, eb - j);
      if (c != 0) return c < 0;
      if (ea == a.size() || eb == b.size()) return ea == a.size() && eb != b.size();
      i = ea + 1;
      j = eb + 1;
    }
    return false;
  }
};

template <typename C, typename W>
class ChunkLogger : public StatsLogger {
 public:
  ChunkLogger(C chunker, W writer)
      : chunker_(std::move(chunker)), writer_(std::move(writer)), chunk_start_(std::chrono::steady_clock::now()) {}
  ~ChunkLogger() override {  // "Flush when dropped"
    try {
      flush();
    } catch (...) {
    }
  }
  void group_start() override {
    if (chunker_.flush_group_start()) flush();
  }
  void group_log(const std::string &id, const LogValue &value) override {
    chunker_.note_log(id, value);
    auto it = summaries_.find(id);
    if (it == summaries_.end()) {
      it = summaries_.emplace(id, Node{ChunkSummary::from(value), true}).first;
    } else {
      it->second.dirty = true;  // Node::push marks the node before the (possibly failing) insert, chunk.rs:144-147
      it->second.summary.push(value);
    }
    chunker_.note_log_summary(it->second.summary);
  }
  void group_end() override {
    if (chunker_.flush_group_end()) flush();
  }
  void flush() override {
    SummaryItems items;
    for (auto &kv : summaries_)
      if (kv.second.dirty) items.emplace_back(&kv.first, &kv.second.summary);
    writer_.write_summaries(items, std::chrono::duration<double>(std::chrono::steady_clock::now() - chunk_start_).count());
    for (auto &kv : summaries_) {
      kv.second.dirty = false;
      kv.second.summary.reset();
    }
    chunk_start_ = std::chrono::steady_clock::now();
    chunker_.note_flush();
  }
  W &writer() { return writer_; }
  C &chunker() { return chunker_; }

 private:
  struct Node {
    ChunkSummary summary;
    bool dirty;
  };
  C chunker_;
  W writer_;
  std::map<std::string, Node, IdLess> summaries_;
  std::chrono::steady_clock::time_point chunk_start_;
};

// ---------------------------------------------------------------- text display (display.rs:41-166, utils/fmt.rs)
namespace fmt {
inline std::string fixed(double v, int prec) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.*f", prec, v);
  return buf;
}
// PrettyPrint<f64> at a precision: exponent form outside (1e-4, 1e6)
inline std::string pretty(double v, int prec) {
  const double m = std::fabs(v);
  if ((m >= 1e6 || m <= 1e-4) && v != 0.0) {
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.*e", prec, v);
    // Rust's LowerExp writes "1.234e6" / "1.234e-5": no '+', no zero padding of the exponent
    std::string s(buf);
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e), ex = s.substr(e + 1);
    const bool neg = !ex.empty() && ex[0] == '-';
    if (!ex.empty() && (ex[0] == '+' || ex[0] == '-')) ex.erase(0, 1);
    while (ex.size() > 1 && ex[0] == '0') ex.erase(0, 1);
    return mant + "e" + (neg ? "-" : "") + ex;
  }
  return fixed(v, prec);
}
// Debug formatting of a Duration: the largest of s / ms / µs / ns whose integer part is non-zero
inline std::string duration(double seconds, int prec) {
  if (seconds >= 1.0) return fixed(seconds, prec) + "s";
  if (seconds >= 1e-3) return fixed(seconds * 1e3, prec) + "ms";
  if (seconds >= 1e-6) return fixed(seconds * 1e6, prec) + "\xc2\xb5s";
  return fixed(seconds * 1e9, prec) + "ns";
}
inline std::string frequency(double hz, int prec) {  // utils/fmt.rs:38-54
  if (hz >= 1e3 && hz < 1e6) return pretty(hz / 1e3, prec) + "kHz";
  if (hz >= 1e6 && hz < 1e9) return pretty(hz / 1e6, prec) + "MHz";
  if (hz >= 1e9 && hz < 1e12) return pretty(hz / 1e9, prec) + "GHz";
  return pretty(hz, prec) + "Hz";
}
}  // namespace fmt

// one summary as the reference's DisplaySummary prints it, without the terminal colours
inline std::string display_summary(const ChunkSummary &s, double elapsed_seconds) {
  std::string out;
  switch (s.kind) {
    case LogValue::Nothing: break;
    case LogValue::CounterIncrement:
      out = std::to_string(s.initial_value + s.increment) + "  (+" + std::to_string(s.increment) + ")";
      if (s.increment > 5) {  // a rate only when the chunk saw several increments (display.rs:84-94)
        const double period = elapsed_seconds / (double)s.increment;
        out += "  " + fmt::frequency(1.0 / period, 2) + "  " + fmt::duration(period, 3);
      }
      break;
    case LogValue::Duration:
      if (s.stats.has_value()) {
        out = fmt::duration(s.stats.mean(), 4);
        if (s.stats.count > 1) out += " (\xcf\x83 " + fmt::duration(s.stats.stddev(), 4) + ")";
        out += " " + fmt::fixed(s.stats.mean() / elapsed_seconds * 100.0, 2) + "%";
      }
      break;
    case LogValue::Scalar:
      if (s.stats.has_value()) {
        out = fmt::pretty(s.stats.mean(), 3);
        if (s.stats.count > 1) out += " (\xcf\x83 " + fmt::pretty(s.stats.stddev(), 3) + ")";
      }
      break;
    case LogValue::Index: {
      size_t n = 0;
      for (size_t c : s.counts) n += c;
      out = "(n " + std::to_string(n) + ")  [";
      for (size_t i = 0; i < s.counts.size(); ++i) {
        if (i) out += " ";
        out += std::to_string(n ? s.counts[i] * 100 / n : 0);
      }
      out += "]%";
      break;
    }
  }
  return out;
}

class DisplayBackend : public SummaryWriter {
 public:
  explicit DisplayBackend(std::ostream &os = std::cout) : os_(&os) {}
  void write_summaries(const SummaryItems &summaries, double elapsed_seconds) override {
    *os_ << "\n";
    for (auto &it : summaries) {
      std::string id = *it.first;
      if (id.size() < 24) id.resize(24, ' ');  // "{:<24} {}"
      *os_ << id << " " << display_summary(*it.second, elapsed_seconds) << "\n";
    }
    os_->flush();
  }

 private:
  std::ostream *os_;
};

// DisplayLogger<C = ByTime> (display.rs:10-19)
template <typename C = ByTime>
class DisplayLogger : public ChunkLogger<C, DisplayBackend> {
 public:
  explicit DisplayLogger(C chunker = C(), std::ostream &os = std::cout)
      : ChunkLogger<C, DisplayBackend>(std::move(chunker), DisplayBackend(os)) {}
};

// ---------------------------------------------------------------- TensorBoard event files (tensorboard.rs:40-124)
// The reference writes through the third-party `tensorboard-rs` crate (not under /root/reference); what reaches the
// disk is TensorFlow's public event-file format, restated here: a stream of records
//     u64 length | u32 masked_crc32c(length) | bytes | u32 masked_crc32c(bytes)
// each holding one `Event` protobuf {wall_time = 1: double, step = 2: int64, file_version = 3: string,
// summary = 5: Summary{value = 1: {tag = 1: string, simple_value = 2: float, histo = 5: HistogramProto{min = 1,
// max = 2, num = 3, sum = 4, sum_squares = 5: double; bucket_limit = 6, bucket = 7: packed double}}}}.
// Per chunk (tensorboard.rs:86-123): counter -> scalar(initial_value + increment), duration / scalar -> scalar(mean),
// index -> histogram with bucket boundaries half way between the integers; step = index of the chunk.
namespace tfevents {
inline uint32_t crc32c(const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}
inline uint32_t masked_crc(const uint8_t *p, size_t n) {
  const uint32_t c = crc32c(p, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
struct Proto {
  std::string b;
  void varint(uint64_t v) {
    while (v >= 0x80) {
      b.push_back((char)(v | 0x80));
      v >>= 7;
    }
    b.push_back((char)v);
  }
  void key(int field, int wire) { varint(((uint64_t)field << 3) | (uint64_t)wire); }
  void f64(int field, double v) {
    key(field, 1);
    b.append((const char *)&v, 8);  // little-endian hosts only (x86-64)
  }
  void f32(int field, float v) {
    key(field, 5);
    b.append((const char *)&v, 4);
  }
  void i64(int field, int64_t v) {
    key(field, 0);
    varint((uint64_t)v);
  }
  void bytes(int field, const std::string &s) {
    key(field, 2);
    varint(s.size());
    b += s;
  }
  void packed_f64(int field, const std::vector<double> &v) {
    key(field, 2);
    varint(v.size() * 8);
    b.append((const char *)v.data(), v.size() * 8);
  }
};
}  // namespace tfevents

class TensorBoardBackend : public SummaryWriter {
 public:
  // creates `<log_dir>/events.out.tfevents.<unix time>.relearn` (the directory must exist)
  explicit TensorBoardBackend(const std::string &log_dir) {
    const double now = wall_time();
    path_ = log_dir + "/events.out.tfevents." + std::to_string((long long)now) + ".relearn";
    file_ = std::fopen(path_.c_str(), "wb");
    if (!file_) throw std::runtime_error("cannot create " + path_);
    tfevents::Proto ev;
    ev.f64(1, now);
    ev.bytes(3, "brain.Event:2");
    record(ev.b);
  }
  TensorBoardBackend(TensorBoardBackend &&o) noexcept
      : path_(std::move(o.path_)), file_(o.file_), summary_index_(o.summary_index_) {
    o.file_ = nullptr;
  }
  TensorBoardBackend(const TensorBoardBackend &) = delete;
  TensorBoardBackend &operator=(const TensorBoardBackend &) = delete;
  ~TensorBoardBackend() override {
    if (file_) std::fclose(file_);
  }
  const std::string &path() const { return path_; }

  void write_summaries(const SummaryItems &summaries, double) override {
    for (auto &it : summaries) {
      const ChunkSummary &s = *it.second;
      tfevents::Proto value;
      value.bytes(1, *it.first);
      switch (s.kind) {
        case LogValue::CounterIncrement: value.f32(2, (float)(s.initial_value + s.increment)); break;
        case LogValue::Duration:
        case LogValue::Scalar:
          if (!s.stats.has_value()) continue;
          value.f32(2, (float)s.stats.mean());
          break;
        case LogValue::Index: {
          double num = 0, sum = 0, sq = 0;
          std::vector<double> limits, buckets;
          for (size_t i = 0; i < s.counts.size(); ++i) {
            const double n = (double)s.counts[i];
            num += n;
            sum += (double)(i * s.counts[i]);
            sq += (double)(i * i * s.counts[i]);
            limits.push_back((double)i + 0.5);
            buckets.push_back(n);
          }
          tfevents::Proto h;
          h.f64(1, -0.5);
          h.f64(2, (double)s.counts.size() - 0.5);
          h.f64(3, num);
          h.f64(4, sum);
          h.f64(5, sq);
          h.packed_f64(6, limits);
          h.packed_f64(7, buckets);
          value.bytes(5, h.b);
          break;
        }
        case LogValue::Nothing: continue;
      }
      tfevents::Proto summary, ev;
      summary.bytes(1, value.b);
      ev.f64(1, wall_time());
      ev.i64(2, (int64_t)summary_index_);
      ev.bytes(5, summary.b);
      record(ev.b);
    }
    summary_index_ += 1;
    std::fflush(file_);
  }

 private:
  static double wall_time() {
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
  }
  void record(const std::string &data) {
    uint8_t head[12];
    const uint64_t len = data.size();
    std::memcpy(head, &len, 8);
    const uint32_t c1 = tfevents::masked_crc(head, 8);
    std::memcpy(head + 8, &c1, 4);
    const uint32_t c2 = tfevents::masked_crc((const uint8_t *)data.data(), data.size());
    if (std::fwrite(head, 1, 12, file_) != 12 || std::fwrite(data.data(), 1, data.size(), file_) != data.size() ||
        std::fwrite(&c2, 1, 4, file_) != 4)
      throw std::runtime_error("short write to " + path_);
  }
  std::string path_;
  std::FILE *file_ = nullptr;
  size_t summary_index_ = 0;
};

// TensorBoardLogger<C = ByTime> (tensorboard.rs:10-19)
template <typename C = ByTime>
class TensorBoardLogger : public ChunkLogger<C, TensorBoardBackend> {
 public:
  TensorBoardLogger(C chunker, const std::string &log_dir)
      : ChunkLogger<C, TensorBoardBackend>(std::move(chunker), TensorBoardBackend(log_dir)) {}
};

}  // namespace relearn
