// prng.hpp — `Prng = rand_chacha::ChaCha8Rng` (reference src/lib.rs:68) for host-side code, with the rand 0.8.5
// sampling rules the path uses.  Block function and seed expansion come from include/rl_chacha.h (shared with the
// kernels, pinned by RFC 7539 / eSTREAM known answers in tests/test_detmath_prng.py).
#pragma once
#include <array>
#include <cstdint>

#include "../../../include/rl_chacha.h"

namespace relearn {

class Prng {
 public:
  Prng() = default;
  // SeedableRng::seed_from_u64
  static Prng seed_from_u64(uint64_t seed) {
    Prng r;
    rl_seed_from_u64(seed, r.key_.data());
    return r;
  }
  // SeedableRng::from_rng: the child's 32-byte seed is the next 8 words of the parent
  static Prng from_rng(Prng &parent) {
    Prng r;
    for (auto &k : r.key_) k = parent.next_u32();
    return r;
  }
  void set_stream(uint64_t stream) {
    stream_ = stream;
    index_ = kBuf;
  }
  void set_word_pos(uint64_t word_pos) {
    next_block_ = word_pos / 16;
    refill();
    index_ = static_cast<uint32_t>(word_pos % 16);
  }

  uint32_t next_u32() {
    if (index_ >= kBuf) {
      refill();
      index_ = 0;
    }
    return buf_[index_++];
  }
  // rand_core BlockRng::next_u64: two consecutive words, low word first, also across a buffer refill
  uint64_t next_u64() {
    if (index_ + 1 < kBuf) {
      uint64_t lo = buf_[index_], hi = buf_[index_ + 1];
      index_ += 2;
      return (hi << 32) | lo;
    }
    if (index_ >= kBuf) {
      refill();
      index_ = 2;
      return (static_cast<uint64_t>(buf_[1]) << 32) | buf_[0];
    }
    uint64_t lo = buf_[kBuf - 1];
    refill();
    index_ = 1;
    return (static_cast<uint64_t>(buf_[0]) << 32) | lo;
  }
  // Standard: f32 from 24 bits, f64 from 53 bits
  float gen_f32() { return rl_u32_to_unit_f32(next_u32()); }
  double gen_f64() { return rl_u64_to_unit_f64(next_u64()); }
  // Rng::gen_range(low..high) for usize on a 64-bit target: widening multiply with a rejection zone
  uint64_t gen_range(uint64_t low, uint64_t high) {
    const uint64_t range = high - low;
    const uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
      const unsigned __int128 wide = static_cast<unsigned __int128>(next_u64()) * range;
      if (static_cast<uint64_t>(wide) <= zone) return low + static_cast<uint64_t>(wide >> 64);
    }
  }
  // Rng::gen_bool(p) = Bernoulli::new(p).sample: p == 1 draws nothing
  bool gen_bool(double p) {
    if (p == 1.0) return true;
    const uint64_t threshold = static_cast<uint64_t>(p * 18446744073709551616.0);
    return next_u64() < threshold;
  }
  // Uniform::new_inclusive(low, high).sample for f64
  double uniform_inclusive(double low, double high) {
    return rl_uniform_f64_from_u64(next_u64(), low, rl_uniform_f64_inclusive_scale(low, high));
  }

 private:
  static constexpr uint32_t kBuf = 64;  // rand_chacha refills four blocks at a time
  void refill() {
    for (uint32_t b = 0; b < 4; ++b) rl_chacha_block(key_.data(), next_block_ + b, stream_, 4, buf_.data() + 16 * b);
    next_block_ += 4;
  }
  std::array<uint32_t, 8> key_{};
  std::array<uint32_t, kBuf> buf_{};
  uint64_t next_block_ = 0, stream_ = 0;
  uint32_t index_ = kBuf;
};

}  // namespace relearn
