/* prng.c — oracle restatement of `Prng = rand_chacha::ChaCha8Rng` and the rand 0.8.5 sampling
 * rules the reference path uses.  TEST INFRASTRUCTURE (see oracle.h).
 *
 * Third-party provenance: rand 0.8.5 / rand_core 0.6.3 / rand_chacha 0.3.1 are pinned in
 * /root/reference/relearn_experiments/Cargo.lock but their sources are NOT under /root/reference.
 * The block function is pinned by published known-answer vectors; the sampling conventions are
 * restated from the crates' published algorithms and are "parity unpinned" against a Rust run.
 *
 * Call sites followed:
 *   seed_from_u64   examples/cartpole-trpo.rs:49, simulation/mod.rs:141-147
 *   from_rng        simulation/train.rs:102-103
 *   Uniform f64     envs/cartpole.rs:105-111
 *   gen::<f32>()    envs/chain.rs:91
 *   gen::<f64>(), gen_range   agents/tabular.rs:223-225
 *   gen_bool        torch/agents/dqn.rs:366
 */
#include "oracle.h"
#include "../include/rl_chacha.h"

#include <string.h>

static void refill(oracle_prng *r) {
  /* rand_chacha fills a 4-block buffer per call; counters c, c+1, c+2, c+3 */
  for (int b = 0; b < 4; ++b) rl_chacha_block(r->key, r->counter + (uint64_t)b, r->stream, 4, r->buf + 16 * b);
  r->counter += 4;
}

void oracle_prng_from_seed(oracle_prng *r, const uint32_t key[8]) {
  memcpy(r->key, key, sizeof(r->key));
  r->counter = 0;
  r->stream = 0;
  r->index = 64;
}

void oracle_prng_seed_from_u64(oracle_prng *r, uint64_t seed) {
  uint32_t key[8];
  rl_seed_from_u64(seed, key);
  oracle_prng_from_seed(r, key);
}

uint32_t oracle_prng_next_u32(oracle_prng *r) {
  if (r->index >= 64) {
    refill(r);
    r->index = 0;
  }
  return r->buf[r->index++];
}

/* rand_core 0.6 BlockRng::next_u64 */
uint64_t oracle_prng_next_u64(oracle_prng *r) {
  const uint32_t len = 64;
  uint32_t index = r->index;
  if (index < len - 1) {
    r->index += 2;
    return ((uint64_t)r->buf[index + 1] << 32) | (uint64_t)r->buf[index];
  } else if (index >= len) {
    refill(r);
    r->index = 2;
    return ((uint64_t)r->buf[1] << 32) | (uint64_t)r->buf[0];
  } else {
    uint64_t x = r->buf[len - 1];
    refill(r);
    r->index = 1;
    uint64_t y = r->buf[0];
    return (y << 32) | x;
  }
}

/* SeedableRng::from_rng: fill the 32-byte seed from the source (8 little-endian words) */
void oracle_prng_from_rng(oracle_prng *out, oracle_prng *src) {
  uint32_t key[8];
  for (int i = 0; i < 8; ++i) key[i] = oracle_prng_next_u32(src);
  oracle_prng_from_seed(out, key);
}

void oracle_prng_set_stream(oracle_prng *r, uint64_t stream) {
  /* rand_chacha set_stream keeps the word position; callers here always follow with set_word_pos */
  r->stream = stream;
  r->index = 64;
}

void oracle_prng_set_word_pos(oracle_prng *r, uint64_t word_pos) {
  uint64_t block = word_pos / 16;
  r->counter = block;
  refill(r);
  r->index = (uint32_t)(word_pos % 16);
}

/* get_word_pos: flat index of the next unread word of the stream */
uint64_t oracle_prng_word_pos(const oracle_prng *r) { return r->counter * 16 - 64 + (uint64_t)r->index; }

float oracle_prng_gen_f32(oracle_prng *r) { return rl_u32_to_unit_f32(oracle_prng_next_u32(r)); }
double oracle_prng_gen_f64(oracle_prng *r) { return rl_u64_to_unit_f64(oracle_prng_next_u64(r)); }

/* UniformInt<usize>::sample_single (64-bit target): widening multiply + rejection zone */
uint64_t oracle_prng_gen_range_u64(oracle_prng *r, uint64_t low, uint64_t high) {
  uint64_t range = high - low;
  int lz = __builtin_clzll(range);
  uint64_t zone = (range << lz) - 1;
  for (;;) {
    uint64_t v = oracle_prng_next_u64(r);
    unsigned __int128 m = (unsigned __int128)v * (unsigned __int128)range;
    uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
    if (lo <= zone) return low + hi;
  }
}

/* Bernoulli::new(p) + sample */
int oracle_prng_gen_bool(oracle_prng *r, double p) {
  if (p == 1.0) return 1; /* ALWAYS_TRUE: no draw */
  uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
  uint64_t v = oracle_prng_next_u64(r);
  return v < p_int;
}

double oracle_prng_uniform_f64_inclusive(oracle_prng *r, double low, double high) {
  double scale = rl_uniform_f64_inclusive_scale(low, high);
  return rl_uniform_f64_from_u64(oracle_prng_next_u64(r), low, scale);
}
