/* rl_chacha.h — ChaCha8 block function + the draw conventions of the engine's random streams.
 *
 * The reference's generator is `Prng = rand_chacha::ChaCha8Rng`
 * (/root/reference/src/lib.rs:68).  The engine keeps that generator family and uses it the way
 * a counter-based generator is used on a GPU: every lane owns the ChaCha8 stream
 *      key    = seed_from_u64(seed) expansion (rand_core 0.6 PCG32 rule),
 *      stream = global lane id                      (`ChaCha8Rng::set_stream(lane)`),
 *      word w = 32-bit word  w % 16  of block  w / 16 (`set_word_pos(w)`),
 * so that a draw is a pure function of (seed, lane, draw index) and results are invariant to
 * how lanes are sharded over GPUs.  A Rust host reproduces a lane's stream with
 * `ChaCha8Rng::seed_from_u64(seed)` + `set_stream(lane)` + `set_word_pos`.
 *
 * State layout (rand_chacha 0.3 / djb): words 0-3 "expand 32-byte k", 4-11 key,
 * 12-13 64-bit block counter, 14-15 64-bit stream id.  8 rounds = 4 double rounds.
 *
 * Header is shared by kernels (hipcc) and the oracle (gcc): integer-only, no UB.
 */
#ifndef RL_CHACHA_H
#define RL_CHACHA_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RL_HDC __host__ __device__ static inline
#else
#define RL_HDC static inline
#endif

RL_HDC uint32_t rl_rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define RL_QR(a, b, c, d)                     \
  do {                                        \
    a += b; d ^= a; d = rl_rotl32(d, 16);     \
    c += d; b ^= c; b = rl_rotl32(b, 12);     \
    a += b; d ^= a; d = rl_rotl32(d, 8);      \
    c += d; b ^= c; b = rl_rotl32(b, 7);      \
  } while (0)

/* One ChaCha block with `double_rounds` double rounds (4 => ChaCha8, 10 => ChaCha20). */
RL_HDC void rl_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream,
                            int double_rounds, uint32_t out[16]) {
  uint32_t s0 = 0x61707865u, s1 = 0x3320646eu, s2 = 0x79622d32u, s3 = 0x6b206574u;
  uint32_t s4 = key[0], s5 = key[1], s6 = key[2], s7 = key[3];
  uint32_t s8 = key[4], s9 = key[5], s10 = key[6], s11 = key[7];
  uint32_t s12 = (uint32_t)counter, s13 = (uint32_t)(counter >> 32);
  uint32_t s14 = (uint32_t)stream, s15 = (uint32_t)(stream >> 32);
  uint32_t x0 = s0, x1 = s1, x2 = s2, x3 = s3, x4 = s4, x5 = s5, x6 = s6, x7 = s7;
  uint32_t x8 = s8, x9 = s9, x10 = s10, x11 = s11, x12 = s12, x13 = s13, x14 = s14, x15 = s15;
  for (int i = 0; i < double_rounds; ++i) {
    RL_QR(x0, x4, x8, x12);
    RL_QR(x1, x5, x9, x13);
    RL_QR(x2, x6, x10, x14);
    RL_QR(x3, x7, x11, x15);
    RL_QR(x0, x5, x10, x15);
    RL_QR(x1, x6, x11, x12);
    RL_QR(x2, x7, x8, x13);
    RL_QR(x3, x4, x9, x14);
  }
  out[0] = x0 + s0; out[1] = x1 + s1; out[2] = x2 + s2; out[3] = x3 + s3;
  out[4] = x4 + s4; out[5] = x5 + s5; out[6] = x6 + s6; out[7] = x7 + s7;
  out[8] = x8 + s8; out[9] = x9 + s9; out[10] = x10 + s10; out[11] = x11 + s11;
  out[12] = x12 + s12; out[13] = x13 + s13; out[14] = x14 + s14; out[15] = x15 + s15;
}

/* rand_core 0.6 `SeedableRng::seed_from_u64`: PCG32 output words fill the 32-byte seed. */
RL_HDC void rl_seed_from_u64(uint64_t state, uint32_t key[8]) {
  const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
  for (int i = 0; i < 8; ++i) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
}

/* rand 0.8 `Standard` f32: 24 high bits scaled into [0, 1). */
RL_HDC float rl_u32_to_unit_f32(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }
/* rand 0.8 `Standard` f64: 53 high bits scaled into [0, 1). */
RL_HDC double rl_u64_to_unit_f64(uint64_t w) { return (double)(w >> 11) * 1.1102230246251565e-16; }

/* rand 0.8.5 `UniformFloat<f64>::sample`: [1,2) mantissa trick, then `v * scale + low`
 * (two roundings, not fused). `scale` comes from rl_uniform_f64_inclusive_scale. */
RL_HDC double rl_uniform_f64_from_u64(uint64_t w, double low, double scale) {
  union { uint64_t u; double d; } c;
  c.u = (w >> 12) | 0x3ff0000000000000ULL;
  double v01 = c.d - 1.0;
  return v01 * scale + low;
}

/* rand 0.8.5 `UniformFloat<f64>::new_inclusive(low, high)` scale computation. */
RL_HDC double rl_uniform_f64_inclusive_scale(double low, double high) {
  const double max_rand = 1.0 - 2.220446049250313e-16; /* (u64::MAX >> 12) as [1,2) float - 1 */
  double scale = (high - low) / max_rand;
  for (;;) {
    if (!(scale * max_rand + low > high)) break;
    union { uint64_t u; double d; } c;
    c.d = scale;
    c.u -= 1;
    scale = c.d;
  }
  return scale;
}

#endif /* RL_CHACHA_H */
