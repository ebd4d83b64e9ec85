/* rl_detmath.h — deterministic elementary functions shared by host and device.
 *
 * Part of the published contract of the relearn-mi355x C ABI (include/relearn_hip.h):
 * every transcendental on the rollout path is one of these functions, written with
 * nothing but IEEE-754 +,-,*,/ and fma, so the SAME bit pattern is produced by
 *   - the gfx950 kernels (hipcc, built with -ffp-contract=off),
 *   - any host caller that wants to reproduce a trajectory (a Rust shim may transcribe
 *     them 1:1 with f64::mul_add / f32::mul_add),
 *   - the parity oracle under oracle/ (gcc, built with -ffp-contract=off).
 *
 * Why they exist: the reference calls platform libm (`f64::sin_cos`,
 * /root/reference/src/envs/cartpole.rs:322) and libtorch (`log_softmax`, `exp`,
 * /root/reference/src/torch/distributions/categorical.rs:31,53).  Device libm and glibc
 * differ in the last ulp, CartPole is chaotic, and a categorical sample flips when p moves by
 * one ulp across the uniform draw — so "bit-exact action indices" (BASELINE.json north_star)
 * is only reachable with a shared definition.  Accuracy vs glibc is pinned by
 * tests/test_detmath_prng.py (sincos <= 1 ulp on the CartPole range, expf/logf <= 2 ulp).
 *
 * Algorithms: Cody-Waite 3-term reduction by pi/2 with fma + the classic fdlibm kernel
 * polynomials (public domain, Sun Microsystems 1993) for f64 sin/cos; fdlibm-style expf/logf.
 * All functions are branch-light and safe for one-env-per-lane execution.
 */
#ifndef RL_DETMATH_H
#define RL_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RL_HD __host__ __device__ static inline
#else
#define RL_HD static inline
#endif

/* ---- bit casts ------------------------------------------------------------------------- */
RL_HD uint64_t rl_f64_bits(double x) { union { double d; uint64_t u; } c; c.d = x; return c.u; }
RL_HD double rl_f64_from_bits(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
RL_HD uint32_t rl_f32_bits(float x) { union { float f; uint32_t u; } c; c.f = x; return c.u; }
RL_HD float rl_f32_from_bits(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

RL_HD double rl_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
RL_HD float rl_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* ---- f64 sin/cos ----------------------------------------------------------------------- */
/* fdlibm __kernel_sin / __kernel_cos on |x| <= pi/4 (tail y = 0). */
RL_HD double rl_kernel_sin(double x) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ax = x < 0.0 ? -x : x;
  if (ax < 1.4901161193847656e-08) return x; /* |x| < 2^-26: sin x == x to the last bit, keeps -0.0 */
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x + v * (S1 + z * r);
}

RL_HD double rl_kernel_cos(double x) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double ax = x < 0.0 ? -x : x;
  if (ax < 0.3) {
    return 1.0 - (0.5 * z - z * r);
  }
  double qx = ax > 0.78125 ? 0.28125 : ax * 0.25; /* fdlibm uses x/4 truncated; any qx near it is valid */
  /* keep qx exactly representable with few bits: truncate low mantissa word */
  qx = rl_f64_from_bits(rl_f64_bits(qx) & 0xffffffff00000000ULL);
  double hz = 0.5 * z - qx;
  double a = 1.0 - qx;
  return a - (hz - z * r);
}

/* sin and cos of x, |x| <= ~1e5 (beyond that the 3-term reduction loses accuracy but the
 * result stays deterministic).  Mirrors the call `phys.pole_angle.sin_cos()`. */
RL_HD void rl_sincos(double x, double *s, double *c) {
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double P1 = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
  const double P2 = 6.07710050630396597660e-11; /* next 33 bits */
  const double P3 = 2.02226624871116645580e-21; /* next 33 bits */
  const double P3T = 8.47842766036889956997e-32; /* tail */
  double ax = x < 0.0 ? -x : x;
  if (ax <= 0.78539816339744830962) {
    *s = rl_kernel_sin(x);
    *c = rl_kernel_cos(x);
    return;
  }
  /* n = nearest integer to x * 2/pi, computed with the 2^52+2^51 trick (round-to-nearest-even) */
  const double SHIFT = 6755399441055744.0;
  double fn = (x * INV_PIO2 + SHIFT) - SHIFT;
  double r = rl_fma(-fn, P1, x);
  r = rl_fma(-fn, P2, r);
  r = rl_fma(-fn, P3, r);
  r = rl_fma(-fn, P3T, r);
  int64_t n = (int64_t)fn;
  double ks = rl_kernel_sin(r), kc = rl_kernel_cos(r);
  switch ((int)(n & 3)) {
    case 0: *s = ks; *c = kc; break;
    case 1: *s = kc; *c = -ks; break;
    case 2: *s = -ks; *c = -kc; break;
    default: *s = -kc; *c = ks; break;
  }
}

/* ---- f32 exp / log --------------------------------------------------------------------- */
/* expf: fdlibm e_expf.c scheme. Returns 0 for x < -103.97, +inf for x > 88.72. NaN -> NaN. */
RL_HD float rl_expf(float x) {
  const float LN2HI = 6.9314575195e-01f, /* 0x3f317200 */
      LN2LO = 1.4286067653e-06f,         /* 0x35bfbe8e */
      INVLN2 = 1.4426950216e+00f,        /* 0x3fb8aa3b */
      P1 = 1.6666625440e-1f,             /* 0x3e2aaa8f */
      P2 = -2.7667332906e-3f;            /* 0xbb355215 */
  if (x != x) return x;
  if (x > 88.7228317f) return rl_f32_from_bits(0x7f800000u);
  if (x < -103.972084f) return 0.0f;
  float ax = x < 0.0f ? -x : x;
  float hi, lo;
  int k;
  if (ax > 0.34657359f) { /* 0.5 ln2 */
    float kf = x * INVLN2 + (x < 0.0f ? -0.5f : 0.5f);
    k = (int)kf; /* truncation toward zero */
    hi = x - (float)k * LN2HI;
    lo = (float)k * LN2LO;
    x = hi - lo;
  } else if (ax > 2.9802322e-08f) { /* 2^-25 */
    k = 0;
    hi = x;
    lo = 0.0f;
  } else {
    return 1.0f + x;
  }
  float xx = x * x;
  float c = x - xx * (P1 + xx * P2);
  float y = 1.0f + (x * c / (2.0f - c) - lo + hi);
  if (k == 0) return y;
  /* scale by 2^k without denormal/overflow loss: split when k is extreme */
  if (k < -125) {
    y = y * rl_f32_from_bits((uint32_t)(k + 100 + 127) << 23);
    return y * 7.888609052210118e-31f; /* 2^-100 */
  }
  if (k > 127) {
    y = y * rl_f32_from_bits((uint32_t)(k - 100 + 127) << 23);
    return y * 1.2676506002282294e+30f; /* 2^100 */
  }
  return y * rl_f32_from_bits((uint32_t)(k + 127) << 23);
}

/* logf: fdlibm e_logf.c scheme for normal positive x (the only inputs the path produces:
 * a sum of exponentials in [1, A]). x <= 0 -> -inf / NaN, subnormals are pre-scaled. */
RL_HD float rl_logf(float x) {
  const float LN2HI = 6.9313812256e-01f, /* 0x3f317180 */
      LN2LO = 9.0580006145e-06f,         /* 0x3717f7d1 */
      LG1 = 0.66666662693f,              /* 0xaaaaaa.0p-24 */
      LG2 = 0.40000972152f,              /* 0xccce13.0p-25 */
      LG3 = 0.28498786688f,              /* 0x91e9ee.0p-25 */
      LG4 = 0.24279078841f;              /* 0xf89e26.0p-26 */
  uint32_t ix = rl_f32_bits(x);
  int k = 0;
  if (x != x) return x;
  if (ix < 0x00800000u || (ix >> 31)) {
    if ((ix << 1) == 0) return rl_f32_from_bits(0xff800000u); /* log(+-0) = -inf */
    if (ix >> 31) return rl_f32_from_bits(0x7fc00000u);       /* log(-#) = NaN */
    k -= 25;
    x *= 33554432.0f; /* 2^25 */
    ix = rl_f32_bits(x);
  } else if (ix >= 0x7f800000u) {
    return x;
  } else if (ix == 0x3f800000u) {
    return 0.0f;
  }
  /* reduce x into [sqrt(2)/2, sqrt(2)] */
  ix += 0x3f800000u - 0x3f3504f3u;
  k += (int)(ix >> 23) - 0x7f;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  x = rl_f32_from_bits(ix);
  float f = x - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  float w = z * z;
  float t1 = w * (LG2 + w * LG4);
  float t2 = z * (LG1 + w * LG3);
  float R = t2 + t1;
  float hfsq = 0.5f * f * f;
  float dk = (float)k;
  return s * (hfsq + R) + dk * LN2LO - hfsq + f + dk * LN2HI;
}

/* ---- f32 sigmoid / tanh (GRU gates, torch gru_cell: sigmoid_, tanh_) ---------------------- */
/* e^y for y <= 0 without a division: y = k ln2 + r, |r| <= ln2 / 2, degree-7 Taylor polynomial of e^r by Horner
 * with fma (truncation < 6e-9 relative), scaled by 2^k through the exponent field.  Returns 0 below -87 (the
 * callers add 1 to the result).  One rounding per fma, so host and device agree bit for bit. */
RL_HD float rl_exp_nonpos(float y) {
  if (!(y > -87.0f)) return y != y ? y : 0.0f;
  const float kf = (float)(int)(y * 1.4426950216e+00f - 0.5f); /* round to nearest, y <= 0 */
  float r = __builtin_fmaf(kf, -6.9314575195e-01f, y);         /* ln2 high part, exact product for |k| < 2^11 */
  r = __builtin_fmaf(kf, -1.4286067653e-06f, r);
  float p = 1.9841270114e-04f;               /* 1/5040 */
  p = __builtin_fmaf(p, r, 1.3888889225e-03f); /* 1/720 */
  p = __builtin_fmaf(p, r, 8.3333337680e-03f); /* 1/120 */
  p = __builtin_fmaf(p, r, 4.1666667908e-02f); /* 1/24 */
  p = __builtin_fmaf(p, r, 1.6666667163e-01f); /* 1/6 */
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  return p * rl_f32_from_bits((uint32_t)((int)kf + 127) << 23);
}

/* sigmoid(x) = 1 / (1 + e^-|x|) for x >= 0 and e^-|x| / (1 + e^-|x|) for x < 0: one exp, one division */
RL_HD float rl_sigmoidf(float x) {
  if (x != x) return x;
  const float t = rl_exp_nonpos(x < 0.0f ? x : -x);
  const float s = 1.0f / (1.0f + t);
  return x < 0.0f ? t * s : s;
}

/* tanh: odd polynomial below 0.25 (truncation error < 4e-11 relative), (1 - e^-2|x|) / (1 + e^-2|x|) above;
 * exactly +-1 once e^-2|x| underflows below half an ulp of 1 (|x| > 9.02).  Both branches are cheap enough to be
 * evaluated unconditionally on the GPU (no divergence). */
RL_HD float rl_tanhf(float x) {
  if (x != x) return x;
  const float ax = x < 0.0f ? -x : x;
  const float x2 = ax * ax;
  float p = -8.8632355e-03f;                 /* -1382/155925 */
  p = 2.1869488e-02f + x2 * p;               /*  62/2835 */
  p = -5.3968254e-02f + x2 * p;              /* -17/315 */
  p = 1.3333334e-01f + x2 * p;               /*  2/15 */
  p = -3.3333334e-01f + x2 * p;              /* -1/3 */
  const float small = ax + ax * (x2 * p);
  const float t = rl_exp_nonpos(-2.0f * ax);
  const float big = (1.0f - t) / (1.0f + t);
  const float y = ax < 0.25f ? small : big;
  return rl_f32_from_bits(rl_f32_bits(y) | (rl_f32_bits(x) & 0x80000000u)); /* odd, tanh(-0) = -0 */
}

#endif /* RL_DETMATH_H */
