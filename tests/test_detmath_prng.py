"""Pins the shared deterministic math (include/rl_detmath.h) against libm and the ChaCha8 generator
(include/rl_chacha.h, oracle/prng.c) against published known-answer vectors.  CPU only."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle as O

L = O.lib()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE_C = r'''
#include <math.h>
#include <stdint.h>
#include "include/rl_detmath.h"
#include "include/rl_chacha.h"
void probe_sincos(const double *x, int n, double *s, double *c) { for (int i = 0; i < n; ++i) rl_sincos(x[i], s + i, c + i); }
void probe_expf(const float *x, int n, float *y) { for (int i = 0; i < n; ++i) y[i] = rl_expf(x[i]); }
void probe_logf(const float *x, int n, float *y) { for (int i = 0; i < n; ++i) y[i] = rl_logf(x[i]); }
void probe_sigmoidf(const float *x, int n, float *y) { for (int i = 0; i < n; ++i) y[i] = rl_sigmoidf(x[i]); }
void probe_tanhf(const float *x, int n, float *y) { for (int i = 0; i < n; ++i) y[i] = rl_tanhf(x[i]); }
void probe_block(const uint32_t *key, uint64_t counter, uint64_t stream, int dr, uint32_t *out) { rl_chacha_block(key, counter, stream, dr, out); }
void probe_seed(uint64_t s, uint32_t *key) { rl_seed_from_u64(s, key); }
double probe_scale(double lo, double hi) { return rl_uniform_f64_inclusive_scale(lo, hi); }
'''


@pytest.fixture(scope="module")
def probe():
    d = tempfile.mkdtemp()
    src = os.path.join(d, "probe.c")
    open(src, "w").write(PROBE_C)
    so = os.path.join(d, "probe.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-I", ROOT, src, "-o", so,
                           "-lm"])
    return C.CDLL(so)


def ulps64(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def ulps32(a, b):
    b32 = b.astype(np.float32)
    return np.abs(a.astype(np.float64) - b) / np.spacing(np.abs(b32)).astype(np.float64)


def test_sincos_accuracy(probe):
    rng = np.random.default_rng(0)
    for scale, bound in ((0.3, 1.0), (0.8, 1.5), (4.0, 2.0), (1000.0, 2.0)):  # CartPole range: <= 1 ulp
        x = rng.uniform(-scale, scale, 200000)
        s, c = np.zeros_like(x), np.zeros_like(x)
        probe.probe_sincos(O.f64p(x), len(x), O.f64p(s), O.f64p(c))
        # numpy's sin/cos are correctly rounded to < 1 ulp; longdouble gives a tighter reference
        rs, rc = np.sin(x.astype(np.longdouble)), np.cos(x.astype(np.longdouble))
        es = np.abs(s - rs).astype(np.float64) / np.spacing(np.abs(rs.astype(np.float64)))
        ec = np.abs(c - rc).astype(np.float64) / np.spacing(np.abs(rc.astype(np.float64)))
        assert es.max() <= bound and ec.max() <= bound, (scale, es.max(), ec.max())
    x = np.array([0.0, -0.0, 1e-300, np.pi / 4, -np.pi / 4])
    s, c = np.zeros_like(x), np.zeros_like(x)
    probe.probe_sincos(O.f64p(x), len(x), O.f64p(s), O.f64p(c))
    assert s[0] == 0.0 and c[0] == 1.0 and np.signbit(s[1]) and s[2] == 1e-300


def test_expf_logf_accuracy(probe):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-20, 20, 300000), rng.uniform(-87, 88, 100000)]).astype(np.float32)
    y = np.zeros_like(x)
    probe.probe_expf(O.f32p(x), len(x), O.f32p(y))
    assert ulps32(y, np.exp(x.astype(np.float64))).max() <= 1.0
    sp = np.array([0.0, -200.0, 200.0, np.nan, -1e-30], np.float32)
    out = np.zeros_like(sp)
    probe.probe_expf(O.f32p(sp), len(sp), O.f32p(out))
    assert out[0] == 1.0 and out[1] == 0.0 and np.isinf(out[2]) and np.isnan(out[3]) and out[4] == 1.0
    x = np.concatenate([rng.uniform(1e-3, 4, 300000), rng.uniform(1e-30, 1e30, 100000)]).astype(np.float32)
    y = np.zeros_like(x)
    probe.probe_logf(O.f32p(x), len(x), O.f32p(y))
    ref = np.log(x.astype(np.float64))
    m = np.abs(ref) > 1e-3
    assert ulps32(y[m], ref[m]).max() <= 1.0
    sp = np.array([1.0, 2.0, 0.0, -1.0], np.float32)
    out = np.zeros_like(sp)
    probe.probe_logf(O.f32p(sp), len(sp), O.f32p(out))
    assert out[0] == 0.0 and out[1] == np.float32(np.log(2.0)) and out[2] == -np.inf and np.isnan(out[3])


def test_sigmoid_tanh_accuracy(probe):
    """GRU gate functions: within 3 ulp of the exact value over the whole range, exact limits and symmetry"""
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-1, 1, 200000), rng.uniform(-12, 12, 200000), rng.uniform(-100, 100, 50000),
                        rng.uniform(-0.3, 0.3, 100000)]).astype(np.float32)
    y = np.zeros_like(x)
    probe.probe_sigmoidf(O.f32p(x), len(x), O.f32p(y))
    ref = 1.0 / (1.0 + np.exp(-x.astype(np.longdouble)))
    m = ref > 1e-35
    assert ulps32(y[m], ref[m].astype(np.float64)).max() <= 3.0
    probe.probe_tanhf(O.f32p(x), len(x), O.f32p(y))
    ref = np.tanh(x.astype(np.longdouble)).astype(np.float64)
    m = np.abs(ref) > 0
    assert ulps32(y[m], ref[m]).max() <= 3.0
    sp = np.array([0.0, -0.0, 30.0, -30.0, 200.0, -200.0, np.nan, 1e-20], np.float32)
    out = np.zeros_like(sp)
    probe.probe_tanhf(O.f32p(sp), len(sp), O.f32p(out))
    assert out[0] == 0.0 and np.signbit(out[1]) and out[2] == 1.0 and out[3] == -1.0 and out[4] == 1.0
    assert out[5] == -1.0 and np.isnan(out[6]) and out[7] == np.float32(1e-20)
    probe.probe_sigmoidf(O.f32p(sp), len(sp), O.f32p(out))
    assert out[0] == 0.5 and out[1] == 0.5 and out[4] == 1.0 and out[5] == 0.0 and np.isnan(out[6])
    a = rng.uniform(-8, 8, 1000).astype(np.float32)
    ya, yb = np.zeros_like(a), np.zeros_like(a)
    probe.probe_tanhf(O.f32p(a), len(a), O.f32p(ya))
    probe.probe_tanhf(O.f32p(-a), len(a), O.f32p(yb))
    assert np.array_equal(ya, -yb)


def test_chacha_known_answers(probe):
    key = np.zeros(8, np.uint32)
    out = np.zeros(16, np.uint32)
    probe.probe_block(key.ctypes.data_as(C.c_void_p), C.c_uint64(0), C.c_uint64(0), 10, out.ctypes.data_as(C.c_void_p))
    # RFC 7539 / djb ChaCha20 zero key, zero nonce, block 0
    assert out.tobytes().hex() == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                   "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    probe.probe_block(key.ctypes.data_as(C.c_void_p), C.c_uint64(0), C.c_uint64(0), 4, out.ctypes.data_as(C.c_void_p))
    # eSTREAM ChaCha8, 256-bit zero key, zero IV
    assert out.tobytes().hex() == ("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
                                   "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")


def test_prng_stream_structure(probe):
    """BlockRng semantics of rand_core 0.6: 64-word buffer, next_u64 pairing incl. the straddling case,
    set_word_pos addressing, from_rng forking."""
    r = O.Prng()
    L.oracle_prng_seed_from_u64(C.byref(r), 0)
    key = np.zeros(8, np.uint32)
    probe.probe_seed(C.c_uint64(0), key.ctypes.data_as(C.c_void_p))
    assert list(r.key) == key.tolist()
    words = [L.oracle_prng_next_u32(C.byref(r)) for _ in range(130)]
    blk = np.zeros(16, np.uint32)
    for b in range(8):
        probe.probe_block(key.ctypes.data_as(C.c_void_p), C.c_uint64(b), C.c_uint64(0), 4,
                          blk.ctypes.data_as(C.c_void_p))
        assert words[16 * b:16 * b + 16] == blk.tolist()
    # next_u64 = lo | hi << 32 of consecutive words; straddling the 64-word buffer end keeps the order
    r2 = O.Prng()
    L.oracle_prng_seed_from_u64(C.byref(r2), 0)
    for _ in range(63):
        L.oracle_prng_next_u32(C.byref(r2))
    v = L.oracle_prng_next_u64(C.byref(r2))
    assert v == words[63] | (words[64] << 32)
    assert L.oracle_prng_next_u32(C.byref(r2)) == words[65]
    # set_word_pos
    L.oracle_prng_set_word_pos(C.byref(r2), 100)
    assert L.oracle_prng_next_u32(C.byref(r2)) == words[100]
    # from_rng consumes 8 words as the child's key
    r3, child = O.Prng(), O.Prng()
    L.oracle_prng_seed_from_u64(C.byref(r3), 0)
    L.oracle_prng_from_rng(C.byref(child), C.byref(r3))
    assert list(child.key) == words[:8]
    assert L.oracle_prng_next_u32(C.byref(r3)) == words[8]


def test_sampling_rules(probe):
    r = O.Prng()
    L.oracle_prng_seed_from_u64(C.byref(r), 42)
    f = [L.oracle_prng_gen_f32(C.byref(r)) for _ in range(2000)]
    assert 0.0 <= min(f) and max(f) < 1.0 and abs(np.mean(f) - 0.5) < 0.03
    d = [L.oracle_prng_gen_f64(C.byref(r)) for _ in range(2000)]
    assert 0.0 <= min(d) and max(d) < 1.0
    g = [L.oracle_prng_gen_range_u64(C.byref(r), 0, 2) for _ in range(4000)]
    assert set(g) == {0, 1} and abs(np.mean(g) - 0.5) < 0.04
    g = [L.oracle_prng_gen_range_u64(C.byref(r), 3, 10) for _ in range(2000)]
    assert min(g) == 3 and max(g) == 9
    assert all(L.oracle_prng_gen_bool(C.byref(r), 1.0) for _ in range(10))
    assert not any(L.oracle_prng_gen_bool(C.byref(r), 0.0) for _ in range(10))
    b = [L.oracle_prng_gen_bool(C.byref(r), 0.2) for _ in range(5000)]
    assert abs(np.mean(b) - 0.2) < 0.03
    # Uniform::new_inclusive(-0.05, 0.05): scale chosen so that the largest mantissa maps to <= high
    probe.probe_scale.restype = C.c_double
    scale = probe.probe_scale(C.c_double(-0.05), C.c_double(0.05))
    max_rand = 1.0 - 2.0 ** -52
    assert scale * max_rand + -0.05 <= 0.05
    u = [L.oracle_prng_uniform_f64_inclusive(C.byref(r), -0.05, 0.05) for _ in range(5000)]
    assert -0.05 <= min(u) and max(u) <= 0.05 and abs(np.mean(u)) < 0.003
