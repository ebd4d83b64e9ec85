"""Accuracy of the device Fisher-vector product against the f64 oracle, next to the f32 oracle's (diagnostic)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import oracle as O  # noqa: E402
import relearn_amd as ra  # noqa: E402

n, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 128)
H = 128
PS = O.MlpShape(5, H, 2)
eng = ra.Engine(0)
if len(sys.argv) > 3:
    eng.set_kernel_variant(int(sys.argv[3]))
env = ra.CartPoleEnv(eng, n, max_steps=500)
pol = ra.Mlp(eng, 5, H, 2)
pol.init(2)
traj = ra.Trajectory(eng, n, T, 5)
ra.rollout(env, pol, traj)
want = traj.read_all()
x, a = O.flat_samples(want)
pp = pol.get_params()
rng = np.random.default_rng(1)
for trial in range(3):
    v = rng.standard_normal(pol.P).astype(np.float32)
    hv_d = ra.policy_fvp(pol, traj, v, 0.0)
    hv32 = np.zeros_like(pp)
    O.lib().oracle_policy_fvp_f32(PS, O.f32p(pp), O.f32p(x), len(a), O.f32p(v), 0.0, O.f32p(hv32))
    hv64 = np.zeros(pol.P, np.float64)
    O.lib().oracle_policy_fvp_f64(PS, O.f64p(pp.astype(np.float64)), O.f64p(x.astype(np.float64)), len(a),
                                  O.f64p(v.astype(np.float64)), 0.0, O.f64p(hv64))
    sc = np.abs(hv64).max()
    print("trial %d: device vs f64 %.3g   f32 oracle vs f64 %.3g   (relative to max|Hv|)" % (
        trial, np.abs(hv_d - hv64).max() / sc, np.abs(hv32 - hv64).max() / sc))
# gradient of the surrogate for comparison
adv = rng.standard_normal(n * T).astype(np.float32)
traj.write(ra.TRAJ_ADVANTAGES, adv.reshape(T, n))
g_d, loss_d, _ = ra.policy_gradient(pol, traj)
g32 = np.zeros_like(pp)
l32 = C.c_float()
O.lib().oracle_policy_grad_f32(PS, O.f32p(pp), O.f32p(x), O.i64p(a), O.f32p(adv), len(a), O.f32p(g32), C.byref(l32))
g64 = np.zeros(pol.P, np.float64)
l64 = C.c_double()
O.lib().oracle_policy_grad_f64(PS, O.f64p(pp.astype(np.float64)), O.f64p(x.astype(np.float64)), O.i64p(a),
                               O.f64p(adv.astype(np.float64)), len(a), O.f64p(g64), C.byref(l64))
sc = np.abs(g64).max()
print("gradient: device vs f64 %.3g   f32 oracle vs f64 %.3g" % (np.abs(g_d - g64).max() / sc, np.abs(g32 - g64).max() / sc))
# per-block view of the FVP error: which parameter block carries it
v = rng.standard_normal(pol.P).astype(np.float32)
hv_d = ra.policy_fvp(pol, traj, v, 0.0)
hv64 = np.zeros(pol.P, np.float64)
O.lib().oracle_policy_fvp_f64(PS, O.f64p(pp.astype(np.float64)), O.f64p(x.astype(np.float64)), len(a),
                              O.f64p(v.astype(np.float64)), 0.0, O.f64p(hv64))
blocks = {"W1": (0, 5 * H), "b1": (5 * H, 6 * H), "W2": (6 * H, 8 * H), "b2": (8 * H, 8 * H + 2)}
for k, (lo, hi) in blocks.items():
    e = np.abs(hv_d[lo:hi] - hv64[lo:hi])
    print("  block %s: max err %.3g (block max %.3g), rel to global max %.3g" % (
        k, e.max(), np.abs(hv64[lo:hi]).max(), e.max() / np.abs(hv64).max()))
for k, (lo, hi) in blocks.items():
    e = np.abs(g_d[lo:hi] - g64[lo:hi])
    print("  grad block %s: max err %.3g (block max %.3g)" % (k, e.max(), np.abs(g64[lo:hi]).max()))
print("loss: device %.9g f32 %.9g f64 %.9g" % (loss_d, l32.value, l64.value))
e = (g_d[:5 * H].astype(np.float64) - g64[:5 * H]).reshape(H, 5)
eb = g_d[5 * H:6 * H].astype(np.float64) - g64[5 * H:6 * H]
W2 = pp[6 * H:8 * H].reshape(2, H).astype(np.float64)
print("corr(err_b1, W2sum) %.4f  corr(err_b1, W2diff) %.4f" % (np.corrcoef(eb, W2[0] + W2[1])[0, 1], np.corrcoef(eb, W2[0] - W2[1])[0, 1]))
print("err_b1 / W2sum (first 6):", (eb / (W2[0] + W2[1]))[:6])
print("err_b1 / W2diff (first 6):", (eb / (W2[0] - W2[1]))[:6])
print("err W1 per k, unit 0..2:", e[:3])
jm = np.unravel_index(np.abs(e).argmax(), e.shape)
print("worst W1 entry", jm, e[jm], "g64", g64[:5 * H].reshape(H, 5)[jm], "W2 col", W2[:, jm[0]], "W1 row", pp[:5 * H].reshape(H, 5)[jm[0]], "b1", pp[5 * H + jm[0]])
bad = np.where(np.abs(eb) > 1e-9)[0]
print("units with b1 error > 1e-9:", bad, eb[bad])
