"""Print the full-size update-kernel errors against the f64 oracle (what tests/test_gpu_fullsize.py asserts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
import relearn_amd as ra
N, T, H = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 128, 128
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)
def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
eng = ra.Engine(0)
env = ra.CartPoleEnv(eng, N, max_steps=500, seed_env=0, seed_actor=1)
pol, cri = ra.Mlp(eng, 5, H, 2), ra.Mlp(eng, 5, H, 1)
pol.init(2); cri.init(3)
traj = ra.Trajectory(eng, N, T, 5)
ra.rollout(env, pol, traj); ra.gae(traj, cri, 0.99, 0.95)
data = traj.read_all()
x, a = O.flat_samples(data)
adv, rtg = traj.read(ra.TRAJ_ADVANTAGES).reshape(-1), traj.read(ra.TRAJ_RETURNS).reshape(-1)
pp, cp = pol.get_params(), cri.get_params()
for variant in (0, 1):
    eng.set_kernel_variant(variant)
    g_d, loss_d, ent_d = ra.policy_gradient(pol, traj)
    g64, l64 = O.grad_f64_mt("policy", PS, pp, x, a.astype(np.uint8), adv)
    print("variant", variant, "policy grad rel err", rel_err(g_d, g64), "loss", loss_d, l64)
    v = np.random.default_rng(7).standard_normal(len(pp)).astype(np.float32)
    h_d = ra.policy_fvp(pol, traj, v, 0.0)
    h64, _ = O.grad_f64_mt("fvp", PS, pp, x, v=v)
    print("  fvp rel err", rel_err(h_d, h64))
    gc_d, lc_d = ra.critic_gradient(cri, traj)
    gc64, lc64 = O.grad_f64_mt("critic", CS, cp, x, aux=rtg)
    print("  critic grad rel err", rel_err(gc_d, gc64), "loss", lc_d, lc64, abs(lc_d - lc64) / lc64)
    k = np.argmax(np.abs(gc_d - gc64)); print("  worst critic entry", k, gc_d[k], gc64[k], np.abs(gc64).max())
# ---- where does the policy gradient's distance from f64 come from?
eng.set_kernel_variant(0)
g_d, _, _ = ra.policy_gradient(pol, traj)
g64, _ = O.grad_f64_mt("policy", PS, pp, x, a.astype(np.uint8), adv)
g32, _ = O.grad_f64_mt("policy", PS, pp, x, a.astype(np.uint8), adv, f32_samples=True)
print("max|g|", np.abs(g64).max(), "f32 oracle rel err", rel_err(g32, g64))
segs = {"W1": slice(0, 640), "b1": slice(640, 768), "W2": slice(768, 1024), "b2": slice(1024, 1026)}
for k, sl in segs.items():
    print("  ", k, "max|g|", np.abs(g64[sl]).max(), "dev err", np.abs(g_d[sl] - g64[sl]).max(), "f32 oracle err", np.abs(g32[sl] - g64[sl]).max())
advc = (adv - adv.mean()).astype(np.float32)
traj.write(ra.TRAJ_ADVANTAGES, advc.reshape(T, N))
g_d2, _, _ = ra.policy_gradient(pol, traj)
g642, _ = O.grad_f64_mt("policy", PS, pp, x, a.astype(np.uint8), advc)
print("centred advantages: max|g|", np.abs(g642).max(), "dev rel err", rel_err(g_d2, g642), "abs", np.abs(g_d2 - g642).max())
