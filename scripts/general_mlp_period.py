"""One TRPO + critic period with MlpConfig { hidden_sizes: [...] } policies and critics on CartPole lanes: what the
general per-layer path (relearn_amd/csrc/kernels_general.hip) costs next to the fused single-hidden-layer kernels.
usage: general_mlp_period.py [lanes] [horizon] [critic steps] [hidden sizes ...]   (default 16384 128 20 64 64)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relearn_amd as ra  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
hidden = [int(v) for v in sys.argv[4:]] or [64, 64]
eng = ra.Engine(0)
out = {"lanes": n, "horizon": T, "critic_steps": steps}
for name, h in (("fused_128", 128), ("general", hidden)):
    env = ra.CartPoleEnv(eng, n, max_steps=500, seed_env=0, seed_actor=1)
    pol, cri = ra.Mlp(eng, 5, h, 2), ra.Mlp(eng, 5, h, 1)
    pol.init(2)
    cri.init(3)
    opt = ra.Adam(cri)
    traj = ra.Trajectory(eng, n, T, 5)

    def period():
        t = {}
        eng.timer_begin(); ra.rollout(env, pol, traj); t["rollout_ms"] = eng.timer_end()
        eng.timer_begin(); ra.gae(traj, cri, 0.99, 0.95); t["gae_ms"] = eng.timer_end()
        eng.timer_begin(); st = ra.trpo_update(pol, traj); t["trpo_ms"] = eng.timer_end()
        t["trpo_cg_iterations"], t["trpo_backtracks"] = int(st.cg_iterations), int(st.num_backtracks)
        eng.timer_begin(); ra.critic_update(cri, opt, traj, steps); t["critic_ms"] = eng.timer_end()
        return t

    period()
    t = period()
    t["period_ms"] = sum(v for k, v in t.items() if k.endswith("_ms"))
    t["hidden_sizes"] = h if isinstance(h, list) else [h]
    t["env_steps_per_s"] = n * T / t["period_ms"] * 1e3
    out[name] = t
print(json.dumps(out))
