"""A recurrent chain with stacked layers (RnnBaseConfig::num_layers = 2) on the lane-per-thread kernels
(relearn_amd/csrc/kernels_seq_stack.hip): partially observed Chain lanes, rollout, GAE with a stacked recurrent critic,
PPO steps and critic steps through time.  Prints one JSON line: device times (HIP events) per phase and kernel class.

    python scripts/stacked_period.py [lanes] [horizon] [hidden] [layers] [cell] [ppo_steps] [critic_steps]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relearn_amd as ra  # noqa: E402

arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d
N, T, H, NL, cell = arg(1, 4096), arg(2, 100), arg(3, 128), arg(4, 2), arg(5, "gru")
ppo_steps, critic_steps = arg(6, 4), arg(7, 8)

eng = ra.Engine(0)
env = ra.ChainEnv(eng, N, max_steps=100)
cls = ra.GruMlp if cell == "gru" else ra.LstmMlp
pol, cri = cls(eng, 5, 2, H, H, num_layers=NL), cls(eng, 5, 1, H, H, num_layers=NL)
pol.init(1)
cri.init(2)
popt, copt = ra.Adam(pol), ra.Adam(cri)
traj = ra.Trajectory(eng, N, T, 5)
ppo = ra.ppo_config_default()
ppo.opt_steps_per_update = ppo_steps


def period():
    t = {}
    eng.timer_begin(); ra.rollout(env, pol, traj); t["rollout_ms"] = eng.timer_end()
    eng.timer_begin(); ra.gae(traj, cri, 0.95, 0.95); t["gae_ms"] = eng.timer_end()
    eng.timer_begin(); ps = ra.ppo_update(pol, popt, traj, ppo); t["ppo_ms"] = eng.timer_end()
    eng.timer_begin(); cs = ra.critic_update(cri, copt, traj, critic_steps); t["critic_ms"] = eng.timer_end()
    return t, ps, cs


period()  # warm-up (allocates the records)
eng.profile_enable(True)
eng.profile_read(reset=True)
t, ps, cs = period()
prof = eng.profile_read(reset=True)
steps = N * T
G = 3 if cell == "gru" else 4
cell_flop = 2.0 * G * H * ((5 + H) + (NL - 1) * 2 * H)      # all layers' gate products per sample-step
head_flop = 2.0 * (H * H + 2 * H)
n_grad = ppo_steps + critic_steps
k = {kk: v[0] for kk, v in prof.items() if v[1]}
out = {"cell": cell, "lanes": N, "horizon": T, "hidden": H, "num_layers": NL, "params": pol.P, **t,
       "period_ms": sum(t.values()), "ppo_steps": ppo_steps, "critic_steps": critic_steps,
       "kernel_ms": k, "kernel_launches": {kk: v[1] for kk, v in prof.items() if v[1]},
       "forward_flop_per_sample_step": cell_flop + head_flop,
       "rollout_gflops": (cell_flop + head_flop) * steps / t["rollout_ms"] / 1e6,
       "ms_per_gradient": (t["ppo_ms"] + t["critic_ms"]) / n_grad,
       "policy_loss": [ps.loss_first, ps.loss_last], "critic_loss": [cs.loss_first, cs.loss_last]}
print(json.dumps(out))
