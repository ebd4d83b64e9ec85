"""The recurrent configuration of scripts/gru_config5.py with the LSTM cell (ChainConfig<LstmConfig, MlpConfig>): Chain under
LatentStepLimit(100), 16,384 lanes, T = 100, PPO 10 steps + critic 80 steps per period.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import relearn_amd as ra

n, T = 16384, 100
eng = ra.Engine(0)
env = ra.ChainEnv(eng, n, max_steps=100, seed_env=0, seed_actor=1)
pol, cri = ra.LstmMlp(eng, 5, 2), ra.LstmMlp(eng, 5, 1)
pol.init(2); cri.init(3)
popt, copt = ra.Adam(pol), ra.Adam(cri)
traj = ra.Trajectory(eng, n, T, 5)
cfg = ra.ppo_config_default()

def period(timing=None):
    def lap(name, f):
        eng.sync(); t0 = time.perf_counter(); r = f(); eng.sync()
        if timing is not None: timing[name] = (time.perf_counter() - t0) * 1e3
        return r
    lap("rollout_ms", lambda: ra.rollout(env, pol, traj))
    lap("gae_ms", lambda: ra.gae(traj, cri, 0.95, 0.95))
    lap("ppo_ms", lambda: ra.ppo_update(pol, popt, traj, cfg))
    lap("critic_ms", lambda: ra.critic_update(cri, copt, traj, 80))

period()
eng.profile_enable(True); eng.profile_read(reset=True)
tm = {}
eng.sync(); t0 = time.perf_counter(); period(tm); eng.sync()
tm["period_ms"] = (time.perf_counter() - t0) * 1e3
prof = eng.profile_read(reset=True)
out = {"cell": "lstm", "lanes": n, "horizon": T, "env_steps_per_period": n * T, **tm,
       "env_steps_per_s": n * T / tm["period_ms"] * 1e3,
       "kernel_ms_per_launch": {k: v[0] / v[1] for k, v in prof.items() if v[1] > 0},
       "kernel_launches_per_period": {k: v[1] for k, v in prof.items() if v[1] > 0}}
print(json.dumps(out))
