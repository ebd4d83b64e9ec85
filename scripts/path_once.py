"""Profiling target: a few periods of the CartPole MLP-TRPO hot path (rollout -> values/GAE -> TRPO -> 80 critic
steps) plus some standalone env steps, nothing else.  Run directly under rocprofv3 (`-- python3 scripts/path_once.py`).
usage: path_once.py [envs=65536] [periods=2] [critic_steps=80]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import relearn_amd as ra  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
periods = int(sys.argv[2]) if len(sys.argv) > 2 else 2
csteps = int(sys.argv[3]) if len(sys.argv) > 3 else 80
T = 128
eng = ra.Engine(0)
env = ra.CartPoleEnv(eng, n, max_steps=500)
pol = ra.Mlp(eng, 5, 128, 2)
cri = ra.Mlp(eng, 5, 128, 1)
pol.init(2)
cri.init(3)
opt = ra.Adam(cri)
traj = ra.Trajectory(eng, n, T, 5)
for _ in range(periods):
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    st = ra.trpo_update(pol, traj)
    cs = ra.critic_update(cri, opt, traj, csteps)
env.upload_actions(np.random.default_rng(0).integers(0, 2, size=n).astype(np.uint8))
for _ in range(10):
    env.step_resident()
eng.sync()
print("path_once: %d envs, %d periods: trpo status %d backtracks %d, critic loss %.4f -> %.4f" % (
    n, periods, st.status, st.num_backtracks, cs.loss_first, cs.loss_last))
