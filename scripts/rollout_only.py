"""Time the fused rollout alone.  usage: rollout_only.py [envs] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relearn_amd as ra  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = ra.Engine(0)
env = ra.CartPoleEnv(eng, n, max_steps=500)
pol = ra.Mlp(eng, 5, 128, 2)
pol.init(2)
traj = ra.Trajectory(eng, n, 128, 5)
for _ in range(3):
    ra.rollout(env, pol, traj)
eng.sync()
eng.timer_begin()
for _ in range(reps):
    ra.rollout(env, pol, traj)
ms = eng.timer_end() / reps
print("rollout %d lanes x 128 steps: %.3f ms (G=%s)" % (n, ms, os.environ.get("RELEARN_ROLLOUT_G", "auto")))
