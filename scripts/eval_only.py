"""Time the loss/KL evaluation pass (k_policy_bf16<EVAL>) alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import relearn_amd as ra
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
eng = ra.Engine(0)
env = ra.CartPoleEnv(eng, n); pol = ra.Mlp(eng, 5, 128, 2); pol.init(2); cri = ra.Mlp(eng, 5, 128, 1); cri.init(3)
traj = ra.Trajectory(eng, n, 128, 5)
ra.rollout(env, pol, traj); ra.gae(traj, cri, 0.99, 0.95)
p0 = pol.get_params()
ra.policy_loss_kl(pol, traj, p0)
eng.sync(); eng.profile_enable(True); eng.profile_read(reset=True)
for _ in range(reps):
    ra.policy_loss_kl(pol, traj, p0)
pr = eng.profile_read()
ms, cnt = pr["policy_fused"]
print("policy_fused launches %d avg %.4f ms (each loss_kl call = 1 INIT + 1 EVAL)" % (cnt, ms / cnt))
