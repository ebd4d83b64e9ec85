"""Run only critic gradient steps on a 65,536 x 128 trajectory (profiling target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relearn_amd as ra
n, T, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = ra.Engine(0)
if len(sys.argv) > 3: eng.set_kernel_variant(int(sys.argv[3]))
env = ra.CartPoleEnv(eng, n)
pol = ra.Mlp(eng, 5, 128, 2); pol.init(2)
cri = ra.Mlp(eng, 5, 128, 1); cri.init(3)
traj = ra.Trajectory(eng, n, T, 5)
ra.rollout(env, pol, traj); ra.gae(traj, cri, 0.99, 0.95)
opt = ra.Adam(cri)
ra.critic_update(cri, opt, traj, 3)
eng.sync(); eng.timer_begin()
st = ra.critic_update(cri, opt, traj, steps)
ms = eng.timer_end()
print("critic step: %.3f ms = %.2f us (loss %.3f -> %.3f)" % (ms / steps, 1e3 * ms / steps, st.loss_first, st.loss_last))
