"""Device time of each update pass of a several-hidden-layer policy / critic on a 16,384 x 128 trajectory.
usage: gen_passes.py [lanes] [hidden sizes ...] [Relu|Tanh|Sigmoid|Identity]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import relearn_amd as ra
n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 128
act = "Relu"
args = sys.argv[2:]
if args and not args[-1].isdigit():
    act = args.pop()
hidden = [int(v) for v in args] or [64, 64]
eng = ra.Engine(0)
env = ra.CartPoleEnv(eng, n)
fpol = ra.Mlp(eng, 5, 128, 2); fpol.init(2)
pol = ra.Mlp(eng, 5, hidden, 2, act, "Identity"); pol.init(4)
cri = ra.Mlp(eng, 5, hidden, 1, act, "Identity"); cri.init(3)
traj = ra.Trajectory(eng, n, T, 5)
ra.rollout(env, fpol, traj); ra.gae(traj, cri, 0.99, 0.95)
p0 = pol.get_params()
vec = np.random.default_rng(0).normal(size=pol.P).astype(np.float32)
def timed(f, reps=5):
    f(); eng.sync(); eng.timer_begin()
    for _ in range(reps): f()
    return eng.timer_end() / reps
print("hidden", hidden, act, "samples", n * T)
print("policy gradient  %.3f ms" % timed(lambda: ra.policy_gradient(pol, traj)))
print("policy loss / KL %.3f ms" % timed(lambda: ra.policy_loss_kl(pol, traj, p0)))
print("Fisher-vector    %.3f ms" % timed(lambda: ra.policy_fvp(pol, traj, vec, 1e-5)))
print("critic gradient  %.3f ms" % timed(lambda: ra.critic_gradient(cri, traj)))
