"""time only the recurrent rollout kernel (profiling / ablation target)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relearn_amd as ra
N, T = 16384, 100
eng = ra.Engine(0)
env = ra.ChainEnv(eng, N, max_steps=100)
pol = ra.GruMlp(eng, 5, 2); pol.init(1)
traj = ra.Trajectory(eng, N, T, 5)
ra.rollout(env, pol, traj)
eng.sync(); eng.timer_begin()
for _ in range(5): ra.rollout(env, pol, traj)
ms = eng.timer_end() / 5
print("rollout %.3f ms  %.1f TFLOP/s" % (ms, 2.0 * (128 * 384 + 128 * 128) * N * T / ms / 1e9))
