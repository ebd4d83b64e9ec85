"""Rehearsal of the library configuration of a multi-rank bench.py process: torch imported FIRST (its bundled ROCm
libraries get mapped), then the engine, then a one-rank RCCL communicator (RELEARN_FORCE_RCCL=1) and a few collective
updates.  Prints which libamdhip64 / librccl / libhsa objects the process ended up with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch  # noqa: F401
os.environ["RELEARN_FORCE_RCCL"] = "1"
import numpy as np
import relearn_amd as ra
eng = ra.Engine(0)
if order != "torch_first":
    import torch  # noqa: F401
eng.comm_init(0, 1, ra.comm_unique_id())
env = ra.CartPoleEnv(eng, 512, max_steps=50, seed_env=0, seed_actor=1)
pol, cri = ra.Mlp(eng, 5, 128, 2), ra.Mlp(eng, 5, 128, 1)
pol.init(2); cri.init(3)
traj = ra.Trajectory(eng, 512, 32, 5)
ra.rollout(env, pol, traj); ra.gae(traj, cri, 0.99, 0.95)
st = ra.trpo_update(pol, traj)
cs = ra.critic_update(cri, ra.Adam(cri), traj, 3)
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("amdhip64", "librccl", "hsa-runtime"))})
print(order, "OK trpo", st.status, "critic", cs.loss_last)
print("BOUND rccl=%s hip=%s" % ra.comm_library_paths())
print("\n".join(maps))
