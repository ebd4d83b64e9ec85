"""BASELINE.json configs[2]: CartPole DQN with the replay buffer in HBM, 4,096 lanes on one MI355X.

examples/cartpole-dqn.rs shape: buffer 50 M steps in total (12,207 per lane), first collection 5 M steps, then
100 k per update; 50 optimisation steps on minibatches of >= 100 k steps of whole episodes; Adam 1e-3.
Prints one JSON line with collection and update rates (device time, HIP events) and per-class kernel times.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relearn_amd as ra  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
first_T = int(sys.argv[2]) if len(sys.argv) > 2 else (5_000_000 + N - 1) // N
rest_T = (100_000 + N - 1) // N
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 5

eng = ra.Engine(0)
env = ra.CartPoleEnv(eng, N, max_steps=500, seed_env=0, seed_actor=1)
q = ra.Mlp(eng, 5, 128, 2)
q.init(2)
opt = ra.Adam(q)
cfg = ra.dqn_config_default()
cfg.buffer_capacity = 50_000_000 // N
cfg.update_first, cfg.update_rest = 5_000_000, 100_000
if os.environ.get("DQN_TD") == "1":  # one-step TD targets instead of the default reward-to-go
    cfg.target = ra.DQN_TARGET_ONE_STEP_TD
for i in range(8):
    cfg.agent_key[i] = 1000 + i
dqn = ra.Dqn(env, q, opt, cfg)

eng.sync()
eng.timer_begin()
dqn.collect(first_T, want_stats=False)
ms_first = eng.timer_end()
out = {"target": "one-step-td" if cfg.target == ra.DQN_TARGET_ONE_STEP_TD else "reward-to-go", "lanes": N, "buffer_capacity_per_lane": int(cfg.buffer_capacity), "first_collect_steps": first_T * N,
       "first_collect_ms": ms_first, "first_collect_steps_per_s": first_T * N / ms_first * 1e3}
st = dqn.update()  # warm-up
# device time of an update without the per-kernel profiling events
t_plain = 0.0
for it in range(updates):
    dqn.collect(rest_T, want_stats=False)
    eng.timer_begin()
    st = dqn.update()
    t_plain += eng.timer_end()
out["update_ms"] = t_plain / updates
out["trained_samples_per_s"] = int(st.last_minibatch_steps) * int(cfg.opt_steps_per_update) / (t_plain / updates) * 1e3
eng.profile_enable(True)
eng.profile_read(reset=True)
t_collect = t_update = 0.0
wall0 = time.time()
for it in range(updates):
    eng.timer_begin()
    dqn.collect(rest_T, want_stats=False)
    t_collect += eng.timer_end()
    eng.timer_begin()
    st = dqn.update()
    t_update += eng.timer_end()
wall = time.time() - wall0
prof = eng.profile_read(reset=True)
out.update({
    "updates": updates, "collect_steps_per_update": rest_T * N, "collect_ms": t_collect / updates,
    "update_ms_with_profiling": t_update / updates, "opt_steps_per_update": int(cfg.opt_steps_per_update),
    "minibatch_steps": int(st.last_minibatch_steps), "minibatch_episodes": int(st.last_minibatch_episodes),
    "wall_s_per_update_with_profiling": wall / updates,
    "loss_first": st.loss_first, "loss_last": st.loss_last, "exploration_rate": dqn.exploration_rate(),
    "kernel_ms_per_update": {k: v[0] / updates for k, v in prof.items() if v[1]},
    "kernel_launches_per_update": {k: v[1] / updates for k, v in prof.items() if v[1]},
})
print(json.dumps(out))
