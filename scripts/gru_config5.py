"""BASELINE.json configs[4]: partially observed Chain (LatentStepLimit 100) with the GRU policy, 16,384 lanes, T = 100,
one MI355X: rollout, GAE with a recurrent critic, PPO (10 steps) and critic (80 steps) updates through time.
Prints one JSON line: device times (HIP events) per phase and per kernel class, and MFMA rates."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import relearn_amd as ra  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
critic_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 80
periods = int(sys.argv[4]) if len(sys.argv) > 4 else 2

eng = ra.Engine(0)
env = ra.ChainEnv(eng, N, max_steps=100)
pol, cri = ra.GruMlp(eng, 5, 2), ra.GruMlp(eng, 5, 1)
pol.init(1)
cri.init(2)
popt, copt = ra.Adam(pol), ra.Adam(cri)
traj = ra.Trajectory(eng, N, T, 5)
ppo = ra.ppo_config_default()


def period():
    t = {}
    eng.timer_begin(); ra.rollout(env, pol, traj); t["rollout_ms"] = eng.timer_end()
    eng.timer_begin(); ra.gae(traj, cri, 0.95, 0.95); t["gae_ms"] = eng.timer_end()
    eng.timer_begin(); ps = ra.ppo_update(pol, popt, traj, ppo); t["ppo_ms"] = eng.timer_end()
    eng.timer_begin(); cs = ra.critic_update(cri, copt, traj, critic_steps); t["critic_ms"] = eng.timer_end()
    return t, ps, cs


period()  # warm-up (allocates the activation records)
eng.profile_enable(True)
eng.profile_read(reset=True)
acc = {}
for _ in range(periods):
    t, ps, cs = period()
    for k, v in t.items():
        acc[k] = acc.get(k, 0.0) + v / periods
prof = eng.profile_read(reset=True)
steps = N * T
fwd_flop = 2.0 * (128 * 384 + 128 * 128) * steps            # one cell + MLP layer per sample-step (MFMA part)
bwd_flop = fwd_flop                                         # W_hh^T and W1^T products
wg_flop = 2.0 * (384 * 128 + 128 * 128) * steps             # dW_hh and dW1 GEMMs
total_ms = sum(acc.values())
out = {"lanes": N, "horizon": T, "env_steps_per_period": steps, **acc, "period_ms": total_ms,
       "env_steps_per_s": steps / total_ms * 1e3,
       "rollout_tflops": fwd_flop / acc["rollout_ms"] / 1e9,
       "policy_loss": [ps.loss_first, ps.loss_last], "critic_loss": [cs.loss_first, cs.loss_last],
       "kernel_ms_per_period": {k: v[0] / periods for k, v in prof.items() if v[1]},
       "kernel_launches_per_period": {k: v[1] / periods for k, v in prof.items() if v[1]},
       "flop_per_pass": {"forward": fwd_flop, "bptt": bwd_flop, "wgrad": wg_flop}}
# per gradient evaluation (90 per period: 10 PPO + 80 critic): the three passes' device time by kernel class, their
# f32-equivalent rate against the f32 matrix peak (157.3 TFLOP/s) and the rate the bf16 pipe actually executes
# (every f32 product = 6 bf16 products of pieces: PIECE_PAIRS in kernels_seq_train.hip) against its dense peak (2,500 TFLOP/s)
k = out["kernel_ms_per_period"]
n_grad = 10 + critic_steps
dw1_flop = 2.0 * 128 * 128 * steps  # the head's weight gradient is accumulated by the head's backward kernel
passes = {"forward": ("policy_fused", fwd_flop), "backward": ("backward", bwd_flop + dw1_flop),
          "weight_gradients": ("critic_fused", wg_flop - dw1_flop)}
out["gradient_passes"] = {}
for name, (cls, flop) in passes.items():
    if cls in k:
        ms = k[cls] / n_grad
        out["gradient_passes"][name] = {
            "ms": ms, "f32_equivalent_tflops": flop / ms / 1e9, "frac_of_f32_mfma_peak": flop / ms / 1e9 / 157.3,
            "executed_bf16_tflops": 6 * flop / ms / 1e9, "frac_of_bf16_mfma_peak": 6 * flop / ms / 1e9 / 2500.0}
out["ms_per_gradient"] = sum(v["ms"] for v in out["gradient_passes"].values()) + (k.get("reduce", 0) + k.get("small", 0)) / n_grad
# ---- roofline of a gradient evaluation, as bench.py reports it for config 4.  The five kernels of an
# evaluation are HBM-bound on the activation record they hand each other: 24 plane transfers of N x T x 128 x 4 B
#  against an algorithmic input of the observations, actions, advantages and flags of the batch.
import roofline_util as ru  # noqa: E402
plane = 4.0 * 128 * steps
accounted = 24.0 * plane
alg_bytes = (5 * 4 + 1 + 4 + 1) * steps
kernels, src = ru.pmc_kernels("r06_pmc_gru_config5_summary.json")
traffic, found = ru.traffic_of(kernels if src["applies"] else {}, {
    "k_gru_recur_fwd": 1.0, "k_seq_head_forward<1>": 80.0 / n_grad, "k_seq_head_forward<2>": 10.0 / n_grad,
    "k_gru_head_backward<1>": 80.0 / n_grad, "k_gru_head_backward<2>": 10.0 / n_grad, "k_gru_recur_bwd": 1.0,
    "k_gru_wgrad_bf16": 1.0})
ms = out["ms_per_gradient"]
moved = traffic if traffic else accounted
out["roofline"] = {
    "kernel": "one gradient evaluation through time (k_gru_recur_fwd, k_seq_head_forward, k_gru_head_backward, "
              "k_gru_recur_bwd, k_gru_wgrad_bf16)",
    "bound": "hbm", "achieved": moved / (ms * 1e-3) / 1e9, "peak": ru.HBM_PEAK_GBS, "unit": "GB/s",
    "frac": moved / (ms * 1e-3) / 1e9 / ru.HBM_PEAK_GBS, "traffic": traffic,
    "bytes_used_for_achieved": "counters" if traffic else "plane accounting (24 planes)",
    "accounted_bytes": accounted, "algorithmic_bytes": alg_bytes, "moved_over_algorithmic": moved / alg_bytes,
    "ms_per_gradient": ms, "gradients_per_period": n_grad, "source": src, "kernels_in_traffic": found,
    "note": "achieved = bytes the evaluation MOVES / its time (the bench contract's HBM form); the algorithmic input of the "
            "batch is ~400x smaller: the traffic is the activation record of backpropagation through time (DESIGN 13a)"}
print(json.dumps(out))
