"""k_env_step throughput vs lane count (100 B per env-step): at 65,536 lanes the launch is
latency-bound (one wave per SIMD); the sweep shows what the kernel sustains when the chip is filled."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import relearn_amd as ra
eng = ra.Engine(0)
out = []
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
    env = ra.CartPoleEnv(eng, n)
    env.upload_actions(np.random.default_rng(0).integers(0, 2, size=n).astype(np.uint8))
    for _ in range(5): env.step_resident()
    reps = 50
    eng.sync(); eng.timer_begin()
    for _ in range(reps): env.step_resident()
    ms = eng.timer_end() / reps
    out.append({"lanes": n, "us_per_launch": 1e3 * ms, "env_steps_per_s": n / ms * 1e3, "GBps_at_100B": 100.0 * n / ms / 1e6,
                "hbm_frac": 100.0 * n / ms / 1e6 / 8000.0})
    env.close()
print(json.dumps(out))
