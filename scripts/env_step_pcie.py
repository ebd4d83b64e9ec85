"""rl_env_step with HOST buffers (actions in, reward / flag / observations / interrupt successors out over PCIe) next to
the resident step (rl_env_step_resident): the PCIe-inclusive rate of the standalone env boundary, for DESIGN.md §9."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import relearn_amd as ra
eng = ra.Engine(0)
out = []
for n in (65536, 1 << 20):
    env = ra.CartPoleEnv(eng, n)
    a = np.random.default_rng(0).integers(0, 2, size=n).astype(np.uint8)
    for _ in range(3): env.step(a)
    reps = 20
    eng.sync(); t0 = time.perf_counter()
    for _ in range(reps): env.step(a)
    eng.sync(); host_ms = (time.perf_counter() - t0) / reps * 1e3
    env.upload_actions(a)
    for _ in range(3): env.step_resident()
    eng.sync(); eng.timer_begin()
    for _ in range(reps): env.step_resident()
    res_ms = eng.timer_end() / reps
    out.append({"lanes": n, "host_buffers_ms": host_ms, "host_buffers_steps_per_s": n / host_ms * 1e3,
                "resident_ms": res_ms, "resident_steps_per_s": n / res_ms * 1e3,
                "bytes_over_pcie_per_step": 1 + 4 + 1 + 2 * 5 * 4})
    env.close()
print(json.dumps(out))
