import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle as O
import relearn_amd as ra
from test_gpu_gru import narrow_modules, synthetic_history
eng = ra.Engine(0)
for cell, D, H, H2, A in [("lstm", 5, 128, 128, 2), ("lstm", 5, 128, 128, 1), ("gru", 5, 128, 128, 2), ("lstm", 5, 20, 16, 2), ("lstm", 4, 20, 16, 2), ("lstm", 4, 128, 128, 2), ("lstm", 5, 128, 16, 2), ("lstm", 5, 20, 128, 2), ("gru", 4, 20, 16, 2)]:
    m, shape = narrow_modules(eng, cell, D, H, H2, A, 31)
    traj, want = synthetic_history(eng, 64, 12, D, 5)
    out_d, succ_d = m.seq_forward(traj)
    out_o, succ_o = O.gru_seq_forward(shape, m.get_params(), want)
    bad = np.argwhere(out_d != out_o)
    print(cell, D, H, H2, A, "mismatches", len(bad), "of", out_d.size, "succ", int((succ_d != succ_o).sum()), bad[:6].tolist(),
          "max abs", np.abs(out_d - out_o).max())
    if len(bad):
        b = bad[0]; print("   first", out_d[tuple(b)], out_o[tuple(b)], "obs", want["obs"][:, b[1], b[2]])
        ts = sorted(set(bad[:, 1].tolist())); print("  steps with mismatches", ts[:12], "flags before first", want["flag"][:ts[0] + 1, bad[0][2]].tolist())
