"""world_size-2 rehearsal of the multi-GPU algorithm on CPU (gloo): lanes sharded by global lane id, every
reduced vector all-reduced (sum) and every rank applying the same deterministic CG / line-search bookkeeping.
Uses the oracle's per-sample kernels in place of the HIP ones; what is tested is the sharding rule and the
algebra of the exchange (which vectors are summed, how they are scaled), not the kernels.  CPU only."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = O.lib()
    H, n_total, T = 16, 64, 24
    ps, cs = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)
    pp, cp = O.mlp_init(ps, 2), O.mlp_init(cs, 3)
    n_local = n_total // world
    sim = O.LaneSim(n_local, max_steps=15, lane_offset=rank * n_local)
    traj = sim.rollout(ps, pp, T, threads=1)
    _, adv, _ = O.lanes_gae(cs, cp, traj, np.float32(0.99), np.float32(0.95))
    x, a = O.flat_samples(traj)
    adv = np.ascontiguousarray(adv.reshape(-1))
    B_local, B_total = len(a), len(a) * world
    P = len(pp)

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    w = np.float32(B_local) / np.float32(B_total)  # local means -> contributions to the global mean

    def grad():
        g = np.zeros(P, np.float32)
        loss = C.c_float()
        L.oracle_policy_grad_f32(ps, O.f32p(pp), O.f32p(x), O.i64p(a), O.f32p(adv), B_local, O.f32p(g),
                                 C.byref(loss))
        return allreduce(g * w), float(allreduce(np.array([loss.value * w]))[0])

    def fvp(v, reg):
        out = np.zeros(P, np.float32)
        L.oracle_policy_fvp_f32(ps, O.f32p(pp), O.f32p(x), B_local, O.f32p(v), 0.0, O.f32p(out))
        return allreduce(out * w) + np.float32(reg) * v

    def loss_kl(p_new):
        lo, kl = C.c_float(), C.c_float()
        L.oracle_policy_loss_kl_f32(ps, O.f32p(p_new), O.f32p(pp), O.f32p(x), O.i64p(a), O.f32p(adv), B_local,
                                    C.byref(lo), C.byref(kl))
        r = allreduce(np.array([lo.value * w, kl.value * w]))
        return float(r[0]), float(r[1])

    # the engine's TRPO step with all-reduces where rl_trpo_update places them
    g, loss0 = grad()
    xk, r, p = np.zeros(P, np.float32), g.copy(), g.copy()
    rr = np.float32(np.dot(r.astype(np.float64), r))
    for _ in range(10):
        z = fvp(p, 1e-5)
        alpha = rr / np.float32(np.dot(p.astype(np.float64), z))
        xk = xk + alpha * p
        r = r - alpha * z
        new_rr = np.float32(np.dot(r.astype(np.float64), r))
        if new_rr < 1e-10:
            break
        p = p * (new_rr / rr) + r
        rr = new_rr
    xhx = float(np.float32(np.dot(xk.astype(np.float64), fvp(xk, 1e-5))))
    step = np.float32(np.sqrt(1.0 / (xhx + 1e-8) * 0.01 * 2.0))
    new_params, accepted = pp, -1
    for i in range(15):
        cand = (pp - np.float32(0.8 ** i) * (step * xk)).astype(np.float32)
        lo, kl = loss_kl(cand)
        if lo < loss0 and kl <= 0.01:
            new_params, accepted = cand, i
            break
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(traj=traj, params=new_params, g=g))
    if rank == 0:
        q.put(dict(gathered=gathered, accepted=accepted, loss0=loss0, step=float(step)))
    dist.destroy_process_group()


def test_two_rank_sharded_trpo_step_equals_single_process():
    import torch.multiprocessing as mp

    import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g0, g1 = res["gathered"]
    # every rank ends with identical parameters (redundant deterministic updates, no broadcast needed)
    assert np.array_equal(g0["params"], g1["params"])
    assert np.array_equal(g0["g"], g1["g"])
    # sharded lanes == the corresponding lanes of one big simulation
    H, n_total, T = 16, 64, 24
    ps, cs = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)
    pp, cp = O.mlp_init(ps, 2), O.mlp_init(cs, 3)
    sim = O.LaneSim(n_total, max_steps=15)
    full = sim.rollout(ps, pp, T, threads=2)
    for k in ("obs", "action", "flag"):
        assert np.array_equal(full[k][..., :32], g0["traj"][k])
        assert np.array_equal(full[k][..., 32:], g1["traj"][k])
    # and the all-reduced gradient / accepted step match the single-process oracle on the full batch
    _, adv, _ = O.lanes_gae(cs, cp, full, np.float32(0.99), np.float32(0.95))
    x, a = O.flat_samples(full)
    adv = np.ascontiguousarray(adv.reshape(-1))
    g = np.zeros(len(pp), np.float32)
    loss = C.c_float()
    O.lib().oracle_policy_grad_f32(ps, O.f32p(pp), O.f32p(x), O.i64p(a), O.f32p(adv), len(a), O.f32p(g),
                                   C.byref(loss))
    assert np.abs(g - g0["g"]).max() <= 2e-6 * np.abs(g).max()
    assert abs(loss.value - res["loss0"]) < 1e-5
    p_new, st, _ = O.trpo_update(ps, pp, x, a, adv)
    assert st.status == O.OPT_OK
    assert abs(st.num_backtracks - res["accepted"]) <= 1
    assert abs(st.step_size - res["step"]) < 0.05 * st.step_size
