"""Several ranks of a peer-mailbox job in ONE process (one engine and one host thread per rank), several such processes
per job: rank = proc * ranks_per_proc + i, all on device 0.  A GPU box of this pool admits at most six processes on its
card, so the 8-rank split of config 4 (8 x 8,192 lanes) is rehearsed as 4 processes x 2 ranks: ranks of one process find
each other's mailboxes by address, ranks of other processes through hipIpcOpenMemHandle (relearn_amd/csrc/comm_ipc.hip).
usage: ipc_multi.py <proc> <n_procs> <ranks_per_proc> <dir> [n_total] [T] [critic_steps] [periods]
Every rank runs `periods` periods of the bench's hot path (rollout, rl_gae, rl_actor_critic_update) on its lane slice
and saves what the test compares (tests/test_gpu_multirank.py)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import relearn_amd as ra  # noqa: E402


def wait_for(paths, what, limit=180.0):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > limit:
            raise SystemExit("timed out waiting for %s" % what)
        time.sleep(0.02)


def config4_rank(eng, rank, world, n_total, T, critic_steps, periods, save_obs=False):
    """the bench's period on this rank's lanes; returns what the 1-rank / N-rank comparison needs (`save_obs`: also the
    observation planes of the first period — 169 MB at the full size, asked of the one-rank run only)"""
    n = n_total // world
    env = ra.CartPoleEnv(eng, n, max_steps=500, lane_offset=rank * n, seed_env=0, seed_actor=1)
    pol, cri = ra.Mlp(eng, 5, 128, 2), ra.Mlp(eng, 5, 128, 1)
    pol.init(2)
    cri.init(3)
    opt = ra.Adam(cri)
    traj = ra.Trajectory(eng, n, T, 5)
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update = critic_steps
    out = {"policy_init": pol.get_params()}
    eng.profile_enable(True)
    for period in range(periods):
        ra.rollout(env, pol, traj)
        ra.gae(traj, cri, 0.99, 0.95)
        say(rank, "period %d: rollout and advantages enqueued" % period)
        if period == 0:
            out["action"], out["flag"] = traj.read(ra.TRAJ_ACTION), traj.read(ra.TRAJ_FLAG)
            out["adv"] = traj.read(ra.TRAJ_ADVANTAGES)
            if save_obs:
                out["obs"] = traj.read(ra.TRAJ_OBS)
            # every kind of all-reduced vector on its own, before anything is updated (collective calls; what
            # tests/test_gpu_multirank.py::check_probe_vectors compares)
            g, loss, ent = ra.policy_gradient(pol, traj)
            v = np.random.default_rng(7).standard_normal(pol.P).astype(np.float32)
            gc, lc = ra.critic_gradient(cri, traj)
            out.update(probe_g=g, probe_loss=np.float64(loss), probe_ent=np.float64(ent),
                       probe_hv=ra.policy_fvp(pol, traj, v, 0.0), probe_gc=gc, probe_lc=np.float64(lc))
            say(rank, "probe vectors done")
        st, cs, losses = ra.actor_critic_update(pol, cri, opt, traj, None, ccfg, want_losses=True)
        say(rank, "period %d: update done" % period)
        out["policy%d" % period], out["critic%d" % period] = pol.get_params(), cri.get_params()
        out["losses%d" % period] = losses
        out["trpo%d" % period] = np.array([st.loss_initial, st.entropy, st.step_size, st.cg_iterations, st.status,
                                           st.num_backtracks, st.step_scale, st.loss_final, st.constraint_val_final])
    eng.sync()
    out["allreduce_launches"] = np.array([eng.profile_read()["allreduce"][1]])
    return out


T0 = time.time()


def say(rank, what):
    """a progress line per phase (the tests keep the output: where a rank was when a peer gave up on it)"""
    print("[%7.2f s] rank %d: %s" % (time.time() - T0, rank, what), flush=True)


def main():
    proc, n_procs, rpp, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    n_total = int(sys.argv[5]) if len(sys.argv) > 5 else 65536
    T = int(sys.argv[6]) if len(sys.argv) > 6 else 128
    critic_steps = int(sys.argv[7]) if len(sys.argv) > 7 else 80
    periods = int(sys.argv[8]) if len(sys.argv) > 8 else 2
    world = n_procs * rpp
    engines = [ra.Engine(0) for _ in range(rpp)]
    ranks = [proc * rpp + i for i in range(rpp)]
    if world > 1:
        for eng, rank in zip(engines, ranks):
            h = eng.comm_ipc_handle(world)
            tmp = os.path.join(d, "h%d.tmp" % rank)
            open(tmp, "wb").write(h)
            os.rename(tmp, os.path.join(d, "h%d.bin" % rank))
        files = [os.path.join(d, "h%d.bin" % r) for r in range(world)]
        wait_for(files, "the peers' mailbox handles")
        handles = [open(f, "rb").read() for f in files]
    errors = []

    def run(eng, rank):
        try:
            if world > 1:
                say(rank, "handles of all %d ranks read" % world)
                eng.comm_init_ipc(rank, world, handles)  # (ends with the job's first all-reduce: every rank must be here)
                say(rank, "mailboxes mapped, first all-reduce done")
                eng.comm_selftest()
                say(rank, "self-test done")
            out = config4_rank(eng, rank, world, n_total, T, critic_steps, periods)
            np.savez(os.path.join(d, "out%d_of_%d.npz" % (rank, world)), **out)
        except BaseException as exc:
            errors.append((rank, exc))
            raise
        finally:
            open(os.path.join(d, ("done%d" if not errors else "failed%d") % rank), "w").close()

    threads = [threading.Thread(target=run, args=(e, r)) for e, r in zip(engines, ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit("proc %d: %r" % (proc, errors))
    if world > 1:
        # nobody unmaps a mailbox a peer may still be writing to
        wait_for([os.path.join(d, "done%d" % r) for r in range(world)], "the peers to finish")
        for eng in engines:
            eng.comm_destroy()
    print("proc %d of %d (%d ranks each) ok" % (proc, n_procs, rpp))


if __name__ == "__main__":
    main()
