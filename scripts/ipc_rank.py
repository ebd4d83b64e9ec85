"""One rank of a multi-PROCESS job over the peer-mailbox collective (rl_comm_init_ipc), all ranks on device 0 — the
functional rehearsal a one-GPU box allows: IPC handles of one device map into other processes exactly like peer windows
do.  Handles are exchanged through files in `dir`.   usage: ipc_rank.py <rank> <world> <dir> [n_total] [T] [desert]
`desert`: the last rank leaves right after the communicator exists; the others must see RL_ERR_COMM within the wait
bound (RELEARN_IPC_TIMEOUT_MS), with their parameters and optimiser state untouched, and fail fast afterwards."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import relearn_amd as ra  # noqa: E402

rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n_total = int(sys.argv[4]) if len(sys.argv) > 4 else 512
T = int(sys.argv[5]) if len(sys.argv) > 5 else 48
desert = len(sys.argv) > 6 and sys.argv[6] in ("desert", "desert_trpo")
desert_trpo = desert and sys.argv[6] == "desert_trpo"


def wait_for(paths, what, limit=120.0):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > limit:
            sys.exit("rank %d: timed out waiting for %s" % (rank, what))
        time.sleep(0.02)


eng = ra.Engine(0)
if world > 1:
    h = eng.comm_ipc_handle(world)
    tmp = os.path.join(d, "h%d.tmp" % rank)
    open(tmp, "wb").write(h)
    os.rename(tmp, os.path.join(d, "h%d.bin" % rank))
    files = [os.path.join(d, "h%d.bin" % r) for r in range(world)]
    wait_for(files, "the peers' mailbox handles")
    eng.comm_init_ipc(rank, world, [open(f, "rb").read() for f in files])
    if desert:
        if rank == world - 1:
            # stay mapped (nobody writes into freed memory) but never take part in a collective
            open(os.path.join(d, "deserted"), "w").close()
            wait_for([os.path.join(d, "seen%d" % r) for r in range(world - 1)], "the others to notice")
            print("rank %d of %d deserted" % (rank, world))
            sys.exit(0)
        wait_for([os.path.join(d, "deserted")], "the deserter")
        cri = ra.Mlp(eng, 5, 128, 1)
        cri.init(3)
        opt = ra.Adam(cri)
        env = ra.CartPoleEnv(eng, 64, max_steps=30, lane_offset=rank * 64)
        pol = ra.Mlp(eng, 5, 128, 2)
        pol.init(2)
        traj = ra.Trajectory(eng, 64, 16, 5)
        ra.rollout(env, pol, traj)
        ra.gae(traj, cri, 0.99, 0.95)
        before, before_pol = cri.get_params(), pol.get_params()
        t0 = time.time()
        try:
            if desert_trpo:  # stand-alone all-reduce launches: gradient, Fisher-vector products, line search
                ra.trpo_update(pol, traj)
            else:            # the exchange inside the reduce + Adam launch
                ra.critic_update(cri, opt, traj, 3)
            eng.sync()
            sys.exit("rank %d: the update with a missing peer did not fail" % rank)
        except ra.RelearnError as err:
            assert err.code == ra.ERR_COMM, err
        first = time.time() - t0
        assert np.array_equal(cri.get_params(), before), "a failed exchange must not step the parameters"
        assert np.array_equal(pol.get_params(), before_pol), "a failed exchange must not move the policy"
        t0 = time.time()
        try:
            eng.comm_selftest()
            sys.exit("rank %d: a collective after the failure did not fail" % rank)
        except ra.RelearnError as err:
            assert err.code == ra.ERR_COMM, err
        again = time.time() - t0
        print("rank %d of %d saw the timeout after %.2f s, then failed fast in %.3f s" % (rank, world, first, again))
        open(os.path.join(d, "seen%d" % rank), "w").close()
        sys.exit(0)
    eng.comm_selftest()
n = n_total // world
env = ra.CartPoleEnv(eng, n, max_steps=30, lane_offset=rank * n, seed_env=0, seed_actor=1)
pol, cri = ra.Mlp(eng, 5, 128, 2), ra.Mlp(eng, 5, 128, 1)
pol.init(2)
cri.init(3)
opt = ra.Adam(cri)
traj = ra.Trajectory(eng, n, T, 5)
out = {"policy_init": pol.get_params(), "critic_init": cri.get_params()}
eng.profile_enable(True)
for period in range(2):
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    if period == 0:
        out["action"] = traj.read(ra.TRAJ_ACTION)
        out["adv"] = traj.read(ra.TRAJ_ADVANTAGES)
        # (what the oracle-based bars of tests/test_gpu_multirank.py need: the samples and the critic's targets)
        out["obs"], out["flag"], out["reward"] = traj.read(ra.TRAJ_OBS), traj.read(ra.TRAJ_FLAG), traj.read(ra.TRAJ_REWARD)
        out["rtg"] = traj.read(ra.TRAJ_RETURNS)
        g, loss, ent = ra.policy_gradient(pol, traj)
        v = np.random.default_rng(7).standard_normal(pol.P).astype(np.float32)
        gc, lc = ra.critic_gradient(cri, traj)
        out.update(probe_g=g, probe_loss=np.float64(loss), probe_ent=np.float64(ent),
                   probe_hv=ra.policy_fvp(pol, traj, v, 0.0), probe_gc=gc, probe_lc=np.float64(lc))
    if period == 0:
        st = ra.trpo_update(pol, traj)
        cs, losses = ra.critic_update(cri, opt, traj, 6, want_losses=True)
    else:  # the two chains side by side, each on its own half of the mailboxes
        ccfg = ra.values_opt_config_default()
        ccfg.opt_steps_per_update = 6
        st, cs, losses = ra.actor_critic_update(pol, cri, opt, traj, None, ccfg, want_losses=True)
    out["policy%d" % period] = pol.get_params()
    out["critic%d" % period] = cri.get_params()
    out["losses%d" % period] = losses
    out["trpo%d" % period] = np.array([st.loss_initial, st.entropy, st.step_size, st.cg_iterations, st.status,
                                       st.loss_final, st.constraint_val_final, st.step_scale, st.num_backtracks])
eng.sync()
out["allreduce_launches"] = np.array([eng.profile_read()["allreduce"][1]])
np.savez(os.path.join(d, "out%d_of_%d.npz" % (rank, world)), **out)
if world > 1:
    # nobody unmaps a mailbox a peer may still be writing to
    open(os.path.join(d, "done%d" % rank), "w").close()
    wait_for([os.path.join(d, "done%d" % r) for r in range(world)], "the peers to finish")
    eng.comm_destroy()
print("rank %d of %d ok" % (rank, world))
