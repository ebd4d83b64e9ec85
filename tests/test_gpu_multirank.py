"""Multi-rank arithmetic on ONE GPU: two engines in one process, each driven by its own host thread, joined by the
library's in-process loopback collective (RELEARN_LOOPBACK_COMM=1, relearn_amd/csrc/abi.hip).  Everything except
the RCCL call itself (exercised separately with a one-rank communicator) is the code the 2/4/8-GPU runs execute:
lane sharding by global lane id, all-reduced gradients / Fisher-vector products / loss sums, sample-weighted means
over all ranks, identical redundant updates on every rank.  The sharded result must equal the single-engine result
up to the order of the f32 partial sums."""
import os
import threading
import time

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

H = 128


PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)


def probe_vectors(pol, cri, traj):
    """the three kinds of vector a sharded update all-reduces, each once: the policy gradient (with its loss and entropy
    sums), a Fisher-vector product with a fixed tangent, the critic gradient (with its loss)"""
    g, loss, ent = ra.policy_gradient(pol, traj)
    v = np.random.default_rng(7).standard_normal(pol.P).astype(np.float32)
    hv = ra.policy_fvp(pol, traj, v, 0.0)
    gc, lc = ra.critic_gradient(cri, traj)
    return dict(g=g, loss=loss, ent=ent, hv=hv, gc=gc, lc=lc)


def check_probe_vectors(sharded, single, tol=2e-6):
    """same samples, other order of the f32 partial sums and one rounding more per all-reduce: the vectors agree to
    rounding (2e-6 of their largest entry; a shard weighted 1.5 x instead of 1 moves them by per cent)"""
    for k in ("g", "hv", "gc"):
        a, b = np.asarray(sharded[k], dtype=np.float64), np.asarray(single[k], dtype=np.float64)
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), k
    for k in ("loss", "ent", "lc"):
        assert abs(sharded[k] - single[k]) <= 1e-6 * max(1.0, abs(single[k])), k


def run_rank(rank, world, uid, lanes, T, out, barrier, critic_steps=10, with_dqn=True):
    """`lanes`: the lane count of every rank (the library's contract: equal shards — the negative test breaks it)"""
    try:
        eng = ra.Engine(0)
        eng.profile_enable(True)
        if world > 1:
            eng.comm_init(rank, world, uid)
        n, first = lanes[rank], sum(lanes[:rank])
        env = ra.CartPoleEnv(eng, n, max_steps=40, lane_offset=first, seed_env=5, seed_actor=6)
        pol, cri = ra.Mlp(eng, 5, H, 2), ra.Mlp(eng, 5, H, 1)
        pol.init(2)
        cri.init(3)
        opt = ra.Adam(cri)
        traj = ra.Trajectory(eng, n, T, 5)
        res = {"policy_init": pol.get_params(), "critic_init": cri.get_params()}
        for period in range(2):
            ra.rollout(env, pol, traj)
            ra.gae(traj, cri, 0.99, 0.95)
            before = traj.read_all() if period == 0 else None
            rtg = traj.read(ra.TRAJ_RETURNS) if period == 0 else None
            if period == 0:  # every kind of all-reduced vector on its own, before anything is updated (collective calls)
                res["probe"] = probe_vectors(pol, cri, traj)
            # period 0: the two updates in turn; period 1: side by side on two streams, each chain with its own
            # collective channel (rl_actor_critic_update) — identical replicas either way
            if period == 0:
                st = ra.trpo_update(pol, traj)
                cs, losses = ra.critic_update(cri, opt, traj, critic_steps, want_losses=True)
            else:
                ccfg = ra.values_opt_config_default()
                ccfg.opt_steps_per_update = critic_steps
                st, cs, losses = ra.actor_critic_update(pol, cri, opt, traj, None, ccfg, want_losses=True)
            res[period] = dict(action=traj.read(ra.TRAJ_ACTION), adv=traj.read(ra.TRAJ_ADVANTAGES), traj=before, rtg=rtg,
                               policy=pol.get_params(), critic=cri.get_params(), trpo=st.as_dict(), losses=losses)
        if with_dqn:
            # DQN on the same lanes: the minibatch size is summed over ranks, gradients all-reduced
            q = ra.Mlp(eng, 5, H, 2)
            q.init(7)
            cfg = ra.dqn_config_default()
            cfg.exploration_kind, cfg.exploration_start = ra.SCHEDULE_CONSTANT, 0.3
            cfg.minibatch_steps, cfg.opt_steps_per_update, cfg.buffer_capacity = 2000 // world, 3, 128
            for i in range(8):
                cfg.agent_key[i] = 100 + i + rank  # every rank samples its own lanes with its own stream
            dqn = ra.Dqn(env, q, ra.Adam(q), cfg)
            dqn.collect(60)
            dst, dl = dqn.update(want_losses=True)
            res["dqn"] = dict(q=q.get_params(), losses=dl, global_steps=dst.global_steps)
        res["allreduce_launches"] = eng.profile_read()["allreduce"][1]
        out[rank] = res
    except BaseException as exc:  # surface the failure in the main thread
        out[rank] = exc
        raise
    finally:
        try:
            barrier.abort() if isinstance(out.get(rank), BaseException) else None
        except Exception:
            pass


def launch(world, n_total, T, lanes=None, **kw):
    os.environ["RELEARN_LOOPBACK_COMM"] = "1"
    try:
        uid = ra.comm_unique_id()
        out = {}
        barrier = threading.Barrier(world)
        lanes = lanes if lanes is not None else [n_total // world] * world
        threads = [threading.Thread(target=run_rank, args=(r, world, uid, lanes, T, out, barrier), kwargs=kw)
                   for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        for r in range(world):
            assert r in out and not isinstance(out[r], BaseException), out.get(r)
        return out
    finally:
        os.environ.pop("RELEARN_LOOPBACK_COMM", None)


def check_sharded_update_against_the_oracles(ranks, single, critic_steps=10):
    """The bar of the single-rank tests (tests/test_gpu_parity.py::test_trpo_update_default_config_vs_f64_truth,
    test_critic_gradient_and_update_match_oracle) applied to a sharded update: same samples, the f32 partial sums in
    another order and one rounding more per all-reduced vector.
      * the ranks' trajectories, laid side by side, ARE the single rank's (lanes are global ids), bit for bit;
      * the sample-weighted scalars (initial loss, entropy) agree to 1e-6, the CG iteration count exactly;
      * step size: no farther from the f64 oracle than 2 x the f32 oracle is;
      * step direction: backward error |(F + reg I) x - g| / |g| under the f64 operator <= 3 x the worse oracle's;
      * the accepted step obeys the trust region and improves the surrogate;
      * critic: the loss before every one of the Adam steps within 1e-5 of the single rank's and of the f32 oracle's, the
        parameters after them within the single-rank test's bound of the f32 oracle's.
    A shard weighted by the wrong B_local / B_total moves the scalars and the per-step losses by per cent (the negative
    test below): none of these bars lets that through."""
    s = single[0]
    if "probe" in single and "probe" in ranks[0]:
        check_probe_vectors(ranks[0]["probe"], single["probe"])
    tr = s["traj"]
    for f in ("action", "flag", "reward", "obs"):
        axis = 2 if f == "obs" else 1
        assert np.array_equal(np.concatenate([r[0]["traj"][f] for r in ranks], axis=axis), tr[f]), f
    assert np.array_equal(np.concatenate([r[0]["adv"] for r in ranks], axis=1), s["adv"])
    x, a = O.flat_samples(tr)
    adv, rtg = s["adv"].reshape(-1), s["rtg"].reshape(-1)
    p0, c0 = single["policy_init"], single["critic_init"]
    st = ranks[0][0]["trpo"]
    p_d = ranks[0][0]["policy"]
    assert abs(st["loss_initial"] - s["trpo"]["loss_initial"]) < 1e-6 and abs(st["entropy"] - s["trpo"]["entropy"]) < 1e-6
    p32, st32, sd32 = O.trpo_update(PS, p0, x, a, adv)
    p64, st64, sd64 = O.trpo_update(PS, p0, x, a, adv, f64=True)
    assert st["status"] == st32.status == st64.status == ra.OPT_OK
    assert st["cg_iterations"] == s["trpo"]["cg_iterations"] == st64.cg_iterations
    assert abs(st["entropy"] - st64.entropy) < 1e-5
    assert abs(st["loss_initial"] - st64.loss_initial) <= 1e-5 * max(1.0, abs(st64.loss_initial))
    err_dev, err_o32 = abs(st["step_size"] - st64.step_size), abs(st32.step_size - st64.step_size)
    assert err_dev <= 2.0 * err_o32 + 1e-3 * st64.step_size, (st["step_size"], st32.step_size, st64.step_size)
    assert st["constraint_val_final"] <= 0.01 and st["loss_final"] < st["loss_initial"]
    g64, _ = O.grad_f64_mt("policy", PS, p0, x, a.astype(np.uint8), adv)

    def residual(direction):
        d = np.asarray(direction, dtype=np.float64)
        fx, _ = O.grad_f64_mt("fvp", PS, p0, x, v=d.astype(np.float32))
        return np.linalg.norm(fx + 1e-5 * d - g64) / np.linalg.norm(g64)

    x_dev = (p0.astype(np.float64) - p_d.astype(np.float64)) / (st["step_scale"] * st["step_size"])
    x_one = (p0.astype(np.float64) - s["policy"].astype(np.float64)) / (s["trpo"]["step_scale"] * s["trpo"]["step_size"])
    r_dev, r_one, r_o32, r_o64 = residual(x_dev), residual(x_one), residual(sd32), residual(sd64)
    print("sharded TRPO backward error |Ax - g| / |g|: %d ranks %.4g, one rank %.4g, f32 oracle %.4g, f64 oracle %.4g" % (
        len(ranks), r_dev, r_one, r_o32, r_o64))
    # (three correct f32 evaluations of the same ten iterations — the f32 oracle, the one-rank device run, the sharded run
    # — differ from each other by a factor of a few in this residual: f32 CG has lost conjugacy by then, tests/
    # test_gpu_parity.py::test_trpo_update_default_config_vs_f64_truth quotes 0.32 / 0.25 / 0.046 on one problem and 0.040 /
    # 0.026 / 0.11 on another.  The bar is 3 x the worst of the yardsticks; a wrong operator, sign or weight gives O(1).)
    assert r_dev <= max(3.0 * max(r_o32, r_o64, r_one), 0.2)  # (the sharp bars on the vectors are check_probe_vectors)
    # critic: the oracle's Adam loop on the same samples and targets
    import ctypes as C
    ac = O.AdamCfg()
    O.lib().oracle_adam_cfg_default(C.byref(ac))
    ad = O.lib().oracle_adam_new(len(c0))
    losses_o = np.zeros(critic_steps, dtype=np.float32)
    c_o = c0.copy()
    O.lib().oracle_critic_update_f32(CS, O.f32p(c_o), ad, C.byref(ac), O.f32p(x), O.f32p(rtg), len(a), critic_steps,
                                     O.f32p(losses_o))
    O.lib().oracle_adam_free(ad)
    losses_d = ranks[0][0]["losses"]
    assert np.max(np.abs(losses_d - s["losses"]) / s["losses"]) < 1e-5
    assert np.max(np.abs(losses_d - losses_o) / losses_o) < 1e-5
    assert np.abs(ranks[0][0]["critic"] - c_o).max() < 2e-5 + 1e-3 * critic_steps * 1e-3


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_ranks_equal_one_rank(world):
    n_total, T = 512, 48
    single = launch(1, n_total, T, with_dqn=(world == 2))
    ranks = launch(world, n_total, T, with_dqn=(world == 2))
    # the collective really ran: gradient + Fisher-vector + line-search + critic (+ DQN) vectors, on every rank
    assert single[0]["allreduce_launches"] == 0
    assert len({ranks[r]["allreduce_launches"] for r in range(world)}) == 1
    assert ranks[0]["allreduce_launches"] > 2 * (1 + 11 + 1 + 10)
    for period in range(2):  # every rank holds identical replicas after every update
        for r in range(1, world):
            assert np.array_equal(ranks[0][period]["policy"], ranks[r][period]["policy"])
            assert np.array_equal(ranks[0][period]["critic"], ranks[r][period]["critic"])
    check_sharded_update_against_the_oracles([ranks[r] for r in range(world)], single[0])
    if world == 2:
        # DQN: both ranks step in lockstep and end with identical networks; global_steps counts all ranks' lanes
        assert np.array_equal(ranks[0]["dqn"]["q"], ranks[1]["dqn"]["q"])
        assert ranks[0]["dqn"]["global_steps"] == single[0]["dqn"]["global_steps"] == 60 * n_total
        assert np.all(np.isfinite(ranks[0]["dqn"]["losses"]))


def test_a_wrongly_weighted_shard_fails_the_sharded_parity_bars():
    """The negative of the test above: one rank of two contributes its gradient-sized vectors —
    the policy gradient, the Fisher-vector products, every critic gradient — with a wrong weight, what a wrong
    B_local / B_total on that rank would do (RELEARN_LOOPBACK_TEST_WEIGHT = "1:1.5": the in-process collective multiplies
    rank 1's vector by 1.5 before it sums; every rank still receives the same sums, so the job runs to its end).  The
    trajectories are still the single rank's and the replicas still identical — and the update is not the single rank's
    update: the bars of check_sharded_update_against_the_oracles must say so.  (Two ranks of unequal lane counts, the
    first form of this test, break the library's contract in a way that lets their line searches disagree and the job
    deadlock.)"""
    n_total, T = 512, 48
    single = launch(1, n_total, T, with_dqn=False)
    os.environ["RELEARN_LOOPBACK_TEST_WEIGHT"] = "1:1.5"
    try:
        skewed = launch(2, n_total, T, with_dqn=False)
    finally:
        os.environ.pop("RELEARN_LOOPBACK_TEST_WEIGHT", None)
    assert np.array_equal(skewed[0][0]["policy"], skewed[1][0]["policy"])  # (identical, and identically wrong)
    assert np.array_equal(np.concatenate([skewed[r][0]["adv"] for r in range(2)], axis=1), single[0][0]["adv"])
    with pytest.raises(AssertionError):
        check_sharded_update_against_the_oracles([skewed[0], skewed[1]], single[0])
    # a milder error on the critic's vectors only would still be caught by the per-step losses: here they are off by
    # far more than the 1e-5 bar from the second step on (the first loss is computed before any wrong step is taken)
    a, b = skewed[0][0], single[0][0]
    assert np.max(np.abs(a["losses"][1:] - b["losses"][1:]) / b["losses"][1:]) > 1e-4


def poisoned_torch(tmp_path):
    """a directory that makes `import torch` fail when it is first on PYTHONPATH"""
    d = os.path.join(str(tmp_path), "no_torch", "torch")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "__init__.py"), "w") as f:
        f.write("raise ImportError('torch is not available in this test')\n")
    return os.path.dirname(d)


def run_bench(world, extra_env=None, workload=("--envs", "2048", "--horizon", "32", "--critic-steps", "5"), limit=150,
              launcher="self"):
    """bench.py on a small workload, one process per rank: `launcher` "self" = `python bench.py --gpus N` (bench.py starts
    its own ranks), "torchrun" = exactly as the driver launches it (torch.distributed.run)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # ONE period: the rollouts of the two runs are then identical and only the order of the f32 sums differs (a second
    # period would start from policies that already differ by TRPO's amplified rounding, and drift apart from there)
    args = ["--gpus", str(world), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + list(workload)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RELEARN_RDZV_PORT"):
        env.pop(k, None)
    import socket
    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    if world == 1 or launcher == "self":
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py")] + args
    # own process group + a hard limit: a rank that hangs must not outlive the test
    proc = subprocess.Popen(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        raise AssertionError("bench.py --gpus %d did not finish within %d s" % (world, limit))

    class out:  # the fields the checks below read
        returncode = proc.returncode
    out.stdout, out.stderr = stdout, stderr
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()  # rank 0 prints ONE JSON line
    res = json.loads(lines[0])
    res["_stderr"] = out.stderr.decode()
    return res


def test_bench_as_two_processes_over_the_host_collective(tmp_path):
    """The multi-process launch path of bench.py — its own spawner (RANK / LOCAL_RANK / WORLD_SIZE as
    torch.distributed.run sets them), the standard-library control plane, lane sharding, barriers, max-over-ranks
    timing, one JSON line — rehearsed with two processes on this box's one GPU, in an environment where torch cannot be
    imported.  RCCL refuses two ranks on one device, so the data-plane collective is the library's host-staged one
    (rl_comm_init_host over the control plane); the RCCL call path itself is covered by the one-rank communicator test.
    The sharded job must report the same update statistics as the one-process job (same samples, other sum order)."""
    no_torch = {"PYTHONPATH": poisoned_torch(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", "")}
    one = run_bench(1, no_torch)
    two = run_bench(2, dict(no_torch, RELEARN_BENCH_SINGLE_DEVICE="1", RELEARN_BENCH_COMM="host"))
    assert two["n_gpus"] == 2 and two["config"]["n_envs_per_gpu"] == 1024 and two["config"]["n_envs_total"] == 2048
    assert "host-staged" in two["config"]["parallelism"] and "tcp" in two["config"]["parallelism"]
    assert two["cpu_baseline"] is None
    assert two["value"] > 0 and abs(two["value"] * two["ms_per_step"] * 1e-3 - 2048 * 32) < 1e-6 * 2048 * 32
    assert two["phases"]["allreduce"]["launches_per_step"] >= 5 + 11 + 2
    # the same launch with the peer-mailbox transport (both ranks map each other's mailbox on the one device)
    ipc = run_bench(2, dict(no_torch, RELEARN_BENCH_SINGLE_DEVICE="1", RELEARN_BENCH_COMM="ipc"))
    assert "mailboxes" in ipc["config"]["parallelism"], ipc["_stderr"][-1500:]
    assert ipc["phases"]["allreduce"]["launches_per_step"] >= 5 + 11 + 2
    assert one["replicas_identical"] is None and two["replicas_identical"] is True and ipc["replicas_identical"] is True
    for other in (two, ipc):
        a, b = one["last_update"], other["last_update"]
        assert a["trpo_status"] == b["trpo_status"]
        # one period of training with the f32 sums in another order: TRPO's CG amplifies rounding-level differences of
        # the Fisher-vector products, so the runs stay close, not identical
        assert abs(a["entropy"] - b["entropy"]) < 5e-3
        assert abs(a["critic_loss_last"] - b["critic_loss_last"]) < 5e-2 * a["critic_loss_last"]


def test_bench_under_torch_distributed_run_and_over_the_gloo_control_plane():
    """the driver's launch line (torch.distributed.run) lands on the same torch-free control plane; `--control gloo`
    (torch imported on request) stays available as a fallback"""
    import subprocess
    import sys
    # page torch in first (a cold image needs a minute or two for its first import), in a child process
    subprocess.run([sys.executable, "-c", "import torch"], check=True, timeout=280)
    tr = run_bench(2, {"RELEARN_BENCH_SINGLE_DEVICE": "1", "RELEARN_BENCH_COMM": "host"}, launcher="torchrun")
    assert tr["n_gpus"] == 2 and "tcp" in tr["config"]["parallelism"] and tr["replicas_identical"] is True
    gl = run_bench(2, {"RELEARN_BENCH_SINGLE_DEVICE": "1", "RELEARN_BENCH_COMM": "gloo"}, launcher="torchrun")
    assert gl["n_gpus"] == 2 and "gloo" in gl["config"]["parallelism"] and gl["replicas_identical"] is True
    assert tr["last_update"] == gl["last_update"]  # the same sums in the same (rank) order, whatever carries them


@pytest.mark.parametrize("order", ["torch_first", "engine_first"])
def test_rccl_is_bound_next_to_the_hip_runtime_in_use(order):
    """A multi-rank bench.py process imports torch (for its gloo control group) before the engine exists, so the library
    runs on the ROCm copy torch bundles; other hosts load the engine first and torch later, which maps a second copy.
    Either way RCCL must come from the directory of the libamdhip64 this library is bound to (rccl_load, abi.hip) — a
    communicator from the other copy fails with "unhandled cuda error".  scripts/rccl_with_torch_first.py creates a
    one-rank communicator and runs collective updates in a fresh process with the given import order."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.Popen([sys.executable, os.path.join(root, "scripts", "rccl_with_torch_first.py"), order], cwd=root,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=280)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        raise AssertionError("rccl_with_torch_first.py %s did not finish within 280 s" % order)
    text = out.decode()
    assert proc.returncode == 0 and ("%s OK trpo 0" % order) in text, text[-1500:]
    # the librccl the library bound (dladdr of its ncclAllReduce) sits next to the libamdhip64 the library runs on
    bound = [l for l in text.splitlines() if l.startswith("BOUND ")]
    assert bound, text[-1500:]
    rccl, hip = (kv.split("=", 1)[1] for kv in bound[0].split()[1:3])
    assert rccl and hip and os.path.realpath(os.path.dirname(rccl)) == os.path.realpath(os.path.dirname(hip)), bound


def _run_ipc_ranks(tmp_path, world, extra=(), env_extra=None, limit=200):
    """`world` processes of scripts/ipc_rank.py on this box's one GPU, handle files in a directory of their own"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "ipc_rank.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    d = os.path.join(str(tmp_path), "world%d%s" % (world, "_".join(extra)))
    os.makedirs(d)
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), d] + list(extra), cwd=root, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            import signal
            for q in procs:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            raise AssertionError("ipc_rank.py (world %d) did not finish within %d s" % (world, limit))
        outs.append(o.decode())
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
    return d, outs


def _ipc_results(tmp_path, world):
    d, _ = _run_ipc_ranks(tmp_path, world)
    return [dict(np.load(os.path.join(d, "out%d_of_%d.npz" % (r, world)))) for r in range(world)]


def _as_launch_result(npz):
    """the arrays scripts/ipc_rank.py saves, in the shape check_sharded_update_against_the_oracles reads"""
    t = npz["trpo0"]
    trpo = dict(loss_initial=float(t[0]), entropy=float(t[1]), step_size=float(t[2]), cg_iterations=int(t[3]),
                status=int(t[4]), loss_final=float(t[5]), constraint_val_final=float(t[6]), step_scale=float(t[7]),
                num_backtracks=int(t[8]))
    traj = dict(obs=npz["obs"], action=npz["action"], flag=npz["flag"], reward=npz["reward"])
    probe = dict(g=npz["probe_g"], hv=npz["probe_hv"], gc=npz["probe_gc"], loss=float(npz["probe_loss"]),
                 ent=float(npz["probe_ent"]), lc=float(npz["probe_lc"]))
    return {"policy_init": npz["policy_init"], "critic_init": npz["critic_init"], "probe": probe,
            0: dict(traj=traj, adv=npz["adv"], rtg=npz["rtg"], policy=npz["policy0"], critic=npz["critic0"], trpo=trpo,
                    losses=npz["losses0"])}


def _check_sharded_against_single(ranks, single):
    world = len(ranks)
    assert len({int(r["allreduce_launches"][0]) for r in ranks}) == 1
    assert ranks[0]["allreduce_launches"][0] > 2 * (1 + 11 + 1 + 6)
    for period in range(2):  # every rank ends every update with the same replica, bit for bit
        for r in ranks[1:]:
            assert np.array_equal(ranks[0]["policy%d" % period], r["policy%d" % period])
            assert np.array_equal(ranks[0]["critic%d" % period], r["critic%d" % period])
    # the same bars as the in-process group's ranks (oracle-based, no per-cent tolerances)
    check_sharded_update_against_the_oracles([_as_launch_result(r) for r in ranks], _as_launch_result(single),
                                             critic_steps=6)
    assert world >= 2


def test_two_processes_over_the_peer_mailbox_collective(tmp_path):
    """rl_comm_init_ipc: two PROCESSES on this box's one GPU exchange their mailbox handles and run sharded updates
    through the single-launch all-reduce (scripts/ipc_rank.py); the library's self-test (240 rounds of random payloads
    over every chunk and both slots) passes on both, both end with identical replicas, and the job agrees with the
    one-process run like the in-process loopback group does."""
    single = _ipc_results(tmp_path, 1)[0]
    assert single["allreduce_launches"][0] == 0
    _check_sharded_against_single(_ipc_results(tmp_path, 2), single)


def test_four_processes_over_the_peer_mailbox_collective(tmp_path):
    """the same with FOUR ranks: four rows per mailbox, sums of four terms in rank order, slot reuse with three peers
    that may each be a collective ahead or behind (VERDICT round 2, next 8 i)"""
    single = _ipc_results(tmp_path, 1)[0]
    _check_sharded_against_single(_ipc_results(tmp_path, 4), single)


def test_four_loopback_ranks_keep_identical_replicas():
    """four ranks in the in-process group: after the sequential and after the two-stream update every rank holds the
    same policy and critic, bit for bit"""
    quad = launch(4, 512, 32)
    for period in range(2):
        for r in range(1, 4):
            assert np.array_equal(quad[0][period]["policy"], quad[r][period]["policy"])
            assert np.array_equal(quad[0][period]["critic"], quad[r][period]["critic"])
        assert np.all(np.isfinite(quad[0][period]["losses"])) and quad[0][period]["trpo"]["status"] == ra.OPT_OK


@pytest.mark.parametrize("what", ["desert", "desert_trpo"])
def test_a_missing_peer_fails_the_mailbox_collective_without_touching_the_replica(tmp_path, what):
    """A rank that never joins a collective: the others' waits end at the wall-clock bound (here 1.5 s), the exchange
    returns before anything is stored — the critic's parameters are bit-identical to what they were —, the engine's
    error word is sticky (the next collective fails in milliseconds, not after another bound), and both surface as
    RL_ERR_COMM (comm_ipc.hpp)."""
    # `desert_trpo`: the same through rl_trpo_update, whose all-reduces are launches of their own — the kernels that write
    # parameters (line-search candidates, the final rollback) test the error word, so the policy is bit-identical to what
    # it was although the vector held local sums (comm_ipc.hip)
    _, outs = _run_ipc_ranks(tmp_path, 3, extra=("512", "48", what), env_extra={"RELEARN_IPC_TIMEOUT_MS": "1500"},
                             limit=120)
    assert "deserted" in outs[2]
    for o in outs[:2]:
        assert "saw the timeout" in o, o[-1500:]
        secs = float(o.split("saw the timeout after ")[1].split(" s")[0])
        fast = float(o.split("failed fast in ")[1].split(" s")[0])
        assert 1.0 < secs < 30.0 and fast < 1.0, o[-500:]


# ---------------------------------------------------------------- config 4 at its exact split: 8 ranks x 8,192 lanes
# (BASELINE.json configs[3]: 65,536 envs over 8 GPUs, T = 128, 80 critic steps; replaces the thread fan-out of
# src/simulation/train.rs:98-158,180).  A one-GPU box can host the eight ranks as eight engines of one process (the
# in-process loopback collective) or — the pool admits at most six GPU processes — as four processes of two ranks each
# over the peer mailboxes.  The driver's 8-process launch of bench.py is rehearsed with four processes for the same
# reason (16,384 lanes per rank, the full workload otherwise).
CONFIG4 = dict(n_total=65536, T=128, critic_steps=80, periods=2)


def _config4_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ipc_multi", os.path.join(root, "scripts", "ipc_multi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _config4_loopback(world):
    mod = _config4_module()
    os.environ["RELEARN_LOOPBACK_COMM"] = "1"
    try:
        uid = ra.comm_unique_id()
        out = {}

        def run(rank):
            try:
                eng = ra.Engine(0)
                if world > 1:
                    eng.comm_init(rank, world, uid)
                out[rank] = mod.config4_rank(eng, rank, world, save_obs=(world == 1), **CONFIG4)
            except BaseException as exc:
                out[rank] = exc
                raise

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=280)
        for r in range(world):
            assert r in out and not isinstance(out[r], BaseException), out.get(r)
        return [out[r] for r in range(world)]
    finally:
        os.environ.pop("RELEARN_LOOPBACK_COMM", None)


@pytest.fixture(scope="module")
def config4_single():
    return _config4_loopback(1)[0]


_CONFIG4_YARDSTICK = {}


def _config4_yardstick(single):
    """f64 quantities of the first period's TRPO problem at the full size (8.39 M samples), computed once per session with
    the OpenMP oracle (O.grad_f64_mt): the gradient, ten CG iterations of conjugate_gradient.rs:371-403 in f64 with its
    Fisher-vector products, the step size they give — and the one-rank DEVICE result's distance from them, the yardstick a
    correct f32 evaluation sets (the f32 oracle's scalar TRPO would take minutes here)."""
    if _CONFIG4_YARDSTICK:
        return _CONFIG4_YARDSTICK
    tr = dict(obs=single["obs"], action=single["action"])
    x, a = O.flat_samples(tr)
    adv, p0 = single["adv"].reshape(-1), single["policy_init"]
    g64, _ = O.grad_f64_mt("policy", PS, p0, x, a.astype(np.uint8), adv)
    reg = 1e-5

    def op(d):
        fx, _ = O.grad_f64_mt("fvp", PS, p0, x, v=np.asarray(d, dtype=np.float32))
        return fx + reg * np.asarray(d, dtype=np.float64)

    xs, r, pv = np.zeros_like(g64), g64.copy(), g64.copy()
    rr = float(r @ r)
    for _ in range(10):
        z = op(pv)
        alpha = rr / float(pv @ z)
        xs += alpha * pv
        r -= alpha * z
        new_rr = float(r @ r)
        if new_rr < 1e-10:
            break
        pv = r + (new_rr / rr) * pv
        rr = new_rr
    ss64 = float(np.sqrt(1.0 / (float(xs @ op(xs)) + 1e-8) * 0.01 * 2.0))

    def residual(policy_after, trpo):
        d = (p0.astype(np.float64) - policy_after.astype(np.float64)) / (float(trpo[6]) * float(trpo[2]))
        return float(np.linalg.norm(op(d) - g64) / np.linalg.norm(g64))

    _CONFIG4_YARDSTICK.update(g64=g64, ss64=ss64, residual=residual, r64=float(np.linalg.norm(op(xs) - g64) / np.linalg.norm(g64)),
                              r_single=residual(single["policy0"], single["trpo0"]),
                              ss_err_single=abs(float(single["trpo0"][2]) - ss64))
    return _CONFIG4_YARDSTICK


def _check_config4(ranks, single):
    """rollouts bit-identical lane for lane; every rank's replica identical after both periods; the first period's update
    held to the bars of check_sharded_update_against_the_oracles with the one-rank device result as the f32 yardstick:
    scalars to 1e-6, step size no farther from the f64 step size than twice the one-rank run is (or 2 %), backward error of the
    step direction under the f64 operator no more than 3 x the worse of the one-rank run's and the f64 iteration's own,
    the loss before every critic step to 1e-5, the critic's parameters to the single-rank test's bound"""
    world = len(ranks)
    assert world == 8 and ranks[0]["action"].shape == (128, 8192)
    for f in ("action", "flag", "adv"):
        assert np.array_equal(np.concatenate([r[f] for r in ranks], axis=1), single[f]), f
    for period in range(CONFIG4["periods"]):
        for r in ranks[1:]:
            assert np.array_equal(ranks[0]["policy%d" % period], r["policy%d" % period]), period
            assert np.array_equal(ranks[0]["critic%d" % period], r["critic%d" % period]), period
            assert np.array_equal(ranks[0]["losses%d" % period], r["losses%d" % period]), period
        assert ranks[0]["trpo%d" % period][4] == ra.OPT_OK
    # 1 gradient + 11 Fisher-vector products + >= 2 line-search pairs + 80 critic steps, per period, on every rank
    assert len({int(r["allreduce_launches"][0]) for r in ranks}) == 1
    assert ranks[0]["allreduce_launches"][0] >= 2 * (1 + 11 + 2 + 80)
    a, b = ranks[0]["trpo0"], single["trpo0"]
    assert abs(a[0] - b[0]) < 1e-6 and abs(a[1] - b[1]) < 1e-6 and a[3] == b[3] and a[4] == b[4]
    y = _config4_yardstick(single)
    # (the one-rank run is ONE f32 evaluation and may land anywhere inside the f32 scatter — 0.1 % from the f64 step size
    # on one run where the eight ranks landed 1.0 % from it: the yardstick is twice its distance or 2 % of the step,
    # whichever is larger; a shard of 1/8 of the samples weighted 1.5 x moves the step size by 6 %)
    assert abs(float(a[2]) - y["ss64"]) <= max(2.0 * y["ss_err_single"], 0.02 * y["ss64"]), (a[2], b[2], y["ss64"])
    r8 = y["residual"](ranks[0]["policy0"], a)
    print("config 4 TRPO backward error |Ax - g| / |g|: 8 ranks %.4g, 1 rank %.4g, f64 CG %.4g" % (r8, y["r_single"], y["r64"]))
    # (0.041 / 0.094 / 0.0089 and 0.046 / 0.010 / 0.0089 for eight ranks / one rank / the f64 iteration on two builds of
    # the library: ten f32 CG iterations scatter by a factor of ten here.  The bar keeps out what is wrong by O(1); the
    # sharp bars are the vectors themselves, below, and the per-step critic losses)
    assert r8 <= max(3.0 * max(y["r_single"], y["r64"]), 0.2)
    check_probe_vectors({k[6:]: ranks[0][k] for k in ranks[0] if k.startswith("probe_")},
                        {k[6:]: single[k] for k in single if k.startswith("probe_")})
    assert a[8] <= 0.01 and a[7] < a[0]  # the accepted step obeys the trust region and improves the surrogate
    steps = CONFIG4["critic_steps"]
    assert np.max(np.abs(ranks[0]["losses0"] - single["losses0"]) / single["losses0"]) < 1e-5
    assert np.abs(ranks[0]["critic0"] - single["critic0"]).max() < 2e-5 + 1e-3 * steps * 1e-3


def test_eight_loopback_ranks_at_the_config4_split(config4_single):
    """(a) eight engines of one process x 8,192 lanes against one engine x 65,536 lanes, two periods of
    rl_actor_critic_update at the full workload"""
    _check_config4(_config4_loopback(8), config4_single)


def _mailbox_ranks_once(d, limit, ipc_timeout_ms):
    """four processes x two ranks of scripts/ipc_multi.py into directory `d`; (finished and all exit codes 0, what the
    processes said).  Output goes into files: what a process had said is still there when the test has to end it."""
    import signal
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "ipc_multi.py")
    os.makedirs(d)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RELEARN_IPC_TIMEOUT_MS=str(ipc_timeout_ms))
    args = [str(CONFIG4[k]) for k in ("n_total", "T", "critic_steps", "periods")]
    logs = [os.path.join(d, "proc%d.log" % p) for p in range(4)]
    procs = []
    for p in range(4):
        with open(logs[p], "wb") as out:  # (the child keeps its own descriptor)
            procs.append(subprocess.Popen([sys.executable, "-u", script, str(p), "4", "2", d] + args, cwd=root, env=env,
                                          stdout=out, stderr=subprocess.STDOUT, start_new_session=True))

    def tail(path):
        with open(path, errors="replace") as f:
            return f.read()[-1500:]

    def said():
        return "\n".join("--- proc %d (exit %s)\n%s" % (i, q.poll(), tail(logs[i])) for i, q in enumerate(procs))

    deadline = time.time() + limit
    for p in procs:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            for q in procs:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
            for q in procs:
                q.wait()
            return False, "ipc_multi.py (4 x 2 ranks) did not finish within %d s\n" % limit + said()
    return all(p.returncode == 0 for p in procs), said()


def test_eight_mailbox_ranks_at_the_config4_split(tmp_path, config4_single):
    """(b) the same eight ranks over the peer-mailbox transport: four processes of two ranks each (the pool admits six
    GPU processes), every rank running the library's 240-round self-test first.  Ranks of one process reach each other's
    mailboxes by address, ranks of other processes through IPC handles.
    Eight ranks share ONE card's hardware queues here, each with kernels that spin on words their peers' kernels write: the
    job itself takes about a second, and about one in five — inside the whole suite only, never alone — has not FINISHED
    (a rank's wait bound hit or a process still starting after minutes; round 6 saw it with a 30 s and with a 150 s bound).  That is the
    rehearsal's property, not the transport's on eight cards, so an attempt that does not finish is repeated ONCE, with a
    warning that carries what its processes had said; a finished attempt is judged as it is — numbers are never retried."""
    import warnings
    # (RELEARN_TEST_FIRST_ATTEMPT_S: a rehearsal of the second attempt — the first one is ended early, mid-kernel)
    first_limit = float(os.environ.get("RELEARN_TEST_FIRST_ATTEMPT_S", "100"))
    ok, said = _mailbox_ranks_once(os.path.join(str(tmp_path), "config4"), limit=first_limit, ipc_timeout_ms=60000)
    d = os.path.join(str(tmp_path), "config4")
    if not ok:
        warnings.warn("the 8-rank mailbox rehearsal did not finish at the first attempt:\n" + said)
        d = os.path.join(str(tmp_path), "config4_second")
        ok, said = _mailbox_ranks_once(d, limit=150, ipc_timeout_ms=90000)
    assert ok, said
    ranks = [dict(np.load(os.path.join(d, "out%d_of_8.npz" % r))) for r in range(8)]
    _check_config4(ranks, config4_single)


def test_bench_as_four_processes_at_the_full_workload(tmp_path):
    """(c) `python bench.py --gpus 4`, FOUR processes (the pool's limit is six on one card) on this box's one GPU at the
    metric's workload (65,536 envs, T = 128, 80 critic steps), over the mailboxes: started by bench.py itself, started by
    torch.distributed.run as the driver does, and started where `import torch` fails — and once over the host-staged
    collective.  The line says n_gpus 4, the replicas are identical, and `allreduce_per_rank` is filled for every rank:
    what the first real multi-GPU run will be read by."""
    import subprocess
    import sys
    subprocess.run([sys.executable, "-c", "import torch"], check=True, timeout=280)
    no_torch = {"PYTHONPATH": poisoned_torch(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", "")}
    for comm, launcher, extra in (("ipc", "self", {}), ("ipc", "torchrun", {}), ("ipc", "self", no_torch),
                                  ("host", "self", no_torch)):
        res = run_bench(4, dict(extra, RELEARN_BENCH_SINGLE_DEVICE="1", RELEARN_BENCH_COMM=comm),
                        workload=("--steps", "2", "--warmup", "1"), limit=280, launcher=launcher)
        assert res["n_gpus"] == 4 and res["config"]["n_envs_per_gpu"] == 16384 and res["config"]["n_envs_total"] == 65536
        assert res["replicas_identical"] is True, res["_stderr"][-1500:]
        assert ("host-staged" if comm == "host" else "mailboxes") in res["config"]["parallelism"], res["_stderr"][-1500:]
        per_rank = res["allreduce_per_rank"]
        assert len(per_rank) == 4 and all(p is not None and p["rank"] == i for i, p in enumerate(per_rank))
        # (over the mailboxes the critic chain's 80 exchanges run INSIDE its reduce + Adam launches)
        assert all(p["launches_per_step"] >= 1 + 11 + 2 + (80 if comm == "host" else 0) for p in per_rank)
        assert res["last_update"]["trpo_status"] == ra.OPT_OK


def test_two_loopback_ranks_pipelined_periods_keep_identical_replicas():
    """rl_actor_critic_update_begin / _finish with a collective: two ranks run three periods with the next rollout
    enqueued under the critic chain (two trajectories per rank); both chains' all-reduces travel on their own channels,
    every rank ends every period with the same policy and critic, and the sequence equals the same ranks running
    rl_actor_critic_update without the split, bit for bit (same launches, same sums; only the host's enqueue order
    differs)."""
    os.environ["RELEARN_LOOPBACK_COMM"] = "1"
    try:
        def job(pipelined):
            uid = ra.comm_unique_id()
            out = {}

            def run(rank):
                try:
                    eng = ra.Engine(0)
                    eng.comm_init(rank, 2, uid)
                    n = 512
                    env = ra.CartPoleEnv(eng, n, max_steps=40, lane_offset=rank * n, seed_env=5, seed_actor=6)
                    pol, cri = ra.Mlp(eng, 5, H, 2), ra.Mlp(eng, 5, H, 1)
                    pol.init(2)
                    cri.init(3)
                    opt = ra.Adam(cri)
                    trajs = [ra.Trajectory(eng, n, 32, 5), ra.Trajectory(eng, n, 32, 5)]
                    ccfg = ra.values_opt_config_default()
                    ccfg.opt_steps_per_update = 8
                    res, pending = [], None
                    for k in range(3):
                        tr = trajs[k & 1]
                        ra.rollout(env, pol, tr)
                        ra.gae(tr, cri, 0.99, 0.95)
                        if pipelined:
                            if pending is not None:
                                cst, losses = ra.actor_critic_update_finish(pending, want_losses=True)
                                res[-1] += (cri.get_params(), losses)
                            pst = ra.actor_critic_update_begin(pol, cri, opt, tr, None, ccfg)
                            res.append((pst.as_dict(),))
                            pending = tr
                        else:
                            pst, cst, losses = ra.actor_critic_update(pol, cri, opt, tr, None, ccfg, want_losses=True)
                            res.append((pst.as_dict(), cri.get_params(), losses))
                    if pipelined:
                        cst, losses = ra.actor_critic_update_finish(pending, want_losses=True)
                        res[-1] += (cri.get_params(), losses)
                    out[rank] = (res, pol.get_params())
                except BaseException as exc:
                    out[rank] = exc
                    raise

            threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=200)
            for r in range(2):
                assert r in out and not isinstance(out[r], BaseException), out.get(r)
            return out

        plain, piped = job(False), job(True)
    finally:
        os.environ.pop("RELEARN_LOOPBACK_COMM", None)
    for run in (plain, piped):  # identical replicas on both ranks, every period
        assert np.array_equal(run[0][1], run[1][1])
        for (s0, c0, l0), (s1, c1, l1) in zip(run[0][0], run[1][0]):
            assert s0 == s1 and np.array_equal(c0, c1) and np.array_equal(l0, l1)
    assert np.array_equal(plain[0][1], piped[0][1])
    for (s0, c0, l0), (s1, c1, l1) in zip(plain[0][0], piped[0][0]):
        assert s0 == s1 and np.array_equal(c0, c1) and np.array_equal(l0, l1)


def test_bench_pipelined_periods_report_the_same_update_as_the_plain_sequence():
    """`bench.py --pipeline` (period k + 1's rollout enqueued under critic chain k, two trajectories) is the same
    computation as the default sequence: after four periods the last update's statistics are identical to the digit."""
    common = ("--envs", "4096", "--horizon", "32", "--critic-steps", "10", "--steps", "4", "--warmup", "0")
    plain = run_bench(1, workload=common)
    piped = run_bench(1, workload=common + ("--pipeline",))
    assert "two trajectories" in piped["config"]["pipeline"] and plain["config"]["pipeline"] == "none"
    assert plain["last_update"] == piped["last_update"]
    assert plain["last_update"]["trpo_status"] == ra.OPT_OK


def test_a_stalled_rank_ends_the_bench_with_the_phase_it_was_stuck_in():
    """bench.py's watchdog (multi-rank jobs): one of two ranks never joins the first collective of the warm-up; both ranks
    must leave within the bound (RELEARN_BENCH_TIMEOUT, here 20 s after the rendezvous) with exit code 4 and a line on
    stderr that names the phase, and the launcher passes that on as a non-zero exit without a result line — a job that
    stalls on a real node explains itself instead of holding it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RELEARN_BENCH_SINGLE_DEVICE="1", RELEARN_BENCH_COMM="host",
               RELEARN_BENCH_TEST_STALL_RANK="1", RELEARN_BENCH_TIMEOUT="20")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RELEARN_RDZV_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--envs", "2048", "--horizon", "32", "--critic-steps", "5", "--no-cpu-baseline"]
    proc = subprocess.Popen(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=200)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        raise AssertionError("the stalled job did not end by itself")
    text = err.decode()
    assert proc.returncode != 0
    assert "stuck in phase `warm-up periods (host collective)`" in text and "giving up (exit 4)" in text, text[-2000:]
    assert not [l for l in out.decode().splitlines() if l.startswith("{")]  # no result line from a job that did not finish
