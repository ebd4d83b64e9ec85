#!/usr/bin/env python3
"""bench.py — env-steps/sec of CartPole MLP-TRPO on MI355X (BASELINE.json metric).

One "step" = one training period of the hot path: a T-step fused rollout of every lane, critic values +
GAE/return scan, the TRPO policy update (gradient, 10 CG iterations with Fisher-vector products, step size,
backtracking line search) and 80 full-batch Adam steps on the critic — inputs resident in HBM, nothing
skipped.  The workload at every GPU count is the configuration the metric is quoted on: 65,536 CartPole
envs (VisibleStepLimit(500)), T = 128, policy 5-128-2, critic 5-128-1; with N GPUs each rank owns 65,536/N
lanes (strong scaling) and every reduced gradient / Hessian-vector / scalar vector is all-reduced with RCCL.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PMC_SUMMARY = "r06_pmc_65536_summary.json"  # profiles/: counter passes of the build named inside (scripts/pmc_summary.py)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: f32 MFMA peak = f32 vector peak
BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def lib_sha16():
    """identity of the kernel library this process runs (what the committed counter summaries are matched against)"""
    import hashlib
    import relearn_amd as ra
    return hashlib.sha256(open(ra.LIB_PATH, "rb").read()).hexdigest()[:16]


def build_record():
    """which binary this line belongs to: the library's identity, and what the last build() did with it"""
    import relearn_amd as ra
    rec = {"lib_sha16": lib_sha16()}
    try:
        last = json.load(open(ra.BUILD_RECORD))
        rec["last_build"] = last
        rec["last_build_is_this_library"] = last.get("lib_sha16") == rec["lib_sha16"]
    except Exception:
        rec["last_build"] = None
    return rec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=65536, help="total env lanes over all GPUs")
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--critic-steps", type=int, default=80)
    ap.add_argument("--max-episode-steps", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm", choices=("rccl", "ipc", "host", "gloo"), default=os.environ.get("RELEARN_BENCH_COMM", "rccl"),
                    help="data-plane collective for N > 1: RCCL on the engine stream (default); `ipc` = the library's "
                         "single-launch all-reduce over peer-mapped mailboxes (self-tested at start-up, RCCL if the test "
                         "fails); `host` = the host-staged fallback over the control plane (also taken when RCCL cannot "
                         "be initialised); `gloo` = `host` with `--control gloo`")
    ap.add_argument("--control", choices=("tcp", "gloo"), default=os.environ.get("RELEARN_BENCH_CONTROL", "tcp"),
                    help="control plane for N > 1 (id hand-out, barriers, max of the timing): `tcp` = the standard-library "
                         "star on rank 0 (relearn_amd/rendezvous.py; no rank imports torch); `gloo` = a torch.distributed "
                         "gloo group, imported only when asked for")
    ap.add_argument("--no-kernel-profile", action="store_true",
                    help="do not wrap launches in HIP events inside the timed region")
    ap.add_argument("--profile-steps", type=int, default=1,
                    help="number of timed periods (the last ones) whose launches are wrapped in HIP events")
    ap.add_argument("--serial-update", action="store_true",
                    help="run the TRPO chain and the critic chain of every period one after the other on one stream "
                         "(default: side by side on two streams, rl_actor_critic_update)")
    ap.add_argument("--pipeline", action="store_true",
                    help="start period k+1's rollout under period k's critic chain (rl_actor_critic_update_begin / "
                         "_finish over a pair of trajectories).  Off by default: it buys nothing on this part — a critic "
                         "step is one 8-wave x 252-register workgroup per CU and cannot be placed on a CU that holds a "
                         "rollout wave, so the two take turns (DESIGN 7b, profiles/r05_pipeline_timeline_8192.csv)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch path rehearsal without a GPU: the ranks meet on the control plane, run one of each of its "
                         "collectives, rank 0 prints one JSON line, nothing else runs (tests/test_bench_launch.py)")
    ap.add_argument("--cpu-sample-steps", type=int, default=262144,
                    help="total env-steps of the bounded CPU-baseline sample (split over the host cores)")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle's train_parallel-structured CPU path (one worker thread per host core running the scalar
    Steps::step loop, then the update), timed on this box's host cores on a bounded sample.

    The update is timed three ways, because the reference's agent runs it on one Rust thread while libtorch splits every
    batched matmul over its intra-op pool (examples/cartpole-trpo.rs:44: num_cpus::get() workers; actor_critic.rs:176-211):
      * on one thread, over a PREFIX of the sample that costs a few seconds (a side figure, scaled by sample count — every
        pass of the update visits every sample once);
      * with every full-batch pass split over 16, 64 and all the cores of this process's affinity mask, on a quarter of
        the sample: which count is fastest depends on the cores the job really gets, so it is MEASURED here;
      * with the fastest of those counts over the whole sample: the leg `value` is computed from."""
    import ctypes as C

    import oracle as O
    t_start = time.time()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    H = args.hidden
    ps, cs = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)
    pp, cp = O.mlp_init(ps, 2), O.mlp_init(cs, 3)
    L = O.lib()
    st = O.PeriodStats()
    per_thread = max(64, args.cpu_sample_steps // cores)
    sample = L.oracle_cpu_sample_collect(0, cores, per_thread, 100, args.max_episode_steps, H, O.f32p(pp), O.f32p(cp),
                                         C.byref(st))
    steps, rollout_s = int(st.steps), st.rollout_seconds
    # one thread: a prefix sized from a calibration on 1/64 of the sample so that the leg costs <= ~2 s
    st1 = O.PeriodStats()
    n_cal = max(1024, steps // 64)
    cal_s = L.oracle_cpu_sample_update_one_thread(sample, n_cal, args.critic_steps, C.byref(st1))
    n_one = int(min(steps, max(n_cal, n_cal * 2.0 / max(cal_s, 1e-6))))
    one_s = L.oracle_cpu_sample_update_one_thread(sample, n_one, args.critic_steps, C.byref(st1))
    one_thread_full_s = one_s * steps / n_one
    # the passes over several thread counts (quarter sample), then the fastest count over the whole sample
    if os.environ.get("RELEARN_BENCH_USABLE_CORES"):
        counts = [max(1, min(cores, int(os.environ["RELEARN_BENCH_USABLE_CORES"])))]
        how = "RELEARN_BENCH_USABLE_CORES"
    else:
        counts = sorted({min(cores, 16), min(cores, 64), cores})
        how = "measured: the update's passes over a quarter of the sample with %s threads, fastest taken" % (
            " / ".join(str(c) for c in counts))
    probe = {}
    for c in counts:  # ascending
        if probe and min(probe.values()) * 1.25 < probe[max(probe)]:
            # the previous, wider pool was already clearly slower than a narrower one: this job does not get more cores
            # than that, and a pool far wider than its cores spends every one of the ~100 barriers of an update in the
            # scheduler (256 threads on a 16-core share: 18 s for this probe, measured) — not run
            probe[c] = None
            continue
        L.oracle_cpu_sample_update_intraop(sample, 2048, 2, c)  # (the OpenMP pool of this width exists before the clock)
        probe[c] = L.oracle_cpu_sample_update_intraop(sample, max(1024, steps // 4), args.critic_steps, c)
    ran = {c: t for c, t in probe.items() if t is not None}
    usable = min(ran, key=ran.get)
    intraop_s = L.oracle_cpu_sample_update_intraop(sample, steps, args.critic_steps, usable)
    L.oracle_cpu_sample_free(sample)
    # HOT LOOP A alone, long enough to be a measurement: a short calibration run, then >= 2 s of stepping on every
    # thread (thread start-up outside the clock; buffers cleared every 8,192 steps)
    got = C.c_uint64()
    cal = L.oracle_cartpole_rollout_only(1, cores, 20000, 8192, args.max_episode_steps, H, O.f32p(pp), C.byref(got))
    per_thread_ro = int(min(max(2.3 * 20000 / max(cal, 1e-6), 50000), 20_000_000))
    for attempt in range(2):  # (the calibration run is cold: repeat once with the measured rate if the leg lasted < 2 s)
        ro_s = L.oracle_cartpole_rollout_only(2 + attempt, cores, per_thread_ro, 8192, args.max_episode_steps, H,
                                              O.f32p(pp), C.byref(got))
        if ro_s >= 2.0:
            break
        per_thread_ro = int(min(per_thread_ro * 2.4 / max(ro_s, 1e-3), 40_000_000))
    ro_steps = int(got.value)
    return {
        "value": steps / (rollout_s + min(intraop_s, one_thread_full_s)),
        "value_definition": "v3 (round 6): sample steps / (rollout + update), the update with its full-batch passes split "
                            "over the MEASURED fastest thread count (round 5: a fixed 16; rounds 1-4: one thread = "
                            "value_single_threaded_update)",
        "value_single_threaded_update": steps / (rollout_s + one_thread_full_s),
        "update_seconds": {"one_thread_scaled_to_the_sample": one_thread_full_s,
                           "one_thread_measured": {"samples": n_one, "seconds": one_s},
                           "passes_over_%d_threads" % usable: intraop_s,
                           "quarter_sample_probe": {str(c): (probe[c] if probe[c] is not None else
                                                                     "not run: a narrower pool was already >= 1.25x faster "
                                                                     "than the one before this") for c in counts}},
        "usable_cores": usable,
        "usable_cores_how": how,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "build": "gcc -O3 -ffp-contract=off -mavx2 -mfma -fopenmp (oracle/Makefile)",
        "sample": "%d worker threads x >=%d scalar Steps::step steps (%d steps, %d episodes) in %.2f s, then TRPO + %d Adam "
                  "steps with every full-batch pass split over %d threads in %.2f s (`value`); on one thread %.2f s for "
                  "the first %d samples = %.1f s for the sample" % (
                      cores, per_thread, steps, int(st.episodes), rollout_s, args.critic_steps, usable, intraop_s, one_s,
                      n_one, one_thread_full_s),
        "rollout_only_steps_per_s": ro_steps / max(ro_s, 1e-9),
        "rollout_only_sample": "%d threads x %d scalar Steps::step steps in %.2f s (worker threads started before the "
                               "clock)" % (cores, per_thread_ro, ro_s),
        "wall_s": time.time() - t_start,
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children — each with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in its environment, exactly what torch.distributed.run would give it — wait for them, pass
    rank 0's one JSON line on, and leave non-zero if any rank did.  The process-level form of train_parallel's scoped
    thread fan-out and join (/root/reference/src/simulation/train.rs:98-158,180).  This process never touches the GPU
    (nothing here imports the engine), and no process is ever replaced: the ranks are fresh children."""
    import signal
    import socket
    import subprocess
    import threading
    with socket.socket() as sock:  # a free port for the ranks' control plane
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    limit = float(os.environ.get("RELEARN_BENCH_TIMEOUT", "420")) + 240.0  # the ranks' own watchdogs fire first

    def die_with_parent():  # a killed launcher must not leave ranks holding GPUs
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)  # PR_SET_PDEATHSIG
        except Exception:
            pass

    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RELEARN_RDZV_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE, preexec_fn=die_with_parent))
    lines = [[] for _ in procs]

    def pump(i):  # rank 0's stdout is the job's stdout; anything another rank prints goes to stderr
        for raw in procs[i].stdout:
            text = raw.decode(errors="replace")
            lines[i].append(text)
            (sys.stdout if i == 0 else sys.stderr).write(text)
            (sys.stdout if i == 0 else sys.stderr).flush()

    pumps = [threading.Thread(target=pump, args=(i,), daemon=True) for i in range(len(procs))]
    for t in pumps:
        t.start()
    t0, first_bad, codes = time.time(), None, [None] * len(procs)
    while any(c is None for c in codes):
        for i, pr in enumerate(procs):
            if codes[i] is None:
                codes[i] = pr.poll()
                if codes[i] not in (None, 0) and first_bad is None:
                    first_bad = time.time()
                    print("bench.py: rank %d left with exit code %d" % (i, codes[i]), file=sys.stderr, flush=True)
        # a rank that failed: the others notice through the control plane / their watchdogs; after a grace period (or at
        # the overall limit) the ranks still running — this process's own children, by pid — are ended
        if (first_bad is not None and time.time() - first_bad > 45.0) or time.time() - t0 > limit:
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    pr.kill()
                    codes[i] = pr.wait()
            if first_bad is None:
                print("bench.py: the ranks did not finish within %.0f s" % limit, file=sys.stderr, flush=True)
            break
        time.sleep(0.05)
    for t in pumps:
        t.join(timeout=5)
    bad = [c for c in codes if c != 0]
    if bad:
        return bad[0] if bad[0] and bad[0] > 0 else 1
    if len([l for l in lines[0] if l.startswith("{")]) != 1:
        print("bench.py: rank 0 did not print exactly one result line", file=sys.stderr, flush=True)
        return 1
    return 0


class GlooControl:
    """`--control gloo`: the control plane's interface (relearn_amd/rendezvous.py) over a torch.distributed gloo group —
    optional, imported only when asked for; the default control plane needs no torch"""

    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def all_gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def broadcast(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def all_min(self, x):
        return min(self.all_gather(x))

    def all_max(self, x):
        return max(self.all_gather(x))

    def all_reduce_sum_f32(self, array):
        self.dist.all_reduce(self.torch.from_numpy(array))
        return array

    def close(self):
        self.dist.destroy_process_group()


def main():
    args = parse_args()
    if args.comm == "gloo":
        args.comm, args.control = "host", "gloo"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # (before anything that could touch a GPU)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    # (multi-process GPU work on this driver stack needs dmabuf IPC: RCCL's peer mappings and the mailbox handles fail
    # with "hipIpcGetMemHandle: invalid argument" otherwise; must be in the environment before the HIP runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import relearn_amd as ra

    # A multi-rank job that stalls (a collective some rank never joins) must end, and say where: every rank runs a
    # watchdog that prints the phase it is stuck in and leaves with a non-zero exit instead of holding the node until the
    # launcher's own limit.  RELEARN_BENCH_TIMEOUT (seconds, default 420) bounds everything after the rendezvous.
    phase = {"name": "start", "since": time.time()}

    def enter(name):
        phase["name"], phase["since"] = name, time.time()

    watchdog_all = None
    ctl = None
    if world > 1:
        import threading

        def stalled():
            print("bench.py: rank %d of %d: no end after %s s, stuck in phase `%s` for %.0f s — giving up (exit 4)" % (
                rank, world, os.environ.get("RELEARN_BENCH_TIMEOUT", "420"), phase["name"],
                time.time() - phase["since"]), file=sys.stderr, flush=True)
            os._exit(4)

        # control plane only (unique-id exchange, barrier, max-reduce of the timing); the data-plane collective is RCCL
        # (or the peer mailboxes) called from the library on its own HIP stream.  Standard library by default: no rank
        # imports torch, so no rank maps a second HIP runtime next to the one the engine runs on.
        enter("rendezvous (%s control plane)" % args.control)
        if args.control == "gloo":
            ctl = GlooControl(rank, world)
        else:
            from relearn_amd import rendezvous
            ctl = rendezvous.from_env()
        watchdog_all = threading.Timer(float(os.environ.get("RELEARN_BENCH_TIMEOUT", "420")), stalled)
        watchdog_all.daemon = True
        watchdog_all.start()

    if args.rendezvous_only:
        out = {"rendezvous_only": True, "n_gpus": world, "control": args.control if world > 1 else None,
               "torch_imported": "torch" in sys.modules}
        if ctl is not None:
            seen = ctl.all_gather({"rank": rank, "local_rank": local_rank, "pid": os.getpid(),
                                   "torch_imported": "torch" in sys.modules})
            ctl.barrier()
            probe = np.full(5, float(rank + 1), dtype=np.float32)
            ctl.all_reduce_sum_f32(probe)
            out.update(ranks=seen, max_rank=ctl.all_max(rank), min_rank=ctl.all_min(rank), sum_probe=float(probe[0]),
                       uid_len=len(ctl.broadcast(b"\x07" * 128 if rank == 0 else None, src=0)))
            ctl.barrier()
            ctl.close()
            watchdog_all.cancel()
        if rank == 0:
            print(json.dumps(out), flush=True)
        return

    assert args.envs % world == 0, "envs must divide evenly over GPUs"
    n_local = args.envs // world
    T, H = args.horizon, args.hidden

    # one process per GPU; RELEARN_BENCH_SINGLE_DEVICE=1 puts every rank on device 0 (a rehearsal of the multi-process
    # path on a one-GPU box, only meaningful with --comm gloo: RCCL refuses two ranks on one device)
    device = 0 if os.environ.get("RELEARN_BENCH_SINGLE_DEVICE") else local_rank
    enter("engine on device %d" % device)
    eng = ra.Engine(device)
    comm_kind = "none"
    enter("collective set-up (%s)" % args.comm)
    if world > 1:
        comm_kind = args.comm
        if comm_kind == "ipc":
            # peer-mailbox transport: exchange the mailbox handles over the control plane, map, and let every rank run
            # the library's self-test; the job uses it only if ALL ranks pass (else RCCL, then the host-staged one)
            ok = 1
            try:
                mine = eng.comm_ipc_handle(world)
            except ra.RelearnError as exc:
                print("bench.py: rank %d: no mailbox (%s)" % (rank, exc), file=sys.stderr)
                mine, ok = None, 0
            handles = ctl.all_gather(mine)
            if ok and all(h is not None for h in handles):
                try:
                    eng.comm_init_ipc(rank, world, handles)
                    eng.comm_selftest()
                except ra.RelearnError as exc:
                    print("bench.py: rank %d: peer-mailbox collective unusable (%s)" % (rank, exc), file=sys.stderr)
                    ok = 0
            else:
                ok = 0
            if ctl.all_min(ok) == 0:
                ctl.barrier()
                eng.comm_destroy()
                comm_kind = "rccl"
                if rank == 0:
                    print("bench.py: peer-mailbox collective not available on every rank, using RCCL", file=sys.stderr)
        if comm_kind == "rccl":
            # single-node job: RCCL's bootstrap sockets may use the loopback interface (the box may have no other)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            # 1. every rank checks that it can bind RCCL at all, and the job agrees on it BEFORE anyone enters the
            #    collective ncclCommInitRank (a rank that failed here would leave its peers blocked in there)
            ok = ctl.all_min(1 if ra.comm_available() else 0)
            uid = None
            if ok and rank == 0:
                try:
                    uid = ra.comm_unique_id()
                except ra.RelearnError as exc:
                    print("bench.py: RCCL unavailable (%s)" % exc, file=sys.stderr)
            uid = ctl.broadcast(uid, src=0)
            if uid is None:
                ok = 0
            if ok:
                # 2. the collective initialisation under a watchdog: a rank stuck in RCCL's bootstrap ends the job with
                #    a non-zero exit (never a re-exec: this process has touched the GPU)
                import threading
                watchdog = threading.Timer(float(os.environ.get("RELEARN_COMM_INIT_TIMEOUT", "180")),
                                           lambda: (print("bench.py: rank %d: ncclCommInitRank did not return" % rank,
                                                          file=sys.stderr, flush=True), os._exit(3)))
                watchdog.daemon = True
                watchdog.start()
                try:
                    eng.comm_init(rank, world, uid)
                except ra.RelearnError as exc:
                    print("bench.py: rank %d: RCCL communicator not created (%s)" % (rank, exc), file=sys.stderr)
                    ok = 0
                finally:
                    watchdog.cancel()
            if ctl.all_min(ok) == 0:  # every rank takes the same path
                eng.comm_destroy()
                comm_kind = "host"
                if rank == 0:
                    print("bench.py: falling back to the host-staged collective over the control plane", file=sys.stderr)
        if comm_kind == "host":
            eng.comm_init_host(rank, world, ctl.all_reduce_sum_f32)

    enter("allocation")
    env = ra.CartPoleEnv(eng, n_local, max_steps=args.max_episode_steps, limit=ra.LIMIT_VISIBLE,
                         lane_offset=rank * n_local, seed_env=0, seed_actor=1)
    policy = ra.Mlp(eng, 5, H, 2)
    critic = ra.Mlp(eng, 5, H, 1)
    policy.init(2)   # every rank initialises identical replicas from the same stream
    critic.init(3)
    opt = ra.Adam(critic)
    # a pair of trajectories: period k + 1's rollout fills one while period k's critic chain still reads the other
    pipelined = args.pipeline and not args.serial_update
    trajs = [ra.Trajectory(eng, n_local, T, 5) for _ in range(2 if pipelined else 1)]
    trpo_cfg = ra.trpo_config_default()
    gamma = min(0.99, 0.99)  # min(max_discount_factor, env discount) (critics/opt.rs:73)
    critic_cfg = ra.values_opt_config_default()  # ValuesOptConfig::default: reward-to-go targets
    critic_cfg.opt_steps_per_update, critic_cfg.discount_factor = args.critic_steps, gamma
    eng.set_serial_update(args.serial_update)

    last = {}

    def periods(count, before_period=None):
        """`count` whole periods: exactly `count` rollouts, advantage passes, TRPO updates and critic updates, nothing in
        flight before or after.  Period k + 1's rollout needs only the policy TRPO k produced (the actors of the next
        collection snapshot the policy: agents/mod.rs:48-59), so it is enqueued as soon as that chain has finished —
        under critic chain k, which still reads the other trajectory of the pair; rl_gae of k + 1 then waits for chain k
        on the device.  The first rollout of a call has nothing to hide under."""
        pending = None
        for k in range(count):
            if before_period is not None:
                before_period(k)
            traj = trajs[k % len(trajs)]
            ra.rollout(env, policy, traj)
            ra.gae(traj, critic, gamma, 0.95)
            if pending is not None:
                last["critic"] = ra.actor_critic_update_finish(pending)
            # policy.update and critic.update of ActorCriticAgent::batch_update_slice (actor_critic.rs:196-208):
            # independent given the trajectory and its advantages, so the engine runs the two launch chains side by
            # side on two streams and hands control back when the policy chain is done
            last["trpo"] = ra.actor_critic_update_begin(policy, critic, opt, traj, trpo_cfg, critic_cfg)
            pending = traj
            if not pipelined:
                last["critic"] = ra.actor_critic_update_finish(pending)
                pending = None
        if pending is not None:
            last["critic"] = ra.actor_critic_update_finish(pending)

    def barrier():
        # eng.sync() = hipStreamSynchronize(engine stream) + hipDeviceSynchronize(): the device-wide wait the bench
        # contract spells torch.cuda.synchronize(), from the one HIP runtime this process has
        eng.sync()
        if ctl is not None:
            ctl.barrier()
        eng.sync()

    enter("warm-up periods (%s collective)" % comm_kind)
    if os.environ.get("RELEARN_BENCH_TEST_STALL_RANK") == str(rank) and world > 1:
        # test hook (tests/test_gpu_multirank.py): this rank never joins the job's first collective — what a rank that
        # lost its GPU or its link looks like to the others; every rank's watchdog must end the job and name the phase
        time.sleep(3600)
    periods(args.warmup)
    barrier()
    # Per-kernel HIP events (two per launch) cost ~2 ms of host time per period when they wrap all ~220 launches, so
    # they are switched on for the LAST `--profile-steps` periods of the timed region only; `roofline` / `phases`
    # are averages over those periods' launches.
    prof_steps = 0 if args.no_kernel_profile else max(1, min(args.profile_steps, args.steps))
    eng.profile_read(reset=True)
    barrier()
    enter("timed periods (%s collective)" % comm_kind)
    t0 = time.perf_counter()
    def switch_to_profiling(k):
        if prof_steps and k == args.steps - prof_steps:
            # the profiled period(s) run the two chains one after the other: a launch timed by events on its stream while
            # another stream's kernels share the CUs would not be that kernel's duration (`roofline`, `phases`)
            eng.profile_enable(True)
            eng.set_serial_update(True)

    periods(args.steps, switch_to_profiling)
    barrier()
    elapsed = time.perf_counter() - t0
    enter("after the timed region")
    prof = eng.profile_read(reset=True) if not args.no_kernel_profile else None
    eng.profile_enable(False)
    eng.set_serial_update(args.serial_update)
    if ctl is not None:
        elapsed = float(ctl.all_max(elapsed))

    total_steps = args.envs * T * args.steps
    value = total_steps / elapsed
    B_local = n_local * T

    # ---- roofline of the dominant kernel: k_critic_step_mfma (class "critic_fused") -------------------------
    # One launch = forward + MSE loss + backward of the 5-128-1 critic over every sample of the rank.
    # The kernel's contractions run on the bf16 matrix pipe as EXACT three-piece splits (relearn_amd/csrc/bf16_tile.hpp:
    # every product exact, f32 accumulation — f32-equivalent arithmetic), so the roof it sits under is the dense bf16 MFMA
    # peak.  `achieved` / `frac` count the ALGORITHMIC flop (3 x critic forward = 4,608 per sample); what the
    # pipe EXECUTES for it — 22 v_mfma_f32_32x32x16_bf16 per 32-sample tile (12 forward, 2 routing the pieces of dy * x, 8
    # backward) x 32,768 flop = 22,528 per sample — is reported beside it as `pipe_occupancy`.
    roofline = None
    roofline_policy = None
    phases = None

    def counters(prefix):
        """counter evidence of a kernel from the committed PMC summary (separate rocprofv3 --pmc passes over
        scripts/path_once.py, scripts/pmc_passes.sh), with the build it was taken from"""
        tpath = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        out = {"traffic": None, "valu_busy_frac": None, "mfma_busy_frac": None, "valu_issue_frac": None,
               "issue_port_frac": None,
               "source": {"file": "profiles/" + PMC_SUMMARY, "kind": "committed rocprofv3 PMC summary, not measured in this run",
                          "applies": False}}
        if not (os.path.exists(tpath) and args.envs // world == 65536 and T == 128):
            out["source"]["why_not"] = "no summary for this lane count / horizon"
            return out
        summary = json.load(open(tpath))
        out["source"]["collected_from_lib_sha16"] = summary.get("lib_sha16")
        out["source"]["this_lib_sha16"] = lib_sha16()
        out["source"]["applies"] = summary.get("lib_sha16") == lib_sha16()
        for kname, row in summary.get("kernels", {}).items():
            if kname.replace("void ", "").startswith(prefix):
                out["traffic"] = row.get("hbm_bytes_per_launch")
                out["valu_busy_frac"], out["mfma_busy_frac"] = row.get("valu_busy_frac"), row.get("mfma_busy_frac")
                if row.get("SQ_INSTS_VALU") and row.get("SQ_BUSY_CU_CYCLES"):
                    # vector instructions x 4 issue cycles against the SIMD cycles of the launch (4 SIMDs per busy CU cycle)
                    out["valu_issue_frac"] = 4.0 * row["SQ_INSTS_VALU"] / (4.0 * row["SQ_BUSY_CU_CYCLES"])
                    # the SIMD's shared issue port: a vector instruction holds it ~4 cycles, a matrix instruction 8, an
                    # LDS instruction ~18 (profiles/r03_slot_cost.txt), against 4 SIMDs x the busy CU cycles of the launch
                    out["issue_port_frac"] = (4.0 * row["SQ_INSTS_VALU"] + 8.0 * row.get("SQ_INSTS_MFMA", 0.0) +
                                              18.0 * row.get("SQ_INSTS_LDS", 0.0)) / (4.0 * row["SQ_BUSY_CU_CYCLES"])
        return out

    if prof is not None:
        flop_c = 3 * 2 * (5 * H + H * 1)
        cf_ms, cf_n = prof["critic_fused"]
        fused = cf_n > 0
        if not fused:  # v1 kernels (hidden != 128): the separate forward + backward pair on the vector unit
            cf_ms = prof["critic_fwd"][0] + prof["backward"][0]
            cf_n = prof["critic_fwd"][1]
        algorithmic = flop_c * B_local * cf_n / (cf_ms * 1e-3) / 1e12 if cf_ms > 0 else 0.0
        bf16_flop = 22 * 32768 // 32
        executed = bf16_flop * B_local * cf_n / (cf_ms * 1e-3) / 1e12 if cf_ms > 0 else 0.0
        ctr = counters("k_critic_step_mfma<1>")
        if fused:
            roofline = {
                # achieved = ALGORITHMIC flop (3 x critic forward = 4,608 per sample) x samples / launch time,
                # against the peak of the unit that runs it (the dense bf16 matrix pipe)
                "kernel": "k_critic_step_mfma", "bound": "mfma", "achieved": algorithmic, "peak": BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": algorithmic / BF16_PEAK_TFLOPS, "traffic": ctr["traffic"],
                "frac_algorithmic": algorithmic / BF16_PEAK_TFLOPS,
                "algorithmic_flop_per_sample": flop_c,
                "frac_of_f32_mfma_peak": algorithmic / F32_PEAK_TFLOPS,
                # what the pipe EXECUTES for that (exact three-piece splits, K padding: 22,528 flop per sample)
                "pipe_occupancy": executed / BF16_PEAK_TFLOPS, "executed_TFLOPs": executed,
                "executed_bf16_flop_per_sample": bf16_flop, "useful_over_executed": flop_c / float(bf16_flop),
                "issue_port_frac": ctr["issue_port_frac"],
                "valu_issue_frac": ctr["valu_issue_frac"], "valu_busy_frac": ctr["valu_busy_frac"],
                "mfma_busy_frac": ctr["mfma_busy_frac"], "source": ctr["source"],
                "launches": int(cf_n), "avg_launch_us": 1e3 * cf_ms / max(cf_n, 1), "samples_per_launch": B_local,
                "note": "frac = algorithmic f32 flop (SURVEY 8d) / launch time / dense bf16 MFMA peak.  The contractions run "
                        "on the bf16 pipe as exact three-piece splits with f32 accumulation (f32-equivalent arithmetic: "
                        "parity at 1e-6), which executes 4.9x the algorithmic flop: pipe_occupancy = executed / peak, "
                        "frac = pipe_occupancy x useful_over_executed.  The same work is 1.4-1.5x the f32 MFMA roof "
                        "(frac_of_f32_mfma_peak), which is why that roof is not the denominator.  What stops the pipe "
                        "is not HBM (traffic = bytes per launch, 1.0x the algorithmic 24 B per sample) but the SIMD's "
                        "shared issue port: issue_port_frac = (4 x vector + 8 x matrix + 18 x LDS instructions) / SIMD "
                        "cycles, from the committed counters (see source).  Per-rank figures.",
            }
        else:
            roofline = {
                "kernel": "k_critic_fwd + k_mlp_backward (v1, vector unit)", "bound": "valu", "achieved": algorithmic,
                "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": algorithmic / F32_PEAK_TFLOPS, "traffic": None,
                "launches": int(cf_n), "avg_launch_us": 1e3 * cf_ms / max(cf_n, 1), "samples_per_launch": B_local,
                "algorithmic_flop_per_sample": flop_c,
                "note": "hidden != 128: the first-generation f32 kernels; achieved = algorithmic f32 flop against the f32 "
                        "vector / MFMA peak",
            }
        tot = sum(v[0] for v in prof.values())
        phases = {k: {"ms_per_step": v[0] / prof_steps, "launches_per_step": v[1] / prof_steps,
                      "share": (v[0] / tot if tot > 0 else 0.0)} for k, v in prof.items() if v[1]}
        fv_ms, fv_n = prof.get("policy_fvp", (0.0, 0))
        if fv_n:
            # Fisher-vector product launch of the fused policy kernel.  Algorithmic: forward + tangent forward + 2x
            # backward of the 5-128-2 MLP = 4 x 2 x (5*128 + 128*2) = 7,168 f32 flop per sample.  Executed on the bf16
            # pipe: 38 matrix instructions per 32-sample tile (12 forward, 8 transposing the relu' masks, 8 for the masked
            # sum that gives the tangent logit, 2 routing pieces, 8 backward) x 32,768 flop = 38,912 per sample.
            flop_p = 4 * 2 * (5 * H + H * 2)
            ach = flop_p * B_local * fv_n / (fv_ms * 1e-3) / 1e12
            exe = (38 * 32768 // 32) * B_local * fv_n / (fv_ms * 1e-3) / 1e12
            pc = counters("k_policy_bf16<2")
            roofline_policy = {"kernel": "k_policy_bf16<PASS_JVP>", "bound": "mfma", "achieved": ach,
                               "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_PEAK_TFLOPS,
                               "frac_algorithmic": ach / BF16_PEAK_TFLOPS,
                               "frac_of_f32_mfma_peak": ach / F32_PEAK_TFLOPS,
                               "pipe_occupancy": exe / BF16_PEAK_TFLOPS, "executed_TFLOPs": exe,
                               "executed_bf16_flop_per_sample": 38 * 32768 // 32,
                               "traffic": pc["traffic"], "issue_port_frac": pc["issue_port_frac"],
                               "valu_issue_frac": pc["valu_issue_frac"], "valu_busy_frac": pc["valu_busy_frac"],
                               "mfma_busy_frac": pc["mfma_busy_frac"], "source": pc["source"],
                               "launches": int(fv_n), "avg_launch_us": 1e3 * fv_ms / fv_n,
                               "algorithmic_flop_per_sample": flop_p, "samples_per_launch": B_local}
        pf_ms, pf_n = prof["policy_fused"]
        if pf_n:
            # gradient / Fisher-vector launches do forward + 2x backward (+ tangent forward), evaluations forward only
            phases["policy_fused"]["avg_launch_us"] = 1e3 * pf_ms / pf_n
        # fused rollout: 26 B/env-step trajectory record; policy forward 1792 flop/env-step
        ro_ms, ro_n = prof["rollout"]
        if ro_n:
            gbs = 26.0 * B_local * ro_n / (ro_ms * 1e-3) / 1e9
            phases["rollout"]["hbm_GBps_at_26B_per_step"] = gbs
            phases["rollout"]["hbm_frac"] = gbs / HBM_PEAK_GBS
            phases["rollout"]["policy_forward_TFLOPs"] = 1792.0 * B_local * ro_n / (ro_ms * 1e-3) / 1e12

    # ---- standalone env-step kernel: 100 B/env-step, HBM roofline --------------------------
    env_step = None
    if rank == 0:
        reps = 200
        env.upload_actions(np.random.default_rng(0).integers(0, 2, size=n_local).astype(np.uint8))
        for _ in range(20):
            env.step_resident()
        eng.sync()
        eng.timer_begin()
        for _ in range(reps):
            env.step_resident()
        ms = eng.timer_end()
        gbs = 100.0 * n_local * reps / (ms * 1e-3) / 1e9
        env_step = {"kernel": "k_env_step", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "avg_launch_us": 1e3 * ms / reps,
                    "env_steps_per_s": n_local * reps / (ms * 1e-3),
                    "note": "at the workload's lane count the launch is latency-bound (one wave per SIMD); "
                            "`filled` is the same kernel on 4,194,304 lanes"}
        try:
            big_n = 1 << 22
            big = ra.CartPoleEnv(eng, big_n, max_steps=args.max_episode_steps, seed_env=0, seed_actor=1)
            big.upload_actions(np.random.default_rng(0).integers(0, 2, size=big_n).astype(np.uint8))
            for _ in range(5):
                big.step_resident()
            eng.sync()
            eng.timer_begin()
            for _ in range(50):
                big.step_resident()
            bms = eng.timer_end() / 50
            bg = 100.0 * big_n / (bms * 1e-3) / 1e9
            env_step["filled"] = {"lanes": big_n, "achieved": bg, "frac": bg / HBM_PEAK_GBS, "unit": "GB/s",
                                  "avg_launch_us": 1e3 * bms, "env_steps_per_s": big_n / (bms * 1e-3)}
            big.close()
        except ra.RelearnError as exc:  # reported, never fatal for the bench line
            env_step["filled"] = {"error": str(exc)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    # replicas must still be bit-identical after the timed region: every rank applied the same all-reduced vectors
    # (whatever the transport).  A rank whose parameters differ invalidates the run.
    replicas_identical = None
    if ctl is not None:
        import hashlib
        mine = hashlib.sha256(policy.get_params().tobytes() + critic.get_params().tobytes()).hexdigest()
        digests = ctl.all_gather(mine)
        replicas_identical = all(d == digests[0] for d in digests)
        if not replicas_identical:
            print("bench.py: rank %d: replicas diverged under the %s collective: %s" % (rank, comm_kind, digests),
                  file=sys.stderr, flush=True)
        ctl.barrier()
    # what the collective cost each rank (launches and time per period), so that a scaling run explains itself
    allreduce_per_rank = None
    if ctl is not None:
        mine = None
        if prof is not None and prof.get("allreduce", (0.0, 0))[1]:
            ar_ms, ar_n = prof["allreduce"]
            mine = {"rank": rank, "launches_per_step": ar_n / prof_steps, "us_per_launch": 1e3 * ar_ms / ar_n,
                    "ms_per_step": ar_ms / prof_steps}
        allreduce_per_rank = ctl.all_gather(mine)
    if rank == 0:
        st = last["trpo"]
        out = {
            "metric": "env-steps/sec (whole node), 64k-env CartPole TRPO",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "CartPole(VisibleStepLimit %d) MLP-TRPO, %d envs total, T=%d, policy 5-%d-2, critic "
                            "5-%d-1, CG 10, <=15 backtracks, %d Adam critic steps per period" % (
                                args.max_episode_steps, args.envs, T, H, H, args.critic_steps),
                "n_envs_total": args.envs, "n_envs_per_gpu": n_local, "horizon": T, "hidden": H,
                "critic_steps": args.critic_steps,
                "update_chains": ("policy and critic chains one after the other" if args.serial_update else
                                  "policy and critic chains side by side on two streams (the %d profiled period(s) of "
                                  "the timed region run them in turn)" % prof_steps),
                "pipeline": ("period k+1's rollout enqueued after TRPO k, under critic chain k (two trajectories; %d of "
                             "the %d timed rollouts have a chain to hide under)" % (
                                 max(args.steps - prof_steps - 1, 0), args.steps) if pipelined else "none"),
                "parallelism": "env-sharded x%d + %s" % (world, {"none": "no collective (one rank)", "rccl": "RCCL all-reduce",
                                                               "ipc": "single-launch all-reduce over peer-mapped mailboxes",
                                                               "host": "host-staged all-reduce over the %s control plane "
                                                                       "(fallback)" % args.control}[comm_kind]),
            },
            "roofline": roofline,
            "roofline_policy": roofline_policy,
            "roofline_env_step": env_step,
            "cpu_baseline": cpu,
            "phases": phases,
            "replicas_identical": replicas_identical,
            "allreduce_per_rank": allreduce_per_rank,
            "build": build_record(),
            "last_update": {"trpo_status": st.status, "num_backtracks": st.num_backtracks,
                            "kl": st.constraint_val_final, "entropy": st.entropy,
                            "critic_loss_first": last["critic"].loss_first,
                            "critic_loss_last": last["critic"].loss_last},
        }
        if replicas_identical is False:  # an invalid run has no headline number
            out["value"] = None
            out["invalid"] = "the ranks' parameter replicas differ after the timed region"
        print(json.dumps(out))
    if watchdog_all is not None:
        watchdog_all.cancel()
    if ctl is not None:
        try:
            eng.comm_destroy()  # every rank releases its communicator before the control plane goes away
        except ra.RelearnError as exc:
            print("bench.py: rank %d: comm_destroy: %s" % (rank, exc), file=sys.stderr)
        ctl.barrier()
        ctl.close()
    if replicas_identical is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
