"""The multi-rank launch path of bench.py without a GPU: the standard-library control plane
(relearn_amd/rendezvous.py — what replaces the gloo group of earlier rounds and, at process granularity, the scoped
thread fan-out and join of /root/reference/src/simulation/train.rs:98-158,180) and bench.py's own spawner.

`bench.py --rendezvous-only` runs everything up to and including the rendezvous and one of each control-plane collective,
then prints its one JSON line; the GPU half of the same path is tests/test_gpu_multirank.py."""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from relearn_amd import rendezvous  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_body(rank, world, port, token, q, scenario):
    try:
        ctl = rendezvous.Control(rank, world, "127.0.0.1", [port], token, timeout=20.0, connect_timeout=20.0)
        out = {}
        if scenario == "collectives":
            out["gather"] = ctl.all_gather({"r": rank, "blob": bytes([rank]) * 64})
            out["bcast"] = ctl.broadcast("from-two" if rank == 2 else None, src=2)
            ctl.barrier()
            out["min"], out["max"] = ctl.all_min(10 - rank), ctl.all_max(rank * 1.5)
            # sum in rank order in f32: every rank gets the same bits, and they are the bits of ((a0 + a1) + a2) + a3
            a = (np.arange(1000, dtype=np.float32) + 1.0) * np.float32(1.0 + 1e-7 * (rank + 1)) * np.float32(3.0 ** rank)
            mine = a.copy()
            ctl.all_reduce_sum_f32(a)
            out["sum"], out["mine"] = a, mine
            two_d = np.full((3, 4), float(rank), dtype=np.float32)
            ctl.all_reduce_sum_f32(two_d)
            out["two_d"] = two_d
        elif scenario == "diverge":
            try:
                if rank == 1:
                    ctl.all_max(1.0)  # the others are in a barrier
                else:
                    ctl.barrier()
                out["error"] = None
            except rendezvous.ControlError as exc:
                out["error"] = str(exc)
        elif scenario == "death":
            ctl.barrier()
            if rank == 1:
                os._exit(7)  # dies between two collectives
            t0 = time.time()
            try:
                ctl.barrier()
                out["error"] = None
            except rendezvous.ControlError as exc:
                out["error"] = str(exc)
            out["seconds"] = time.time() - t0
        ctl.close()
        q.put((rank, out))
    except BaseException as exc:  # noqa: BLE001
        q.put((rank, {"exception": repr(exc)}))


def run_world(world, scenario, expect=None):
    port, token = free_port(), b"t" * 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_body, args=(r, world, port, token, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(expect if expect is not None else world):
        r, out = q.get(timeout=60)
        res[r] = out
    for p in procs:
        p.join(timeout=30)
    return res


def test_control_plane_collectives_over_four_processes():
    res = run_world(4, "collectives")
    assert all("exception" not in res[r] for r in range(4)), res
    want = res[0]["mine"].copy()
    for r in range(1, 4):
        want = want + res[r]["mine"]  # f32, rank order
    for r in range(4):
        assert [g["r"] for g in res[r]["gather"]] == [0, 1, 2, 3]
        assert res[r]["gather"][3]["blob"] == b"\x03" * 64
        assert res[r]["bcast"] == "from-two"
        assert res[r]["min"] == 7 and res[r]["max"] == 4.5
        assert res[r]["sum"].dtype == np.float32 and np.array_equal(res[r]["sum"], want)
        assert np.array_equal(res[r]["two_d"], np.full((3, 4), 6.0, dtype=np.float32))


def test_ranks_that_take_different_paths_get_an_error_not_a_hang():
    res = run_world(3, "diverge")
    for r in range(3):
        assert "exception" not in res[r], res
        assert res[r]["error"] and "while rank 0 is in" in res[r]["error"], res


def test_a_rank_that_dies_is_seen_at_once_by_the_others():
    res = run_world(3, "death", expect=2)
    for r in (0, 2):
        assert res[r]["error"] and res[r]["seconds"] < 10.0, res


def test_foreign_connections_are_turned_away():
    """something else that connects to rank 0's port (another job's rank with another token, a scanner) is answered NO /
    dropped and the job still assembles"""
    port, token = free_port(), b"j" * 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_rank_body, args=(0, 2, port, token, q, "none"))
    p0.start()
    deadline = time.time() + 20
    answered = None
    while time.time() < deadline and answered is None:
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=2) as s:
                import struct
                s.sendall(rendezvous.MAGIC + b"x" * 16 + struct.pack(">II", 1, 2))
                answered = s.recv(2)
        except OSError:
            time.sleep(0.1)
    assert answered == b"NO"
    with socket.create_connection(("127.0.0.1", port), timeout=2) as s:
        s.sendall(b"GET / HTTP/1.0\r\n\r\n")
    p1 = ctx.Process(target=_rank_body, args=(1, 2, port, token, q, "none"))
    p1.start()
    got = dict(q.get(timeout=40) for _ in range(2))
    p0.join(timeout=20)
    p1.join(timeout=20)
    assert got == {0: {}, 1: {}}


def test_port_search_under_a_launcher_that_holds_master_port():
    """under torch.distributed.run MASTER_PORT itself is the launcher's store: rank 0 takes the first free port above it,
    the others find it by the handshake (here the first candidate is occupied as well)"""
    base = free_port()
    holder = socket.socket()
    try:
        holder.bind(("127.0.0.1", base))
        holder.listen(1)
    except OSError:
        pytest.skip("port vanished")
    blocker = socket.socket()
    try:
        blocker.bind(("127.0.0.1", base + 1))
        blocker.listen(1)
    except OSError:
        blocker = None
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base))
    env.pop("RELEARN_RDZV_PORT", None)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from relearn_amd import rendezvous\n"
            "c = rendezvous.from_env(timeout=20)\n"
            "print(c.port, c.all_gather(c.rank)); c.barrier(); c.close()\n" % ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=60) for p in procs]
    holder.close()
    if blocker:
        blocker.close()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-1500:]
    ports = {int(o.decode().split()[0]) for o, _ in outs}
    assert len(ports) == 1 and base < ports.pop() <= base + rendezvous.PORT_SPAN
    assert all(o.decode().strip().endswith("[0, 1]") for o, _ in outs)


# ---------------------------------------------------------------- bench.py's launch path
def run_bench(args, env_extra=None, launcher=None, limit=120):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RELEARN_RDZV_PORT"):
        env.pop(k, None)
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py")] + args
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        proc.communicate()
        raise AssertionError("bench.py %s did not finish within %d s" % (args, limit))
    return proc.returncode, out.decode(), err.decode()


def poisoned_torch(tmp_path):
    """a directory that makes `import torch` fail, to put in front of PYTHONPATH"""
    d = tmp_path / "no_torch" / "torch"
    d.mkdir(parents=True)
    (d / "__init__.py").write_text("raise ImportError('torch is not available in this test')\n")
    return str(tmp_path / "no_torch")


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 4` with no launcher: four children with RANK / LOCAL_RANK / WORLD_SIZE, one JSON line from
    rank 0, exit 0 — also in an environment where torch cannot be imported at all"""
    for extra in ({}, {"PYTHONPATH": poisoned_torch(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", "")}):
        code, out, err = run_bench(["--gpus", "4", "--rendezvous-only"], extra)
        assert code == 0, err[-2000:]
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and len(out.strip().splitlines()) == 1, out
        res = json.loads(lines[0])
        assert res["n_gpus"] == 4 and res["control"] == "tcp" and res["sum_probe"] == 10.0 and res["uid_len"] == 128
        assert [r["rank"] for r in res["ranks"]] == [0, 1, 2, 3] == [r["local_rank"] for r in res["ranks"]]
        assert len({r["pid"] for r in res["ranks"]}) == 4
        assert not any(r["torch_imported"] for r in res["ranks"])  # no rank maps a second HIP runtime


def test_bench_under_torch_distributed_run_uses_the_same_control_plane():
    """the driver's launch line: the ranks come from torch.distributed.run, which holds MASTER_PORT; the ranks themselves
    still do not import torch"""
    port = free_port()
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    code, out, err = run_bench(["--gpus", "2", "--rendezvous-only"], launcher=launcher, limit=280)
    assert code == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["control"] == "tcp" and res["sum_probe"] == 3.0
    assert not any(r["torch_imported"] for r in res["ranks"])


def test_bench_gloo_control_plane_is_still_available():
    code, out, err = run_bench(["--gpus", "2", "--rendezvous-only", "--control", "gloo"], limit=280)
    assert code == 0, err[-2000:]
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert res["control"] == "gloo" and res["sum_probe"] == 3.0 and res["max_rank"] == 1
    assert all(r["torch_imported"] for r in res["ranks"])


def test_a_failing_rank_fails_the_launch_without_a_result_line():
    """no GPU visible to the ranks: every rank's engine creation fails loudly (there is no CPU fallback), the launcher
    reports which rank left first, prints no result line and exits non-zero"""
    hide = {"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""}
    code, out, err = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], hide, limit=200)
    assert code != 0
    assert not [l for l in out.splitlines() if l.startswith("{")]
    assert "left with exit code" in err and "no CPU fallback" in err, err[-2000:]
