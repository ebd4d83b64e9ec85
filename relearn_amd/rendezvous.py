"""Control plane of a multi-rank job, in the standard library only: a TCP star on rank 0.

What a one-process-per-GPU job needs next to the data plane (RCCL / the peer mailboxes, both inside
librelearn_hip.so) is small: hand out the collective's unique id or the mailbox handles, agree on a path, a
barrier on both sides of the timed region, the maximum of one scalar.  This replaces the thread fan-out and
join of the reference's `train_parallel` (/root/reference/src/simulation/train.rs:98-158,180: scoped worker
threads, a join, the summed statistics) at process granularity, and it replaces the gloo control group earlier
rounds took from PyTorch — no rank of the job imports torch, so no rank maps a second HIP runtime.

Topology: rank 0 listens, ranks 1..N-1 connect and identify themselves with the job's token.  Every call is a
collective over all ranks and carries (sequence number, operation name): a rank that took another path shows
up as a `ControlError` on every rank instead of a hang.  Every socket operation is bounded by `timeout`
seconds; a rank that died closes its socket, which the others see as an error at once.

Messages: 8-byte big-endian length + pickle (the peers are this job's own processes on one node).

Finding rank 0: `RELEARN_RDZV_PORT` names the port exactly (set by `bench.py` when it spawns its own ranks).
Under `torch.distributed.run` only MASTER_ADDR / MASTER_PORT are known and MASTER_PORT itself is taken by the
launcher's store, so rank 0 binds the first free port of MASTER_PORT+1 .. +32 and the others try those ports
in turn until one answers the handshake with this job's token.
"""
import hashlib
import os
import pickle
import socket
import struct
import time

import numpy as np

MAGIC = b"RLRDZV01"
PORT_SPAN = 32


class ControlError(RuntimeError):
    pass


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack(">Q", len(data)) + data)


def _recv_exact(sock, n, who):
    buf = bytearray()
    while len(buf) < n:
        try:
            chunk = sock.recv(n - len(buf))
        except socket.timeout:
            raise ControlError("control plane: no message from %s within the time limit" % who)
        except OSError as exc:
            raise ControlError("control plane: connection to %s failed (%s)" % (who, exc))
        if not chunk:
            raise ControlError("control plane: %s closed its connection (the process ended?)" % who)
        buf += chunk
    return bytes(buf)


def _recv(sock, who):
    (n,) = struct.unpack(">Q", _recv_exact(sock, 8, who))
    if n > (1 << 30):
        raise ControlError("control plane: absurd message length %d from %s" % (n, who))
    return pickle.loads(_recv_exact(sock, n, who))


def job_token(world, master_port, extra=""):
    text = "%s|%s|%s|%s|%s" % (world, master_port, os.environ.get("TORCHELASTIC_RUN_ID", ""),
                               os.environ.get("RELEARN_RDZV_TOKEN", ""), extra)
    return hashlib.sha256(text.encode()).digest()[:16]


class Control:
    """One rank's end of the star.  Every public method is a collective: all ranks call it, in the same order."""

    def __init__(self, rank, world, addr="127.0.0.1", ports=(29501,), token=b"", timeout=420.0, connect_timeout=None):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.seq = 0
        self.peers = {}     # rank 0: rank -> socket
        self.hub = None     # other ranks: socket to rank 0
        self.listener = None
        self.port = None
        if not 0 <= self.rank < self.world:
            raise ControlError("control plane: rank %d of %d" % (self.rank, self.world))
        deadline = time.time() + (connect_timeout if connect_timeout is not None else self.timeout)
        if self.world == 1:
            return
        if self.rank == 0:
            self._listen(addr, list(ports), token, deadline)
        else:
            self._connect(addr, list(ports), token, deadline)

    # ------------------------------------------------------------------ set-up
    def _listen(self, addr, ports, token, deadline):
        last = None
        for port in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            except OSError as exc:
                last = exc
                s.close()
                continue
            self.listener, self.port = s, s.getsockname()[1]
            break
        if self.listener is None:
            raise ControlError("control plane: rank 0 could not bind any of the ports %s on %s (%s)" % (ports, addr, last))
        self.listener.listen(self.world + 8)
        try:
            while len(self.peers) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    missing = sorted(set(range(1, self.world)) - set(self.peers))
                    raise ControlError("control plane: ranks %s never reached rank 0 (port %d)" % (missing, self.port))
                self.listener.settimeout(min(left, 5.0))
                try:
                    conn, _ = self.listener.accept()
                except socket.timeout:
                    continue
                conn.settimeout(5.0)
                try:
                    hello = _recv_exact(conn, len(MAGIC) + 16 + 8, "a connecting process")
                    r, w = struct.unpack(">II", hello[len(MAGIC) + 16:])
                    ok = (hello[:len(MAGIC)] == MAGIC and hello[len(MAGIC):len(MAGIC) + 16] == token and w == self.world
                          and 0 < r < self.world and r not in self.peers)
                    conn.sendall(b"OK" if ok else b"NO")
                except (ControlError, OSError, struct.error):
                    ok = False
                if not ok:  # another job's rank, a port scanner, a duplicate: not ours
                    conn.close()
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(self.timeout)
                self.peers[r] = conn
        except BaseException:
            self.close()
            raise

    def _connect(self, addr, ports, token, deadline):
        hello = MAGIC + token + struct.pack(">II", self.rank, self.world)
        last = "never tried"
        while time.time() < deadline:
            for port in ports:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.settimeout(3.0)
                try:
                    s.connect((addr, port))
                    s.sendall(hello)
                    if _recv_exact(s, 2, "rank 0") == b"OK":
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(self.timeout)
                        self.hub, self.port = s, port
                        return
                    last = "port %d answered, but not for this job" % port
                except (OSError, ControlError) as exc:
                    last = "port %d: %s" % (port, exc)
                s.close()
            time.sleep(0.05)
        raise ControlError("control plane: rank %d found no rank 0 on %s ports %s (%s)" % (self.rank, addr, ports, last))

    # ------------------------------------------------------------------ the one primitive
    def _exchange(self, op, payload, combine):
        """every rank sends (seq, op, payload) to rank 0, which applies `combine` to the payloads in rank order and
        returns one result to everyone"""
        self.seq += 1
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            items, bad = [payload], None
            for r in range(1, self.world):
                try:
                    seq, their_op, item = _recv(self.peers[r], "rank %d" % r)
                except ControlError as exc:
                    bad = str(exc)
                    break
                if seq != self.seq or their_op != op:
                    bad = "control plane: rank %d is in `%s` #%d while rank 0 is in `%s` #%d" % (r, their_op, seq, op, self.seq)
                    break
                items.append(item)
            if bad is None:
                try:
                    result = ("ok", combine(items))
                except Exception as exc:  # e.g. arrays of different lengths
                    bad = "control plane: `%s` could not be combined (%s)" % (op, exc)
            if bad is not None:
                result = ("error", bad)
            for r in range(1, self.world):
                try:
                    _send(self.peers[r], result)
                except OSError:
                    pass
            if bad is not None:
                self.close()
                raise ControlError(bad)
            return result[1]
        try:
            _send(self.hub, (self.seq, op, payload))
        except OSError as exc:
            raise ControlError("control plane: rank %d could not reach rank 0 (%s)" % (self.rank, exc))
        status, value = _recv(self.hub, "rank 0")
        if status != "ok":
            self.close()
            raise ControlError(value)
        return value

    # ------------------------------------------------------------------ collectives
    def all_gather(self, obj):
        """the objects of all ranks, in rank order"""
        return self._exchange("all_gather", obj, list)

    def broadcast(self, obj, src=0):
        return self._exchange("broadcast:%d" % src, obj if self.rank == src else None, lambda items: items[src])

    def barrier(self):
        self._exchange("barrier", None, lambda items: None)

    def all_min(self, x):
        return self._exchange("min", x, min)

    def all_max(self, x):
        return self._exchange("max", x, max)

    def all_reduce_sum_f32(self, array):
        """sum a float32 numpy array over all ranks IN PLACE — the terms added in rank order in f32, so every rank holds
        the same bits (the host-staged data plane of rl_comm_init_host: a fallback, never the fast path)"""
        a = np.ascontiguousarray(array, dtype=np.float32)

        def total(items):
            acc = np.frombuffer(items[0], dtype=np.float32).copy()
            for it in items[1:]:
                term = np.frombuffer(it, dtype=np.float32)
                if term.shape != acc.shape:
                    raise ValueError("lengths differ: %d and %d" % (acc.size, term.size))
                acc += term
            return acc.tobytes()

        out = np.frombuffer(self._exchange("sum_f32", a.tobytes(), total), dtype=np.float32)
        array[...] = out.reshape(array.shape)
        return array

    def close(self):
        for s in list(self.peers.values()) + [self.hub, self.listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.hub, self.listener = {}, None, None


def from_env(timeout=None):
    """the control plane of this process's job, from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR,
    MASTER_PORT as torch.distributed.run and bench.py's own spawner set them)"""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    master_port = int(os.environ.get("MASTER_PORT", "29500"))
    if timeout is None:
        # (a little longer than the job's own watchdog, which names the phase a rank is stuck in: that message first)
        timeout = float(os.environ.get("RELEARN_BENCH_TIMEOUT", "420")) + 30.0
    exact = os.environ.get("RELEARN_RDZV_PORT")
    ports = [int(exact)] if exact else [master_port + 1 + i for i in range(PORT_SPAN) if master_port + 1 + i < 65536]
    return Control(rank, world, addr, ports, job_token(world, master_port), timeout=timeout,
                   connect_timeout=float(os.environ.get("RELEARN_RDZV_CONNECT_TIMEOUT", "180")))
