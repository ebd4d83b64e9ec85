"""The C++ host API's statistics logging (relearn_amd/csrc/host/logging.hpp) against a Python restatement of the
reference's chunked logger (src/logging/chunk.rs:40-266, chunk_by_counter.rs, display.rs, tensorboard.rs): which
values land in which chunk, counters carrying their running total, Welford mean / population sigma, the index
histogram, incompatible values refused, ids in component order, and a TensorBoard event file a reader can parse.
Host-only code: no GPU and no device library needed."""
import json
import math
import os
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "logging_demo.cpp")


@pytest.fixture(scope="module")
def demo():
    out = os.path.join(tempfile.mkdtemp(), "logging_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", ROOT, SRC, "-o", out])
    return out


class Welford:  # utils/stats.rs:119-127
    def __init__(self):
        self.mean, self.s, self.n = 0.0, 0.0, 0

    def push(self, v):
        pre = v - self.mean
        self.n += 1
        self.mean = self.mean + pre / self.n
        self.s = self.s + pre * (v - self.mean)


def id_key(i):
    return i.split("/")


class ChunkSim:
    """chunk.rs ChunkLogger + chunk_by_counter.rs ByCounter."""

    def __init__(self, counter, interval):
        self.counter, self.interval = counter, interval
        self.nodes, self.chunks, self.flush_pending, self.depth = {}, [], False, 0

    def group_start(self):
        self.depth += 1

    def group_end(self):
        self.depth -= 1
        if self.depth == 0 and self.flush_pending:
            self.flush()

    def log(self, i, kind, v):
        own = self.depth == 0
        if own:
            self.group_start()
        node = self.nodes.get(i)
        if node is None:
            node = self.nodes[i] = {"kind": kind, "dirty": True, "increment": 0, "initial": 0, "w": Welford(),
                                    "counts": [0] * (v[1] if kind == "Index" else 0)}
        assert node["kind"] == kind
        node["dirty"] = True
        if kind == "CounterIncrement":
            node["increment"] += v
        elif kind in ("Scalar", "Duration"):
            node["w"].push(v)
        else:
            node["counts"][v[0]] += 1
        if i == self.counter and not self.flush_pending:
            self.flush_pending = (node["increment"] + node["initial"]) % self.interval == 0
        if own:
            self.group_end()

    def flush(self):
        items = []
        for i in sorted(self.nodes, key=id_key):
            n = self.nodes[i]
            if not n["dirty"]:
                continue
            it = {"id": i, "kind": n["kind"]}
            if n["kind"] == "CounterIncrement":
                it.update(increment=n["increment"], initial_value=n["initial"])
            elif n["kind"] == "Index":
                it.update(counts=list(n["counts"]))
            else:
                w = n["w"]
                it.update(count=w.n, mean=w.mean, stddev=math.sqrt(w.s / w.n))
            items.append(it)
        self.chunks.append(items)
        for n in self.nodes.values():
            n["dirty"] = False
            n["initial"] += n["increment"]
            n["increment"] = 0
            n["w"] = Welford()
            n["counts"] = [0] * len(n["counts"])
        self.flush_pending = False


def test_chunks_match_the_restatement(demo):
    lines = [json.loads(l) for l in subprocess.check_output([demo, "chunks"]).decode().splitlines()]
    sim = ChunkSim("agent_update/count", 2)
    x = 0.5
    for period in range(5):
        sim.group_start()
        sim.log("agent_update/count", "CounterIncrement", 1)
        sim.log("agent_update/time", "Duration", 0.001 * (period + 1))
        sim.group_end()
        for k in range(period + 1):
            x = x * 1.7 - 0.3 * k
            sim.log("policy/entropy", "Scalar", x)
        sim.log("worker0/step/action", "Index", (period % 3, 3))
        if period == 3:
            sim.log("sim/step/count", "CounterIncrement", 7)
    sim.flush()  # the logger flushes when destroyed
    chunks = [l for l in lines if "chunk" in l]
    errors = [l["error"] for l in lines if "error" in l]
    assert [c["chunk"] for c in chunks] == list(range(len(sim.chunks)))
    assert len(chunks) == 3  # after updates 2 and 4 (at the END of their group), then the final flush
    for got, want in zip(chunks, sim.chunks):
        assert [g["id"] for g in got["items"]] == [w["id"] for w in want]
        for g, w in zip(got["items"], want):
            assert g == w, (g, w)  # doubles printed with 17 digits: exact
    # the group that brings the counter to 2 also holds that update's duration: both are in chunk 0
    c0 = {i["id"]: i for i in chunks[0]["items"]}
    assert c0["agent_update/time"]["count"] == 2 and c0["agent_update/count"]["increment"] == 2
    c1 = {i["id"]: i for i in chunks[1]["items"]}
    assert c1["agent_update/count"] == {"id": "agent_update/count", "kind": "CounterIncrement", "increment": 2,
                                        "initial_value": 2}
    assert "sim/step/count" not in c0 and "sim/step/count" not in c1  # logged after chunk 1 closed
    assert {i["id"] for i in chunks[2]["items"]} >= {"sim/step/count", "policy/entropy"}
    # incompatible values are refused with the reference's messages (logging/mod.rs LogError)
    assert errors[0] == "incompatible value type; previously CounterIncrement, now Scalar"
    assert errors[1] == "incompatible index size; previously 3, now 4"


def test_by_time_flushes_at_group_start(demo):
    lines = [json.loads(l) for l in subprocess.check_output([demo, "bytime"]).decode().splitlines()]
    # chunk_by_time.rs:32-39: the check happens BEFORE the value is logged, so 1.0 and 3.0 are in different chunks
    assert [[(i["id"], i["count"], i["mean"]) for i in c["items"]] for c in lines if c["items"]] == \
        [[("x", 1, 1.0)], [("x", 1, 3.0)]]


def test_display_lines(demo):
    out = subprocess.check_output([demo, "display"]).decode().splitlines()
    assert out[0] == ""
    rows = {l[:24].rstrip(): l[25:] for l in out[1:]}
    assert [l[:24].rstrip() for l in out[1:]] == ["a", "a/action", "b", "huge", "policy/entropy", "tiny", "z/time"]
    assert rows["a"] == "1.000"
    assert rows["a/action"] == "(n 3)  [33 66]%"          # integer percentages, display.rs:147-160
    assert rows["b"] == "3  (+3)"                          # no rate below 6 increments
    assert rows["huge"] == "1.250e7" and rows["tiny"] == "2.500e-5"  # PrettyPrint<f64>: exponent form
    mean = (0.6931 + 0.6) / 2
    sd = math.sqrt(((0.6931 - mean) ** 2 + (0.6 - mean) ** 2) / 2)
    assert rows["policy/entropy"] == "%.3f (σ %.3f)" % (mean, sd)
    assert rows["z/time"].startswith("250.0000ms ") and rows["z/time"].endswith("%")


# ---- a minimal TensorBoard event-file reader (TFRecord framing + the Event / Summary protobuf fields used)
def crc32c(data):
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    c = 0xFFFFFFFF
    for b in data:
        c = table[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def fields(buf):
    out, i = [], 0
    while i < len(buf):
        key = shift = 0
        while True:
            b = buf[i]
            i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if b < 0x80:
                break
        f, wire = key >> 3, key & 7
        if wire == 0:
            v = shift = 0
            while True:
                b = buf[i]
                i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if b < 0x80:
                    break
        elif wire == 1:
            v = buf[i:i + 8]
            i += 8
        elif wire == 5:
            v = buf[i:i + 4]
            i += 4
        else:
            assert wire == 2
            n = shift = 0
            while True:
                b = buf[i]
                i += 1
                n |= (b & 0x7F) << shift
                shift += 7
                if b < 0x80:
                    break
            v = buf[i:i + n]
            i += n
        out.append((f, v))
    return out


def test_tensorboard_event_file(demo):
    d = tempfile.mkdtemp()
    path = subprocess.check_output([demo, "tensorboard", d]).decode().strip()
    assert os.path.dirname(path) == d and os.path.basename(path).startswith("events.out.tfevents.")
    raw = open(path, "rb").read()
    events, i = [], 0
    while i < len(raw):
        (n,) = struct.unpack_from("<Q", raw, i)
        assert struct.unpack_from("<I", raw, i + 8)[0] == masked(raw[i:i + 8])
        data = raw[i + 12:i + 12 + n]
        assert struct.unpack_from("<I", raw, i + 12 + n)[0] == masked(data)
        events.append(dict(fields(data)))
        i += 16 + n
    assert events[0][3] == b"brain.Event:2"
    got = {}
    for ev in events[1:]:
        step = ev.get(2, 0)
        value = dict(fields(dict(fields(ev[5]))[1]))
        tag = value[1].decode()
        if 2 in value:
            got[(tag, step)] = struct.unpack("<f", value[2])[0]
        else:
            h = dict(fields(value[5]))
            dbl = lambda b: list(struct.unpack("<%dd" % (len(b) // 8), b))
            got[(tag, step)] = {k: dbl(h[k]) for k in range(1, 8)}
    for step in range(3):  # ByCounter("n", 1): one chunk per group; tensorboard.rs:86-123
        assert got[("n", step)] == float(step + 1)                       # running total of the counter
        assert got[("policy/entropy", step)] == 0.5 + step               # chunk mean
        assert got[("time", step)] == 0.125 * (step + 1)
        counts = [0.0, 0.0]
        counts[step % 2] += 1
        counts[1] += 1
        h = got[("action", step)]
        assert h[1] == [-0.5] and h[2] == [1.5] and h[3] == [2.0] and h[6] == [0.5, 1.5] and h[7] == counts
        assert h[4] == [counts[1]] and h[5] == [counts[1]]               # sum i*n, sum i*i*n
