"""The oracle's MemoryGame (oracle/envs.c, oracle/seq.c lanes) against an independent pure-Python restatement of
src/envs/memory.rs:24-115 and of rand 0.8.5's `gen_range(0..n)` on the raw ChaCha8 stream.  The reference's only
test of this env is structural (memory.rs:122-125 via envs/testing.rs:23-57: observations and rewards stay inside
their spaces for 1000 random steps), repeated here as invariants."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

L = O.lib()


class Stream:
    """the raw ChaCha8 stream `stream` of Prng::seed_from_u64(seed)"""

    def __init__(self, seed, stream):
        self.r = O.Prng()
        L.oracle_prng_seed_from_u64(C.byref(self.r), seed)
        L.oracle_prng_set_stream(C.byref(self.r), stream)

    def next_u64(self):
        return int(L.oracle_prng_next_u64(C.byref(self.r)))


def gen_range_replay(rng, n):
    """UniformInt::<u64>::sample_single (rand 0.8.5): widening multiply with a zone; returns (value, u64s consumed)"""
    lz = 64 - int(n).bit_length()
    zone = ((n << lz) - 1) & (2 ** 64 - 1)
    used = 0
    while True:
        v = rng.next_u64()
        used += 1
        m = v * n
        if (m & (2 ** 64 - 1)) <= zone:
            return m >> 64, used


@pytest.mark.parametrize("num_actions,history_len", [(2, 3), (2, 1), (3, 2), (5, 4)])
def test_lanes_follow_memory_rs(num_actions, history_len):
    n, lane_offset = 40, 9
    sim = O.MemoryLaneSim(n, num_actions, history_len, lane_offset=lane_offset, seed_env=7)
    S = num_actions + history_len
    assert sim.D == S
    # independent model: per lane a Prng on stream lane_offset + i, consumed only by initial_state
    rngs = []
    for i in range(n):
        rngs.append(Stream(7, lane_offset + i))
    cur, ini, pos = np.zeros(n, int), np.zeros(n, int), np.zeros(n, int)
    for i in range(n):
        v, used = gen_range_replay(rngs[i], num_actions)
        cur[i] = ini[i] = v
        pos[i] += 2 * used
    rs = np.random.default_rng(0)
    ep_len = np.zeros(n, int)
    for t in range(6 * (history_len + 1) + 3):
        c, i0, p, _, rc = sim.get_state()
        assert np.array_equal(c, cur) and np.array_equal(i0, ini) and np.array_equal(p, pos)
        obs = sim.observe()
        assert np.array_equal(obs, (np.arange(S)[:, None] == cur[None, :]).astype(np.float32))  # one-hot
        a = rs.integers(0, num_actions, n).astype(np.uint8)
        reward, flag, obs_next, _ = sim.step(a)
        for i in range(n):
            ep_len[i] += 1
            if cur[i] == S - 1:  # the answer step: terminal, +-1
                assert flag[i] == O.TERMINATE and reward[i] == (1.0 if a[i] == ini[i] else -1.0)
                assert ep_len[i] == history_len + 1  # "Every episode has length HISTORY_LEN + 1"
                ep_len[i] = 0
                v, used = gen_range_replay(rngs[i], num_actions)
                cur[i] = ini[i] = v
                pos[i] += 2 * used
            else:
                assert flag[i] == O.CONTINUE and reward[i] == 0.0
                cur[i] = num_actions if cur[i] < num_actions else cur[i] + 1
        assert np.array_equal(obs_next, (np.arange(S)[:, None] == cur[None, :]).astype(np.float32))
    assert set(np.unique(ini)) == set(range(num_actions)) or n < 4 * num_actions


def test_rejected_draws_advance_the_stream():
    # gen_range(0..2) rejects every u64 whose bit 62 is set: about half of the lanes need more than one draw
    sim = O.MemoryLaneSim(2000, 2, 3, seed_env=1)
    _, ini, pos, _, rc = sim.get_state()
    assert np.all(pos % 2 == 0) and np.all(pos >= 2) and np.all(rc == 1)
    frac_more = (pos > 2).mean()
    assert 0.45 < frac_more < 0.55
    assert 0.45 < ini.mean() < 0.55


def test_step_limit_interrupts_before_the_answer():
    # LatentStepLimit(2) cuts MemoryGame(2, 3) (4-step episodes) after two steps: Interrupt with the successor state
    sim = O.MemoryLaneSim(16, 2, 3, max_steps=2, limit=O.LIMIT_LATENT, seed_env=5)
    a = np.zeros(16, np.uint8)
    r, f, _, _ = sim.step(a)
    assert np.all(f == O.CONTINUE) and np.all(r == 0)
    r, f, obs, term = sim.step(a)
    assert np.all(f == O.INTERRUPT) and np.all(r == 0)
    assert np.all(term[3] == 1.0)  # states 0/1 -> 2 -> 3: the successor of the cut episode is state 3
    assert np.all(obs[:2].sum(axis=0) == 1.0)  # and the lane restarted in state 0 or 1
    # with the limit equal to the episode length the Terminate of the answer step passes through
    sim = O.MemoryLaneSim(16, 2, 3, max_steps=4, limit=O.LIMIT_LATENT, seed_env=5)
    for t in range(4):
        r, f, _, _ = sim.step(a)
    assert np.all(f == O.TERMINATE) and set(np.unique(r)) <= {-1.0, 1.0}


@pytest.mark.parametrize("num_actions,history_len,max_steps", [(2, 3, 0), (2, 1, 0), (4, 2, 0), (2, 3, 3)])
def test_scalar_host_env_matches_the_oracle(num_actions, history_len, max_steps):
    """relearn_amd/csrc/host/envs.hpp MemoryGame (the scalar Environment of the C++ host API, any size) against lane 0
    of the oracle: same observations, rewards and successors for the same actions and env seed."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tempfile.mkdtemp(), "host_envs_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", root, "-I",
                           os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "host_envs_demo.cpp"),
                           "-o", exe])
    steps = 45
    lines = subprocess.check_output([exe, str(num_actions), str(history_len), str(max_steps), "11", str(steps)])
    rows = [l.split() for l in lines.decode().splitlines()]
    sim = O.MemoryLaneSim(1, num_actions, history_len, max_steps=max_steps,
                          limit=O.LIMIT_LATENT if max_steps else O.LIMIT_NONE, seed_env=11)
    for t, (obs, action, reward, succ) in enumerate(rows):
        assert int(obs) == int(np.argmax(sim.observe()[:, 0]))
        assert int(action) == (t * 7 + 3) % num_actions
        r, f, _, _ = sim.step(np.array([int(action)], np.uint8))
        assert float(reward) == float(r[0]) and int(succ) == int(f[0]), t
    assert len(rows) == steps
