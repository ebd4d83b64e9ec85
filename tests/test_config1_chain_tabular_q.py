"""BASELINE.json configs[0] — examples/chain-tabular-q.rs (CPU plumbing, no GPU): the product's C++ host path
(relearn_amd/csrc/host/*.hpp behind rl_chain_tabular_q_*) against the C oracle.  Bar: bit-exact Q-table, visit
counts and evaluation action stream for a fixed thread count.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
import relearn_amd as ra


@pytest.mark.parametrize("threads,periods,steps", [(1, 3, 500), (4, 10, 10000), (3, 2, 1234)])
def test_q_table_bit_exact_against_oracle(threads, periods, steps):
    q, counts, total = ra.chain_tabular_q_train(seed=0, n_threads=threads, n_periods=periods, min_worker_steps=steps)
    q_o = np.zeros((5, 2), np.float64)
    c_o = np.zeros((5, 2), np.uint64)
    t_o = C.c_uint64()
    O.lib().oracle_chain_tabular_q_train(0, threads, periods, steps, O.f64p(q_o), O.u64p(c_o), C.byref(t_o))
    assert np.array_equal(q, q_o)
    assert np.array_equal(counts, c_o)
    assert total == t_o.value == threads * periods * (steps - 1)  # the dangling last step of each thread is dropped


def test_example_configuration_learns_and_evaluates():
    """10 periods, 10,000 steps per worker (examples/chain-tabular-q.rs:15-20): the greedy policy always moves
    right; evaluation with SimSeed::Root(0) is bit-exact against the oracle."""
    q, counts, total = ra.chain_tabular_q_train(seed=0, n_threads=4, n_periods=10, min_worker_steps=10000)
    assert np.all(q[:, 1] > q[:, 0])
    actions, reward = ra.chain_tabular_q_eval(q, seed=0, n_steps=10000)
    acts_o = np.zeros(10000, np.int32)
    reward_o = O.lib().oracle_chain_tabular_q_eval(O.f64p(q), 0, 10000, O.i32p(acts_o))
    assert np.array_equal(actions, acts_o.astype(np.uint8))
    assert reward == reward_o
    assert np.all(actions == 1) and reward / 10000 > 3.0
    # the initial (all-zero) table ties: argmax picks the first action, i.e. always Left (2.0 per step, minus slips)
    a0, r0 = ra.chain_tabular_q_eval(np.zeros((5, 2)), seed=0, n_steps=2000)
    assert np.all(a0 == 0)


def test_different_seeds_and_thread_counts_differ_but_are_deterministic():
    a = ra.chain_tabular_q_train(seed=1, n_threads=2, n_periods=2, min_worker_steps=300)
    b = ra.chain_tabular_q_train(seed=1, n_threads=2, n_periods=2, min_worker_steps=300)
    c = ra.chain_tabular_q_train(seed=2, n_threads=2, n_periods=2, min_worker_steps=300)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], c[0])
