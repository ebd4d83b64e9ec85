"""The LSTM restatement (oracle/seq_impl.inc, cell = ORACLE_CELL_LSTM; reference Lstm = RnnBase<LstmImpl>,
src/torch/modules/seq/rnn/lstm.rs:12-51) against what pins it: torch.lstm_cell + autograd through time on lane
trajectories with episode boundaries (tests/golden/torch_golden_lstm.json, generator committed next to it), the
reference's own module property "packed sequence == iterated step" (modules/testing.rs:124-156), and structural
properties of the initialisation (RnnBaseConfig::default, seq/rnn/mod.rs:36-45).  CPU only."""
import json
import os

import numpy as np
import pytest

import oracle as O

L = O.lib()
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "torch_golden_lstm.json")) as f:
    GOLD = json.load(f)


def _case(name):
    c = GOLD[name]
    D, H, H2, A = c["dims"]
    n, T = c["n"], c["T"]
    s = O.LstmShape(D, H, H2, A)
    traj = dict(obs=np.array(c["obs"]).reshape(D, T + 1, n), term_obs=np.array(c["term_obs"]).reshape(D, T, n),
                flag=np.array(c["flag"], np.uint8).reshape(T, n))
    return c, s, traj


def test_parameter_count():
    # 4 gate blocks instead of the GRU's 3 (GATES_MULTIPLE, lstm.rs:20), same head
    g, l = O.GruShape(5, 128, 128, 2), O.LstmShape(5, 128, 128, 2)
    assert L.oracle_gru_num_params(l) - L.oracle_gru_num_params(g) == 128 * 5 + 128 * 128 + 2 * 128
    assert L.oracle_gru_num_params(l) == 4 * 128 * 5 + 4 * 128 * 128 + 8 * 128 + 128 * 128 + 128 + 2 * 128 + 2


@pytest.mark.parametrize("name,tol", [("lstm_f32", 3e-6), ("lstm_f64", 1e-13)])
def test_forward_outputs_and_successor_outputs(name, tol):
    c, s, traj = _case(name)
    f64 = name.endswith("f64")
    out, succ = O.gru_seq_forward(s, np.array(c["params"]), traj, f64=f64)
    assert np.max(np.abs(out.reshape(-1) - np.array(c["out"]))) < tol
    assert np.max(np.abs(succ.reshape(-1) - np.array(c["succ_out"]))) < tol
    assert np.count_nonzero(np.array(c["succ_out"])) > 0


@pytest.mark.parametrize("name,rtol", [("lstm_f32", 2e-5), ("lstm_f64", 1e-11)])
def test_backward_through_time_against_autograd(name, rtol):
    c, s, traj = _case(name)
    f64 = name.endswith("f64")
    A, T, n = s.out_dim, c["T"], c["n"]
    g = O.gru_seq_backward(s, np.array(c["params"]), traj, np.array(c["dout"]).reshape(A, T, n), f64=f64)
    want = np.array(c["grad"])
    assert np.max(np.abs(g - want)) <= rtol * np.max(np.abs(want))
    assert np.count_nonzero(want) > 0.8 * len(want)  # every block gets gradient (a few ReLU units of the tiny head are dead)


def test_step_equals_sequence_forward():
    """modules/testing.rs:124-156: the sequence forward equals the cell stepped by hand (here bit for bit)"""
    c, s, traj = _case("lstm_f32")
    p = np.array(c["params"], np.float32)
    out, _ = O.gru_seq_forward(s, p, traj)
    obs = traj["obs"].astype(np.float32)
    for i in range(c["n"]):
        state = np.zeros(2 * s.hidden, np.float32)  # [h; c], LstmImpl::initial_cell_state (lstm.rs:22-31)
        for t in range(c["T"]):
            y = np.zeros(s.out_dim, np.float32)
            L.oracle_gru_step_f32(s, O.f32p(p), O.f32p(np.ascontiguousarray(obs[:, t, i])), O.f32p(state), O.f32p(y))
            assert np.array_equal(y, out[:, t, i])
            if traj["flag"][t, i] != 0:
                state[:] = 0


def test_initialisation_structure():
    s = O.LstmShape(5, 128, 128, 2)
    p = O.gru_init(s, 9)
    H, D = 128, 5
    wih = p[:4 * H * D].reshape(4 * H, D)
    whh = p[4 * H * D:4 * H * D + 4 * H * H].reshape(4 * H, H).astype(np.float64)
    lim = np.sqrt(6.0 / (D + 4 * H))  # Glorot-uniform over the whole [4H, in] matrix
    assert np.abs(wih).max() <= lim * (1 + 1e-6) and np.abs(wih).max() > 0.9 * lim
    assert np.max(np.abs(whh.T @ whh - np.eye(H))) < 1e-5  # orthonormal columns (initializers.rs:328-364, test :382-458)
    biases = p[4 * H * D + 4 * H * H:4 * H * D + 4 * H * H + 8 * H]
    assert np.all(biases == 0)
    assert not np.array_equal(p, O.gru_init(s, 10))


def test_memory_lanes_rollout_with_the_lstm_policy():
    """the lane rollouts take either cell: teacher-forcing the recorded observations reproduces the logits that sampled
    the recorded actions (the recurrent state restarts at the same places)"""
    s = O.LstmShape(5, 16, 16, 2)
    p = O.gru_init(s, 3)
    sim = O.MemoryLaneSim(8, seed_env=1, seed_actor=2)
    tr = sim.rollout_gru(s, p, 24)
    logits, _ = O.gru_seq_forward(s, p, tr, want_succ=False)
    assert logits.shape == (2, 24, 8) and np.all(np.isfinite(logits))
    assert (tr["flag"] == O.TERMINATE).sum() == 8 * 6
    again = O.MemoryLaneSim(8, seed_env=1, seed_actor=2).rollout_gru(s, p, 24)
    assert all(np.array_equal(tr[k], again[k]) for k in tr)


@pytest.mark.parametrize("name,rtol", [("lstm_f32", 2e-5), ("lstm_f64", 1e-11)])
def test_forward_mode_derivative_through_time(name, rtol):
    """the LSTM JVP against torch.autograd.functional.jvp of all step outputs along a parameter tangent"""
    c, s, traj = _case(name)
    f64 = name.endswith("f64")
    od = O.gru_seq_jvp(s, np.array(c["params"]), np.array(c["tangent"]), traj, f64=f64)
    want = np.array(c["out_dot"]).reshape(od.shape)
    assert np.max(np.abs(od - want)) <= rtol * np.max(np.abs(want))


def test_fisher_vector_product_is_symmetric_and_positive():
    """F = J^T (diag p - p p^T) J / B through the LSTM: v.Fw == w.Fv and v.Fv > 0"""
    s = O.LstmShape(5, 16, 12, 2)
    sim = O.ChainLaneSim(24, max_steps=7, seed_env=2, seed_actor=3)
    p = O.gru_init(s, 4).astype(np.float64)
    traj = sim.rollout_gru(s, p.astype(np.float32), 18, threads=2)
    rng = np.random.default_rng(0)
    v, w = rng.normal(size=len(p)), rng.normal(size=len(p))
    Fv = O.gru_policy_fvp(s, p, v, traj, 0.0, f64=True)
    Fw = O.gru_policy_fvp(s, p, w, traj, 0.0, f64=True)
    assert abs(v @ Fw - w @ Fv) <= 1e-10 * abs(v @ Fw)
    assert v @ Fv > 0
