"""The stacked-layer oracles (oracle/stacked.py in NumPy f64; oracle/stack_impl.inc in C, forward) against the PyTorch
vectors of tests/golden/torch_golden_stacked.json (torch.gru_cell / lstm_cell layer by layer, asserted equal to
torch.nn.GRU / LSTM(num_layers) by the generator), against each other, and the initialisation stream."""
import json
import os

import numpy as np
import pytest

import oracle as O
from oracle import stacked as S

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "torch_golden_stacked.json")) as f:
    GOLD = json.load(f)

CASES = [k for k in GOLD if k != "generator"]


def load(c):
    D, H, L, H2, A = c["dims"]
    n, T = c["n"], c["T"]
    cell = S.LSTM if c["cell"] == "lstm" else S.GRU
    spec = S.Spec(cell, D, H, L, H2, A)
    traj = {"obs": np.array(c["obs"]).reshape(D, T + 1, n), "term_obs": np.array(c["term_obs"]).reshape(D, T, n),
            "flag": np.array(c["flag"], dtype=np.uint8).reshape(T, n)}
    return spec, traj, np.array(c["params"]), (A, T, n)


@pytest.mark.parametrize("name", CASES)
def test_numpy_restatement_against_torch(name):
    c = GOLD[name]
    spec, traj, params, shp = load(c)
    tol = 1e-11 if c["dtype"] == "float64" else 2e-5
    out, succ, _ = S.forward(spec, params, traj)
    assert np.allclose(out, np.array(c["out"]).reshape(shp), rtol=tol, atol=tol)
    assert np.allclose(succ, np.array(c["succ_out"]).reshape(shp), rtol=tol, atol=tol)
    assert np.abs(np.array(c["succ_out"])).max() > 0
    g = S.backward(spec, params, traj, np.array(c["dout"]).reshape(shp))
    want = np.array(c["grad"])
    assert np.abs(g - want).max() <= tol * 10 * max(1.0, np.abs(want).max())
    od = S.jvp(spec, params, np.array(c["tangent"]), traj)
    wd = np.array(c["out_dot"]).reshape(shp)
    assert np.abs(od - wd).max() <= tol * 10 * max(1.0, np.abs(wd).max())


@pytest.mark.parametrize("name", CASES)
def test_c_forward_against_torch_and_numpy(name):
    c = GOLD[name]
    spec, traj, params, shp = load(c)
    shape = O.GruShape(spec.D, spec.H, spec.H2, spec.A, O.CELL_LSTM if spec.cell == S.LSTM else O.CELL_GRU)
    assert O.lib().oracle_stack_num_params(shape, spec.L) == spec.num_params() == len(params)
    o64, s64 = O.stack_seq_forward(shape, spec.L, params, traj, f64=True)
    on, sn, _ = S.forward(spec, params, traj)
    assert np.allclose(o64, on, rtol=1e-12, atol=1e-13) and np.allclose(s64, sn, rtol=1e-12, atol=1e-13)
    tol = 1e-11 if c["dtype"] == "float64" else 2e-5
    assert np.allclose(o64, np.array(c["out"]).reshape(shp), rtol=tol, atol=tol)
    o32, s32 = O.stack_seq_forward(shape, spec.L, params.astype(np.float32), traj, f64=False)
    assert np.allclose(o32, o64, rtol=1e-4, atol=1e-5) and np.allclose(s32, s64, rtol=1e-4, atol=1e-5)


def test_one_layer_is_the_single_layer_chain():
    """num_layers = 1 of the stacked restatements == the single-layer oracle (seq_impl.inc), bit for bit in f32"""
    rng = np.random.default_rng(3)
    for cell in (O.CELL_GRU, O.CELL_LSTM):
        shape = O.GruShape(3, 6, 4, 2, cell)
        p = O.gru_init(shape, 9)
        assert np.array_equal(p, O.stack_init(shape, 1, 9))
        n, T = 5, 9
        flag = (rng.random((T, n)) < 0.2).astype(np.uint8) * rng.integers(1, 3, (T, n)).astype(np.uint8)
        traj = {"obs": rng.standard_normal((3, T + 1, n)).astype(np.float32),
                "term_obs": rng.standard_normal((3, T, n)).astype(np.float32), "flag": flag}
        a, b = O.gru_seq_forward(shape, p, traj)
        c, d = O.stack_seq_forward(shape, 1, p, traj)
        assert np.array_equal(a, c) and np.array_equal(b, d)
        spec = S.Spec(S.LSTM if cell == O.CELL_LSTM else S.GRU, 3, 6, 1, 4, 2)
        dout = rng.standard_normal((2, T, n))
        g1 = O.gru_seq_backward(shape, p, traj, dout, f64=True)
        g2 = S.backward(spec, p, traj, dout)
        assert np.allclose(g1, g2, rtol=1e-10, atol=1e-12)
        v = rng.standard_normal(len(p))
        assert np.allclose(O.gru_seq_jvp(shape, p, v, traj, f64=True), S.jvp(spec, p, v, traj), rtol=1e-10, atol=1e-12)
        assert np.allclose(O.gru_policy_fvp(shape, p, v, traj, 1e-5, f64=True), S.policy_fvp(spec, p, v, traj, 1e-5),
                           rtol=1e-9, atol=1e-12)


def test_init_layers():
    """RnnWeights::new's layer loop: every layer's W_ih within its Glorot limit, W_hh with orthonormal columns, biases
    zero; the first layer's draws are those of the single-layer module"""
    shape = O.GruShape(5, 16, 8, 2, O.CELL_GRU)
    L = 3
    p = O.stack_init(shape, L, 21)
    spec = S.Spec(S.GRU, 5, 16, L, 8, 2)
    layers, head = spec.unpack(p)
    one = O.gru_init(shape, 21)
    n0 = 48 * 5 + 48 * 16 + 96
    assert np.array_equal(p[:n0], one[:n0])
    for l, w in enumerate(layers):
        K = 5 if l == 0 else 16
        assert np.abs(w["Wih"]).max() <= np.sqrt(6.0 / (K + 48)) + 1e-7 and np.abs(w["Wih"]).max() > 0.1
        assert np.allclose(w["Whh"].T @ w["Whh"], np.eye(16), atol=1e-6)
        assert not w["bih"].any() and not w["bhh"].any()
    assert head["W1"].any() and head["W2"].any()


def test_init_with_initializers():
    """RnnBaseConfig's initializers spelled out: the defaults reproduce oracle_stack_init bit for bit; other choices have
    the statistics their variance scale prescribes (1-D biases: fan_in 1, fan_out = gate rows) and draw from one stream"""
    shape = O.GruShape(5, 16, 8, 2, O.CELL_LSTM)
    for L in (1, 3):
        assert np.array_equal(O.stack_init_with(shape, L, 33), O.stack_init(shape, L, 33))
    inits = (("Normal", "FanIn", 0.0), ("Uniform", "Constant", 0.25), ("Normal", "FanOut", 0.0), ("Orthogonal", "FanAvg", 0.0),
             ("Constant", "FanAvg", 0.5))
    p = O.stack_init_with(shape, 2, 7, inits)
    spec = S.Spec(S.LSTM, 5, 16, 2, 8, 2)
    layers, head = spec.unpack(p)
    assert abs(layers[1]["Wih"].std() - np.sqrt(1.0 / 16)) < 0.03           # Normal, variance 1 / fan_in (16)
    assert np.abs(layers[0]["Whh"]).max() <= np.sqrt(3 * 0.25) + 1e-6        # Uniform, variance 0.25
    assert abs(layers[0]["Whh"].var() - 0.25) < 0.03
    assert abs(layers[0]["bih"].std() - np.sqrt(1.0 / 64)) < 0.04            # Normal, variance 1 / fan_out (64 gate rows)
    assert not np.array_equal(layers[0]["bih"], layers[0]["bhh"])
    assert np.allclose(head["W1"] @ head["W1"].T, np.eye(8), atol=1e-6)      # wide [8, 16]: orthonormal rows
    assert np.all(head["b1"] == 0.5) and np.all(head["b2"] == 0.5)
