"""The oracle's hand-derived update math against vectors produced by PyTorch CPU autograd
(tests/golden/torch_golden.json; generator: tests/golden/make_torch_golden.py).  CPU only.

The Hessian-vector product there is the reference's own construction (double backward of the mean KL,
conjugate_gradient.rs:262-339); the oracle restates it analytically as the Fisher form — this test pins
that equivalence, the gradient of the surrogate, the CG/line-search bookkeeping and libtorch-style Adam."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "torch_golden.json")))
L = O.lib()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _case(name):
    c = G[name]
    D, H, A = c["dims"]
    return c, O.MlpShape(D, H, A)


def test_policy_gradient_and_hvp_f64():
    c, s = _case("trpo_f64")
    p = np.array(c["params0"])
    x = np.ascontiguousarray(c["obs"], dtype=np.float64)
    a = np.array(c["actions"], np.int64)
    adv = np.array(c["adv"])
    g = np.zeros_like(p)
    loss = C.c_double()
    L.oracle_policy_grad_f64(s, O.f64p(p), O.f64p(x), O.i64p(a), O.f64p(adv), len(a), O.f64p(g), C.byref(loss))
    assert rel(g, c["grad"]) < 1e-12
    assert abs(loss.value - c["loss0"]) < 1e-13
    hv = np.zeros_like(p)
    L.oracle_policy_fvp_f64(s, O.f64p(p), O.f64p(x), len(a), O.f64p(np.array(c["v"])), c["reg"], O.f64p(hv))
    # analytic Fisher-vector product == autograd double backward of KL(pi0 || pi) at theta0
    assert rel(hv, c["hvp_v"]) < 1e-10
    lo, kl = C.c_double(), C.c_double()
    L.oracle_policy_loss_kl_f64(s, O.f64p(np.array(c["params_pert"])), O.f64p(p), O.f64p(x), O.i64p(a),
                                O.f64p(adv), len(a), C.byref(lo), C.byref(kl))
    assert abs(lo.value - c["loss_pert"]) < 1e-12 and abs(kl.value - c["kl_pert"]) < 1e-12


def test_trpo_step_f64_matches_torch_transcription():
    c, s = _case("trpo_f64")
    x = np.ascontiguousarray(c["obs"], dtype=np.float64)
    p_new, st, sd = O.trpo_update(s, np.array(c["params0"]), x, np.array(c["actions"], np.int64),
                                  np.array(c["adv"]), f64=True)
    assert st.cg_iterations == c["cg_iterations"]
    # 24 samples < 66 parameters: the Fisher matrix is singular up to the 1e-5 regulariser, and 10 CG iterations
    # amplify the 1e-10 difference between the analytic product and autograd's double backward to ~5e-4
    assert rel(sd, c["step_dir"]) < 5e-3
    assert abs(st.step_size - c["step_size"]) < 2e-3 * c["step_size"]
    assert st.num_backtracks == c["num_backtracks"]
    assert (st.status == O.OPT_OK) == c["status_ok"]
    assert abs(st.entropy - c["entropy"]) < 1e-12
    assert abs(st.loss_final - c["loss_final"]) < 2e-3 * abs(c["loss_final"])
    assert abs(st.constraint_val_final - c["kl_final"]) < 5e-3 * c["kl_final"]
    assert rel(p_new, c["new_params"]) < 2e-3


def test_policy_gradient_and_hvp_f32():
    c, s = _case("trpo_f32")
    p = np.array(c["params0"], np.float32)
    x = np.ascontiguousarray(c["obs"], dtype=np.float32)
    a = np.array(c["actions"], np.int64)
    adv = np.array(c["adv"], np.float32)
    g = np.zeros_like(p)
    loss = C.c_float()
    L.oracle_policy_grad_f32(s, O.f32p(p), O.f32p(x), O.i64p(a), O.f32p(adv), len(a), O.f32p(g), C.byref(loss))
    assert rel(g, c["grad"]) < 2e-6
    hv = np.zeros_like(p)
    L.oracle_policy_fvp_f32(s, O.f32p(p), O.f32p(x), len(a), O.f32p(np.array(c["v"], np.float32)), c["reg"],
                            O.f32p(hv))
    assert rel(hv, c["hvp_v"]) < 5e-6  # autograd's f32 double backward carries its own rounding
    # the full f32 TRPO step: same bookkeeping; CG amplifies f32 rounding (see test_gpu_parity), so compare
    # with the f64 ground truth of the same problem rather than bit-for-bit with torch's f32 run
    t64 = G["trpo_f64"]
    p_new, st, sd = O.trpo_update(s, p, x, a, adv)
    assert st.status == O.OPT_OK and st.cg_iterations == 10
    assert abs(st.step_size - t64["step_size"]) < 0.1 * t64["step_size"]
    assert abs(st.num_backtracks - t64["num_backtracks"]) <= 1


def test_critic_gradient_and_adam_f32():
    c, s = _case("critic_f32")
    p = np.array(c["params0"], np.float32)
    x = np.ascontiguousarray(c["obs"], dtype=np.float32)
    t = np.array(c["targets"], np.float32)
    g = np.zeros_like(p)
    loss = C.c_float()
    L.oracle_critic_grad_f32(s, O.f32p(p), O.f32p(x), O.f32p(t), len(t), O.f32p(g), C.byref(loss))
    assert rel(g, c["grad0"]) < 2e-6
    assert abs(loss.value - c["losses"][0]) < 1e-5 * c["losses"][0]
    ad = L.oracle_adam_new(len(p))
    ac = O.AdamCfg()
    L.oracle_adam_cfg_default(C.byref(ac))
    losses = np.zeros(c["steps"], np.float32)
    L.oracle_critic_update_f32(s, O.f32p(p), ad, C.byref(ac), O.f32p(x), O.f32p(t), len(t), c["steps"],
                               O.f32p(losses))
    L.oracle_adam_free(ad)
    assert np.allclose(losses, c["losses"], rtol=2e-6)
    # Adam's first steps move every parameter by ~lr: agreement to 1e-6 absolute pins m/v/bias-correction/eps
    assert np.abs(p - np.array(c["params_final"])).max() < 2e-6


def test_gae_on_reference_history_fixture():
    """GAE(0.9, 0.8) + reward-to-go on the reference's 4-episode history (features.rs:293-333), packed the
    reference way, vs per-episode torch arithmetic."""
    g = G["gae_f32"]
    D, H, _ = g["dims"]
    cs = O.MlpShape(D, H, 1)
    cp = np.array(g["critic_params"], np.float32)
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_fixtures.json")))
    b = L.oracle_vecbuffer_new(1)
    succ = {"C": O.CONTINUE, "T": O.TERMINATE}
    eps = fix["history_features"]["episodes"]
    for ei, ep in enumerate(eps):
        for i, (obs, act, rew, nxt) in enumerate(ep):
            o = np.array([obs], np.float32)
            if nxt.startswith("I"):
                L.oracle_vecbuffer_write_step(b, O.f32p(o), act, rew, O.INTERRUPT,
                                              O.f32p(np.array([float(nxt[1:])], np.float32)))
            elif nxt == "C" and i == len(ep) - 1:
                # dangling episode: ends without a successor -> invalid extended row (value 0)
                L.oracle_vecbuffer_write_step(b, O.f32p(o), act, rew, O.TERMINATE, None)
            else:
                L.oracle_vecbuffer_write_step(b, O.f32p(o), act, rew, succ[nxt], None)
    arr = (C.POINTER(O.VecBuffer) * 1)(b)
    feat = L.oracle_features_from_buffers(arr, 1)
    n = feat.contents.n_steps
    adv = np.zeros(n, np.float32)
    rtg = np.zeros(n, np.float32)
    L.oracle_gae_packed(cs, O.f32p(cp), feat, g["gamma"], g["lambda"], O.f32p(adv), None)
    L.oracle_reward_to_go_packed(feat, g["gamma"], O.f32p(rtg))
    td = np.zeros(n, np.float32)  # StepValueTarget::OneStepTd (critics/mod.rs:139-150)
    L.oracle_one_step_values_packed(cs, O.f32p(cp), feat, g["gamma"], O.f32p(td))
    # un-pack by source index (buffer order = episodes in fixture order)
    src = np.array([feat.contents.src_index[i] for i in range(n)])
    adv_flat, rtg_flat, td_flat = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    adv_flat[src], rtg_flat[src], td_flat[src] = adv, rtg, td
    k = 0
    for ep, exp in zip(eps, g["episodes"]):
        m = len(ep)
        assert np.allclose(adv_flat[k:k + m], exp["adv"], atol=2e-6)
        assert np.allclose(rtg_flat[k:k + m], exp["rtg"], atol=2e-6)
        assert np.allclose(td_flat[k:k + m], exp["td"], atol=2e-6)
        k += m
    L.oracle_features_free(feat)
    L.oracle_vecbuffer_free(b)
