"""The C++ host side above the C ABI (relearn_amd/csrc/host/agents.hpp: the reference's trait surface — BuildAgent,
Agent, BatchUpdate, StatsLogger, train loop — in C++ because the reference is compiled code and no Rust toolchain
exists here).  CPU: the demo program compiles and links against the library.  GPU: it runs, and every number it
reports (parameter checksums, logged scalars, counters) equals the same runs driven through the ctypes binding."""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import relearn_amd as ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_api_demo.cpp")


def build_demo():
    ra.build()
    out = os.path.join(tempfile.mkdtemp(), "host_api_demo")
    libdir = os.path.join(ROOT, "relearn_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", ROOT, SRC, "-o", out,
                           "-L", libdir, "-lrelearn_hip", "-Wl,-rpath," + libdir])
    return out


def test_cpp_host_api_compiles_and_links():
    exe = build_demo()
    assert os.path.exists(exe)
    hdr = open(os.path.join(ROOT, "relearn_amd", "csrc", "host", "agents.hpp")).read()
    for name in ("BuildAgentError", "StatsLogger", "ActorCriticConfig", "TrpoConfig", "PpoConfig", "ValuesOptConfig",
                 "DqnConfig", "batch_update", "build_agent", "train_batched", "ActorMode", "class Policy", "class Critic",
                 "class Optimizer", "class TrustRegionOptimizer", "class Actor", "class PolicyActor",
                 "class Trpo final : public Policy", "class ValuesOpt final : public Critic",
                 "class ConjugateGradientOptimizer final : public TrustRegionOptimizer"):
        assert name in hdr


def checksum(p):
    w = 1 + (np.arange(len(p)) % 7)
    return float((p.astype(np.float64) * w).sum())


@pytest.mark.gpu
def test_cpp_host_api_matches_the_ctypes_path(engine):
    exe = build_demo()
    out = json.loads(subprocess.check_output([exe], timeout=300).decode())
    # ---- examples/cartpole-trpo.rs shape, through the ctypes binding
    env = ra.CartPoleEnv(engine, 256, max_steps=500, seed_env=0, seed_actor=1)
    pol, cri = ra.Mlp(engine, 5, 128, 2), ra.Mlp(engine, 5, 128, 1)
    pol.init(2)
    cri.init(3)
    opt = ra.Adam(cri)
    traj = ra.Trajectory(engine, 256, 32, 5)
    episodes = 0
    for _ in range(2):
        ra.rollout(env, pol, traj)
        episodes += int((traj.read(ra.TRAJ_FLAG) != 0).sum())
        ra.gae(traj, cri, 0.99, 0.95)
        st = ra.trpo_update(pol, traj)
        cs = ra.critic_update(cri, opt, traj, 5)
    t = out["trpo"]
    assert t["policy_checksum"] == checksum(pol.get_params()) and t["critic_checksum"] == checksum(cri.get_params())
    s = t["scalars"]
    assert s["policy/entropy"] == st.entropy and s["policy/step_size"] == st.step_size
    assert s["policy/loss_initial"] == st.loss_initial and s["policy/loss_final"] == st.loss_final
    assert s["policy/constraint_val_final"] == st.constraint_val_final and s["critic/loss"] == cs.loss_last
    assert s["policy/num_backtracks"] == st.num_backtracks and s["policy/step_scale"] == st.step_scale
    assert t["counters"] == {"agent_update/count": 2, "sim/ep/count": episodes, "sim/step/count": 2 * 256 * 32}
    assert set(t["durations"]) == {"adv_est_time", "agent_update/time", "critic/update_time", "policy/update_time",
                                   "sim/time"}
    # ---- PPO over the recurrent module on Chain
    cenv = ra.ChainEnv(engine, 64, max_steps=100, seed_env=3, seed_actor=4)
    gp, gc = ra.GruMlp(engine, 5, 2), ra.GruMlp(engine, 5, 1)
    gp.init(11)
    gc.init(12)
    popt, copt = ra.Adam(gp), ra.Adam(gc)
    ctraj = ra.Trajectory(engine, 64, 20, 5)
    ra.rollout(cenv, gp, ctraj)
    ra.gae(ctraj, gc, 0.95, 0.95)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 2
    ps = ra.ppo_update(gp, popt, ctraj, cfg)
    gcs = ra.critic_update(gc, copt, ctraj, 2)
    g = out["ppo_gru"]
    assert g["policy_checksum"] == checksum(gp.get_params()) and g["critic_checksum"] == checksum(gc.get_params())
    assert g["scalars"]["policy/entropy"] == ps.entropy and g["scalars"]["critic/loss"] == gcs.loss_last
    # ---- the same with a stacked chain and RnnBaseConfig's / LinearConfig's initializers spelled out
    senv = ra.ChainEnv(engine, 64, max_steps=100, seed_env=3, seed_actor=4)
    sp, sc = ra.GruMlp(engine, 5, 2, 24, 16, num_layers=2), ra.GruMlp(engine, 5, 1, 24, 16, num_layers=2)
    kw = dict(input_weights=("Normal", "FanIn", 0.0), mlp_bias=("Constant", "FanAvg", 0.125))
    sp.init_with(15, **kw)
    sc.init_with(16, **kw)
    spopt, scopt = ra.Adam(sp), ra.Adam(sc)
    straj = ra.Trajectory(engine, 64, 20, 5)
    ra.rollout(senv, sp, straj)
    ra.gae(straj, sc, 0.95, 0.95)
    sps = ra.ppo_update(sp, spopt, straj, cfg)
    scs = ra.critic_update(sc, scopt, straj, 2)
    g = out["ppo_gru_stacked"]
    assert g["policy_checksum"] == checksum(sp.get_params()) and g["critic_checksum"] == checksum(sc.get_params())
    assert g["scalars"]["policy/entropy"] == sps.entropy and g["scalars"]["critic/loss"] == scs.loss_last
    # ---- PPO + reward-to-go on MemoryGame lanes (discount factor 1.0: memory.rs:74-76)
    menv = ra.MemoryEnv(engine, 64, seed_env=5, seed_actor=6)
    mp = ra.GruMlp(engine, 5, 2)
    mp.init(13)
    mtraj = ra.Trajectory(engine, 64, 24, 5)
    ra.rollout(menv, mp, mtraj)
    ra.reward_to_go(mtraj, 1.0)  # RewardToGo uses the env's own discount factor (critics/rtg.rs:14-20)
    mcfg = ra.ppo_config_default()
    mcfg.opt_steps_per_update = 2
    ms = ra.ppo_update(mp, ra.Adam(mp), mtraj, mcfg)
    g = out["ppo_memory"]
    assert g["policy_checksum"] == checksum(mp.get_params()) and g["scalars"]["policy/entropy"] == ms.entropy
    assert g["counters"]["sim/ep/count"] == 64 * 6 and g["counters"]["sim/step/count"] == 64 * 24
    # ---- examples/cartpole-dqn.rs shape
    denv = ra.CartPoleEnv(engine, 128, max_steps=500, seed_env=0, seed_actor=1)
    q = ra.Mlp(engine, 5, 128, 2)
    q.init(7)
    dcfg = ra.dqn_config_default()
    dcfg.minibatch_steps, dcfg.opt_steps_per_update, dcfg.buffer_capacity = 1000, 3, 256
    dcfg.update_first, dcfg.update_rest, dcfg.exploration_period = 128 * 40, 128 * 10, 100000
    dcfg.discount_factor = 0.99
    for i in range(8):
        dcfg.agent_key[i] = i + 1
    dqn = ra.Dqn(denv, q, ra.Adam(q), dcfg)
    steps = eps = 0
    for _ in range(2):
        m, _slack = dqn.min_update_size()
        cst = dqn.collect((m + 127) // 128)
        steps += cst.steps
        eps += cst.episodes_ended
        rate = dqn.exploration_rate(True)
        ust = dqn.update()
    d = out["dqn"]
    assert d["policy_checksum"] == checksum(q.get_params())
    assert d["scalars"]["loss"] == ust.loss_last and d["scalars"]["exploration_rate"] == rate
    assert d["scalars"]["global_steps"] == ust.global_steps == steps
    assert d["counters"] == {"sim/ep/count": eps, "sim/step/count": steps}
    # ---- MlpConfig { hidden_sizes: [64, 64] } for policy and critic: the general per-layer path
    env2 = ra.CartPoleEnv(engine, 128, max_steps=500, seed_env=7, seed_actor=8)
    p2, c2 = ra.Mlp(engine, 5, [64, 64], 2), ra.Mlp(engine, 5, [64, 64], 1)
    p2.init(9)
    c2.init(10)
    t2 = ra.Trajectory(engine, 128, 16, 5)
    ra.rollout(env2, p2, t2)
    ra.gae(t2, c2, 0.99, 0.95)
    st2 = ra.trpo_update(p2, t2)
    ra.critic_update(c2, ra.Adam(c2), t2, 3)
    g2 = out["trpo_two_layers"]
    assert g2["policy_checksum"] == checksum(p2.get_params()) and g2["critic_checksum"] == checksum(c2.get_params())
    assert g2["scalars"]["policy/step_size"] == st2.step_size
    # ---- Actor::act on single observations repeats the batched rollout's actions (same stream words, same arithmetic)
    assert out["actor"]["checked"] == 6 * 10 and out["actor"]["mismatches"] == 0


def build_example():
    ra.build()
    out = os.path.join(tempfile.mkdtemp(), "cartpole_trpo")
    libdir = os.path.join(ROOT, "relearn_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", ROOT,
                           os.path.join(ROOT, "examples", "cartpole_trpo.cpp"), "-o", out, "-L", libdir,
                           "-lrelearn_hip", "-Wl,-rpath," + libdir])
    return out


def test_cartpole_trpo_example_compiles():
    assert os.path.exists(build_example())


@pytest.mark.gpu
def test_cartpole_trpo_example_trains_saves_and_evaluates():
    """examples/cartpole_trpo.cpp, the counterpart of the reference's examples/cartpole-trpo.rs: trains with the display
    and TensorBoard loggers, saves the evaluation actor as CBOR, reloads it and evaluates — and the agent has learned"""
    exe = build_example()
    out_dir = tempfile.mkdtemp()
    log = subprocess.check_output([exe, "--lanes", "2048", "--periods", "24", "--out", out_dir], timeout=300).decode()
    lengths = [float(l.split()[1]) for l in log.splitlines() if l.startswith("sim/ep/length_mean")]
    assert len(lengths) == 12 and lengths[0] < 40 and lengths[-1] > 3 * lengths[0], lengths  # one display chunk per 2 updates
    for name in ("policy/entropy", "policy/step_size", "critic/loss", "agent_update/count", "sim/step/count"):
        assert any(l.startswith(name) for l in log.splitlines()), name
    assert any(f.startswith("events.out.tfevents.") for f in os.listdir(out_dir))
    actor = os.path.join(out_dir, "actor.cbor")
    assert os.path.getsize(actor) > 4 * 1026
    ev = subprocess.check_output([exe, actor], timeout=120).decode()
    mean_len = float(ev.strip().split()[-1])
    assert mean_len > 3 * lengths[0], ev  # the reloaded actor plays as well as the trained one
