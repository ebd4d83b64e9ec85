"""The C-ABI library loads without a GPU and exports exactly the entry points include/relearn_hip.h declares;
compute entry points fail loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import pytest

import relearn_amd as ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "relearn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    ra.build()
    lib = ra.lib()
    decl = declared_symbols()
    assert len(decl) >= 45
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(ra.ABI_SYMBOLS) == decl, set(ra.ABI_SYMBOLS) ^ set(decl)
    assert lib.rl_abi_version() == 6


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "relearn_hip.h")).read()
    assert text.count("src/") >= 20  # file:line citations of the interfaces each entry point replaces
    for needle in ("src/envs/mod.rs:76-127", "src/torch/agents/policies/trpo.rs:97-164",
                   "src/torch/agents/critics/opt.rs:100-126", "src/simulation/steps.rs:113-167"):
        assert needle in text


def test_defaults_match_reference_configs():
    lib = ra.lib()
    p = ra.cartpole_params_default()
    assert (p.gravity, p.mass_cart, p.mass_pole, p.length_half_pole) == (9.8, 1.0, 0.1, 0.5)
    assert (p.friction_cart, p.friction_pole, p.time_step) == (0.01, 0.01, 0.02)
    assert (p.action_force, p.max_pos, p.discount_factor) == (10.0, 2.4, 0.99)
    assert abs(p.max_angle - 0.20943951023931956) < 1e-16
    t = ra.trpo_config_default()
    assert (t.iterations, t.max_backtracks, t.backtrack_ratio, t.hpv_reg_coeff, t.max_policy_step_kl,
            t.accept_violation) == (10, 15, 0.8, 1e-5, 0.01, 0)
    a = ra.adam_config_default()
    assert (a.learning_rate, a.beta1, a.beta2, a.weight_decay, a.eps) == (1e-3, 0.9, 0.999, 0.0, 1e-8)
    v = ra.values_opt_config_default()  # ValuesOptConfig::default (critics/opt.rs:40-51)
    assert (v.opt_steps_per_update, v.target) == (80, ra.VALUE_TARGET_REWARD_TO_GO)
    assert abs(v.discount_factor - 0.99) < 1e-7
    d = ra.dqn_config_default()  # DqnConfig::default (dqn.rs:57-72), ExplorationRateSchedule::default
    assert (d.target, d.minibatch_steps, d.opt_steps_per_update) == (ra.DQN_TARGET_REWARD_TO_GO, 100_000, 50)
    assert (d.exploration_kind, d.exploration_start, d.exploration_end, d.exploration_period) == (
        ra.SCHEDULE_LINEAR_ANNEALED, 1.0, 0.1, 10_000_000)
    assert (d.update_kind, d.update_first, d.update_rest) == (ra.COLLECT_FIRST_REST, 1_000_000, 100_000)


def test_no_cpu_fallback():
    n = C.c_int32(-1)
    assert ra.lib().rl_device_count(C.byref(n)) == ra.OK
    if n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(ra.RelearnError) as e:
        ra.Engine(0)
    assert e.value.code == ra.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """The product path must never import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "relearn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                bad = re.findall(r"import\s+oracle|from\s+oracle|#include\s*[<\"][^>\"]*oracle|liboracle|oracle/", src)
                assert not bad, (os.path.join(dirpath, f), bad)


def test_integration_doc_binds_every_symbol():
    """INTEGRATION.md shows the reference-side `extern "C"` block: it must name every function the header declares
    (a maintainer copying it gets the complete surface, and the struct layouts it shows are the current ones)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "relearn_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    declared = sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", header)))
    assert [s for s in declared if "fn %s(" % s not in doc] == []
    for field in ("chain_size", "memory_num_actions", "memory_history_len", "bandit_values"):  # rl_env_config's latest fields
        assert field in doc


def test_integration_doc_struct_layouts_match_the_header():
    """every plain-data struct of the header appears in INTEGRATION.md as #[repr(C)] with the same fields in the same order"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "relearn_hip.h")).read().split("\n")
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    rust = {}
    for m in re.finditer(r"pub struct (rl_[a-z_]+)\s*\{(.*?)\}", doc, re.S):
        rust[m.group(1)] = re.findall(r"pub ([a-z_0-9]+)\s*:", re.sub(r"//.*", "", m.group(2)))
    checked = 0
    for end, line in enumerate(header):
        m = re.match(r"\} (rl_[a-z_]+);", line)
        if not m:
            continue
        start = max(i for i in range(end) if header[i].startswith("typedef "))
        if not header[start].startswith("typedef struct {"):
            continue  # an enum
        body = re.sub(r"/\*.*?\*/", "", "\n".join(header[start + 1:end]), flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                fields += [re.sub(r"\[.*\]", "", n).strip() for n in decl.split(None, 1)[1].split(",")]
        assert rust.get(m.group(1)) == fields, (m.group(1), fields, rust.get(m.group(1)))
        checked += 1
    assert checked >= 11


def test_headers_are_plain_c():
    """the boundary is a C ABI: every header under include/ compiles as C99 with -pedantic (what a cgo / bindgen / ctypes
    generator would be fed), and relearn_hip.h alone declares everything its prototypes use"""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for headers in (["relearn_hip.h"], ["rl_chacha.h", "rl_detmath.h"], ["relearn_hip.h", "rl_chacha.h", "rl_detmath.h"]):
        src = "".join('#include "%s"\n' % h for h in headers) + "int main(void) { return 0; }\n"
        with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
            f.write(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I",
                               os.path.join(root, "include"), "-fsyntax-only", f.name])
        os.unlink(f.name)
