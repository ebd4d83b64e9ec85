"""CPU-side hygiene: the oracle (oracle/*.c) and the host side of the C ABI (relearn_amd/csrc/host_abi.cpp, host/*.hpp,
tests/cpp/*.cpp) built ONCE with -fsanitize=address,undefined and driven through what the CPU suite already exercises.
(GPU AddressSanitizer / XNACK runs are not available on the pool; the device code is covered by the parity tests.)
Any report — heap overflow, use after free, misaligned or out-of-range access, signed overflow — fails the test."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
# leaks are not checked: the interpreter the oracle is loaded into keeps its own allocations alive at exit
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
           "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}


def _runtime(name):
    path = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
    if not os.path.isabs(path):
        pytest.skip("%s is not installed with this gcc" % name)
    return path


@pytest.fixture(scope="module")
def workdir():
    return tempfile.mkdtemp(prefix="relearn_san_")


def test_oracle_under_asan_and_ubsan(workdir):
    """the oracle's own tests — reference fixtures, NumPy double-entry table, lane simulator, packed features, TRPO / CG,
    DQN store, the recurrent passes — against a sanitizer build of oracle/*.c loaded into a fresh interpreter"""
    asan = _runtime("libasan.so")
    lib = os.path.join(workdir, "liboracle_san.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("prng.c", "envs.c", "sim.c", "packed.c", "nn.c", "lanes.c",
                                                      "dqn.c", "seq.c")]
    subprocess.check_call(["gcc", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-mavx2", "-mfma", "-fopenmp",
                           "-Wall"] + SAN + srcs + ["-o", lib, "-lm"])
    tests = ["tests/test_oracle_reference_fixtures.py", "tests/test_oracle_numpy_golden.py", "tests/test_oracle_lanes.py",
             "tests/test_oracle_dqn.py", "tests/test_oracle_gru.py", "tests/test_oracle_lstm.py",
             "tests/test_oracle_ppo.py", "tests/test_detmath_prng.py"]
    env = dict(os.environ, ORACLE_LIB=lib, LD_PRELOAD=asan, OMP_NUM_THREADS="4", **SAN_ENV)
    proc = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests,
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=280)
    out = proc.stdout.decode()
    assert proc.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out


def test_cpu_baseline_period_under_asan_and_ubsan(workdir):
    """the train_parallel-structured CPU baseline of bench.py (worker threads, VecBuffers, packed features, GAE, TRPO, 20
    Adam steps, the intra-op-parallel update leg, the rollout-only leg) under the sanitizers"""
    asan = _runtime("libasan.so")
    lib = os.path.join(workdir, "liboracle_san.so")
    assert os.path.exists(lib), "built by test_oracle_under_asan_and_ubsan"
    code = (
        "import ctypes as C, oracle as O\n"
        "ps, cs = O.MlpShape(5, 128, 2), O.MlpShape(5, 128, 1)\n"
        "pp, cp = O.mlp_init(ps, 2), O.mlp_init(cs, 3)\n"
        "opt = O.lib().oracle_adam_new(len(cp)); st = O.PeriodStats()\n"
        "O.lib().oracle_cartpole_trpo_period_ex(0, 0, 4, 1500, 100, 500, 128, O.f32p(pp), O.f32p(cp), opt, 20, 4, C.byref(st))\n"
        "got = C.c_uint64()\n"
        "O.lib().oracle_cartpole_rollout_only(1, 4, 3000, 1024, 500, 128, O.f32p(pp), C.byref(got))\n"
        "O.lib().oracle_adam_free(opt)\n"
        "assert st.steps >= 6000 and st.update_intraop_seconds > 0 and got.value == 12000\n"
        "print('baseline ok', st.steps)\n")
    env = dict(os.environ, ORACLE_LIB=lib, LD_PRELOAD=asan, **SAN_ENV)
    proc = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=280)
    out = proc.stdout.decode()
    assert proc.returncode == 0 and "baseline ok" in out, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]


@pytest.mark.parametrize("program,sources,args", [
    ("host_sanitize", ["tests/cpp/host_sanitize_main.cpp", "relearn_amd/csrc/host_abi.cpp"], []),
    ("logging_demo", ["tests/cpp/logging_demo.cpp"], None),
    ("host_envs_demo", ["tests/cpp/host_envs_demo.cpp"], ["3", "4", "9", "5", "300"]),
])
def test_host_code_under_asan_and_ubsan(workdir, program, sources, args):
    """host_abi.cpp (configuration 1: train_parallel + tabular Q on CPU threads), the logging mirror and the scalar
    environments, each as a sanitizer-instrumented executable; leak checking ON here (plain C++ programs)"""
    exe = os.path.join(workdir, program)
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", "-Wall", "-pthread", "-I", ROOT, "-I",
                           os.path.join(ROOT, "include")] + SAN + [os.path.join(ROOT, s) for s in sources] + ["-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS=SAN_ENV["UBSAN_OPTIONS"])
    runs = [args]
    if args is None:  # logging_demo: every mode of the program (tensorboard takes an output directory)
        modes = sorted(set(__import__("re").findall(r'mode == "([a-z_]+)"', open(os.path.join(ROOT, sources[0])).read())))
        assert "chunks" in modes and "tensorboard" in modes
        runs = [[m] + ([tempfile.mkdtemp(prefix="relearn_san_log_")] if m == "tensorboard" else []) for m in modes]
    for run_args in runs:
        proc = subprocess.run([exe] + run_args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              timeout=200)
        out = proc.stdout.decode()
        assert proc.returncode == 0, (run_args, out[-4000:])
        assert "AddressSanitizer" not in out and "LeakSanitizer" not in out and "runtime error:" not in out, out[-4000:]
