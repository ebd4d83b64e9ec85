"""relearn_amd — thin ctypes binding over the C ABI of ``librelearn_hip.so`` (include/relearn_hip.h).

The product is the HIP library; this module is plumbing for tests and bench.py.  There is no CPU
fallback: importing works anywhere (so the symbol table can be checked without a GPU), but creating an
``Engine`` raises unless a gfx950 device is visible, and a missing/unbuilt library raises at import of
``lib()``.
"""
import atexit
import ctypes as C
import os
import subprocess
import sys
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RELEARN_LIB", os.path.join(_HERE, "librelearn_hip.so"))  # override: ablation builds
CSRC = os.path.join(_HERE, "csrc")

# status codes (include/relearn_hip.h)
OK = 0
ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_BUILD_AGENT, ERR_BUILD_ENV = 1, 2, 3, 4, 5
ERR_BUFFER_FULL, ERR_PACKING, ERR_COMM, ERR_OPT_NAN, ERR_UNSUPPORTED = 6, 7, 8, 9, 10
SUCC_CONTINUE, SUCC_TERMINATE, SUCC_INTERRUPT = 0, 1, 2
OPT_OK, OPT_LOSS_NOT_IMPROVING, OPT_CONSTRAINT_VIOLATED, OPT_NAN_LOSS, OPT_NAN_CONSTRAINT = range(5)
ENV_CARTPOLE, ENV_CHAIN, ENV_MEMORY, ENV_BANDIT = 0, 1, 2, 3
LIMIT_NONE, LIMIT_LATENT, LIMIT_VISIBLE = 0, 1, 2
(TRAJ_OBS, TRAJ_ACTION, TRAJ_REWARD, TRAJ_FLAG, TRAJ_TERM_OBS, TRAJ_VALUES, TRAJ_ADVANTAGES,
 TRAJ_RETURNS, TRAJ_TARGETS) = range(9)
VALUE_TARGET_REWARD_TO_GO, VALUE_TARGET_ONE_STEP_TD = 0, 1
KERNEL_CLASSES = ["env_step", "rollout", "values", "gae", "policy_pass", "backward", "reduce", "small",
                  "critic_fwd", "allreduce", "critic_fused", "policy_fused", "policy_fvp"]

# every symbol include/relearn_hip.h declares (checked by tests/test_abi_symbols.py against the header)
ABI_SYMBOLS = [
    "rl_abi_version", "rl_debug_stream_words", "rl_device_count", "rl_engine_create", "rl_engine_destroy", "rl_engine_sync",
    "rl_last_error", "rl_engine_info", "rl_engine_set_kernel_variant", "rl_timer_begin", "rl_timer_end", "rl_profile_enable", "rl_profile_read",
    "rl_comm_available", "rl_comm_library_paths", "rl_comm_unique_id", "rl_comm_init", "rl_comm_destroy", "rl_comm_init_host",
    "rl_comm_ipc_handle", "rl_comm_init_ipc", "rl_comm_selftest",
    "rl_cartpole_params_default", "rl_env_create", "rl_env_destroy", "rl_env_dims", "rl_env_reset",
    "rl_env_observe", "rl_env_step", "rl_env_upload_actions", "rl_env_step_resident", "rl_env_get_state",
    "rl_env_set_state",
    "rl_mlp_create", "rl_mlp_create_layers", "rl_mlp_create_config", "rl_mlp_destroy", "rl_mlp_num_params", "rl_mlp_init", "rl_mlp_init_with", "rl_params_get", "rl_params_set",
    "rl_mlp_forward", "rl_gru_mlp_create", "rl_lstm_mlp_create", "rl_rnn_mlp_create", "rl_rnn_mlp_create_config", "rl_rnn_mlp_init_with", "rl_seq_forward",
    "rl_traj_create", "rl_traj_destroy", "rl_traj_field_bytes", "rl_traj_read", "rl_traj_write",
    "rl_rollout", "rl_gae",
    "rl_trpo_config_default", "rl_trpo_update", "rl_policy_gradient", "rl_policy_fvp", "rl_policy_loss_kl",
    "rl_adam_config_default", "rl_adam_create", "rl_adam_destroy", "rl_adam_step_host",
    "rl_critic_update", "rl_critic_gradient", "rl_values_opt_config_default", "rl_values_opt_update",
    "rl_actor_critic_update", "rl_actor_critic_update_begin", "rl_actor_critic_update_finish", "rl_engine_set_serial_update",
    "rl_ppo_config_default", "rl_ppo_update", "rl_reinforce_update", "rl_reward_to_go",
    "rl_actor_to_cbor", "rl_module_from_cbor", "rl_tensor_def_to_cbor", "rl_tensor_def_from_cbor",
    "rl_indexed_type_space_to_cbor",
    "rl_dqn_config_default", "rl_dqn_create", "rl_dqn_destroy", "rl_dqn_exploration_rate",
    "rl_dqn_min_update_size", "rl_dqn_collect", "rl_dqn_update", "rl_dqn_replay_field_bytes", "rl_dqn_replay_read",
    "rl_dqn_minibatch_sample", "rl_dqn_minibatch_read", "rl_dqn_minibatch_gradient", "rl_dqn_agent_rng_pos",
    "rl_chain_tabular_q_train", "rl_chain_tabular_q_eval",
]


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.c_uint64)


class RelearnError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rl status %d: %s" % (code, message))
        self.code = code


class CartPoleParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "gravity", "mass_cart", "mass_pole", "length_half_pole", "friction_cart", "friction_pole", "time_step",
        "action_force", "max_pos", "max_angle", "discount_factor")]


class EnvConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("limit_kind", C.c_int32), ("max_steps", C.c_uint64), ("n_lanes", C.c_uint64),
                ("lane_offset", C.c_uint64), ("seed_env", C.c_uint64), ("seed_actor", C.c_uint64),
                ("cartpole", CartPoleParams), ("chain_size", C.c_uint64), ("memory_num_actions", C.c_uint64),
                ("memory_history_len", C.c_uint64), ("bandit_values", C.c_double * 2)]


class TrpoConfig(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("max_backtracks", C.c_uint64), ("backtrack_ratio", C.c_double),
                ("hpv_reg_coeff", C.c_double), ("max_policy_step_kl", C.c_double), ("accept_violation", C.c_int32)]


class TrpoStats(C.Structure):
    _fields_ = [("entropy", C.c_double), ("step_size", C.c_double), ("loss_initial", C.c_double),
                ("loss_final", C.c_double), ("constraint_val_final", C.c_double), ("step_scale", C.c_double),
                ("num_backtracks", C.c_int64), ("status", C.c_int32), ("cg_iterations", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AdamConfig(C.Structure):
    _fields_ = [("learning_rate", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("weight_decay", C.c_double), ("eps", C.c_double)]


class CriticStats(C.Structure):
    _fields_ = [("loss_first", C.c_double), ("loss_last", C.c_double), ("steps", C.c_uint64)]


def build(force=False, check=False):
    """Compile librelearn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `check`: go through make even
    when the library exists (a no-op when it is newer than every source) — what `__graft_entry__.build()` does; the
    default leaves an existing library alone, so that a process which has it loaded never relinks it under itself."""
    if force or check or not os.path.exists(LIB_PATH):
        before = os.path.getmtime(LIB_PATH) if os.path.exists(LIB_PATH) else None
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"] + (["-B"] if force else []))
        _write_build_record(before)
    return LIB_PATH


BUILD_RECORD = os.path.join(_HERE, "build_record.json")


def _write_build_record(mtime_before):
    """what the last build() did — linked a new library, or found the shipped one newer than every source — with the
    identity of the library; bench.py copies it into its line so that a reader knows which binary a number belongs to"""
    import hashlib
    import json
    import time
    try:
        after = os.path.getmtime(LIB_PATH)
        try:
            hipcc = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, timeout=30).stdout.decode().splitlines()[0]
        except Exception:
            hipcc = None
        rec = {"mode": "linked a new library" if mtime_before is None or after > mtime_before else
                       "make: library newer than every source (nothing compiled)",
               "lib_sha16": hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()[:16],
               "lib_mtime": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(after)),
               "checked_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "hipcc": hipcc,
               "arch": "gfx950"}
        json.dump(rec, open(BUILD_RECORD, "w"), indent=1)
    except Exception:
        pass  # (a record, never a reason for a build to fail)


_lib = None


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("librelearn_hip.so is not built: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.rl_last_error.restype = C.c_char_p
        L.rl_last_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(code, eng=None):
    if code != OK:
        msg = lib().rl_last_error(eng)
        raise RelearnError(code, msg.decode() if msg else "")


def _ptr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


# Handles are released before the interpreter (and with it the HIP runtime) shuts down: children first, then
# engines.  A __del__ that runs during finalisation must not call into a runtime that is already gone.
_live = weakref.WeakSet()


def _register(obj):
    _live.add(obj)


@atexit.register
def _close_all():
    objs = list(_live)
    order = {"Dqn": -1, "Adam": 0, "Trajectory": 1, "CartPoleEnv": 2, "ChainEnv": 2, "Mlp": 3, "GruMlp": 3,
             "Engine": 4}
    for o in sorted(objs, key=lambda o: order.get(type(o).__name__, 2)):
        o.close()


class _Handle:
    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Engine(_Handle):
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().rl_engine_create(C.c_int32(device), C.byref(self.h)))
        _register(self)

    def close(self):
        if self.h:
            lib().rl_engine_destroy(self.h)
            self.h = C.c_void_p()

    def sync(self):
        _check(lib().rl_engine_sync(self.h), self.h)

    def info(self):
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(256)
        cus = C.c_int32()
        _check(lib().rl_engine_info(self.h, name, C.c_size_t(256), arch, C.c_size_t(256), C.byref(cus)), self.h)
        return name.value.decode(), arch.value.decode(), cus.value

    def stream_words(self, seed, stream, first_word, n_words):
        """raw 32-bit words of the engine's ChaCha8 stream (seed, stream), computed on the device (test hook)"""
        out = np.zeros(n_words, dtype=np.uint32)
        _check(lib().rl_debug_stream_words(self.h, C.c_uint64(seed), C.c_uint64(stream), C.c_uint64(first_word),
                                           C.c_uint32(n_words), out.ctypes.data_as(C.c_void_p)), self.h)
        return out

    def set_kernel_variant(self, variant):
        _check(lib().rl_engine_set_kernel_variant(self.h, C.c_int32(variant)), self.h)

    def set_serial_update(self, serial):
        """rl_actor_critic_update: True = the two chains one after the other on the main stream"""
        _check(lib().rl_engine_set_serial_update(self.h, C.c_int32(1 if serial else 0)), self.h)

    def timer_begin(self):
        _check(lib().rl_timer_begin(self.h), self.h)

    def timer_end(self):
        ms = C.c_float()
        _check(lib().rl_timer_end(self.h, C.byref(ms)), self.h)
        return ms.value

    def profile_enable(self, on=True):
        _check(lib().rl_profile_enable(self.h, C.c_int32(1 if on else 0)), self.h)

    def profile_read(self, reset=True):
        ms = (C.c_double * len(KERNEL_CLASSES))()
        cnt = (C.c_uint64 * len(KERNEL_CLASSES))()
        _check(lib().rl_profile_read(self.h, ms, cnt, C.c_int32(1 if reset else 0)), self.h)
        return {k: (ms[i], cnt[i]) for i, k in enumerate(KERNEL_CLASSES)}

    def comm_init(self, rank, n_ranks, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().rl_comm_init(self.h, C.c_int32(rank), C.c_int32(n_ranks), buf), self.h)

    def comm_ipc_handle(self, n_ranks):
        """this rank's mailbox for the peer-mailbox collective, as 64 bytes to hand to the other ranks"""
        buf = (C.c_uint8 * 64)()
        _check(lib().rl_comm_ipc_handle(self.h, C.c_int32(n_ranks), buf), self.h)
        return bytes(buf)

    def comm_init_ipc(self, rank, n_ranks, handles):
        """`handles`: the comm_ipc_handle() bytes of all ranks, in rank order"""
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == 64 * n_ranks
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(lib().rl_comm_init_ipc(self.h, C.c_int32(rank), C.c_int32(n_ranks), buf), self.h)

    def comm_selftest(self):
        _check(lib().rl_comm_selftest(self.h), self.h)

    def comm_destroy(self):
        _check(lib().rl_comm_destroy(self.h), self.h)
        self._host_allreduce = None

    def comm_init_host(self, rank, n_ranks, allreduce):
        """Host-staged collective (rl_comm_init_host): `allreduce(array)` must sum the float32 numpy array over all ranks
        in place, e.g. `lambda a: dist.all_reduce(torch.from_numpy(a))` on a gloo group."""
        def thunk(_ctx, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:  # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._host_allreduce = HOST_ALLREDUCE_FN(thunk)  # keep the callback object alive
        _check(lib().rl_comm_init_host(self.h, C.c_int32(rank), C.c_int32(n_ranks), self._host_allreduce, None),
               self.h)


def comm_available():
    """can RCCL be bound in this process?  (no GPU touched; agree on it across ranks before comm_init)"""
    return lib().rl_comm_available() == OK


def comm_library_paths():
    """(librccl bound by the library — '' before it is bound —, libamdhip64 the library runs on)"""
    a, b = C.create_string_buffer(1024), C.create_string_buffer(1024)
    _check(lib().rl_comm_library_paths(a, C.c_size_t(1024), b, C.c_size_t(1024)))
    return a.value.decode(), b.value.decode()


def comm_unique_id():
    buf = (C.c_uint8 * 128)()
    _check(lib().rl_comm_unique_id(buf))
    return bytes(buf)


def cartpole_params_default():
    p = CartPoleParams()
    _check(lib().rl_cartpole_params_default(C.byref(p)))
    return p


class CartPoleEnv(_Handle):
    """N CartPole lanes wrapped in a step limit — `CartPole::default().wrap(VisibleStepLimit::new(500))`."""

    def __init__(self, engine, n_lanes, max_steps=500, limit=LIMIT_VISIBLE, lane_offset=0, seed_env=0,
                 seed_actor=1, params=None):
        self.eng = engine
        cfg = EnvConfig()
        cfg.kind = ENV_CARTPOLE
        cfg.limit_kind = limit
        cfg.max_steps = max_steps
        cfg.n_lanes = n_lanes
        cfg.lane_offset = lane_offset
        cfg.seed_env = seed_env
        cfg.seed_actor = seed_actor
        cfg.cartpole = params if params is not None else cartpole_params_default()
        self.cfg = cfg
        self.h = C.c_void_p()
        _check(lib().rl_env_create(engine.h, C.byref(cfg), C.byref(self.h)), engine.h)
        _register(self)
        self.n = n_lanes
        d, a = C.c_uint32(), C.c_uint32()
        _check(lib().rl_env_dims(self.h, C.byref(d), C.byref(a)), engine.h)
        self.D, self.A = d.value, a.value

    def close(self):
        if self.h:
            lib().rl_env_destroy(self.h)
            self.h = C.c_void_p()

    def reset(self):
        _check(lib().rl_env_reset(self.h), self.eng.h)

    def observe(self):
        obs = np.zeros((self.D, self.n), dtype=np.float32)
        _check(lib().rl_env_observe(self.h, obs.ctypes.data_as(C.c_void_p)), self.eng.h)
        return obs

    def step(self, actions):
        a, ap = _ptr(actions, np.uint8)
        assert a.shape == (self.n,)
        reward = np.zeros(self.n, dtype=np.float32)
        flag = np.zeros(self.n, dtype=np.uint8)
        obs = np.zeros((self.D, self.n), dtype=np.float32)
        term = np.zeros((self.D, self.n), dtype=np.float32)
        _check(lib().rl_env_step(self.h, ap, reward.ctypes.data_as(C.c_void_p), flag.ctypes.data_as(C.c_void_p),
                                 obs.ctypes.data_as(C.c_void_p), term.ctypes.data_as(C.c_void_p)), self.eng.h)
        return reward, flag, obs, term

    def upload_actions(self, actions):
        a, ap = _ptr(actions, np.uint8)
        _check(lib().rl_env_upload_actions(self.h, ap), self.eng.h)

    def step_resident(self):
        _check(lib().rl_env_step_resident(self.h), self.eng.h)

    def get_state(self):
        st = np.zeros((4, self.n), dtype=np.float64)
        nv = np.zeros(self.n, dtype=np.int32)
        rem = np.zeros(self.n, dtype=np.uint64)
        rc = np.zeros(self.n, dtype=np.uint64)
        _check(lib().rl_env_get_state(self.h, st.ctypes.data_as(C.c_void_p), nv.ctypes.data_as(C.c_void_p),
                                      rem.ctypes.data_as(C.c_void_p), rc.ctypes.data_as(C.c_void_p)), self.eng.h)
        return st, nv, rem, rc

    def set_state(self, st, nv, rem, rc):
        st, sp = _ptr(st, np.float64)
        nv, np_ = _ptr(nv, np.int32)
        rem, rp = _ptr(rem, np.uint64)
        rc, cp = _ptr(rc, np.uint64)
        _check(lib().rl_env_set_state(self.h, sp, np_, rp, cp), self.eng.h)


class ChainEnv(_Handle):
    """N Chain lanes wrapped in a step limit — `Chain::default().wrap(LatentStepLimit::new(100))`."""

    def __init__(self, engine, n_lanes, max_steps=100, limit=LIMIT_LATENT, lane_offset=0, seed_env=0, seed_actor=1):
        self.eng = engine
        cfg = EnvConfig()
        cfg.kind = ENV_CHAIN
        cfg.limit_kind = limit
        cfg.max_steps = max_steps
        cfg.n_lanes = n_lanes
        cfg.lane_offset = lane_offset
        cfg.seed_env = seed_env
        cfg.seed_actor = seed_actor
        cfg.cartpole = cartpole_params_default()
        cfg.chain_size = 5
        self.cfg = cfg
        self.h = C.c_void_p()
        _check(lib().rl_env_create(engine.h, C.byref(cfg), C.byref(self.h)), engine.h)
        _register(self)
        self.n = n_lanes
        d, a = C.c_uint32(), C.c_uint32()
        _check(lib().rl_env_dims(self.h, C.byref(d), C.byref(a)), engine.h)
        self.D, self.A = d.value, a.value

    close = CartPoleEnv.close
    reset = CartPoleEnv.reset
    observe = CartPoleEnv.observe
    step = CartPoleEnv.step

    def get_state(self):
        """(state index, steps_remaining, reset_count) per lane"""
        st, _, rem, rc = CartPoleEnv.get_state(self)
        return st[0].astype(np.uint64), rem, rc


class BanditEnv(ChainEnv):
    """N DeterministicBandit lanes (src/envs/bandits.rs:109-116) — `DeterministicBandit::from_values([v0, v1])`: every step
    is a whole episode, the reward is the chosen arm's value.  Observations: one-hot(5) of the single state."""

    def __init__(self, engine, n_lanes, values=(0.0, 1.0), lane_offset=0, seed_env=0, seed_actor=1):
        self.eng = engine
        cfg = EnvConfig()
        cfg.kind = ENV_BANDIT
        cfg.limit_kind = LIMIT_NONE
        cfg.n_lanes = n_lanes
        cfg.lane_offset = lane_offset
        cfg.seed_env = seed_env
        cfg.seed_actor = seed_actor
        cfg.cartpole = cartpole_params_default()
        cfg.bandit_values[0], cfg.bandit_values[1] = values
        self.cfg = cfg
        self.h = C.c_void_p()
        _check(lib().rl_env_create(engine.h, C.byref(cfg), C.byref(self.h)), engine.h)
        _register(self)
        self.n = n_lanes
        d, a = C.c_uint32(), C.c_uint32()
        _check(lib().rl_env_dims(self.h, C.byref(d), C.byref(a)), engine.h)
        self.D, self.A = d.value, a.value


class MemoryEnv(ChainEnv):
    """N MemoryGame lanes (src/envs/memory.rs) — `MemoryGame::new(2, 3)`, optionally wrapped in a step limit: the lane
    starts in state 0 or 1 at random, walks through states 2, 3, 4 whatever the action, and on the step from state 4
    earns +1 iff the action equals the state it started in (else -1); that step terminates the episode."""

    def __init__(self, engine, n_lanes, num_actions=2, history_len=3, max_steps=0, limit=LIMIT_NONE, lane_offset=0,
                 seed_env=0, seed_actor=1):
        self.eng = engine
        cfg = EnvConfig()
        cfg.kind = ENV_MEMORY
        cfg.limit_kind = limit
        cfg.max_steps = max_steps
        cfg.n_lanes = n_lanes
        cfg.lane_offset = lane_offset
        cfg.seed_env = seed_env
        cfg.seed_actor = seed_actor
        cfg.cartpole = cartpole_params_default()
        cfg.chain_size = 0
        cfg.memory_num_actions = num_actions
        cfg.memory_history_len = history_len
        self.cfg = cfg
        self.h = C.c_void_p()
        _check(lib().rl_env_create(engine.h, C.byref(cfg), C.byref(self.h)), engine.h)
        _register(self)
        self.n = n_lanes
        d, a = C.c_uint32(), C.c_uint32()
        _check(lib().rl_env_dims(self.h, C.byref(d), C.byref(a)), engine.h)
        self.D, self.A = d.value, a.value

    def get_state(self):
        """(current state, initial state, env-stream word position, steps_remaining, reset_count) per lane"""
        st, _, rem, rc = CartPoleEnv.get_state(self)
        return st[0].astype(np.uint64), st[1].astype(np.uint64), st[2].astype(np.uint64), rem, rc


class GruMlp(_Handle):
    """`GruMlpConfig::default().build_module(in, out)`: GRU(128) -> ReLU -> MLP([128]); flat reference order."""

    CELL = 0  # RL_CELL_GRU

    def __init__(self, engine, in_dim, out_dim, gru_hidden=128, mlp_hidden=128, num_layers=1, rnn_bias=True):
        """num_layers: RnnBaseConfig::num_layers (stacked layers, 1..4); rnn_bias False: RnnBaseConfig::bias_init = None
        (recurrent layers without bias vectors)"""
        self.eng = engine
        self.h = C.c_void_p()
        _check(lib().rl_rnn_mlp_create_config(engine.h, C.c_int32(self.CELL), C.c_uint32(in_dim), C.c_uint32(gru_hidden),
                                              C.c_uint32(num_layers), C.c_int32(1 if rnn_bias else 0),
                                              C.c_uint32(mlp_hidden), C.c_uint32(out_dim), C.byref(self.h)), engine.h)
        _register(self)
        n = C.c_uint64()
        _check(lib().rl_mlp_num_params(self.h, C.byref(n)), engine.h)
        self.P = n.value
        self.in_dim, self.hidden, self.out_dim, self.gru_hidden = in_dim, mlp_hidden, out_dim, gru_hidden
        self.num_layers = num_layers

    close = None  # set below
    init = None
    get_params = None
    set_params = None

    def init_with(self, seed, input_weights=("Uniform", "FanAvg", 0.0), hidden_weights=("Orthogonal", "FanAvg", 0.0),
                  bias=("Zeros", "FanAvg", 0.0), mlp_kernel=("Uniform", "FanAvg", 0.0), mlp_bias=("Uniform", "FanAvg", 0.0)):
        """RnnBaseConfig's initializers and the chain MLP's LinearConfig as (kind, scale, value) triples"""
        spec = [Initializer.of(x) if x is not None else None for x in (input_weights, hidden_weights, bias, mlp_kernel, mlp_bias)]
        _check(lib().rl_rnn_mlp_init_with(self.h, C.c_uint64(seed), *[C.byref(x) if x is not None else None for x in spec]),
               self.eng.h)

    def seq_forward(self, traj, want_succ=True):
        out = np.zeros((self.out_dim, traj.T, traj.n), dtype=np.float32)
        succ = np.zeros_like(out) if want_succ else None
        _check(lib().rl_seq_forward(self.h, traj.h, out.ctypes.data_as(C.c_void_p),
                                    succ.ctypes.data_as(C.c_void_p) if want_succ else None), self.eng.h)
        return out, succ


class LstmMlp(GruMlp):
    """`ChainConfig<LstmConfig, MlpConfig>::default().build_module(in, out)`: LSTM(128) -> ReLU -> MLP([128])."""

    CELL = 1  # RL_CELL_LSTM

    def __init__(self, engine, in_dim, out_dim, lstm_hidden=128, mlp_hidden=128, num_layers=1, rnn_bias=True):
        GruMlp.__init__(self, engine, in_dim, out_dim, lstm_hidden, mlp_hidden, num_layers, rnn_bias)


ACTIVATIONS = ["Identity", "Relu", "Sigmoid", "Tanh"]  # rl_activation, in the reference enum's order
INIT_KINDS = ["Zeros", "Constant", "Uniform", "Normal", "Orthogonal"]  # rl_init_kind
VARIANCE_SCALES = ["Constant", "FanIn", "FanOut", "FanAvg"]             # rl_variance_scale


class Initializer(C.Structure):
    """rl_initializer"""
    _fields_ = [("kind", C.c_int32), ("scale", C.c_int32), ("value", C.c_double)]

    @staticmethod
    def of(spec):
        kind, scale, value = spec
        return Initializer(INIT_KINDS.index(kind), VARIANCE_SCALES.index(scale), float(value))


class Mlp(_Handle):
    """`MlpConfig{hidden_sizes:[H], activation: Relu}.build_module(in, out)`; flat params in reference order."""

    def __init__(self, engine, in_dim, hidden, out_dim, activation="Relu", output_activation="Identity", bias=True):
        """`hidden`: the width of the single hidden layer, or MlpConfig's `hidden_sizes` as a list (any number of layers:
        shapes other than one layer of <= 128 units, and activations other than Relu / Identity, run the general
        per-layer kernels); activations by the reference's variant names (ff/activation.rs:11-20)"""
        self.eng = engine
        self.h = C.c_void_p()
        self.activation, self.output_activation, self.bias = activation, output_activation, bias
        if isinstance(hidden, (list, tuple)) or (activation, output_activation) != ("Relu", "Identity") or not bias:
            hidden = list(hidden) if isinstance(hidden, (list, tuple)) else [hidden]
            sizes = (C.c_uint32 * max(len(hidden), 1))(*hidden)
            # bias = False: LinearConfig::bias_init = None, layers without a bias vector (rl_mlp_create_config)
            _check(lib().rl_mlp_create_config(engine.h, C.c_uint32(in_dim), sizes, C.c_uint32(len(hidden)),
                                              C.c_uint32(out_dim), C.c_int32(ACTIVATIONS.index(activation)),
                                              C.c_int32(ACTIVATIONS.index(output_activation)),
                                              C.c_int32(1 if bias else 0), C.byref(self.h)),
                   engine.h)
        else:
            _check(lib().rl_mlp_create(engine.h, C.c_uint32(in_dim), C.c_uint32(hidden), C.c_uint32(out_dim),
                                       C.byref(self.h)), engine.h)
        _register(self)
        n = C.c_uint64()
        _check(lib().rl_mlp_num_params(self.h, C.byref(n)), engine.h)
        self.P = n.value
        self.in_dim, self.hidden, self.out_dim = in_dim, hidden, out_dim

    def close(self):
        if self.h:
            lib().rl_mlp_destroy(self.h)
            self.h = C.c_void_p()

    def init(self, seed, kernel_init=None, bias_init=None):
        """Linear::new for every layer; `kernel_init` / `bias_init`: (kind, scale, value) with the reference's variant
        names, e.g. ("Normal", "FanIn", 0.0), ("Constant", "Constant", 0.1), ("Orthogonal", "FanIn", 0.0); default
        both Uniform(FanAvg)"""
        if kernel_init is None and bias_init is None:
            _check(lib().rl_mlp_init(self.h, C.c_uint64(seed)), self.eng.h)
            return
        default = ("Uniform", "FanAvg", 0.0)
        k = Initializer.of(kernel_init or default)
        if not getattr(self, "bias", True):  # LinearConfig::bias_init = None
            assert bias_init is None, "a module without bias vectors takes no bias initializer"
            _check(lib().rl_mlp_init_with(self.h, C.c_uint64(seed), C.byref(k), None), self.eng.h)
            return
        b = Initializer.of(bias_init or default)
        _check(lib().rl_mlp_init_with(self.h, C.c_uint64(seed), C.byref(k), C.byref(b)), self.eng.h)

    def get_params(self):
        p = np.zeros(self.P, dtype=np.float32)
        _check(lib().rl_params_get(self.h, p.ctypes.data_as(C.c_void_p), C.c_uint64(self.P)), self.eng.h)
        return p

    def set_params(self, p):
        p, pp = _ptr(p, np.float32)
        _check(lib().rl_params_set(self.h, pp, C.c_uint64(p.size)), self.eng.h)

    def forward(self, rows):
        rows, rp = _ptr(rows, np.float32)
        out = np.zeros((rows.shape[0], self.out_dim), dtype=np.float32)
        _check(lib().rl_mlp_forward(self.h, rp, C.c_uint64(rows.shape[0]), out.ctypes.data_as(C.c_void_p)),
               self.eng.h)
        return out


GruMlp.close = Mlp.close
GruMlp.init = Mlp.init
GruMlp.get_params = Mlp.get_params
GruMlp.set_params = Mlp.set_params

_TRAJ_DTYPES = {TRAJ_OBS: np.float32, TRAJ_ACTION: np.uint8, TRAJ_REWARD: np.float32, TRAJ_FLAG: np.uint8,
                TRAJ_TERM_OBS: np.float32, TRAJ_VALUES: np.float32, TRAJ_ADVANTAGES: np.float32,
                TRAJ_RETURNS: np.float32, TRAJ_TARGETS: np.float32}


class Trajectory(_Handle):
    def __init__(self, engine, n_lanes, horizon, obs_dim):
        self.eng = engine
        self.h = C.c_void_p()
        _check(lib().rl_traj_create(engine.h, C.c_uint64(n_lanes), C.c_uint64(horizon), C.c_uint32(obs_dim),
                                    C.byref(self.h)), engine.h)
        _register(self)
        self.n, self.T, self.D = n_lanes, horizon, obs_dim

    def close(self):
        if self.h:
            lib().rl_traj_destroy(self.h)
            self.h = C.c_void_p()

    def shape(self, field):
        n, T, D = self.n, self.T, self.D
        return {TRAJ_OBS: (D, T + 1, n), TRAJ_ACTION: (T, n), TRAJ_REWARD: (T, n), TRAJ_FLAG: (T, n),
                TRAJ_TERM_OBS: (D, T, n), TRAJ_VALUES: (T + 1, n), TRAJ_ADVANTAGES: (T, n),
                TRAJ_RETURNS: (T, n), TRAJ_TARGETS: (T, n)}[field]

    def read(self, field):
        out = np.zeros(self.shape(field), dtype=_TRAJ_DTYPES[field])
        _check(lib().rl_traj_read(self.h, C.c_int32(field), out.ctypes.data_as(C.c_void_p),
                                  C.c_uint64(out.nbytes)), self.eng.h)
        return out

    def write(self, field, arr):
        arr, ap = _ptr(arr, _TRAJ_DTYPES[field])
        assert arr.shape == self.shape(field), (arr.shape, self.shape(field))
        _check(lib().rl_traj_write(self.h, C.c_int32(field), ap, C.c_uint64(arr.nbytes)), self.eng.h)

    def read_all(self):
        return dict(obs=self.read(TRAJ_OBS), action=self.read(TRAJ_ACTION), reward=self.read(TRAJ_REWARD),
                    flag=self.read(TRAJ_FLAG), term_obs=self.read(TRAJ_TERM_OBS))

    def write_all(self, traj):
        self.write(TRAJ_OBS, traj["obs"])
        self.write(TRAJ_ACTION, traj["action"])
        self.write(TRAJ_REWARD, traj["reward"])
        self.write(TRAJ_FLAG, traj["flag"])
        self.write(TRAJ_TERM_OBS, traj["term_obs"])


def rollout(env, policy, traj):
    _check(lib().rl_rollout(env.h, policy.h, traj.h), env.eng.h)


def gae(traj, critic, gamma, lam):
    _check(lib().rl_gae(traj.h, critic.h, C.c_float(gamma), C.c_float(lam)), traj.eng.h)


def trpo_config_default():
    c = TrpoConfig()
    _check(lib().rl_trpo_config_default(C.byref(c)))
    return c


def trpo_update(policy, traj, cfg=None):
    cfg = cfg if cfg is not None else trpo_config_default()
    st = TrpoStats()
    _check(lib().rl_trpo_update(policy.h, traj.h, C.byref(cfg), C.byref(st)), traj.eng.h)
    return st


def policy_gradient(policy, traj):
    g = np.zeros(policy.P, dtype=np.float32)
    loss, ent = C.c_float(), C.c_float()
    _check(lib().rl_policy_gradient(policy.h, traj.h, g.ctypes.data_as(C.c_void_p), C.byref(loss), C.byref(ent)),
           traj.eng.h)
    return g, loss.value, ent.value


def policy_fvp(policy, traj, v, reg):
    v, vp = _ptr(v, np.float32)
    out = np.zeros(policy.P, dtype=np.float32)
    _check(lib().rl_policy_fvp(policy.h, traj.h, vp, C.c_float(reg), out.ctypes.data_as(C.c_void_p)), traj.eng.h)
    return out


def policy_loss_kl(policy, traj, params0):
    p0, pp = _ptr(params0, np.float32)
    loss, kl = C.c_float(), C.c_float()
    _check(lib().rl_policy_loss_kl(policy.h, traj.h, pp, C.byref(loss), C.byref(kl)), traj.eng.h)
    return loss.value, kl.value


def adam_config_default():
    c = AdamConfig()
    _check(lib().rl_adam_config_default(C.byref(c)))
    return c


class Adam(_Handle):
    def __init__(self, module, cfg=None):
        self.mod = module
        self.cfg = cfg if cfg is not None else adam_config_default()
        self.h = C.c_void_p()
        _check(lib().rl_adam_create(module.h, C.byref(self.cfg), C.byref(self.h)), module.eng.h)
        _register(self)

    def close(self):
        if self.h:
            lib().rl_adam_destroy(self.h)
            self.h = C.c_void_p()

    def step_host(self, grad):
        g, gp = _ptr(grad, np.float32)
        _check(lib().rl_adam_step_host(self.h, gp), self.mod.eng.h)


def critic_update(critic, opt, traj, opt_steps=80, want_losses=False):
    st = CriticStats()
    losses = np.zeros(max(opt_steps, 1), dtype=np.float32)
    _check(lib().rl_critic_update(critic.h, opt.h, traj.h, C.c_uint64(opt_steps), C.byref(st),
                                  losses.ctypes.data_as(C.c_void_p) if want_losses else None), traj.eng.h)
    return (st, losses[:opt_steps]) if want_losses else st


class ValuesOptConfig(C.Structure):
    _fields_ = [("opt_steps_per_update", C.c_uint64), ("target", C.c_int32), ("discount_factor", C.c_float)]


def values_opt_config_default():
    c = ValuesOptConfig()
    _check(lib().rl_values_opt_config_default(C.byref(c)))
    return c


def values_opt_update(critic, opt, traj, cfg=None, want_losses=False):
    """ValuesOpt::update (critics/opt.rs:100-126) with its target selection (StepValueTarget)"""
    cfg = cfg if cfg is not None else values_opt_config_default()
    st = CriticStats()
    K = cfg.opt_steps_per_update
    losses = np.zeros(max(K, 1), dtype=np.float32)
    _check(lib().rl_values_opt_update(critic.h, opt.h, traj.h, C.byref(cfg), C.byref(st),
                                      losses.ctypes.data_as(C.c_void_p) if want_losses else None), traj.eng.h)
    return (st, losses[:K]) if want_losses else st


def actor_critic_update(policy, critic, critic_opt, traj, trpo_cfg=None, critic_cfg=None, want_losses=False):
    """policy.update + critic.update of ActorCriticAgent::batch_update_slice (actor_critic.rs:196-208) as one call:
    the TRPO chain and the critic chain side by side on two streams (rl_actor_critic_update)"""
    trpo_cfg = trpo_cfg if trpo_cfg is not None else trpo_config_default()
    critic_cfg = critic_cfg if critic_cfg is not None else values_opt_config_default()
    pst, cst = TrpoStats(), CriticStats()
    K = critic_cfg.opt_steps_per_update
    losses = np.zeros(max(K, 1), dtype=np.float32)
    _check(lib().rl_actor_critic_update(policy.h, critic.h, critic_opt.h, traj.h, C.byref(trpo_cfg), C.byref(critic_cfg),
                                        C.byref(pst), C.byref(cst),
                                        losses.ctypes.data_as(C.c_void_p) if want_losses else None), traj.eng.h)
    return (pst, cst, losses[:K]) if want_losses else (pst, cst)


def actor_critic_update_begin(policy, critic, critic_opt, traj, trpo_cfg=None, critic_cfg=None):
    """first half of actor_critic_update: returns the TRPO statistics with the critic chain possibly still in flight on
    the auxiliary stream (a rollout into ANOTHER trajectory may run beside it); actor_critic_update_finish ends it"""
    trpo_cfg = trpo_cfg if trpo_cfg is not None else trpo_config_default()
    critic_cfg = critic_cfg if critic_cfg is not None else values_opt_config_default()
    pst = TrpoStats()
    _check(lib().rl_actor_critic_update_begin(policy.h, critic.h, critic_opt.h, traj.h, C.byref(trpo_cfg),
                                              C.byref(critic_cfg), C.byref(pst)), traj.eng.h)
    traj._pending_steps = int(critic_cfg.opt_steps_per_update)
    return pst


def actor_critic_update_finish(traj, want_losses=False):
    cst = CriticStats()
    K = getattr(traj, "_pending_steps", 0)
    losses = np.zeros(max(K, 1), dtype=np.float32)
    _check(lib().rl_actor_critic_update_finish(traj.h, C.byref(cst),
                                               losses.ctypes.data_as(C.c_void_p) if want_losses else None), traj.eng.h)
    return (cst, losses[:K]) if want_losses else cst


def critic_gradient(critic, traj):
    g = np.zeros(critic.P, dtype=np.float32)
    loss = C.c_float()
    _check(lib().rl_critic_gradient(critic.h, traj.h, g.ctypes.data_as(C.c_void_p), C.byref(loss)), traj.eng.h)
    return g, loss.value


DQN_TARGET_REWARD_TO_GO, DQN_TARGET_ONE_STEP_TD = 0, 1
SCHEDULE_CONSTANT, SCHEDULE_LINEAR_ANNEALED = 0, 1
COLLECT_CONSTANT, COLLECT_FIRST_REST = 0, 1
(REPLAY_HEAD, REPLAY_COUNT, REPLAY_EP_HEAD, REPLAY_EP_COUNT, REPLAY_TOTAL, REPLAY_EP_END, REPLAY_OBS,
 REPLAY_NEXT_OBS, REPLAY_ACTION, REPLAY_REWARD, REPLAY_FLAG, REPLAY_ACTOR_POS, REPLAY_LAST_FLAGS) = range(13)
MB_EP_LANE, MB_EP_START, MB_EP_LEN, MB_EP_OFFSET, MB_OBS, MB_ACTION, MB_TARGET = range(7)


class DqnConfig(C.Structure):
    _fields_ = [("target", C.c_int32), ("exploration_kind", C.c_int32), ("exploration_start", C.c_double),
                ("exploration_end", C.c_double), ("exploration_period", C.c_uint64),
                ("minibatch_steps", C.c_uint64), ("opt_steps_per_update", C.c_uint64),
                ("buffer_capacity", C.c_uint64), ("episode_capacity", C.c_uint64), ("update_kind", C.c_int32),
                ("update_first", C.c_uint64), ("update_rest", C.c_uint64), ("discount_factor", C.c_float),
                ("agent_key", C.c_uint32 * 8)]


class DqnCollectStats(C.Structure):
    _fields_ = [("exploration_rate", C.c_double), ("steps", C.c_uint64), ("episodes_ended", C.c_uint64)]


class DqnUpdateStats(C.Structure):
    _fields_ = [("loss_first", C.c_double), ("loss_last", C.c_double), ("opt_steps", C.c_uint64),
                ("global_steps", C.c_uint64), ("last_minibatch_steps", C.c_uint64),
                ("last_minibatch_episodes", C.c_uint64)]


def dqn_config_default():
    c = DqnConfig()
    _check(lib().rl_dqn_config_default(C.byref(c)))
    return c


class Dqn(_Handle):
    """`DqnConfig::build_agent` + the replay store of all lanes, resident in HBM (src/torch/agents/dqn.rs)."""

    def __init__(self, env, qnet, opt, cfg):
        self.eng, self.env, self.qnet, self.opt, self.cfg = env.eng, env, qnet, opt, cfg
        self.h = C.c_void_p()
        _check(lib().rl_dqn_create(env.h, qnet.h, opt.h, C.byref(cfg), C.byref(self.h)), env.eng.h)
        _register(self)
        self.n, self.D = env.n, env.D
        self.C = cfg.buffer_capacity
        self.E = cfg.episode_capacity or cfg.buffer_capacity
        self.n_eps = self.n_steps = 0

    def close(self):
        if self.h:
            lib().rl_dqn_destroy(self.h)
            self.h = C.c_void_p()

    def exploration_rate(self, training=True):
        r = C.c_double()
        _check(lib().rl_dqn_exploration_rate(self.h, C.c_int32(1 if training else 0), C.byref(r)), self.eng.h)
        return r.value

    def min_update_size(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().rl_dqn_min_update_size(self.h, C.byref(a), C.byref(b)), self.eng.h)
        return a.value, b.value

    def collect(self, horizon, want_stats=True):
        st = DqnCollectStats()
        _check(lib().rl_dqn_collect(self.h, C.c_uint64(horizon), C.byref(st) if want_stats else None), self.eng.h)
        self.last_horizon = horizon
        return st

    def update(self, want_losses=False):
        st = DqnUpdateStats()
        K = self.cfg.opt_steps_per_update
        losses = np.zeros(max(K, 1), dtype=np.float32)
        _check(lib().rl_dqn_update(self.h, C.byref(st), losses.ctypes.data_as(C.c_void_p) if want_losses else None),
               self.eng.h)
        if K:  # the update's last minibatch is the one minibatch_read / minibatch_gradient see now
            self.n_eps, self.n_steps = int(st.last_minibatch_episodes), int(st.last_minibatch_steps)
        return (st, losses[:K]) if want_losses else st

    def replay_read(self, field):
        n, Cc, E, D = self.n, self.C, self.E, self.D
        shape, dt = {
            REPLAY_HEAD: ((n,), np.uint32), REPLAY_COUNT: ((n,), np.uint32), REPLAY_EP_HEAD: ((n,), np.uint32),
            REPLAY_EP_COUNT: ((n,), np.uint32), REPLAY_TOTAL: ((n,), np.uint32), REPLAY_EP_END: ((E, n), np.uint32),
            REPLAY_OBS: ((D, Cc, n), np.float32), REPLAY_NEXT_OBS: ((D, Cc, n), np.float32),
            REPLAY_ACTION: ((Cc, n), np.uint8), REPLAY_REWARD: ((Cc, n), np.float32),
            REPLAY_FLAG: ((Cc, n), np.uint8), REPLAY_ACTOR_POS: ((n,), np.uint64),
            REPLAY_LAST_FLAGS: ((getattr(self, "last_horizon", 0), n), np.uint8)}[field]
        out = np.zeros(shape, dtype=dt)
        _check(lib().rl_dqn_replay_read(self.h, C.c_int32(field), out.ctypes.data_as(C.c_void_p),
                                        C.c_uint64(out.nbytes)), self.eng.h)
        return out

    def minibatch_sample(self, sequential=False):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().rl_dqn_minibatch_sample(self.h, C.c_int32(1 if sequential else 0), C.byref(a), C.byref(b)),
               self.eng.h)
        self.n_eps, self.n_steps = a.value, b.value
        return a.value, b.value

    def minibatch_read(self, field):
        ne, ns, D = self.n_eps, self.n_steps, self.D
        shape, dt = {MB_EP_LANE: ((ne,), np.uint32), MB_EP_START: ((ne,), np.uint32), MB_EP_LEN: ((ne,), np.uint32),
                     MB_EP_OFFSET: ((ne,), np.uint32), MB_OBS: ((D, ns), np.float32), MB_ACTION: ((ns,), np.uint8),
                     MB_TARGET: ((ns,), np.float32)}[field]
        out = np.zeros(shape, dtype=dt)
        _check(lib().rl_dqn_minibatch_read(self.h, C.c_int32(field), out.ctypes.data_as(C.c_void_p),
                                           C.c_uint64(out.nbytes)), self.eng.h)
        return out

    def minibatch_gradient(self):
        g = np.zeros(self.qnet.P, dtype=np.float32)
        loss = C.c_float()
        _check(lib().rl_dqn_minibatch_gradient(self.h, g.ctypes.data_as(C.c_void_p), C.byref(loss)), self.eng.h)
        return g, loss.value

    def agent_rng_pos(self):
        p = C.c_uint64()
        _check(lib().rl_dqn_agent_rng_pos(self.h, C.byref(p)), self.eng.h)
        return p.value


class PpoConfig(C.Structure):
    _fields_ = [("opt_steps_per_update", C.c_uint64), ("clip_distance", C.c_double)]


class PolicyOptStats(C.Structure):
    _fields_ = [("entropy", C.c_double), ("loss_first", C.c_double), ("loss_last", C.c_double),
                ("steps", C.c_uint64)]


def ppo_config_default():
    c = PpoConfig()
    _check(lib().rl_ppo_config_default(C.byref(c)))
    return c


def ppo_update(policy, opt, traj, cfg=None, want_losses=False):
    cfg = cfg if cfg is not None else ppo_config_default()
    st = PolicyOptStats()
    K = cfg.opt_steps_per_update
    losses = np.zeros(max(K, 1), dtype=np.float32)
    _check(lib().rl_ppo_update(policy.h, opt.h, traj.h, C.byref(cfg), C.byref(st),
                               losses.ctypes.data_as(C.c_void_p) if want_losses else None), traj.eng.h)
    return (st, losses[:K]) if want_losses else st


def reinforce_update(policy, opt, traj):
    st = PolicyOptStats()
    _check(lib().rl_reinforce_update(policy.h, opt.h, traj.h, C.byref(st)), traj.eng.h)
    return st


def reward_to_go(traj, gamma):
    _check(lib().rl_reward_to_go(traj.h, C.c_float(gamma)), traj.eng.h)


ACTOR_POLICY, ACTOR_DQN = 0, 1


def actor_to_cbor(env, module, actor_kind=ACTOR_POLICY, exploration_rate=0.0):
    """the bytes `serde_cbor::to_writer(file, &agent.actor(ActorMode::Evaluation))` writes for this env / module"""
    n = C.c_uint64()
    _check(lib().rl_actor_to_cbor(env.h, module.h, C.c_int32(actor_kind), C.c_double(exploration_rate), None,
                                  C.c_uint64(0), C.byref(n)), env.eng.h)
    buf = (C.c_uint8 * n.value)()
    _check(lib().rl_actor_to_cbor(env.h, module.h, C.c_int32(actor_kind), C.c_double(exploration_rate), buf,
                                  C.c_uint64(n.value), C.byref(n)), env.eng.h)
    return bytes(buf)


def module_from_cbor(module, data):
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    _check(lib().rl_module_from_cbor(module.h, buf, C.c_uint64(len(data))), module.eng.h)


# KindDef variants in declaration order (src/torch/serialize.rs:12-31) and the numpy dtype carrying their bytes
TENSOR_KINDS = ["Uint8", "Int8", "Int16", "Int", "Int64", "Half", "Float", "Double", "ComplexHalf", "ComplexFloat",
                "ComplexDouble", "Bool", "QInt8", "QUInt8", "QInt32", "BFloat16"]


def tensor_def_to_cbor(kind, shape, requires_grad, data):
    """TensorDef::from(&tensor) serialised by serde_cbor (host-only): `kind` a KindDef variant name, `data` the raw
    little-endian element bytes in row-major order"""
    data = bytes(data)
    shp = (C.c_int64 * max(len(shape), 1))(*shape)
    args = (C.c_int32(TENSOR_KINDS.index(kind)), shp, C.c_uint32(len(shape)), C.c_int32(1 if requires_grad else 0),
            data, C.c_uint64(len(data)))
    n = C.c_uint64()
    _check(lib().rl_tensor_def_to_cbor(*args, None, C.c_uint64(0), C.byref(n)))
    buf = (C.c_uint8 * max(n.value, 1))()
    _check(lib().rl_tensor_def_to_cbor(*args, buf, C.c_uint64(n.value), C.byref(n)))
    return bytes(buf[:n.value])


def tensor_def_from_cbor(doc):
    """-> (kind name, shape list, requires_grad, data bytes); raises RelearnError for anything serde would refuse"""
    doc = bytes(doc)
    kind, rank, rg, nbytes = C.c_int32(), C.c_uint32(), C.c_int32(), C.c_uint64()
    _check(lib().rl_tensor_def_from_cbor(doc, C.c_uint64(len(doc)), C.byref(kind), None, C.c_uint32(0), C.byref(rank),
                                         C.byref(rg), None, C.c_uint64(0), C.byref(nbytes)))
    shape = (C.c_int64 * max(rank.value, 1))()
    data = (C.c_uint8 * max(nbytes.value, 1))()
    _check(lib().rl_tensor_def_from_cbor(doc, C.c_uint64(len(doc)), C.byref(kind), shape, C.c_uint32(rank.value),
                                         C.byref(rank), C.byref(rg), data, C.c_uint64(nbytes.value), C.byref(nbytes)))
    return TENSOR_KINDS[kind.value], list(shape[:rank.value]), bool(rg.value), bytes(data[:nbytes.value])


def indexed_type_space_to_cbor():
    n = C.c_uint64()
    buf = (C.c_uint8 * 16)()
    _check(lib().rl_indexed_type_space_to_cbor(buf, C.c_uint64(16), C.byref(n)))
    return bytes(buf[:n.value])


def chain_tabular_q_train(seed=0, n_threads=4, n_periods=10, min_worker_steps=10000, exploration_rate=0.2):
    """examples/chain-tabular-q.rs on the host (CPU-only configuration, like the reference)."""
    q = np.zeros((5, 2), dtype=np.float64)
    counts = np.zeros((5, 2), dtype=np.uint64)
    total = C.c_uint64()
    _check(lib().rl_chain_tabular_q_train(C.c_uint64(seed), C.c_uint64(n_threads), C.c_uint64(n_periods),
                                          C.c_uint64(min_worker_steps), C.c_double(exploration_rate),
                                          q.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                          C.byref(total)))
    return q, counts, total.value


def chain_tabular_q_eval(q, seed=0, n_steps=10000):
    q, qp = _ptr(q, np.float64)
    actions = np.zeros(n_steps, dtype=np.uint8)
    total = C.c_double()
    _check(lib().rl_chain_tabular_q_eval(qp, C.c_uint64(seed), C.c_uint64(n_steps),
                                         actions.ctypes.data_as(C.c_void_p), C.byref(total)))
    return actions, total.value
