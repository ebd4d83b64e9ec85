"""ctypes loader for the CPU parity oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``relearn_amd/`` may import this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

CONTINUE, TERMINATE, INTERRUPT = 0, 1, 2
LIMIT_NONE, LIMIT_LATENT, LIMIT_VISIBLE = 0, 1, 2
OPT_OK, OPT_LOSS_NOT_IMPROVING, OPT_CONSTRAINT_VIOLATED, OPT_NAN_LOSS, OPT_NAN_CONSTRAINT = range(5)


def build(force=False, check=False):
    """Compile the oracle with gcc (seconds).  `check`: go through make even when the library exists."""
    if force or check or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class Prng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("counter", C.c_uint64), ("stream", C.c_uint64),
                ("buf", C.c_uint32 * 64), ("index", C.c_uint32)]


class CartPole(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "gravity", "mass_cart", "mass_pole", "length_half_pole", "friction_cart", "friction_pole", "time_step",
        "action_force", "max_pos", "max_angle", "discount_factor",
        "total_weight", "inv_total_mass", "mass_length_pole")] + [("use_libm", C.c_int)]


class CartPoleState(C.Structure):
    _fields_ = [("x", C.c_double), ("xdot", C.c_double), ("th", C.c_double), ("thdot", C.c_double),
                ("nv_pos", C.c_int32)]


class Chain(C.Structure):
    _fields_ = [("size", C.c_uint64), ("discount_factor", C.c_double)]


class Bound(C.Structure):
    _fields_ = [("min_steps", C.c_uint64), ("slack_steps", C.c_uint64)]


class MlpShape(C.Structure):
    _fields_ = [("in_dim", C.c_uint32), ("hidden", C.c_uint32), ("out_dim", C.c_uint32)]


class TrpoCfg(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("max_backtracks", C.c_uint64), ("backtrack_ratio", C.c_double),
                ("hpv_reg_coeff", C.c_double), ("max_kl", C.c_double), ("accept_violation", C.c_int)]


class TrpoStats(C.Structure):
    _fields_ = [("entropy", C.c_double), ("step_size", C.c_double), ("loss_initial", C.c_double),
                ("loss_final", C.c_double), ("constraint_val_final", C.c_double), ("step_scale", C.c_double),
                ("num_backtracks", C.c_int64), ("status", C.c_int32), ("cg_iterations", C.c_int32)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double)]


class AdamState(C.Structure):
    _fields_ = [("step", C.c_uint64), ("m", C.POINTER(C.c_float)), ("v", C.POINTER(C.c_float)), ("n", C.c_uint64)]


class VecBuffer(C.Structure):
    _fields_ = [("obs_dim", C.c_uint32), ("len", C.c_uint64), ("cap", C.c_uint64),
                ("obs", C.POINTER(C.c_float)), ("next_obs", C.POINTER(C.c_float)),
                ("action", C.POINTER(C.c_int32)), ("reward", C.POINTER(C.c_double)),
                ("next", C.POINTER(C.c_uint8)), ("n_episode_ends", C.c_uint64), ("cap_episode_ends", C.c_uint64),
                ("episode_ends", C.POINTER(C.c_uint64))]


class Features(C.Structure):
    _fields_ = [("obs_dim", C.c_uint32), ("n_steps", C.c_uint64), ("n_episodes", C.c_uint64), ("n_ext", C.c_uint64),
                ("n_batches", C.c_uint64), ("n_ext_batches", C.c_uint64),
                ("batch_sizes", C.POINTER(C.c_uint64)), ("ext_batch_sizes", C.POINTER(C.c_uint64)),
                ("obs", C.POINTER(C.c_float)), ("ext_obs", C.POINTER(C.c_float)),
                ("is_invalid", C.POINTER(C.c_uint8)), ("actions", C.POINTER(C.c_int64)),
                ("rewards", C.POINTER(C.c_float)), ("src_index", C.POINTER(C.c_uint64))]


class GruShape(C.Structure):
    """shape of a recurrent chain module; `cell`: CELL_GRU (default) or CELL_LSTM"""
    _fields_ = [("in_dim", C.c_uint32), ("hidden", C.c_uint32), ("mlp_hidden", C.c_uint32), ("out_dim", C.c_uint32),
                ("cell", C.c_uint32)]


CELL_GRU, CELL_LSTM = 0, 1


def LstmShape(in_dim, hidden, mlp_hidden, out_dim):
    return GruShape(in_dim, hidden, mlp_hidden, out_dim, CELL_LSTM)


class MemoryGame(C.Structure):
    _fields_ = [("num_actions", C.c_uint64), ("history_len", C.c_uint64), ("discount_factor", C.c_double)]


class ChainLanes(C.Structure):
    _fields_ = [("env", Chain), ("memory", MemoryGame), ("limit_kind", C.c_int), ("max_steps", C.c_uint64),
                ("seed_env", C.c_uint64), ("seed_actor", C.c_uint64), ("n_lanes", C.c_uint64),
                ("lane_offset", C.c_uint64), ("state", C.POINTER(C.c_uint64)),
                ("steps_remaining", C.POINTER(C.c_uint64)), ("reset_count", C.POINTER(C.c_uint64)),
                ("initial", C.POINTER(C.c_uint64)), ("env_pos", C.POINTER(C.c_uint64)), ("t_global", C.c_uint64),
                ("bandit", C.c_int), ("bandit_values", C.c_double * 2)]


class Lanes(C.Structure):
    _fields_ = [("env", CartPole), ("limit_kind", C.c_int), ("max_steps", C.c_uint64), ("seed_env", C.c_uint64),
                ("seed_actor", C.c_uint64), ("n_lanes", C.c_uint64), ("lane_offset", C.c_uint64),
                ("state", C.POINTER(CartPoleState)), ("steps_remaining", C.POINTER(C.c_uint64)),
                ("reset_count", C.POINTER(C.c_uint64)), ("t_global", C.c_uint64)]


class PeriodStats(C.Structure):
    _fields_ = [("rollout_seconds", C.c_double), ("update_seconds", C.c_double), ("steps", C.c_uint64),
                ("episodes", C.c_uint64), ("mean_episode_length", C.c_double), ("trpo", TrpoStats),
                ("critic_loss_first", C.c_float), ("critic_loss_last", C.c_float),
                ("update_intraop_seconds", C.c_double), ("update_intraop_threads", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # ORACLE_LIB: another build of the same sources (the sanitizer build of tests/test_sanitizers_cpu.py)
        path = os.environ.get("ORACLE_LIB")
        if not path:
            build()
            path = _LIB_PATH
        _lib = C.CDLL(path)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def f32p(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return _p(a, C.c_float)


def f64p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return _p(a, C.c_double)


def u8p(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return _p(a, C.c_uint8)


def i32p(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return _p(a, C.c_int32)


def i64p(a):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return _p(a, C.c_int64)


def u64p(a):
    assert a.dtype == np.uint64 and a.flags.c_contiguous
    return _p(a, C.c_uint64)


def _declare(L):
    P = C.POINTER
    L.oracle_prng_seed_from_u64.argtypes = [P(Prng), C.c_uint64]
    L.oracle_prng_from_rng.argtypes = [P(Prng), P(Prng)]
    L.oracle_prng_set_stream.argtypes = [P(Prng), C.c_uint64]
    L.oracle_prng_set_word_pos.argtypes = [P(Prng), C.c_uint64]
    L.oracle_prng_next_u32.argtypes = [P(Prng)]
    L.oracle_prng_next_u32.restype = C.c_uint32
    L.oracle_prng_next_u64.argtypes = [P(Prng)]
    L.oracle_prng_next_u64.restype = C.c_uint64
    L.oracle_prng_gen_f32.argtypes = [P(Prng)]
    L.oracle_prng_gen_f32.restype = C.c_float
    L.oracle_prng_gen_f64.argtypes = [P(Prng)]
    L.oracle_prng_gen_f64.restype = C.c_double
    L.oracle_prng_gen_range_u64.argtypes = [P(Prng), C.c_uint64, C.c_uint64]
    L.oracle_prng_gen_range_u64.restype = C.c_uint64
    L.oracle_prng_gen_bool.argtypes = [P(Prng), C.c_double]
    L.oracle_prng_gen_bool.restype = C.c_int
    L.oracle_prng_uniform_f64_inclusive.argtypes = [P(Prng), C.c_double, C.c_double]
    L.oracle_prng_uniform_f64_inclusive.restype = C.c_double

    L.oracle_cartpole_default.argtypes = [P(CartPole)]
    L.oracle_cartpole_finish.argtypes = [P(CartPole)]
    L.oracle_cartpole_initial_state.argtypes = [P(CartPole), P(Prng), P(CartPoleState)]
    L.oracle_cartpole_next_state.argtypes = [P(CartPole), P(CartPoleState), C.c_double, P(CartPoleState)]
    L.oracle_cartpole_step.argtypes = [P(CartPole), P(CartPoleState), C.c_int, P(C.c_double)]
    L.oracle_cartpole_step.restype = C.c_int
    L.oracle_chain_default.argtypes = [P(Chain)]
    L.oracle_chain_step.argtypes = [P(Chain), P(C.c_uint64), C.c_int, P(Prng), P(C.c_double)]
    L.oracle_chain_step.restype = C.c_int
    L.oracle_step_limit_apply.argtypes = [C.c_int, P(C.c_uint64)]
    L.oracle_step_limit_apply.restype = C.c_int
    L.oracle_step_limit_remaining.argtypes = [C.c_uint64, C.c_uint64]
    L.oracle_step_limit_remaining.restype = C.c_double
    L.oracle_cartpole_features.argtypes = [P(CartPoleState), C.c_int, C.c_uint64, C.c_uint64, P(C.c_float)]
    L.oracle_index_features.argtypes = [C.c_uint64, C.c_uint64, P(C.c_float)]

    L.oracle_bound_divide.argtypes = [Bound, C.c_uint64]
    L.oracle_bound_divide.restype = Bound
    L.oracle_bound_max.argtypes = [Bound, Bound]
    L.oracle_bound_max.restype = Bound
    L.oracle_bound_with_default_slack.argtypes = [C.c_uint64]
    L.oracle_bound_with_default_slack.restype = Bound
    L.oracle_take_aligned_count.argtypes = [P(C.c_uint8), C.c_uint64, C.c_uint64, C.c_uint64]
    L.oracle_take_aligned_count.restype = C.c_uint64

    L.oracle_vecbuffer_new.argtypes = [C.c_uint32]
    L.oracle_vecbuffer_new.restype = P(VecBuffer)
    L.oracle_vecbuffer_free.argtypes = [P(VecBuffer)]
    L.oracle_vecbuffer_clear.argtypes = [P(VecBuffer)]
    L.oracle_vecbuffer_write_step.argtypes = [P(VecBuffer), P(C.c_float), C.c_int32, C.c_double, C.c_int,
                                              P(C.c_float)]
    L.oracle_vecbuffer_end_experience.argtypes = [P(VecBuffer)]

    L.oracle_replay_new.argtypes = [C.c_uint64]
    L.oracle_replay_new.restype = C.c_void_p
    L.oracle_replay_free.argtypes = [C.c_void_p]
    L.oracle_replay_write_step.argtypes = [C.c_void_p, C.c_int32, C.c_int]
    L.oracle_replay_write_step.restype = C.c_int
    L.oracle_replay_end_experience.argtypes = [C.c_void_p]
    for n in ("num_steps", "num_episodes", "total_step_count"):
        f = getattr(L, "oracle_replay_" + n)
        f.argtypes = [C.c_void_p]
        f.restype = C.c_uint64
    L.oracle_replay_dump.argtypes = [C.c_void_p, P(C.c_int32), P(C.c_uint64)]

    L.oracle_packed_batch_sizes.argtypes = [P(C.c_uint64), C.c_uint64, P(C.c_uint64), C.c_uint64]
    L.oracle_packed_batch_sizes.restype = C.c_int64
    L.oracle_packed_order.argtypes = [P(C.c_uint64), C.c_uint64, P(C.c_uint64), P(C.c_uint64)]
    L.oracle_discounted_cumsum_from_end_f32.argtypes = [P(C.c_float), C.c_uint64, C.c_float, P(C.c_uint64),
                                                        C.c_uint64]
    L.oracle_packed_trim_batch_sizes.argtypes = [P(C.c_uint64), C.c_uint64, C.c_uint64, P(C.c_uint64)]
    L.oracle_packed_trim_batch_sizes.restype = C.c_uint64
    L.oracle_packed_trim_end_f32.argtypes = [P(C.c_float), P(C.c_uint64), C.c_uint64, C.c_uint64, P(C.c_float)]

    L.oracle_features_from_buffers.argtypes = [P(P(VecBuffer)), C.c_uint64]
    L.oracle_features_from_buffers.restype = P(Features)
    L.oracle_features_free.argtypes = [P(Features)]

    L.oracle_mlp_num_params.argtypes = [MlpShape]
    L.oracle_mlp_num_params.restype = C.c_uint64
    L.oracle_mlp_init.argtypes = [MlpShape, C.c_uint64, P(C.c_float)]
    L.oracle_mlp_forward_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_float)]
    L.oracle_mlp_forward_batch_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), C.c_uint64, P(C.c_float)]
    L.oracle_mlp_layers_init.argtypes = [C.c_uint32, P(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int,
                                         C.c_double, C.c_int, C.c_int, C.c_double, P(C.c_float)]
    L.oracle_mlp_layers_forward_f32.argtypes = [C.c_uint32, P(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                P(C.c_float), P(C.c_float), C.c_uint64, P(C.c_float)]

    L.oracle_log_softmax_f32.argtypes = [P(C.c_float), C.c_uint32, P(C.c_float), C.c_int]
    L.oracle_categorical_sample_u.argtypes = [P(C.c_float), C.c_uint32, C.c_float, C.c_int]
    L.oracle_categorical_sample_u.restype = C.c_int
    L.oracle_categorical_entropy_f32.argtypes = [P(C.c_float), C.c_uint32, C.c_int]
    L.oracle_categorical_entropy_f32.restype = C.c_float
    L.oracle_categorical_kl_f32.argtypes = [P(C.c_float), P(C.c_float), C.c_uint32, C.c_int]
    L.oracle_categorical_kl_f32.restype = C.c_float

    L.oracle_gae_packed.argtypes = [MlpShape, P(C.c_float), P(Features), C.c_float, C.c_float, P(C.c_float),
                                    P(C.c_float)]
    L.oracle_reward_to_go_packed.argtypes = [P(Features), C.c_float, P(C.c_float)]
    L.oracle_one_step_values_packed.argtypes = [MlpShape, P(C.c_float), P(Features), C.c_float, P(C.c_float)]
    L.oracle_lanes_one_step_targets.argtypes = [MlpShape, P(C.c_float), C.c_uint64, C.c_uint64, C.c_uint32,
                                                P(C.c_float), P(C.c_float), P(C.c_uint8), P(C.c_float), C.c_float,
                                                P(C.c_float)]
    L.oracle_seq_one_step_targets.argtypes = [C.c_uint64, C.c_uint64, P(C.c_float), P(C.c_float), P(C.c_float),
                                              P(C.c_uint8), C.c_float, P(C.c_float)]

    L.oracle_trpo_cfg_default.argtypes = [P(TrpoCfg)]
    L.oracle_policy_grad_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_int64), P(C.c_float),
                                         C.c_uint64, P(C.c_float), P(C.c_float)]
    L.oracle_policy_fvp_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), C.c_uint64, P(C.c_float), C.c_float,
                                        P(C.c_float)]
    L.oracle_policy_loss_kl_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_int64),
                                            P(C.c_float), C.c_uint64, P(C.c_float), P(C.c_float)]
    L.oracle_trpo_update_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_int64), P(C.c_float),
                                         C.c_uint64, P(TrpoCfg), P(TrpoStats), P(C.c_float)]
    L.oracle_policy_grad_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_int64), P(C.c_double),
                                         C.c_uint64, P(C.c_double), P(C.c_double)]
    L.oracle_policy_fvp_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), C.c_uint64, P(C.c_double),
                                        C.c_double, P(C.c_double)]
    L.oracle_cg_dense_f32.argtypes = [P(C.c_float), P(C.c_float), C.c_uint32, C.c_uint64, C.c_double, P(C.c_float)]
    L.oracle_cg_dense_f64.argtypes = [P(C.c_double), P(C.c_double), C.c_uint32, C.c_uint64, C.c_double,
                                      P(C.c_double)]
    L.oracle_trpo_update_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_int64), P(C.c_double),
                                         C.c_uint64, P(TrpoCfg), P(TrpoStats), P(C.c_double)]
    L.oracle_policy_loss_kl_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_int64),
                                            P(C.c_double), C.c_uint64, P(C.c_double), P(C.c_double)]

    L.oracle_adam_cfg_default.argtypes = [P(AdamCfg)]
    L.oracle_adam_new.argtypes = [C.c_uint64]
    L.oracle_adam_new.restype = P(AdamState)
    L.oracle_adam_free.argtypes = [P(AdamState)]
    L.oracle_adam_step_f32.argtypes = [P(AdamState), P(AdamCfg), P(C.c_float), P(C.c_float)]
    L.oracle_critic_grad_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_float), C.c_uint64,
                                         P(C.c_float), P(C.c_float)]
    L.oracle_critic_update_f32.argtypes = [MlpShape, P(C.c_float), P(AdamState), P(AdamCfg), P(C.c_float),
                                           P(C.c_float), C.c_uint64, C.c_uint64, P(C.c_float)]
    L.oracle_grad_f64_mt.argtypes = [C.c_int, MlpShape, P(C.c_float), P(C.c_float), P(C.c_uint8), P(C.c_float),
                                     P(C.c_float), C.c_uint64, P(C.c_double), P(C.c_double)]
    L.oracle_grad_f32_mt.argtypes = L.oracle_grad_f64_mt.argtypes

    L.oracle_tabular_q_new.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
    L.oracle_tabular_q_new.restype = C.c_void_p
    L.oracle_tabular_q_free.argtypes = [C.c_void_p]
    L.oracle_tabular_q_step_update.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_uint64]
    L.oracle_tabular_q_read.argtypes = [C.c_void_p, P(C.c_double), P(C.c_uint64)]
    L.oracle_chain_step_draw.argtypes = [P(Chain), P(C.c_uint64), C.c_int, C.c_float, P(C.c_double)]
    L.oracle_chain_step_draw.restype = C.c_int
    L.oracle_chain_tabular_q_train.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, P(C.c_double),
                                               P(C.c_uint64), P(C.c_uint64)]
    L.oracle_chain_tabular_q_eval.argtypes = [P(C.c_double), C.c_uint64, C.c_uint64, P(C.c_int32)]
    L.oracle_chain_tabular_q_eval.restype = C.c_double

    L.oracle_lanes_new.argtypes = [P(CartPole), C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.oracle_lanes_new.restype = P(Lanes)
    L.oracle_lanes_free.argtypes = [P(Lanes)]
    L.oracle_lanes_reset.argtypes = [P(Lanes)]
    L.oracle_lanes_get_state.argtypes = [P(Lanes), P(C.c_double), P(C.c_int32), P(C.c_uint64), P(C.c_uint64)]
    L.oracle_lanes_set_state.argtypes = [P(Lanes), P(C.c_double), P(C.c_int32), P(C.c_uint64), P(C.c_uint64)]
    L.oracle_lanes_step.argtypes = [P(Lanes), P(C.c_uint8), P(C.c_float), P(C.c_uint8), P(C.c_float), P(C.c_float)]
    L.oracle_lanes_observe.argtypes = [P(Lanes), P(C.c_float)]
    L.oracle_lanes_rollout.argtypes = [P(Lanes), MlpShape, P(C.c_float), C.c_uint64, P(C.c_float), P(C.c_uint8),
                                       P(C.c_float), P(C.c_uint8), P(C.c_float), C.c_int]
    L.oracle_lanes_gae.argtypes = [MlpShape, P(C.c_float), C.c_uint64, C.c_uint64, C.c_uint32, P(C.c_float),
                                   P(C.c_float), P(C.c_uint8), P(C.c_float), C.c_float, C.c_float, P(C.c_float),
                                   P(C.c_float), P(C.c_float)]
    L.oracle_lanes_to_vecbuffer.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, P(C.c_float), P(C.c_uint8),
                                            P(C.c_float), P(C.c_uint8), P(C.c_float), C.c_int, P(C.c_uint64)]
    L.oracle_lanes_to_vecbuffer.restype = P(VecBuffer)
    L.oracle_gru_num_params.argtypes = [GruShape]
    L.oracle_gru_num_params.restype = C.c_uint64
    L.oracle_gru_init.argtypes = [GruShape, C.c_uint64, P(C.c_float)]
    L.oracle_stack_num_params.argtypes = [GruShape, C.c_uint32]
    L.oracle_stack_num_params.restype = C.c_uint64
    L.oracle_stack_init.argtypes = [GruShape, C.c_uint32, C.c_uint64, P(C.c_float)]
    L.oracle_stack_init_with.argtypes = [GruShape, C.c_uint32, C.c_uint64, C.c_void_p, P(C.c_float)]
    L.oracle_stack_seq_forward_f32.argtypes = [GruShape, C.c_uint32, P(C.c_float), C.c_uint64, C.c_uint64, P(C.c_float),
                                               P(C.c_uint8), P(C.c_float), P(C.c_float), P(C.c_float)]
    L.oracle_stack_seq_forward_f64.argtypes = [GruShape, C.c_uint32, P(C.c_double), C.c_uint64, C.c_uint64,
                                               P(C.c_double), P(C.c_uint8), P(C.c_double), P(C.c_double), P(C.c_double)]
    L.oracle_gru_step_f32.argtypes = [GruShape, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float)]
    L.oracle_gru_step_f64.argtypes = [GruShape, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double)]
    L.oracle_gru_seq_forward_f32.argtypes = [GruShape, P(C.c_float), C.c_uint64, C.c_uint64, P(C.c_float),
                                             P(C.c_uint8), P(C.c_float), P(C.c_float), P(C.c_float)]
    L.oracle_gru_seq_forward_f64.argtypes = [GruShape, P(C.c_double), C.c_uint64, C.c_uint64, P(C.c_double),
                                             P(C.c_uint8), P(C.c_double), P(C.c_double), P(C.c_double)]
    L.oracle_gru_seq_backward_f32.argtypes = [GruShape, P(C.c_float), C.c_uint64, C.c_uint64, P(C.c_float),
                                              P(C.c_uint8), P(C.c_float), P(C.c_float)]
    L.oracle_gru_seq_backward_f64.argtypes = [GruShape, P(C.c_double), C.c_uint64, C.c_uint64, P(C.c_double),
                                              P(C.c_uint8), P(C.c_double), P(C.c_double)]
    L.oracle_gru_seq_jvp_f32.argtypes = [GruShape, P(C.c_float), P(C.c_float), C.c_uint64, C.c_uint64, P(C.c_float),
                                         P(C.c_uint8), P(C.c_float)]
    L.oracle_gru_seq_jvp_f64.argtypes = [GruShape, P(C.c_double), P(C.c_double), C.c_uint64, C.c_uint64,
                                         P(C.c_double), P(C.c_uint8), P(C.c_double)]
    L.oracle_chain_lanes_new.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_uint64]
    L.oracle_chain_lanes_new.restype = P(ChainLanes)
    L.oracle_memory_lanes_new.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                          C.c_uint64, C.c_uint64]
    L.oracle_memory_lanes_new.restype = P(ChainLanes)
    L.oracle_bandit_lanes_new.argtypes = [P(C.c_double), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.oracle_bandit_lanes_new.restype = P(ChainLanes)
    L.oracle_memory_lanes_get_extra.argtypes = [P(ChainLanes), P(C.c_uint64), P(C.c_uint64)]
    L.oracle_chain_lanes_free.argtypes = [P(ChainLanes)]
    L.oracle_chain_lanes_reset.argtypes = [P(ChainLanes)]
    L.oracle_chain_lanes_obs_dim.argtypes = [P(ChainLanes)]
    L.oracle_chain_lanes_obs_dim.restype = C.c_uint32
    L.oracle_chain_lanes_observe.argtypes = [P(ChainLanes), P(C.c_float)]
    L.oracle_chain_lanes_get_state.argtypes = [P(ChainLanes), P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
    L.oracle_chain_lanes_step.argtypes = [P(ChainLanes), P(C.c_uint8), P(C.c_float), P(C.c_uint8), P(C.c_float),
                                          P(C.c_float)]
    L.oracle_chain_lanes_rollout_gru.argtypes = [P(ChainLanes), GruShape, P(C.c_float), C.c_uint64, P(C.c_float),
                                                 P(C.c_uint8), P(C.c_float), P(C.c_uint8), P(C.c_float), C.c_int]
    L.oracle_chain_lanes_rollout_mlp.argtypes = [P(ChainLanes), MlpShape, P(C.c_float), C.c_uint64, P(C.c_float),
                                                 P(C.c_uint8), P(C.c_float), P(C.c_uint8), P(C.c_float)]
    L.oracle_lanes_rollout_gru.argtypes = [P(Lanes), GruShape, P(C.c_float), C.c_uint64, P(C.c_float), P(C.c_uint8),
                                           P(C.c_float), P(C.c_uint8), P(C.c_float), C.c_int]
    L.oracle_seq_gae.argtypes = [C.c_uint64, C.c_uint64, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_uint8),
                                 C.c_float, C.c_float, P(C.c_float), P(C.c_float)]
    L.oracle_seq_policy_dlogits_f32.argtypes = [C.c_uint64, P(C.c_float), P(C.c_uint8), P(C.c_float), P(C.c_float),
                                                C.c_int, C.c_float, C.c_float, P(C.c_float), P(C.c_float),
                                                P(C.c_double), P(C.c_double)]
    L.oracle_policy_logp_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_int64), C.c_uint64,
                                         P(C.c_float), P(C.c_float)]
    L.oracle_policy_logp_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_int64), C.c_uint64,
                                         P(C.c_double), P(C.c_double)]
    L.oracle_ppo_grad_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_int64), P(C.c_float),
                                      P(C.c_float), C.c_uint64, C.c_float, C.c_float, P(C.c_float), P(C.c_float)]
    L.oracle_ppo_grad_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_int64), P(C.c_double),
                                      P(C.c_double), C.c_uint64, C.c_double, C.c_double, P(C.c_double),
                                      P(C.c_double)]
    L.oracle_reinforce_loss_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_int64), P(C.c_float),
                                            C.c_uint64]
    L.oracle_reinforce_loss_f32.restype = C.c_float
    L.oracle_reinforce_loss_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_int64), P(C.c_double),
                                            C.c_uint64]
    L.oracle_reinforce_loss_f64.restype = C.c_double
    L.oracle_ppo_update_f32.argtypes = [MlpShape, P(C.c_float), P(AdamState), P(AdamCfg), P(C.c_float),
                                        P(C.c_int64), P(C.c_float), C.c_uint64, C.c_uint64, C.c_double,
                                        P(C.c_float), P(C.c_float)]
    L.oracle_reinforce_update_f32.argtypes = [MlpShape, P(C.c_float), P(AdamState), P(AdamCfg), P(C.c_float),
                                              P(C.c_int64), P(C.c_float), C.c_uint64, P(C.c_float), P(C.c_float)]
    L.oracle_prng_word_pos.argtypes = [P(Prng)]
    L.oracle_prng_word_pos.restype = C.c_uint64
    L.oracle_prng_from_seed.argtypes = [P(Prng), P(C.c_uint32)]
    L.oracle_dqn_store_new.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    L.oracle_dqn_store_new.restype = C.c_void_p
    L.oracle_dqn_store_free.argtypes = [C.c_void_p]
    L.oracle_dqn_store_actor_pos.argtypes = [C.c_void_p, C.c_uint64]
    L.oracle_dqn_store_actor_pos.restype = C.c_uint64
    L.oracle_dqn_store_lane_info.argtypes = [C.c_void_p, C.c_uint64, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)]
    L.oracle_dqn_store_lane_dump.argtypes = [C.c_void_p, C.c_uint64, P(C.c_int32), P(C.c_uint64)]
    L.oracle_dqn_store_step.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, P(C.c_float), P(C.c_uint8),
                                        P(C.c_float), P(C.c_uint8), P(C.c_float)]
    L.oracle_lanes_rollout_dqn.argtypes = [P(Lanes), C.c_void_p, MlpShape, P(C.c_float), C.c_uint64, C.c_double,
                                           P(C.c_uint8)]
    L.oracle_lanes_rollout_dqn.restype = C.c_int
    L.oracle_dqn_sample.argtypes = [C.c_void_p, P(Prng), C.c_uint64, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32),
                                    C.c_uint64, P(C.c_uint64)]
    L.oracle_dqn_sample.restype = C.c_int64
    L.oracle_dqn_minibatch.argtypes = [C.c_void_p, C.c_uint64, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32), MlpShape,
                                       P(C.c_float), C.c_float, C.c_int, P(C.c_float), P(C.c_int64), P(C.c_float)]
    L.oracle_dqn_grad_f32.argtypes = [MlpShape, P(C.c_float), P(C.c_float), P(C.c_int64), P(C.c_float), C.c_uint64,
                                      P(C.c_float), P(C.c_float)]
    L.oracle_dqn_grad_f64.argtypes = [MlpShape, P(C.c_double), P(C.c_double), P(C.c_int64), P(C.c_double),
                                      C.c_uint64, P(C.c_double), P(C.c_double)]
    L.oracle_dqn_update_f32.argtypes = [C.c_void_p, P(Prng), MlpShape, P(C.c_float), P(AdamState), P(AdamCfg),
                                        C.c_uint64, C.c_uint64, C.c_float, C.c_int, P(C.c_float)]
    L.oracle_dqn_update_f32.restype = C.c_int
    L.oracle_exploration_rate.argtypes = [C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_int]
    L.oracle_exploration_rate.restype = C.c_double
    L.oracle_collection_update_size.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
    L.oracle_collection_update_size.restype = Bound
    L.oracle_cartpole_rollout_only.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                               P(C.c_float), P(C.c_uint64)]
    L.oracle_cartpole_rollout_only.restype = C.c_double
    L.oracle_cartpole_trpo_period.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64,
                                              C.c_uint64, C.c_uint32, P(C.c_float), P(C.c_float), P(AdamState),
                                              C.c_uint64, P(PeriodStats)]
    L.oracle_cartpole_trpo_period_ex.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64,
                                                 C.c_uint64, C.c_uint32, P(C.c_float), P(C.c_float), P(AdamState),
                                                 C.c_uint64, C.c_uint32, P(PeriodStats)]
    L.oracle_cpu_sample_collect.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                            P(C.c_float), P(C.c_float), P(PeriodStats)]
    L.oracle_cpu_sample_collect.restype = C.c_void_p
    L.oracle_cpu_sample_update_one_thread.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, P(PeriodStats)]
    L.oracle_cpu_sample_update_one_thread.restype = C.c_double
    L.oracle_cpu_sample_update_intraop.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]
    L.oracle_cpu_sample_update_intraop.restype = C.c_double
    L.oracle_cpu_sample_free.argtypes = [C.c_void_p]
    L.oracle_cpu_sample_free.restype = None


# ---------------------------------------------------------------------------------------------
# numpy-level conveniences used by the tests


def cartpole_default(use_libm=False):
    env = CartPole()
    lib().oracle_cartpole_default(C.byref(env))
    env.use_libm = 1 if use_libm else 0
    return env


def mlp_init(shape, seed):
    p = np.zeros(int(lib().oracle_mlp_num_params(shape)), dtype=np.float32)
    lib().oracle_mlp_init(shape, seed, f32p(p))
    return p


def mlp_forward_batch(shape, params, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((x.shape[0], shape.out_dim), dtype=np.float32)
    lib().oracle_mlp_forward_batch_f32(shape, f32p(params), f32p(x), x.shape[0], f32p(out))
    return out


INIT_KINDS = ["Zeros", "Constant", "Uniform", "Normal", "Orthogonal"]  # Initializer (initializers.rs:8-21)
SCALES = ["Constant", "FanIn", "FanOut", "FanAvg"]                          # VarianceScale (initializers.rs:40-62)


def mlp_layers_init(in_dim, hidden, out_dim, seed, kernel_init=("Uniform", "FanAvg", 0.0), bias_init=("Uniform", "FanAvg", 0.0)):
    """Linear::new for every layer with the given (kind, scale, value) initializers (oracle_mlp_layers_init)"""
    P_ = sum(fi * fo + fo for fi, fo in zip([in_dim] + list(hidden), list(hidden) + [out_dim]))
    p = np.zeros(P_, dtype=np.float32)
    hs = (C.c_uint32 * max(len(hidden), 1))(*hidden)
    lib().oracle_mlp_layers_init(in_dim, hs, len(hidden), out_dim, seed, INIT_KINDS.index(kernel_init[0]),
                                 SCALES.index(kernel_init[1]), float(kernel_init[2]), INIT_KINDS.index(bias_init[0]),
                                 SCALES.index(bias_init[1]), float(bias_init[2]), f32p(p))
    return p


ACTIVATIONS = ["Identity", "Relu", "Sigmoid", "Tanh"]  # the reference enum's declaration order (ff/activation.rs:11-20)


def mlp_layers_forward(in_dim, hidden, out_dim, params, x, activation="Relu", output_activation="Identity"):
    """Mlp::forward for any hidden_sizes / activations (oracle_mlp_layers_forward_f32); x [rows][in] -> [rows][out]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros((x.shape[0], out_dim), dtype=np.float32)
    hs = (C.c_uint32 * max(len(hidden), 1))(*hidden)
    lib().oracle_mlp_layers_forward_f32(in_dim, hs, len(hidden), out_dim, ACTIVATIONS.index(activation),
                                        ACTIVATIONS.index(output_activation), f32p(np.ascontiguousarray(params, dtype=np.float32)),
                                        f32p(x), x.shape[0], f32p(out))
    return out


class LaneSim:
    """numpy view of oracle_lanes: the engine's vectorised env restated with the scalar pieces."""

    def __init__(self, n_lanes, max_steps=500, limit=LIMIT_VISIBLE, lane_offset=0, seed_env=0, seed_actor=1,
                 env=None):
        self.env = env if env is not None else cartpole_default()
        self.ptr = lib().oracle_lanes_new(C.byref(self.env), limit, max_steps, n_lanes, lane_offset, seed_env,
                                          seed_actor)
        self.n = n_lanes
        self.D = 5 if limit == LIMIT_VISIBLE else 4

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().oracle_lanes_free(self.ptr)
            self.ptr = None

    def reset(self):
        lib().oracle_lanes_reset(self.ptr)

    def get_state(self):
        st = np.zeros((4, self.n), dtype=np.float64)
        nv = np.zeros(self.n, dtype=np.int32)
        rem = np.zeros(self.n, dtype=np.uint64)
        rc = np.zeros(self.n, dtype=np.uint64)
        lib().oracle_lanes_get_state(self.ptr, f64p(st), i32p(nv), u64p(rem), u64p(rc))
        return st, nv, rem, rc

    def set_state(self, st, nv, rem, rc):
        lib().oracle_lanes_set_state(self.ptr, f64p(np.ascontiguousarray(st, dtype=np.float64)),
                                     i32p(np.ascontiguousarray(nv, dtype=np.int32)),
                                     u64p(np.ascontiguousarray(rem, dtype=np.uint64)),
                                     u64p(np.ascontiguousarray(rc, dtype=np.uint64)))

    def observe(self):
        obs = np.zeros((self.D, self.n), dtype=np.float32)
        lib().oracle_lanes_observe(self.ptr, f32p(obs))
        return obs

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.uint8)
        reward = np.zeros(self.n, dtype=np.float32)
        flag = np.zeros(self.n, dtype=np.uint8)
        obs = np.zeros((self.D, self.n), dtype=np.float32)
        term = np.zeros((self.D, self.n), dtype=np.float32)
        lib().oracle_lanes_step(self.ptr, u8p(actions), f32p(reward), u8p(flag), f32p(obs), f32p(term))
        return reward, flag, obs, term

    def rollout(self, pshape, pparams, T, threads=8):
        n, D = self.n, self.D
        obs = np.zeros((D, T + 1, n), dtype=np.float32)
        action = np.zeros((T, n), dtype=np.uint8)
        reward = np.zeros((T, n), dtype=np.float32)
        flag = np.zeros((T, n), dtype=np.uint8)
        term = np.zeros((D, T, n), dtype=np.float32)
        lib().oracle_lanes_rollout(self.ptr, pshape, f32p(pparams), T, f32p(obs), u8p(action), f32p(reward),
                                   u8p(flag), f32p(term), threads)
        return dict(obs=obs, action=action, reward=reward, flag=flag, term_obs=term)


def _lanesim_rollout_gru(self, shape, params, T, threads=8):
    n, D = self.n, self.D
    obs = np.zeros((D, T + 1, n), dtype=np.float32)
    action = np.zeros((T, n), dtype=np.uint8)
    reward = np.zeros((T, n), dtype=np.float32)
    flag = np.zeros((T, n), dtype=np.uint8)
    term = np.zeros((D, T, n), dtype=np.float32)
    lib().oracle_lanes_rollout_gru(self.ptr, shape, f32p(params), T, f32p(obs), u8p(action), f32p(reward), u8p(flag),
                                   f32p(term), threads)
    return dict(obs=obs, action=action, reward=reward, flag=flag, term_obs=term)


LaneSim.rollout_gru = _lanesim_rollout_gru


def trpo_update(pshape, params, x, a, adv, cfg=None, f64=False):
    """Run the oracle TRPO step; returns (new_params, stats, step_dir)."""
    if cfg is None:
        cfg = TrpoCfg()
        lib().oracle_trpo_cfg_default(C.byref(cfg))
    st = TrpoStats()
    if f64:
        p = np.ascontiguousarray(params, dtype=np.float64).copy()
        sd = np.zeros_like(p)
        lib().oracle_trpo_update_f64(pshape, f64p(p), f64p(np.ascontiguousarray(x, dtype=np.float64)), i64p(a),
                                     f64p(np.ascontiguousarray(adv, dtype=np.float64)), len(a), C.byref(cfg),
                                     C.byref(st), f64p(sd))
    else:
        p = np.ascontiguousarray(params, dtype=np.float32).copy()
        sd = np.zeros_like(p)
        lib().oracle_trpo_update_f32(pshape, f32p(p), f32p(x), i64p(a), f32p(adv), len(a), C.byref(cfg), C.byref(st),
                                     f32p(sd))
    return p, st, sd


def lanes_gae(cshape, cparams, traj, gamma, lam):
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    values = np.zeros((T + 1, n), dtype=np.float32)
    adv = np.zeros((T, n), dtype=np.float32)
    rtg = np.zeros((T, n), dtype=np.float32)
    lib().oracle_lanes_gae(cshape, f32p(cparams), n, T, D, f32p(obs), f32p(traj["reward"]), u8p(traj["flag"]),
                           f32p(traj["term_obs"]), gamma, lam, f32p(values), f32p(adv), f32p(rtg))
    return values, adv, rtg


def lanes_one_step_targets(cshape, cparams, traj, gamma):
    """StepValueTarget::OneStepTd on a lane trajectory: r + gamma * V(successor), [T][n]"""
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    out = np.zeros((T, n), dtype=np.float32)
    lib().oracle_lanes_one_step_targets(cshape, f32p(cparams), n, T, D, f32p(obs), f32p(traj["reward"]),
                                        u8p(traj["flag"]), f32p(traj["term_obs"]), gamma, f32p(out))
    return out


def flat_samples(traj):
    """[D][T+1][n] trajectory -> (obs [B][D], actions [B] i64) in the engine's flat sample order b = t*n + lane."""
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    x = np.ascontiguousarray(obs[:, :T, :].reshape(D, T * n).T, dtype=np.float32)
    a = np.ascontiguousarray(traj["action"].reshape(T * n).astype(np.int64))
    return x, a


def grad_f64_mt(kind, shape, params, x, actions=None, aux=None, v=None, f32_samples=False):
    """f64 ground truth over all samples, OpenMP over chunks (oracle_grad_f64_mt): kind "policy" (surrogate gradient,
    aux = advantages), "fvp" (Fisher-vector product with tangent v, no regulariser), "critic" (MSE gradient, aux =
    targets); x [B][D] f32, actions [B] u8.  Returns (gradient f64 [P], loss f64).  f32_samples: the per-sample
    arithmetic of the f32 oracle functions instead (chunk sums still in f64) — a correct f32 evaluation as yardstick."""
    k = {"policy": 0, "fvp": 1, "critic": 2}[kind]
    params = np.ascontiguousarray(params, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    g = np.zeros(len(params), dtype=np.float64)
    loss = C.c_double(0.0)
    keep = [np.ascontiguousarray(t, dtype=d) if t is not None else None
            for t, d in ((actions, np.uint8), (aux, np.float32), (v, np.float32))]
    ptr = lambda a, ct: _p(a, ct) if a is not None else None  # noqa: E731
    fn = lib().oracle_grad_f32_mt if f32_samples else lib().oracle_grad_f64_mt
    fn(k, shape, _p(params, C.c_float), _p(x, C.c_float), ptr(keep[0], C.c_uint8),
                             ptr(keep[1], C.c_float), ptr(keep[2], C.c_float), n, _p(g, C.c_double), C.byref(loss))
    return g, loss.value


def u32p(a):
    assert a.dtype == np.uint32 and a.flags.c_contiguous
    return _p(a, C.c_uint32)


class DqnSim:
    """The DQN path restated on the lane model: per-lane ReplayBuffers + agent Prng + Adam (oracle/dqn.c)."""

    def __init__(self, sim, qshape, qparams, capacity, agent_key, minibatch_steps, gamma=0.99, one_step_td=False,
                 adam_cfg=None):
        L = lib()
        self.sim = sim
        self.qshape = qshape
        self.qparams = np.ascontiguousarray(qparams, dtype=np.float32).copy()
        self.capacity = capacity
        self.store = L.oracle_dqn_store_new(sim.n, capacity, sim.D)
        self.agent_rng = Prng()
        key = (C.c_uint32 * 8)(*[int(k) for k in agent_key])
        L.oracle_prng_from_seed(C.byref(self.agent_rng), key)
        self.minibatch_steps = minibatch_steps
        self.gamma = gamma
        self.td = 1 if one_step_td else 0
        self.acfg = adam_cfg
        if self.acfg is None:
            self.acfg = AdamCfg()
            L.oracle_adam_cfg_default(C.byref(self.acfg))
        self.opt = L.oracle_adam_new(len(self.qparams))

    def __del__(self):
        if getattr(self, "store", None):
            lib().oracle_dqn_store_free(self.store)
            lib().oracle_adam_free(self.opt)
            self.store = None

    def collect(self, T, eps):
        flags = np.zeros((T, self.sim.n), dtype=np.uint8)
        full = lib().oracle_lanes_rollout_dqn(self.sim.ptr, self.store, self.qshape, f32p(self.qparams), T, eps,
                                              u8p(flags))
        return flags, bool(full)

    def lane_info(self, lane):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().oracle_dqn_store_lane_info(self.store, lane, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value  # num_steps, num_episodes, total_step_count

    def lane_dump(self, lane):
        ns, ne, _ = self.lane_info(lane)
        tags = np.zeros(max(ns, 1), dtype=np.int32)
        lens = np.zeros(max(ne, 1), dtype=np.uint64)
        lib().oracle_dqn_store_lane_dump(self.store, lane, i32p(tags), u64p(lens))
        return tags[:ns], lens[:ne]

    def actor_pos(self):
        return np.array([lib().oracle_dqn_store_actor_pos(self.store, i) for i in range(self.sim.n)],
                        dtype=np.uint64)

    def step_data(self, lane, abs_index):
        D = self.sim.D
        obs = np.zeros(D, dtype=np.float32)
        nobs = np.zeros(D, dtype=np.float32)
        a, nx, r = C.c_uint8(), C.c_uint8(), C.c_float()
        lib().oracle_dqn_store_step(self.store, lane, abs_index, f32p(obs), C.byref(a), C.byref(r), C.byref(nx),
                                    f32p(nobs))
        return obs, a.value, r.value, nx.value, nobs

    def agent_pos(self):
        return lib().oracle_prng_word_pos(C.byref(self.agent_rng))

    def sample(self):
        cap = self.minibatch_steps
        lanes = np.zeros(cap, dtype=np.uint32)
        starts = np.zeros(cap, dtype=np.uint32)
        lens = np.zeros(cap, dtype=np.uint32)
        ns = C.c_uint64()
        ne = lib().oracle_dqn_sample(self.store, C.byref(self.agent_rng), self.minibatch_steps, u32p(lanes),
                                     u32p(starts), u32p(lens), cap, C.byref(ns))
        assert ne >= 0, "a lane holds no complete episode"
        return lanes[:ne].copy(), starts[:ne].copy(), lens[:ne].copy(), ns.value

    def minibatch(self, lanes, starts, lens):
        n = int(lens.sum())
        obs = np.zeros((n, self.sim.D), dtype=np.float32)
        actions = np.zeros(n, dtype=np.int64)
        targets = np.zeros(n, dtype=np.float32)
        lib().oracle_dqn_minibatch(self.store, len(lanes), u32p(lanes), u32p(starts), u32p(lens), self.qshape,
                                   f32p(self.qparams), self.gamma, self.td, f32p(obs), i64p(actions), f32p(targets))
        return obs, actions, targets

    def grad(self, obs, actions, targets, f64=False):
        L = lib()
        if f64:
            g = np.zeros(len(self.qparams), dtype=np.float64)
            loss = C.c_double()
            L.oracle_dqn_grad_f64(self.qshape, f64p(self.qparams.astype(np.float64)),
                                  f64p(np.ascontiguousarray(obs, dtype=np.float64)), i64p(actions),
                                  f64p(np.ascontiguousarray(targets, dtype=np.float64)), len(actions), f64p(g),
                                  C.byref(loss))
        else:
            g = np.zeros(len(self.qparams), dtype=np.float32)
            loss = C.c_float()
            L.oracle_dqn_grad_f32(self.qshape, f32p(self.qparams), f32p(obs), i64p(actions), f32p(targets),
                                  len(actions), f32p(g), C.byref(loss))
        return g, loss.value

    def update(self, opt_steps):
        losses = np.zeros(max(opt_steps, 1), dtype=np.float32)
        rc = lib().oracle_dqn_update_f32(self.store, C.byref(self.agent_rng), self.qshape, f32p(self.qparams),
                                         self.opt, C.byref(self.acfg), self.minibatch_steps, opt_steps, self.gamma,
                                         self.td, f32p(losses))
        assert rc == 0
        return losses[:opt_steps]


# ---------------------------------------------------------------------------------------------
# recurrent configuration (oracle/seq.c)


def gru_init(shape, seed):
    p = np.zeros(int(lib().oracle_gru_num_params(shape)), dtype=np.float32)
    lib().oracle_gru_init(shape, seed, f32p(p))
    return p


def gru_seq_forward(shape, params, traj, f64=False, want_succ=True):
    """teacher-forced forward over a lane trajectory dict; returns (out [A][T][n], succ_out or None)"""
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    dt = np.float64 if f64 else np.float32
    ptr = f64p if f64 else f32p
    out = np.zeros((shape.out_dim, T, n), dtype=dt)
    succ = np.zeros((shape.out_dim, T, n), dtype=dt) if want_succ else None
    fn = lib().oracle_gru_seq_forward_f64 if f64 else lib().oracle_gru_seq_forward_f32
    fn(shape, ptr(np.ascontiguousarray(params, dtype=dt)), n, T, ptr(np.ascontiguousarray(obs, dtype=dt)),
       u8p(traj["flag"]), ptr(np.ascontiguousarray(traj["term_obs"], dtype=dt)), ptr(out),
       ptr(succ) if want_succ else None)
    return out, succ


def stack_init(shape, num_layers, seed):
    """RnnBaseConfig { num_layers } chain (oracle/stack_impl.inc): the flat vector of oracle_stack_init"""
    p = np.zeros(int(lib().oracle_stack_num_params(shape, num_layers)), dtype=np.float32)
    lib().oracle_stack_init(shape, num_layers, seed, f32p(p))
    return p


class InitSpec(C.Structure):
    """oracle_init_spec"""
    _fields_ = [("kind", C.c_int32), ("scale", C.c_int32), ("value", C.c_double)]


_INIT_KINDS = ["Zeros", "Constant", "Uniform", "Normal", "Orthogonal"]
_INIT_SCALES = ["Constant", "FanIn", "FanOut", "FanAvg"]
RNN_DEFAULT_INITS = (("Uniform", "FanAvg", 0.0), ("Orthogonal", "FanAvg", 0.0), ("Zeros", "FanAvg", 0.0),
                     ("Uniform", "FanAvg", 0.0), ("Uniform", "FanAvg", 0.0))


def stack_init_with(shape, num_layers, seed, inits=RNN_DEFAULT_INITS):
    """RnnBaseConfig's (input, hidden, bias) initializers and the chain MLP's (kernel, bias) as (kind, scale, value)"""
    p = np.zeros(int(lib().oracle_stack_num_params(shape, num_layers)), dtype=np.float32)
    arr = (InitSpec * 5)(*[InitSpec(_INIT_KINDS.index(k), _INIT_SCALES.index(sc), float(v)) for k, sc, v in inits])
    lib().oracle_stack_init_with(shape, num_layers, seed, arr, f32p(p))
    return p


def stack_seq_forward(shape, num_layers, params, traj, f64=False, want_succ=True):
    """gru_seq_forward for a chain of `num_layers` stacked recurrent layers"""
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    dt = np.float64 if f64 else np.float32
    ptr = f64p if f64 else f32p
    out = np.zeros((shape.out_dim, T, n), dtype=dt)
    succ = np.zeros((shape.out_dim, T, n), dtype=dt) if want_succ else None
    fn = lib().oracle_stack_seq_forward_f64 if f64 else lib().oracle_stack_seq_forward_f32
    fn(shape, num_layers, ptr(np.ascontiguousarray(params, dtype=dt)), n, T, ptr(np.ascontiguousarray(obs, dtype=dt)),
       u8p(traj["flag"]), ptr(np.ascontiguousarray(traj["term_obs"], dtype=dt)), ptr(out),
       ptr(succ) if want_succ else None)
    return out, succ


def gru_seq_backward(shape, params, traj, dout, f64=False):
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    dt = np.float64 if f64 else np.float32
    ptr = f64p if f64 else f32p
    g = np.zeros(len(params), dtype=dt)
    fn = lib().oracle_gru_seq_backward_f64 if f64 else lib().oracle_gru_seq_backward_f32
    fn(shape, ptr(np.ascontiguousarray(params, dtype=dt)), n, T, ptr(np.ascontiguousarray(obs, dtype=dt)),
       u8p(traj["flag"]), ptr(np.ascontiguousarray(dout, dtype=dt)), ptr(g))
    return g


def gru_seq_jvp(shape, params, tangent, traj, f64=False):
    obs = traj["obs"]
    D, T1, n = obs.shape
    T = T1 - 1
    dt = np.float64 if f64 else np.float32
    ptr = f64p if f64 else f32p
    out = np.zeros((shape.out_dim, T, n), dtype=dt)
    fn = lib().oracle_gru_seq_jvp_f64 if f64 else lib().oracle_gru_seq_jvp_f32
    fn(shape, ptr(np.ascontiguousarray(params, dtype=dt)), ptr(np.ascontiguousarray(tangent, dtype=dt)), n, T,
       ptr(np.ascontiguousarray(obs, dtype=dt)), u8p(traj["flag"]), ptr(out))
    return out


def gru_policy_fvp(shape, params, v, traj, reg, f64=False):
    """Fisher-vector product of the mean KL at params (J^T (diag(p) - p p^T) J v / B + reg v) through time"""
    dt = np.float64 if f64 else np.float32
    logits, _ = gru_seq_forward(shape, params, traj, f64=f64, want_succ=False)
    od = gru_seq_jvp(shape, params, v, traj, f64=f64)
    z = logits - logits.max(0)
    lp = z - np.log(np.exp(z).sum(0))
    pr = np.exp(lp).astype(dt)
    B = logits[0].size
    pdz = (pr * od).sum(0)
    dz = (pr * (od - pdz) / dt(B)).astype(dt)
    return gru_seq_backward(shape, params, traj, dz, f64=f64) + dt(reg) * np.asarray(v, dtype=dt)


def seq_gae(values, succ_values, traj, gamma, lam):
    T, n = traj["reward"].shape
    adv = np.zeros((T, n), dtype=np.float32)
    rtg = np.zeros((T, n), dtype=np.float32)
    lib().oracle_seq_gae(n, T, f32p(np.ascontiguousarray(values, dtype=np.float32)),
                         f32p(np.ascontiguousarray(succ_values, dtype=np.float32)), f32p(traj["reward"]),
                         u8p(traj["flag"]), gamma, lam, f32p(adv), f32p(rtg))
    return adv, rtg


def seq_one_step_targets(values, succ_values, traj, gamma):
    T, n = traj["reward"].shape
    out = np.zeros((T, n), dtype=np.float32)
    lib().oracle_seq_one_step_targets(n, T, f32p(np.ascontiguousarray(values, dtype=np.float32)),
                                      f32p(np.ascontiguousarray(succ_values, dtype=np.float32)), f32p(traj["reward"]),
                                      u8p(traj["flag"]), gamma, f32p(out))
    return out


def seq_policy_dlogits(logits, actions, adv, logp0=None, clip=None):
    """logits [2][T][n] -> (dlogits [2][T][n], logp [T][n], loss_sum, entropy_sum); PPO mode when logp0 is given"""
    A, T, n = logits.shape
    B = T * n
    dl = np.zeros((A, T, n), dtype=np.float32)
    lp = np.zeros((T, n), dtype=np.float32)
    loss, ent = C.c_double(), C.c_double()
    mode = 0 if logp0 is None else 1
    lo, hi = (0.0, 0.0) if clip is None else (np.float32(1.0 - clip), np.float32(1.0 + clip))
    lib().oracle_seq_policy_dlogits_f32(B, f32p(np.ascontiguousarray(logits, dtype=np.float32)),
                                        u8p(np.ascontiguousarray(actions, dtype=np.uint8)),
                                        f32p(np.ascontiguousarray(adv, dtype=np.float32)),
                                        f32p(np.ascontiguousarray(logp0, dtype=np.float32)) if mode else None,
                                        mode, lo, hi, f32p(dl), f32p(lp), C.byref(loss), C.byref(ent))
    return dl, lp, loss.value, ent.value


class ChainLaneSim:
    """numpy view of oracle_chain_lanes: Chain (+ step limit) over lanes with the engine's stream discipline
    (slip draw of global step t = word t of the lane's env stream; actor draw = word t of its actor stream)."""

    def __init__(self, n_lanes, max_steps=100, limit=LIMIT_LATENT, lane_offset=0, seed_env=0, seed_actor=1, size=5):
        self.ptr = lib().oracle_chain_lanes_new(size, limit, max_steps, n_lanes, lane_offset, seed_env, seed_actor)
        self.n = n_lanes
        self.D = int(lib().oracle_chain_lanes_obs_dim(self.ptr))

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().oracle_chain_lanes_free(self.ptr)
            self.ptr = None

    def reset(self):
        lib().oracle_chain_lanes_reset(self.ptr)

    def observe(self):
        obs = np.zeros((self.D, self.n), dtype=np.float32)
        lib().oracle_chain_lanes_observe(self.ptr, f32p(obs))
        return obs

    def get_state(self):
        a, b, c = (np.zeros(self.n, dtype=np.uint64) for _ in range(3))
        lib().oracle_chain_lanes_get_state(self.ptr, u64p(a), u64p(b), u64p(c))
        return a, b, c

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.uint8)
        reward = np.zeros(self.n, dtype=np.float32)
        flag = np.zeros(self.n, dtype=np.uint8)
        obs = np.zeros((self.D, self.n), dtype=np.float32)
        term = np.zeros((self.D, self.n), dtype=np.float32)
        lib().oracle_chain_lanes_step(self.ptr, u8p(actions), f32p(reward), u8p(flag), f32p(obs), f32p(term))
        return reward, flag, obs, term

    def rollout_mlp(self, shape, params, T):
        n, D = self.n, self.D
        obs = np.zeros((D, T + 1, n), dtype=np.float32)
        action = np.zeros((T, n), dtype=np.uint8)
        reward = np.zeros((T, n), dtype=np.float32)
        flag = np.zeros((T, n), dtype=np.uint8)
        term = np.zeros((D, T, n), dtype=np.float32)
        lib().oracle_chain_lanes_rollout_mlp(self.ptr, shape, f32p(params), T, f32p(obs), u8p(action), f32p(reward),
                                             u8p(flag), f32p(term))
        return dict(obs=obs, action=action, reward=reward, flag=flag, term_obs=term)

    def rollout_gru(self, shape, params, T, threads=8):
        n, D = self.n, self.D
        obs = np.zeros((D, T + 1, n), dtype=np.float32)
        action = np.zeros((T, n), dtype=np.uint8)
        reward = np.zeros((T, n), dtype=np.float32)
        flag = np.zeros((T, n), dtype=np.uint8)
        term = np.zeros((D, T, n), dtype=np.float32)
        lib().oracle_chain_lanes_rollout_gru(self.ptr, shape, f32p(params), T, f32p(obs), u8p(action), f32p(reward),
                                             u8p(flag), f32p(term), threads)
        return dict(obs=obs, action=action, reward=reward, flag=flag, term_obs=term)


class BanditLaneSim(ChainLaneSim):
    """DeterministicBandit lanes (src/envs/bandits.rs:109-116): reward = the chosen arm's value, every step ends the
    episode; observations are one-hot(5) of the single state."""

    def __init__(self, n_lanes, values=(0.0, 1.0), lane_offset=0, seed_env=0, seed_actor=1):
        v = (C.c_double * 2)(*values)
        self.ptr = lib().oracle_bandit_lanes_new(v, n_lanes, lane_offset, seed_env, seed_actor)
        self.n = n_lanes
        self.D = int(lib().oracle_chain_lanes_obs_dim(self.ptr))


class MemoryLaneSim(ChainLaneSim):
    """MemoryGame lanes (src/envs/memory.rs): the shared index-env lane oracle with `memory.num_actions` set; the
    initial-state draws come sequentially from each lane's env stream."""

    def __init__(self, n_lanes, num_actions=2, history_len=3, max_steps=0, limit=LIMIT_NONE, lane_offset=0, seed_env=0,
                 seed_actor=1):
        self.ptr = lib().oracle_memory_lanes_new(num_actions, history_len, limit, max_steps, n_lanes, lane_offset,
                                                 seed_env, seed_actor)
        self.n = n_lanes
        self.D = int(lib().oracle_chain_lanes_obs_dim(self.ptr))

    def get_state(self):
        """(current, initial, env word position, steps_remaining, reset_count)"""
        cur, rem, rc = ChainLaneSim.get_state(self)
        ini, pos = (np.zeros(self.n, dtype=np.uint64) for _ in range(2))
        lib().oracle_memory_lanes_get_extra(self.ptr, u64p(ini), u64p(pos))
        return cur, ini, pos, rem, rc
