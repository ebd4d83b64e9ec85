"""Actor serialisation through the C ABI (rl_actor_to_cbor / rl_module_from_cbor): the CBOR document has the
structure serde derives for PolicyActor / DqnActor / Mlp / Chain<Gru, Mlp> / TensorDef and is byte-identical to an
independent encoding of the same structure (tests/cbor_ref.py, pinned by RFC 8949 vectors); loading it back restores
the parameters bit for bit."""
import numpy as np
import pytest

from cbor_ref import decode, encode

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

F64_MIN, F64_MAX = -1.7976931348623157e308, 1.7976931348623157e308
MAX_ANGLE = 12.0 * (np.pi / 180.0)  # 12.0f64.to_radians() (CartPole::default, envs/cartpole.rs:203-216)


def tensor(arr):
    return {"kind": "Float", "shape": list(arr.shape), "requires_grad": True, "byte_order": "LittleEndian",
            "data": np.ascontiguousarray(arr, dtype="<f4").tobytes()}


def mlp_doc(p, i, h, o):
    k1, b1 = p[:h * i].reshape(h, i), p[h * i:h * i + h]
    k2, b2 = p[h * i + h:h * i + h + o * h].reshape(o, h), p[h * i + h + o * h:]
    return {"layers": [{"kernel": tensor(k1), "bias": tensor(b1)}, {"kernel": tensor(k2), "bias": tensor(b2)}],
            "activation": "Relu", "output_activation": "Identity"}


def cartpole_space(visible):
    inner = {"cart_position": {"low": -2.4, "high": 2.4}, "cart_velocity": {"low": F64_MIN, "high": F64_MAX},
             "pole_angle": {"low": -MAX_ANGLE, "high": MAX_ANGLE},
             "pole_angular_velocity": {"low": F64_MIN, "high": F64_MAX}}
    return {"inner": {"inner": inner, "remaining": {"low": 0.0, "high": 1.0}} if visible else inner}


@pytest.mark.parametrize("visible", [True, False])
def test_policy_actor_document(engine, visible):
    limit = ra.LIMIT_VISIBLE if visible else ra.LIMIT_NONE
    env = ra.CartPoleEnv(engine, 64, limit=limit)
    D = 5 if visible else 4
    pol = ra.Mlp(engine, D, 128, 2)
    pol.init(5)
    data = ra.actor_to_cbor(env, pol)
    doc = decode(data)
    assert list(doc) == ["observation_space", "action_space", "policy_module"]
    assert doc["action_space"] == {}
    want = {"observation_space": cartpole_space(visible), "action_space": {},
            "policy_module": mlp_doc(pol.get_params(), D, 128, 2)}
    assert doc["observation_space"]["inner"].keys() == want["observation_space"]["inner"].keys()
    assert data == encode(want)
    # serde_cbor's float shrinking: 0.0 / 1.0 as half floats, 2.4 and f64::MAX as doubles
    if visible:
        assert encode({"low": 0.0, "high": 1.0}) in data and bytes.fromhex("636c6f77f90000") in data
    assert bytes.fromhex("fb7fefffffffffffff") in data and bytes.fromhex("fb4003333333333333") in data
    # round trip into a fresh module
    other = ra.Mlp(engine, D, 128, 2)
    other.init(99)
    ra.module_from_cbor(other, data)
    assert np.array_equal(other.get_params(), pol.get_params())
    # a module of another shape is refused
    small = ra.Mlp(engine, D, 64, 2)
    with pytest.raises(ra.RelearnError) as e:
        ra.module_from_cbor(small, data)
    assert e.value.code == ra.ERR_INVALID_ARGUMENT


def test_dqn_actor_document(engine):
    env = ra.CartPoleEnv(engine, 64)
    q = ra.Mlp(engine, 5, 128, 2)
    q.init(8)
    data = ra.actor_to_cbor(env, q, ra.ACTOR_DQN, exploration_rate=0.0)
    doc = decode(data)
    assert list(doc) == ["observation_space", "action_space", "action_value_fn", "exploration_rate"]
    assert doc["exploration_rate"] == 0.0 and data.endswith(bytes.fromhex("f90000"))
    assert ra.actor_to_cbor(env, q, ra.ACTOR_DQN, exploration_rate=0.1).endswith(bytes.fromhex("fb3fb999999999999a"))
    other = ra.Mlp(engine, 5, 128, 2)
    ra.module_from_cbor(other, data)
    assert np.array_equal(other.get_params(), q.get_params())


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_recurrent_policy_actor_document(engine, cell):
    """Gru and Lstm are both RnnBase<impl> (seq/rnn/gru.rs:17, lstm.rs:12): one document shape, 3H or 4H gate rows"""
    env = ra.ChainEnv(engine, 64, max_steps=100)
    make = ra.GruMlp if cell == "gru" else ra.LstmMlp
    G = 3 if cell == "gru" else 4
    pol = make(engine, 5, 2)
    pol.init(13)
    p = pol.get_params()
    H, D = 128, 5
    o = [G * H * D, G * H * H, G * H, G * H]
    c = np.cumsum([0] + o)
    flat = [tensor(p[c[0]:c[1]].reshape(G * H, D)), tensor(p[c[1]:c[2]].reshape(G * H, H)), tensor(p[c[2]:c[3]]),
            tensor(p[c[3]:c[4]])]
    want = {"observation_space": {"inner": {"size": 5}}, "action_space": {},
            "policy_module": {"first": {"weights": {"flat_weights": flat, "has_biases": True}, "hidden_size": 128,
                                        "dropout": 0.0, "type_": None},
                              "second": mlp_doc(p[c[4]:], 128, 128, 2), "activation": "Relu"}}
    data = ra.actor_to_cbor(env, pol)
    assert data == encode(want)
    other = make(engine, 5, 2)
    other.init(1)
    ra.module_from_cbor(other, data)
    assert np.array_equal(other.get_params(), p)
    with pytest.raises(ra.RelearnError):
        ra.module_from_cbor(other, data[:-10])  # truncated document
    wrong = (ra.LstmMlp if cell == "gru" else ra.GruMlp)(engine, 5, 2)
    with pytest.raises(ra.RelearnError):
        ra.module_from_cbor(wrong, data)  # the other cell's gate blocks do not fit


def test_actor_document_leaves_match_the_reference_token_fixtures(engine):
    """every TensorDef of an actor document and its action space tokenise the way the reference's serde_test
    fixtures do (tests/golden/serde_token_fixtures.json; src/torch/serialize.rs:269-302, indexed_type.rs:409-422)"""
    import json
    import os

    from serde_tokens import INDEXED_TYPE_SPACE, TENSOR_DEF, tokens
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "serde_token_fixtures.json")))
    want = [f for f in fix["tensor_def"] if "1d_f32_tensor_requires_grad" in f["source"]][0]["tokens"]
    env = ra.CartPoleEnv(engine, 64)
    pol = ra.Mlp(engine, 5, 128, 2)
    pol.init(5)
    doc = decode(ra.actor_to_cbor(env, pol))
    assert tokens(doc["action_space"], INDEXED_TYPE_SPACE) == fix["indexed_type_space"]["tokens"]
    n = 0
    for layer in doc["policy_module"]["layers"]:
        for name in ("kernel", "bias"):
            got = tokens(layer[name], TENSOR_DEF)
            # same token kinds in the same order; kind / requires_grad / byte_order tokens identical to the fixture's
            strip = lambda ts: [t for t in ts if t[0] not in ("I64", "BorrowedBytes", "Seq")]
            assert strip(got) == strip(want)
            # and the leaf writer of the ABI produces the same bytes for the same tensor
            t = layer[name]
            assert encode(t) == ra.tensor_def_to_cbor(t["kind"], t["shape"], t["requires_grad"], t["data"])
            n += 1
    assert n == 4
