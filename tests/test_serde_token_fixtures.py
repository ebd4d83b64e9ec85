"""The reference's own serialisation fixtures (serde_test token streams, src/torch/serialize.rs:187-352 and
src/spaces/indexed_type.rs:409-422, transcribed in tests/golden/serde_token_fixtures.json) against the C ABI's
TensorDef / IndexedTypeSpace encodings: the bytes the library writes decode to exactly those token sequences, and the
library reads them back (assert_tokens checks both directions).  Host-only entry points: CPU only."""
import json
import os

import numpy as np
import pytest

import relearn_amd as ra
from cbor_ref import decode, encode
from serde_tokens import INDEXED_TYPE_SPACE, TENSOR_DEF, tokens

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "serde_token_fixtures.json")))


@pytest.mark.parametrize("fx", FIX["tensor_def"], ids=[f["source"].split()[-1] for f in FIX["tensor_def"]])
def test_tensor_def_tokens(fx):
    t = fx["tensor"]
    raw = np.asarray(t["values"], dtype=t["dtype"]).tobytes()
    doc = ra.tensor_def_to_cbor(t["kind"], t["shape"], t["requires_grad"], raw)
    assert tokens(decode(doc), TENSOR_DEF) == fx["tokens"]          # Serialize
    assert ra.tensor_def_from_cbor(doc) == (t["kind"], t["shape"], t["requires_grad"], raw)  # Deserialize
    # byte-identical to the independent codec's encoding of the same structure
    assert doc == encode({"kind": t["kind"], "shape": t["shape"], "requires_grad": t["requires_grad"],
                          "byte_order": "LittleEndian", "data": raw})


def test_kind_def_variants_in_declaration_order():
    assert ra.TENSOR_KINDS == FIX["kind_def_variants"]["variants"]
    sizes = [1, 1, 2, 4, 8, 2, 4, 8, 4, 8, 16, 1, 1, 1, 4, 2]  # tch Kind::elt_size_in_bytes
    for kind, size in zip(ra.TENSOR_KINDS, sizes):
        doc = ra.tensor_def_to_cbor(kind, [3], False, bytes(3 * size))
        assert decode(doc)["kind"] == kind
        assert ra.tensor_def_from_cbor(doc) == (kind, [3], False, bytes(3 * size))
        with pytest.raises(ra.RelearnError):
            ra.tensor_def_to_cbor(kind, [3], False, bytes(3 * size + 1))


def test_indexed_type_space_tokens():
    doc = ra.indexed_type_space_to_cbor()
    assert tokens(decode(doc), INDEXED_TYPE_SPACE) == FIX["indexed_type_space"]["tokens"]
    assert doc == bytes([0xA0])


def test_reader_refuses_what_serde_refuses():
    good = {"kind": "Float", "shape": [2], "requires_grad": True, "byte_order": "LittleEndian", "data": bytes(8)}
    assert ra.tensor_def_from_cbor(encode(good)) == ("Float", [2], True, bytes(8))
    for bad in (dict(good, kind="Float32"), dict(good, byte_order="BigEndian"), dict(good, byte_order="Native"),
                dict(good, data=bytes(7)), dict(good, shape=[-2]), dict(good, requires_grad=1),
                dict(good, shape=[1 << 32, 1 << 32], data=b""),  # the element count must not wrap to 0
                dict(good, shape=[1 << 62, 4], data=b""),
                {k: good[k] for k in ("kind", "shape", "requires_grad", "byte_order")}):
        with pytest.raises(ra.RelearnError):
            ra.tensor_def_from_cbor(encode(bad))
    with pytest.raises(ra.RelearnError):
        ra.tensor_def_from_cbor(encode(good)[:-1])
    # a map is keyed, not positional: serde's derived Deserialize takes the fields in any order
    for order in (("shape", "kind", "requires_grad", "byte_order", "data"),
                  ("data", "byte_order", "requires_grad", "shape", "kind")):
        assert ra.tensor_def_from_cbor(encode({k: good[k] for k in order})) == ("Float", [2], True, bytes(8))
