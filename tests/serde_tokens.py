"""serde data-model view of a CBOR document: turns a decoded document (tests/cbor_ref.py) into the serde_test token
stream the reference's `assert_tokens` fixtures are written in (tests/golden/serde_token_fixtures.json).

CBOR carries no Rust type names, so the caller passes a schema naming them — serde_cbor's (non-packed) mapping is
  struct -> map keyed by field name in declaration order      (Token::Struct{name, len} Str(field) ... StructEnd)
  unit variant -> text string of the variant name              (Token::UnitVariant{name, variant})
  Vec / slice -> definite array                                (Token::Seq{len} ... SeqEnd)
  serde_with::Bytes -> byte string                             (Token::BorrowedBytes)
  i64 / usize -> integer, bool -> bool, f64 -> float."""

TENSOR_DEF = ("struct", "TensorDef", [("kind", ("enum", "KindDef")), ("shape", ("seq", ("i64",))),
                                      ("requires_grad", ("bool",)), ("byte_order", ("enum", "ByteOrder")),
                                      ("data", ("bytes",))])
INDEXED_TYPE_SPACE = ("struct", "IndexedTypeSpace", [])


def tokens(value, schema):
    kind = schema[0]
    if kind == "struct":
        _, name, fields = schema
        assert isinstance(value, dict), (name, type(value))
        assert list(value) == [f for f, _ in fields], "%s: fields %r, expected %r" % (name, list(value), fields)
        out = [["Struct", {"name": name, "len": len(fields)}]]
        for f, sub in fields:
            out.append(["Str", f])
            out += tokens(value[f], sub)
        return out + [["StructEnd"]]
    if kind == "enum":
        assert isinstance(value, str), (schema, value)
        return [["UnitVariant", {"name": schema[1], "variant": value}]]
    if kind == "seq":
        assert isinstance(value, list)
        out = [["Seq", {"len": len(value)}]]
        for v in value:
            out += tokens(v, schema[1])
        return out + [["SeqEnd"]]
    if kind == "i64":
        assert isinstance(value, int) and not isinstance(value, bool)
        return [["I64", value]]
    if kind == "bool":
        assert isinstance(value, bool)
        return [["Bool", value]]
    if kind == "bytes":
        assert isinstance(value, (bytes, bytearray))
        return [["BorrowedBytes", list(value)]]
    raise ValueError(schema)
