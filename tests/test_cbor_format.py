"""The test-side CBOR codec against RFC 8949 Appendix A known-answer vectors (the C-ABI's serialiser is compared
with this codec byte for byte in tests/test_gpu_cbor.py).  CPU only."""
import math

import pytest

from cbor_ref import decode, encode

# (value, hex) pairs from RFC 8949 Appendix A
VECTORS = [
    (0, "00"), (1, "01"), (10, "0a"), (23, "17"), (24, "1818"), (25, "1819"), (100, "1864"), (1000, "1903e8"),
    (1000000, "1a000f4240"), (1000000000000, "1b000000e8d4a51000"), (18446744073709551615, "1bffffffffffffffff"),
    (-1, "20"), (-10, "29"), (-100, "3863"), (-1000, "3903e7"),
    (0.0, "f90000"), (1.0, "f93c00"), (1.1, "fb3ff199999999999a"), (1.5, "f93e00"), (65504.0, "f97bff"),
    (100000.0, "fa47c35000"), (3.4028234663852886e+38, "fa7f7fffff"), (1.0e+300, "fb7e37e43c8800759c"),
    (5.960464477539063e-8, "f90001"), (0.00006103515625, "f90400"), (-4.0, "f9c400"), (-4.1, "fbc010666666666666"),
    (math.inf, "f97c00"), (-math.inf, "f9fc00"),
    (False, "f4"), (True, "f5"), (None, "f6"),
    (b"", "40"), (bytes([1, 2, 3, 4]), "4401020304"),
    ("", "60"), ("a", "6161"), ("IETF", "6449455446"),
    ([], "80"), ([1, 2, 3], "83010203"), ([1, [2, 3], [4, 5]], "8301820203820405"),
    (list(range(1, 26)), "98190102030405060708090a0b0c0d0e0f101112131415161718181819"),
    ({}, "a0"), ({"a": 1, "b": [2, 3]}, "a26161016162820203"), (["a", {"b": "c"}], "826161a161626163"),
    ({"a": "A", "b": "B", "c": "C", "d": "D", "e": "E"}, "a56161614161626142616361436164614461656145"),
]


@pytest.mark.parametrize("value,hexstr", VECTORS, ids=[h for _, h in VECTORS])
def test_rfc8949_appendix_a(value, hexstr):
    assert encode(value).hex() == hexstr
    back = decode(bytes.fromhex(hexstr))
    assert back == value and type(back) is type(value)


def test_nan_and_negative_zero():
    assert encode(math.nan).hex() == "f97e00" and math.isnan(decode(bytes.fromhex("f97e00")))
    assert encode(-0.0).hex() == "f98000"
    assert encode(1.7976931348623157e308).hex() == "fb7fefffffffffffff"  # f64::MAX stays 64-bit
    assert encode(2.4).hex().startswith("fb")
