"""A small, independent CBOR codec (RFC 8949) for the tests: a generic decoder, and an encoder that follows
serde_cbor 0.11's choices (definite lengths, shortest integers, floats shrunk to the shortest exact width).
Pinned by the RFC's Appendix A examples in tests/test_cbor_format.py."""
import math
import struct


def _head(major, v):
    m = major << 5
    if v < 24:
        return bytes([m | v])
    if v <= 0xFF:
        return bytes([m | 24, v])
    if v <= 0xFFFF:
        return bytes([m | 25]) + struct.pack(">H", v)
    if v <= 0xFFFFFFFF:
        return bytes([m | 26]) + struct.pack(">I", v)
    return bytes([m | 27]) + struct.pack(">Q", v)


def _float(v):
    if math.isinf(v):
        return bytes([0xF9, 0x7C if v > 0 else 0xFC, 0x00])
    if math.isnan(v):
        return bytes([0xF9, 0x7E, 0x00])
    try:
        h = struct.pack(">e", v)
        if struct.unpack(">e", h)[0] == v:
            return b"\xf9" + h
    except (OverflowError, struct.error):
        pass
    f = struct.pack(">f", v) if abs(v) < 3.5e38 else None
    if f is not None and struct.unpack(">f", f)[0] == v:
        return b"\xfa" + f
    return b"\xfb" + struct.pack(">d", v)


def encode(x):
    """dict -> map (insertion order), list/tuple -> array, str -> text, bytes -> byte string, None -> null"""
    if x is None:
        return b"\xf6"
    if x is True:
        return b"\xf5"
    if x is False:
        return b"\xf4"
    if isinstance(x, int):
        return _head(0, x) if x >= 0 else _head(1, -1 - x)
    if isinstance(x, float):
        return _float(x)
    if isinstance(x, str):
        b = x.encode()
        return _head(3, len(b)) + b
    if isinstance(x, (bytes, bytearray)):
        return _head(2, len(x)) + bytes(x)
    if isinstance(x, (list, tuple)):
        return _head(4, len(x)) + b"".join(encode(i) for i in x)
    if isinstance(x, dict):
        return _head(5, len(x)) + b"".join(encode(k) + encode(v) for k, v in x.items())
    raise TypeError(type(x))


def decode(data):
    v, pos = _dec(memoryview(bytes(data)), 0)
    assert pos == len(data), "trailing bytes"
    return v


def _arg(b, pos, info):
    if info < 24:
        return info, pos
    n = {24: 1, 25: 2, 26: 4, 27: 8}[info]
    return int.from_bytes(b[pos:pos + n], "big"), pos + n


def _dec(b, pos):
    ib = b[pos]
    pos += 1
    major, info = ib >> 5, ib & 31
    if major == 7:
        if info == 20:
            return False, pos
        if info == 21:
            return True, pos
        if info == 22:
            return None, pos
        if info == 25:
            return struct.unpack(">e", b[pos:pos + 2])[0], pos + 2
        if info == 26:
            return struct.unpack(">f", b[pos:pos + 4])[0], pos + 4
        if info == 27:
            return struct.unpack(">d", b[pos:pos + 8])[0], pos + 8
        raise ValueError("simple value %d" % info)
    v, pos = _arg(b, pos, info)
    if major == 0:
        return v, pos
    if major == 1:
        return -1 - v, pos
    if major == 2:
        return bytes(b[pos:pos + v]), pos + v
    if major == 3:
        return bytes(b[pos:pos + v]).decode(), pos + v
    if major == 4:
        out = []
        for _ in range(v):
            x, pos = _dec(b, pos)
            out.append(x)
        return out, pos
    if major == 5:
        out = {}
        for _ in range(v):
            k, pos = _dec(b, pos)
            x, pos = _dec(b, pos)
            out[k] = x
        return out, pos
    raise ValueError("major type %d" % major)
