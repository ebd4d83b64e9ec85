// cbor.hpp — the subset of CBOR (RFC 8949) that serde_cbor 0.11 emits for relearn's serialised actors, and a reader for
// the same subset.
//
// Reference: actors are written with `serde_cbor::to_writer(file, &agent.actor(ActorMode::Evaluation))`
// (examples/cartpole-trpo.rs:71-76) and read back with `serde_cbor::from_reader` (:82-89).  With serde's derived
// impls and serde_cbor's default (non-packed) mode:
//   struct            -> definite-length map, keys = field names as text strings, in declaration order
//   unit enum variant -> text string of the variant name
//   Vec / slice       -> definite-length array;  bytes (serde_with::Bytes) -> byte string
//   usize / i64       -> shortest integer encoding;  bool -> true / false;  None, PhantomData -> null
//   f64 / f32         -> the SHORTEST of f16 / f32 / f64 that represents the value exactly (serde_cbor's
//                        serialize_f64 / serialize_f32), infinities and NaN as f16
// serde_cbor itself is a third-party crate whose source is not under /root/reference: the rules above are its
// published behaviour, restated; byte-compatibility with a real relearn build could not be exercised here (no Rust
// toolchain), which DESIGN.md records.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace cbor {

class Writer {
 public:
  std::vector<uint8_t> out;

  void head(int major, uint64_t v) {
    const uint8_t m = (uint8_t)(major << 5);
    if (v < 24) {
      out.push_back(m | (uint8_t)v);
    } else if (v <= 0xff) {
      out.push_back(m | 24);
      out.push_back((uint8_t)v);
    } else if (v <= 0xffff) {
      out.push_back(m | 25);
      be(v, 2);
    } else if (v <= 0xffffffffull) {
      out.push_back(m | 26);
      be(v, 4);
    } else {
      out.push_back(m | 27);
      be(v, 8);
    }
  }
  void uint(uint64_t v) { head(0, v); }
  void sint(int64_t v) {
    if (v >= 0) head(0, (uint64_t)v);
    else head(1, (uint64_t)(-(v + 1)));
  }
  void text(const char *s) {
    const size_t n = std::strlen(s);
    head(3, n);
    out.insert(out.end(), s, s + n);
  }
  void bytes(const void *p, size_t n) {
    head(2, n);
    const uint8_t *b = (const uint8_t *)p;
    out.insert(out.end(), b, b + n);
  }
  void array(uint64_t n) { head(4, n); }
  void map(uint64_t n) { head(5, n); }
  void boolean(bool b) { out.push_back(b ? 0xf5 : 0xf4); }
  void null() { out.push_back(0xf6); }
  void key(const char *s) { text(s); }

  // serde_cbor::Serializer::serialize_f64 / serialize_f32
  void f64(double v) {
    if (!std::isfinite(v) || (double)(float)v == v) {
      f32((float)v);
      return;
    }
    out.push_back(0xfb);
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    be(bits, 8);
  }
  void f32(float v) {
    if (std::isinf(v)) {
      out.push_back(0xf9);
      out.push_back(v > 0 ? 0x7c : 0xfc);
      out.push_back(0x00);
      return;
    }
    if (std::isnan(v)) {
      out.push_back(0xf9);
      out.push_back(0x7e);
      out.push_back(0x00);
      return;
    }
    uint16_t h;
    if (to_half_exact(v, &h)) {
      out.push_back(0xf9);
      be(h, 2);
      return;
    }
    out.push_back(0xfa);
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    be(bits, 4);
  }

 private:
  void be(uint64_t v, int n) {
    for (int i = n - 1; i >= 0; --i) out.push_back((uint8_t)(v >> (8 * i)));
  }
  // f32 -> f16 when the conversion is exact (what `f32::from(f16::from_f32(v)) == v` tests)
  static bool to_half_exact(float v, uint16_t *h) {
    uint32_t b;
    std::memcpy(&b, &v, 4);
    const uint32_t sign = (b >> 16) & 0x8000u, exp = (b >> 23) & 0xffu, man = b & 0x7fffffu;
    if (exp == 0 && man == 0) {
      *h = (uint16_t)sign;
      return true;
    }
    const int e = (int)exp - 127;
    if (e >= -14 && e <= 15) {  // normal half
      if (man & 0x1fffu) return false;
      *h = (uint16_t)(sign | (uint32_t)((e + 15) << 10) | (man >> 13));
      return true;
    }
    if (e >= -24 && e < -14) {  // subnormal half: value = m * 2^-24
      const uint32_t full = man | 0x800000u;
      const int shift = 13 + (-14 - e);
      if (full & ((1u << shift) - 1)) return false;
      *h = (uint16_t)(sign | (full >> shift));
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------- reader (tree of values)
struct Value;
using ValuePtr = std::shared_ptr<Value>;
struct Value {
  enum Kind { UINT, NINT, BYTES, TEXT, ARRAY, MAP, BOOL, NIL, FLOAT } kind = NIL;
  uint64_t u = 0;  // UINT value; NINT: -1 - u
  double f = 0.0;
  bool b = false;
  std::string s;  // TEXT, or BYTES payload
  std::vector<ValuePtr> items;
  std::vector<std::pair<std::string, ValuePtr>> fields;  // MAP with text keys, in document order

  const Value &at(const char *key) const {
    if (kind != MAP) throw std::runtime_error(std::string("CBOR: expected a map to look up '") + key + "'");
    for (auto &kv : fields)
      if (kv.first == key) return *kv.second;
    throw std::runtime_error(std::string("CBOR: missing field '") + key + "'");
  }
  bool has(const char *key) const {
    if (kind != MAP) return false;
    for (auto &kv : fields)
      if (kv.first == key) return true;
    return false;
  }
  int64_t as_int() const {
    if (kind == UINT) return (int64_t)u;
    if (kind == NINT) return -1 - (int64_t)u;
    throw std::runtime_error("CBOR: expected an integer");
  }
  double as_float() const {
    if (kind == FLOAT) return f;
    if (kind == UINT || kind == NINT) return (double)as_int();
    throw std::runtime_error("CBOR: expected a number");
  }
};

class Reader {
 public:
  Reader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
  ValuePtr parse() {
    ValuePtr v = value(0);
    if (pos_ != n_) throw std::runtime_error("CBOR: trailing bytes after the document");
    return v;
  }

 private:
  const uint8_t *p_;
  size_t n_, pos_ = 0;
  uint8_t byte() {
    if (pos_ >= n_) throw std::runtime_error("CBOR: truncated document");
    return p_[pos_++];
  }
  uint64_t be(int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; ++i) v = (v << 8) | byte();
    return v;
  }
  uint64_t arg(uint8_t info) {
    if (info < 24) return info;
    if (info == 24) return be(1);
    if (info == 25) return be(2);
    if (info == 26) return be(4);
    if (info == 27) return be(8);
    throw std::runtime_error("CBOR: indefinite lengths are not part of the supported subset");
  }
  static double half_to_double(uint16_t h) {
    const int sign = h >> 15, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    double v;
    if (exp == 0) v = std::ldexp((double)man, -24);
    else if (exp == 31) v = man ? NAN : INFINITY;
    else v = std::ldexp((double)(man | 0x400), exp - 25);
    return sign ? -v : v;
  }
  ValuePtr value(int depth) {
    if (depth > 32) throw std::runtime_error("CBOR: nesting too deep");
    auto v = std::make_shared<Value>();
    const uint8_t ib = byte();
    const int major = ib >> 5;
    const uint8_t info = ib & 31;
    switch (major) {
      case 0: v->kind = Value::UINT; v->u = arg(info); break;
      case 1: v->kind = Value::NINT; v->u = arg(info); break;
      case 2:
      case 3: {
        const uint64_t len = arg(info);
        if (len > n_ - pos_) throw std::runtime_error("CBOR: string runs past the end of the document");
        v->kind = major == 2 ? Value::BYTES : Value::TEXT;
        v->s.assign((const char *)p_ + pos_, (size_t)len);
        pos_ += (size_t)len;
        break;
      }
      case 4: {
        const uint64_t len = arg(info);
        if (len > n_ - pos_) throw std::runtime_error("CBOR: array longer than the document");
        v->kind = Value::ARRAY;
        for (uint64_t i = 0; i < len; ++i) v->items.push_back(value(depth + 1));
        break;
      }
      case 5: {
        const uint64_t len = arg(info);
        if (len > n_ - pos_) throw std::runtime_error("CBOR: map longer than the document");
        v->kind = Value::MAP;
        for (uint64_t i = 0; i < len; ++i) {
          ValuePtr k = value(depth + 1);
          if (k->kind != Value::TEXT) throw std::runtime_error("CBOR: map keys must be text strings");
          v->fields.emplace_back(k->s, value(depth + 1));
        }
        break;
      }
      case 7:
        if (info == 20 || info == 21) {
          v->kind = Value::BOOL;
          v->b = info == 21;
        } else if (info == 22) {
          v->kind = Value::NIL;
        } else if (info == 25) {
          v->kind = Value::FLOAT;
          v->f = half_to_double((uint16_t)be(2));
        } else if (info == 26) {
          uint32_t bits = (uint32_t)be(4);
          float f;
          std::memcpy(&f, &bits, 4);
          v->kind = Value::FLOAT;
          v->f = f;
        } else if (info == 27) {
          uint64_t bits = be(8);
          std::memcpy(&v->f, &bits, 8);
          v->kind = Value::FLOAT;
        } else {
          throw std::runtime_error("CBOR: unsupported simple value");
        }
        break;
      default: throw std::runtime_error("CBOR: tags are not part of the supported subset");
    }
    return v;
  }
};

}  // namespace cbor
