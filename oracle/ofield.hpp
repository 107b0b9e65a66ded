// ORACLE (test infrastructure, never shipped / never linked into the product).
// Plain CPU restatement of Stwo's field tower used by cairo-m's prover.
//
// PARITY UNPINNED for everything restated from Stwo (starkware-libs/stwo @ ab57a1c is an
// un-vendored, empty submodule in the reference tree: /root/reference/.gitmodules:1-3,
// Cargo.toml:46-53).  In-tree evidence for the layout used here:
//   * QM31 = ((a,b),(c,d)) with to_m31_array = [a,b,c,d]:
//       crates/prover/src/public_data.rs:146-149, crates/common/src/execution.rs:59-62
//   * M31 modulus 2^31-1, M31::inverse, -M31::one(): crates/prover/src/adapter/memory.rs:440,
//       crates/prover/src/components/merkle.rs:133
// Arithmetic is deliberately the slow, obviously-correct form (64-bit %), so that it is
// independent of the folded reductions used by the HIP product code.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

constexpr uint32_t P = 0x7fffffffu;  // 2^31 - 1

struct M31 {
  uint32_t v;
  M31() : v(0) {}
  explicit M31(uint32_t x) : v(x % P) {}
  static M31 raw(uint32_t x) { M31 r; r.v = x; return r; }
  static M31 from_i64(int64_t x) {
    int64_t m = x % (int64_t)P;
    if (m < 0) m += P;
    return raw((uint32_t)m);
  }
  M31 operator+(M31 o) const { return raw((uint32_t)(((uint64_t)v + o.v) % P)); }
  M31 operator-(M31 o) const { return raw((uint32_t)(((uint64_t)v + P - o.v) % P)); }
  M31 operator*(M31 o) const { return raw((uint32_t)(((uint64_t)v * o.v) % P)); }
  M31 operator-() const { return raw(v == 0 ? 0 : P - v); }
  M31& operator+=(M31 o) { *this = *this + o; return *this; }
  M31& operator-=(M31 o) { *this = *this - o; return *this; }
  M31& operator*=(M31 o) { *this = *this * o; return *this; }
  bool operator==(M31 o) const { return v == o.v; }
  bool operator!=(M31 o) const { return v != o.v; }
  bool is_zero() const { return v == 0; }
  M31 pow(uint64_t e) const {
    M31 r = raw(1), b = *this;
    while (e) {
      if (e & 1) r = r * b;
      b = b * b;
      e >>= 1;
    }
    return r;
  }
  // Fermat inverse x^(P-2).  inverse(0) is an error in Stwo; the oracle asserts.
  M31 inverse() const {
    assert(v != 0);
    return pow(P - 2);
  }
};

struct CM31 {
  M31 a, b;  // a + b*i, i^2 = -1
  CM31() {}
  CM31(M31 a_, M31 b_) : a(a_), b(b_) {}
  explicit CM31(M31 a_) : a(a_), b() {}
  CM31 operator+(CM31 o) const { return {a + o.a, b + o.b}; }
  CM31 operator-(CM31 o) const { return {a - o.a, b - o.b}; }
  CM31 operator-() const { return {-a, -b}; }
  CM31 operator*(CM31 o) const { return {a * o.a - b * o.b, a * o.b + b * o.a}; }
  CM31 operator*(M31 o) const { return {a * o, b * o}; }
  bool operator==(CM31 o) const { return a == o.a && b == o.b; }
  bool is_zero() const { return a.is_zero() && b.is_zero(); }
  CM31 inverse() const {
    M31 n = a * a + b * b;
    M31 ni = n.inverse();
    return {a * ni, -(b * ni)};
  }
};

// QM31 = CM31[u] / (u^2 - (2 + i))
struct QM31 {
  CM31 a, b;
  QM31() {}
  QM31(CM31 a_, CM31 b_) : a(a_), b(b_) {}
  explicit QM31(M31 x) : a(x), b() {}
  static QM31 from_u32(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    return QM31(CM31(M31(x0), M31(x1)), CM31(M31(x2), M31(x3)));
  }
  static QM31 from_m31s(M31 x0, M31 x1, M31 x2, M31 x3) { return QM31(CM31(x0, x1), CM31(x2, x3)); }
  static QM31 one() { return QM31(M31(1)); }
  static QM31 zero() { return QM31(); }
  void to_u32(uint32_t out[4]) const {
    out[0] = a.a.v; out[1] = a.b.v; out[2] = b.a.v; out[3] = b.b.v;
  }
  M31 coord(int i) const { return i == 0 ? a.a : i == 1 ? a.b : i == 2 ? b.a : b.b; }
  QM31 operator+(QM31 o) const { return {a + o.a, b + o.b}; }
  QM31 operator-(QM31 o) const { return {a - o.a, b - o.b}; }
  QM31 operator-() const { return {-a, -b}; }
  QM31 operator*(QM31 o) const {
    const CM31 R(M31(2), M31(1));
    return {a * o.a + R * (b * o.b), a * o.b + b * o.a};
  }
  QM31 operator*(M31 o) const { return {a * o, b * o}; }
  QM31 operator+(M31 o) const { return {CM31(a.a + o, a.b), b}; }
  QM31 operator-(M31 o) const { return {CM31(a.a - o, a.b), b}; }
  QM31 mul_cm31(CM31 o) const { return {a * o, b * o}; }
  QM31& operator+=(QM31 o) { *this = *this + o; return *this; }
  QM31& operator-=(QM31 o) { *this = *this - o; return *this; }
  QM31& operator*=(QM31 o) { *this = *this * o; return *this; }
  bool operator==(QM31 o) const { return a == o.a && b == o.b; }
  bool operator!=(QM31 o) const { return !(*this == o); }
  bool is_zero() const { return a.is_zero() && b.is_zero(); }
  QM31 complex_conjugate() const { return {a, -b}; }  // a - b*u
  QM31 square() const { return *this * *this; }
  QM31 inverse() const {
    const CM31 R(M31(2), M31(1));
    CM31 den = a * a - R * (b * b);
    CM31 di = den.inverse();
    return {a * di, -(b * di)};
  }
  QM31 pow(uint64_t e) const {
    QM31 r = one(), x = *this;
    while (e) {
      if (e & 1) r = r * x;
      x = x * x;
      e >>= 1;
    }
    return r;
  }
};

// Montgomery batch inversion (FieldExpOps::batch_inverse; used public_data.rs:391).
template <class F>
inline std::vector<F> batch_inverse(const std::vector<F>& xs) {
  std::vector<F> out(xs.size());
  for (size_t i = 0; i < xs.size(); i++) out[i] = xs[i].inverse();
  return out;
}

}  // namespace orc
