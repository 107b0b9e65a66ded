// ORACLE (test infrastructure).  Circle group, cosets, canonic domains, bit reversal.
// Restated from Stwo `core::circle`, `core::poly::circle::{canonic,domain}`, `core::utils`
// (PARITY UNPINNED — Stwo @ ab57a1c is not vendored; see ofield.hpp).
// In-tree call sites this follows: `CanonicCoset::new(log).circle_domain().half_coset`
// (crates/prover/src/prover.rs:56-60), `CanonicCoset::new(n).circle_domain()` for every trace
// column (crates/prover/src/preprocessed/range_check/mod.rs:62-67).
#pragma once
#include "ofield.hpp"

namespace orc {

template <class F>
struct CirclePoint {
  F x, y;
  CirclePoint operator+(const CirclePoint& o) const {
    return {x * o.x - y * o.y, x * o.y + y * o.x};
  }
  CirclePoint conjugate() const { return {x, -y}; }
  CirclePoint neg() const { return conjugate(); }
  CirclePoint operator-(const CirclePoint& o) const { return *this + o.conjugate(); }
  CirclePoint dbl() const { return *this + *this; }
};
using PointM = CirclePoint<M31>;
using PointQ = CirclePoint<QM31>;

inline PointQ into_ef(PointM p) { return {QM31(p.x), QM31(p.y)}; }
inline M31 double_x(M31 x) { return x * x + x * x - M31(1); }
inline QM31 double_x(QM31 x) { return x * x + x * x - M31(1); }

constexpr uint32_t CIRCLE_LOG_ORDER = 31;
inline PointM circle_gen() { return {M31(2), M31(1268011823u)}; }

// index -> G^index (index taken mod 2^31)
inline PointM point_at_index(uint32_t idx) {
  idx &= 0x7fffffffu;
  PointM res{M31(1), M31(0)};
  PointM cur = circle_gen();
  while (idx) {
    if (idx & 1) res = res + cur;
    cur = cur.dbl();
    idx >>= 1;
  }
  return res;
}
inline uint32_t subgroup_gen_index(uint32_t log_size) { return 1u << (CIRCLE_LOG_ORDER - log_size); }

struct Coset {
  uint32_t initial_index;  // mod 2^31
  uint32_t step_size;      // mod 2^31
  uint32_t log_size;
  static Coset make(uint32_t initial_index, uint32_t log_size) {
    return {initial_index & 0x7fffffffu, subgroup_gen_index(log_size), log_size};
  }
  static Coset subgroup(uint32_t log_size) { return make(0, log_size); }
  static Coset odds(uint32_t log_size) { return make(subgroup_gen_index(log_size + 1), log_size); }
  static Coset half_odds(uint32_t log_size) { return make(subgroup_gen_index(log_size + 2), log_size); }
  size_t size() const { return (size_t)1 << log_size; }
  uint32_t index_at(size_t i) const {
    return (uint32_t)((initial_index + (uint64_t)step_size * i) & 0x7fffffffu);
  }
  PointM at(size_t i) const { return point_at_index(index_at(i)); }
  PointM initial() const { return point_at_index(initial_index); }
  PointM step() const { return point_at_index(step_size); }
  Coset dbl() const {
    return {(initial_index * 2) & 0x7fffffffu, (step_size * 2) & 0x7fffffffu,
            log_size ? log_size - 1 : 0};
  }
};

struct CircleDomain {
  Coset half_coset;
  uint32_t log_size() const { return half_coset.log_size + 1; }
  size_t size() const { return (size_t)1 << log_size(); }
  uint32_t index_at(size_t i) const {
    size_t h = half_coset.size();
    if (i < h) return half_coset.index_at(i);
    return (0x80000000u - half_coset.index_at(i - h)) & 0x7fffffffu;
  }
  PointM at(size_t i) const { return point_at_index(index_at(i)); }
};

struct CanonicCoset {
  Coset coset;
  explicit CanonicCoset(uint32_t log_size) : coset(Coset::odds(log_size)) { assert(log_size > 0); }
  uint32_t log_size() const { return coset.log_size; }
  Coset half_coset() const { return Coset::half_odds(coset.log_size - 1); }
  CircleDomain circle_domain() const { return {half_coset()}; }
  uint32_t step_size() const { return coset.step_size; }
  PointM step() const { return coset.step(); }
};

inline size_t bit_reverse_index(size_t i, uint32_t log_size) {
  if (log_size == 0) return i;
  size_t r = 0;
  for (uint32_t b = 0; b < log_size; b++) r |= ((i >> b) & 1) << (log_size - 1 - b);
  return r;
}
template <class T>
inline void bit_reverse(std::vector<T>& v) {
  size_t n = v.size();
  uint32_t log = 0;
  while (((size_t)1 << log) < n) log++;
  for (size_t i = 0; i < n; i++) {
    size_t j = bit_reverse_index(i, log);
    if (j > i) std::swap(v[i], v[j]);
  }
}

// Stwo `coset_index_to_circle_domain_index` / `circle_domain_index_to_coset_index`.
inline size_t coset_index_to_circle_domain_index(size_t coset_index, uint32_t log_domain_size) {
  if (coset_index % 2 == 0) return coset_index / 2;
  return (((size_t)2 << log_domain_size) - coset_index) / 2;
}

// Vanishing polynomial of a coset evaluated at p (Stwo `constraints::coset_vanishing`).
template <class F>
inline F coset_vanishing(Coset coset, CirclePoint<F> p, CirclePoint<F> (*lift)(PointM)) {
  // p - initial + step/2
  PointM shift = point_at_index((0x80000000u - coset.initial_index + (coset.step_size >> 1)) & 0x7fffffffu);
  p = p + lift(shift);
  F x = p.x;
  for (uint32_t i = 1; i < coset.log_size; i++) x = double_x(x);
  return x;
}
inline PointM lift_m(PointM p) { return p; }

}  // namespace orc
