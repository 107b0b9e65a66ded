// ORACLE (test infrastructure).  Circle FFT on canonic domains, bit-reversed evaluations.
// Restates Stwo `core::backend::cpu::circle::{interpolate, evaluate, eval_at_point}` and
// `core::poly::circle::poly` (PARITY UNPINNED, Stwo not vendored).  Reached in the reference from
// `tree_builder.extend_evals` / `.commit` (crates/prover/src/prover.rs:71-73, 80-82, 100-102).
//
// Basis: coefficient index bit 0 <-> y, bit 1 <-> x, bit k>=2 <-> pi^{k-1}(x), pi(x)=2x^2-1.
// Twiddles are computed from the group law (no shared tables with the product) and memoised per
// (domain log size, layer) so that the many columns of one proof do not redo the same point walk.
#pragma once
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <map>
#include <memory>
#include <mutex>
#include "ocircle.hpp"

namespace orc {

// Twiddles of layer `layer` (0 = circle/y layer, i>=1 = line layers) for the canonic domain of
// log size n:  tw[h] for butterfly group h.
inline std::vector<M31> compute_layer_twiddles(uint32_t n, uint32_t layer) {
  Coset half = CanonicCoset(n).half_coset();  // size 2^(n-1)
  if (layer == 0) {
    size_t cnt = (size_t)1 << (n - 1);
    std::vector<M31> t(cnt);
    // sequential walk of the coset, then bit-reverse
    PointM p = half.initial(), s = half.step();
    std::vector<M31> nat(cnt);
    for (size_t j = 0; j < cnt; j++) { nat[j] = p.y; p = p + s; }
    for (size_t h = 0; h < cnt; h++) t[h] = nat[bit_reverse_index(h, n - 1)];
    return t;
  }
  Coset c = half;
  for (uint32_t i = 1; i < layer; i++) c = c.dbl();
  // coset c has size 2^(n-layer); first half, bit-reversed
  size_t cnt = (size_t)1 << (n - 1 - layer);
  std::vector<M31> nat(cnt), t(cnt);
  PointM p = c.initial(), s = c.step();
  for (size_t j = 0; j < cnt; j++) { nat[j] = p.x; p = p + s; }
  for (size_t h = 0; h < cnt; h++) t[h] = nat[bit_reverse_index(h, n - 1 - layer)];
  return t;
}

// memoised (n, layer, inverse?) -> table; entries are never evicted (a few tens of MB for the test sizes)
inline const std::vector<M31>& layer_twiddles(uint32_t n, uint32_t layer, bool inverse = false) {
  static std::mutex mu;
  static std::map<uint64_t, std::unique_ptr<std::vector<M31>>> cache;
  const uint64_t key = ((uint64_t)n << 33) | ((uint64_t)layer << 1) | (inverse ? 1u : 0u);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return *it->second;
  std::unique_ptr<std::vector<M31>> t(new std::vector<M31>(compute_layer_twiddles(n, layer)));
  if (inverse) {
    // Montgomery's trick: one field inversion per table instead of one per entry (the tables of one proof hold ~2^23 entries;
    // per-entry inversions were 1.5 s of the first proof of a process, single-threaded under this mutex)
    std::vector<M31>& v = *t;
    bool any_zero = false;
    for (auto& x : v) any_zero |= x.v == 0;
    if (any_zero || v.size() < 2) { for (auto& x : v) x = x.inverse(); }
    else {
      std::vector<M31> pre(v.size());
      M31 acc = M31(1);
      for (size_t i = 0; i < v.size(); i++) { pre[i] = acc; acc = acc * v[i]; }
      M31 inv = acc.inverse();
      for (size_t i = v.size(); i-- > 0;) { const M31 x = v[i]; v[i] = inv * pre[i]; inv = inv * x; }
    }
  }
  return *(cache[key] = std::move(t));
}

// threads inside ONE transform: only outside a parallel region (the callers loop over columns in parallel when there are many)
inline bool fft_inner_parallel(size_t N) {
#ifdef _OPENMP
  return N >= ((size_t)1 << 15) && !omp_in_parallel() && omp_get_max_threads() > 1;
#else
  (void)N;
  return false;
#endif
}
// a layer of 2^20 butterflies is ~0.3 ms of work for 16 threads: larger teams spend longer forming than computing
inline int fft_inner_threads() {
#ifdef _OPENMP
  return std::min(omp_get_max_threads(), 16);
#else
  return 1;
#endif
}
// values: bit-reversed evaluations on CanonicCoset(n).circle_domain(); returns coefficients.
inline std::vector<M31> interpolate(std::vector<M31> values) {
  size_t N = values.size();
  uint32_t n = 0;
  while (((size_t)1 << n) < N) n++;
  assert(n >= 1);
  // a lone large column (composition, preprocessed and FRI trees: 4-10 columns) is transformed by all threads, butterfly pairs of a
  // layer split among them; inside a loop over many columns (the callers' own parallel regions) the layers stay sequential
  const bool par = fft_inner_parallel(N);
  M31* const v = values.data();
  for (uint32_t layer = 0; layer < n; layer++) {
    const M31* const tw = layer_twiddles(n, layer, true).data();
    const size_t stride = (size_t)1 << layer;
#pragma omp parallel for schedule(static) if (par) num_threads(fft_inner_threads())
    for (size_t p = 0; p < N / 2; p++) {
      const size_t h = p >> layer, l = p & (stride - 1);
      const size_t i0 = (h << (layer + 1)) + l, i1 = i0 + stride;
      M31 a = v[i0], b = v[i1];
      v[i0] = a + b;
      v[i1] = (a - b) * tw[h];
    }
  }
  const M31 inv = M31((uint32_t)N).inverse();
#pragma omp parallel for schedule(static) if (par) num_threads(fft_inner_threads())
  for (size_t i = 0; i < N; i++) v[i] = v[i] * inv;
  return values;
}

// coeffs of log size m; evaluate on CanonicCoset(n).circle_domain(), n >= m. Bit-reversed output.
inline std::vector<M31> evaluate(const std::vector<M31>& coeffs, uint32_t n) {
  size_t N = (size_t)1 << n;
  std::vector<M31> values(N);
  for (size_t i = 0; i < coeffs.size(); i++) values[i] = coeffs[i];
  const bool par = fft_inner_parallel(N);
  M31* const v = values.data();
  for (int layer = (int)n - 1; layer >= 0; layer--) {
    const M31* const tw = layer_twiddles(n, (uint32_t)layer).data();
    const size_t stride = (size_t)1 << layer;
#pragma omp parallel for schedule(static) if (par) num_threads(fft_inner_threads())
    for (size_t p = 0; p < N / 2; p++) {
      const size_t h = p >> layer, l = p & (stride - 1);
      const size_t i0 = (h << (layer + 1)) + l, i1 = i0 + stride;
      M31 a = v[i0], b = v[i1] * tw[h];
      v[i0] = a + b;
      v[i1] = a - b;
    }
  }
  return values;
}

// Stwo `CirclePoly::eval_at_point` (fold with mappings [.., pi(x), x, y]).
inline QM31 eval_at_point(const std::vector<M31>& coeffs, PointQ p) {
  size_t N = coeffs.size();
  uint32_t n = 0;
  while (((size_t)1 << n) < N) n++;
  if (n == 0) return QM31(coeffs[0]);
  std::vector<QM31> maps;  // maps[k] multiplies index bit k
  maps.push_back(p.y);
  QM31 x = p.x;
  for (uint32_t i = 1; i < n; i++) { maps.push_back(x); x = double_x(x); }
  std::vector<QM31> cur(N);
  for (size_t i = 0; i < N; i++) cur[i] = QM31(coeffs[i]);
  for (uint32_t k = 0; k < n; k++) {
    size_t half = cur.size() / 2;
    std::vector<QM31> nxt(half);
    // bit 0 of the (current) index pairs adjacent elements
    for (size_t i = 0; i < half; i++) nxt[i] = cur[2 * i] + maps[k] * cur[2 * i + 1];
    cur.swap(nxt);
  }
  return cur[0];
}

// Direct O(N) evaluation of the basis polynomial sum at a base-field point (test helper).
inline M31 eval_at_base_point(const std::vector<M31>& coeffs, PointM p) {
  PointQ q = into_ef(p);
  QM31 r = eval_at_point(coeffs, q);
  return r.a.a;
}

}  // namespace orc
