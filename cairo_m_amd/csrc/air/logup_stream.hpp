// Streaming restatement of Stwo's LogupAtRow / `finalize_logup_in_pairs` (stwo-constraint-framework;
// semantics described in-tree at docs/stwo-debug.md:123-216): relation entries are batched in pairs
// (0,0,1,1,...), every batch but the last yields the constraint (c_j - c_{j-1})*D - N = 0 on a fresh
// committed QM31 column, the last batch uses the [-1, 0] row mask and the cumsum shift.
// O(1) state: one pending entry + one pending batch, so GPU evaluators never hold all fractions.
//
// CRTP: D provides
//   EF  combine(int rel, const F* vals, int n)   — sum alpha^i * v_i - z
//   EF  ef_from(F)
//   void on_entry(int rel, F mult, const F* vals, int n)   — histogram hook (may be empty)
//   void emit_batch(bool last, EF num, EF den)
#pragma once
#include "air_common.hpp"

namespace air {

template <class D, class F_, class EF_>
struct LogupStream {
  using F = F_;
  using EF = EF_;
  bool have_first = false, have_pending = false;
  F first_n;
  EF first_d, pend_n, pend_d;

  AIR_HD D& self() { return *static_cast<D*>(this); }

  AIR_HD void rel_arr(int r, F mult, const F* vals, int n) {
    D& d = self();
    d.on_entry(r, mult, vals, n);
    EF den = d.combine(r, vals, n);
    if (have_first) {
      // Fraction sum: (n0/d0) + (n1/d1) = (n0*d1 + n1*d0) / (d0*d1)
      EF N = den * first_n + first_d * mult;
      EF Dd = first_d * den;
      have_first = false;
      complete(N, Dd);
    } else {
      first_n = mult;
      first_d = den;
      have_first = true;
    }
  }
  template <class... V>
  AIR_HD void rel(int r, F mult, V... vals) {
    F arr[] = {vals...};
    self().rel_arr(r, mult, arr, (int)sizeof...(V));   // (an evaluator may take the entries one by one: gpu_air.hpp LogupEval)
  }
  AIR_HD void complete(EF N, EF Dd) {
    if (have_pending) self().emit_batch(false, pend_n, pend_d);
    pend_n = N;
    pend_d = Dd;
    have_pending = true;
  }
  AIR_HD void finalize_pairs() {
    if (have_first) {
      have_first = false;
      complete(self().ef_from(first_n), first_d);
    }
    if (have_pending) self().emit_batch(true, pend_n, pend_d);
    have_pending = false;
  }
  // `finalize_logup()` (one entry per batch) is only used by components with a single entry
  // (range_check_macro.rs:181, bitwise.rs:236), where both batchings coincide.
  AIR_HD void finalize_single() { finalize_pairs(); }
};

}  // namespace air
