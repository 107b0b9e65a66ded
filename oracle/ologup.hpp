// ORACLE (test infrastructure).  The oracle's OWN restatement of Stwo's `LogupAtRow` batching
// (stwo-constraint-framework: `add_to_relation` queues fractions, `finalize_logup_in_pairs()` =
// `finalize_logup_batched(&[0, 0, 1, 1, ...])`, `finalize_logup()` = one fraction per batch; semantics described
// in-tree at docs/stwo-debug.md:123-216).  Deliberately NOT the product's streaming form
// (cairo_m_amd/csrc/air/logup_stream.hpp keeps O(1) state for GPU threads): here every fraction of the row is
// collected first, then grouped by its batch index and summed — the way Stwo's own code is laid out — so the
// two sides only agree if both restate the same batching.  PARITY UNPINNED against Stwo itself (not vendored).
//
// CRTP: D provides  combine(rel, vals, n) -> EF,  ef_from(F) -> EF,  on_entry(rel, mult, vals, n),
//                   emit_batch(is_last, numerator, denominator).
#pragma once
#include <stddef.h>
#include <stdexcept>

namespace orc {

template <class D, class F_, class EF_>
struct LogupAtRow {
  using F = F_;
  using EF = EF_;
  struct Frac { EF num, den; };
  static constexpr size_t MAX_FRACS = 64;   // the largest component (u32_store_div_fp_fp) queues 58
  Frac fracs[MAX_FRACS];                    // in add_to_relation order
  size_t n_fracs = 0;

  D& self() { return *static_cast<D*>(this); }

  // EvalAtRow::add_to_relation(RelationEntry::new(relation, multiplicity, values))
  void rel_arr(int r, F mult, const F* vals, int n) {
    D& d = self();
    d.on_entry(r, mult, vals, n);
    if (n_fracs == MAX_FRACS) throw std::runtime_error("LogupAtRow: too many relation entries in one row");
    fracs[n_fracs].num = d.ef_from(mult);
    fracs[n_fracs].den = d.combine(r, vals, n);
    n_fracs++;
  }
  template <class... V>
  void rel(int r, F mult, V... vals) {
    F arr[] = {vals...};
    rel_arr(r, mult, arr, (int)sizeof...(V));
  }
  // finalize_logup_batched(batching): fractions with the same batch index are summed into one committed column;
  // every batch but the last constrains (c_j - c_{j-1}) * D - N, the last one carries the cumulative-sum mask.
  void finalize_batched(const size_t* batching) {
    size_t n_batches = 0;
    for (size_t i = 0; i < n_fracs; i++) if (batching[i] + 1 > n_batches) n_batches = batching[i] + 1;
    for (size_t b = 0; b < n_batches; b++) {
      bool have = false;
      Frac acc{};
      for (size_t i = 0; i < n_fracs; i++) {
        if (batching[i] != b) continue;
        if (!have) { acc = fracs[i]; have = true; }
        else {   // Fraction + Fraction: (a.n * b.d + b.n * a.d) / (a.d * b.d)
          Frac s;
          s.num = acc.num * fracs[i].den + fracs[i].num * acc.den;
          s.den = acc.den * fracs[i].den;
          acc = s;
        }
      }
      if (have) self().emit_batch(b + 1 == n_batches, acc.num, acc.den);
    }
    n_fracs = 0;
  }
  void finalize_pairs() {   // finalize_logup_in_pairs: batching = [0, 0, 1, 1, 2, 2, ...]
    size_t batching[MAX_FRACS];
    for (size_t i = 0; i < n_fracs; i++) batching[i] = i / 2;
    finalize_batched(batching);
  }
  void finalize_single() {  // finalize_logup: batching = [0, 1, 2, ...]
    size_t batching[MAX_FRACS];
    for (size_t i = 0; i < n_fracs; i++) batching[i] = i;
    finalize_batched(batching);
  }
};

}  // namespace orc
