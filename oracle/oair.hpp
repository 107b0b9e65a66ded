// ORACLE (test infrastructure).  CPU evaluators over the shared AIR descriptions
// (cairo_m_amd/csrc/air/*.hpp restate the reference's component files — their `eval` half is pinned to vectors derived
// mechanically from the reference's evaluate() text, tests/golden/air_eval_vectors.json; the evaluators below restate
// Stwo's constraint framework, incl. the oracle's own LogUp batching (ologup.hpp), independently of the HIP evaluators):
//   * trace generation            <- Claim::write_trace  (crates/prover/src/components/mod.rs:106-194)
//   * rc / bitwise histograms     <- range_check_macro.rs:62-112, preprocessed/bitwise.rs:72-157
//   * LogUp interaction trace     <- InteractionClaim::write_interaction_trace (components/mod.rs:198-281)
//                                    + Stwo LogupTraceGenerator::{write_frac, finalize_col, finalize_last}
//   * row-wise constraint check   <- debug_tools/assert_constraints.rs:24-60
//   * constraint quotients        <- Stwo FrameworkComponent::evaluate_constraint_quotients_on_domain
//   * point evaluation            <- Stwo FrameworkComponent::evaluate_constraint_quotients_at_point
// PARITY UNPINNED for the Stwo parts (framework not vendored); the component AIRs are checked by
// tests/test_oracle_air.py (every constraint vanishes on every row + LogUp sums cancel), the same
// check the reference runs in tests/prover.rs:351-370.
#pragma once
#include "ofield.hpp"
#include "ocircle.hpp"
#include "ologup.hpp"
#include "../cairo_m_amd/csrc/air/components.hpp"
#include "../include/cairom_hip.h"
#include <vector>
#include <string>
#include <cstring>
#include <atomic>

namespace orc {

struct OrcOps {
  using M = M31;
  static M mk(uint32_t v) { return M31::raw(v % P); }
  static M inv(M x) { return x.inverse(); }
};

struct Relations {
  QM31 z[air::N_RELATIONS];
  QM31 alpha_pow[air::N_RELATIONS][air::MAX_REL_SIZE];
  QM31 combine(int rel, const M31* vals, int n) const {
    QM31 acc;
    for (int i = 0; i < n; i++) acc += alpha_pow[rel][i] * vals[i];
    return acc - z[rel];
  }
  QM31 combine_q(int rel, const QM31* vals, int n) const {
    QM31 acc;
    for (int i = 0; i < n; i++) acc += alpha_pow[rel][i] * vals[i];
    return acc - z[rel];
  }
};

using Col = std::vector<M31>;

struct ComponentTrace {
  int cid = 0;
  uint32_t log_size = 0;
  size_t n_rows = 0;  // non padded length
  std::vector<Col> trace;        // tree-1 columns (2^log_size)
  std::vector<Col> interaction;  // tree-2 columns
  QM31 claimed_sum;
};

inline uint32_t log_size_for(size_t n) {  // max(LOG_N_LANES, ceil_log2(n))
  uint32_t l = 4;
  while (((size_t)1 << l) < n) l++;
  return l;
}

// ---- trace generation -----------------------------------------------------------------------------
template <class C>
void gen_opcode_trace(ComponentTrace& ct, const cm_bundle* bundles, size_t n, const cm_data_access* acc) {
  ct.n_rows = n;
  ct.log_size = log_size_for(n);
  size_t N = (size_t)1 << ct.log_size;
  ct.trace.assign(C::N_TRACE, Col(N));
  const air::Access* a = reinterpret_cast<const air::Access*>(acc);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    air::Bundle b = r < n ? *reinterpret_cast<const air::Bundle*>(&bundles[r]) : air::default_bundle();
    M31 out[64];
    C::template witness<OrcOps>(b, a, r < n ? 1u : 0u, out);
    for (int c = 0; c < C::N_TRACE; c++) ct.trace[c][r] = out[c];
  }
}
inline void gen_opcode_trace_dispatch(int cid, ComponentTrace& ct, const cm_bundle* b, size_t n, const cm_data_access* acc) {
  switch (cid) {
#define ORC_X(id, T) case air::id: gen_opcode_trace<air::T>(ct, b, n, acc); break;
    AIR_OPCODE_COMPONENTS(ORC_X)
#undef ORC_X
    default: break;
  }
}

// ---- row evaluators over trace-domain rows ---------------------------------------------------------
// Histogram (range_check_macro.rs:72-79, bitwise.rs:86-109): every lane of every lookup, padding included.
struct HistTables {
  std::vector<uint32_t> rc8, rc16, rc20, bitwise;
  HistTables() : rc8(1 << 8), rc16(1 << 16), rc20(1 << 20), bitwise(1 << 18) {}
};
struct NoEF {};
inline NoEF operator*(NoEF, NoEF) { return {}; }
inline NoEF operator*(NoEF, M31) { return {}; }
inline NoEF operator+(NoEF, NoEF) { return {}; }
struct HistEval : LogupAtRow<HistEval, M31, NoEF> {
  const std::vector<Col>* cols;
  size_t row;
  int ci = 0;
  HistTables* h;
  std::string* err;
  M31 next() { return (*cols)[ci++][row]; }
  M31 preproc(int) { return M31(); }
  M31 c(uint32_t v) { return M31::raw(v); }
  void constraint(M31) {}
  NoEF combine(int, const M31*, int) { return {}; }
  NoEF ef_from(M31) { return {}; }
  void bump(std::vector<uint32_t>& t, uint32_t idx, const char* name) {
    if (idx >= t.size()) {
#pragma omp critical
      *err = std::string("lookup value out of range for ") + name;
      return;
    }
#pragma omp atomic
    t[idx]++;
  }
  void on_entry(int rel, M31, const M31* v, int) {
    if (rel == air::REL_RC8) bump(h->rc8, v[0].v, "rc8");
    else if (rel == air::REL_RC16) bump(h->rc16, v[0].v, "rc16");
    else if (rel == air::REL_RC20) bump(h->rc20, v[0].v, "rc20");
    else if (rel == air::REL_BITWISE) {
      uint64_t idx = (uint64_t)v[0].v * 65536 + ((uint64_t)v[1].v << 8) + v[2].v;
      bump(h->bitwise, idx < (1u << 18) ? (uint32_t)idx : 0xffffffffu, "bitwise");
    }
  }
  void emit_batch(bool, NoEF, NoEF) {}
};

// Row dump: constraint values and relation entries of one row in evaluation order (golden-vector replay,
// tests/test_air_eval_golden.py).
struct DumpEval : LogupAtRow<DumpEval, M31, NoEF> {
  const uint32_t* row;
  const uint32_t* ppv;
  int ci = 0;
  std::vector<uint32_t>* constraints;
  std::vector<uint32_t>* entries;   // per entry: relation id, multiplicity, n, values...
  int n_batches = 0;
  M31 next() { return M31::raw(row[ci++]); }
  M31 preproc(int id) { return M31::raw(ppv[id]); }
  M31 c(uint32_t v) { return M31::raw(v); }
  void constraint(M31 x) { constraints->push_back(x.v); }
  NoEF combine(int, const M31*, int) { return {}; }
  NoEF ef_from(M31) { return {}; }
  void on_entry(int rel, M31 mult, const M31* v, int n) {
    entries->push_back((uint32_t)rel); entries->push_back(mult.v); entries->push_back((uint32_t)n);
    for (int i = 0; i < n; i++) entries->push_back(v[i].v);
  }
  void emit_batch(bool, NoEF, NoEF) { n_batches++; }
};
template <class C>
int dump_row(const uint32_t* row, const uint32_t* ppv, std::vector<uint32_t>& cons, std::vector<uint32_t>& ents) {
  DumpEval e;
  e.row = row; e.ppv = ppv; e.constraints = &cons; e.entries = &ents;
  C::eval(e);
  return e.n_batches;
}
inline int dump_row_dispatch(int cid, const uint32_t* row, const uint32_t* ppv, std::vector<uint32_t>& cons, std::vector<uint32_t>& ents) {
  switch (cid) {
#define ORC_X(id, T) case air::id: return dump_row<air::T>(row, ppv, cons, ents);
    AIR_ALL_COMPONENTS(ORC_X)
#undef ORC_X
  }
  return -1;
}

// LogUp column generator (Stwo LogupTraceGenerator): column j = column j-1 + num/den.
struct LogupGenEval : LogupAtRow<LogupGenEval, M31, QM31> {
  const std::vector<Col>* cols;
  const std::vector<Col>* pp;  // preprocessed columns (trace domain), indexed by PreprocId
  size_t row;
  int ci = 0, batch = 0;
  const Relations* rels;
  std::vector<Col>* out;
  QM31 prev;
  M31 next() { return (*cols)[ci++][row]; }
  M31 preproc(int id) { return (*pp)[id][row]; }
  M31 c(uint32_t v) { return M31::raw(v); }
  void constraint(M31) {}
  QM31 combine(int r, const M31* v, int n) { return rels->combine(r, v, n); }
  QM31 ef_from(M31 m) { return QM31(m); }
  void on_entry(int, M31, const M31*, int) {}
  void emit_batch(bool, QM31 num, QM31 den) {
    QM31 v = prev + num * den.inverse();
    for (int k = 0; k < 4; k++) (*out)[4 * batch + k][row] = v.coord(k);
    prev = v;
    batch++;
  }
};

// Index (bit-reversed storage of log `n`) of the row that is `offset` trace-steps away, where the
// trace step is the generator of the canonic coset of log `trace_log` (n >= trace_log).
inline size_t shifted_row(size_t r, uint32_t n, uint32_t trace_log, int offset) {
  size_t i = bit_reverse_index(r, n);             // natural circle-domain index
  size_t half = (size_t)1 << (n - 1);
  // domain point index (exponent) — work in units of the half-coset step G_{2^(n-1)} = 4 units of g=G_{2^(n+1)}
  // point(i) = g^(1+4i) for i<half, g^-(1+4(i-half)) otherwise; trace step = G_{2^trace_log} = g^(2^(n+1-trace_log))
  int64_t mod = (int64_t)1 << (n + 1);
  int64_t e = i < half ? (int64_t)(1 + 4 * i) : -(int64_t)(1 + 4 * (i - half));
  e += (int64_t)offset * ((int64_t)1 << (n + 1 - trace_log));
  e = ((e % mod) + mod) % mod;
  // invert: e == 1 mod 4 -> first half; e == 3 mod 4 -> second half
  size_t j;
  if ((e & 3) == 1) j = (size_t)((e - 1) / 4);
  else j = half + (size_t)((mod - e - 1) / 4);
  return bit_reverse_index(j, n);
}

// finalize_last: subtract claimed_sum/2^log from the last column, inclusive prefix sum in coset order.
inline QM31 finalize_last(std::vector<Col>& inter, uint32_t log_size) {
  size_t N = (size_t)1 << log_size;
  size_t base = inter.size() - 4;
  QM31 sum;
  for (size_t r = 0; r < N; r++) sum += QM31::from_m31s(inter[base][r], inter[base + 1][r], inter[base + 2][r], inter[base + 3][r]);
  QM31 shift = sum * M31((uint32_t)N).inverse();
  QM31 acc;
  for (size_t k = 0; k < N; k++) {
    size_t r = bit_reverse_index(coset_index_to_circle_domain_index(k, log_size), log_size);
    QM31 v = QM31::from_m31s(inter[base][r], inter[base + 1][r], inter[base + 2][r], inter[base + 3][r]) - shift;
    acc += v;
    for (int c = 0; c < 4; c++) inter[base + c][r] = acc.coord(c);
  }
  return sum;
}

template <class C>
void gen_interaction(ComponentTrace& ct, const Relations& rel, const std::vector<Col>& pp) {
  const air::ComponentInfo& info = air::component_info(ct.cid);
  size_t N = (size_t)1 << ct.log_size;
  ct.interaction.assign(info.n_interaction, Col(N));
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    LogupGenEval e;
    e.cols = &ct.trace; e.pp = &pp; e.row = r; e.rels = &rel; e.out = &ct.interaction;
    C::eval(e);
  }
  ct.claimed_sum = finalize_last(ct.interaction, ct.log_size);
}
inline void gen_interaction_dispatch(ComponentTrace& ct, const Relations& rel, const std::vector<Col>& pp) {
  switch (ct.cid) {
#define ORC_X(id, T) case air::id: gen_interaction<air::T>(ct, rel, pp); break;
    AIR_ALL_COMPONENTS(ORC_X)
#undef ORC_X
  }
}
template <class C>
void run_hist(const ComponentTrace& ct, HistTables& h, std::string& err) {
  size_t N = (size_t)1 << ct.log_size;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    HistEval e;
    e.cols = &ct.trace; e.row = r; e.h = &h; e.err = &err;
    C::eval(e);
  }
}
inline void run_hist_dispatch(const ComponentTrace& ct, HistTables& h, std::string& err) {
  switch (ct.cid) {
#define ORC_X(id, T) case air::id: run_hist<air::T>(ct, h, err); break;
    AIR_OPCODE_COMPONENTS(ORC_X)
#undef ORC_X
    default: break;
  }
}

// ---- constraint evaluators -------------------------------------------------------------------------
// Generic row evaluator: reads columns at `row` of storage of log `eval_log` (trace domain when
// eval_log == trace_log, LDE otherwise).  Accumulates sum_k coeff[k] * C_k (coeff = 1 => assert mode collects
// the first non-zero constraint instead).
struct RowConstraintEval : LogupAtRow<RowConstraintEval, M31, QM31> {
  const std::vector<const M31*>* tr;  // tree-1 columns of the component
  const std::vector<const M31*>* it;  // tree-2 columns
  const M31* const* pp;               // preprocessed columns by PreprocId (same log as the component)
  size_t row, prev_row;
  const Relations* rels;
  const QM31* coeff;                  // per-constraint coefficients, or null (assert mode)
  int n_base;
  QM31 cumsum_shift;
  int ci = 0, ii = 0, kb = 0, kl = 0;
  QM31 prev_col, acc;
  int first_bad = -1;
  M31 next() { return (*tr)[ci++][row]; }
  M31 preproc(int id) { return pp[id][row]; }
  M31 c(uint32_t v) { return M31::raw(v); }
  void constraint(M31 x) {
    int k = kb++;
    if (coeff) acc += coeff[k] * x;
    else if (!x.is_zero() && first_bad < 0) first_bad = k;
  }
  void constraint_q(QM31 x) {
    int k = n_base + kl++;
    if (coeff) acc += coeff[k] * x;
    else if (!x.is_zero() && first_bad < 0) first_bad = k;
  }
  QM31 combine(int r, const M31* v, int n) { return rels->combine(r, v, n); }
  QM31 ef_from(M31 m) { return QM31(m); }
  void on_entry(int, M31, const M31*, int) {}
  QM31 mask(size_t r) {
    return QM31::from_m31s((*it)[ii][r], (*it)[ii + 1][r], (*it)[ii + 2][r], (*it)[ii + 3][r]);
  }
  void emit_batch(bool last, QM31 num, QM31 den) {
    if (!last) {
      QM31 cur = mask(row);
      ii += 4;
      QM31 diff = cur - prev_col;
      prev_col = cur;
      constraint_q(diff * den - num);
    } else {
      QM31 prev_row_v = mask(prev_row), cur = mask(row);
      ii += 4;
      QM31 diff = cur - prev_row_v - prev_col;
      constraint_q((diff + cumsum_shift) * den - num);
    }
  }
};

// Point evaluator (F = QM31) over OODS mask values.
struct PointEval : LogupAtRow<PointEval, QM31, QM31> {
  const QM31* tr;           // tree-1 sampled values (offset 0), one per column
  const QM31* it;           // tree-2 sampled values, flattened in mask order (see build_mask_layout)
  const QM31* pp;           // preprocessed sampled values by PreprocId
  const Relations* rels;
  const QM31* coeff;
  int n_base;
  QM31 cumsum_shift;
  int ci = 0, ii = 0, kb = 0, kl = 0;
  QM31 prev_col, acc;
  QM31 next() { return tr[ci++]; }
  QM31 preproc(int id) { return pp[id]; }
  QM31 c(uint32_t v) { return QM31(M31::raw(v)); }
  void constraint(QM31 x) { acc += coeff[kb++] * x; }
  void constraint_q(QM31 x) { acc += coeff[n_base + kl++] * x; }
  QM31 combine(int r, const QM31* v, int n) { return rels->combine_q(r, v, n); }
  QM31 ef_from(QM31 m) { return m; }
  void on_entry(int, QM31, const QM31*, int) {}
  // SecureField from 4 coordinate-column samples: sum_k coord_k * basis_k (Stwo combine_ef)
  static QM31 combine_ef(const QM31* c4) {
    const QM31 b1 = QM31::from_u32(0, 1, 0, 0), b2 = QM31::from_u32(0, 0, 1, 0), b3 = QM31::from_u32(0, 0, 0, 1);
    return c4[0] + c4[1] * b1 + c4[2] * b2 + c4[3] * b3;
  }
  void emit_batch(bool last, QM31 num, QM31 den) {
    if (!last) {
      QM31 cur = combine_ef(it + ii);
      ii += 4;
      QM31 diff = cur - prev_col;
      prev_col = cur;
      constraint_q(diff * den - num);
    } else {
      // last 4 columns carry two samples each: [offset -1, offset 0]
      QM31 pr[4], cu[4];
      for (int k = 0; k < 4; k++) { pr[k] = it[ii + 2 * k]; cu[k] = it[ii + 2 * k + 1]; }
      ii += 8;
      QM31 diff = combine_ef(cu) - combine_ef(pr) - prev_col;
      constraint_q((diff + cumsum_shift) * den - num);
    }
  }
};

}  // namespace orc
