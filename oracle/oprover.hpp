// ORACLE (test infrastructure).  CPU restatement of the whole hot path:
//   prove_cairo_m  (crates/prover/src/prover.rs:23-147, transcript order is normative)
//   stwo `prove`, CommitmentSchemeProver::{commit, prove_values}, compute_fri_quotients,
//   FriProver::{commit, decommit}  (PARITY UNPINNED — Stwo not vendored; restated from upstream).
// Scalar/OpenMP, obviously-correct forms (per-element inverses, recomputed twiddles).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "oair.hpp"
#include "offt.hpp"
#include "omerkle.hpp"
#include "ochannel.hpp"
#include "oproof.hpp"
#include <map>
#include <set>
#include <stdexcept>

namespace orc {

struct CommitmentTree {
  std::vector<Col> polys;            // coefficients
  std::vector<uint32_t> poly_logs;
  std::vector<Col> evals;            // LDE on CanonicCoset(log + blowup)
  MerkleProver merkle;
};

// a loop over columns runs in parallel when the LARGE columns alone can keep the threads busy
inline bool columns_in_parallel(const std::vector<Col>& cols) {
  size_t large = 0;
  for (auto& c : cols) large += c.size() >= ((size_t)1 << 15) ? 1 : 0;
#ifdef _OPENMP
  // (measured on the GPU box's 256-CPU host, metric segment: with 64 threads and ~100 large columns the one-after-the-other
  // form took 17 s for a tree the column-parallel form commits in 0.6 s — a team of 64 re-formed for every layer of every
  // column; a quarter of the threads' worth of large columns is enough to prefer the column-parallel form)
  return large == 0 || 4 * large >= (size_t)omp_get_max_threads();
#else
  return false;
#endif
}
struct PcsProver {
  PcsConfig cfg;
  std::vector<CommitmentTree> trees;
  void commit_polys(std::vector<Col>&& polys, Channel& ch) {
    CommitmentTree t;
    t.polys = std::move(polys);
    t.evals.resize(t.polys.size());
    t.poly_logs.resize(t.polys.size());
    // few columns (preprocessed, composition): one after the other, every transform on all threads (offt.hpp); many: one per thread
    const bool outer = columns_in_parallel(t.polys);
#pragma omp parallel for schedule(dynamic) if (outer)
    for (size_t i = 0; i < t.polys.size(); i++) {
      t.poly_logs[i] = ilog2(t.polys[i].size());
      t.evals[i] = evaluate(t.polys[i], t.poly_logs[i] + cfg.log_blowup);
    }
    std::vector<const Column*> ptrs;
    for (auto& e : t.evals) ptrs.push_back(&e);
    t.merkle = MerkleProver::commit(ptrs);
    ch.mix_root(t.merkle.root());
    trees.push_back(std::move(t));
  }
  void commit_evals(std::vector<Col>&& evals, Channel& ch) {  // TreeBuilder::extend_evals + commit
    std::vector<Col> polys(evals.size());
    const bool outer = columns_in_parallel(evals);
#pragma omp parallel for schedule(dynamic) if (outer)
    for (size_t i = 0; i < evals.size(); i++) polys[i] = interpolate(std::move(evals[i]));
    commit_polys(std::move(polys), ch);
  }
};

inline PointQ random_point(Channel& ch) {  // CirclePoint::get_random_point
  QM31 t = ch.draw_felt();
  QM31 t2 = t.square();
  QM31 inv = (t2 + M31(1)).inverse();
  QM31 x = (QM31::one() - t2) * inv;
  QM31 y = (t + t) * inv;
  return {x, y};
}
inline Relations draw_relations(Channel& ch) {  // Relations::draw (components/mod.rs:311-323)
  Relations r;
  for (int rel = 0; rel < air::N_RELATIONS; rel++) {
    std::vector<QM31> za = ch.draw_felts(2);
    r.z[rel] = za[0];
    QM31 cur = QM31::one();
    for (int i = 0; i < air::MAX_REL_SIZE; i++) { r.alpha_pow[rel][i] = cur; cur = cur * za[1]; }
  }
  return r;
}

inline PublicData make_public_data(const cm_prover_input& in) {  // PublicData::new (public_data.rs:244-272)
  PublicData d;
  d.initial_pc = in.initial_pc; d.initial_fp = in.initial_fp; d.final_pc = in.final_pc; d.final_fp = in.final_fp;
  uint64_t steps = 0;
  for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) steps += in.n_bundles[i];
  d.clock = M31((uint32_t)steps).v;
  d.initial_root = in.initial_root; d.final_root = in.final_root;
  std::map<uint32_t, const cm_memory_cell*> init, fin;
  for (uint64_t i = 0; i < in.n_initial_memory; i++) init[in.initial_memory[i].address] = &in.initial_memory[i];
  for (uint64_t i = 0; i < in.n_final_memory; i++) fin[in.final_memory[i].address] = &in.final_memory[i];
  auto extract = [](const std::map<uint32_t, const cm_memory_cell*>& m, const uint32_t range[2]) {
    std::vector<PublicEntry> v;
    for (uint32_t a = range[0]; a < range[1]; a++) {
      PublicEntry e{};
      auto it = m.find(a);
      if (it != m.end()) { e.present = true; e.addr = a; for (int k = 0; k < 4; k++) e.value[k] = it->second->value[k]; e.clock = it->second->clock; }
      v.push_back(e);
    }
    return v;
  };
  d.program = extract(init, in.program_range);
  d.input = extract(init, in.input_range);
  d.output = extract(fin, in.output_range);
  return d;
}
inline void mix_public_data(const PublicData& d, Channel& ch) {  // public_data.rs:401-412, 132-186
  uint32_t w[7] = {d.initial_pc, d.initial_fp, d.final_pc, d.final_fp, d.clock, d.initial_root, d.final_root};
  ch.mix_u32s(w, 7);
  uint32_t lens[3] = {(uint32_t)d.program.size(), (uint32_t)d.input.size(), (uint32_t)d.output.size()};
  ch.mix_u32s(lens, 3);
  for (const auto* v : {&d.program, &d.input, &d.output}) {
    std::vector<uint32_t> words;
    for (auto& e : *v) if (e.present) { words.push_back(e.addr); for (int k = 0; k < 4; k++) words.push_back(e.value[k]); words.push_back(e.clock); }
    ch.mix_u32s(words);
  }
}
inline QM31 initial_logup_sum(const PublicData& d, const Relations& r) {  // public_data.rs:291-394
  std::vector<QM31> vals;
  M31 a[3] = {M31(d.initial_pc), M31(d.initial_fp), M31(1)};
  vals.push_back(r.combine(air::REL_REGISTERS, a, 3));
  M31 b[3] = {M31(d.final_pc), M31(d.final_fp), M31(d.clock) + M31(1)};
  vals.push_back(-r.combine(air::REL_REGISTERS, b, 3));
  M31 c[4] = {M31(0), M31(0), M31(d.initial_root), M31(d.initial_root)};
  vals.push_back(r.combine(air::REL_MERKLE, c, 4));
  M31 e[4] = {M31(0), M31(0), M31(d.final_root), M31(d.final_root)};
  vals.push_back(r.combine(air::REL_MERKLE, e, 4));
  auto add = [&](const std::vector<PublicEntry>& es, bool plus) {
    M31 root = M31(plus ? d.initial_root : d.final_root);
    for (auto& x : es) {
      if (!x.present) continue;
      M31 m[6] = {M31(x.addr), M31(x.clock), M31(x.value[0]), M31(x.value[1]), M31(x.value[2]), M31(x.value[3])};
      QM31 den = r.combine(air::REL_MEMORY, m, 6);
      vals.push_back(plus ? den : -den);
      for (uint32_t j = 0; j < 4; j++) {
        M31 k[4] = {M31(4) * M31(x.addr) + M31(j), M31(air::TREE_HEIGHT), M31(x.value[j]), root};
        vals.push_back(-r.combine(air::REL_MERKLE, k, 4));
      }
    }
  };
  add(d.program, true);
  add(d.input, true);
  add(d.output, false);
  QM31 s;
  for (auto& v : vals) s += v.inverse();
  return s;
}

// ---- component traces -----------------------------------------------------------------------------
inline std::vector<Col> preprocessed_columns() {
  std::vector<Col> pp(air::N_PREPROC);
  for (int id = 0; id < air::N_PREPROC; id++) {
    size_t N = (size_t)1 << air::PREPROC_LOG[id];
    pp[id].resize(N);
    for (size_t i = 0; i < N; i++) pp[id][i] = M31::raw(air::preproc_value(id, (uint32_t)i));
  }
  return pp;
}

template <class C, class Row>
void gen_builtin(ComponentTrace& ct, size_t n, Row row_fn) {
  ct.n_rows = n;
  ct.log_size = log_size_for(n);
  size_t N = (size_t)1 << ct.log_size;
  ct.trace.assign(C::N_TRACE, Col(N));
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    M31 out[448];
    row_fn(r, r < n ? 1u : 0u, out);
    for (int c = 0; c < C::N_TRACE; c++) ct.trace[c][r] = out[c];
  }
}

inline std::vector<ComponentTrace> write_traces(const cm_prover_input& in, std::string& err) {
  std::vector<ComponentTrace> cts(air::N_COMPONENTS);
  for (int c = 0; c < air::N_COMPONENTS; c++) cts[c].cid = c;
  for (int c = 0; c < air::N_OPCODE_COMPONENTS; c++) gen_opcode_trace_dispatch(c, cts[c], in.bundles[c], in.n_bundles[c], in.data_accesses);
  // memory (memory.rs:93-195): initial rows then final rows
  size_t ni = in.n_initial_memory, nf = in.n_final_memory;
  gen_builtin<air::MemoryC>(cts[air::C_MEMORY], ni + nf, [&](size_t r, uint32_t en, M31* o) {
    const air::MemoryCell* cell = nullptr;
    uint32_t root = 0;
    if (r < ni) { cell = reinterpret_cast<const air::MemoryCell*>(&in.initial_memory[r]); root = in.initial_root; }
    else if (r < ni + nf) { cell = reinterpret_cast<const air::MemoryCell*>(&in.final_memory[r - ni]); root = in.final_root; }
    air::MemoryC::witness<OrcOps>(cell, root, en, o);
  });
  size_t ti = in.n_initial_tree, tf = in.n_final_tree;
  gen_builtin<air::MerkleC>(cts[air::C_MERKLE], ti + tf, [&](size_t r, uint32_t en, M31* o) {
    const air::MerkleNode* n = nullptr;
    uint32_t root = 0;
    if (r < ti) { n = reinterpret_cast<const air::MerkleNode*>(&in.initial_tree[r]); root = in.initial_root; }
    else if (r < ti + tf) { n = reinterpret_cast<const air::MerkleNode*>(&in.final_tree[r - ti]); root = in.final_root; }
    air::MerkleC::witness<OrcOps>(n, root, en, o);
  });
  gen_builtin<air::ClockUpdateC>(cts[air::C_CLOCK_UPDATE], in.n_clock_updates, [&](size_t r, uint32_t en, M31* o) {
    const air::ClockUpdateRow* c = r < in.n_clock_updates ? reinterpret_cast<const air::ClockUpdateRow*>(&in.clock_updates[r]) : nullptr;
    air::ClockUpdateC::witness<OrcOps>(c, en, o);
  });
  // poseidon2 inputs = (left, right) of every merkle node, initial tree then final tree (adapter/mod.rs:163-172)
  gen_builtin<air::Poseidon2C>(cts[air::C_POSEIDON2], ti + tf, [&](size_t r, uint32_t en, M31* o) {
    uint32_t st[16] = {0};
    const uint32_t* p = nullptr;
    if (r < ti + tf) {
      const cm_merkle_node& n = r < ti ? in.initial_tree[r] : in.final_tree[r - ti];
      st[0] = n.left_value; st[1] = n.right_value;
      p = st;
    }
    air::Poseidon2C::witness<OrcOps>(p, en, o);
  });
  // histograms over every opcode component (components/mod.rs:139-160)
  HistTables h;
  for (int c = 0; c < air::N_OPCODE_COMPONENTS; c++) run_hist_dispatch(cts[c], h, err);
  auto fill = [&](int cid, const std::vector<uint32_t>& t, uint32_t log) {
    ComponentTrace& ct = cts[cid];
    ct.log_size = log; ct.n_rows = (size_t)1 << log;
    ct.trace.assign(1, Col((size_t)1 << log));
    for (size_t i = 0; i < t.size(); i++) ct.trace[0][i] = M31(t[i]);
  };
  fill(air::C_RC8, h.rc8, 8);
  fill(air::C_RC16, h.rc16, 16);
  fill(air::C_RC20, h.rc20, 20);
  fill(air::C_BITWISE, h.bitwise, 18);
  return cts;
}

// ---- constraints on the evaluation domain -----------------------------------------------------------
struct TraceLocation { size_t tr0, it0; };  // first column of the component in tree 1 / tree 2

template <class C>
void accumulate_constraints(const ComponentTrace& ct, const TraceLocation& loc, const PcsProver& pcs, const Relations& rel,
                            const QM31* coeff, std::vector<Col>& acc /*4 cols, 2^(log+1)*/) {
  const air::ComponentInfo& info = air::component_info(ct.cid);
  uint32_t n = ct.log_size, en = n + 1;
  size_t N = (size_t)1 << en;
  std::vector<const M31*> tr(info.n_trace), it(info.n_interaction);
  const M31* pp[air::N_PREPROC];
  // The constraints are evaluated on CanonicCoset(log + 1) (max constraint degree bound 2 x the trace size).  With
  // log_blowup_factor = 1 that IS the committed LDE domain; with a larger blowup the polynomials are evaluated there
  // separately (Stwo: `poly.evaluate(eval_domain)` when the committed evaluation is on another domain).
  std::vector<Col> tmp;
  if (pcs.cfg.log_blowup == 1) {
    for (int i = 0; i < info.n_trace; i++) tr[i] = pcs.trees[1].evals[loc.tr0 + i].data();
    for (int i = 0; i < info.n_interaction; i++) it[i] = pcs.trees[2].evals[loc.it0 + i].data();
    for (int i = 0; i < air::N_PREPROC; i++) pp[i] = pcs.trees[0].evals[i].data();
  } else {
    tmp.reserve(info.n_trace + info.n_interaction + air::N_PREPROC);
    for (int i = 0; i < info.n_trace; i++) { tmp.push_back(evaluate(pcs.trees[1].polys[loc.tr0 + i], en)); tr[i] = tmp.back().data(); }
    for (int i = 0; i < info.n_interaction; i++) { tmp.push_back(evaluate(pcs.trees[2].polys[loc.it0 + i], en)); it[i] = tmp.back().data(); }
    for (int i = 0; i < air::N_PREPROC; i++) {
      pp[i] = nullptr;      // a component only reads preprocessed columns of its own size
      if (pcs.trees[0].poly_logs[i] == n) { tmp.push_back(evaluate(pcs.trees[0].polys[i], en)); pp[i] = tmp.back().data(); }
    }
  }
  // 1 / vanishing of the trace coset on the two cosets of the evaluation domain
  CircleDomain ed = CanonicCoset(en).circle_domain();
  M31 dinv[2];
  for (int k = 0; k < 2; k++) dinv[k] = coset_vanishing<M31>(CanonicCoset(n).coset, ed.at(k), lift_m).inverse();
  QM31 shift = ct.claimed_sum * M31((uint32_t)1 << n).inverse();
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    RowConstraintEval e;
    e.tr = &tr; e.it = &it; e.pp = pp; e.row = r; e.prev_row = shifted_row(r, en, n, -1);
    e.rels = &rel; e.coeff = coeff; e.n_base = info.n_base_constraints; e.cumsum_shift = shift;
    C::eval(e);
    QM31 v = e.acc * dinv[r >> n];
    for (int k = 0; k < 4; k++) acc[k][r] += v.coord(k);
  }
}
inline void accumulate_constraints_dispatch(const ComponentTrace& ct, const TraceLocation& loc, const PcsProver& pcs,
                                            const Relations& rel, const QM31* coeff, std::vector<Col>& acc) {
  switch (ct.cid) {
#define ORC_X(id, T) case air::id: accumulate_constraints<air::T>(ct, loc, pcs, rel, coeff, acc); break;
    AIR_ALL_COMPONENTS(ORC_X)
#undef ORC_X
  }
}

// row-wise check on the trace domain (debug_tools/assert_constraints.rs). Returns "" or a description.
template <class C>
std::string assert_component(const ComponentTrace& ct, const Relations& rel, const std::vector<Col>& pp) {
  const air::ComponentInfo& info = air::component_info(ct.cid);
  uint32_t n = ct.log_size;
  size_t N = (size_t)1 << n;
  std::vector<const M31*> tr(info.n_trace), it(info.n_interaction);
  for (int i = 0; i < info.n_trace; i++) tr[i] = ct.trace[i].data();
  for (int i = 0; i < info.n_interaction; i++) it[i] = ct.interaction[i].data();
  const M31* ppp[air::N_PREPROC];
  for (int i = 0; i < air::N_PREPROC; i++) ppp[i] = pp[i].data();
  QM31 shift = ct.claimed_sum * M31((uint32_t)1 << n).inverse();
  std::string err;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    RowConstraintEval e;
    e.tr = &tr; e.it = &it; e.pp = ppp; e.row = r; e.prev_row = shifted_row(r, n, n, -1);
    e.rels = &rel; e.coeff = nullptr; e.n_base = info.n_base_constraints; e.cumsum_shift = shift;
    C::eval(e);
    if (e.first_bad >= 0) {
#pragma omp critical
      if (err.empty()) err = std::string(air::component_name(ct.cid)) + ": constraint " + std::to_string(e.first_bad) + " fails on row " + std::to_string(r);
    }
  }
  return err;
}
inline std::string assert_component_dispatch(const ComponentTrace& ct, const Relations& rel, const std::vector<Col>& pp) {
  switch (ct.cid) {
#define ORC_X(id, T) case air::id: return assert_component<air::T>(ct, rel, pp);
    AIR_ALL_COMPONENTS(ORC_X)
#undef ORC_X
  }
  return "?";
}

template <class C>
QM31 point_eval_component(const QM31* tr, const QM31* it, const QM31* pp, const Relations& rel, const QM31* coeff, int n_base, QM31 shift) {
  PointEval e;
  e.tr = tr; e.it = it; e.pp = pp; e.rels = &rel; e.coeff = coeff; e.n_base = n_base; e.cumsum_shift = shift;
  C::eval(e);
  return e.acc;
}
inline QM31 point_eval_dispatch(int cid, const QM31* tr, const QM31* it, const QM31* pp, const Relations& rel, const QM31* coeff, int n_base, QM31 shift) {
  switch (cid) {
#define ORC_X(id, T) case air::id: return point_eval_component<air::T>(tr, it, pp, rel, coeff, n_base, shift);
    AIR_ALL_COMPONENTS(ORC_X)
#undef ORC_X
  }
  return QM31();
}

// ---- quotients (Stwo core::pcs::quotients) ----------------------------------------------------------
struct SampleBatch {
  PointQ point;
  std::vector<std::pair<size_t, QM31>> cols;  // (column index within the size group, value)
};
inline bool same_point(const PointQ& a, const PointQ& b) { return a.x == b.x && a.y == b.y; }
// derived Ord of CirclePoint<SecureField>: x then y, each QM31 as its words (a, b, c, d) — framing switch sample_batch=sorted
inline bool point_less(const PointQ& a, const PointQ& b) {
  uint32_t wa[8], wb[8];
  a.x.to_u32(wa); a.y.to_u32(wa + 4);
  b.x.to_u32(wb); b.y.to_u32(wb + 4);
  for (int i = 0; i < 8; i++) if (wa[i] != wb[i]) return wa[i] < wb[i];
  return false;
}

// columns: LDE columns of one size (log), samples[c] = list of (point, value)
inline std::vector<Col> accumulate_quotients(uint32_t log, const std::vector<const Col*>& columns,
                                             const std::vector<std::vector<std::pair<PointQ, QM31>>>& samples, QM31 random_coeff) {
  std::vector<SampleBatch> batches;  // insertion-ordered grouping by point (ColumnSampleBatch::new_vec)
  for (size_t c = 0; c < columns.size(); c++)
    for (auto& s : samples[c]) {
      size_t b = 0;
      for (; b < batches.size(); b++) if (same_point(batches[b].point, s.first)) break;
      if (b == batches.size()) batches.push_back(SampleBatch{s.first, {}});
      batches[b].cols.push_back({c, s.second});
    }
  if (framing().sample_batch_sorted) std::stable_sort(batches.begin(), batches.end(), [](const SampleBatch& a, const SampleBatch& b) { return point_less(a.point, b.point); });
  // line coefficients (column_line_coeffs) and per-batch random coefficient powers
  struct LC { QM31 a, b, c; };
  std::vector<std::vector<LC>> lcs(batches.size());
  std::vector<QM31> batch_coeff(batches.size());
  for (size_t b = 0; b < batches.size(); b++) {
    QM31 alpha = QM31::one();
    for (auto& cv : batches[b].cols) {
      alpha = alpha * random_coeff;
      QM31 a = cv.second.complex_conjugate() - cv.second;
      QM31 c = batches[b].point.y.complex_conjugate() - batches[b].point.y;
      QM31 bb = cv.second * c - a * batches[b].point.y;
      lcs[b].push_back(LC{alpha * a, alpha * bb, alpha * c});
    }
    batch_coeff[b] = random_coeff.pow(batches[b].cols.size());
  }
  size_t N = (size_t)1 << log;
  std::vector<Col> out(4, Col(N));
  CircleDomain dom = CanonicCoset(log).circle_domain();
#pragma omp parallel for schedule(static)
  for (size_t row = 0; row < N; row++) {
    PointM p = dom.at(bit_reverse_index(row, log));
    QM31 acc;
    for (size_t b = 0; b < batches.size(); b++) {
      // denominator: (Pr.x - p.x) * Pi.y - (Pr.y - p.y) * Pi.x  in CM31
      CM31 prx = batches[b].point.x.a, pry = batches[b].point.y.a, pix = batches[b].point.x.b, piy = batches[b].point.y.b;
      CM31 den = (prx - CM31(p.x)) * piy - (pry - CM31(p.y)) * pix;
      CM31 dinv = den.inverse();
      QM31 num;
      for (size_t k = 0; k < batches[b].cols.size(); k++) {
        const LC& lc = lcs[b][k];
        QM31 value = lc.c * (*columns[batches[b].cols[k].first])[row];
        QM31 linear = lc.a * p.y + lc.b;
        num += value - linear;
      }
      acc = acc * batch_coeff[b] + num.mul_cm31(dinv);
    }
    for (int k = 0; k < 4; k++) out[k][row] = acc.coord(k);
  }
  return out;
}

// ---- FRI (Stwo core::fri) ---------------------------------------------------------------------------
inline QM31 qat(const std::vector<Col>& c, size_t i) { return QM31::from_m31s(c[0][i], c[1][i], c[2][i], c[3][i]); }
inline void qset(std::vector<Col>& c, size_t i, QM31 v) { for (int k = 0; k < 4; k++) c[k][i] = v.coord(k); }

// dst (line eval of size 2^(log-1)) = dst*alpha^2 + fold(src circle eval on CanonicCoset(log))
inline void fold_circle_into_line(std::vector<Col>& dst, const std::vector<Col>& src, uint32_t log, QM31 alpha) {
  CircleDomain dom = CanonicCoset(log).circle_domain();
  QM31 a2 = alpha * alpha;
  size_t n = (size_t)1 << (log - 1);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    PointM p = dom.at(bit_reverse_index(i << 1, log));
    QM31 f0 = qat(src, 2 * i), f1 = qat(src, 2 * i + 1);
    QM31 s = f0 + f1, d = (f0 - f1) * p.y.inverse();
    qset(dst, i, qat(dst, i) * a2 + (s + alpha * d));
  }
}
// line evaluation on LineDomain(half_odds(log)) folded once
inline std::vector<Col> fold_line(const std::vector<Col>& src, uint32_t log, QM31 alpha) {
  Coset c = Coset::half_odds(log);
  size_t n = (size_t)1 << (log - 1);
  std::vector<Col> out(4, Col(n));
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    M31 x = c.at(bit_reverse_index(i << 1, log)).x;
    QM31 f0 = qat(src, 2 * i), f1 = qat(src, 2 * i + 1);
    QM31 s = f0 + f1, d = (f0 - f1) * x.inverse();
    qset(out, i, s + alpha * d);
  }
  return out;
}

struct Queries {
  std::vector<size_t> positions;
  uint32_t log_domain_size;
  static Queries generate(Channel& ch, uint32_t log, size_t n_queries) {
    std::set<size_t> s;
    size_t cnt = 0;
    size_t mask = ((size_t)1 << log) - 1;
    for (;;) {
      Hash32 b = ch.draw_random_bytes();
      for (int k = 0; k < 8; k++) {
        uint32_t w;
        memcpy(&w, b.data() + 4 * k, 4);
        s.insert(w & mask);
        if (++cnt == n_queries) return Queries{std::vector<size_t>(s.begin(), s.end()), log};
      }
    }
  }
  Queries fold(uint32_t n) const {
    Queries q;
    q.log_domain_size = log_domain_size - n;
    for (size_t p : positions) { size_t f = p >> n; if (q.positions.empty() || q.positions.back() != f) q.positions.push_back(f); }
    return q;
  }
};
// compute_decommitment_positions_and_witness_evals
inline void decommit_positions(const std::vector<Col>& column, const std::vector<size_t>& queries, uint32_t fold_step,
                               std::vector<size_t>& positions, std::vector<QM31>& witness) {
  size_t i = 0;
  while (i < queries.size()) {
    size_t start = (queries[i] >> fold_step) << fold_step;
    size_t j = i;
    while (j < queries.size() && (queries[j] >> fold_step) == (queries[i] >> fold_step)) j++;
    size_t qi = i;
    for (size_t pos = start; pos < start + ((size_t)1 << fold_step); pos++) {
      positions.push_back(pos);
      if (qi < j && queries[qi] == pos) { qi++; continue; }
      witness.push_back(qat(column, pos));
    }
    i = j;
  }
}

// ---- the prover ---------------------------------------------------------------------------------------
struct ProveOutput {
  Proof proof;
  uint64_t cells = 0;
  std::vector<ComponentTrace> traces;  // kept for tests (trace-domain columns)
  Relations relations;
  TranscriptLog transcript;            // every Fiat-Shamir step (ochannel.hpp)
};

// optional phase timing to stderr (ORC_TIMING=1) — used to find the oracle's own slow spots
struct OrcTick {
  bool on; std::chrono::steady_clock::time_point t;
  OrcTick() : on(getenv("ORC_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void operator()(const char* what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[oracle] %-22s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};
#define ORC_TICK(x) tick(x)
inline ProveOutput prove_segment(const cm_prover_input& in, const PcsConfig& cfg, bool keep_traces = false) {
  ProveOutput out;
  Proof& pf = out.proof;
  pf.config = cfg;
  Channel ch;
  ch.log.p = &out.transcript;
  OrcTick tick;
  // PcsConfig::mix_into (order inside FriConfig::mix_into: framing switch pcs_mix)
  ch.mix_u64(cfg.pow_bits);
  ch.mix_u64(cfg.log_blowup);
  if (framing().pcs_mix_blq) { ch.mix_u64(cfg.log_last_layer); ch.mix_u64(cfg.n_queries); }
  else { ch.mix_u64(cfg.n_queries); ch.mix_u64(cfg.log_last_layer); }
  PcsProver pcs;
  pcs.cfg = cfg;
  pf.public_data = make_public_data(in);
  mix_public_data(pf.public_data, ch);
  // tree 0: preprocessed
  std::vector<Col> pp = preprocessed_columns();
  ORC_TICK("preproc gen");
  { std::vector<Col> c = pp; pcs.commit_evals(std::move(c), ch); }
  ORC_TICK("preproc commit");
  // tree 1: execution trace
  std::string err;
  std::vector<ComponentTrace> cts = write_traces(in, err);
  if (!err.empty()) throw std::runtime_error(err);
  ORC_TICK("write_traces");
  for (auto& ct : cts) { pf.claim_log_sizes.push_back(ct.log_size); ch.mix_u64(ct.log_size); }
  std::vector<TraceLocation> loc(cts.size());
  {
    std::vector<Col> cols;
    for (size_t c = 0; c < cts.size(); c++) { loc[c].tr0 = cols.size(); for (auto& col : cts[c].trace) cols.push_back(col); }
    pcs.commit_evals(std::move(cols), ch);
  }
  ORC_TICK("trace commit");
  pf.interaction_pow = grind(ch, 2);  // relations::INTERACTION_POW_BITS
  ch.mix_u64(pf.interaction_pow);
  Relations rel = draw_relations(ch);
  // tree 2: interaction trace
  for (auto& ct : cts) gen_interaction_dispatch(ct, rel, pp);
  ORC_TICK("interaction gen");
  for (auto& ct : cts) { pf.claimed_sums.push_back(ct.claimed_sum); ch.mix_felts(&ct.claimed_sum, 1); }
  {
    std::vector<Col> cols;
    for (size_t c = 0; c < cts.size(); c++) { loc[c].it0 = cols.size(); for (auto& col : cts[c].interaction) cols.push_back(col); }
    pcs.commit_evals(std::move(cols), ch);
  }
  ORC_TICK("interaction commit");
  for (int t = 0; t < 3; t++) for (auto l : pcs.trees[t].poly_logs) out.cells += (uint64_t)1 << l;
  // ---- stwo prove ----
  QM31 random_coeff = ch.draw_felt();
  size_t total_constraints = 0;
  for (auto& ct : cts) total_constraints += air::component_info(ct.cid).n_constraints;
  std::vector<QM31> powers(total_constraints);  // powers[g] = rho^(total-1-g)
  {
    QM31 cur = QM31::one();
    for (size_t g = total_constraints; g-- > 0;) { powers[g] = cur; cur = cur * random_coeff; }
  }
  uint32_t max_log = 0;
  for (auto& ct : cts) max_log = std::max(max_log, ct.log_size);
  uint32_t comp_log = max_log + 1;
  std::map<uint32_t, std::vector<Col>> accs;  // eval log -> 4 coordinate columns
  {
    size_t g = 0;
    for (size_t c = 0; c < cts.size(); c++) {
      uint32_t el = cts[c].log_size + 1;
      if (!accs.count(el)) accs[el] = std::vector<Col>(4, Col((size_t)1 << el));
      accumulate_constraints_dispatch(cts[c], loc[c], pcs, rel, &powers[g], accs[el]);
      g += air::component_info(cts[c].cid).n_constraints;
    }
  }
  ORC_TICK("constraints");
  std::vector<Col> comp_poly;  // DomainEvaluationAccumulator::finalize
  for (auto& kv : accs) {
    std::vector<Col> vals = kv.second;
    if (!comp_poly.empty())
      for (int k = 0; k < 4; k++) {
        Col e = evaluate(comp_poly[k], kv.first);
        for (size_t i = 0; i < e.size(); i++) vals[k][i] += e[i];
      }
    comp_poly.resize(4);
    for (int k = 0; k < 4; k++) comp_poly[k] = interpolate(std::move(vals[k]));
  }
  if (ilog2(comp_poly[0].size()) != comp_log) throw std::runtime_error("composition log size mismatch");
  pcs.commit_polys(std::move(comp_poly), ch);
  ORC_TICK("composition commit");
  PointQ oods = random_point(ch);
  // mask points
  std::vector<std::vector<std::vector<PointQ>>> pts(4);
  pts[0].assign(pcs.trees[0].polys.size(), {oods});
  pts[1].assign(pcs.trees[1].polys.size(), {oods});
  pts[2].assign(pcs.trees[2].polys.size(), {oods});
  for (size_t c = 0; c < cts.size(); c++) {
    int ni = air::component_info(cts[c].cid).n_interaction;
    PointM step = CanonicCoset(cts[c].log_size).step();
    PointQ prev = oods + into_ef(step.conjugate());
    for (int k = ni - 4; k < ni; k++) pts[2][loc[c].it0 + k] = {prev, oods};
  }
  pts[3].assign(4, {oods});
  // sampled values
  pf.sampled_values.resize(4);
  for (int t = 0; t < 4; t++) {
    pf.sampled_values[t].resize(pcs.trees[t].polys.size());
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < pcs.trees[t].polys.size(); c++)
      for (auto& p : pts[t][c]) pf.sampled_values[t][c].push_back(eval_at_point(pcs.trees[t].polys[c], p));
  }
  {
    std::vector<QM31> flat;
    for (auto& t : pf.sampled_values) for (auto& c : t) for (auto& s : c) flat.push_back(s);
    ch.mix_felts(flat);
  }
  ORC_TICK("oods sampling");
  // sanity check: composition OODS value == constraints evaluated on the sampled mask (stwo prove)
  {
    QM31 comp_at_oods = PointEval::combine_ef(std::vector<QM31>{pf.sampled_values[3][0][0], pf.sampled_values[3][1][0],
                                                               pf.sampled_values[3][2][0], pf.sampled_values[3][3][0]}.data());
    QM31 ppv[air::N_PREPROC];
    for (int i = 0; i < air::N_PREPROC; i++) ppv[i] = pf.sampled_values[0][i][0];
    QM31 total;
    size_t g = 0;
    for (size_t c = 0; c < cts.size(); c++) {
      const air::ComponentInfo& info = air::component_info(cts[c].cid);
      std::vector<QM31> tr, it;
      for (int k = 0; k < info.n_trace; k++) tr.push_back(pf.sampled_values[1][loc[c].tr0 + k][0]);
      for (int k = 0; k < info.n_interaction; k++) for (auto& s : pf.sampled_values[2][loc[c].it0 + k]) it.push_back(s);
      QM31 shift = cts[c].claimed_sum * M31((uint32_t)1 << cts[c].log_size).inverse();
      QM31 num = point_eval_dispatch(cts[c].cid, tr.data(), it.data(), ppv, rel, &powers[g], info.n_base_constraints, shift);
      QM31 den = coset_vanishing<QM31>(CanonicCoset(cts[c].log_size).coset, oods, into_ef);
      total += num * den.inverse();
      g += info.n_constraints;
    }
    if (total != comp_at_oods) throw std::runtime_error("ConstraintsNotSatisfied: composition OODS mismatch");
  }
  // FRI quotients: group all LDE columns by size (descending, stable)
  QM31 qcoeff = ch.draw_felt();
  struct ColRef { int t; size_t c; };
  std::map<uint32_t, std::vector<ColRef>, std::greater<uint32_t>> groups;
  for (int t = 0; t < 4; t++)
    for (size_t c = 0; c < pcs.trees[t].evals.size(); c++) groups[pcs.trees[t].poly_logs[c] + cfg.log_blowup].push_back({t, c});
  std::vector<uint32_t> q_logs;
  std::vector<std::vector<Col>> quotients;
  for (auto& kv : groups) {
    std::vector<const Col*> cols;
    std::vector<std::vector<std::pair<PointQ, QM31>>> smp;
    for (auto& r : kv.second) {
      cols.push_back(&pcs.trees[r.t].evals[r.c]);
      std::vector<std::pair<PointQ, QM31>> s;
      for (size_t k = 0; k < pts[r.t][r.c].size(); k++) s.push_back({pts[r.t][r.c][k], pf.sampled_values[r.t][r.c][k]});
      smp.push_back(s);
    }
    q_logs.push_back(kv.first);
    quotients.push_back(accumulate_quotients(kv.first, cols, smp, qcoeff));
  }
  // FRI commit
  MerkleProver first_tree;
  {
    std::vector<const Column*> ptrs;
    for (auto& q : quotients) for (auto& c : q) ptrs.push_back(&c);
    ORC_TICK("quotients");
    first_tree = MerkleProver::commit(ptrs);
    ch.mix_root(first_tree.root());
  }
  QM31 circle_alpha = ch.draw_felt();
  uint32_t layer_log = q_logs[0] - 1;
  std::vector<Col> layer(4, Col((size_t)1 << layer_log));
  struct InnerLayer { std::vector<Col> eval; uint32_t log; MerkleProver tree; };
  std::vector<InnerLayer> inner;
  size_t qi = 0;
  uint32_t last_log = cfg.log_last_layer + cfg.log_blowup;
  while (layer_log > last_log) {
    while (qi < quotients.size() && q_logs[qi] - 1 == layer_log) { fold_circle_into_line(layer, quotients[qi], q_logs[qi], circle_alpha); qi++; }
    InnerLayer il;
    il.eval = layer; il.log = layer_log;
    std::vector<const Column*> ptrs;
    for (auto& c : il.eval) ptrs.push_back(&c);
    il.tree = MerkleProver::commit(ptrs);
    ch.mix_root(il.tree.root());
    QM31 alpha = ch.draw_felt();
    layer = fold_line(il.eval, layer_log, alpha);
    layer_log--;
    inner.push_back(std::move(il));
  }
  if (qi != quotients.size()) throw std::runtime_error("fri: not all columns consumed");
  // last layer: interpolate the line evaluation (2^last_log points) and keep 2^log_last_layer coefficients
  {
    size_t n = (size_t)1 << last_log;
    std::vector<QM31> vals(n);
    for (size_t i = 0; i < n; i++) vals[i] = qat(layer, i);
    // line IFFT on LineDomain(half_odds(last_log)), values in bit-reversed order
    uint32_t lg = last_log;
    Coset c = Coset::half_odds(lg);
    for (uint32_t l = 0; l < lg; l++) {
      size_t stride = (size_t)1 << l;
      for (size_t h = 0; h < (n >> (l + 1)); h++) {
        // layer l pairs (x, -x) of the coset doubled l times; twiddle of group h:
        Coset cl = c;
        for (uint32_t d = 0; d < l; d++) cl = cl.dbl();
        M31 x = cl.at(bit_reverse_index(h, lg - 1 - l)).x;
        for (size_t k = 0; k < stride; k++) {
          size_t i0 = (h << (l + 1)) + k, i1 = i0 + stride;
          QM31 a = vals[i0], b = vals[i1];
          vals[i0] = a + b;
          vals[i1] = (a - b) * x.inverse();
        }
      }
    }
    M31 ninv = M31((uint32_t)n).inverse();
    for (auto& v : vals) v = v * ninv;
    size_t keep = (size_t)1 << cfg.log_last_layer;
    // Stwo: bit_reverse(values); line_ifft (natural order) -> LinePoly coefficients in ITS bit-reversed order;
    // into_ordered_coefficients() bit-reverses again.  Working in place on the bit-reversed evaluations, position p of
    // `vals` therefore already is ordered coefficient p (degree p).  The proof keeps from_ordered_coefficients(first keep).
    for (size_t i = keep; i < n; i++) if (!vals[i].is_zero()) throw std::runtime_error("fri: invalid last layer degree");
    pf.last_layer_poly.assign(keep, QM31());
    for (size_t i = 0; i < keep; i++) pf.last_layer_poly[bit_reverse_index(i, cfg.log_last_layer)] = vals[i];
    pf.last_layer_log_size = cfg.log_last_layer;
    ch.mix_felts(pf.last_layer_poly);
  }
  ORC_TICK("fri commit");
  pf.proof_of_work = grind(ch, cfg.pow_bits);
  ch.mix_u64(pf.proof_of_work);
  // FRI decommit
  Queries queries = Queries::generate(ch, q_logs[0], cfg.n_queries);
  std::map<uint32_t, std::vector<size_t>> qpos_by_log;
  for (auto l : q_logs) qpos_by_log[l] = queries.fold(queries.log_domain_size - l).positions;
  {
    std::map<uint32_t, std::vector<size_t>> dpos;
    for (size_t k = 0; k < quotients.size(); k++) {
      std::vector<size_t> pos;
      decommit_positions(quotients[k], qpos_by_log[q_logs[k]], 1, pos, pf.fri_first.fri_witness);
      dpos[q_logs[k]] = pos;
    }
    std::vector<const Column*> ptrs;
    for (auto& q : quotients) for (auto& c : q) ptrs.push_back(&c);
    auto r = first_tree.decommit(dpos, ptrs);
    pf.fri_first.decommitment = r.second;
    pf.fri_first.commitment = first_tree.root();
  }
  {
    Queries lq = queries.fold(1);
    for (auto& il : inner) {
      FriLayerProof lp;
      std::vector<size_t> pos;
      decommit_positions(il.eval, lq.positions, 1, pos, lp.fri_witness);
      std::map<uint32_t, std::vector<size_t>> dpos;
      dpos[il.log] = pos;
      std::vector<const Column*> ptrs;
      for (auto& c : il.eval) ptrs.push_back(&c);
      auto r = il.tree.decommit(dpos, ptrs);
      lp.decommitment = r.second;
      lp.commitment = il.tree.root();
      pf.fri_inner.push_back(std::move(lp));
      lq = lq.fold(1);
    }
  }
  // decommit the commitment trees
  for (int t = 0; t < 4; t++) {
    std::vector<const Column*> ptrs;
    for (auto& c : pcs.trees[t].evals) ptrs.push_back(&c);
    auto r = pcs.trees[t].merkle.decommit(qpos_by_log, ptrs);
    pf.queried_values.push_back(r.first);
    pf.decommitments.push_back(r.second);
    pf.commitments.push_back(pcs.trees[t].merkle.root());
  }
  out.relations = rel;
  if (keep_traces) out.traces = std::move(cts);
  return out;
}

}  // namespace orc
