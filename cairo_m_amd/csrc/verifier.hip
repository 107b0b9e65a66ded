// Product-side verifier: `verify_cairo_m::<Blake2sMerkleChannel>` (/root/reference/crates/prover/src/verifier.rs:17-95)
// and the Stwo `verify` it calls (CommitmentSchemeVerifier::verify_values, fri_answers, FriVerifier,
// MerkleVerifier).  Host code only — verification touches O(queries * log n) values, there is nothing to put
// on the GPU — so `cm_verify_proof*` also work on a machine without one.  Written against the product's own
// types (host_channel.hpp, field.hpp, proof.hpp, point_eval.hpp); the CPU oracle has its own, separate verifier.
#include "../../include/cairom_hip.h"
#include "host_channel.hpp"
#include "point_eval.hpp"
#include "proof.hpp"
#include "air/components.hpp"
#include <algorithm>
#include <functional>
#include <map>
#include <set>
#include <string>

namespace cm {

using hostch::Channel;

// ---- words -> ProofData (inverse of proof_to_words) ---------------------------------------------------------
bool proof_from_words(const uint32_t* w, uint64_t n, ProofData& p, std::string& err) {
  uint64_t i = 0;
  bool ok = true;
  auto u = [&]() -> uint32_t { if (i >= n) { ok = false; return 0; } return w[i++]; };
  auto cnt = [&](uint64_t unit_words) -> uint32_t {  // a length field followed by that many records
    uint32_t c = u();
    if (ok && (uint64_t)c * unit_words > n - i) ok = false;
    return ok ? c : 0;
  };
  auto u64 = [&]() -> uint64_t { uint64_t lo = u(); uint64_t hi = u(); return lo | (hi << 32); };
  auto q = [&]() -> QM31 { uint32_t t[4]; for (int k = 0; k < 4; k++) t[k] = u(); for (int k = 0; k < 4; k++) if (t[k] >= P) ok = false; return QM31::from_u32(t); };
  auto h = [&]() -> Hash32 { uint32_t t[8]; for (int k = 0; k < 8; k++) t[k] = u(); Hash32 x; memcpy(x.data(), t, 32); return x; };
  auto dec = [&](MerkleDecommitment& d) {
    uint32_t nh = cnt(8);
    for (uint32_t k = 0; k < nh; k++) d.hash_witness.push_back(h());
    uint32_t nc = cnt(1);
    for (uint32_t k = 0; k < nc; k++) { uint32_t v = u(); if (v >= P) ok = false; d.column_witness.push_back(v); }
  };
  auto layer = [&](FriLayerProofData& l) {
    uint32_t nw = cnt(4);
    for (uint32_t k = 0; k < nw; k++) l.fri_witness.push_back(q());
    dec(l.decommitment);
    l.commitment = h();
  };
  auto entries = [&](std::vector<PublicEntry>& v) {
    uint32_t c = cnt(7);
    for (uint32_t k = 0; k < c; k++) {
      PublicEntry e;
      e.present = u(); e.addr = u();
      for (int j = 0; j < 4; j++) e.value[j] = u();
      e.clock = u();
      if (e.present > 1 || e.addr >= P || e.clock >= P) ok = false;      // canonical M31 words only (M31(x) needs x < P)
      for (int j = 0; j < 4; j++) if (e.value[j] >= P) ok = false;
      v.push_back(e);
    }
  };
  if (u() != 0x434d5031u) { err = "not a proof word stream (bad magic)"; return false; }
  p.config.pow_bits = u(); p.config.log_blowup_factor = u(); p.config.log_last_layer_degree_bound = u(); p.config.n_queries = u();
  uint32_t nc = cnt(5);
  for (uint32_t k = 0; k < nc; k++) p.claim_log_sizes.push_back(u());
  for (uint32_t k = 0; k < nc; k++) p.claimed_sums.push_back(q());
  PublicData& d = p.public_data;
  d.initial_pc = u(); d.initial_fp = u(); d.final_pc = u(); d.final_fp = u(); d.clock = u(); d.initial_root = u(); d.final_root = u();
  for (uint32_t w : {d.initial_pc, d.initial_fp, d.final_pc, d.final_fp, d.clock, d.initial_root, d.final_root}) if (w >= P) ok = false;
  entries(d.program); entries(d.input); entries(d.output);
  p.interaction_pow = u64();
  uint32_t nt = cnt(8);
  for (uint32_t k = 0; k < nt; k++) p.commitments.push_back(h());
  p.sampled_values.resize(nt);
  for (uint32_t t = 0; t < nt && ok; t++) {
    uint32_t ncol = cnt(1);
    p.sampled_values[t].resize(ncol);
    for (uint32_t c = 0; c < ncol && ok; c++) {
      uint32_t ns = cnt(4);
      for (uint32_t s = 0; s < ns; s++) p.sampled_values[t][c].push_back(q());
    }
  }
  p.decommitments.resize(nt);
  for (uint32_t t = 0; t < nt && ok; t++) dec(p.decommitments[t]);
  p.queried_values.resize(nt);
  for (uint32_t t = 0; t < nt && ok; t++) {
    uint32_t nv = cnt(1);
    for (uint32_t k = 0; k < nv; k++) { uint32_t v = u(); if (v >= P) ok = false; p.queried_values[t].push_back(v); }
  }
  p.proof_of_work = u64();
  layer(p.fri_first);
  uint32_t nl = cnt(1);
  p.fri_inner.resize(nl);
  for (uint32_t k = 0; k < nl && ok; k++) layer(p.fri_inner[k]);
  uint32_t np = cnt(4);
  for (uint32_t k = 0; k < np; k++) p.last_layer_poly.push_back(q());
  p.last_layer_log_size = u();
  if (!ok || i != n) { err = "malformed proof word stream"; return false; }
  return true;
}

namespace {

// ---- Merkle (Stwo MerkleVerifier::verify over Blake2sMerkleHasher) ------------------------------------------
Hash32 hash_node(const Hash32* left, const Hash32* right, const uint32_t* vals, size_t n) {
  if (framing().hash_node_rfc) {   // framing.hpp `hash_node=rfc`: Blake2s-256 of left || right || le32(values)
    std::vector<uint8_t> buf((left ? 64 : 0) + 4 * n);
    if (left) { memcpy(buf.data(), left->data(), 32); memcpy(buf.data() + 32, right->data(), 32); }
    if (n) memcpy(buf.data() + (left ? 64 : 0), vals, 4 * n);
    return hostch::blake2s256(buf.data(), buf.size());
  }
  uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m[16];
  if (left) {
    memcpy(m, left->data(), 32);
    memcpy(m + 8, right->data(), 32);
    hostch::compress(st, m, 0, 0);
  }
  for (size_t c0 = 0; c0 < n; c0 += 16) {
    for (size_t k = 0; k < 16; k++) m[k] = c0 + k < n ? vals[c0 + k] : 0u;
    hostch::compress(st, m, 0, 0);
  }
  Hash32 out;
  memcpy(out.data(), st, 32);
  return out;
}
// col_logs: log size of every column of the tree in commitment order; queries: log -> sorted unique positions
std::string merkle_verify(const Hash32& root, const std::vector<uint32_t>& col_logs, const std::map<uint32_t, std::vector<uint32_t>>& queries,
                          const std::vector<uint32_t>& queried_values, const MerkleDecommitment& d) {
  if (col_logs.empty()) return "empty tree";
  uint32_t max_log = *std::max_element(col_logs.begin(), col_logs.end());
  std::map<uint32_t, uint32_t> n_cols;
  for (auto l : col_logs) n_cols[l]++;
  size_t qi = 0, hi = 0, ci = 0;
  std::vector<std::pair<uint32_t, Hash32>> last;
  for (int log = (int)max_log; log >= 0; log--) {
    const uint32_t nc = n_cols.count((uint32_t)log) ? n_cols[(uint32_t)log] : 0;
    static const std::vector<uint32_t> none;
    auto it = queries.find((uint32_t)log);
    const std::vector<uint32_t>& colq = (nc && it != queries.end()) ? it->second : none;
    std::vector<std::pair<uint32_t, Hash32>> cur;
    size_t pi = 0, cq = 0;
    const bool has_prev = log < (int)max_log;
    while (pi < last.size() || cq < colq.size()) {
      uint32_t node;
      if (pi < last.size() && cq < colq.size()) node = std::min(last[pi].first / 2, colq[cq]);
      else if (pi < last.size()) node = last[pi].first / 2;
      else node = colq[cq];
      Hash32 l, r;
      if (has_prev) {
        if (pi < last.size() && last[pi].first == 2 * node) l = last[pi++].second;
        else { if (hi >= d.hash_witness.size()) return "WitnessTooShort"; l = d.hash_witness[hi++]; }
        if (pi < last.size() && last[pi].first == 2 * node + 1) r = last[pi++].second;
        else { if (hi >= d.hash_witness.size()) return "WitnessTooShort"; r = d.hash_witness[hi++]; }
      }
      std::vector<uint32_t> vals(nc);
      const bool isq = cq < colq.size() && colq[cq] == node;
      if (isq) {
        cq++;
        if (qi + nc > queried_values.size()) return "TooFewQueriedValues";
        for (uint32_t k = 0; k < nc; k++) vals[k] = queried_values[qi++];
      } else {
        if (ci + nc > d.column_witness.size()) return "WitnessTooShort";
        for (uint32_t k = 0; k < nc; k++) vals[k] = d.column_witness[ci++];
      }
      cur.push_back({node, hash_node(has_prev ? &l : nullptr, has_prev ? &r : nullptr, vals.data(), nc)});
    }
    last.swap(cur);
  }
  if (hi != d.hash_witness.size() || ci != d.column_witness.size()) return "WitnessTooLong";
  if (qi != queried_values.size()) return "TooManyQueriedValues";
  if (last.size() != 1 || last[0].second != root) return "RootMismatch";
  return "";
}

// ---- public data: transcript framing and LogUp contribution (public_data.rs:291-412) --------------------------
void mix_public_data(const PublicData& d, Channel& ch) {
  uint32_t w[7] = {d.initial_pc, d.initial_fp, d.final_pc, d.final_fp, d.clock, d.initial_root, d.final_root};
  ch.mix_u32s(w, 7);
  uint32_t lens[3] = {(uint32_t)d.program.size(), (uint32_t)d.input.size(), (uint32_t)d.output.size()};
  ch.mix_u32s(lens, 3);
  for (const auto* v : {&d.program, &d.input, &d.output}) {
    std::vector<uint32_t> words;
    for (auto& e : *v) if (e.present) { words.push_back(e.addr); for (int k = 0; k < 4; k++) words.push_back(e.value[k]); words.push_back(e.clock); }
    ch.mix_u32s(words.data(), words.size());
  }
}
QM31 combine(const HostRelations& rel, int r, std::initializer_list<M31> vals) {
  QM31 a;
  int i = 0;
  for (M31 v : vals) a += rel.alpha_pow[r][i++] * v;
  return a - rel.z[r];
}
QM31 initial_logup_sum(const PublicData& d, const HostRelations& rel) {
  const M31 one(1), zero;
  std::vector<QM31> dens;
  dens.push_back(combine(rel, air::REL_REGISTERS, {M31(d.initial_pc), M31(d.initial_fp), one}));
  dens.push_back(-combine(rel, air::REL_REGISTERS, {M31(d.final_pc), M31(d.final_fp), M31(d.clock) + one}));
  dens.push_back(combine(rel, air::REL_MERKLE, {zero, zero, M31(d.initial_root), M31(d.initial_root)}));
  dens.push_back(combine(rel, air::REL_MERKLE, {zero, zero, M31(d.final_root), M31(d.final_root)}));
  auto add = [&](const std::vector<PublicEntry>& es, bool emit) {
    const M31 root(emit ? d.initial_root : d.final_root), height(air::TREE_HEIGHT), four(4);
    for (auto& e : es) {
      if (!e.present) continue;
      QM31 mem = combine(rel, air::REL_MEMORY, {M31(e.addr), M31(e.clock), M31(e.value[0]), M31(e.value[1]), M31(e.value[2]), M31(e.value[3])});
      dens.push_back(emit ? mem : -mem);
      for (uint32_t k = 0; k < 4; k++) dens.push_back(-combine(rel, air::REL_MERKLE, {four * M31(e.addr) + M31(k), height, M31(e.value[k]), root}));
    }
  };
  add(d.program, true);
  add(d.input, true);
  add(d.output, false);
  QM31 s;
  for (auto& x : dens) s += inv(x);
  return s;
}

CPoint<QM31> into_ef(CPoint<M31> p) { return CPoint<QM31>{QM31(p.x), QM31(p.y)}; }
// vanishing polynomial of CanonicCoset(log).coset at a QM31 point (shift is zero for a canonic coset)
QM31 canonic_vanishing(uint32_t log, CPoint<QM31> p) {
  QM31 x = p.x;
  for (uint32_t i = 1; i < log; i++) x = double_x(x);
  return x;
}
CPoint<M31> domain_point(uint32_t log, uint32_t row) { return point_at_index(domain_index_at(log, bit_reverse(row, log))); }

struct Sample { CPoint<QM31> pt; QM31 value; };
// DEEP quotient of one queried row (Stwo accumulate_row_quotients): cols[c] = samples of column c of the size group
QM31 row_quotient(const std::vector<std::vector<Sample>>& cols, QM31 random_coeff, const std::vector<uint32_t>& row, CPoint<M31> p) {
  struct Batch { CPoint<QM31> pt; std::vector<std::pair<size_t, QM31>> entries; };
  std::vector<Batch> batches;
  for (size_t c = 0; c < cols.size(); c++)
    for (auto& s : cols[c]) {
      size_t b = 0;
      for (; b < batches.size(); b++) if (batches[b].pt.x == s.pt.x && batches[b].pt.y == s.pt.y) break;
      if (b == batches.size()) batches.push_back(Batch{s.pt, {}});
      batches[b].entries.push_back({c, s.value});
    }
  if (framing().sample_batch_sorted)
    std::stable_sort(batches.begin(), batches.end(), [](const Batch& a, const Batch& b) { return secure_point_less(a.pt, b.pt); });
  QM31 acc;
  for (auto& b : batches) {
    QM31 alpha(M31(1)), num;
    const QM31 cdiff = conj_u(b.pt.y) - b.pt.y;
    for (auto& e : b.entries) {
      alpha = alpha * random_coeff;
      QM31 a = conj_u(e.second) - e.second;
      QM31 bb = e.second * cdiff - a * b.pt.y;
      num += alpha * (cdiff * M31(row[e.first]) - (a * p.y + bb));
    }
    CM31 prx = b.pt.x.a, pix = b.pt.x.b, pry = b.pt.y.a, piy = b.pt.y.b;
    CM31 den = (prx - CM31(p.x)) * piy - (pry - CM31(p.y)) * pix;
    acc = acc * qpow(random_coeff, b.entries.size()) + mul_cm31(num, inv(den));
  }
  return acc;
}

struct FoldQueries {
  std::vector<uint32_t> positions;
  FoldQueries fold(uint32_t n) const {
    FoldQueries q;
    for (auto p : positions) { uint32_t f = p >> n; if (q.positions.empty() || q.positions.back() != f) q.positions.push_back(f); }
    return q;
  }
};
// compute_decommitment_positions_and_rebuild_evals (fold step 1)
bool rebuild_evals(const std::vector<uint32_t>& queries, const std::vector<QM31>& query_evals, const std::vector<QM31>& witness, size_t& wi,
                   std::vector<uint32_t>& positions, std::vector<std::array<QM31, 2>>& pairs, std::vector<uint32_t>& starts) {
  size_t i = 0;
  while (i < queries.size()) {
    uint32_t start = (queries[i] >> 1) << 1;
    size_t j = i;
    while (j < queries.size() && (queries[j] >> 1) == (queries[i] >> 1)) j++;
    size_t qi = i;
    std::array<QM31, 2> pr;
    for (uint32_t k = 0; k < 2; k++) {
      positions.push_back(start + k);
      if (qi < j && queries[qi] == start + k) pr[k] = query_evals[qi++];
      else { if (wi >= witness.size()) return false; pr[k] = witness[wi++]; }
    }
    pairs.push_back(pr);
    starts.push_back(start);
    i = j;
  }
  return true;
}

}  // namespace

// "" = the proof verifies; otherwise the name of the failed check
// `expected` is the verifier's OWN PcsConfig (verify_cairo_m takes it from the caller, defaulting to REGULAR_96_BITS,
// verifier.rs:17-31): the security level is never read from the proof.  A proof made under another config fails.
std::string verify_proof(const ProofData& pf, const cm_pcs_config& expected) {
  FramingUse framing_use;   // (see Prover: one framing for the whole verification)
  const cm_pcs_config& cfg = expected;
  if (pf.config.pow_bits != cfg.pow_bits || pf.config.log_blowup_factor != cfg.log_blowup_factor ||
      pf.config.n_queries != cfg.n_queries || pf.config.log_last_layer_degree_bound != cfg.log_last_layer_degree_bound)
    return "InvalidStructure(config): the proof was made under a different PcsConfig than the verifier's";
  if (pf.claim_log_sizes.size() != (size_t)air::N_COMPONENTS || pf.claimed_sums.size() != (size_t)air::N_COMPONENTS ||
      pf.commitments.size() != 4 || pf.sampled_values.size() != 4 || pf.decommitments.size() != 4 || pf.queried_values.size() != 4)
    return "InvalidStructure";
  for (auto l : pf.claim_log_sizes) if (l < 4 || l > 26) return "InvalidStructure(log size)";
  if (cfg.log_blowup_factor < 1 || cfg.log_blowup_factor > 4 || cfg.n_queries == 0 || cfg.n_queries > 4096 || cfg.pow_bits > 64 ||
      cfg.log_last_layer_degree_bound > 20) return "InvalidStructure(config)";
  Channel ch;
  ch.mix_u64(cfg.pow_bits);
  ch.mix_u64(cfg.log_blowup_factor);
  if (framing().pcs_mix_blq) { ch.mix_u64(cfg.log_last_layer_degree_bound); ch.mix_u64(cfg.n_queries); }
  else { ch.mix_u64(cfg.n_queries); ch.mix_u64(cfg.log_last_layer_degree_bound); }
  mix_public_data(pf.public_data, ch);
  // column log sizes per tree (preprocessed, trace, interaction, composition)
  std::vector<std::vector<uint32_t>> logs(4);
  for (int i = 0; i < air::N_PREPROC; i++) logs[0].push_back(air::PREPROC_LOG[i]);
  std::vector<size_t> tr0(air::N_COMPONENTS), it0(air::N_COMPONENTS);
  for (int c = 0; c < air::N_COMPONENTS; c++) {
    const air::ComponentInfo& info = air::component_info(c);
    tr0[c] = logs[1].size(); it0[c] = logs[2].size();
    logs[1].insert(logs[1].end(), info.n_trace, pf.claim_log_sizes[c]);
    logs[2].insert(logs[2].end(), info.n_interaction, pf.claim_log_sizes[c]);
  }
  ch.mix_root(pf.commitments[0]);
  for (auto l : pf.claim_log_sizes) ch.mix_u64(l);
  ch.mix_root(pf.commitments[1]);
  ch.mix_u64(pf.interaction_pow);
  if (ch.trailing_zeros() < INTERACTION_POW_BITS) return "ProofOfWork(interaction)";  // relations::INTERACTION_POW_BITS (verifier.rs:55-58)
  HostRelations rel;
  for (int r = 0; r < air::N_RELATIONS; r++) {
    QM31 z, alpha;
    ch.draw_two_felts(z, alpha);
    rel.z[r] = z;
    QM31 cur(M31(1));
    for (int i = 0; i < air::MAX_REL_SIZE; i++) { rel.alpha_pow[r][i] = cur; cur = cur * alpha; }
  }
  {
    QM31 s = initial_logup_sum(pf.public_data, rel);  // verifier.rs:69-77
    for (auto& c : pf.claimed_sums) s += c;
    if (!s.is_zero()) return "InvalidLogupSum";
  }
  for (auto& c : pf.claimed_sums) ch.mix_felts(&c, 1);
  ch.mix_root(pf.commitments[2]);
  // ---- stwo verify ----
  const QM31 random_coeff = ch.draw_felt();
  uint32_t max_log = *std::max_element(pf.claim_log_sizes.begin(), pf.claim_log_sizes.end());
  logs[3].assign(4, max_log + 1);
  ch.mix_root(pf.commitments[3]);
  CPoint<QM31> oods;
  {
    QM31 t = ch.draw_felt();
    QM31 t2 = t * t;
    QM31 iv = inv(t2 + M31(1));
    oods.x = (QM31(M31(1)) - t2) * iv;
    oods.y = (t + t) * iv;
  }
  // mask points: every column at the OODS point; the last LogUp column group of a component also one step back
  std::vector<std::vector<std::vector<CPoint<QM31>>>> pts(4);
  for (int t = 0; t < 4; t++) pts[t].assign(logs[t].size(), {oods});
  for (int c = 0; c < air::N_COMPONENTS; c++) {
    int ni = air::component_info(c).n_interaction;
    CPoint<M31> step = point_at_index(subgroup_gen_index(pf.claim_log_sizes[c]));
    CPoint<QM31> prev = cadd(oods, CPoint<QM31>{QM31(step.x), QM31(-step.y)});
    for (int k = ni - 4; k < ni; k++) pts[2][it0[c] + k] = {prev, oods};
  }
  for (int t = 0; t < 4; t++) {
    if (pf.sampled_values[t].size() != logs[t].size()) return "InvalidStructure(sampled columns)";
    for (size_t c = 0; c < logs[t].size(); c++) if (pf.sampled_values[t][c].size() != pts[t][c].size()) return "InvalidStructure(samples)";
  }
  {  // composition OODS value == sum_c constraints_c(mask) / vanishing_c(oods)
    size_t total = 0;
    for (int c = 0; c < air::N_COMPONENTS; c++) total += air::component_info(c).n_constraints;
    std::vector<QM31> powers(total);
    QM31 cur(M31(1));
    for (size_t g = total; g-- > 0;) { powers[g] = cur; cur = cur * random_coeff; }
    QM31 c4[4] = {pf.sampled_values[3][0][0], pf.sampled_values[3][1][0], pf.sampled_values[3][2][0], pf.sampled_values[3][3][0]};
    QM31 ppv[air::N_PREPROC];
    for (int i = 0; i < air::N_PREPROC; i++) ppv[i] = pf.sampled_values[0][i][0];
    QM31 sum;
    size_t g = 0;
    for (int c = 0; c < air::N_COMPONENTS; c++) {
      const air::ComponentInfo& info = air::component_info(c);
      std::vector<QM31> tr, it;
      for (int k = 0; k < info.n_trace; k++) tr.push_back(pf.sampled_values[1][tr0[c] + k][0]);
      for (int k = 0; k < info.n_interaction; k++) for (auto& s : pf.sampled_values[2][it0[c] + k]) it.push_back(s);
      QM31 shift = pf.claimed_sums[c] * inv(M31::from_u32(1u << pf.claim_log_sizes[c]));
      QM31 num = point_eval(c, tr.data(), it.data(), ppv, rel, &powers[g], info.n_base_constraints, shift);
      sum += num * inv(canonic_vanishing(pf.claim_log_sizes[c], oods));
      g += info.n_constraints;
    }
    if (sum != combine_ef(c4)) return "OodsNotMatching";
  }
  {
    std::vector<QM31> flat;
    for (auto& t : pf.sampled_values) for (auto& c : t) for (auto& s : c) flat.push_back(s);
    ch.mix_felts(flat.data(), flat.size());
  }
  const QM31 qcoeff = ch.draw_felt();
  std::set<uint32_t, std::greater<uint32_t>> ext;
  for (int t = 0; t < 4; t++) for (auto l : logs[t]) ext.insert(l + cfg.log_blowup_factor);
  const std::vector<uint32_t> q_logs(ext.begin(), ext.end());
  // FRI commit phase replay
  ch.mix_root(pf.fri_first.commitment);
  const QM31 circle_alpha = ch.draw_felt();
  const uint32_t last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup_factor;
  if (q_logs[0] < last_log + 1 || pf.fri_inner.size() != (size_t)(q_logs[0] - 1 - last_log)) return "Fri(InvalidNumFriLayers)";
  std::vector<QM31> alphas;
  for (auto& l : pf.fri_inner) { ch.mix_root(l.commitment); alphas.push_back(ch.draw_felt()); }
  if (pf.last_layer_poly.size() != ((size_t)1 << cfg.log_last_layer_degree_bound) ||
      pf.last_layer_log_size != cfg.log_last_layer_degree_bound) return "Fri(LastLayerDegreeInvalid)";
  ch.mix_felts(pf.last_layer_poly.data(), pf.last_layer_poly.size());
  ch.mix_u64(pf.proof_of_work);
  if (ch.trailing_zeros() < cfg.pow_bits) return "ProofOfWork";
  FoldQueries queries;
  {
    std::set<uint32_t> s;
    uint32_t cnt = 0;
    const uint32_t mask = (1u << q_logs[0]) - 1;
    bool done = false;
    while (!done) {
      hostch::Hash32 b = ch.draw_random_bytes();
      for (int k = 0; k < 8 && !done; k++) {
        uint32_t wv;
        memcpy(&wv, b.data() + 4 * k, 4);
        s.insert(wv & mask);
        if (++cnt == cfg.n_queries) done = true;
      }
    }
    queries.positions.assign(s.begin(), s.end());
  }
  std::map<uint32_t, std::vector<uint32_t>> qpos;
  for (auto l : q_logs) qpos[l] = queries.fold(q_logs[0] - l).positions;
  for (int t = 0; t < 4; t++) {
    std::vector<uint32_t> e;
    for (auto l : logs[t]) e.push_back(l + cfg.log_blowup_factor);
    std::string err = merkle_verify(pf.commitments[t], e, qpos, pf.queried_values[t], pf.decommitments[t]);
    if (!err.empty()) return "Merkle(tree " + std::to_string(t) + "): " + err;
  }
  // fri_answers: DEEP quotient of every size group at its query positions
  std::vector<size_t> cursor(4, 0);
  std::vector<std::vector<QM31>> answers;
  for (auto l : q_logs) {
    std::vector<std::vector<Sample>> cols;
    std::vector<size_t> ncols(4, 0);
    for (int t = 0; t < 4; t++)
      for (size_t c = 0; c < logs[t].size(); c++)
        if (logs[t][c] + cfg.log_blowup_factor == l) {
          ncols[t]++;
          std::vector<Sample> s;
          for (size_t k = 0; k < pts[t][c].size(); k++) s.push_back(Sample{pts[t][c][k], pf.sampled_values[t][c][k]});
          cols.push_back(s);
        }
    std::vector<QM31> ans;
    for (uint32_t qx : qpos[l]) {
      std::vector<uint32_t> row;
      for (int t = 0; t < 4; t++)
        for (size_t k = 0; k < ncols[t]; k++) {
          if (cursor[t] >= pf.queried_values[t].size()) return "InvalidStructure(queried values)";
          row.push_back(pf.queried_values[t][cursor[t]++]);
        }
      ans.push_back(row_quotient(cols, qcoeff, row, domain_point(l, qx)));
    }
    answers.push_back(ans);
  }
  // FRI first layer: rebuild the pairs, check their decommitment, fold every column into the line domain
  std::vector<std::vector<QM31>> folded_first;
  {
    size_t wi = 0;
    std::map<uint32_t, std::vector<uint32_t>> dpos;
    std::vector<uint32_t> dvals, col_logs;
    for (size_t k = 0; k < q_logs.size(); k++) {
      const uint32_t l = q_logs[k];
      std::vector<uint32_t> positions, starts;
      std::vector<std::array<QM31, 2>> pairs;
      if (!rebuild_evals(qpos[l], answers[k], pf.fri_first.fri_witness, wi, positions, pairs, starts)) return "Fri(FirstLayerEvaluationsInvalid)";
      dpos[l] = positions;
      for (auto& pr : pairs) for (auto& v : pr) { uint32_t w4[4]; v.to_u32(w4); dvals.insert(dvals.end(), w4, w4 + 4); }
      col_logs.insert(col_logs.end(), 4, l);
      std::vector<QM31> f;
      for (size_t s = 0; s < pairs.size(); s++) {
        CPoint<M31> p = domain_point(l, starts[s]);
        f.push_back((pairs[s][0] + pairs[s][1]) + circle_alpha * ((pairs[s][0] - pairs[s][1]) * inv(p.y)));
      }
      folded_first.push_back(f);
    }
    if (wi != pf.fri_first.fri_witness.size()) return "Fri(FirstLayerEvaluationsInvalid)";
    std::string err = merkle_verify(pf.fri_first.commitment, col_logs, dpos, dvals, pf.fri_first.decommitment);
    if (!err.empty()) return "Fri(FirstLayerCommitmentInvalid): " + err;
  }
  // inner layers
  FoldQueries lq = queries.fold(1);
  std::vector<QM31> evals(lq.positions.size());
  size_t col = 0;
  uint32_t layer_log = q_logs[0] - 1;
  const QM31 a2 = circle_alpha * circle_alpha;
  for (size_t li = 0; li < pf.fri_inner.size(); li++, layer_log--) {
    while (col < q_logs.size() && q_logs[col] - 1 == layer_log) {
      if (folded_first[col].size() != evals.size()) return "Fri(InnerLayerEvaluationsInvalid)";
      for (size_t i = 0; i < evals.size(); i++) evals[i] = evals[i] * a2 + folded_first[col][i];
      col++;
    }
    const FriLayerProofData& lp = pf.fri_inner[li];
    size_t wi = 0;
    std::vector<uint32_t> positions, starts;
    std::vector<std::array<QM31, 2>> pairs;
    if (!rebuild_evals(lq.positions, evals, lp.fri_witness, wi, positions, pairs, starts) || wi != lp.fri_witness.size())
      return "Fri(InnerLayerEvaluationsInvalid)";
    std::vector<uint32_t> dvals;
    for (auto& pr : pairs) for (auto& v : pr) { uint32_t w4[4]; v.to_u32(w4); dvals.insert(dvals.end(), w4, w4 + 4); }
    std::map<uint32_t, std::vector<uint32_t>> dpos;
    dpos[layer_log] = positions;
    std::string err = merkle_verify(lp.commitment, std::vector<uint32_t>(4, layer_log), dpos, dvals, lp.decommitment);
    if (!err.empty()) return "Fri(InnerLayerCommitmentInvalid " + std::to_string(li) + "): " + err;
    std::vector<QM31> nxt;
    for (size_t s = 0; s < pairs.size(); s++) {
      // LineDomain(half_odds(layer_log)) at bit-reversed position starts[s]
      uint32_t idx = subgroup_gen_index(layer_log + 2) + subgroup_gen_index(layer_log) * bit_reverse(starts[s], layer_log);
      M31 x = point_at_index(idx).x;
      nxt.push_back((pairs[s][0] + pairs[s][1]) + alphas[li] * ((pairs[s][0] - pairs[s][1]) * inv(x)));
    }
    evals = nxt;
    lq = lq.fold(1);
  }
  if (col != q_logs.size()) return "Fri(InvalidNumFriLayers)";
  {  // last layer: evaluate the line polynomial at the remaining query points
    const size_t n = pf.last_layer_poly.size();
    for (size_t i = 0; i < lq.positions.size(); i++) {
      uint32_t idx = subgroup_gen_index(layer_log + 2) + subgroup_gen_index(layer_log) * bit_reverse(lq.positions[i], layer_log);
      M31 x = point_at_index(idx).x;
      QM31 v;
      for (size_t j = 0; j < n; j++) {
        QM31 term = pf.last_layer_poly[j];
        M31 cur = x;
        for (uint32_t b = 0; b < pf.last_layer_log_size; b++) { if ((j >> (pf.last_layer_log_size - 1 - b)) & 1) term = term * cur; cur = double_x(cur); }
        v += term;
      }
      if (v != evals[i]) return "Fri(LastLayerEvaluationsInvalid)";
    }
  }
  return "";
}

}  // namespace cm
