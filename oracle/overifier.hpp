// ORACLE (test infrastructure).  verify_cairo_m (crates/prover/src/verifier.rs:17-95) + Stwo `verify`,
// CommitmentSchemeVerifier::verify_values, fri_answers, FriVerifier (PARITY UNPINNED — Stwo not
// vendored).  Used as the end-to-end self-check: proofs from the oracle prover AND from the HIP prover
// must be accepted.
#pragma once
#include "oprover.hpp"

namespace orc {

// quotient value of one row from the queried values of that row (accumulate_row_quotients)
struct RowQuotient {
  std::vector<SampleBatch> batches;
  struct LC { QM31 a, b, c; };
  std::vector<std::vector<LC>> lcs;
  std::vector<QM31> batch_coeff;
  void init(const std::vector<std::vector<std::pair<PointQ, QM31>>>& samples, QM31 random_coeff) {
    for (size_t c = 0; c < samples.size(); c++)
      for (auto& s : samples[c]) {
        size_t b = 0;
        for (; b < batches.size(); b++) if (same_point(batches[b].point, s.first)) break;
        if (b == batches.size()) batches.push_back(SampleBatch{s.first, {}});
        batches[b].cols.push_back({c, s.second});
      }
    if (framing().sample_batch_sorted) std::stable_sort(batches.begin(), batches.end(), [](const SampleBatch& a, const SampleBatch& b) { return point_less(a.point, b.point); });
    lcs.resize(batches.size());
    batch_coeff.resize(batches.size());
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
  }
  QM31 eval(const std::vector<M31>& row_values, PointM p) const {
    QM31 acc;
    for (size_t b = 0; b < batches.size(); b++) {
      CM31 prx = batches[b].point.x.a, pry = batches[b].point.y.a, pix = batches[b].point.x.b, piy = batches[b].point.y.b;
      CM31 dinv = ((prx - CM31(p.x)) * piy - (pry - CM31(p.y)) * pix).inverse();
      QM31 num;
      for (size_t k = 0; k < batches[b].cols.size(); k++) {
        const LC& lc = lcs[b][k];
        num += lc.c * row_values[batches[b].cols[k].first] - (lc.a * p.y + lc.b);
      }
      acc = acc * batch_coeff[b] + num.mul_cm31(dinv);
    }
    return acc;
  }
};

// compute_decommitment_positions_and_rebuild_evals
inline bool rebuild_evals(const std::vector<size_t>& queries, const std::vector<QM31>& query_evals, const std::vector<QM31>& witness,
                          size_t& wi, uint32_t fold_step, std::vector<size_t>& positions, std::vector<std::vector<QM31>>& subsets,
                          std::vector<size_t>& subset_starts) {
  size_t i = 0;
  while (i < queries.size()) {
    size_t start = (queries[i] >> fold_step) << fold_step;
    size_t j = i;
    while (j < queries.size() && (queries[j] >> fold_step) == (queries[i] >> fold_step)) j++;
    size_t qi = i;
    std::vector<QM31> sub;
    for (size_t pos = start; pos < start + ((size_t)1 << fold_step); pos++) {
      positions.push_back(pos);
      if (qi < j && queries[qi] == pos) { sub.push_back(query_evals[qi]); qi++; }
      else { if (wi >= witness.size()) return false; sub.push_back(witness[wi++]); }
    }
    subsets.push_back(sub);
    subset_starts.push_back(start);
    i = j;
  }
  return true;
}

// `expected`: the verifier's own PcsConfig (verify_cairo_m(proof, pcs_config), verifier.rs:17-31) — never the proof's.
inline std::string verify_proof(const Proof& pf, const PcsConfig& expected = PcsConfig()) {
  const PcsConfig& cfg = expected;
  if (pf.config.pow_bits != cfg.pow_bits || pf.config.log_blowup != cfg.log_blowup || pf.config.n_queries != cfg.n_queries ||
      pf.config.log_last_layer != cfg.log_last_layer) return "InvalidStructure(config)";
  if (cfg.n_queries == 0 || cfg.n_queries > 4096 || cfg.pow_bits > 64 || cfg.log_last_layer > 20 || cfg.log_blowup < 1 || cfg.log_blowup > 4)
    return "InvalidStructure(config)";
  if (pf.claim_log_sizes.size() != (size_t)air::N_COMPONENTS || pf.commitments.size() != 4) return "InvalidStructure";
  Channel ch;
  ch.mix_u64(cfg.pow_bits);
  ch.mix_u64(cfg.log_blowup);
  if (framing().pcs_mix_blq) { ch.mix_u64(cfg.log_last_layer); ch.mix_u64(cfg.n_queries); }
  else { ch.mix_u64(cfg.n_queries); ch.mix_u64(cfg.log_last_layer); }
  mix_public_data(pf.public_data, ch);
  // column log sizes per tree
  std::vector<std::vector<uint32_t>> logs(4);
  for (int i = 0; i < air::N_PREPROC; i++) logs[0].push_back(air::PREPROC_LOG[i]);
  struct Loc { size_t tr0, it0; };
  std::vector<Loc> loc(air::N_COMPONENTS);
  for (int c = 0; c < air::N_COMPONENTS; c++) {
    const air::ComponentInfo& info = air::component_info(c);
    loc[c].tr0 = logs[1].size(); loc[c].it0 = logs[2].size();
    for (int k = 0; k < info.n_trace; k++) logs[1].push_back(pf.claim_log_sizes[c]);
    for (int k = 0; k < info.n_interaction; k++) logs[2].push_back(pf.claim_log_sizes[c]);
  }
  ch.mix_root(pf.commitments[0]);
  for (auto l : pf.claim_log_sizes) ch.mix_u64(l);
  ch.mix_root(pf.commitments[1]);
  ch.mix_u64(pf.interaction_pow);
  if (ch.trailing_zeros() < 2) return "ProofOfWork(interaction)";
  Relations rel = draw_relations(ch);
  {
    QM31 s = initial_logup_sum(pf.public_data, rel);
    for (auto& c : pf.claimed_sums) s += c;
    if (!s.is_zero()) return "InvalidLogupSum";
  }
  for (auto& c : pf.claimed_sums) ch.mix_felts(&c, 1);
  ch.mix_root(pf.commitments[2]);
  // ---- stwo verify ----
  QM31 random_coeff = ch.draw_felt();
  uint32_t max_log = 0;
  for (auto l : pf.claim_log_sizes) max_log = std::max(max_log, l);
  logs[3].assign(4, max_log + 1);
  ch.mix_root(pf.commitments[3]);
  PointQ oods = random_point(ch);
  std::vector<std::vector<std::vector<PointQ>>> pts(4);
  pts[0].assign(logs[0].size(), {oods});
  pts[1].assign(logs[1].size(), {oods});
  pts[2].assign(logs[2].size(), {oods});
  for (int c = 0; c < air::N_COMPONENTS; c++) {
    int ni = air::component_info(c).n_interaction;
    PointQ prev = oods + into_ef(CanonicCoset(pf.claim_log_sizes[c]).step().conjugate());
    for (int k = ni - 4; k < ni; k++) pts[2][loc[c].it0 + k] = {prev, oods};
  }
  pts[3].assign(4, {oods});
  if (pf.sampled_values.size() != 4) return "InvalidStructure";
  for (int t = 0; t < 4; t++) {
    if (pf.sampled_values[t].size() != logs[t].size()) return "InvalidStructure(sampled cols)";
    for (size_t c = 0; c < logs[t].size(); c++) if (pf.sampled_values[t][c].size() != pts[t][c].size()) return "InvalidStructure(samples)";
  }
  // composition OODS check
  {
    size_t total = 0;
    for (int c = 0; c < air::N_COMPONENTS; c++) total += air::component_info(c).n_constraints;
    std::vector<QM31> powers(total);
    QM31 cur = QM31::one();
    for (size_t g = total; g-- > 0;) { powers[g] = cur; cur = cur * random_coeff; }
    QM31 c4[4] = {pf.sampled_values[3][0][0], pf.sampled_values[3][1][0], pf.sampled_values[3][2][0], pf.sampled_values[3][3][0]};
    QM31 comp = PointEval::combine_ef(c4);
    QM31 ppv[air::N_PREPROC];
    for (int i = 0; i < air::N_PREPROC; i++) ppv[i] = pf.sampled_values[0][i][0];
    QM31 sum;
    size_t g = 0;
    for (int c = 0; c < air::N_COMPONENTS; c++) {
      const air::ComponentInfo& info = air::component_info(c);
      std::vector<QM31> tr, it;
      for (int k = 0; k < info.n_trace; k++) tr.push_back(pf.sampled_values[1][loc[c].tr0 + k][0]);
      for (int k = 0; k < info.n_interaction; k++) for (auto& s : pf.sampled_values[2][loc[c].it0 + k]) it.push_back(s);
      QM31 shift = pf.claimed_sums[c] * M31((uint32_t)1 << pf.claim_log_sizes[c]).inverse();
      QM31 num = point_eval_dispatch(c, tr.data(), it.data(), ppv, rel, &powers[g], info.n_base_constraints, shift);
      sum += num * coset_vanishing<QM31>(CanonicCoset(pf.claim_log_sizes[c]).coset, oods, into_ef).inverse();
      g += info.n_constraints;
    }
    if (sum != comp) return "OodsNotMatching";
  }
  // ---- verify_values ----
  {
    std::vector<QM31> flat;
    for (auto& t : pf.sampled_values) for (auto& c : t) for (auto& s : c) flat.push_back(s);
    ch.mix_felts(flat);
  }
  QM31 qcoeff = ch.draw_felt();
  std::set<uint32_t, std::greater<uint32_t>> ext_logs;  // distinct LDE sizes, descending
  for (int t = 0; t < 4; t++) for (auto l : logs[t]) ext_logs.insert(l + cfg.log_blowup);
  std::vector<uint32_t> q_logs(ext_logs.begin(), ext_logs.end());
  // FRI commit phase
  ch.mix_root(pf.fri_first.commitment);
  QM31 circle_alpha = ch.draw_felt();
  uint32_t last_log = cfg.log_last_layer + cfg.log_blowup;
  if (q_logs[0] < last_log + 1 || pf.fri_inner.size() != (size_t)(q_logs[0] - 1 - last_log)) return "FRI: InvalidNumFriLayers";
  std::vector<QM31> alphas;
  for (auto& l : pf.fri_inner) { ch.mix_root(l.commitment); alphas.push_back(ch.draw_felt()); }
  if (pf.last_layer_poly.size() != ((size_t)1 << cfg.log_last_layer) || pf.last_layer_log_size != cfg.log_last_layer)
    return "FRI: LastLayerDegreeInvalid";
  ch.mix_felts(pf.last_layer_poly);
  ch.mix_u64(pf.proof_of_work);
  if (ch.trailing_zeros() < cfg.pow_bits) return "ProofOfWork";
  Queries queries = Queries::generate(ch, q_logs[0], cfg.n_queries);
  std::map<uint32_t, std::vector<size_t>> qpos;
  for (auto l : q_logs) qpos[l] = queries.fold(queries.log_domain_size - l).positions;
  // Merkle decommitments of the 4 trees
  for (int t = 0; t < 4; t++) {
    std::vector<uint32_t> ext;
    for (auto l : logs[t]) ext.push_back(l + cfg.log_blowup);
    std::string e = merkle_verify(pf.commitments[t], ext, qpos, pf.queried_values[t], pf.decommitments[t]);
    if (!e.empty()) return "Merkle(tree " + std::to_string(t) + "): " + e;
  }
  // fri_answers
  std::vector<size_t> qv_cursor(4, 0);
  std::vector<std::vector<QM31>> answers;  // per q_log: quotient value at each query
  for (auto l : q_logs) {
    std::vector<std::vector<std::pair<PointQ, QM31>>> smp;
    std::vector<size_t> ncols(4, 0);
    for (int t = 0; t < 4; t++)
      for (size_t c = 0; c < logs[t].size(); c++)
        if (logs[t][c] + cfg.log_blowup == l) {
          ncols[t]++;
          std::vector<std::pair<PointQ, QM31>> s;
          for (size_t k = 0; k < pts[t][c].size(); k++) s.push_back({pts[t][c][k], pf.sampled_values[t][c][k]});
          smp.push_back(s);
        }
    RowQuotient rq;
    rq.init(smp, qcoeff);
    CircleDomain dom = CanonicCoset(l).circle_domain();
    std::vector<QM31> ans;
    for (size_t q : qpos[l]) {
      std::vector<M31> row;
      for (int t = 0; t < 4; t++)
        for (size_t k = 0; k < ncols[t]; k++) {
          if (qv_cursor[t] >= pf.queried_values[t].size()) return "InvalidStructure(queried values)";
          row.push_back(pf.queried_values[t][qv_cursor[t]++]);
        }
      ans.push_back(rq.eval(row, dom.at(bit_reverse_index(q, l))));
    }
    answers.push_back(ans);
  }
  // FRI first layer
  std::vector<std::vector<QM31>> folded_first;  // per column: folded value per folded query
  {
    size_t wi = 0;
    std::map<uint32_t, std::vector<size_t>> dpos;
    std::vector<M31> dvals;
    std::vector<uint32_t> col_logs;
    for (size_t k = 0; k < q_logs.size(); k++) {
      uint32_t l = q_logs[k];
      std::vector<size_t> positions, starts;
      std::vector<std::vector<QM31>> subsets;
      if (!rebuild_evals(qpos[l], answers[k], pf.fri_first.fri_witness, wi, 1, positions, subsets, starts)) return "FRI: FirstLayerEvaluationsInvalid";
      dpos[l] = positions;
      for (auto& s : subsets) for (auto& v : s) for (int c = 0; c < 4; c++) dvals.push_back(v.coord(c));
      for (int c = 0; c < 4; c++) col_logs.push_back(l);
      CircleDomain dom = CanonicCoset(l).circle_domain();
      std::vector<QM31> f;
      for (size_t s = 0; s < subsets.size(); s++) {
        PointM p = dom.at(bit_reverse_index(starts[s], l));
        QM31 f0 = subsets[s][0], f1 = subsets[s][1];
        f.push_back((f0 + f1) + circle_alpha * ((f0 - f1) * p.y.inverse()));
      }
      folded_first.push_back(f);
    }
    if (wi != pf.fri_first.fri_witness.size()) return "FRI: FirstLayerEvaluationsInvalid(extra)";
    std::string e = merkle_verify(pf.fri_first.commitment, col_logs, dpos, dvals, pf.fri_first.decommitment);
    if (!e.empty()) return "FRI first layer: " + e;
  }
  // inner layers
  Queries lq = queries.fold(1);
  std::vector<QM31> evals(lq.positions.size());
  size_t col = 0;
  uint32_t layer_log = q_logs[0] - 1;
  QM31 a2 = circle_alpha * circle_alpha;
  for (size_t li = 0; li < pf.fri_inner.size(); li++, layer_log--) {
    while (col < q_logs.size() && q_logs[col] - 1 == layer_log) {
      if (folded_first[col].size() != evals.size()) return "FRI: internal size mismatch";
      for (size_t i = 0; i < evals.size(); i++) evals[i] = evals[i] * a2 + folded_first[col][i];
      col++;
    }
    const FriLayerProof& lp = pf.fri_inner[li];
    size_t wi = 0;
    std::vector<size_t> positions, starts;
    std::vector<std::vector<QM31>> subsets;
    if (!rebuild_evals(lq.positions, evals, lp.fri_witness, wi, 1, positions, subsets, starts)) return "FRI: InnerLayerEvaluationsInvalid";
    if (wi != lp.fri_witness.size()) return "FRI: InnerLayerEvaluationsInvalid(extra)";
    std::vector<M31> dvals;
    for (auto& s : subsets) for (auto& v : s) for (int c = 0; c < 4; c++) dvals.push_back(v.coord(c));
    std::map<uint32_t, std::vector<size_t>> dpos;
    dpos[layer_log] = positions;
    std::string e = merkle_verify(lp.commitment, std::vector<uint32_t>(4, layer_log), dpos, dvals, lp.decommitment);
    if (!e.empty()) return "FRI inner layer " + std::to_string(li) + ": " + e;
    Coset lc = Coset::half_odds(layer_log);
    std::vector<QM31> nxt;
    for (size_t s = 0; s < subsets.size(); s++) {
      M31 x = lc.at(bit_reverse_index(starts[s], layer_log)).x;
      QM31 f0 = subsets[s][0], f1 = subsets[s][1];
      nxt.push_back((f0 + f1) + alphas[li] * ((f0 - f1) * x.inverse()));
    }
    evals = nxt;
    lq = lq.fold(1);
  }
  if (col != q_logs.size()) return "FRI: columns left unfolded";
  // last layer: constant (log_last_layer = 0) or general line poly
  {
    Coset lc = Coset::half_odds(layer_log);
    for (size_t i = 0; i < lq.positions.size(); i++) {
      M31 x = lc.at(bit_reverse_index(lq.positions[i], layer_log)).x;
      // LinePoly::eval_at_point = fold(coeffs, [x, pi(x), pi^2(x), ...]): the top index bit pairs with x
      QM31 v;
      size_t n = pf.last_layer_poly.size();
      for (size_t j = 0; j < n; j++) {
        QM31 term = pf.last_layer_poly[j];
        M31 cur = x;
        for (uint32_t b = 0; b < pf.last_layer_log_size; b++) { if ((j >> (pf.last_layer_log_size - 1 - b)) & 1) term = term * cur; cur = double_x(cur); }
        v += term;
      }
      if (v != evals[i]) return "FRI: LastLayerEvaluationsInvalid";
    }
  }
  return "";
}

}  // namespace orc
