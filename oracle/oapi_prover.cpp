// ORACLE (test infrastructure): extern "C" surface for the whole-segment CPU prover/verifier.
#include "oprover.hpp"
#include "overifier.hpp"
using namespace orc;

static thread_local std::string g_err;

extern "C" {
const char* orc_last_error() { return g_err.c_str(); }

struct orc_proof_handle {
  std::vector<uint32_t> words;
  uint64_t cells;
  std::string transcript_json;
};
// same JSON as the product's cm_proof_transcript: [{"op", "digest" (hex, after the call), "n_words", "words"}, ...]
static std::string transcript_json(const TranscriptLog& log) {
  std::string out = "[";
  char buf[64];
  for (size_t i = 0; i < log.size(); i++) {
    const TranscriptEntry& e = log[i];
    out += i ? ",\n {\"op\": \"" : "{\"op\": \"";
    out += e.op;
    out += "\", \"digest\": \"";
    for (int k = 0; k < 32; k++) { snprintf(buf, sizeof(buf), "%02x", e.digest[k]); out += buf; }
    snprintf(buf, sizeof(buf), "\", \"n_words\": %u, \"words\": [", e.n_words);
    out += buf;
    for (size_t k = 0; k < e.words.size(); k++) { snprintf(buf, sizeof(buf), k ? ", %u" : "%u", e.words[k]); out += buf; }
    out += "]}";
  }
  return out + "]";
}
int orc_set_framing(const char* spec) {
  std::string e = set_framing(spec ? spec : "");
  if (!e.empty()) { g_err = e; return 1; }
  return 0;
}
// cfg = {pow_bits, log_blowup, log_last_layer, n_queries} or NULL
int orc_prove(const cm_prover_input* in, const uint32_t* cfg, orc_proof_handle** out) {
  try {
    PcsConfig c;
    if (cfg) { c.pow_bits = cfg[0]; c.log_blowup = cfg[1]; c.log_last_layer = cfg[2]; c.n_queries = cfg[3]; }
    ProveOutput po = prove_segment(*in, c);
    orc_proof_handle* h = new orc_proof_handle();
    h->words = serialize(po.proof);
    h->cells = po.cells;
    h->transcript_json = transcript_json(po.transcript);
    *out = h;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
uint64_t orc_proof_n_words(const orc_proof_handle* h) { return h->words.size(); }
void orc_proof_words(const orc_proof_handle* h, uint32_t* dst) { memcpy(dst, h->words.data(), h->words.size() * 4); }
uint64_t orc_proof_cells(const orc_proof_handle* h) { return h->cells; }
const char* orc_proof_transcript(const orc_proof_handle* h) { return h->transcript_json.c_str(); }
void orc_proof_free(orc_proof_handle* h) { delete h; }

// cfg: {pow_bits, log_blowup, log_last_layer, n_queries} the VERIFIER expects; NULL = REGULAR_96_BITS
int orc_verify(const uint32_t* words, uint64_t n, const uint32_t* cfg) {
  try {
    Proof p;
    if (!deserialize(words, n, p)) { g_err = "malformed proof words"; return 2; }
    PcsConfig c;
    if (cfg) { c.pow_bits = cfg[0]; c.log_blowup = cfg[1]; c.log_last_layer = cfg[2]; c.n_queries = cfg[3]; }
    std::string e = verify_proof(p, c);
    if (!e.empty()) { g_err = e; return 1; }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 3; }
}

// One row through a component's AIR description: constraint values (add_constraint order) and relation entries
// (relation id, multiplicity, n, values...) in add_to_relation order.  Returns the number of LogUp batches, -1 on error.
int orc_component_eval_row(int cid, const uint32_t* row, const uint32_t* preproc7, uint32_t* cons_out, uint32_t* n_cons,
                           uint32_t* ents_out, uint32_t* n_ent_words, uint32_t cap) {
  try {
    std::vector<uint32_t> cons, ents;
    int nb = dump_row_dispatch(cid, row, preproc7, cons, ents);
    if (cons.size() > cap || ents.size() > cap) { g_err = "orc_component_eval_row: output buffer too small"; return -1; }
    memcpy(cons_out, cons.data(), cons.size() * 4);
    memcpy(ents_out, ents.data(), ents.size() * 4);
    *n_cons = (uint32_t)cons.size();
    *n_ent_words = (uint32_t)ents.size();
    return nb;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
const char* orc_component_name(int cid) { return air::component_name(cid); }

// AIR consistency check without PCS (reference: debug_tools/assert_constraints.rs, tests/prover.rs:351-370):
// relations drawn from a default channel, every constraint must vanish on every row, LogUp sums cancel.
int orc_assert_constraints(const cm_prover_input* in) {
  try {
    std::string err;
    std::vector<ComponentTrace> cts = write_traces(*in, err);
    if (!err.empty()) { g_err = err; return 1; }
    Channel ch;
    Relations rel = draw_relations(ch);
    std::vector<Col> pp = preprocessed_columns();
    QM31 total = initial_logup_sum(make_public_data(*in), rel);
    for (auto& ct : cts) {
      gen_interaction_dispatch(ct, rel, pp);
      total += ct.claimed_sum;
      std::string e = assert_component_dispatch(ct, rel, pp);
      if (!e.empty()) { g_err = e; return 2; }
    }
    if (!total.is_zero()) { g_err = "LogUp sums do not cancel"; return 3; }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 4; }
}
// trace-domain columns of one component (for GPU trace-gen parity): returns n columns of 2^log each
int orc_component_trace(const cm_prover_input* in, int cid, uint32_t* log_out, uint32_t* n_cols_out, uint32_t* dst, uint64_t dst_cap) {
  try {
    std::string err;
    std::vector<ComponentTrace> cts = write_traces(*in, err);
    if (!err.empty()) { g_err = err; return 1; }
    const ComponentTrace& ct = cts[cid];
    *log_out = ct.log_size; *n_cols_out = (uint32_t)ct.trace.size();
    uint64_t need = ct.trace.size() << ct.log_size;
    if (dst && dst_cap >= need)
      for (size_t c = 0; c < ct.trace.size(); c++) memcpy(dst + (c << ct.log_size), ct.trace[c].data(), (size_t)4 << ct.log_size);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 4; }
}
// ---- per-component interaction trace / constraint accumulation (parity of cm_interaction_write / cm_constraints_accumulate) ----
static Relations relations_from_words(const uint32_t* w) {   // layout of cm_relations: z[8][4], alpha_pow[8][16][4]
  auto q = [](const uint32_t* x) { return QM31::from_m31s(M31(x[0]), M31(x[1]), M31(x[2]), M31(x[3])); };
  Relations r;
  for (int i = 0; i < air::N_RELATIONS; i++) r.z[i] = q(w + 4 * i);
  const uint32_t* a = w + 4 * air::N_RELATIONS;
  for (int i = 0; i < air::N_RELATIONS; i++)
    for (int k = 0; k < air::MAX_REL_SIZE; k++) r.alpha_pow[i][k] = q(a + 4 * (i * air::MAX_REL_SIZE + k));
  return r;
}
// interaction columns (n_interaction x 2^log words into dst) and claimed sum of one component
int orc_component_interaction(const cm_prover_input* in, int cid, const uint32_t* rel_words, uint32_t* dst, uint64_t dst_cap,
                              uint32_t* claimed_sum) {
  try {
    std::string err;
    std::vector<ComponentTrace> cts = write_traces(*in, err);
    if (!err.empty()) { g_err = err; return 1; }
    ComponentTrace& ct = cts[cid];
    std::vector<Col> pp = preprocessed_columns();
    gen_interaction_dispatch(ct, relations_from_words(rel_words), pp);
    uint64_t need = ct.interaction.size() << ct.log_size;
    if (dst_cap < need) { g_err = "buffer too small"; return 2; }
    for (size_t c = 0; c < ct.interaction.size(); c++) memcpy(dst + (c << ct.log_size), ct.interaction[c].data(), (size_t)4 << ct.log_size);
    for (int k = 0; k < 4; k++) claimed_sum[k] = ct.claimed_sum.coord(k).v;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 4; }
}
// acc (4 x 2^(log+1) words, zero-initialised here) = sum_k coeff_k * C_k / vanishing of one component on its evaluation domain
int orc_component_constraints(const cm_prover_input* in, int cid, const uint32_t* rel_words, const uint32_t* coeff_words,
                              uint32_t* acc_out) {
  try {
    std::string err;
    std::vector<ComponentTrace> cts = write_traces(*in, err);
    if (!err.empty()) { g_err = err; return 1; }
    ComponentTrace& ct = cts[cid];
    Relations rel = relations_from_words(rel_words);
    std::vector<Col> pp = preprocessed_columns();
    gen_interaction_dispatch(ct, rel, pp);
    PcsProver pcs;
    pcs.trees.resize(3);
    auto lde = [](const Col& evals) { Col c = evals; uint32_t lg = ilog2(c.size()); return evaluate(interpolate(std::move(c)), lg + 1); };
    for (auto& c : pp) pcs.trees[0].evals.push_back(lde(c));
    for (auto& c : ct.trace) pcs.trees[1].evals.push_back(lde(c));
    for (auto& c : ct.interaction) pcs.trees[2].evals.push_back(lde(c));
    const air::ComponentInfo& info = air::component_info(cid);
    std::vector<QM31> coeff(info.n_constraints);
    for (int k = 0; k < info.n_constraints; k++)
      coeff[k] = QM31::from_m31s(M31(coeff_words[4 * k]), M31(coeff_words[4 * k + 1]), M31(coeff_words[4 * k + 2]), M31(coeff_words[4 * k + 3]));
    size_t N = (size_t)2 << ct.log_size;
    std::vector<Col> acc(4, Col(N));
    accumulate_constraints_dispatch(ct, TraceLocation{0, 0}, pcs, rel, coeff.data(), acc);
    for (int k = 0; k < 4; k++) for (size_t i = 0; i < N; i++) acc_out[k * N + i] = acc[k][i].v;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 4; }
}
// FriOps::decompose restated from the Stwo CPU backend: in place on 4 x 2^log words; lambda out
void orc_fri_decompose(uint32_t* f, uint32_t log, uint32_t* lambda_out) {
  size_t N = (size_t)1 << log, H = N / 2;
  QM31 a, b;
  auto at = [&](size_t i) { return QM31::from_m31s(M31(f[i]), M31(f[N + i]), M31(f[2 * N + i]), M31(f[3 * N + i])); };
  for (size_t i = 0; i < H; i++) a += at(i);
  for (size_t i = H; i < N; i++) b += at(i);
  QM31 lambda = (a - b) * M31((uint32_t)N).inverse();
  for (size_t i = 0; i < N; i++) {
    QM31 v = i < H ? at(i) - lambda : at(i) + lambda;
    for (int k = 0; k < 4; k++) f[k * N + i] = v.coord(k).v;
  }
  for (int k = 0; k < 4; k++) lambda_out[k] = lambda.coord(k).v;
}
// Poseidon2 permutation of 16 words (KAT: crates/prover/tests/poseidon2.rs:14-34)
void orc_poseidon2_permute(uint32_t* state) {
  M31 s[16];
  for (int i = 0; i < 16; i++) s[i] = M31(state[i]);
  M31 o[448];
  // run the component witness and read the final state from the last 16 columns of the last full round
  uint32_t in[16];
  for (int i = 0; i < 16; i++) in[i] = s[i].v;
  air::Poseidon2C::witness<OrcOps>(in, 1, o);
  for (int i = 0; i < 16; i++) state[i] = o[air::Poseidon2C::N_TRACE - 16 + i].v;
}
// ---- per-op exports (parity tests of the FriOps / QuotientOps C-ABI entry points) ----
// cols: 4 coordinate columns, each 2^log (src) / 2^(log-1) (dst), contiguous
void orc_fold_circle_into_line(uint32_t* dst, const uint32_t* src, uint32_t log, const uint32_t* alpha) {
  size_t N = (size_t)1 << log, H = N / 2;
  std::vector<Col> d(4, Col(H)), s(4, Col(N));
  for (int k = 0; k < 4; k++) {
    for (size_t i = 0; i < H; i++) d[k][i] = M31(dst[k * H + i]);
    for (size_t i = 0; i < N; i++) s[k][i] = M31(src[k * N + i]);
  }
  fold_circle_into_line(d, s, log, QM31::from_m31s(M31(alpha[0]), M31(alpha[1]), M31(alpha[2]), M31(alpha[3])));
  for (int k = 0; k < 4; k++) for (size_t i = 0; i < H; i++) dst[k * H + i] = d[k][i].v;
}
void orc_fold_line(const uint32_t* src, uint32_t log, const uint32_t* alpha, uint32_t* out) {
  size_t N = (size_t)1 << log, H = N / 2;
  std::vector<Col> s(4, Col(N));
  for (int k = 0; k < 4; k++) for (size_t i = 0; i < N; i++) s[k][i] = M31(src[k * N + i]);
  std::vector<Col> o = fold_line(s, log, QM31::from_m31s(M31(alpha[0]), M31(alpha[1]), M31(alpha[2]), M31(alpha[3])));
  for (int k = 0; k < 4; k++) for (size_t i = 0; i < H; i++) out[k * H + i] = o[k][i].v;
}
// cols: n_cols columns of 2^log contiguous; samples given as batches in the same layout as cm_sample_batches
void orc_accumulate_quotients(uint32_t log, const uint32_t* cols, uint32_t n_cols, uint32_t n_batches, const uint32_t* points,
                              const uint32_t* batch_off, const uint32_t* col_index, const uint32_t* values,
                              const uint32_t* coeff, uint32_t* out) {
  size_t N = (size_t)1 << log;
  std::vector<Col> c(n_cols, Col(N));
  std::vector<const Col*> cp(n_cols);
  for (uint32_t j = 0; j < n_cols; j++) {
    for (size_t i = 0; i < N; i++) c[j][i] = M31(cols[j * N + i]);
    cp[j] = &c[j];
  }
  auto q = [](const uint32_t* w) { return QM31::from_m31s(M31(w[0]), M31(w[1]), M31(w[2]), M31(w[3])); };
  // the oracle groups samples by point in insertion order; feed them column-major so that the grouping
  // reproduces the given batches (the caller passes batches with increasing first-appearance order)
  std::vector<std::vector<std::pair<PointQ, QM31>>> samples(n_cols);
  for (uint32_t b = 0; b < n_batches; b++)
    for (uint32_t e = batch_off[b]; e < batch_off[b + 1]; e++) {
      PointQ pt; pt.x = q(points + 8 * b); pt.y = q(points + 8 * b + 4);
      samples[col_index[e]].push_back({pt, q(values + 4 * e)});
    }
  std::vector<Col> o = accumulate_quotients(log, cp, samples, q(coeff));
  for (int k = 0; k < 4; k++) for (size_t i = 0; i < N; i++) out[k * N + i] = o[k][i].v;
}
}
