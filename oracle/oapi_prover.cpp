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
};
// cfg = {pow_bits, log_blowup, log_last_layer, n_queries} or NULL
int orc_prove(const cm_prover_input* in, const uint32_t* cfg, orc_proof_handle** out) {
  try {
    PcsConfig c;
    if (cfg) { c.pow_bits = cfg[0]; c.log_blowup = cfg[1]; c.log_last_layer = cfg[2]; c.n_queries = cfg[3]; }
    ProveOutput po = prove_segment(*in, c);
    orc_proof_handle* h = new orc_proof_handle();
    h->words = serialize(po.proof);
    h->cells = po.cells;
    *out = h;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
uint64_t orc_proof_n_words(const orc_proof_handle* h) { return h->words.size(); }
void orc_proof_words(const orc_proof_handle* h, uint32_t* dst) { memcpy(dst, h->words.data(), h->words.size() * 4); }
uint64_t orc_proof_cells(const orc_proof_handle* h) { return h->cells; }
void orc_proof_free(orc_proof_handle* h) { delete h; }

int orc_verify(const uint32_t* words, uint64_t n) {
  try {
    Proof p;
    if (!deserialize(words, n, p)) { g_err = "malformed proof words"; return 2; }
    std::string e = verify_proof(p);
    if (!e.empty()) { g_err = e; return 1; }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 3; }
}

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
