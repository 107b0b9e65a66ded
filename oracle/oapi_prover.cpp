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
}
