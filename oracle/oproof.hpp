// ORACLE (test infrastructure).  Proof object + flat u32 serialisation shared (as a FORMAT, not code)
// with the product: tests compare the product's words with the oracle's, and feed the product's words to
// the oracle verifier.  Mirrors `Proof<H>` (crates/prover/src/lib.rs:61-73) and Stwo's
// CommitmentSchemeProof / FriProof / MerkleDecommitment (PARITY UNPINNED — field order of the Stwo
// structs is restated from upstream knowledge).
#pragma once
#include "ofield.hpp"
#include "omerkle.hpp"
#include <vector>

namespace orc {

struct PcsConfig {
  uint32_t pow_bits = 16, log_blowup = 1, log_last_layer = 0, n_queries = 80;  // REGULAR_96_BITS (prover_config.rs:13-20)
};
struct PublicEntry { bool present; uint32_t addr; uint32_t value[4]; uint32_t clock; };
struct PublicData {
  uint32_t initial_pc, initial_fp, final_pc, final_fp, clock, initial_root, final_root;
  std::vector<PublicEntry> program, input, output;
};
struct FriLayerProof {
  std::vector<QM31> fri_witness;
  MerkleDecommitment decommitment;
  Hash32 commitment;
};
struct Proof {
  PcsConfig config;
  std::vector<uint32_t> claim_log_sizes;   // 34, components in provers() order
  std::vector<QM31> claimed_sums;          // 34
  PublicData public_data;
  uint64_t interaction_pow = 0;
  std::vector<Hash32> commitments;                              // 4
  std::vector<std::vector<std::vector<QM31>>> sampled_values;  // tree -> column -> samples
  std::vector<MerkleDecommitment> decommitments;                // 4
  std::vector<std::vector<M31>> queried_values;                 // 4
  uint64_t proof_of_work = 0;
  FriLayerProof fri_first;
  std::vector<FriLayerProof> fri_inner;
  std::vector<QM31> last_layer_poly;
  uint32_t last_layer_log_size = 0;
};

constexpr uint32_t PROOF_MAGIC = 0x434d5031;  // "CMP1"

struct WordWriter {
  std::vector<uint32_t> w;
  void u(uint32_t x) { w.push_back(x); }
  void u64(uint64_t x) { u((uint32_t)x); u((uint32_t)(x >> 32)); }
  void q(const QM31& x) { uint32_t t[4]; x.to_u32(t); for (int i = 0; i < 4; i++) u(t[i]); }
  void h(const Hash32& x) { uint32_t t[8]; memcpy(t, x.data(), 32); for (int i = 0; i < 8; i++) u(t[i]); }
  void dec(const MerkleDecommitment& d) {
    u((uint32_t)d.hash_witness.size());
    for (auto& x : d.hash_witness) h(x);
    u((uint32_t)d.column_witness.size());
    for (auto& x : d.column_witness) u(x.v);
  }
  void layer(const FriLayerProof& l) {
    u((uint32_t)l.fri_witness.size());
    for (auto& x : l.fri_witness) q(x);
    dec(l.decommitment);
    h(l.commitment);
  }
  void entries(const std::vector<PublicEntry>& v) {
    u((uint32_t)v.size());
    for (auto& e : v) { u(e.present); u(e.addr); for (int i = 0; i < 4; i++) u(e.value[i]); u(e.clock); }
  }
};
inline std::vector<uint32_t> serialize(const Proof& p) {
  WordWriter W;
  W.u(PROOF_MAGIC);
  W.u(p.config.pow_bits); W.u(p.config.log_blowup); W.u(p.config.log_last_layer); W.u(p.config.n_queries);
  W.u((uint32_t)p.claim_log_sizes.size());
  for (auto x : p.claim_log_sizes) W.u(x);
  for (auto& x : p.claimed_sums) W.q(x);
  const PublicData& d = p.public_data;
  W.u(d.initial_pc); W.u(d.initial_fp); W.u(d.final_pc); W.u(d.final_fp); W.u(d.clock); W.u(d.initial_root); W.u(d.final_root);
  W.entries(d.program); W.entries(d.input); W.entries(d.output);
  W.u64(p.interaction_pow);
  W.u((uint32_t)p.commitments.size());
  for (auto& x : p.commitments) W.h(x);
  for (auto& tree : p.sampled_values) {
    W.u((uint32_t)tree.size());
    for (auto& col : tree) { W.u((uint32_t)col.size()); for (auto& s : col) W.q(s); }
  }
  for (auto& x : p.decommitments) W.dec(x);
  for (auto& x : p.queried_values) { W.u((uint32_t)x.size()); for (auto& v : x) W.u(v.v); }
  W.u64(p.proof_of_work);
  W.layer(p.fri_first);
  W.u((uint32_t)p.fri_inner.size());
  for (auto& l : p.fri_inner) W.layer(l);
  W.u((uint32_t)p.last_layer_poly.size());
  for (auto& x : p.last_layer_poly) W.q(x);
  W.u(p.last_layer_log_size);
  return W.w;
}

struct WordReader {
  const uint32_t* w; size_t n, i = 0; bool ok = true;
  uint32_t u() { if (i >= n) { ok = false; return 0; } return w[i++]; }
  uint64_t u64() { uint64_t lo = u(), hi = u(); return lo | (hi << 32); }
  QM31 q() { uint32_t a = u(), b = u(), c = u(), d = u(); return QM31::from_u32(a, b, c, d); }
  Hash32 h() { uint32_t t[8]; for (int k = 0; k < 8; k++) t[k] = u(); Hash32 x; memcpy(x.data(), t, 32); return x; }
  MerkleDecommitment dec() {
    MerkleDecommitment d;
    uint32_t nh = u(); if (nh > n) { ok = false; return d; }
    for (uint32_t k = 0; k < nh; k++) d.hash_witness.push_back(h());
    uint32_t nc = u(); if (nc > n) { ok = false; return d; }
    for (uint32_t k = 0; k < nc; k++) d.column_witness.push_back(M31(u()));
    return d;
  }
  FriLayerProof layer() {
    FriLayerProof l;
    uint32_t nw = u(); if (nw > n) { ok = false; return l; }
    for (uint32_t k = 0; k < nw; k++) l.fri_witness.push_back(q());
    l.decommitment = dec();
    l.commitment = h();
    return l;
  }
  std::vector<PublicEntry> entries() {
    std::vector<PublicEntry> v;
    uint32_t c = u(); if (c > n) { ok = false; return v; }
    for (uint32_t k = 0; k < c; k++) {
      PublicEntry e; e.present = u() != 0; e.addr = u();
      for (int t = 0; t < 4; t++) e.value[t] = u();
      e.clock = u();
      v.push_back(e);
    }
    return v;
  }
};
inline bool deserialize(const uint32_t* words, size_t n, Proof& p) {
  WordReader R{words, n};
  if (R.u() != PROOF_MAGIC) return false;
  p.config.pow_bits = R.u(); p.config.log_blowup = R.u(); p.config.log_last_layer = R.u(); p.config.n_queries = R.u();
  uint32_t nc = R.u(); if (nc > 1024) return false;
  p.claim_log_sizes.resize(nc);
  for (auto& x : p.claim_log_sizes) x = R.u();
  p.claimed_sums.resize(nc);
  for (auto& x : p.claimed_sums) x = R.q();
  PublicData& d = p.public_data;
  d.initial_pc = R.u(); d.initial_fp = R.u(); d.final_pc = R.u(); d.final_fp = R.u(); d.clock = R.u();
  d.initial_root = R.u(); d.final_root = R.u();
  d.program = R.entries(); d.input = R.entries(); d.output = R.entries();
  p.interaction_pow = R.u64();
  uint32_t nt = R.u(); if (nt > 16) return false;
  p.commitments.resize(nt);
  for (auto& x : p.commitments) x = R.h();
  p.sampled_values.resize(nt);
  for (auto& tree : p.sampled_values) {
    uint32_t ncol = R.u(); if (ncol > n) return false;
    tree.resize(ncol);
    for (auto& col : tree) { uint32_t ns = R.u(); if (ns > 16) return false; col.resize(ns); for (auto& s : col) s = R.q(); }
  }
  p.decommitments.resize(nt);
  for (auto& x : p.decommitments) x = R.dec();
  p.queried_values.resize(nt);
  for (auto& x : p.queried_values) { uint32_t c = R.u(); if (c > n) return false; x.resize(c); for (auto& v : x) v = M31(R.u()); }
  p.proof_of_work = R.u64();
  p.fri_first = R.layer();
  uint32_t nl = R.u(); if (nl > 64) return false;
  p.fri_inner.resize(nl);
  for (auto& l : p.fri_inner) l = R.layer();
  uint32_t ncf = R.u(); if (ncf > 1024) return false;
  p.last_layer_poly.resize(ncf);
  for (auto& x : p.last_layer_poly) x = R.q();
  p.last_layer_log_size = R.u();
  return R.ok && R.i == n;
}

}  // namespace orc
