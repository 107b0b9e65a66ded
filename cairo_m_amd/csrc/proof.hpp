// Proof object of the HIP prover + its two serialisations:
//   * proof_to_json  : serde layout of `Proof<Blake2sMerkleHasher>` (crates/prover/src/lib.rs:61-73) as emitted by
//                      `sonic_rs::to_string(&proof)` (crates/prover/src/main.rs:86-91): M31 as a number, CM31/QM31 as
//                      nested 2-tuples, hashes as 32-number arrays, Option::None as null.  Field order of the Stwo
//                      structs (StarkProof/CommitmentSchemeProof/FriProof/...) is restated from upstream (unpinned).
//   * proof_to_words : flat u32 stream used by the parity tests (same format as oracle/oproof.hpp).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <array>
#include "../../include/cairom_hip.h"
#include "field.hpp"
#include "merkle_tree.hpp"
#include "host_channel.hpp"
#include "air/air_common.hpp"

namespace cm {

struct PublicEntry { uint32_t present, addr, value[4], clock; };
struct PublicData {
  uint32_t initial_pc = 0, initial_fp = 0, final_pc = 0, final_fp = 0, clock = 0, initial_root = 0, final_root = 0;
  std::vector<PublicEntry> program, input, output;
};
struct FriLayerProofData {
  std::vector<QM31> fri_witness;
  MerkleDecommitment decommitment;
  Hash32 commitment;
};
// The sampled values of one column: one or two in every AIR of this prover (mask [0] or [-1, 0]).  A std::vector per column
// meant ~1 600 heap blocks per proof, allocated during OODS sampling and freed in cm_proof_free — between two lone proofs, with
// the GPU idle.  Two values inline; more (a foreign proof the verifier is asked about) spill into a vector.
struct SampleVec {
  QM31 in_[2];
  uint32_t n_ = 0;
  std::vector<QM31> more_;   // holds every value once there are more than two
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  const QM31* data() const { return more_.empty() ? in_ : more_.data(); }
  QM31* data() { return more_.empty() ? in_ : more_.data(); }
  void push_back(const QM31& v) {
    if (more_.empty() && n_ < 2) { in_[n_++] = v; return; }
    if (more_.empty()) more_.assign(in_, in_ + n_);
    more_.push_back(v);
    n_++;
  }
  QM31& operator[](size_t i) { return data()[i]; }
  const QM31& operator[](size_t i) const { return data()[i]; }
  QM31* begin() { return data(); }
  QM31* end() { return data() + n_; }
  const QM31* begin() const { return data(); }
  const QM31* end() const { return data() + n_; }
  QM31& front() { return data()[0]; }
  QM31& back() { return data()[n_ - 1]; }
  const QM31& front() const { return data()[0]; }
  const QM31& back() const { return data()[n_ - 1]; }
};
struct ProofData {
  cm_pcs_config config;
  std::vector<uint32_t> claim_log_sizes;
  std::vector<QM31> claimed_sums;
  PublicData public_data;
  uint64_t interaction_pow = 0;
  std::vector<Hash32> commitments;
  std::vector<std::vector<SampleVec>> sampled_values;   // [tree][column] -> values in mask order
  std::vector<MerkleDecommitment> decommitments;
  std::vector<std::vector<uint32_t>> queried_values;
  uint64_t proof_of_work = 0;
  FriLayerProofData fri_first;
  std::vector<FriLayerProofData> fri_inner;
  std::vector<QM31> last_layer_poly;
  uint32_t last_layer_log_size = 0;
  // statistics (not part of the proof)
  hostch::TranscriptLog transcript;   // every Fiat-Shamir step of this proof, when cm_set_transcript_log(1) (host_channel.hpp)
  uint64_t cells = 0, steps = 0;
  std::vector<double> phase_ms;
  double total_ms = 0;
};

// [{"op": "mix_u64", "digest": "<64 hex digits: the channel digest after the call>", "n_words": 2, "words": [..first <= 16..]}, ...]
inline std::string transcript_to_json(const hostch::TranscriptLog& log) {
  std::string out = "[";
  char buf[80];
  for (size_t i = 0; i < log.size(); i++) {
    const auto& e = log[i];
    out += i ? ",\n {\"op\": \"" : "{\"op\": \"";
    out += e.op;
    out += "\", \"digest\": \"";
    for (int k = 0; k < 32; k++) { snprintf(buf, sizeof(buf), "%02x", e.digest[k]); out += buf; }
    snprintf(buf, sizeof(buf), "\", \"n_words\": %u, \"words\": [", e.n_words);
    out += buf;
    for (size_t k = 0; k < e.words.size(); k++) { snprintf(buf, sizeof(buf), k ? ", %u" : "%u", e.words[k]); out += buf; }
    out += "]}";
  }
  out += "]";
  return out;
}

inline std::vector<uint32_t> proof_to_words(const ProofData& p) {
  std::vector<uint32_t> w;
  auto u = [&](uint32_t x) { w.push_back(x); };
  auto u64 = [&](uint64_t x) { u((uint32_t)x); u((uint32_t)(x >> 32)); };
  auto q = [&](const QM31& x) { uint32_t t[4]; x.to_u32(t); for (int i = 0; i < 4; i++) u(t[i]); };
  auto h = [&](const Hash32& x) { uint32_t t[8]; memcpy(t, x.data(), 32); for (int i = 0; i < 8; i++) u(t[i]); };
  auto dec = [&](const MerkleDecommitment& d) {
    u((uint32_t)d.hash_witness.size());
    for (auto& x : d.hash_witness) h(x);
    u((uint32_t)d.column_witness.size());
    for (auto x : d.column_witness) u(x);
  };
  auto layer = [&](const FriLayerProofData& l) {
    u((uint32_t)l.fri_witness.size());
    for (auto& x : l.fri_witness) q(x);
    dec(l.decommitment);
    h(l.commitment);
  };
  auto entries = [&](const std::vector<PublicEntry>& v) {
    u((uint32_t)v.size());
    for (auto& e : v) { u(e.present); u(e.addr); for (int i = 0; i < 4; i++) u(e.value[i]); u(e.clock); }
  };
  u(0x434d5031);
  u(p.config.pow_bits); u(p.config.log_blowup_factor); u(p.config.log_last_layer_degree_bound); u(p.config.n_queries);
  u((uint32_t)p.claim_log_sizes.size());
  for (auto x : p.claim_log_sizes) u(x);
  for (auto& x : p.claimed_sums) q(x);
  const PublicData& d = p.public_data;
  u(d.initial_pc); u(d.initial_fp); u(d.final_pc); u(d.final_fp); u(d.clock); u(d.initial_root); u(d.final_root);
  entries(d.program); entries(d.input); entries(d.output);
  u64(p.interaction_pow);
  u((uint32_t)p.commitments.size());
  for (auto& x : p.commitments) h(x);
  for (auto& tree : p.sampled_values) {
    u((uint32_t)tree.size());
    for (auto& col : tree) { u((uint32_t)col.size()); for (auto& s : col) q(s); }
  }
  for (auto& x : p.decommitments) dec(x);
  for (auto& x : p.queried_values) { u((uint32_t)x.size()); for (auto v : x) u(v); }
  u64(p.proof_of_work);
  layer(p.fri_first);
  u((uint32_t)p.fri_inner.size());
  for (auto& l : p.fri_inner) layer(l);
  u((uint32_t)p.last_layer_poly.size());
  for (auto& x : p.last_layer_poly) q(x);
  u(p.last_layer_log_size);
  return w;
}

// field names of the opcode claim structs = module names in macro order (components/opcodes/mod.rs:223-268)
inline const char* component_field_name(int cid) {
  static const char* names[air::N_COMPONENTS] = {
      "assert_eq_fp_imm", "call_abs_imm", "jmp_imm", "jnz_fp_imm", "ret", "store_imm", "store_fp_fp", "store_fp_imm",
      "double_deref_fp_imm", "double_deref_fp_fp", "store_frame_pointer", "u32_store_imm", "u32_store_add_fp_imm",
      "u32_store_mul_fp_imm", "u32_store_div_fp_imm", "u32_store_eq_fp_fp", "u32_store_eq_fp_imm", "u32_store_lt_fp_imm",
      "u32_store_lt_fp_fp", "u32_store_add_fp_fp", "u32_store_sub_fp_fp", "u32_store_mul_fp_fp", "u32_store_div_fp_fp",
      "u32_store_bitwise_fp_fp", "u32_store_bitwise_fp_imm", "store_le_fp_imm",
      "memory", "merkle", "clock_update", "poseidon2", "range_check_8", "range_check_16", "range_check_20", "bitwise"};
  return names[cid];
}

inline std::string proof_to_json(const ProofData& p) {
  std::string s;
  s.reserve(1 << 20);
  auto num = [&](uint64_t v) { s += std::to_string(v); };
  auto q = [&](const QM31& x) {
    s += "[["; num(x.a.a.v); s += ','; num(x.a.b.v); s += "],["; num(x.b.a.v); s += ','; num(x.b.b.v); s += "]]";
  };
  auto hash = [&](const Hash32& h) {
    s += '[';
    for (int i = 0; i < 32; i++) { if (i) s += ','; num(h[i]); }
    s += ']';
  };
  auto dec = [&](const MerkleDecommitment& d) {
    s += "{\"hash_witness\":[";
    for (size_t i = 0; i < d.hash_witness.size(); i++) { if (i) s += ','; hash(d.hash_witness[i]); }
    s += "],\"column_witness\":[";
    for (size_t i = 0; i < d.column_witness.size(); i++) { if (i) s += ','; num(d.column_witness[i]); }
    s += "]}";
  };
  auto layer = [&](const FriLayerProofData& l) {
    s += "{\"fri_witness\":[";
    for (size_t i = 0; i < l.fri_witness.size(); i++) { if (i) s += ','; q(l.fri_witness[i]); }
    s += "],\"decommitment\":"; dec(l.decommitment);
    s += ",\"commitment\":"; hash(l.commitment);
    s += '}';
  };
  auto entries = [&](const std::vector<PublicEntry>& v) {
    s += '[';
    for (size_t i = 0; i < v.size(); i++) {
      if (i) s += ',';
      if (!v[i].present) { s += "null"; continue; }
      s += '['; num(v[i].addr); s += ',';
      q(QM31(M31(v[i].value[0]), M31(v[i].value[1]), M31(v[i].value[2]), M31(v[i].value[3])));
      s += ','; num(v[i].clock); s += ']';
    }
    s += ']';
  };
  auto claims = [&](bool interaction) {
    s += "{\"opcodes\":{";
    for (int c = 0; c < air::N_OPCODE_COMPONENTS; c++) {
      if (c) s += ',';
      s += '"'; s += component_field_name(c); s += "\":{";
      if (!interaction) { s += "\"log_size\":"; num(p.claim_log_sizes[c]); }
      else { s += "\"claimed_sum\":"; q(p.claimed_sums[c]); }
      s += '}';
    }
    s += '}';
    for (int c = air::N_OPCODE_COMPONENTS; c < air::N_COMPONENTS; c++) {
      s += ",\""; s += component_field_name(c); s += "\":{";
      if (!interaction) { s += "\"log_size\":"; num(p.claim_log_sizes[c]); }
      else { s += "\"claimed_sum\":"; q(p.claimed_sums[c]); }
      s += '}';
    }
    s += '}';
  };
  s += "{\"claim\":"; claims(false);
  s += ",\"interaction_claim\":"; claims(true);
  const PublicData& d = p.public_data;
  s += ",\"public_data\":{\"initial_registers\":{\"pc\":"; num(d.initial_pc); s += ",\"fp\":"; num(d.initial_fp);
  s += "},\"final_registers\":{\"pc\":"; num(d.final_pc); s += ",\"fp\":"; num(d.final_fp);
  s += "},\"clock\":"; num(d.clock); s += ",\"initial_root\":"; num(d.initial_root); s += ",\"final_root\":"; num(d.final_root);
  s += ",\"public_memory\":{\"program\":"; entries(d.program); s += ",\"input\":"; entries(d.input);
  s += ",\"output\":"; entries(d.output); s += "}}";
  s += ",\"stark_proof\":{\"config\":{\"pow_bits\":"; num(p.config.pow_bits);
  s += ",\"fri_config\":{\"log_blowup_factor\":"; num(p.config.log_blowup_factor);
  s += ",\"log_last_layer_degree_bound\":"; num(p.config.log_last_layer_degree_bound);
  s += ",\"n_queries\":"; num(p.config.n_queries); s += "}},\"commitments\":[";
  for (size_t i = 0; i < p.commitments.size(); i++) { if (i) s += ','; hash(p.commitments[i]); }
  s += "],\"sampled_values\":[";
  for (size_t t = 0; t < p.sampled_values.size(); t++) {
    if (t) s += ',';
    s += '[';
    for (size_t c = 0; c < p.sampled_values[t].size(); c++) {
      if (c) s += ',';
      s += '[';
      for (size_t k = 0; k < p.sampled_values[t][c].size(); k++) { if (k) s += ','; q(p.sampled_values[t][c][k]); }
      s += ']';
    }
    s += ']';
  }
  s += "],\"decommitments\":[";
  for (size_t i = 0; i < p.decommitments.size(); i++) { if (i) s += ','; dec(p.decommitments[i]); }
  s += "],\"queried_values\":[";
  for (size_t t = 0; t < p.queried_values.size(); t++) {
    if (t) s += ',';
    s += '[';
    for (size_t i = 0; i < p.queried_values[t].size(); i++) { if (i) s += ','; num(p.queried_values[t][i]); }
    s += ']';
  }
  s += "],\"proof_of_work\":"; num(p.proof_of_work);
  s += ",\"fri_proof\":{\"first_layer\":"; layer(p.fri_first);
  s += ",\"inner_layers\":[";
  for (size_t i = 0; i < p.fri_inner.size(); i++) { if (i) s += ','; layer(p.fri_inner[i]); }
  s += "],\"last_layer_poly\":{\"coeffs\":[";
  for (size_t i = 0; i < p.last_layer_poly.size(); i++) { if (i) s += ','; q(p.last_layer_poly[i]); }
  s += "],\"log_size\":"; num(p.last_layer_log_size);
  s += "}}},\"interaction_pow\":"; num(p.interaction_pow); s += '}';
  return s;
}

}  // namespace cm
