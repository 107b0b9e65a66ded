// Named switches for every Stwo-side framing choice that no in-tree reference vector settles (the Stwo half of
// the path is "parity unpinned" — Stwo @ ab57a1c is an empty submodule of the reference, `.gitmodules:1-3`).  The DEFAULT of
// each switch is the reading the restatement believes correct for that revision; the ALTERNATE is written out and tested
// (product == oracle under every combination, both verifiers accept), so that the day a reference-produced transcript
// (integration/prover-hip/tests/golden_dump.rs -> tests/golden/ref_*.json) disagrees at some step, the fix is a switch flip,
// not a rewrite.  The oracle has the same switches under the same names (oracle/oframing.hpp).
//
//   mix_u64       raw   : digest' = F(digest, [lo, hi, 0 x 14], t=0, f=0)   one raw compression (what SimdBackend::grind
//                         searches over; PoW predicate pinned by verifier.rs:55-58)                                  [default]
//                 u32s  : digest' = Blake2s256(digest || le32(lo) || le32(hi))  ( = mix_u32s(&[lo, hi]) )
//   hash_node     raw   : state = 0^32; [F(state, left || right)]; F per 16 column words, zero padded; t = f = 0        [default]
//                 rfc   : RFC 7693 Blake2s-256 of  left || right || le32(column values)
//   sample_batch  insertion : ColumnSampleBatch::new_vec groups the samples by point in first-seen order (IndexMap)    [default]
//                 sorted    : groups ordered by point (BTreeMap<CirclePoint<SecureField>, _>: x then y, QM31 words
//                             compared lexicographically as (a, b, c, d))
//   pcs_mix       bql   : PcsConfig::mix_into = mix_u64(pow_bits), mix_u64(log_blowup), mix_u64(n_queries),
//                         mix_u64(log_last_layer_degree_bound)                                                          [default]
//                 blq   : ... mix_u64(log_blowup), mix_u64(log_last_layer_degree_bound), mix_u64(n_queries) (struct field order)
//
// Selected process-wide with cm_set_framing("mix_u64=u32s,hash_node=rfc") or the environment variable CM_FRAMING (read once);
// "default" / "" restores the defaults.  Proofs made under different settings are different proofs: the verifier must run
// under the same setting.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>

namespace cm {

struct Framing {
  bool mix_u64_u32s = false;
  bool hash_node_rfc = false;
  bool sample_batch_sorted = false;
  bool pcs_mix_blq = false;
  std::string describe() const {
    return std::string("mix_u64=") + (mix_u64_u32s ? "u32s" : "raw") + ",hash_node=" + (hash_node_rfc ? "rfc" : "raw") +
           ",sample_batch=" + (sample_batch_sorted ? "sorted" : "insertion") + ",pcs_mix=" + (pcs_mix_blq ? "blq" : "bql");
  }
  // returns an error message, or "" on success
  static std::string parse(const char* spec, Framing& out) {
    Framing f;
    std::string s = spec ? spec : "";
    size_t pos = 0;
    while (pos < s.size()) {
      size_t end = s.find(',', pos);
      if (end == std::string::npos) end = s.size();
      std::string item = s.substr(pos, end - pos);
      pos = end + 1;
      while (!item.empty() && item.front() == ' ') item.erase(item.begin());
      while (!item.empty() && item.back() == ' ') item.pop_back();
      if (item.empty() || item == "default") continue;
      size_t eq = item.find('=');
      if (eq == std::string::npos) return "framing: expected name=value in '" + item + "'";
      const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
      auto pick = [&](const char* a, const char* b, bool& dst) -> bool {
        if (v == a) { dst = false; return true; }
        if (v == b) { dst = true; return true; }
        return false;
      };
      bool ok;
      if (k == "mix_u64") ok = pick("raw", "u32s", f.mix_u64_u32s);
      else if (k == "hash_node") ok = pick("raw", "rfc", f.hash_node_rfc);
      else if (k == "sample_batch") ok = pick("insertion", "sorted", f.sample_batch_sorted);
      else if (k == "pcs_mix") ok = pick("bql", "blq", f.pcs_mix_blq);
      else return "framing: unknown switch '" + k + "' (mix_u64, hash_node, sample_batch, pcs_mix)";
      if (!ok) return "framing: bad value '" + v + "' for " + k;
    }
    out = f;
    return "";
  }
};

// process-wide setting (framing.cpp part of capi.hip); reads CM_FRAMING on first use
const Framing& framing();                    // a malformed CM_FRAMING is a hard error (CmError) on first use, not a silent default
std::string set_framing(const char* spec);   // "" on success; refused while a proof or a verification is running
// Held by every prover / verifier for its lifetime: the kernels, the transcript and the quotient planning of ONE proof read the
// process-wide setting at their own points, so it must not change underneath them (cm_set_framing then returns an error).
struct FramingUse {
  FramingUse();
  ~FramingUse();
  FramingUse(const FramingUse&) = delete;
  FramingUse& operator=(const FramingUse&) = delete;
};

}  // namespace cm
