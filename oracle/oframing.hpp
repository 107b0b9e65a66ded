// ORACLE (test infrastructure).  The oracle's copy of the named framing switches (same names and values as the product's
// cairo_m_amd/csrc/framing.hpp; parsed independently): every Stwo-side convention no in-tree reference vector settles.
//   mix_u64 = raw | u32s      hash_node = raw | rfc      sample_batch = insertion | sorted      pcs_mix = bql | blq
// Set with orc_set_framing("hash_node=rfc,...") (oapi_prover.cpp); "" / "default" = defaults (first value of each switch).
#pragma once
#include <string>

namespace orc {

struct Framing {
  bool mix_u64_u32s = false, hash_node_rfc = false, sample_batch_sorted = false, pcs_mix_blq = false;
};
inline Framing& framing_mut() { static Framing f; return f; }
inline const Framing& framing() { return framing_mut(); }
inline std::string set_framing(const std::string& spec) {
  Framing f;
  size_t pos = 0;
  while (pos <= spec.size()) {
    size_t end = spec.find(',', pos);
    if (end == std::string::npos) end = spec.size();
    std::string item = spec.substr(pos, end - pos);
    pos = end + 1;
    size_t a = item.find_first_not_of(' '), b = item.find_last_not_of(' ');
    if (a == std::string::npos) continue;
    item = item.substr(a, b - a + 1);
    if (item == "default") continue;
    size_t eq = item.find('=');
    if (eq == std::string::npos) return "framing: expected name=value in '" + item + "'";
    const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
    bool* dst = nullptr;
    const char *v0 = "", *v1 = "";
    if (k == "mix_u64") { dst = &f.mix_u64_u32s; v0 = "raw"; v1 = "u32s"; }
    else if (k == "hash_node") { dst = &f.hash_node_rfc; v0 = "raw"; v1 = "rfc"; }
    else if (k == "sample_batch") { dst = &f.sample_batch_sorted; v0 = "insertion"; v1 = "sorted"; }
    else if (k == "pcs_mix") { dst = &f.pcs_mix_blq; v0 = "bql"; v1 = "blq"; }
    else return "framing: unknown switch '" + k + "'";
    if (v == v0) *dst = false;
    else if (v == v1) *dst = true;
    else return "framing: bad value '" + v + "' for " + k;
  }
  framing_mut() = f;
  return "";
}

}  // namespace orc
