// ORACLE (test infrastructure).  Mixed-degree Blake2s Merkle tree: commit / decommit / verify.
// Restates Stwo `core::vcs::{blake2_merkle, prover, verifier}` (PARITY UNPINNED — Stwo not vendored).
// Reached from `tree_builder.commit(channel)` (crates/prover/src/prover.rs:73, 82, 102).
//
// hash_node framing ("candidate A", believed current at ab57a1c): state = 0^32;
//   if children: state = F(state, left||right, 0,0,0,0);
//   column values (LE u32) zero-padded to a multiple of 16 words, one F(state, chunk, 0,0,0,0) each.
// No IV / parameter block / length / finalisation flag.
#pragma once
#include "oblake2s.hpp"
#include "ofield.hpp"
#include "oframing.hpp"
#include <map>
#include <string>
#include <algorithm>

namespace orc {

inline Hash32 hash_node(const Hash32* left, const Hash32* right, const uint32_t* vals, size_t nvals) {
  if (framing().hash_node_rfc) {   // "candidate B" (framing switch hash_node=rfc): RFC 7693 Blake2s-256 of left || right || le32(values)
    std::vector<uint8_t> buf((left ? 64 : 0) + 4 * nvals);
    if (left) { memcpy(buf.data(), left->data(), 32); memcpy(buf.data() + 32, right->data(), 32); }
    if (nvals) memcpy(buf.data() + (left ? 64 : 0), vals, 4 * nvals);
    return blake2s256(buf);
  }
  uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (left) {
    uint32_t m[16];
    memcpy(m, left->data(), 32);
    memcpy(m + 8, right->data(), 32);
    b2s_compress(st, m, 0, 0, 0, 0);
  }
  for (size_t off = 0; off < nvals; off += 16) {
    uint32_t m[16] = {0};
    size_t c = std::min<size_t>(16, nvals - off);
    for (size_t i = 0; i < c; i++) m[i] = vals[off + i];
    b2s_compress(st, m, 0, 0, 0, 0);
  }
  Hash32 out;
  memcpy(out.data(), st, 32);
  return out;
}

using Column = std::vector<M31>;

inline uint32_t ilog2(size_t n) {
  uint32_t l = 0;
  while (((size_t)1 << (l + 1)) <= n) l++;
  return l;
}

struct MerkleDecommitment {
  std::vector<Hash32> hash_witness;
  std::vector<M31> column_witness;
};

struct MerkleProver {
  // layers[k] has 2^k hashes; layers[0][0] = root.
  std::vector<std::vector<Hash32>> layers;

  static MerkleProver commit(const std::vector<const Column*>& columns_in) {
    MerkleProver mp;
    if (columns_in.empty()) {
      mp.layers.push_back({hash_node(nullptr, nullptr, nullptr, 0)});
      return mp;
    }
    // stable sort by length descending
    std::vector<const Column*> cols(columns_in);
    std::stable_sort(cols.begin(), cols.end(),
                     [](const Column* a, const Column* b) { return a->size() > b->size(); });
    uint32_t max_log = ilog2(cols[0]->size());
    mp.layers.resize(max_log + 1);
    size_t ci = 0;
    for (int log = (int)max_log; log >= 0; log--) {
      std::vector<const Column*> layer_cols;
      while (ci < cols.size() && ilog2(cols[ci]->size()) == (uint32_t)log) layer_cols.push_back(cols[ci++]);
      size_t n = (size_t)1 << log;
      std::vector<Hash32>& out = mp.layers[log];
      out.resize(n);
      const std::vector<Hash32>* prev = (log < (int)max_log) ? &mp.layers[log + 1] : nullptr;
      std::vector<uint32_t> vals(layer_cols.size());
#pragma omp parallel for firstprivate(vals) schedule(static)
      for (size_t i = 0; i < n; i++) {
        for (size_t c = 0; c < layer_cols.size(); c++) vals[c] = (*layer_cols[c])[i].v;
        out[i] = hash_node(prev ? &(*prev)[2 * i] : nullptr, prev ? &(*prev)[2 * i + 1] : nullptr,
                           vals.data(), vals.size());
      }
    }
    return mp;
  }
  Hash32 root() const { return layers[0][0]; }

  // queries_per_log_size: log_size -> sorted unique positions.  Returns (queried_values, decommitment).
  std::pair<std::vector<M31>, MerkleDecommitment> decommit(
      const std::map<uint32_t, std::vector<size_t>>& queries_per_log_size,
      const std::vector<const Column*>& columns_in) const {
    std::vector<M31> queried_values;
    MerkleDecommitment d;
    std::vector<const Column*> cols(columns_in);
    std::stable_sort(cols.begin(), cols.end(),
                     [](const Column* a, const Column* b) { return a->size() > b->size(); });
    size_t ci = 0;
    std::vector<size_t> last_layer_queries;
    for (int layer_log = (int)layers.size() - 1; layer_log >= 0; layer_log--) {
      std::vector<size_t> layer_total_queries;
      std::vector<const Column*> layer_cols;
      while (ci < cols.size() && ilog2(cols[ci]->size()) == (uint32_t)layer_log) layer_cols.push_back(cols[ci++]);
      const std::vector<Hash32>* prev_hashes =
          ((size_t)layer_log + 1 < layers.size()) ? &layers[layer_log + 1] : nullptr;
      static const std::vector<size_t> empty;
      auto it = queries_per_log_size.find((uint32_t)layer_log);
      const std::vector<size_t>& colq = it == queries_per_log_size.end() ? empty : it->second;
      size_t pi = 0, qi = 0;
      const std::vector<size_t>& prevq = last_layer_queries;
      while (pi < prevq.size() || qi < colq.size()) {
        size_t node;
        if (pi < prevq.size() && qi < colq.size()) node = std::min(prevq[pi] / 2, colq[qi]);
        else if (pi < prevq.size()) node = prevq[pi] / 2;
        else node = colq[qi];
        if (prev_hashes) {
          if (pi < prevq.size() && prevq[pi] == 2 * node) pi++;
          else d.hash_witness.push_back((*prev_hashes)[2 * node]);
          if (pi < prevq.size() && prevq[pi] == 2 * node + 1) pi++;
          else d.hash_witness.push_back((*prev_hashes)[2 * node + 1]);
        }
        if (qi < colq.size() && colq[qi] == node) {
          qi++;
          for (auto c : layer_cols) queried_values.push_back((*c)[node]);
        } else {
          for (auto c : layer_cols) d.column_witness.push_back((*c)[node]);
        }
        layer_total_queries.push_back(node);
      }
      last_layer_queries = layer_total_queries;
    }
    return {queried_values, d};
  }
};

// Stwo `MerkleVerifier::verify`.  column_log_sizes in commitment (column) order.
// Returns empty string on success, else an error description.
inline std::string merkle_verify(const Hash32& root, const std::vector<uint32_t>& column_log_sizes,
                                 const std::map<uint32_t, std::vector<size_t>>& queries_per_log_size,
                                 const std::vector<M31>& queried_values, const MerkleDecommitment& d) {
  uint32_t max_log = 0;
  for (auto l : column_log_sizes) max_log = std::max(max_log, l);
  std::map<uint32_t, size_t> n_columns_per_log_size;
  for (auto l : column_log_sizes) n_columns_per_log_size[l]++;
  size_t qv = 0, hw = 0, cw = 0;
  std::vector<std::pair<size_t, Hash32>> last_layer_hashes;
  bool have_last = false;
  for (int layer_log = (int)max_log; layer_log >= 0; layer_log--) {
    size_t n_cols = 0;
    auto nc = n_columns_per_log_size.find((uint32_t)layer_log);
    if (nc != n_columns_per_log_size.end()) n_cols = nc->second;
    std::vector<std::pair<size_t, Hash32>> layer_total;
    static const std::vector<size_t> empty;
    auto it = queries_per_log_size.find((uint32_t)layer_log);
    const std::vector<size_t>& colq = it == queries_per_log_size.end() ? empty : it->second;
    size_t pi = 0, qi = 0;
    while (pi < last_layer_hashes.size() || qi < colq.size()) {
      size_t node;
      if (pi < last_layer_hashes.size() && qi < colq.size()) node = std::min(last_layer_hashes[pi].first / 2, colq[qi]);
      else if (pi < last_layer_hashes.size()) node = last_layer_hashes[pi].first / 2;
      else node = colq[qi];
      Hash32 l, r;
      bool has_children = have_last;
      if (has_children) {
        if (pi < last_layer_hashes.size() && last_layer_hashes[pi].first == 2 * node) l = last_layer_hashes[pi++].second;
        else { if (hw >= d.hash_witness.size()) return "WitnessTooShort"; l = d.hash_witness[hw++]; }
        if (pi < last_layer_hashes.size() && last_layer_hashes[pi].first == 2 * node + 1) r = last_layer_hashes[pi++].second;
        else { if (hw >= d.hash_witness.size()) return "WitnessTooShort"; r = d.hash_witness[hw++]; }
      }
      std::vector<uint32_t> vals(n_cols);
      if (qi < colq.size() && colq[qi] == node) {
        qi++;
        if (qv + n_cols > queried_values.size()) return "TooFewQueriedValues";
        for (size_t c = 0; c < n_cols; c++) vals[c] = queried_values[qv++].v;
      } else {
        if (cw + n_cols > d.column_witness.size()) return "WitnessTooShort";
        for (size_t c = 0; c < n_cols; c++) vals[c] = d.column_witness[cw++].v;
      }
      layer_total.push_back({node, hash_node(has_children ? &l : nullptr, has_children ? &r : nullptr, vals.data(), n_cols)});
    }
    last_layer_hashes = layer_total;
    have_last = true;
  }
  if (hw != d.hash_witness.size()) return "WitnessTooLong";
  if (qv != queried_values.size()) return "TooManyQueriedValues";
  if (cw != d.column_witness.size()) return "WitnessTooLong";
  if (last_layer_hashes.size() != 1 || last_layer_hashes[0].second != root) return "RootMismatch";
  return "";
}

}  // namespace orc
