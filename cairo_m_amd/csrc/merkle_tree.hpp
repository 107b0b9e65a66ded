// Device-resident mixed-degree Merkle tree (Stwo MerkleProver over Blake2sMerkleHasher):
// layers live in HBM (8 u32 per node); commit() drives k_merkle_layer from the largest layer down;
// decommit() replays Stwo's decommitment walk on the host and fetches only the needed hashes /
// column values with two gather kernels.
// Reference call sites: tree_builder.commit (crates/prover/src/prover.rs:73, 82, 102) and
// commitment_scheme.prove_values -> tree.decommit inside stwo `prove` (prover.rs:131).
#pragma once
#include "engine.hpp"
#include <algorithm>
#include <map>
#include <array>
#include <string.h>

namespace cm {

using Hash32 = std::array<uint8_t, 32>;

struct MerkleDecommitment {
  std::vector<Hash32> hash_witness;
  std::vector<uint32_t> column_witness;
};

struct MerkleTree {
  std::vector<DevBuf> layers;            // layers[k]: 2^k nodes
  std::vector<const uint32_t*> cols;     // columns sorted by size desc (stable)
  std::vector<uint32_t> col_logs;        // same order
  DevBuf d_cols;                         // device copy of `cols`
  DevBuf d_layers;                       // device array of layer pointers

  static uint32_t ilog2(uint64_t n) { uint32_t l = 0; while ((1ull << (l + 1)) <= n) l++; return l; }

  void commit(const std::vector<const uint32_t*>& columns, const std::vector<uint32_t>& logs, hipStream_t st) {
    std::vector<uint32_t> order(columns.size());
    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return logs[a] > logs[b]; });
    cols.clear(); col_logs.clear();
    for (auto i : order) { cols.push_back(columns[i]); col_logs.push_back(logs[i]); }
    uint32_t max_log = cols.empty() ? 0 : col_logs[0];
    layers.clear();
    layers.resize(max_log + 1);
    d_cols = upload(cols, st);
    size_t ci = 0;
    const int tail_top = (int)std::min<uint32_t>(max_log, MERKLE_TAIL_LOG);
    for (int log = (int)max_log; log > tail_top; log--) {
      size_t c0 = ci;
      while (ci < cols.size() && col_logs[ci] == (uint32_t)log) ci++;
      layers[log].alloc((size_t)32 << log);
      const uint32_t* prev = (log < (int)max_log) ? layers[log + 1].u32() : nullptr;
      merkle_layer((uint32_t)log, prev, d_cols.as<const uint32_t*>() + c0, (uint32_t)(ci - c0), layers[log].u32(), st);
    }
    {
      // layers 2^tail_top .. 2^0: one fused launch
      MerkleTailArgs a;
      a.top_log = (uint32_t)tail_top;
      a.prev = (tail_top < (int)max_log) ? layers[tail_top + 1].u32() : nullptr;
      a.cols = d_cols.as<const uint32_t*>();
      for (int log = tail_top; log >= 0; log--) {
        a.col_begin[log] = (uint32_t)ci;
        while (ci < cols.size() && col_logs[ci] == (uint32_t)log) ci++;
        a.col_end[log] = (uint32_t)ci;
        layers[log].alloc((size_t)32 << log);
        a.layers[log] = layers[log].u32();
      }
      merkle_tail(a, st);
    }
    std::vector<const uint32_t*> lp(layers.size());
    for (size_t i = 0; i < layers.size(); i++) lp[i] = layers[i].u32();
    d_layers = upload(lp, st);
  }
  void root(uint8_t out[32], hipStream_t st) const {
    CM_HIP(hipMemcpyAsync(out, layers[0].p, 32, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
  }

  // queries_per_log_size: log -> sorted unique positions.
  void decommit(const std::map<uint32_t, std::vector<uint32_t>>& queries_per_log_size, std::vector<uint32_t>& queried_values,
                MerkleDecommitment& d, hipStream_t st) const {
    // pass 1: symbolic walk, record what to fetch
    std::vector<uint32_t> h_layer, h_node;          // hash witness requests
    std::vector<uint32_t> v_col, v_row;             // value requests, in walk order
    std::vector<uint8_t> v_is_query;                // 1 = goes to queried_values, 0 = column_witness
    size_t ci = 0;
    std::vector<uint32_t> last;
    for (int layer_log = (int)layers.size() - 1; layer_log >= 0; layer_log--) {
      size_t c0 = ci;
      while (ci < cols.size() && col_logs[ci] == (uint32_t)layer_log) ci++;
      bool has_prev = (size_t)layer_log + 1 < layers.size();
      static const std::vector<uint32_t> empty;
      auto it = queries_per_log_size.find((uint32_t)layer_log);
      const std::vector<uint32_t>& colq = it == queries_per_log_size.end() ? empty : it->second;
      std::vector<uint32_t> total;
      size_t pi = 0, qi = 0;
      while (pi < last.size() || qi < colq.size()) {
        uint32_t node;
        if (pi < last.size() && qi < colq.size()) node = std::min(last[pi] / 2, colq[qi]);
        else if (pi < last.size()) node = last[pi] / 2;
        else node = colq[qi];
        if (has_prev) {
          if (pi < last.size() && last[pi] == 2 * node) pi++;
          else { h_layer.push_back(layer_log + 1); h_node.push_back(2 * node); }
          if (pi < last.size() && last[pi] == 2 * node + 1) pi++;
          else { h_layer.push_back(layer_log + 1); h_node.push_back(2 * node + 1); }
        }
        bool isq = qi < colq.size() && colq[qi] == node;
        if (isq) qi++;
        for (size_t c = c0; c < ci; c++) { v_col.push_back((uint32_t)c); v_row.push_back(node); v_is_query.push_back(isq); }
        total.push_back(node);
      }
      last.swap(total);
    }
    // pass 2: fetch
    std::vector<uint32_t> hashes(h_layer.size() * 8), vals(v_col.size());
    if (!h_layer.empty()) {
      DevBuf dl = upload(h_layer, st), dn = upload(h_node, st), dout(hashes.size() * 4);
      gather_hashes(d_layers.as<const uint32_t*>(), dl.u32(), dn.u32(), (uint32_t)h_layer.size(), dout.u32(), st);
      CM_HIP(hipMemcpyAsync(hashes.data(), dout.p, hashes.size() * 4, hipMemcpyDeviceToHost, st));
      CM_HIP(hipStreamSynchronize(st));
    }
    if (!v_col.empty()) {
      DevBuf dc = upload(v_col, st), dr = upload(v_row, st), dout(vals.size() * 4);
      gather_values(d_cols.as<const uint32_t*>(), dc.u32(), dr.u32(), (uint32_t)v_col.size(), dout.u32(), st);
      CM_HIP(hipMemcpyAsync(vals.data(), dout.p, vals.size() * 4, hipMemcpyDeviceToHost, st));
      CM_HIP(hipStreamSynchronize(st));
    }
    d.hash_witness.resize(h_layer.size());
    for (size_t i = 0; i < h_layer.size(); i++) memcpy(d.hash_witness[i].data(), &hashes[8 * i], 32);
    for (size_t i = 0; i < vals.size(); i++) {
      if (v_is_query[i]) queried_values.push_back(vals[i]);
      else d.column_witness.push_back(vals[i]);
    }
  }
};

}  // namespace cm
