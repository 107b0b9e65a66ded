// Device-resident mixed-degree Merkle tree (Stwo MerkleProver over Blake2sMerkleHasher):
// layers live in HBM (8 u32 per node); commit() drives k_merkle_layer from the largest layer down and
// finishes the small layers in one fused launch; decommitment is split in two so that ALL trees of a proof
// share one gather launch: plan_decommit() replays Stwo's decommitment walk on the host and appends the
// device addresses it needs to a GatherBatch; after GatherBatch::run() finish_decommit() distributes the
// fetched words.
// Reference call sites: tree_builder.commit (crates/prover/src/prover.rs:73, 82, 102) and
// commitment_scheme.prove_values -> tree.decommit inside stwo `prove` (prover.rs:131).
#pragma once
#include "engine.hpp"
#include <algorithm>
#include <functional>
#include <map>
#include <array>
#include <string.h>

namespace cm {

using Hash32 = std::array<uint8_t, 32>;

struct MerkleDecommitment {
  std::vector<Hash32> hash_witness;
  std::vector<uint32_t> column_witness;
};

// One batched device->host gather of single words, 32-byte hashes and "row runs" (one row of a run of columns
// given by a DEVICE pointer table: the kernel expands it, so the host neither builds nor uploads one address
// per queried cell — a 2^22 proof queries ~3*10^5 cells).
struct GatherBatch {
  std::vector<const uint32_t*> word_addrs, hash_addrs;
  std::vector<RowRun> runs;
  size_t run_words = 0;
  // results (hashes: 8 words each): they stay in the calling thread's pinned landing buffer — valid until that thread's next
  // stage_download_async (a proof carries ~1 MB of witnesses; zero-filling and copying them into vectors first cost ~80 us
  // with the GPU idle)
  const uint32_t* words = nullptr;
  const uint32_t* hashes = nullptr;
  const uint32_t* run_out = nullptr;
  size_t add_word(const uint32_t* p) { word_addrs.push_back(p); return word_addrs.size() - 1; }
  size_t add_hash(const uint32_t* p) { hash_addrs.push_back(p); return hash_addrs.size() - 1; }
  // values d_cols[c][row], c < n_cols  ->  run_out[returned offset + c]
  size_t add_run(const uint32_t* const* d_cols, uint32_t n_cols, uint32_t row) {
    size_t off = run_words;
    runs.push_back(RowRun{d_cols, n_cols, row, (uint32_t)off, 0});
    run_words += n_cols;
    return off;
  }
  size_t total_words() const { return word_addrs.size() + hash_addrs.size() * 8 + run_words; }
  // one upload (all address tables), three gather launches into one output buffer, ONE device->host copy into `land` (pinned host
  // memory of at least total_words() words, or null = the thread's landing buffer); no wait: the results are valid once the stream
  // has passed this point.  The device buffers stay with the batch.
  DevBuf tables_, out_;
  void enqueue(hipStream_t st, uint32_t* land = nullptr) {
    const size_t nw = word_addrs.size(), nh = hash_addrs.size() * 8, nr = run_words, total = nw + nh + nr;
    if (!total) return;
    UploadBatch ub;
    const uint32_t** d_w = nullptr;
    const uint32_t** d_h = nullptr;
    RowRun* d_r = nullptr;
    if (nw) ub.add(word_addrs, &d_w);
    if (nh) ub.add(hash_addrs, &d_h);
    if (nr) ub.add(runs, &d_r);
    tables_ = ub.flush(st);
    out_.alloc(total * 4);
    if (nw) gather_words(d_w, (uint32_t)word_addrs.size(), 1, out_.u32(), st);
    if (nh) gather_words(d_h, (uint32_t)hash_addrs.size(), 8, out_.u32() + nw, st);
    if (nr) gather_runs(d_r, (uint32_t)runs.size(), out_.u32() + nw + nh, st);
    const uint32_t* host;
    if (land) { CM_HIP(hipMemcpyAsync(land, out_.p, total * 4, hipMemcpyDeviceToHost, st)); host = land; }
    else host = (const uint32_t*)stage_download_async(out_.p, total * 4, st);
    words = host;
    hashes = host + nw;
    run_out = host + nw + nh;
  }
  void run(hipStream_t st) {
    if (!total_words()) return;
    enqueue(st);
    CM_HIP(hipStreamSynchronize(st));
  }
};

struct DecommitPlan {
  size_t hash0 = 0, n_hash = 0;    // range in GatherBatch::hash_addrs
  // requested column values in decommitment order: segments of single words or of one row run
  struct Seg { uint32_t is_run, is_query; size_t off; uint32_t count; };
  std::vector<Seg> segs;
};

struct MerkleTree {
  std::vector<DevBuf> layers;            // layers[k]: 2^k nodes
  std::vector<const uint32_t*> cols;     // columns sorted by size desc (stable)
  std::vector<uint32_t> col_logs;        // same order
  DevBuf d_cols;                         // device copy of `cols`

  // Optional: recorded on the commit stream in front of the first layer of at most 2^MERKLE_PACE_LOG nodes — what follows is the
  // latency-bound top of the tree (~0.15-0.2 ms).  The prover's host thread waits for THIS event, not for the whole tree, before
  // it enqueues the next phase (Prover::pace): its wake-up and launch latency hide behind the tree top.
  hipEvent_t pace_ev = nullptr;
  bool pace_recorded = false;
  // (round 6) the leaf layer's buffer, handed in by a caller whose own kernel writes that layer (the DEEP-quotient kernel hashes the
  // rows it produces: Prover::deep_quotients): plan_commit() adopts it as layers[max_log] instead of allocating; the caller then
  // skips the plan's first launch (hi == lo == max_log)
  DevBuf leaf_prealloc;
  MerkleTopExtra top_extra;                      // extras of the tree-top launch (engine.hpp): set before plan_commit()
  bool top_launch_has_extras = false, top_launch_has_fold = false;   // what the last plan_commit() could place
  const uint32_t* const* d_cols_view = nullptr;  // device copy of `cols` inside somebody else's upload (UploadBatch)
  const uint32_t* const* dcols() const { return d_cols_view ? d_cols_view : d_cols.as<const uint32_t*>(); }

  // sort the columns (stable, by size descending): fills `cols` / `col_logs`
  void prepare(const std::vector<const uint32_t*>& columns, const std::vector<uint32_t>& logs) {
    // a stable counting sort over the (at most 32) sizes: this runs on the host between two phases, with the GPU idle
    const size_t n = columns.size();
    uint32_t off[34] = {0};
    for (size_t i = 0; i < n; i++) { CM_CHECK(logs[i] < 32, "merkle tree: column of 2^32 rows or more"); off[32 - logs[i]]++; }
    for (uint32_t k = 0, run = 0; k < 34; k++) { const uint32_t c = off[k]; off[k] = run; run += c; }   // slot 32 - log: descending size
    cols.resize(n); col_logs.resize(n);
    for (size_t i = 0; i < n; i++) { const uint32_t p = off[32 - logs[i]]++; cols[p] = columns[i]; col_logs[p] = logs[i]; }
    d_cols_view = nullptr;
  }
  void commit(const std::vector<const uint32_t*>& columns, const std::vector<uint32_t>& logs, hipStream_t st) {
    prepare(columns, logs);
    d_cols = upload(cols, st);
    commit_prepared(st);
  }
  // One launch of the commitment: the layers 2^hi .. 2^lo it produces (it reads the columns of exactly those sizes and the
  // hashes of layer hi + 1).  `pace_before`: the tree's pace event is recorded in front of it.
  struct CommitLaunch { int hi, lo; bool pace_before; std::function<void(hipStream_t)> run; };
  // after prepare(); the device copy of `cols` is either d_cols or d_cols_view
  void commit_prepared(hipStream_t st) {
    for (auto& l : plan_commit()) run_launch(l, st);
  }
  void run_launch(const CommitLaunch& l, hipStream_t st) {
    if (l.pace_before && pace_ev) { CM_HIP(hipEventRecord(pace_ev, st)); pace_recorded = true; }
    l.run(st);
  }
  // Allocates every layer and returns the launches of the commitment in order (largest layer first).  The caller may interleave
  // them with the transforms that produce the columns (Prover::commit_enqueue: launch k only needs the columns of >= 2^lo rows).
  std::vector<CommitLaunch> plan_commit() {
    std::vector<CommitLaunch> plan;
    uint32_t max_log = cols.empty() ? 0 : col_logs[0];
    layers.clear();
    layers.resize(max_log + 1);
    size_t ci = 0;
    const int tail_top = (int)std::min<uint32_t>(max_log, MERKLE_TAIL_LOG);
    static const bool use_top = getenv("CM_NO_MERKLE_TOP") == nullptr;   // A/B switch
    pace_recorded = false;
    top_launch_has_extras = top_launch_has_fold = false;
    bool pace_planned = false;
    auto push = [&](int hi, int lo, std::function<void(hipStream_t)> f) {
      const bool pace = !pace_planned && hi <= (int)MERKLE_PACE_LOG;
      if (pace) pace_planned = true;
      plan.push_back(CommitLaunch{hi, lo, pace, std::move(f)});
    };
    for (int log = (int)max_log; log > tail_top;) {
      // the whole top of the tree in one launch once no wide layer is left among the per-lane levels
      if (use_top && log <= (int)MERKLE_TOP_MAX_LOG && log >= 9) {
        bool wide_inside = false;
        size_t cj = ci;
        for (int l = log; l >= log - 8; l--) {
          size_t cnt = 0;
          while (cj + cnt < cols.size() && col_logs[cj + cnt] == (uint32_t)l) cnt++;
          if (cnt >= MERKLE_QUAD_MIN_COLS) wide_inside = true;
          cj += cnt;
        }
        if (!wide_inside) {
          MerkleTopArgs a;
          a.top_log = (uint32_t)log;
          a.prev = (log < (int)max_log) ? layers[log + 1].u32() : nullptr;
          a.cols = dcols();
          for (int l = log; l >= 0; l--) {
            a.col_begin[l] = (uint32_t)ci;
            while (ci < cols.size() && col_logs[ci] == (uint32_t)l) ci++;
            a.col_end[l] = (uint32_t)ci;
            layers[l].alloc((size_t)32 << l);
            a.layers[l] = layers[l].u32();
          }
          a.x = top_extra;
          if (log != (int)max_log) a.x.fold_mode = 0;   // the fold makes the leaf level: only a launch that starts there may carry it
          top_launch_has_extras = a.x.chan != nullptr;
          top_launch_has_fold = a.x.fold_mode != 0;
          push(log, 0, [a](hipStream_t st) mutable { merkle_top(a, st); });
          return plan;
        }
      }
      // group of up to MERKLE_MULTI_LEVELS layers per launch (the top layer of a group needs >= 256 nodes)
      int levels = std::min<int>((int)MERKLE_MULTI_LEVELS, log - tail_top);
      if (log < 8) levels = 1;
      // big layers are throughput-bound: one node per thread with every lane busy beats the fused kernel
      // (whose parent levels run on half / quarter of the block); fusion pays only once launches are latency-bound
      if (log >= tune(T_MERKLE_MULTI_TOP)) levels = 1;   // (default MERKLE_MULTI_MAX_TOP = 19; A/B: "merkle_multi_top")
      // a mid-size layer carrying many columns is one long compression chain per node: quad-lane kernel
      size_t n_here = 0;
      while (ci + n_here < cols.size() && col_logs[ci + n_here] == (uint32_t)log) n_here++;
      const bool wide = log <= (int)MERKLE_QUAD_MAX_LOG && n_here >= MERKLE_QUAD_MIN_COLS;
      if (!wide && levels > 1) {  // a fused group must stop in front of a wide layer further down
        size_t cj = ci + n_here;
        for (int lv = 1; lv < levels; lv++) {
          size_t cnt = 0;
          while (cj + cnt < cols.size() && col_logs[cj + cnt] == (uint32_t)(log - lv)) cnt++;
          if (log - lv <= (int)MERKLE_QUAD_MAX_LOG && cnt >= MERKLE_QUAD_MIN_COLS) { levels = lv; break; }
          cj += cnt;
        }
      }
      if (levels == 1 || wide) {
        size_t c0 = ci;
        ci += n_here;
        if (log == (int)max_log && leaf_prealloc.p && leaf_prealloc.bytes >= ((size_t)32 << log)) layers[log] = std::move(leaf_prealloc);
        else layers[log].alloc((size_t)32 << log);
        const uint32_t* prev = (log < (int)max_log) ? layers[log + 1].u32() : nullptr;
        const uint32_t* const* dc = dcols() + c0;
        const uint32_t nc = (uint32_t)(ci - c0);
        uint32_t* outp = layers[log].u32();
        if (wide) push(log, log, [=](hipStream_t st) { merkle_layer_quad((uint32_t)log, prev, dc, nc, outp, st); });
        else push(log, log, [=](hipStream_t st) { merkle_layer((uint32_t)log, prev, dc, nc, outp, st); });
        log--;
        continue;
      }
      MerkleMultiArgs a;
      a.top_log = (uint32_t)log;
      a.n_levels = (uint32_t)levels;
      a.prev = (log < (int)max_log) ? layers[log + 1].u32() : nullptr;
      a.cols = dcols();
      double bytes = 0;
      for (int lv = 0; lv < levels; lv++) {
        int l = log - lv;
        a.col_begin[lv] = (uint32_t)ci;
        while (ci < cols.size() && col_logs[ci] == (uint32_t)l) ci++;
        a.col_end[lv] = (uint32_t)ci;
        layers[l].alloc((size_t)32 << l);
        a.layers[lv] = layers[l].u32();
        bytes += (4.0 * (a.col_end[lv] - a.col_begin[lv]) + 32.0 + ((lv == 0 && a.prev) ? 64.0 : 0.0)) * (double)((size_t)1 << l);
      }
      push(log, log - levels + 1, [a, bytes](hipStream_t st) { merkle_multi(a, bytes, st); });
      log -= levels;
    }
    {
      // layers 2^tail_top .. 2^0: one fused launch
      MerkleTailArgs a;
      a.top_log = (uint32_t)tail_top;
      a.prev = (tail_top < (int)max_log) ? layers[tail_top + 1].u32() : nullptr;
      a.cols = dcols();
      for (int log = tail_top; log >= 0; log--) {
        a.col_begin[log] = (uint32_t)ci;
        while (ci < cols.size() && col_logs[ci] == (uint32_t)log) ci++;
        a.col_end[log] = (uint32_t)ci;
        layers[log].alloc((size_t)32 << log);
        a.layers[log] = layers[log].u32();
      }
      push(tail_top, 0, [a](hipStream_t st) { merkle_tail(a, st); });
    }
    return plan;
  }
  void root(uint8_t out[32], hipStream_t st) const {
    uint32_t* pin = pinned_words() + PIN_ROOT;
    CM_HIP(hipMemcpyAsync(pin, layers[0].p, 32, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    memcpy(out, pin, 32);
  }

  // Symbolic decommitment walk (Stwo MerkleProver::decommit).  queries_per_log_size: log -> sorted unique positions.
  DecommitPlan plan_decommit(const std::map<uint32_t, std::vector<uint32_t>>& queries_per_log_size, GatherBatch& gb) const {
    const std::vector<uint32_t>* by_log[64] = {};
    for (auto& kv : queries_per_log_size) if (kv.first < 64) by_log[kv.first] = &kv.second;
    return plan_decommit(by_log, gb);
  }
  // one queried size only (FRI layer trees)
  DecommitPlan plan_decommit(uint32_t log, const std::vector<uint32_t>& positions, GatherBatch& gb) const {
    const std::vector<uint32_t>* by_log[64] = {};
    by_log[log] = &positions;
    return plan_decommit(by_log, gb);
  }
  DecommitPlan plan_decommit(const std::vector<uint32_t>* const (&by_log)[64], GatherBatch& gb) const {
    DecommitPlan plan;
    plan.hash0 = gb.hash_addrs.size();
    size_t ci = 0;
    size_t max_q = 0;
    for (auto* v : by_log) if (v) max_q = std::max(max_q, v->size());
    std::vector<uint32_t> last, total;   // scratch reused across the layers (no per-layer allocation)
    last.reserve(max_q);
    total.reserve(max_q);
    {
      size_t col_layers = 0;   // layers that carry columns (a FRI layer tree: one)
      for (size_t c = 0; c < col_logs.size(); c++) if (c == 0 || col_logs[c] != col_logs[c - 1]) col_layers++;
      plan.segs.reserve(max_q * col_layers);
    }
    for (int layer_log = (int)layers.size() - 1; layer_log >= 0; layer_log--) {
      size_t c0 = ci;
      while (ci < cols.size() && col_logs[ci] == (uint32_t)layer_log) ci++;
      bool has_prev = (size_t)layer_log + 1 < layers.size();
      static const std::vector<uint32_t> empty;
      const std::vector<uint32_t>& colq = (layer_log < 64 && by_log[layer_log]) ? *by_log[layer_log] : empty;
      total.clear();
      size_t pi = 0, qi = 0;
      while (pi < last.size() || qi < colq.size()) {
        uint32_t node;
        if (pi < last.size() && qi < colq.size()) node = std::min(last[pi] / 2, colq[qi]);
        else if (pi < last.size()) node = last[pi] / 2;
        else node = colq[qi];
        if (has_prev) {
          const uint32_t* child = layers[layer_log + 1].u32();
          if (pi < last.size() && last[pi] == 2 * node) pi++;
          else gb.add_hash(child + (size_t)(2 * node) * 8);
          if (pi < last.size() && last[pi] == 2 * node + 1) pi++;
          else gb.add_hash(child + (size_t)(2 * node + 1) * 8);
        }
        bool isq = qi < colq.size() && colq[qi] == node;
        if (isq) qi++;
        if (ci > c0) {
          if (dcols()) {
            size_t off = gb.add_run(dcols() + c0, (uint32_t)(ci - c0), node);
            plan.segs.push_back(DecommitPlan::Seg{1u, isq ? 1u : 0u, off, (uint32_t)(ci - c0)});
          } else {  // trees whose column table never went to the device (FRI tail layers: 4 columns)
            size_t off = gb.word_addrs.size();
            for (size_t c = c0; c < ci; c++) gb.add_word(cols[c] + node);
            plan.segs.push_back(DecommitPlan::Seg{0u, isq ? 1u : 0u, off, (uint32_t)(ci - c0)});
          }
        }
        total.push_back(node);
      }
      last.swap(total);
    }
    plan.n_hash = gb.hash_addrs.size() - plan.hash0;
    return plan;
  }
  static void finish_decommit(const DecommitPlan& plan, const GatherBatch& gb, std::vector<uint32_t>& queried_values,
                              MerkleDecommitment& d) {
    d.hash_witness.resize(plan.n_hash);
    if (plan.n_hash) memcpy(d.hash_witness[0].data(), &gb.hashes[8 * plan.hash0], 32 * plan.n_hash);   // Hash32 = 32 contiguous bytes
    size_t nq = 0, nwit = 0;
    for (const auto& sg : plan.segs) (sg.is_query ? nq : nwit) += sg.count;
    queried_values.reserve(queried_values.size() + nq);
    d.column_witness.reserve(d.column_witness.size() + nwit);
    for (const auto& sg : plan.segs) {
      const uint32_t* v = (sg.is_run ? gb.run_out : gb.words) + sg.off;
      std::vector<uint32_t>& dst = sg.is_query ? queried_values : d.column_witness;
      dst.insert(dst.end(), v, v + sg.count);
    }
  }
  // single-tree convenience (per-op C ABI)
  void decommit(const std::map<uint32_t, std::vector<uint32_t>>& queries_per_log_size, std::vector<uint32_t>& queried_values,
                MerkleDecommitment& d, hipStream_t st) const {
    GatherBatch gb;
    DecommitPlan plan = plan_decommit(queries_per_log_size, gb);
    gb.run(st);
    finish_decommit(plan, gb, queried_values, d);
  }
};

}  // namespace cm
