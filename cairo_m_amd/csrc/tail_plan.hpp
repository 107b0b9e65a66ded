// Host side of the device-side proof tail (tail_device.hpp): which pieces the decommitment of a proof consists of (built
// BEFORE the queries exist — only list ids, layer bases and column tables), the launches, and the distribution of what came
// back into the proof object.  Used by the single-GPU segment prover (prover.hip).
#pragma once
#include "prover_common.hpp"
#include "tail_device.hpp"

namespace cm {

inline std::atomic<int> g_device_tail{-1};   // cm_set_device_tail: -1 = CM_DEVICE_TAIL / default on, 0 = host walk, 1 = device
inline bool device_tail_enabled() {
  const int v = g_device_tail.load(std::memory_order_relaxed);
  if (v >= 0) return v != 0;
  static const bool env_on = !(getenv("CM_DEVICE_TAIL") && atoi(getenv("CM_DEVICE_TAIL")) == 0);
  return env_on;
}

struct DeviceTail {
  enum Target : int { T_TREE_HASH = 0, T_FIRST_WITNESS, T_FIRST_HASH, T_INNER_WITNESS, T_INNER_HASH, T_TREE_ROWS };
  struct Span { Target what; uint32_t index; size_t d0, d1; };
  std::vector<TailDesc> desc;
  std::vector<Span> spans;
  uint32_t L0 = 0, nq = 0, nq_pad = 0, qmask = 0, n_small = 0;
  DevBuf d_desc, d_off, d_tab, d_nonce;
  uint32_t *hdr = nullptr, *h_pos = nullptr, *h_out = nullptr;
  size_t out_bound = 0;
  bool enqueued = false;
  // the host follows the tail in three steps instead of one wait at its end: behind k_tail_last it replays the commit phase
  // (FriPhase::commit_finish) while the proof of work runs, behind k_tail_tables it checks the nonce, draws the queries and builds
  // its tables while the gathers run, and only the witness copy waits for the last kernel
  hipEvent_t ev_last = nullptr, ev_tables = nullptr;
  std::vector<size_t> off;   // first output word of every piece, from the host's tables (plan())

  static bool supported(const cm_pcs_config& cfg, const std::vector<uint32_t>& q_logs, const FriPhase& fri) {
    if (!device_tail_enabled() || q_logs.empty() || !fri.have_first) return false;
    const uint32_t last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup_factor;
    return cfg.n_queries >= 1 && cfg.n_queries <= TAIL_MAX_QUERIES && q_logs[0] < TAIL_MAX_SHIFTS && q_logs[0] >= 1 &&
           (1u << last_log) <= TAIL_MAX_LAST && 4u << last_log <= PIN_WORDS - PIN_LAST_LAYER && cfg.pow_bits <= TAIL_MAX_POW_BITS &&
           (fri.n_inner_ + 1) * 12 <= PIN_LAST_LAYER - PIN_ALPHAS;
  }

  void begin(Target what, uint32_t index) { spans.push_back(Span{what, index, desc.size(), desc.size()}); }
  void add(uint32_t kind, uint32_t k, uint32_t width, const void* p0, const void* p1 = nullptr, const void* p2 = nullptr, const void* p3 = nullptr) {
    desc.push_back(TailDesc{{p0, p1, p2, p3}, kind, k, width, 0});
    spans.back().d1 = desc.size();
  }
  static size_t bound_items(uint32_t kind, uint32_t k, uint32_t L0, uint32_t nq) {
    const uint64_t layer = (uint64_t)1 << (L0 - k);
    const size_t n = (size_t)std::min<uint64_t>(nq, layer);
    return kind == TD_HASH_F ? 3 * n : n;
  }

  // everything behind the last FRI fold, enqueued on the prover's stream (FriPhase::commit_enqueue has run)
  void enqueue(Prover& P, FriPhase& fri, const std::vector<ColumnSet>& quotients, const std::vector<uint32_t>& q_logs) {
    hipStream_t st = P.st;
    const cm_pcs_config& cfg = P.cfg;
    L0 = q_logs[0];
    nq = cfg.n_queries;
    nq_pad = 64;
    while (nq_pad < nq) nq_pad <<= 1;
    qmask = 0;
    for (auto l : q_logs) qmask |= 1u << l;
    desc.clear(); spans.clear();
    desc.reserve(40 * (6 + fri.inner.size()));
    // ---- hash and witness pieces
    for (uint32_t t = 0; t < 4; t++) {
      const MerkleTree& mt = P.trees[t].merkle;
      begin(T_TREE_HASH, t);
      for (int j = (int)mt.layers.size() - 1; j >= 1; j--) add(TD_HASH_W, L0 - (uint32_t)j, 8, mt.layers[j].p);
    }
    begin(T_FIRST_WITNESS, 0);
    for (size_t g = 0; g < quotients.size(); g++)
      add(TD_COORDS_W, L0 - q_logs[g], 4, quotients[g].ptrs[0], quotients[g].ptrs[1], quotients[g].ptrs[2], quotients[g].ptrs[3]);
    begin(T_FIRST_HASH, 0);
    CM_CHECK(fri.first_tree.layers.size() == (size_t)L0 + 1, "device tail: first FRI tree does not span the query domain");
    for (int j = (int)L0; j >= 1; j--) add(TD_HASH_F, L0 - (uint32_t)j + 1, 8, fri.first_tree.layers[j].p);
    for (size_t i = 0; i < fri.inner.size(); i++) {
      const FriPhase::InnerLayer& il = *fri.inner[i];
      CM_CHECK(il.log + fri.inner_fold0 + i == L0, "device tail: FRI layer sizes do not follow the query folds");
      begin(T_INNER_WITNESS, (uint32_t)i);
      add(TD_COORDS_W, L0 - il.log, 4, il.eval.ptrs[0], il.eval.ptrs[1], il.eval.ptrs[2], il.eval.ptrs[3]);
      begin(T_INNER_HASH, (uint32_t)i);
      for (int j = (int)il.log - 1; j >= 1; j--) add(TD_HASH_W, L0 - (uint32_t)j, 8, il.tree.layers[j].p);
    }
    n_small = (uint32_t)desc.size();
    // ---- queried rows of the four commitment trees: one piece per column-bearing layer, largest first
    for (uint32_t t = 0; t < 4; t++) {
      const MerkleTree& mt = P.trees[t].merkle;
      begin(T_TREE_ROWS, t);
      for (size_t c0 = 0; c0 < mt.cols.size();) {
        size_t c1 = c0;
        while (c1 < mt.cols.size() && mt.col_logs[c1] == mt.col_logs[c0]) c1++;
        CM_CHECK(mt.col_logs[c0] <= L0 && ((qmask >> mt.col_logs[c0]) & 1u), "device tail: a committed column size has no query set");
        add(TD_ROWS_U, L0 - mt.col_logs[c0], (uint32_t)(c1 - c0), mt.dcols() + c0);
        c0 = c1;
      }
    }
    out_bound = 0;
    for (auto& d : desc) out_bound += bound_items(d.kind, d.k, L0, nq) * TailTables::words_per_item(d);
    // pinned: descriptors in; {header | positions | witnesses} out
    TailDesc* h_desc = (TailDesc*)tail_pinned_desc(desc.size() * sizeof(TailDesc));
    memcpy(h_desc, desc.data(), desc.size() * sizeof(TailDesc));
    uint32_t* ob = (uint32_t*)tail_pinned_out((TAIL_HDR_WORDS + (size_t)nq_pad + out_bound) * 4);
    hdr = ob; h_pos = ob + TAIL_HDR_WORDS; h_out = h_pos + nq_pad;
    memset(hdr, 0xFF, TAIL_HDR_WORDS * 4);
    d_desc.alloc(desc.size() * sizeof(TailDesc));
    d_off.alloc(desc.size() * 4);
    d_tab.alloc(tail_tab_words(nq_pad) * 4);
    d_nonce.alloc(8);
    // K1: last layer
    {
      TailLastArgs a;
      memset(&a, 0, sizeof(a));
      a.d_ar = fri.d_ar.u32();
      a.n_ar_words = (fri.n_inner_ + 1) * 12;
      const uint32_t last_log = fri.last_log_, n = 1u << last_log;
      for (int c = 0; c < 4; c++) a.last[c] = fri.last_layer.ptrs[c];
      a.log_n = last_log;
      a.log_keep = cfg.log_last_layer_degree_bound;
      a.ninv = inv(M31::from_u32(n)).v;
      for (uint32_t l = 0; l < last_log; l++)
        for (uint32_t h = 0; h < (n >> (l + 1)); h++) {
          const uint32_t clog_ = last_log - l;
          const uint32_t idx = subgroup_gen_index(clog_ + 2) + subgroup_gen_index(clog_) * bit_reverse(h, clog_ - 1);
          a.xinv[(n - (n >> l)) + h] = inv(point_at_index(idx).x).v;
        }
      a.chan = fri.d_chan_;
      a.h_ar = pinned_words() + PIN_ALPHAS;
      a.h_last = pinned_words() + PIN_LAST_LAYER;
      a.hdr = hdr;
      a.nonce = (unsigned long long*)d_nonce.p;
      tail_last_layer(a, st);
    }
    {
      static thread_local hipEvent_t evs[2] = {nullptr, nullptr};
      for (auto& e : evs)
        if (!e) { CM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); thread_event_owned(e); }
      ev_last = evs[0]; ev_tables = evs[1];
    }
    flags = tune(T_TAIL_FLAGS) != 0;   // the host watches two header words instead of two events (A/B: cm_set_tuning("tail_flags", 0))
    if (!flags) CM_HIP(hipEventRecord(ev_last, st));
    P.tick("fri_commit");
    tail_grind(fri.d_chan_, cfg.pow_bits, (unsigned long long*)d_nonce.p, st);
    P.tick("pow");
    {
      TailTablesArgs a;
      a.chan = fri.d_chan_;
      a.nonce = (const unsigned long long*)d_nonce.p;
      a.n_queries = nq; a.log_domain = L0; a.qmask = qmask;
      a.n_desc = (uint32_t)desc.size();
      a.h_desc = h_desc; a.d_desc = d_desc.as<TailDesc>(); a.d_off = d_off.u32();
      a.tab = d_tab.u32(); a.nq_pad = nq_pad; a.hdr = hdr; a.h_positions = h_pos;
      tail_tables(a, st);
    }
    if (!flags) CM_HIP(hipEventRecord(ev_tables, st));
    tail_gather(d_desc.as<TailDesc>(), d_off.u32(), d_tab.u32(), nq_pad, n_small, (uint32_t)desc.size(), nq, h_out, st);
    P.tick("decommit");
    enqueued = true;
  }

  bool flags = true;
  // the host's two waits: a header word the kernel sets behind its pinned writes, or (flags off / the word does not come within
  // ~2 ms of spinning) the event / the stream
  void wait_word(uint32_t word, hipEvent_t ev, hipStream_t st) const {
    if (flags) {
      const volatile uint32_t* w = hdr + word;
      for (int spin = 0; spin < 200000; spin++) {
        if (*w == 1u) { std::atomic_thread_fence(std::memory_order_acquire); return; }
        __builtin_ia32_pause();
      }
      CM_HIP(hipStreamSynchronize(st));
      return;
    }
    CM_HIP(hipEventSynchronize(ev));
  }
  void wait_last(hipStream_t st) const {
    wait_word(TAIL_HDR_LAST_DONE, ev_last, st);
    // k_tail_last found a non-zero coefficient above the degree bound of the last layer (the host replay checks it again)
    CM_CHECK(hdr[6] == 0, "fri: the last layer's degree exceeds the bound (device tail)");
  }
  void wait_tables(hipStream_t st) const { wait_word(TAIL_HDR_TABLES_DONE, ev_tables, st); }
  // behind the tables: the header and the positions are in pinned memory
  bool nonce_found() const { return hdr[0] == TAIL_OK; }
  uint64_t nonce() const { return (uint64_t)hdr[1] | ((uint64_t)hdr[2] << 32); }

  // `queries`: drawn by the HOST channel after mix_u64(nonce) — must equal what the device drew.  Host tables -> where every piece
  // of the witness stream starts (while k_tail_gather is still writing it).
  void plan(const Queries& queries) {
    CM_CHECK(hdr[3] == queries.positions.size() && memcmp(h_pos, queries.positions.data(), 4 * queries.positions.size()) == 0,
             "decommit: device query positions diverged from the host channel");
    TailTables tt;
    tt.build(queries.positions, L0, qmask);
    off.assign(desc.size() + 1, 0);
    for (size_t d = 0; d < desc.size(); d++) off[d + 1] = off[d] + tt.count(desc[d].kind, desc[d].k) * TailTables::words_per_item(desc[d]);
    CM_CHECK(off.back() == hdr[4] && off.back() <= out_bound, "decommit: device witness size differs from the host's table walk");
  }
  // after the stream is synchronised: the witnesses into the proof object
  void copy(Prover& P, FriPhase& fri, ProofData& pf) {
    pf.decommitments.resize(4);
    pf.queried_values.resize(4);
    pf.fri_inner.resize(fri.inner.size());
    // one pass per vector: bulk assign from the pinned words (no zero-fill first, no element-wise push_back)
    static_assert(sizeof(Hash32) == 32 && sizeof(QM31) == 16, "witness words map onto the proof's element types");
    auto hashes = [&](std::vector<Hash32>& v, const uint32_t* w, size_t words) {
      const Hash32* p = reinterpret_cast<const Hash32*>(w);
      v.assign(p, p + words / 8);
    };
    auto felts = [&](std::vector<QM31>& v, const uint32_t* w, size_t words) {
      const QM31* p = reinterpret_cast<const QM31*>(w);   // QM31 = four canonical M31 words (QM31::from_u32 copies them)
      v.assign(p, p + words / 4);
    };
    for (const Span& sp : spans) {
      const uint32_t* w = h_out + off[sp.d0];
      const size_t words = off[sp.d1] - off[sp.d0];
      switch (sp.what) {
        case T_TREE_HASH: hashes(pf.decommitments[sp.index].hash_witness, w, words); break;
        case T_TREE_ROWS: pf.queried_values[sp.index].assign(w, w + words); break;
        case T_FIRST_WITNESS: felts(pf.fri_first.fri_witness, w, words); break;
        case T_FIRST_HASH: hashes(pf.fri_first.decommitment.hash_witness, w, words); break;
        case T_INNER_WITNESS: felts(pf.fri_inner[sp.index].fri_witness, w, words); break;
        case T_INNER_HASH: hashes(pf.fri_inner[sp.index].decommitment.hash_witness, w, words); break;
      }
    }
    for (size_t i = 0; i < fri.inner.size(); i++) pf.fri_inner[i].commitment = fri.inner[i]->root;
    for (int t = 0; t < 4; t++) pf.commitments.push_back(P.trees[t].root);
  }
};

}  // namespace cm
