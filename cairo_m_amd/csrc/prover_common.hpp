// Shared pieces of the segment provers (single-GPU prover.hip, sharded prover_sharded.inc) and of the FRI phase
// (fri_phase.hip): column arenas, committed trees, the per-proof driver state `Prover` (stream, channel, phase events,
// commitment of one tree), query sets and the FRI phase's interface.
#pragma once
#include "../../include/cairom_hip.h"
#include "engine.hpp"
#include "merkle_tree.hpp"
#include "fri_kernels.hpp"
#include "host_channel.hpp"
#include "framing.hpp"
#include "proof.hpp"
#include "kprof.hpp"
#include <atomic>
#include <array>
#include <chrono>
#include <memory>
#include <map>
#include <algorithm>
#include <functional>

namespace cm {

using hostch::Channel;

// ---- column sets -------------------------------------------------------------------------------------------
constexpr size_t COL_SKEW_WORDS_DEFAULT = 0;   // A/B: CM_COL_SKEW_BYTES
struct ColumnSet {
  std::vector<uint32_t> logs;
  std::vector<uint32_t*> ptrs;
  DevBuf buf, d_ptrs;
  uint32_t** d_view = nullptr;  // device pointer table living in somebody else's upload (UploadBatch)
  // Column skew: the columns of a set are powers of two long, so without padding row r of EVERY column has the same address
  // modulo the column size — a kernel that reads one row of many columns (Merkle leaves, DEEP quotients, constraints, LogUp:
  // every lane-coalesced 256-byte run of a wave) then keeps hitting the same HBM channel / bank group.  Large columns are
  // therefore laid out `skew_words()` apart in addition to their length (a multiple of 64 words: runs stay 256-byte aligned).
  static size_t skew_words() {
    static const size_t w = getenv("CM_COL_SKEW_BYTES") ? (size_t)atol(getenv("CM_COL_SKEW_BYTES")) / 4 : COL_SKEW_WORDS_DEFAULT;
    return w & ~(size_t)63;
  }
  void alloc(const std::vector<uint32_t>& logs_, hipStream_t st, bool upload_ptrs = true, bool contiguous = false) {
    logs = logs_;
    const size_t skew = contiguous ? 0 : skew_words();
    size_t total = 0;
    for (auto l : logs) total += ((size_t)1 << l) + (l >= 14 ? skew : 0);
    buf.alloc(total * 4);
    ptrs.resize(logs.size());
    size_t off = 0;
    for (size_t i = 0; i < logs.size(); i++) { ptrs[i] = buf.u32() + off; off += ((size_t)1 << logs[i]) + (logs[i] >= 14 ? skew : 0); }
    d_view = nullptr;
    if (upload_ptrs) d_ptrs = upload(ptrs, st);
  }
  uint32_t* const* dev(size_t first = 0) const {
    if (d_view) return d_view + first;
    CM_CHECK(d_ptrs.p, "ColumnSet::dev(): pointer table was not uploaded");
    return d_ptrs.as<uint32_t*>() + first;
  }
  size_t size() const { return logs.size(); }
};

// groups column indices by log size (descending) — used to batch FFT launches
inline std::map<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> by_log(const std::vector<uint32_t>& logs) {
  std::map<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> m;
  for (uint32_t i = 0; i < logs.size(); i++) m[logs[i]].push_back(i);
  return m;
}

struct CommittedTree {
  ColumnSet coeffs, lde;
  MerkleTree merkle;
  hostch::Hash32 root;
  DevBuf tables;  // one upload: pointer tables of coeffs / lde / the FFT groups / the Merkle column order
};

constexpr int PIPE_PRIO_DEFAULT = 0;          // commitment pipeline: priority class of the transform stream (A/B: CM_PIPE_PRIO)
constexpr uint32_t FFT_CHUNK_MB_DEFAULT = 0;   // Infinity-Cache blocking of the transform sweeps (commit_enqueue); A/B: CM_FFT_CHUNK_MB
inline std::atomic<int> g_transcript_log{0};     // cm_set_transcript_log: proofs record every Fiat-Shamir step (ProofData::transcript)
inline std::atomic<int> g_proofs_in_flight{0};   // proofs being made by cm_prove_many runners right now (0 outside of it)
inline void thread_event_owned(hipEvent_t e) { at_thread_exit([e] { (void)hipEventDestroy(e); }); }   // destroyed when the creating thread ends
struct Prover {
  FramingUse framing_use;   // the process-wide framing cannot change while this proof is being made (framing.hpp)
  hipStream_t st = 0;
  cm_pcs_config cfg;
  Twiddles* tw = nullptr;
  Channel ch;
  CommittedTree trees[4];
  std::vector<double> phase_ms;
  std::chrono::steady_clock::time_point t0;

  // Phase boundaries are HIP events on the prover stream, read back at the end of the proof: a host-side
  // hipStreamSynchronize per phase drained the GPU at boundaries that need no host round trip (constraints ->
  // composition commit, quotients -> FRI).  CM_HOST_TRACE=1 restores the synchronising form and prints host / wait times.
  std::vector<hipEvent_t> evs;
  static std::vector<hipEvent_t>& event_cache() {
    static thread_local std::vector<hipEvent_t>* c = nullptr;
    if (!c) {
      c = new std::vector<hipEvent_t>();
      std::vector<hipEvent_t>* own = c;
      at_thread_exit([own] { for (hipEvent_t e : *own) (void)hipEventDestroy(e); delete own; });
    }
    return *c;
  }
  hipEvent_t next_event() {
    auto& c = event_cache();
    if (evs.size() == c.size()) { hipEvent_t e; CM_HIP(hipEventCreate(&e)); c.push_back(e); }
    evs.push_back(c[evs.size()]);
    return evs.back();
  }
  void start() {
    t0 = std::chrono::steady_clock::now();
    CM_HIP(hipEventRecord(next_event(), st));
  }
  void tick(const char* name) {
    static const bool trace = getenv("CM_HOST_TRACE") != nullptr;
    CM_HIP(hipEventRecord(next_event(), st));
    if (!trace) return;
    auto te = std::chrono::steady_clock::now();
    CM_HIP(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[phase] %-20s host %8.1f us, then waited %8.1f us for the GPU\n", name,
            std::chrono::duration<double, std::micro>(te - t0).count(), std::chrono::duration<double, std::micro>(t1 - te).count());
    t0 = t1;
  }
  void finish() {   // the stream is idle (the decommitment gather has been read back)
    CM_HIP(hipEventSynchronize(evs.back()));
    for (size_t k = 1; k < evs.size(); k++) {
      float ms = 0;
      CM_HIP(hipEventElapsedTime(&ms, evs[k - 1], evs[k]));
      phase_ms.push_back(ms);
    }
  }

  // IFFT src(evals, trace domain) -> tree.coeffs; LDE -> tree.lde; Merkle; mix root.
  // If `in_place`, coeffs aliases src (src is consumed).
  void commit(CommittedTree& t, ColumnSet* evals, bool from_coeffs) {
    commit_enqueue(t, evals, from_coeffs, st);
    commit_finish(t);
  }
  // the root comes back on the prover stream (which has joined the stream the tree was built on) and goes into the transcript
  void commit_finish(CommittedTree& t) {
    t.merkle.root(t.root.data(), st);
    ch.mix_root(t.root);
  }
  // everything of a commitment except the root read-back, on stream `s` (tree 0 is built on a side stream while the
  // execution trace is generated: both are chains of small launches)
  // evals_in_place: with from_coeffs, the columns of t.coeffs still hold EVALUATIONS: every size group is interpolated in place
  // right in front of its extension (the small columns by the fused small-column kernel)
  // s_tr: SOFTWARE PIPELINE of the commitment (prover.rs:71-73, 80-82, 100-102: `extend_evals` + `commit`).  The transforms run on
  // `s_tr`, size group by size group from the largest down, the Merkle launches on `s`: the hashing of layer 2^L starts as soon as
  // the groups of >= 2^L rows are extended, while the next group is still being transformed.  The Blake2s layers are bound by VALU
  // issue, the transform passes wait on HBM / LDS for ~40 % of their cycles (profiles/r03h_pmc_sq.json) — back to back on one queue
  // they never met.  `s_tr` must already be ordered behind everything the transforms read (a Fork side stream of `s`); when this
  // returns `s` is ordered behind all of `s_tr`'s work.  null = everything on `s` (CM_COMMIT_PIPE=0 forces that form: A/B).
  // Columns whose evaluations are still being finished on another stream when the tree is enqueued (tree 2: the four running-sum
  // columns of every component while the LogUp tail runs): transformed last within their size group, behind `ready`.
  struct DeferredCols { std::vector<char> late; hipEvent_t ready = nullptr; };
  // commit_enqueue = commit_prepare (host-only work + ONE host->device copy of every table of the tree, enqueued on `s`) followed
  // by commit_launch (the transforms and Merkle launches).  Callers that have GPU work in flight call commit_prepare EARLY — the
  // interaction phase right behind its LogUp launches: the ~0.1 ms the host spends on the 1036 columns of tree 2 (size groups,
  // pointer tables, layer buffers, the launch plan) then runs while the LogUp kernels execute instead of between the LogUp tail
  // and the first transform (round-5 timeline: 50 us of idle GPU there).
  struct Grp { uint32_t log, n; size_t off; uint32_t n_early; };
  // (round 6) GUEST columns: the columns of ANOTHER tree (the preprocessed tree next to the execution trace) ride in this tree's
  // size-group launches — one more column in the y-dimension of launches that are there anyway, instead of a chain of ~12 small
  // transform launches of its own on a side stream (a 2^18-row group of four columns is 64 blocks: a quarter of the chip for
  // 0.16 ms at high priority, inside the throughput-bound part of tree 1's commitment).  The guest's Merkle tree stays its own
  // (`stream`), launch by launch behind the size groups that carry its columns.
  struct GuestTree { CommittedTree* t; ColumnSet* evals; hipStream_t stream; };
  struct CommitPrep {
    CommittedTree* t = nullptr;
    bool from_coeffs = false, with_merkle = true, evals_in_place = false;
    hipStream_t s = nullptr, s_tr = nullptr;
    const DeferredCols* defer = nullptr;
    std::vector<uint32_t> logs;
    std::vector<Grp> grps;
    const uint32_t** d_table = nullptr;
    std::vector<SmallCommitJob> sjobs;
    SmallCommitJob* d_sjobs = nullptr;
    uint32_t small_max = 0;
    std::vector<MerkleTree::CommitLaunch> plan;
    GuestTree guest{nullptr, nullptr, nullptr};
    std::vector<MerkleTree::CommitLaunch> guest_plan;
  };
  void commit_enqueue(CommittedTree& t, ColumnSet* evals, bool from_coeffs, hipStream_t s, bool with_merkle = true,
                      bool evals_in_place = false, hipStream_t s_tr = nullptr, const DeferredCols* defer = nullptr) {
    CommitPrep cp = commit_prepare(t, evals, from_coeffs, s, with_merkle, evals_in_place, s_tr, defer, nullptr);
    commit_launch(cp);
  }
  // `defer->late` must be final; `defer->ready` may still be null (it is read by commit_launch)
  CommitPrep commit_prepare(CommittedTree& t, ColumnSet* evals, bool from_coeffs, hipStream_t s, bool with_merkle = true,
                            bool evals_in_place = false, hipStream_t s_tr = nullptr, const DeferredCols* defer = nullptr,
                            const GuestTree* guest = nullptr) {
    CM_CHECK(!guest || (!from_coeffs && !defer && guest->t && guest->evals), "commit: guest columns ride with a tree committed from evaluations");
    const bool pipe_on = tune(T_COMMIT_PIPE) != 0;
    if (!pipe_on || !with_merkle || s_tr == s) s_tr = nullptr;
    CommitPrep cp;
    cp.t = &t; cp.from_coeffs = from_coeffs; cp.with_merkle = with_merkle; cp.evals_in_place = evals_in_place; cp.s = s; cp.defer = defer;
    cp.logs = from_coeffs ? t.coeffs.logs : evals->logs;
    const std::vector<uint32_t>& logs = cp.logs;
    // (the host side of this function sits between two phases with the GPU idle: ~150 us for the 474 / 1036 columns of trees 1 / 2
    // before the size groups were built once instead of three times and the small columns' 1 / n came from a table)
    auto groups = by_log(logs);
    if (s_tr) {   // a tree of ONE large size group (composition, preprocessed-like trees) has nothing to overlap: two cross-queue
                  // hand-overs (~20 us each) for nothing — composition_commit 0.77 -> 0.81 ms when it went through the pipeline
      uint32_t big_groups = 0;
      for (auto& kv : groups) if (!small_commit_serves(kv.first, cfg.log_blowup_factor)) big_groups++;
      if (big_groups < 2) s_tr = nullptr;
    }
    cp.s_tr = s_tr;
    UploadBatch ub;
    if (!from_coeffs) { t.coeffs.alloc(logs, s, false); ub.add(t.coeffs.ptrs, &t.coeffs.d_view); }
    std::vector<uint32_t> lde_logs(logs);
    for (auto& l : lde_logs) l += cfg.log_blowup_factor;
    t.lde.alloc(lde_logs, s, false);
    ub.add(t.lde.ptrs, &t.lde.d_view);
    std::map<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> ggroups;   // guest columns by log
    if (guest) {
      cp.guest = *guest;
      CommittedTree& gt = *guest->t;
      const std::vector<uint32_t>& gl = guest->evals->logs;
      gt.coeffs.alloc(gl, s, false);
      ub.add(gt.coeffs.ptrs, &gt.coeffs.d_view);
      std::vector<uint32_t> glde(gl);
      for (auto& l : glde) l += cfg.log_blowup_factor;
      gt.lde.alloc(glde, s, false);
      ub.add(gt.lde.ptrs, &gt.lde.d_view);
      ggroups = by_log(gl);
      for (auto& kv : ggroups) groups[kv.first];   // a size only the guest has becomes a group of its own
      std::vector<const uint32_t*> gcols(gt.lde.ptrs.begin(), gt.lde.ptrs.end());
      gt.merkle.prepare(gcols, gt.lde.logs);
      ub.add(gt.merkle.cols, &gt.merkle.d_cols_view);
    }
    // pointer table of all size groups of the tree: [src | coeffs | lde] per group
    std::vector<Grp>& grps = cp.grps;
    std::vector<const uint32_t*> table;
    table.reserve(3 * logs.size());
    for (auto& kv : groups) {
      const std::vector<uint32_t>* gi = nullptr;   // the guest's columns of this size
      if (guest) { auto it = ggroups.find(kv.first); if (it != ggroups.end()) gi = &it->second; }
      const uint32_t n_all = (uint32_t)(kv.second.size() + (gi ? gi->size() : 0));
      Grp g{kv.first, n_all, table.size(), n_all};
      if (defer)
        g.n_early = (uint32_t)(std::stable_partition(kv.second.begin(), kv.second.end(), [&](size_t i) { return !defer->late[i]; }) -
                               kv.second.begin());
      for (auto i : kv.second) table.push_back(from_coeffs ? nullptr : evals->ptrs[i]);
      if (gi) for (auto i : *gi) table.push_back(guest->evals->ptrs[i]);
      for (auto i : kv.second) table.push_back(t.coeffs.ptrs[i]);
      if (gi) for (auto i : *gi) table.push_back(guest->t->coeffs.ptrs[i]);
      for (auto i : kv.second) table.push_back(t.lde.ptrs[i]);
      if (gi) for (auto i : *gi) table.push_back(guest->t->lde.ptrs[i]);
      grps.push_back(g);
    }
    ub.add(table, &cp.d_table);
    std::vector<const uint32_t*> cols(t.lde.ptrs.begin(), t.lde.ptrs.end());
    if (with_merkle) {        // (the sharded prover hashes row slices of the LDE instead: prover_sharded.inc)
      t.merkle.prepare(cols, t.lde.logs);
      ub.add(t.merkle.cols, &t.merkle.d_cols_view);
    }
    // small columns of every size: ONE fused interpolate + extend launch for all of them (k_small_commit)
    std::vector<SmallCommitJob>& sjobs = cp.sjobs;
    sjobs.reserve(logs.size());
    static const std::array<uint32_t, 31> inv_pow2 = [] {   // 1 / 2^k in M31 (a host inversion per small column was ~50 us per tree)
      std::array<uint32_t, 31> a{};
      for (uint32_t k = 0; k < 31; k++) a[k] = inv(M31::from_u32(1u << k)).v;
      return a;
    }();
    for (size_t i = 0; i < logs.size(); i++)
      if (small_commit_serves(logs[i], cfg.log_blowup_factor)) {
        const uint32_t* src = !from_coeffs ? evals->ptrs[i] : evals_in_place ? t.coeffs.ptrs[i] : nullptr;
        sjobs.push_back(SmallCommitJob{src, t.coeffs.ptrs[i], t.lde.ptrs[i], logs[i], inv_pow2[logs[i]]});
        cp.small_max = std::max(cp.small_max, logs[i]);
      }
    if (guest)
      for (size_t i = 0; i < guest->evals->logs.size(); i++) {
        const uint32_t l = guest->evals->logs[i];
        if (!small_commit_serves(l, cfg.log_blowup_factor)) continue;
        sjobs.push_back(SmallCommitJob{guest->evals->ptrs[i], guest->t->coeffs.ptrs[i], guest->t->lde.ptrs[i], l, inv_pow2[l]});
        cp.small_max = std::max(cp.small_max, l);
      }
    if (!sjobs.empty()) ub.add(sjobs, &cp.d_sjobs);
    t.tables = ub.flush(s);   // ONE host->device copy for the whole tree
    if (with_merkle && s_tr) cp.plan = t.merkle.plan_commit();
    if (guest) cp.guest_plan = guest->t->merkle.plan_commit();   // (behind the flush: the plan captures the device column table)
    return cp;
  }
  void commit_launch(CommitPrep& cp) {
    CommittedTree& t = *cp.t;
    const bool from_coeffs = cp.from_coeffs, with_merkle = cp.with_merkle, evals_in_place = cp.evals_in_place;
    hipStream_t s = cp.s, s_tr = cp.s_tr;
    const DeferredCols* defer = cp.defer;
    const std::vector<uint32_t>& logs = cp.logs;
    std::vector<Grp>& grps = cp.grps;
    const uint32_t** d_table = cp.d_table;
    std::vector<SmallCommitJob>& sjobs = cp.sjobs;
    SmallCommitJob* d_sjobs = cp.d_sjobs;
    const uint32_t small_max = cp.small_max;
    hipStream_t ts = s;       // the stream of the transforms
    if (s_tr) {               // the tables (and whatever else sits in front of this tree on `s`) first
      hipEvent_t e = pipe_event();
      CM_HIP(hipEventRecord(e, s));
      CM_HIP(hipStreamWaitEvent(s_tr, e, 0));
      ts = s_tr;
    }
    std::vector<MerkleTree::CommitLaunch>& plan = cp.plan;
    size_t next_launch = 0;
    // the Merkle launches whose columns all have at least 2^ready_log rows: behind what `ts` holds right now
    auto hash_ready = [&](int ready_log) {
      if (!s_tr) return;
      bool waited = false;
      while (next_launch < plan.size() && plan[next_launch].lo >= ready_log) {
        if (!waited) {
          hipEvent_t e = pipe_event();
          CM_HIP(hipEventRecord(e, ts));
          CM_HIP(hipStreamWaitEvent(s, e, 0));
          waited = true;
        }
        t.merkle.run_launch(plan[next_launch++], s);
      }
    };
    // the guest tree's Merkle launches whose columns all have at least 2^ready_log rows, on the guest's stream behind `ts`
    size_t guest_next = 0;
    auto guest_ready = [&](int ready_log) {
      if (!cp.guest.t) return;
      bool waited = false;
      while (guest_next < cp.guest_plan.size() && cp.guest_plan[guest_next].lo >= ready_log) {
        if (!waited) {
          hipEvent_t e = pipe_event();
          CM_HIP(hipEventRecord(e, ts));
          CM_HIP(hipStreamWaitEvent(cp.guest.stream, e, 0));
          waited = true;
        }
        cp.guest.t->merkle.run_launch(cp.guest_plan[guest_next++], cp.guest.stream);
      }
    };
    // the small columns (and the late ones of every group) need `defer->ready`: not in front of the first large group's early
    // columns, which is the work that hides the wait
    bool small_done = false;
    uint64_t total_cells = 0, early_cells = 0;
    for (auto l : logs) total_cells += (uint64_t)1 << l;
    auto late_ready = [&]() {
      if (small_done) return;
      small_done = true;
      if (defer) CM_HIP(hipStreamWaitEvent(ts, defer->ready, 0));
      small_commit(d_sjobs, (uint32_t)sjobs.size(), small_max, cfg.log_blowup_factor, *tw, ts);
    };
    if (!defer) late_ready();
    for (size_t gi = 0; gi < grps.size(); gi++) {
      const Grp& g = grps[gi];
      if (small_commit_serves(g.log, cfg.log_blowup_factor)) continue;
      const uint32_t* const* dsrc = d_table + g.off;
      uint32_t* const* dco = (uint32_t* const*)(d_table + g.off + g.n);
      uint32_t* const* dld = (uint32_t* const*)(d_table + g.off + 2 * g.n);
      // Infinity-Cache blocking: the four sweeps of a column (IFFT 2 passes, LDE 2 passes) are issued back to back for a CHUNK
      // of columns whose working set (evaluations + coefficients + LDE) fits the 256 MiB L3, so every sweep after the first
      // reads what the previous one just wrote from the on-die cache instead of HBM.  CM_FFT_CHUNK_MB: working-set budget
      // (0 = whole group per sweep, the round-2 order).
      const uint32_t chunk_mb = (uint32_t)tune(T_FFT_CHUNK_MB);
      uint32_t per = g.n;
      if (chunk_mb) {
        const uint64_t col_bytes = ((uint64_t)4 << g.log) * (from_coeffs ? 1 : 2) + ((uint64_t)4 << (g.log + cfg.log_blowup_factor));
        per = (uint32_t)std::max<uint64_t>(1, ((uint64_t)chunk_mb << 20) / col_bytes);
        if (per >= g.n || (g.log < 16)) per = g.n;
      }
      // (groups are split only until a fifth of the tree's cells is in the stream in front of the wait — about the length of
      // the LogUp tail; later groups go through whole: a launch of four columns fills the chip badly)
      if (defer && !small_done && early_cells * 5 >= total_cells) late_ready();
      const uint32_t n_early = small_done ? g.n : g.n_early;
      early_cells += (uint64_t)n_early << g.log;
      for (uint32_t c0 = 0, nc = 0; c0 < g.n; c0 += nc) {
        if (c0 >= n_early) late_ready();
        nc = std::min(per, (c0 < n_early ? n_early : g.n) - c0);
        if (cfg.log_blowup_factor == 1 && (!from_coeffs || evals_in_place)) {
          // interpolate + extend by two: the inverse transform's last pass and the forward one's first are one sweep (engine.hpp)
          interpolate_extend(from_coeffs ? (const uint32_t* const*)dco + c0 : dsrc + c0, dco + c0, dld + c0, nc, g.log, *tw, ts);
          continue;
        }
        if (!from_coeffs) interpolate_oop(dsrc + c0, dco + c0, nc, g.log, *tw, ts);
        else if (evals_in_place) interpolate(dco + c0, nc, g.log, *tw, ts);
        evaluate((const uint32_t* const*)dco + c0, dld + c0, nc, g.log, g.log + cfg.log_blowup_factor, *tw, ts);
      }
      late_ready();
      // what can be hashed now: every layer above the next (smaller) group that is still to be transformed
      int next_log = -1;
      for (size_t gj = gi + 1; gj < grps.size(); gj++)
        if (!small_commit_serves(grps[gj].log, cfg.log_blowup_factor)) { next_log = (int)(grps[gj].log + cfg.log_blowup_factor); break; }
      hash_ready(next_log + 1);
      guest_ready(next_log + 1);
    }
    late_ready();
    if (with_merkle) {
      if (s_tr) hash_ready(0);   // (a tree of small columns only: nothing was hashed inside the loop)
      else t.merkle.commit_prepared(s);
    }
    guest_ready(0);
  }
  // the stream the transforms of a pipelined commitment run on (a side stream of the calling thread; A/B: CM_PIPE_STREAM = index)
  static hipStream_t pipe_stream() {
    // CM_PIPE_PRIO: -1 = a stream of the highest priority class (its own hardware queue; the transforms feed the Merkle launches,
    // so they go first), 1 = lowest class, 0 = side stream CM_PIPE_STREAM of the normal class
    static const int prio = getenv("CM_PIPE_PRIO") ? atoi(getenv("CM_PIPE_PRIO")) : PIPE_PRIO_DEFAULT;
    static const int idx = getenv("CM_PIPE_STREAM") ? atoi(getenv("CM_PIPE_STREAM")) : 0;
    return prio ? thread_priority_stream(prio) : thread_side_stream(idx);
  }
  // events of the commitment pipeline: a ring per host thread (a wait captures the record that precedes it, so a slot may be
  // re-recorded while an earlier wait on it is still pending)
  static hipEvent_t pipe_event() {
    static thread_local std::vector<hipEvent_t> ring;
    static thread_local size_t pos = 0;
    if (ring.empty()) {
      ring.resize(64);
      for (auto& e : ring) CM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      const std::vector<hipEvent_t> own = ring;
      at_thread_exit([own] { for (hipEvent_t e : own) (void)hipEventDestroy(e); });
    }
    return ring[pos++ % ring.size()];
  }
  // Host pacing.  With the transcript steps behind a tree on the device the host COULD enqueue the whole next phase while the
  // tree is still being built — but the next phase forks over side streams, and fork waits that sit blocked at the head of
  // the other hardware queues for milliseconds slow the dispatch of the running stream's ~100 small launches: +0.2 ms per
  // tree (CM_PACE=0 shows it; even 0.8 ms of blocked waits cost 50-80 us).  So the host lets the stream drain behind the
  // device-side step and only then enqueues the next phase: one launch latency instead of two or three host round trips.
  // With several proofs in flight (cm_prove_many) other proofs' kernels fill the dispatch slack and running ahead is the
  // better choice (10.3 vs 10.5 ms per proof with 4 in flight), so pacing applies to a lone proof only.
  // `tree`: wait only until that tree's commitment has reached its latency-bound top (MerkleTree::pace_ev) — the fork waits of
  // the next phase then sit blocked for ~0.2 ms instead of milliseconds, and the host's wake-up + launch latency (~50 us of idle
  // GPU per site with the full drain) hides behind the tree top.  CM_PACE_EARLY=0: the full drain.
  void pace(MerkleTree* tree = nullptr) {
    const int mode = tune(T_PACE);   // 0 = always run ahead, 1 = always drain (A/B)
    const bool early = tune(T_PACE_EARLY) != 0;
    const bool drain = mode == 1 || (mode != 0 && g_proofs_in_flight.load(std::memory_order_relaxed) <= 1);
    if (!drain) return;
    if (early && tree && tree->pace_ev && tree->pace_recorded) CM_HIP(hipEventSynchronize(tree->pace_ev));
    else CM_HIP(hipStreamSynchronize(st));
  }
  // one event per tree slot, created once per host thread
  static hipEvent_t pace_event(int slot) {
    static thread_local hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (!ev[slot]) { CM_HIP(hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming)); thread_event_owned(ev[slot]); }
    return ev[slot];
  }
};

// FRI layers of at most 2^fri_tail_log() points are all handled by one single-block launch (k_fri_tail).
inline uint32_t fri_tail_log() {
  static const uint32_t v = [] {
    const char* e = getenv("CM_FRI_TAIL_LOG");
    uint32_t x = e ? (uint32_t)atoi(e) : FRI_TAIL_DEFAULT_LOG;
    return std::min(std::max(x, 1u), FRI_TAIL_MAX_LOG);
  }();
  return v;
}

struct Queries {
  std::vector<uint32_t> positions;
  uint32_t log_domain_size;
  // Queries::generate: n_queries draws of log_domain_size bits, sorted and de-duplicated (BTreeSet order)
  template <class Ch>
  static Queries draw(Ch& ch, uint32_t n_queries, uint32_t log_domain_size) {
    Queries q;
    q.log_domain_size = log_domain_size;
    std::vector<uint32_t>& s = q.positions;
    s.reserve(n_queries);
    const uint32_t mask = (1u << log_domain_size) - 1;
    uint32_t cnt = 0;
    bool done = false;
    while (!done) {
      auto b = ch.draw_random_bytes();
      for (int k = 0; k < 8 && !done; k++) {
        uint32_t w;
        memcpy(&w, b.data() + 4 * k, 4);
        s.push_back(w & mask);
        if (++cnt == n_queries) done = true;
      }
    }
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    return q;
  }
  Queries fold(uint32_t n) const {
    Queries q;
    q.log_domain_size = log_domain_size - n;
    for (auto p : positions) { uint32_t f = p >> n; if (q.positions.empty() || q.positions.back() != f) q.positions.push_back(f); }
    return q;
  }
};

// 4-coordinate values at `pos` of a SecureColumnByCoords, through a GatherBatch
struct QGather { size_t w0 = 0, n = 0; };
inline QGather plan_gather_q(const uint32_t* const col4[4], const std::vector<uint32_t>& pos, GatherBatch& gb) {
  QGather g;
  g.w0 = gb.word_addrs.size();
  g.n = pos.size();
  for (auto p : pos) for (int k = 0; k < 4; k++) gb.add_word(col4[k] + p);
  return g;
}
inline void finish_gather_q(const QGather& g, const GatherBatch& gb, std::vector<QM31>& out) {
  out.reserve(out.size() + g.n);
  for (size_t i = 0; i < g.n; i++) out.push_back(QM31::from_u32(&gb.words[g.w0 + 4 * i]));
}
// compute_decommitment_positions_and_witness_evals (fold step 1): decommitment positions + witness requests
inline QGather plan_fri_positions(const uint32_t* const col4[4], const std::vector<uint32_t>& queries, std::vector<uint32_t>& positions,
                                  GatherBatch& gb) {
  std::vector<uint32_t> wpos;
  wpos.reserve(queries.size());
  positions.reserve(2 * queries.size());
  size_t i = 0;
  while (i < queries.size()) {
    uint32_t start = (queries[i] >> 1) << 1;
    size_t j = i;
    while (j < queries.size() && (queries[j] >> 1) == (queries[i] >> 1)) j++;
    size_t qi = i;
    for (uint32_t pos = start; pos < start + 2; pos++) {
      positions.push_back(pos);
      if (qi < j && queries[qi] == pos) { qi++; continue; }
      wpos.push_back(pos);
    }
    i = j;
  }
  return plan_gather_q(col4, wpos, gb);
}

// FRI commit phase of stwo `prove` (prover.rs:131): first-layer tree over the DEEP quotient columns, circle / line folds, one
// tree per inner layer, the last layer's polynomial — with the transcript steps between the layers on the device.  Shared by
// the single-GPU prover and the sharded one (where FRI is replicated on every rank).  Leaves the trees and layer evaluations
// in place for the decommitment.
// The commit phase picks up BEHIND layers somebody else committed (the sharded prover's row-sharded first layers, whose transcript
// steps already ran on the host channel): the first-layer tree is skipped and the loop starts at `layer_log`.
struct FriResume {
  ColumnSet* layer = nullptr;   // evaluations of layer `layer_log`: line folds done, the circle quotients of that size NOT yet folded in;
                                // null = nothing folded into it yet (the first inner layer)
  uint32_t layer_log = 0;
  size_t qi = 0;                // first quotient group that is not folded yet
  uint32_t n_inner_before = 0;  // inner layers committed before `layer_log` (their queries are folded away in plan_decommit)
  QM31 alpha_c;                 // the circle-fold challenge (the first FRI challenge)
  // device sources (both or none): the layers in front ran their transcript steps on the device and the host has NOT replayed them
  // yet — the device channel {digest[8], n_sent} and the circle-fold challenge are copied from here on the stream, P.ch is stale
  // until the caller's replay (which must run before commit_finish)
  const uint32_t* d_chan = nullptr;
  const uint32_t* d_alpha_c = nullptr;
};
struct FriPhase {
  struct InnerLayer {
    ColumnSet eval; uint32_t log; MerkleTree tree; hostch::Hash32 root;
    std::vector<MerkleTree::CommitLaunch> plan;   // the tree's launches, planned when the fold INTO this layer is enqueued ...
    bool planned = false, leaf_done = false;      // ... because that fold may already have hashed the leaves (fold_line_leaf)
  };
  MerkleTree first_tree;
  std::vector<std::unique_ptr<InnerLayer>> inner;
  bool have_first = true;       // false after a resumed commit: no first-layer tree here
  bool first_leaf_done = false; // the DEEP-quotient kernel wrote first_tree's leaf layer (first_tree.leaf_prealloc)
  uint32_t inner_fold0 = 1;     // folds between the query domain and inner[0]
  void commit(Prover& P, const cm_pcs_config& cfg, std::vector<ColumnSet>& quotients, const std::vector<uint32_t>& q_logs, ProofData& pf,
              const std::function<void()>& while_gpu_busy, const FriResume* resume = nullptr) {
    commit_enqueue(P, cfg, quotients, q_logs, resume);
    while_gpu_busy();   // host-only work of the caller, overlapped with the quotient / FRI kernels enqueued above
    commit_finish(P, cfg, pf, false);
  }
  // the two halves of commit(): everything the GPU does (no host round trip), then the host's replay of the transcript steps and
  // the last layer.  Between them the single-GPU prover enqueues the device-side tail of the proof (tail_device.hpp).
  void commit_enqueue(Prover& P, const cm_pcs_config& cfg, std::vector<ColumnSet>& quotients, const std::vector<uint32_t>& q_logs,
                      const FriResume* resume = nullptr);
  void commit_finish(Prover& P, const cm_pcs_config& cfg, ProofData& pf, bool from_pinned);
  DevBuf d_ar, fri_tables;        // {challenges | roots} of the commit phase; the trees' column tables + the device channel
  ColumnSet last_layer;           // evaluations of the last layer (2^last_log_ values)
  uint32_t n_inner_ = 0, last_log_ = 0;
  bool resumed_ = false;
  uint32_t* d_chan_ = nullptr;    // device copy of the channel {digest[8], n_sent} (inside fri_tables)
  // Decommitment of the FRI trees (first layer over the quotient columns, then one tree per inner layer): decommitment
  // positions + witness evaluations of every layer are requested through the caller's GatherBatch (one gather launch for the
  // whole proof), finish_decommit() distributes what came back.
  std::vector<QGather> first_w, inner_w;
  DecommitPlan first_plan;
  std::vector<DecommitPlan> inner_plan;
  void plan_decommit(const Queries& queries, const std::map<uint32_t, std::vector<uint32_t>>& qpos, const std::vector<ColumnSet>& quotients,
                     const std::vector<uint32_t>& q_logs, GatherBatch& gb) {
    if (have_first) {
      std::map<uint32_t, std::vector<uint32_t>> first_dpos;
      for (size_t k = 0; k < quotients.size(); k++) {
        const uint32_t* c4[4] = {quotients[k].ptrs[0], quotients[k].ptrs[1], quotients[k].ptrs[2], quotients[k].ptrs[3]};
        std::vector<uint32_t> pos;
        first_w.push_back(plan_fri_positions(c4, qpos.at(q_logs[k]), pos, gb));
        first_dpos[q_logs[k]] = std::move(pos);
      }
      first_plan = first_tree.plan_decommit(first_dpos, gb);
    }
    Queries lq = queries.fold(inner_fold0);
    static const bool generic = getenv("CM_FRI_PLAN_GENERIC") != nullptr;   // A/B switch: the per-tree symbolic walk
    if (generic) {
      for (auto& il : inner) {
        const uint32_t* c4[4] = {il->eval.ptrs[0], il->eval.ptrs[1], il->eval.ptrs[2], il->eval.ptrs[3]};
        std::vector<uint32_t> pos;
        inner_w.push_back(plan_fri_positions(c4, lq.positions, pos, gb));
        inner_plan.push_back(il->tree.plan_decommit(il->log, pos, gb));
        lq = lq.fold(1);
      }
      return;
    }
    // The inner layers' trees have columns at the leaves only and their decommitment positions are whole sibling pairs, so the
    // walks of ALL of them come from one family of sets: B[m] = the query positions folded m more times (layer i is queried at
    // B[i]) and W[m] = the siblings of B[m]'s elements that are not in B[m] themselves, ascending.  Layer i's witness evaluations
    // sit at W[i]; its tree needs no hash below level L_i - 1 (both leaves of every touched pair are known), and from level L_i - k
    // exactly the nodes W[i + k] (k = 1 .. L_i - 1: the missing child of every touched parent, parents ascending — Stwo's order).
    // One pass builds the W lists; every tree then only turns indices into addresses.  (The queried leaf values the generic walk
    // also fetched were dropped afterwards: the verifier recomputes them.)
    std::vector<std::vector<uint32_t>> W;
    {
      std::vector<uint32_t> B = lq.positions;
      uint32_t log = lq.log_domain_size;
      for (;; log--) {
        std::vector<uint32_t> w;
        for (size_t j = 0; j < B.size(); j++) {
          const uint32_t sib = B[j] ^ 1u;
          const bool present = (j + 1 < B.size() && B[j + 1] == sib) || (j > 0 && B[j - 1] == sib);
          if (!present) w.push_back(sib);
        }
        W.push_back(std::move(w));
        if (log == 0) break;
        size_t o = 0;
        for (size_t j = 0; j < B.size(); j++) { const uint32_t p = B[j] >> 1; if (o == 0 || B[o - 1] != p) B[o++] = p; }
        B.resize(o);
      }
    }
    for (size_t i = 0; i < inner.size(); i++) {
      InnerLayer& il = *inner[i];
      CM_CHECK(il.log + i == lq.log_domain_size, "fri decommit: layer sizes do not follow the query folds");
      const uint32_t* c4[4] = {il.eval.ptrs[0], il.eval.ptrs[1], il.eval.ptrs[2], il.eval.ptrs[3]};
      inner_w.push_back(plan_gather_q(c4, W[i], gb));
      DecommitPlan plan;
      plan.hash0 = gb.hash_addrs.size();
      for (uint32_t k = 1; k < il.log; k++) {
        const uint32_t* layer = il.tree.layers[il.log - k].u32();
        for (uint32_t idx : W[i + k]) gb.add_hash(layer + (size_t)idx * 8);
      }
      plan.n_hash = gb.hash_addrs.size() - plan.hash0;
      inner_plan.push_back(std::move(plan));
    }
  }
  void finish_decommit(const GatherBatch& gb, ProofData& pf) const {
    if (have_first) {
      for (auto& g : first_w) finish_gather_q(g, gb, pf.fri_first.fri_witness);
      std::vector<uint32_t> qv;
      MerkleTree::finish_decommit(first_plan, gb, qv, pf.fri_first.decommitment);
    }
    for (size_t i = 0; i < inner.size(); i++) {
      FriLayerProofData lp;
      finish_gather_q(inner_w[i], gb, lp.fri_witness);
      std::vector<uint32_t> qv;
      MerkleTree::finish_decommit(inner_plan[i], gb, qv, lp.decommitment);
      lp.commitment = inner[i]->root;
      pf.fri_inner.push_back(std::move(lp));
    }
  }
};
}  // namespace cm
