// Device-side tail of a segment proof: everything stwo `prove` (crates/prover/src/prover.rs:131) does BEHIND the last FRI
// fold — the last layer's polynomial and its mix_felts, the proof of work (GrindOps::grind), mix_u64(nonce), the query draws
// (Queries::generate), the folds of the query positions, and the decommitment of every tree of the proof (the four
// commitment trees, the FRI first-layer tree, one tree per inner FRI layer) with the FRI witness evaluations — as four
// launches enqueued right behind k_fri_tail, with no host round trip in between.  The witnesses land in pinned host memory in
// proof order; the host replays the transcript steps afterwards from what came back (FriPhase::commit_finish, then
// DeviceTail::finish) and refuses the proof on any mismatch, as it does for the other device-side transcript steps.
//
// Round 4 timeline of a lone 2^22-row proof (profiles/r04q_gaps.txt): k_fri_tail -> host (replay, last layer) -> k_grind ->
// host (mix_u64, queries, folds, symbolic decommitment walk of 17 trees: ~0.14 ms) -> upload -> three gathers -> download:
// 0.53 ms from the end of k_fri_tail to the end of the proof with ~0.06 ms of kernels in it.
//
// What makes the walk a table lookup: every query set of a proof is a fold of ONE sorted set S of positions on the largest
// domain (2^L0 points).  With U[k] = unique(S >> k) (the queried nodes of a layer of 2^(L0-k) nodes) and
// W[k] = { u ^ 1 : u in U[k], u ^ 1 not in U[k] } (ascending: the siblings nobody queried),
//  * a commitment tree's decommitment (MerkleProver::decommit) asks layer j (j = top .. 1) for the hashes W[L0 - j] and
//    every column-bearing layer l for the rows U[L0 - l] of its columns (all of them queried values; the column witness is
//    empty because the queries of all sizes are folds of the same set);
//  * FRI layer i (2^Li values) opens whole sibling pairs: witness evaluations at W[L0 - Li], no hash from the leaf layer, and
//    W[L0 - j] from layer j < Li;
//  * the FRI first-layer tree carries the quotient columns of EVERY size: a sibling-only node of a column-bearing layer l has
//    no queried descendant, so both its children are witnesses — the list F[L0 - l] (hashes of layer l + 1 the walk at layer
//    l asks for) interleaves those with the missing children of the queried nodes.
// `tests/test_gpu_prove.py::test_device_tail_equals_host_walk` proves with both forms (cm_set_device_tail) and compares.
#pragma once
#include "engine.hpp"
#include <vector>

namespace cm {

// one contiguous piece of the decommitment, in output order
struct TailDesc {
  const void* p[4];   // HASH*: p[0] = the layer's hashes (8 words per node); ROWS_U: p[0] = DEVICE table of `width` column pointers;
                      // COORDS_W: the four coordinate columns
  uint32_t kind, k, width, pad;
};
enum TailKind : uint32_t {
  TD_HASH_W = 0,    // hashes of the nodes W[k]
  TD_HASH_F = 1,    // hashes of the nodes F[k] (first FRI tree)
  TD_COORDS_W = 2,  // 4 words per row W[k]
  TD_ROWS_U = 3,    // `width` words per row U[k]
};
constexpr uint32_t TAIL_MAX_QUERIES = 1024, TAIL_MAX_SHIFTS = 32, TAIL_MAX_LAST = 64, TAIL_MAX_POW_BITS = 26;
constexpr uint32_t TAIL_HDR_WORDS = 16;   // pinned header: {status, nonce lo, nonce hi, n_unique, total_words, pow miss, degree error}
constexpr uint32_t TAIL_HDR_LAST_DONE = 7, TAIL_HDR_TABLES_DONE = 8;   // header words the host watches (1 = that kernel's pinned words are written)
enum TailStatus : uint32_t { TAIL_OK = 0, TAIL_NO_NONCE = 1 };   // (a degree violation of the last layer is hdr[6], read by DeviceTail::wait_last)

struct TailLastArgs {
  const uint32_t* d_ar;          // {alphas | roots} of the FRI commit phase
  uint32_t n_ar_words;
  const uint32_t* last[4];       // the last layer's evaluations (4 coordinates x n)
  uint32_t log_n, log_keep;      // n = 2^log_n values, 2^log_keep coefficients kept
  uint32_t ninv;                 // 1 / n
  uint32_t xinv[TAIL_MAX_LAST];  // line-IFFT twiddles: level l, pair h at [(n - (n >> l)) + h]  (n - 1 words)
  uint32_t* chan;                // device channel {digest[8], n_sent}
  uint32_t* h_ar;                // pinned: copy of d_ar
  uint32_t* h_last;              // pinned: the last layer's evaluations, coordinate-major (4 n words)
  uint32_t* hdr;                 // pinned header
  unsigned long long* nonce;     // device: set to ~0
};
void tail_last_layer(const TailLastArgs& a, hipStream_t st);
// smallest nonce in [0, 2^(bits + 4)) whose mix_u64 hash has `bits` trailing zero bits; digest read from `chan`
void tail_grind(const uint32_t* d_chan, uint32_t bits, unsigned long long* d_nonce, hipStream_t st);

struct TailTablesArgs {
  uint32_t* chan;
  const unsigned long long* nonce;
  uint32_t n_queries, log_domain;   // L0
  uint32_t qmask;                   // bit l set: the first FRI tree carries columns of 2^l rows (F lists)
  uint32_t n_desc;
  const TailDesc* h_desc;           // pinned host memory (read once by the kernel)
  TailDesc* d_desc;                 // device copy
  uint32_t* d_off;                  // per descriptor: first output word
  uint32_t* tab;                    // device tables: {cntU[32] | cntW[32] | cntF[32] | U[32][NQ] | W[32][NQ] | F[32][4 NQ]}
  uint32_t nq_pad;                  // NQ: power of two >= n_queries
  uint32_t* hdr;                    // pinned header
  uint32_t* h_positions;            // pinned: S (n_unique words)
};
void tail_tables(const TailTablesArgs& a, hipStream_t st);
inline size_t tail_tab_words(uint32_t nq_pad) { return 3 * TAIL_MAX_SHIFTS + (size_t)TAIL_MAX_SHIFTS * nq_pad * 6; }
// the gathers: descriptors [0, n_small) are hash / coordinate pieces, [n_small, n_desc) row pieces
void tail_gather(const TailDesc* d_desc, const uint32_t* d_off, const uint32_t* d_tab, uint32_t nq_pad, uint32_t n_small, uint32_t n_desc,
                 uint32_t n_queries, uint32_t* out, hipStream_t st);

// host mirror of the tables (counts and lists): same definitions as the kernel, as linear merges over the sorted set
struct TailTables {
  uint32_t L0 = 0, n = 0;
  std::vector<uint32_t> buf;                         // [U | W | F] per shift: n + n + 3 n words
  uint32_t cnt[3][TAIL_MAX_SHIFTS] = {};             // 0 = U, 1 = W, 2 = F
  const uint32_t* list(uint32_t which, uint32_t k) const { return buf.data() + (size_t)k * 5 * n + (which == 0 ? 0 : which == 1 ? n : 2 * n); }
  uint32_t* list(uint32_t which, uint32_t k) { return buf.data() + (size_t)k * 5 * n + (which == 0 ? 0 : which == 1 ? n : 2 * n); }
  void build(const std::vector<uint32_t>& S, uint32_t log_domain, uint32_t qmask);
  size_t count(uint32_t kind, uint32_t k) const { return cnt[kind == TD_ROWS_U ? 0 : kind == TD_HASH_F ? 2 : 1][k]; }
  static uint32_t words_per_item(const TailDesc& d) { return d.kind == TD_ROWS_U ? d.width : d.kind == TD_COORDS_W ? 4u : 8u; }
};

// pinned buffers of the calling thread for the tail's results (pool.hip): descriptors in, witnesses out
void* tail_pinned_desc(size_t bytes);
void* tail_pinned_out(size_t bytes);

}  // namespace cm
