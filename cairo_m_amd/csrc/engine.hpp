// Internal C++ interface of the HIP engine: host-side launch wrappers shared by the C ABI
// (capi.hip) and the prover driver (prover.hip).  All pointer-array arguments named d_* are
// DEVICE arrays of device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <sched.h>
#include <vector>
#include <functional>
#include <string.h>
#include <utility>
#include "field.hpp"
#include "device_common.hpp"

namespace cm {

struct Twiddles {
  uint32_t R = 0;  // tables cover CanonicCoset(R).circle_domain() and every smaller canonic domain
  uint32_t *xtw = nullptr, *ixtw = nullptr, *ytw = nullptr, *iytw = nullptr;   // every entry is 2 * twiddle (< 2^32): see mul_tw2
  uint32_t* scratch = nullptr;   // point tables the build kernels read (twiddles_scratch_words(R) words)
};
Twiddles* twiddles_create(uint32_t R, hipStream_t st);
size_t twiddles_scratch_words(uint32_t R);
void twiddles_build(const Twiddles& t, hipStream_t st);   // caller-allocated buffers (R set, 2^(R-1) / 2^R words)
void twiddles_destroy(Twiddles* t);

// in-place IFFT of ncols columns of 2^n (bit-reversed evals -> coefficients)
void interpolate(uint32_t* const* d_cols, uint32_t ncols, uint32_t n, const Twiddles& tw, hipStream_t st);
void interpolate_oop(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n, const Twiddles& tw,
                     hipStream_t st);
// coefficients (2^n_in each) -> evaluations on the canonic domain of log n_out (n_out >= n_in)
void evaluate(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n_in, uint32_t n_out,
              const Twiddles& tw, hipStream_t st);
// interpolate + evaluate on the domain of twice the size (extend_evals at log_blowup_factor 1); d_src may equal d_coeffs
void interpolate_extend(const uint32_t* const* d_src, uint32_t* const* d_coeffs, uint32_t* const* d_lde, uint32_t ncols, uint32_t n,
                        const Twiddles& tw, hipStream_t st);
void bit_reverse_columns(uint32_t* const* d_cols, uint32_t ncols, uint32_t n, hipStream_t st);
// interpolate + extend of small columns of ANY mix of sizes in one launch (one block per column): src (2^log_n evaluations;
// null = the coefficients are already in `coeffs`; may alias `coeffs`) -> coeffs (2^log_n) -> lde (2^(log_n + blowup));
// inv_n = 2^-log_n as a canonical M31 word
constexpr uint32_t SMALL_COMMIT_MAX_LOG = 10;
struct SmallCommitJob { const uint32_t* src; uint32_t* coeffs; uint32_t* lde; uint32_t log_n, inv_n; };
struct TwiddleTables { uint32_t R; const uint32_t *xtw, *ixtw, *ytw, *iytw; };
void small_commit(const SmallCommitJob* d_jobs, uint32_t n_jobs, uint32_t max_log, uint32_t blowup, const Twiddles& tw, hipStream_t st);
inline bool small_commit_serves(uint32_t log, uint32_t blowup) {   // A/B switch: CM_NO_SMALL_COMMIT=1
  static const bool on = getenv("CM_NO_SMALL_COMMIT") == nullptr;
  return on && log >= 1 && log <= SMALL_COMMIT_MAX_LOG && SMALL_COMMIT_MAX_LOG + blowup <= 14;
}

// eval_at_point for a batch of equal-size coefficient columns at ONE point.
// d_scratch: >= 4*(2^10 + 2^max(0,n-10)) + 4*ncols*2^max(0,n-10) + 32*4 u32.  d_out: 4*ncols u32 (device).
size_t eval_at_point_scratch_words(uint32_t ncols, uint32_t n);
void eval_at_point_batch(const uint32_t* const* d_coeffs, uint32_t ncols, uint32_t n, const QM31& px, const QM31& py,
                         uint32_t* d_scratch, uint32_t* d_out, hipStream_t st);

// every (log size, point) sampling group of a proof in three launches; d_out receives 4 * ncols words per job
struct EapJob { uint32_t log_n, ncols; const uint32_t* const* d_coeffs; QM31 px, py; uint32_t* d_out;
                bool has_shift = false; uint32_t shift_x = 0, shift_y = 0;
                uint32_t* h_out = nullptr; };   // optional mirror of d_out in PINNED HOST memory, written by the same kernel (no copy command)
// d_oods_t != null: every job samples at the OODS point derived ON THE DEVICE from the felt at d_oods_t (4 words; the draw of
// CirclePoint::get_random_point), plus the job's M31 shift (has_shift: the previous-row mask point) — px / py are ignored
void eval_at_point_multi(const std::vector<EapJob>& jobs, hipStream_t st, const uint32_t* d_oods_t = nullptr);

// Merkle
void merkle_layer(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                  uint32_t* d_out, hipStream_t st);
// layers 2^top_log .. 2^0 in one launch (top_log <= MERKLE_TAIL_LOG)
constexpr uint32_t MERKLE_TAIL_LOG = 8;  // 256 nodes = one 1024-thread block of lane quads
struct MerkleTailArgs {
  uint32_t top_log;
  const uint32_t* prev;                       // hashes of layer top_log + 1, or null
  const uint32_t* const* cols;                // device array of all (sorted) column pointers of the tree
  uint32_t col_begin[MERKLE_TAIL_LOG + 1];    // column range of layer l in `cols`
  uint32_t col_end[MERKLE_TAIL_LOG + 1];
  uint32_t* layers[MERKLE_TAIL_LOG + 1];      // output buffer of layer l
};
void merkle_tail(const MerkleTailArgs& a, hipStream_t st);
// one layer with one node per quad of lanes (wide-and-short layers)
void merkle_layer_quad(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                       uint32_t* d_out, hipStream_t st);
constexpr uint32_t MERKLE_QUAD_MAX_LOG = 12, MERKLE_QUAD_MIN_COLS = 64;
// up to MERKLE_MULTI_LEVELS consecutive layers per launch (top layer must have >= 256 nodes)
constexpr uint32_t MERKLE_MULTI_LEVELS = 4;
constexpr uint32_t MERKLE_PACE_LOG = 18;     // MerkleTree::pace_ev sits in front of the first layer of at most 2^18 nodes
constexpr uint32_t MERKLE_MULTI_MAX_TOP = 19;  // layers of 2^19 nodes and more get their own launch
struct MerkleMultiArgs {
  uint32_t top_log, n_levels;
  const uint32_t* prev;                        // hashes of layer top_log + 1, or null
  const uint32_t* const* cols;                 // device array of all (sorted) column pointers of the tree
  uint32_t col_begin[MERKLE_MULTI_LEVELS];     // column range of level lv (layer top_log - lv) in `cols`
  uint32_t col_end[MERKLE_MULTI_LEVELS];
  uint32_t* layers[MERKLE_MULTI_LEVELS];       // output buffer of level lv
};
void merkle_multi(const MerkleMultiArgs& a, double alg_bytes, hipStream_t st);
// The whole top of a tree — layers 2^top_log (9 <= top_log <= MERKLE_TOP_MAX_LOG) down to the root — in ONE launch:
// every block reduces 256 nodes of layer top_log to one node of layer top_log - 8, the last block to finish (ticket)
// hashes the remaining <= 256 nodes up to the root.  Replaces 2-3 k_merkle_multi launches + k_merkle_tail.
constexpr uint32_t MERKLE_TOP_MAX_LOG = 16;
// Optional extras of the tree-top launch (FRI layer trees, round 5):
//  * the transcript step behind the tree — Blake2sChannel::mix_root(root) + draw_felt() — by the block that hashes the root
//    (chan != null): one launch less on the protocol-serial chain of every FRI layer (k_chan_mix_root_draw: ~5 us each);
//  * the fold that PRODUCES the layer (fold_mode 1 = fold_line of `fold_src`, 2 = fold_line + fold_circle of the quotient columns
//    `fold_circ` of that size): the leaf level computes the folded value, stores it to `fold_dst` (= the tree's four columns) and
//    hashes it in registers.  Only when the launch covers the whole tree (top_log = the layer's log, prev = null).
struct MerkleTopExtra {
  uint32_t *chan = nullptr, *felt_out = nullptr, *root_log = nullptr;
  uint32_t fold_mode = 0;
  const uint32_t* fold_src[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint32_t* fold_circ[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* fold_dst[4] = {nullptr, nullptr, nullptr, nullptr};
  const uint32_t *ixt = nullptr, *iyt = nullptr;      // inverse twiddles of the fold, indexed by output position (entries are 2 * value)
  const uint32_t *alpha = nullptr, *alpha_c = nullptr;
};
struct MerkleTopArgs {
  uint32_t top_log;
  const uint32_t* prev;                          // hashes of layer top_log + 1, or null
  const uint32_t* const* cols;                   // device array of all (sorted) column pointers of the tree
  uint32_t col_begin[MERKLE_TOP_MAX_LOG + 1];    // column range of layer l in `cols`
  uint32_t col_end[MERKLE_TOP_MAX_LOG + 1];
  uint32_t* layers[MERKLE_TOP_MAX_LOG + 1];      // output buffer of layer l
  uint32_t* ticket;                              // zero on entry; the last block leaves it zero again
  MerkleTopExtra x;
};
void merkle_top(MerkleTopArgs& a, hipStream_t st);   // fills a.ticket
// device-side transcript step of the FRI commit phase: chan = {digest[8], n_sent} (9 u32);
// chan <- mix_root(root); felt_out[4] <- draw_felt(); root_log[8] <- root (read back once at the end)
void chan_mix_root_draw(uint32_t* d_chan, const uint32_t* d_root, uint32_t* d_felt_out, uint32_t* d_root_log, hipStream_t st);
// sharded FRI layer (kernels_hash.hip k_shard_top_step): N gathered sub-roots -> top levels -> root -> mix_root + draw_felt on d_chan
void shard_top_step(const uint32_t* d_sub, uint32_t n, uint32_t* d_chan, uint32_t* d_felt_out, uint32_t* d_root_log, uint32_t* d_sub_log, hipStream_t st);
// the top levels alone (kernels_hash.hip k_shard_top): N gathered sub-roots -> root (8 words at d_root_out), sub-roots kept at d_sub_log
void shard_top(const uint32_t* d_sub, uint32_t n, uint32_t* d_root_out, uint32_t* d_sub_log, hipStream_t st);
// same, with the channel state {digest[8], n_sent} passed in the kernel arguments and stored to d_chan first
void chan_init_mix_root_draw(const uint32_t init9[9], uint32_t* d_chan, const uint32_t* d_root, uint32_t* d_felt_out, uint32_t* d_root_log,
                             hipStream_t st);
// transcript step behind the trace commitment on the device: mix_root(root), interaction PoW, mix_u64(nonce), Relations::draw.
// d_rel_z[n_rel][4], d_rel_pow[n_rel][max_rel][4] (= DevRelations); d_out16 = {root[8], nonce[2], n_sent, error, z0[4]}
void step_pow_relations(const uint32_t init9[9], const uint32_t* d_root, uint32_t pow_bits, uint32_t n_rel, uint32_t max_rel, uint32_t* d_rel_z,
                        uint32_t* d_rel_pow, uint32_t* d_out16, hipStream_t st);
// d_powers[g] = rho^(n - 1 - g) (QM31 words), rho read from device memory
void coeff_powers(const uint32_t* d_rho, uint32_t* d_powers, uint32_t n, hipStream_t st);
uint64_t grind_gpu(const uint8_t digest[32], uint32_t bits, hipStream_t st);
// out[q * width + w] = addrs[q][w]  (decommitment gathers: width 1 = values, 8 = hashes)
void gather_words(const uint32_t* const* d_addrs, uint32_t n, uint32_t width, uint32_t* d_out, hipStream_t st);
// out[run.out_off + c] = run.d_cols[c][run.row]  (decommitment: one queried row of a run of columns)
struct RowRun { const uint32_t* const* d_cols; uint32_t n_cols, row, out_off, pad; };
void gather_runs(const RowRun* d_runs, uint32_t n_runs, uint32_t* d_out, hipStream_t st);

// Per-thread resources (streams, events, pinned buffers, the upload ring) are created on first use by whichever host thread enters
// the library; `at_thread_exit` registers their release for the moment that thread ends (reverse order).  Without it a service
// that proves from short-lived threads leaked ~35 MB of pinned host memory, nine streams and a few dozen events per thread — the
// round-4 soak (tools/stress_pipeline.py: three new threads per round for ten minutes) took the test box down twice that way.
void at_thread_exit(std::function<void()> f);
// pool.cpp-style services implemented in pool.hip
void* pool_get(size_t bytes);
void pool_put(void* p);
void pool_trim();
// Buffers a finished proof keeps OUTSIDE the pool for a while (the deferred teardown of prover.hip) register one release function per
// host thread here; pool_trim(), cm_shutdown(), the pool's out-of-memory retry and prove_sharded() run it first, so that "hand
// everything back" means everything and a parked FRI phase can never be the reason an allocation fails.
void set_thread_parked_release(std::function<void()> f);
void release_thread_parked();
void stage_upload(void* dst, const void* src, size_t bytes, hipStream_t st);

// The HIP device cm_init() selected (one device per process: one process per GPU).  hipSetDevice is per host
// thread, so every host thread that enters the library is bound to that device on its first call.
void set_library_device(int device);
void bind_thread_to_library_device();
// CPU placement of proving threads (pool.hip): scoped to one cm_prove_* call on caller threads, permanent on the library's own
// cm_prove_many workers; cm_set_cpu_affinity(0) / CM_CPU_AFFINITY=0 / CM_NO_CPU_AFFINITY=1 turn it off.
struct AffinityScope {
  cpu_set_t saved;
  bool active = false;
  AffinityScope();
  ~AffinityScope();
  AffinityScope(const AffinityScope&) = delete;
  AffinityScope& operator=(const AffinityScope&) = delete;
};
void set_cpu_affinity_mode(int mode);
int cpu_affinity_mode();
void bind_worker_thread_cpus();
// forget a caller stream that is about to be destroyed (waits for its copies out of the calling thread's upload ring)
void stage_forget_stream(hipStream_t st);

// The prover's main stream of the calling host thread (created on first use, non-blocking): concurrent proofs
// from different host threads run on different streams and overlap on the GPU.
hipStream_t thread_main_stream();
// measurement switches (cm_set_tuning; include/cairom_hip.h lists them): initial values from the environment, flipped at run time
// so that two forms can be timed alternately inside one process (tools/ab_switch.py)
enum TuneKey { T_OODS_POLL, T_OODS_HOST_WRITE, T_STAGE_COPY_KERNEL, T_STAGE_LAZY_EVENTS, T_DEFER_TEARDOWN, T_FLAG_JOIN, T_FLAG_FORK,
               T_COMMIT_PREP_EARLY, T_TRACE_HIST_FUSE, T_FRI_TOP_FUSE, T_LOGUP_DEFER, T_OODS_SPLIT,
               // policy choices of rounds 2-4 (numeric where the old environment variable was)
               T_FORK_MAIN, T_MERKLE_NPW, T_FORK_WIDTH, T_PP_SIDE, T_TREE0_PRIO, T_TREE1_FIRST, T_LOGUP_WIDTH, T_QUOT_ROWS, T_FRI_FOLD_LEAF,
               T_FFT_FUSED, T_COMMIT_PIPE, T_FFT_CHUNK_MB, T_PACE, T_PACE_EARLY, T_TAIL_FLAGS,
               // test hook: > 0 caps the device tail's proof-of-work search at 2^(value-1) nonces so that the host fallback runs
               T_TAIL_GRIND_CAP,
               // round 6: launch order inside the fork regions
               T_CONS_WIDE_FIRST, T_LOGUP_SMALL_STREAM, T_CONS_PLAN, T_QUOT_LEAF, T_SHARD_FRI_STREAM, T_FFT_HALF_OCC, T_SHARD_HALO, T_TW_BATCH, T_TREE0_GUEST, T_MERKLE_MULTI_TOP, T_SHARD_TREE_STREAM, T_SHARD_FRI_STOP_LOG, T_COUNT };
struct TuneEntry { const char* key; const char* env; int dflt, lo, hi; };
extern const TuneEntry TUNE_TABLE[T_COUNT];
std::atomic<int>* tune_values();
inline int tune(TuneKey k) { return tune_values()[k].load(std::memory_order_relaxed); }
// side stream i of the calling host thread (the streams Fork hands out), with NO ordering against anything: the caller orders it
// with events (Prover::commit_enqueue runs the transforms of a commitment there, next to the Merkle launches on the main stream)
hipStream_t thread_side_stream(int i);
hipStream_t thread_priority_stream(int rel);   // rel < 0: highest priority class, > 0: lowest (own hardware queues; pool.hip)

// Fork/join over a small set of side streams (thread-local, created once): independent per-component
// launches of one phase run concurrently instead of serialising 34 tiny kernels on one stream.
// Fork f(main); launch on f.stream(i) ...; f.join();  — every side stream first waits for everything
// enqueued on `main` before the fork, and `main` waits for all side work at join().
// stream(Fork::MAIN) is the main stream itself: the LONGEST job of a region goes there — it starts without the cross-queue
// hand-over of the fork (event -> barrier packet on another hardware queue: 15-50 us on gfx950) and, finishing last, finds the
// side streams' join events already signalled (A/B: CM_FORK_MAIN=0 sends it to a side stream like everything else).
struct Fork {
  static constexpr int N = 8;
  static constexpr int MAIN = 1 << 20;
  static int main_or(int side_index);   // MAIN, or side_index when CM_FORK_MAIN=0
  hipStream_t main;
  uint32_t used = 0, fork_epoch = 0;
  bool joined = false;
  explicit Fork(hipStream_t main_stream);
  hipStream_t stream(int i);
  void join();
  ~Fork();
};

void fork_join_check();   // throws if a flag-join collector of the calling thread timed out (pool.hip); main stream synchronised

// simple RAII device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() {}
  explicit DevBuf(size_t b) { alloc(b); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  // Buffers come from a caching pool (pool.hpp): after the first proof no hipMalloc/hipFree (and none of
  // hipFree's implicit device syncs) happen on the hot path.  Reuse is stream-ordered: the prover runs on
  // one stream, and the per-op C-ABI wrappers synchronise before their temporaries are released.
  void alloc(size_t b) {
    release();
    if (b == 0) b = 4;
    p = pool_get(b);
    bytes = b;
  }
  void release() {
    if (p) pool_put(p);
    p = nullptr;
    bytes = 0;
  }
  uint32_t* u32() const { return (uint32_t*)p; }
  template <class T> T* as() const { return (T*)p; }
};
// Device->host results land in a per-thread pinned buffer: truly asynchronous (pool.hip).  Valid after the stream is
// synchronised and until the next call on this thread.
const void* stage_download_async(const void* src, size_t bytes, hipStream_t st);
void* stage_landing(size_t bytes, hipStream_t st);   // the buffer alone (>= bytes): the caller enqueues its own copies into it
// 1024 pinned words per host thread; fixed slots (words): 0 range-check flag, 2-3 grind nonce, 8-15 Merkle root,
// 16-23 root of tree 0, 32-167 claimed sums, 172-175 random coefficient + 176-183 root of tree 2 (one copy), 208-219 the OODS felt + root 3 of the device-side step, 256-351 FRI challenges, 384-575 FRI roots, 640-1023 last FRI layer
uint32_t* pinned_words();
constexpr uint32_t INTERACTION_POW_BITS = 2;   // relations::INTERACTION_POW_BITS (prover.rs:90, verifier.rs:55-58)
enum PinnedSlot : uint32_t { PIN_FLAG = 0, PIN_NONCE = 2, PIN_ROOT = 8, PIN_ROOT0 = 16, PIN_SUMS = 32, PIN_COEFF = 172, PIN_ROOT2 = 176, PIN_STEP1 = 192, PIN_STEP3 = 208, PIN_ALPHAS = 256, PIN_ROOTS = 384,
                            PIN_LAST_LAYER = 640, PIN_WORDS = 1024 };
// Small host->device uploads (pointer arrays, coefficients, positions) go through a pinned staging ring
// and hipMemcpyAsync on the launch stream: no host sync, no pageable-copy stall.
template <class T>
inline DevBuf upload(const std::vector<T>& v, hipStream_t st) {
  DevBuf b(v.size() * sizeof(T));
  if (!v.empty()) stage_upload(b.p, v.data(), v.size() * sizeof(T), st);
  return b;
}

// Several small host tables -> ONE host->device copy.  add() records where the device address of a table has
// to be stored; flush() uploads the concatenation and patches those pointers.  The returned buffer owns the
// tables and must outlive their users.
struct UploadBatch {
  std::vector<uint8_t> blob;
  std::vector<std::pair<void**, size_t>> fixups;
  template <class T, class U>
  void add(const std::vector<T>& v, U** out_dev_ptr) {
    size_t off = (blob.size() + 15) & ~(size_t)15;
    blob.resize(off + v.size() * sizeof(T));
    if (!v.empty()) memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    fixups.push_back({(void**)out_dev_ptr, off});
  }
  DevBuf flush(hipStream_t st) {
    DevBuf b = upload(blob, st);
    for (auto& f : fixups) *f.first = (uint8_t*)b.p + f.second;
    return b;
  }
};

}  // namespace cm
