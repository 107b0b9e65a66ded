// Device-memory caching pool + pinned upload ring.  Sized for a 288 GB part: freed blocks are kept
// (never returned to the driver on the hot path) and reused best-fit, so a steady-state proof performs
// zero hipMalloc/hipFree calls; cm_shutdown()/pool_trim() hands everything back.
#include <algorithm>
#include "engine.hpp"
#include "tail_device.hpp"
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string.h>
#include <ctype.h>
#include <sched.h>
#include <stdio.h>
#include <string>

namespace cm {
namespace {
struct Pool {
  std::mutex mu;
  // Free blocks by capacity class.  A proof asks for the same ~1500 sizes every time and they fall into ~100 classes: a class
  // is a vector (push / pop at the back), the ordered map only holds the classes — it stays in cache, and neither path
  // allocates a node (a multimap entry per block cost 0.2-0.5 us per get / put once the tree had a few thousand scattered nodes;
  // the teardown of one proof is ~600 puts with the GPU idle).
  std::map<size_t, std::vector<void*>> free_;  // capacity -> blocks
  std::unordered_map<void*, size_t> cap_;      // every live or cached block
  static size_t round(size_t b) {
    if (b <= (1u << 20)) return (b + 511) & ~(size_t)511;
    return (b + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
  }
  void* get(size_t bytes) {
    size_t want = round(bytes);
    {
      std::lock_guard<std::mutex> lk(mu);
      // accept a cached block if it wastes at most 25 % (exact sizes recur proof after proof)
      for (auto it = free_.lower_bound(want); it != free_.end() && it->first <= want + want / 4; ++it)
        if (!it->second.empty()) {
          void* p = it->second.back();
          it->second.pop_back();
          return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      release_thread_parked();   // a previous proof's deferred teardown (prover.hip) goes back to the pool first ...
      trim();                    // ... and the pool back to the driver
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) throw CmError(2, std::string("device allocation of ") + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(mu);
    cap_[p] = want;
    return p;
  }
  void put(void* p) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cap_.find(p);
    if (it == cap_.end()) { (void)hipFree(p); return; }
    free_[it->second].push_back(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : free_)
      for (void* p : kv.second) { (void)hipFree(p); cap_.erase(p); }
    free_.clear();
  }
};
// One pool per host thread: a freed block may still be read by kernels queued on the freeing thread's stream,
// so it may only be handed to work that is ordered after them, i.e. to the same thread (= same main stream).
// Concurrent proofs (one host thread each) therefore never exchange blocks.
// The pool object itself is leaked on purpose (DevBufs released by late destructors still find it); the CACHED blocks
// of a host thread go back to the driver when that thread exits — prover threads that come and go would otherwise pin
// a few GB of HBM each.
struct PoolExitGuard { Pool* p = nullptr; ~PoolExitGuard() { if (p) p->trim(); } };
Pool& pool() {
  static thread_local Pool* p = new Pool();
  static thread_local PoolExitGuard guard;
  guard.p = p;
  return *p;
}

struct Stage {
  std::mutex mu;
  uint8_t* base = nullptr;
  size_t size = (size_t)32 << 20, off = 0;
  // One event per stream with copies out of the ring since the last wrap (main stream + Fork side streams + caller streams
  // of the per-op C ABI), re-recorded behind every copy.  The wrap waits on the EVENTS, never on the stream handles: a caller
  // may have destroyed its cm_stream_t in the meantime (hipStreamDestroy lets queued work finish; a recorded event stays valid).
  struct User { hipStream_t st; hipEvent_t ev; bool dirty = false; };   // dirty: copies behind the last record of `ev` (lazy form)
  std::vector<User> users;
  std::vector<hipEvent_t> spare;
  void ensure() {
    if (!base) CM_HIP(hipHostMalloc((void**)&base, size, hipHostMallocDefault));
  }
};
Stage& stage() {   // ring wrap syncs only the owner's stream
  static thread_local Stage* s = nullptr;
  if (!s) {
    s = new Stage();
    Stage* own = s;
    at_thread_exit([own] {
      for (const Stage::User& u : own->users) {
        if (u.dirty) (void)hipStreamSynchronize(u.st);   // (the thread's main stream: alive, or already drained by its own exit hook)
        (void)hipEventSynchronize(u.ev); (void)hipEventDestroy(u.ev);
      }
      for (hipEvent_t e : own->spare) (void)hipEventDestroy(e);
      if (own->base) (void)hipHostFree(own->base);
      delete own;
    });
  }
  return *s;
}
}  // namespace

namespace {
struct ThreadExit {
  std::vector<std::function<void()>> fns;
  ~ThreadExit() {
    for (auto it = fns.rbegin(); it != fns.rend(); ++it) {
      try { (*it)(); } catch (...) {}   // never throw out of a thread's teardown
    }
  }
};
}  // namespace
void at_thread_exit(std::function<void()> f) {
  static thread_local ThreadExit te;
  te.fns.push_back(std::move(f));
}
void* pool_get(size_t bytes) { return pool().get(bytes); }
void pool_put(void* p) { pool().put(p); }
namespace { std::function<void()>& parked_release_fn() { static thread_local std::function<void()> f; return f; } }
void set_thread_parked_release(std::function<void()> f) { parked_release_fn() = std::move(f); }
void release_thread_parked() { if (parked_release_fn()) parked_release_fn()(); }
void pool_trim() { release_thread_parked(); pool().trim(); }

// The copy out of the ring as a KERNEL that reads the pinned words over PCIe: a hipMemcpyAsync of more than a few KB goes through
// the SDMA engine, and a copy-engine command between two kernels of a compute stream costs ~10 us on either side of its 7 us
// (four of them sat on the critical path of a proof: profiles/r05p_gaps.txt).  A/B: CM_STAGE_COPY_KERNEL=0.
__global__ void __launch_bounds__(256) k_stage_copy(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t n16,
                                                    uint8_t* dst_tail, const uint8_t* src_tail, uint32_t n_tail) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
void stage_upload(void* dst, const void* src, size_t bytes, hipStream_t st) {
  const bool copy_kernel = tune(T_STAGE_COPY_KERNEL) != 0;
  // the thread's own main stream outlives every copy enqueued on it: its event is only recorded when the ring wraps (an event
  // record per upload is a barrier packet per upload, ~20 per proof).  A/B: CM_STAGE_LAZY_EVENTS=0.
  const bool lazy_events = tune(T_STAGE_LAZY_EVENTS) != 0;
  if (!bytes) return;
  Stage& s = stage();
  std::lock_guard<std::mutex> lk(s.mu);
  s.ensure();
  size_t need = (bytes + 255) & ~(size_t)255;
  if (need > s.size / 2) {
    // oversize: straight from the caller's (pageable) memory, but ORDERED on `st` — the destination usually comes from the
    // caching pool, whose reuse is only stream-ordered, and a NULL-stream hipMemcpy is not ordered against the library's
    // non-blocking streams (earlier kernels on `st` could still be using the recycled block).
    CM_HIP(hipStreamSynchronize(st));
    CM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    CM_HIP(hipStreamSynchronize(st));
    return;
  }
  if (s.off + need > s.size) {
    // wrap: every copy still reading the ring must have executed, on whichever stream it was enqueued
    for (const Stage::User& u : s.users) {
      if (u.dirty) CM_HIP(hipEventRecord(u.ev, u.st));
      CM_HIP(hipEventSynchronize(u.ev)); s.spare.push_back(u.ev);
    }
    s.users.clear();
    s.off = 0;
  }
  auto it = std::find_if(s.users.begin(), s.users.end(), [&](const Stage::User& u) { return u.st == st; });
  if (it == s.users.end()) {
    hipEvent_t ev;
    if (!s.spare.empty()) { ev = s.spare.back(); s.spare.pop_back(); }
    else CM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    s.users.push_back({st, ev});
    it = s.users.end() - 1;
  }
  memcpy(s.base + s.off, src, bytes);
  if (copy_kernel && ((uintptr_t)dst & 15) == 0) {
    const uint32_t n16 = (uint32_t)(bytes / 16), n_tail = (uint32_t)(bytes % 16);
    const uint32_t blocks = std::min<uint32_t>(std::max<uint32_t>((n16 + 255) / 256, 1u), 64u);
    hipLaunchKernelGGL(k_stage_copy, dim3(blocks), dim3(256), 0, st, (uint4*)dst, (const uint4*)(s.base + s.off), n16,
                       (uint8_t*)dst + (size_t)n16 * 16, s.base + s.off + (size_t)n16 * 16, n_tail);
    CM_HIP(hipGetLastError());
  } else {
    CM_HIP(hipMemcpyAsync(dst, s.base + s.off, bytes, hipMemcpyHostToDevice, st));
  }
  if (lazy_events && st == thread_main_stream()) it->dirty = true;
  else { CM_HIP(hipEventRecord(it->ev, st)); it->dirty = false; }
  s.off += need;
}
const TuneEntry TUNE_TABLE[T_COUNT] = {
    {"oods_poll", "CM_OODS_POLL", 1, 0, 1},           {"oods_host_write", "CM_OODS_HOST_WRITE", 1, 0, 1}, {"stage_copy_kernel", "CM_STAGE_COPY_KERNEL", 1, 0, 1},
    {"stage_lazy_events", "CM_STAGE_LAZY_EVENTS", 1, 0, 1}, {"defer_teardown", "CM_DEFER_TEARDOWN", 1, 0, 1}, {"flag_join", "CM_FLAG_JOIN", 1, 0, 1},
    {"flag_fork", "CM_FLAG_FORK", 1, 0, 1},           {"commit_prep_early", "CM_COMMIT_PREP_EARLY", 1, 0, 1}, {"trace_hist_fuse", "CM_TRACE_HIST_FUSE", 1, 0, 1},
    {"fri_top_fuse", "CM_FRI_TOP_FUSE", 1, 0, 1},     {"logup_defer", "CM_LOGUP_DEFER", 1, 0, 1},         {"oods_split", "CM_OODS_SPLIT", 780, 0, 1000},
    {"fork_main", "CM_FORK_MAIN", 1, 0, 1},           {"merkle_npw", "CM_MERKLE_NPW", -1, -1, 8},         {"fork_width", "CM_FORK_WIDTH", 0, 0, 8},
    {"pp_side", "CM_PP_SIDE", 1, 0, 1},               {"tree0_prio", "CM_TREE0_PRIO", -1, -1, 1},         {"tree1_first", "CM_TREE1_FIRST", 1, 0, 1},
    {"logup_width", "CM_LOGUP_WIDTH", 4, 1, 7},       {"quot_rows", "CM_QUOT_ROWS", 2, 1, 4},             {"fri_fold_leaf", "CM_FRI_FOLD_LEAF", 1, 0, 1},
    {"fft_fused", "CM_FFT_FUSED", 1, 0, 1},           {"commit_pipe", "CM_COMMIT_PIPE", 1, 0, 1},         {"fft_chunk_mb", "CM_FFT_CHUNK_MB", 0, 0, 4096},
    {"pace", "CM_PACE", -1, -1, 1},                   {"pace_early", "CM_PACE_EARLY", 1, 0, 1},           {"tail_flags", "CM_TAIL_FLAGS", 1, 0, 1},
    {"tail_grind_cap", "CM_TAIL_GRIND_CAP", 0, 0, 40},
    {"cons_wide_first", "CM_CONS_WIDE_FIRST", 1, 0, 1}, {"logup_small_stream", "CM_LOGUP_SMALL_STREAM", -1, -1, 7},
    {"cons_plan", "CM_CONS_PLAN", 01237456, 0, 077777777}, {"quot_leaf", "CM_QUOT_LEAF", 1, 0, 1},
    {"shard_fri_stream", "CM_SHARD_FRI_STREAM", 1, 0, 1}, {"fft_half_occ", "CM_FFT_HALF_OCC", 0, 0, 3},
    {"shard_halo", "CM_SHARD_HALO", 1, 0, 1},           {"tw_batch", "CM_TW_BATCH", 8, 2, 16},
    {"tree0_guest", "CM_TREE0_GUEST", 0, 0, 1},         {"merkle_multi_top", "CM_MERKLE_MULTI_TOP", 19, 17, 23},
    {"shard_tree_stream", "CM_SHARD_TREE_STREAM", 1, 0, 1}, {"shard_fri_stop_log", "CM_SHARD_FRI_STOP_LOG", 16, 0, 99},
};
std::atomic<int>* tune_values() {
  static std::atomic<int>* v = [] {
    std::atomic<int>* x = new std::atomic<int>[T_COUNT];
    for (int k = 0; k < T_COUNT; k++) {
      const char* e = getenv(TUNE_TABLE[k].env);
      x[k].store(e ? std::min(std::max((int)strtol(e, nullptr, 0), TUNE_TABLE[k].lo), TUNE_TABLE[k].hi) : TUNE_TABLE[k].dflt);   // (base 0: CM_CONS_PLAN=01237456 is octal)
    }
    return x;
  }();
  return v;
}
void stage_forget_stream(hipStream_t st) {
  // cm_stream_destroy on the owning thread: the handle value may be recycled for a new stream, whose copies must get their own
  // event history — wait for this stream's copies out of the ring and drop the entry
  Stage& s = stage();
  std::lock_guard<std::mutex> lk(s.mu);
  for (size_t i = 0; i < s.users.size(); i++)
    if (s.users[i].st == st) {
      if (s.users[i].dirty) (void)hipStreamSynchronize(st);
      (void)hipEventSynchronize(s.users[i].ev);
      s.spare.push_back(s.users[i].ev);
      s.users.erase(s.users.begin() + i);
      break;
    }
}

// Pinned landing buffer for device->host results (one per host thread, grown on demand): a hipMemcpyAsync into
// pageable memory blocks the caller until the stream has drained, so host work meant to overlap the producing
// kernels never overlapped.  The returned pointer is valid once `st` is synchronised, until the next call.
namespace {
struct Landing {
  uint8_t* base = nullptr;
  size_t cap = 0;
};
Landing& landing() {
  static thread_local Landing* l = nullptr;
  if (!l) {
    l = new Landing();
    Landing* own = l;
    at_thread_exit([own] { if (own->base) (void)hipHostFree(own->base); delete own; });
  }
  return *l;
}
}  // namespace
// the landing buffer itself (>= bytes), for callers that fill it with several copies
void* stage_landing(size_t bytes, hipStream_t st) { return const_cast<void*>(stage_download_async(nullptr, bytes ? bytes : 1, st)); }
const void* stage_download_async(const void* src, size_t bytes, hipStream_t st) {
  Landing& l = landing();
  if (bytes > l.cap) {
    if (l.base) { CM_HIP(hipStreamSynchronize(st)); CM_HIP(hipHostFree(l.base)); l.base = nullptr; }
    l.cap = std::max(bytes * 2, (size_t)1 << 20);
    CM_HIP(hipHostMalloc((void**)&l.base, l.cap, hipHostMallocDefault));
  }
  if (bytes && src) CM_HIP(hipMemcpyAsync(l.base, src, bytes, hipMemcpyDeviceToHost, st));
  return l.base;
}

// Pinned buffers of the device-side proof tail (tail_device.hpp), one pair per host thread, grown on demand: the kernels read
// the decommitment descriptors from the first and write the witnesses into the second (no copy command in the tail).
namespace {
struct TailPinned { uint8_t* base[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0}; };
void* tail_pinned(int which, size_t bytes) {
  static thread_local TailPinned* t = nullptr;
  if (!t) {
    t = new TailPinned();
    TailPinned* own = t;
    at_thread_exit([own] { for (int i = 0; i < 2; i++) if (own->base[i]) (void)hipHostFree(own->base[i]); delete own; });
  }
  if (bytes > t->cap[which]) {
    // (a proof ends with a synchronised stream: nothing of the previous proof still reads or writes the old buffer)
    if (t->base[which]) { CM_HIP(hipDeviceSynchronize()); CM_HIP(hipHostFree(t->base[which])); t->base[which] = nullptr; t->cap[which] = 0; }
    const size_t cap = std::max(bytes + bytes / 2, (size_t)1 << 16);
    CM_HIP(hipHostMalloc((void**)&t->base[which], cap, hipHostMallocDefault));
    t->cap[which] = cap;
  }
  return t->base[which];
}
}  // namespace
void* tail_pinned_desc(size_t bytes) { return tail_pinned(0, bytes); }
void* tail_pinned_out(size_t bytes) { return tail_pinned(1, bytes); }

// 1024 pinned words per host thread for the small fixed-slot results of a proof (flag, nonce, roots, claimed sums,
// FRI challenges): copies into pageable memory block the caller and cost 15-25 us each
uint32_t* pinned_words() {
  static thread_local uint32_t* p = nullptr;
  if (!p) {
    CM_HIP(hipHostMalloc((void**)&p, 4096, hipHostMallocDefault));
    uint32_t* own = p;
    at_thread_exit([own] { (void)hipHostFree(own); });
  }
  return p;
}

// ---- fork/join side streams ---------------------------------------------------------------------------
namespace {
// Join by flags (round 5).  An event join puts one barrier packet per joined side stream into the main stream's hardware queue
// and each costs ~5.5 us whether its event has fired or not: 42-52 us behind a region of seven side streams
// (tools/join_lab.hip, tools/flagjoin_lab.hip).  Instead every side stream ends with a one-thread kernel that stores the
// join's epoch into its flag word, and the main stream runs ONE collector kernel whose lanes poll the flags: 4.7-4.9 us for the
// same region.  Submission order makes it safe on shared hardware queues: the collector is enqueued after every flag kernel it
// waits for.  A lane gives up after JOIN_LIMIT_S seconds of device clock and reports it through a pinned word that the prover
// reads at the end of the proof (fork_join_check).  CM_FLAG_JOIN=0: the event form.
constexpr double JOIN_LIMIT_S = 10.0;
__global__ void k_join_flag(uint32_t* flag, uint32_t epoch) { __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_join_collect(const uint32_t* flags, uint32_t mask, uint32_t epoch, unsigned long long limit_ticks, uint32_t* timed_out) {
  const uint32_t i = threadIdx.x;
  if (i >= (uint32_t)Fork::N || !((mask >> i) & 1u)) return;
  const unsigned long long t0 = wall_clock64();
  // SYSTEM scope: the load must not be served from this XCD's L2 (the flag kernel ran on whichever XCD; its store reaches memory
  // when that kernel ends) — with agent scope on ordinary device memory the lanes polled a stale line until the time limit
  while ((int32_t)(__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
    if (wall_clock64() - t0 > limit_ticks) {   // {1, lane, epoch, flag value seen}: read by fork_join_check
      timed_out[1] = i; timed_out[2] = epoch; timed_out[3] = __hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      *timed_out = 1;
      return;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}
struct SideStreams {
  hipStream_t s[Fork::N];
  hipEvent_t done[Fork::N], fork_ev;
  // device: one join word per side stream, then (own 64-byte line) the fork word.  All of it is allocated and zeroed here:
  // a fork word holding a stale value >= the thread's fork epoch would release the side streams before the main stream
  // reaches the fork point (round-5 advice: it used to sit one word past a N-word allocation)
  static constexpr int FORK_SLOT = 16;                       // word index of the fork flag (Fork::N <= 16)
  static constexpr size_t FLAG_WORDS = FORK_SLOT + 16;
  static_assert(Fork::N <= FORK_SLOT, "join words and the fork word must not overlap");
  uint32_t* flags = nullptr;
  uint32_t* fork_flag() const { return flags + FORK_SLOT; }
  uint32_t* timed_out = nullptr;   // pinned
  uint32_t epoch = 0, fork_epoch = 0;
  unsigned long long limit_ticks = 0;
  SideStreams() {
    for (int i = 0; i < Fork::N; i++) {
      CM_HIP(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
      CM_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    CM_HIP(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    const int mem = getenv("CM_FLAG_MEM") ? atoi(getenv("CM_FLAG_MEM")) : 0;   // development: 0 device, 1 fine-grained device, 2 pinned host
    if (mem == 2) CM_HIP(hipHostMalloc((void**)&flags, FLAG_WORDS * 4, hipHostMallocDefault));
    else if (mem == 1) CM_HIP(hipExtMallocWithFlags((void**)&flags, FLAG_WORDS * 4, hipDeviceMallocFinegrained));
    else CM_HIP(hipMalloc((void**)&flags, FLAG_WORDS * 4));
    CM_HIP(hipMemset(flags, 0, FLAG_WORDS * 4));
    CM_HIP(hipDeviceSynchronize());   // (hipMemset runs on the NULL stream, the library's streams are non-blocking)
    CM_HIP(hipHostMalloc((void**)&timed_out, 64, hipHostMallocDefault));
    memset(timed_out, 0, 64);
    int dev = 0, rate_khz = 100000;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);
    const double limit_s = getenv("CM_JOIN_LIMIT_S") ? atof(getenv("CM_JOIN_LIMIT_S")) : JOIN_LIMIT_S;
    limit_ticks = (unsigned long long)(limit_s * 1e3 * (double)rate_khz);
  }
};
SideStreams& side() {
  static thread_local SideStreams* s = nullptr;
  if (!s) {
    s = new SideStreams();
    SideStreams* own = s;
    at_thread_exit([own] {
      for (int i = 0; i < Fork::N; i++) { (void)hipStreamSynchronize(own->s[i]); (void)hipStreamDestroy(own->s[i]); (void)hipEventDestroy(own->done[i]); }
      (void)hipEventDestroy(own->fork_ev);
      (void)hipFree(own->flags);
      (void)hipHostFree(own->timed_out);
      delete own;
    });
  }
  return *s;
}
}  // namespace

hipStream_t thread_side_stream(int i) { return side().s[((i % Fork::N) + Fork::N) % Fork::N]; }
// A stream of its own PRIORITY class: the runtime keeps separate hardware queues per priority, so this stream never shares a
// hardware queue with the main / side streams (eight normal-priority streams land on four queues, and two streams on one queue
// run strictly one after the other: the transforms of tree 1 once sat in front of the whole preprocessed tree that way).
// rel: -1 = the highest priority the device offers, +1 = the lowest.
hipStream_t thread_priority_stream(int rel) {
  static thread_local hipStream_t hi = nullptr, lo = nullptr;
  hipStream_t& s = rel < 0 ? hi : lo;
  if (!s) {
    int least = 0, greatest = 0;
    CM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));   // numerically: greatest <= 0 <= least
    CM_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, rel < 0 ? greatest : least));
    hipStream_t own = s;
    at_thread_exit([own] { (void)hipStreamSynchronize(own); (void)hipStreamDestroy(own); });
  }
  return s;
}
// Fork by flags (round 5), the mirror image of the flag join: a one-thread kernel on the main stream stores the fork's epoch, and
// every side stream starts with a one-wave collector that polls it — 3-8 us from the end of the main stream's kernel to the start
// of the side kernels instead of 10-30 us through an event and a barrier packet per side stream (tools/flagjoin_lab.hip).
// CM_FLAG_FORK=0: the event form.
// Flag synchronisation needs kernels of different streams to RUN CONCURRENTLY: a collector spins until a flag kernel on another
// stream has run.  Tools that serialise kernel execution — rocprofv3 with --pmc (counter collection runs one kernel at a time),
// some debuggers — turn that into a wait for the collector's time limit (the first PMC pass of round 5's measurement script sat in
// it until gpurun's limit).  One self-test per process decides: a collector with a 100 ms limit on one stream, THEN its flag kernel
// on another; where kernels overlap it returns in microseconds, under a serialising tool it times out once and every fork / join
// of the process uses events.
static bool flag_sync_usable() {
  static const bool ok = [] {
    hipStream_t a = nullptr, b = nullptr;
    uint32_t *flag = nullptr, *to = nullptr;
    bool good = false;
    if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&b, hipStreamNonBlocking) == hipSuccess &&
        hipMalloc((void**)&flag, 4) == hipSuccess && hipHostMalloc((void**)&to, 64, hipHostMallocDefault) == hipSuccess) {
      memset(to, 0, 64);
      int dev = 0, rate_khz = 100000;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);
      if (hipMemsetAsync(flag, 0, 4, a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess) {
        hipLaunchKernelGGL(k_join_collect, dim3(1), dim3(64), 0, a, flag, 1u, 1u, (unsigned long long)(0.1 * 1e3 * (double)rate_khz), to);
        hipLaunchKernelGGL(k_join_flag, dim3(1), dim3(1), 0, b, flag, 1u);
        good = hipStreamSynchronize(b) == hipSuccess && hipStreamSynchronize(a) == hipSuccess && to[0] == 0;
      }
    }
    if (flag) (void)hipFree(flag);
    if (to) (void)hipHostFree(to);
    if (a) (void)hipStreamDestroy(a);
    if (b) (void)hipStreamDestroy(b);
    if (!good && getenv("CM_QUIET") == nullptr)
      fprintf(stderr, "[cairom_hip] kernels of different streams do not run concurrently here (a profiler collecting counters?): "
                      "fork / join through events instead of flags\n");
    return good;
  }();
  return ok;
}
static bool flag_fork_on() {
  const bool on = tune(T_FLAG_FORK) != 0 && flag_sync_usable();
  return on;
}
Fork::Fork(hipStream_t main_stream) : main(main_stream) {
  SideStreams& ss = side();
  if (flag_fork_on()) {
    fork_epoch = ++ss.fork_epoch;
    hipLaunchKernelGGL(k_join_flag, dim3(1), dim3(1), 0, main, ss.fork_flag(), fork_epoch);
    CM_HIP(hipGetLastError());
  } else CM_HIP(hipEventRecord(ss.fork_ev, main));
}
int Fork::main_or(int side_index) {
  return tune(T_FORK_MAIN) != 0 ? MAIN : side_index;
}
hipStream_t Fork::stream(int i) {
  if (i == MAIN) return main;
  i = ((i % N) + N) % N;
  SideStreams& ss = side();
  if (!(used & (1u << i))) {
    if (flag_fork_on()) {
      hipLaunchKernelGGL(k_join_collect, dim3(1), dim3(64), 0, ss.s[i], ss.fork_flag(), 1u, fork_epoch, ss.limit_ticks, ss.timed_out);
      CM_HIP(hipGetLastError());
    } else CM_HIP(hipStreamWaitEvent(ss.s[i], ss.fork_ev, 0));
    used |= 1u << i;
  }
  return ss.s[i];
}
namespace { int g_library_device = -1; }
void set_library_device(int device) { g_library_device = device; }
void bind_thread_to_library_device() {
  static thread_local int bound = -1;
  if (g_library_device >= 0 && bound != g_library_device) {
    CM_HIP(hipSetDevice(g_library_device));   // hipSetDevice is per host thread
    bound = g_library_device;
  }
}
// ---- CPU placement of proving threads --------------------------------------------------------------------------------------
// A proof is ~450 launches and ~10 host round trips; on a two-socket host every doorbell write, pinned-memory read and
// completion signal of a thread running on the socket AWAY from the GPU crosses the inter-socket fabric (0.1-0.5 ms per proof
// between otherwise identical boxes).  Policy (cm_set_cpu_affinity / CM_CPU_AFFINITY):
//   0  never touch any thread's affinity;
//   1  (default) SCOPED: a caller thread is moved onto the GPU's local_cpulist for the duration of one cm_prove_* call and gets
//      its own mask back before the call returns — nothing leaks to the host application, to threads it creates later or to
//      child processes; the library's own cm_prove_many workers stay on the GPU's node for good.
// The mask is only ever narrowed inside what the caller (cgroup, taskset) allowed.
namespace {
std::atomic<int> g_affinity_mode{-1};
int affinity_mode() {
  int m = g_affinity_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("CM_CPU_AFFINITY");
    m = (getenv("CM_NO_CPU_AFFINITY") != nullptr) ? 0 : (e ? (atoi(e) == 2 ? 2 : atoi(e) != 0) : 1);
    g_affinity_mode.store(m);
  }
  return m;
}
// sysfs local_cpulist of the device's PCI function ("0-63,128-191"), parsed once per device under a lock (no strtok: the
// cm_prove_many workers arrive here together)
bool local_cpus_of_device(int device, cpu_set_t* set) {
  static std::mutex mu;
  static std::map<int, std::pair<bool, cpu_set_t>> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(device);
  if (it == cache.end()) {
    std::pair<bool, cpu_set_t> ent;
    ent.first = false;
    CPU_ZERO(&ent.second);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) == hipSuccess) {
      for (char* c = bus; *c; c++) *c = (char)tolower(*c);
      const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
      if (FILE* f = fopen(path.c_str(), "r")) {
        char line[4096] = {0};
        if (fgets(line, sizeof(line), f)) {
          int n = 0;
          const char* p = line;
          while (*p) {
            if (!isdigit((unsigned char)*p)) { p++; continue; }
            char* end = nullptr;
            long a = strtol(p, &end, 10), b = a;
            if (*end == '-') b = strtol(end + 1, &end, 10);
            for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &ent.second); n++; }
            p = end;
          }
          ent.first = n > 0;
        }
        fclose(f);
      }
    }
    it = cache.emplace(device, ent).first;
  }
  *set = it->second.second;
  return it->second.first;
}
bool narrow_to_device(cpu_set_t* saved) {
  cpu_set_t local, both;
  if (g_library_device < 0 || !local_cpus_of_device(g_library_device, &local)) return false;
  if (sched_getaffinity(0, sizeof(*saved), saved) != 0) return false;
  CPU_AND(&both, &local, saved);                       // never widen what the caller (cgroup, taskset) allowed
  if (CPU_COUNT(&both) == 0 || CPU_EQUAL(&both, saved)) return false;
  return sched_setaffinity(0, sizeof(both), &both) == 0;   // tid 0 = the calling thread
}
}  // namespace
void set_cpu_affinity_mode(int mode) { g_affinity_mode.store(mode == 2 ? 2 : (mode ? 1 : 0)); }
int cpu_affinity_mode() { return affinity_mode(); }
AffinityScope::AffinityScope() {
  const int m = affinity_mode();
  if (m == 2) {   // STICKY: narrowed once per thread, never restored (three syscalls per proof cost 30-50 us on the pool's hosts)
    static thread_local bool done = false;
    if (!done) { cpu_set_t dropped; (void)narrow_to_device(&dropped); done = true; }
    active = false;
    return;
  }
  active = m == 1 && narrow_to_device(&saved);
}
AffinityScope::~AffinityScope() { if (active) (void)sched_setaffinity(0, sizeof(saved), &saved); }
void bind_worker_thread_cpus() {
  cpu_set_t saved;
  if (affinity_mode() >= 1) (void)narrow_to_device(&saved);
}
hipStream_t thread_main_stream() {
  static thread_local hipStream_t s = nullptr;
  if (!s) {
    bind_thread_to_library_device();
    CM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t own = s;
    at_thread_exit([own] { (void)hipStreamSynchronize(own); (void)hipStreamDestroy(own); });
  }
  return s;
}
void Fork::join() {
  if (joined) return;
  joined = true;
  SideStreams& ss = side();
  const bool flag_join = tune(T_FLAG_JOIN) != 0 && flag_sync_usable();
  if (flag_join && used) {
    const uint32_t epoch = ++ss.epoch;
    for (int i = 0; i < N; i++)
      if (used & (1u << i)) hipLaunchKernelGGL(k_join_flag, dim3(1), dim3(1), 0, ss.s[i], ss.flags + i, epoch);
    hipLaunchKernelGGL(k_join_collect, dim3(1), dim3(64), 0, main, ss.flags, used, epoch, ss.limit_ticks, ss.timed_out);
    static const bool dbg = getenv("CM_FLAG_JOIN_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[join] epoch %u used %02x main %p\n", epoch, used, (void*)main);
    CM_HIP(hipGetLastError());
    return;
  }
  for (int i = 0; i < N; i++)
    if (used & (1u << i)) {
      CM_HIP(hipEventRecord(ss.done[i], ss.s[i]));
      CM_HIP(hipStreamWaitEvent(main, ss.done[i], 0));
    }
}
// a collector of this thread gave up (a side stream never reached its flag kernel): call with the main stream synchronised
void fork_join_check() {
  SideStreams& ss = side();
  if (*ss.timed_out) {
    const std::string what = "fork/join: a side stream did not reach its join flag (collector timed out: side " + std::to_string(ss.timed_out[1]) +
                             ", epoch " + std::to_string(ss.timed_out[2]) + ", flag " + std::to_string(ss.timed_out[3]) + ")";
    *ss.timed_out = 0;
    throw CmError(2, what);
  }
}
Fork::~Fork() {
  // never leave side work un-joined (exception paths): block the host instead of throwing from a destructor
  if (!joined) {
    SideStreams& ss = side();
    for (int i = 0; i < N; i++) if (used & (1u << i)) (void)hipStreamSynchronize(ss.s[i]);
  }
}

}  // namespace cm
