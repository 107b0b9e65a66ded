// Whole-segment HIP prover: `prove_cairo_m::<Blake2sMerkleChannel>`
// (/root/reference/crates/prover/src/prover.rs:23-147; transcript order is normative) driving the gfx950
// kernels.  Host code here only sequences launches, runs the Fiat-Shamir channel, and assembles the proof;
// every pass over trace data (trace gen, LogUp, IFFT/LDE, Merkle, constraints, OODS sampling, DEEP
// quotients, FRI folds, PoW search, decommit gathers) is a kernel.  No CPU fallback exists.
#include "../../include/cairom_hip.h"
#include "prover_common.hpp"
#include "air_kernels.hpp"
#include "gpu_air.hpp"
#include "point_eval.hpp"
#include "host_adapter.hpp"
#include "shard_kernels.hpp"
#include "tail_plan.hpp"
#include <atomic>
#include <chrono>
#include <deque>
#include <condition_variable>
#include <thread>
#include <memory>
#include <mutex>
#include <map>
#include <set>
#include <algorithm>
#include <functional>

namespace cm {

static inline uint32_t log_size_for(uint64_t n) {  // max(LOG_N_LANES, ceil_log2(n))
  uint32_t l = 4;
  while ((1ull << l) < n) l++;
  return l;
}

// ---- device-resident prover input ------------------------------------------------------------------------
struct DeviceInput {
  cm_prover_input meta;  // scalar fields + counts; pointers are replaced by device pointers below
  DevBuf bundles[CM_N_OPCODE_COMPONENTS], data_accesses, init_mem, fin_mem, clock_updates, init_tree, fin_tree;
  PublicData public_data;
};

static PublicData make_public_data(const cm_prover_input& in) {  // PublicData::new (public_data.rs:244-272)
  PublicData d;
  d.initial_pc = in.initial_pc; d.initial_fp = in.initial_fp; d.final_pc = in.final_pc; d.final_fp = in.final_fp;
  uint64_t steps = 0;
  for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) steps += in.n_bundles[i];
  d.clock = M31::reduce(steps).v;
  d.initial_root = in.initial_root; d.final_root = in.final_root;
  for (uint32_t w : {d.initial_pc, d.initial_fp, d.final_pc, d.final_fp, d.initial_root, d.final_root})
    CM_CHECK(w < P, "ProverInput: a public register / root word is not a canonical M31");
  std::map<uint32_t, const cm_memory_cell*> init, fin;
  for (uint64_t i = 0; i < in.n_initial_memory; i++) init[in.initial_memory[i].address] = &in.initial_memory[i];
  for (uint64_t i = 0; i < in.n_final_memory; i++) fin[in.final_memory[i].address] = &in.final_memory[i];
  auto extract = [](const std::map<uint32_t, const cm_memory_cell*>& m, const uint32_t range[2]) {
    std::vector<PublicEntry> v;
    for (uint32_t a = range[0]; a < range[1]; a++) {
      PublicEntry e{};
      auto it = m.find(a);
      if (it != m.end()) {
        e.present = 1; e.addr = a; for (int k = 0; k < 4; k++) e.value[k] = it->second->value[k]; e.clock = it->second->clock;
        CM_CHECK(a < P && e.clock < P && e.value[0] < P && e.value[1] < P && e.value[2] < P && e.value[3] < P,
                 "ProverInput: a public memory entry is not made of canonical M31 words");
      }
      v.push_back(e);
    }
    return v;
  };
  d.program = extract(init, in.program_range);
  d.input = extract(init, in.input_range);
  d.output = extract(fin, in.output_range);
  return d;
}

// `st` = null: blocking copies on the NULL stream + a device-wide synchronisation (the one-shot form).  Otherwise every copy is
// enqueued on `st` and only `st` is waited for — the streaming ingest (cm_prove_many_host) uploads segment k + 1 on the calling
// thread's stream while the workers' proofs of segments <= k keep the GPU busy; a device-wide wait would stall the uploader
// behind every proof in flight.
DeviceInput* upload_input(const cm_prover_input& in, hipStream_t st = nullptr) {
  bind_thread_to_library_device();
  std::unique_ptr<DeviceInput> dh(new DeviceInput());
  DeviceInput* d = dh.get();
  d->meta = in;
  auto up = [st](DevBuf& b, const void* p, size_t bytes) {
    b.alloc(bytes);
    if (!bytes) return;
    if (st) CM_HIP(hipMemcpyAsync(b.p, p, bytes, hipMemcpyHostToDevice, st));
    else CM_HIP(hipMemcpy(b.p, p, bytes, hipMemcpyHostToDevice));
  };
  for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) up(d->bundles[i], in.bundles[i], in.n_bundles[i] * sizeof(cm_bundle));
  up(d->data_accesses, in.data_accesses, in.n_data_accesses * sizeof(cm_data_access));
  up(d->init_mem, in.initial_memory, in.n_initial_memory * sizeof(cm_memory_cell));
  up(d->fin_mem, in.final_memory, in.n_final_memory * sizeof(cm_memory_cell));
  up(d->clock_updates, in.clock_updates, in.n_clock_updates * sizeof(cm_clock_update));
  up(d->init_tree, in.initial_tree, in.n_initial_tree * sizeof(cm_merkle_node));
  up(d->fin_tree, in.final_tree, in.n_final_tree * sizeof(cm_merkle_node));
  d->public_data = make_public_data(in);
  if (st) CM_HIP(hipStreamSynchronize(st));
  else CM_HIP(hipDeviceSynchronize());   // the copies ran on the NULL stream; the prover's streams are non-blocking
  return dh.release();
}

// device adapter (adapter_device.hip): bulk arrays already live in HBM, the small boundary-memory / Merkle-tree
// arrays come from the host (pointers in `meta`, valid during the call)
DeviceInput* make_device_input(const cm_prover_input& meta, DevBuf (&bundles)[CM_N_OPCODE_COMPONENTS], DevBuf& data_accesses,
                               DevBuf& clock_updates, DevBuf* init_tree_dev, DevBuf* fin_tree_dev) {
  DeviceInput* d = new DeviceInput();
  d->meta = meta;
  auto up = [](DevBuf& b, const void* p, size_t bytes) {
    b.alloc(bytes);
    if (bytes) CM_HIP(hipMemcpy(b.p, p, bytes, hipMemcpyHostToDevice));
  };
  for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) d->bundles[i] = std::move(bundles[i]);
  d->data_accesses = std::move(data_accesses);
  d->clock_updates = std::move(clock_updates);
  up(d->init_mem, meta.initial_memory, meta.n_initial_memory * sizeof(cm_memory_cell));
  up(d->fin_mem, meta.final_memory, meta.n_final_memory * sizeof(cm_memory_cell));
  if (init_tree_dev) d->init_tree = std::move(*init_tree_dev);   // trees hashed on the GPU by the device adapter
  else up(d->init_tree, meta.initial_tree, meta.n_initial_tree * sizeof(cm_merkle_node));
  if (fin_tree_dev) d->fin_tree = std::move(*fin_tree_dev);
  else up(d->fin_tree, meta.final_tree, meta.n_final_tree * sizeof(cm_merkle_node));
  d->public_data = make_public_data(meta);
  // the host pointers of `meta` die with the caller's temporaries
  for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) d->meta.bundles[i] = nullptr;
  d->meta.data_accesses = nullptr; d->meta.initial_memory = nullptr; d->meta.final_memory = nullptr;
  d->meta.clock_updates = nullptr; d->meta.initial_tree = nullptr; d->meta.final_tree = nullptr;
  CM_HIP(hipDeviceSynchronize());   // NULL-stream copies above; the prover's streams are non-blocking
  return d;
}
DeviceInput* adapt_segment_device(const cm_runner_segment& seg);  // adapter_device.hip
// copy a device-resident input back to the host (tests)
void download_input(const DeviceInput& d, host::ProverInputOwned& o) {
  CM_HIP(hipDeviceSynchronize());
  const cm_prover_input& m = d.meta;
  o.initial_pc = m.initial_pc; o.initial_fp = m.initial_fp; o.final_pc = m.final_pc; o.final_fp = m.final_fp;
  o.initial_root = m.initial_root; o.final_root = m.final_root;
  for (int i = 0; i < 2; i++) { o.program_range[i] = m.program_range[i]; o.input_range[i] = m.input_range[i]; o.output_range[i] = m.output_range[i]; }
  auto down = [](auto& vec, const DevBuf& b, uint64_t n) {
    vec.resize(n);
    if (n) CM_HIP(hipMemcpy(vec.data(), b.p, n * sizeof(vec[0]), hipMemcpyDeviceToHost));
  };
  o.n_steps = 0;
  for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) { down(o.bundles[i], d.bundles[i], m.n_bundles[i]); o.n_steps += m.n_bundles[i]; }
  down(o.data_accesses, d.data_accesses, m.n_data_accesses);
  down(o.initial_memory, d.init_mem, m.n_initial_memory);
  down(o.final_memory, d.fin_mem, m.n_final_memory);
  down(o.clock_updates, d.clock_updates, m.n_clock_updates);
  down(o.initial_tree, d.init_tree, m.n_initial_tree);
  down(o.final_tree, d.fin_tree, m.n_final_tree);
}

// Optional cache of tree 0.  The preprocessed columns are constants (preprocessed/mod.rs:75-82), so their
// coefficients, LDE and Merkle tree are the same for every proof of one PCS config.  OFF by default — a proof then
// recomputes them like the reference does (prover.rs:70-73) and bench.py times that; cm_set_preprocessed_cache(1) or
// CM_PREPROCESSED_CACHE=1 keeps the committed tree per host thread (it lives in that thread's device pool).
static std::atomic<int> g_pp_cache{-1};
static bool pp_cache_enabled() {
  int v = g_pp_cache.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("CM_PREPROCESSED_CACHE");
    v = (e && *e && *e != '0') ? 1 : 0;
    g_pp_cache.store(v, std::memory_order_relaxed);
  }
  return v == 1;
}
struct PreprocessedCache {
  bool valid = false;
  uint32_t log_blowup = 0;
  ColumnSet evals;     // the tables on their trace domains: the LogUp pass of rc8/16/20 and bitwise reads them
  CommittedTree tree;
};
static thread_local PreprocessedCache tl_pp_cache;


// CM_NO_SMALL_BATCH=1: one launch per small component again (A/B of the batched small-component kernels)
static bool no_small_batch() { static const bool v = getenv("CM_NO_SMALL_BATCH") != nullptr; return v; }

// Twiddle tables.  The reference recomputes them in every prove_cairo_m (`SimdBackend::precompute_twiddles`, prover.rs:56-60),
// so by default every proof builds its own tables (in pool memory, on a side stream next to trace generation); they
// depend only on the domain size, and cm_set_twiddle_cache(1) / CM_TWIDDLE_CACHE=1 keeps one table per size for the
// process instead (OFF in every quoted number, like the preprocessed-tree cache).
static std::atomic<int> g_tw_cache{-1};
static bool tw_cache_enabled() {
  int v = g_tw_cache.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("CM_TWIDDLE_CACHE");
    v = (e && *e && *e != '0') ? 1 : 0;
    g_tw_cache.store(v, std::memory_order_relaxed);
  }
  return v == 1;
}
static Twiddles* cached_twiddles(uint32_t R, hipStream_t st) {
  static std::mutex mu;
  static std::map<uint32_t, Twiddles*> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(R);
  if (it != cache.end()) return it->second;
  Twiddles* t = twiddles_create(R, st);
  CM_HIP(hipStreamSynchronize(st));  // other host threads use the tables from their own streams
  cache[R] = t;
  return t;
}
struct ProofTwiddles {   // per-proof tables in pool memory
  Twiddles t;
  DevBuf x, ix, y, iy, scratch;
  void build(uint32_t R, hipStream_t s) {
    t.R = R;
    x.alloc((size_t)4 << (R - 1)); ix.alloc((size_t)4 << (R - 1)); y.alloc((size_t)4 << R); iy.alloc((size_t)4 << R);
    scratch.alloc(twiddles_scratch_words(R) * 4);
    t.xtw = x.u32(); t.ixtw = ix.u32(); t.ytw = y.u32(); t.iytw = iy.u32(); t.scratch = scratch.u32();
    twiddles_build(t, s);
  }
};

// coset_vanishing of CanonicCoset(log).coset at p (QM31 or M31 point)
template <class F>
static F coset_vanishing_canonic(uint32_t log, CPoint<F> p) {
  // shift = -initial + step/2 = 0 for a canonic (odds) coset: initial = G_{2^(log+1)} = step/2
  F x = p.x;
  for (uint32_t i = 1; i < log; i++) x = double_x(x);
  return x;
}

// log2 rows of every component, known from the input lengths (Claim::log_sizes)
static void component_logs(const cm_prover_input& in, uint32_t* clog) {
  uint64_t nrows[air::N_COMPONENTS] = {0};
  for (int c = 0; c < air::N_OPCODE_COMPONENTS; c++) nrows[c] = in.n_bundles[c];
  nrows[air::C_MEMORY] = in.n_initial_memory + in.n_final_memory;
  nrows[air::C_MERKLE] = in.n_initial_tree + in.n_final_tree;
  nrows[air::C_CLOCK_UPDATE] = in.n_clock_updates;
  nrows[air::C_POSEIDON2] = in.n_initial_tree + in.n_final_tree;
  for (int c = 0; c <= air::C_POSEIDON2; c++) clog[c] = log_size_for(nrows[c]);
  clog[air::C_RC8] = 8; clog[air::C_RC16] = 16; clog[air::C_RC20] = 20; clog[air::C_BITWISE] = 18;
}
// ---- transcript steps shared by the single-GPU and the sharded prover ------------------------------------------------------
// PcsConfig::mix_into + PublicData::mix_into (prover.rs:33-36, 62-66)
static void mix_config_and_public_data(Channel& ch, const cm_pcs_config& cfg, const PublicData& d) {
  // PcsConfig::mix_into + FriConfig::mix_into (prover.rs:36); the order inside FriConfig is a framing switch (framing.hpp `pcs_mix`)
  ch.mix_u64(cfg.pow_bits);
  ch.mix_u64(cfg.log_blowup_factor);
  if (framing().pcs_mix_blq) { ch.mix_u64(cfg.log_last_layer_degree_bound); ch.mix_u64(cfg.n_queries); }
  else { ch.mix_u64(cfg.n_queries); ch.mix_u64(cfg.log_last_layer_degree_bound); }
  uint32_t w[7] = {d.initial_pc, d.initial_fp, d.final_pc, d.final_fp, d.clock, d.initial_root, d.final_root};
  ch.mix_u32s(w, 7);
  uint32_t lens[3] = {(uint32_t)d.program.size(), (uint32_t)d.input.size(), (uint32_t)d.output.size()};
  ch.mix_u32s(lens, 3);
  for (const auto* v : {&d.program, &d.input, &d.output}) {
    std::vector<uint32_t> words;
    for (auto& e : *v) if (e.present) { words.push_back(e.addr); for (int k = 0; k < 4; k++) words.push_back(e.value[k]); words.push_back(e.clock); }
    ch.mix_u32s(words.data(), words.size());
  }
}
// Relations::draw (prover.rs:94): (z, alpha) per relation, alpha powers for the device
static void draw_relations(Channel& ch, HostRelations& hrel, DevRelations& drel_h) {
  for (int r = 0; r < air::N_RELATIONS; r++) {
    QM31 z, alpha;
    ch.draw_two_felts(z, alpha);
    hrel.z[r] = z;
    QM31 cur(M31(1));
    for (int i = 0; i < air::MAX_REL_SIZE; i++) { hrel.alpha_pow[r][i] = cur; cur = cur * alpha; }
    z.to_u32(drel_h.z[r]);
    for (int i = 0; i < air::MAX_REL_SIZE; i++) hrel.alpha_pow[r][i].to_u32(drel_h.alpha_pow[r][i]);
  }
}
// random coefficient of the composition polynomial: powers[g] = rho^(total - 1 - g), uploaded as words
static void draw_constraint_powers(Channel& ch, std::vector<QM31>& powers, DevBuf& d_powers, hipStream_t st) {
  QM31 random_coeff = ch.draw_felt();
  QM31 cur(M31(1));
  const size_t total = powers.size();
  for (size_t g = total; g-- > 0;) { powers[g] = cur; cur = cur * random_coeff; }
  std::vector<uint32_t> powers_w(4 * total);
  for (size_t g = 0; g < total; g++) powers[g].to_u32(&powers_w[4 * g]);
  stage_upload(d_powers.p, powers_w.data(), powers_w.size() * 4, st);
}

// Sanity check of stwo `prove`: the composition polynomial's OODS value equals the constraints evaluated on the sampled mask
// values (host-only work; both provers run it while the GPU is busy with the quotient / FRI kernels).
static void check_composition_at_oods(const ProofData& pf, const std::vector<size_t>& tr0, const std::vector<size_t>& it0, const uint32_t* clog,
                                      const HostRelations& hrel, const std::vector<QM31>& powers, const std::vector<size_t>& coff,
                                      const CPoint<QM31>& oods) {
  QM31 c4[4] = {pf.sampled_values[3][0][0], pf.sampled_values[3][1][0], pf.sampled_values[3][2][0], pf.sampled_values[3][3][0]};
  QM31 comp = combine_ef(c4);
  QM31 ppv[air::N_PREPROC];
  for (int i = 0; i < air::N_PREPROC; i++) ppv[i] = pf.sampled_values[0][i][0];
  QM31 sum;
  for (int c = 0; c < air::N_COMPONENTS; c++) {
    const air::ComponentInfo& info = air::component_info(c);
    std::vector<QM31> tr, it;
    for (int k = 0; k < info.n_trace; k++) tr.push_back(pf.sampled_values[1][tr0[c] + k][0]);
    for (int k = 0; k < info.n_interaction; k++) for (auto& s : pf.sampled_values[2][it0[c] + k]) it.push_back(s);
    QM31 shift = pf.claimed_sums[c] * inv(M31::from_u32(1u << clog[c]));
    QM31 num = point_eval(c, tr.data(), it.data(), ppv, hrel, &powers[coff[c]], info.n_base_constraints, shift);
    sum += num * inv(coset_vanishing_canonic<QM31>(clog[c], oods));
  }
  if (sum != comp) throw CmError(10, "ConstraintsNotSatisfied: composition polynomial does not match the constraints at the OODS point");
}
// the OODS point from one drawn felt t: ((1 - t^2) / (1 + t^2), 2t / (1 + t^2))
static CPoint<QM31> draw_oods_point(Channel& ch, QM31* t_out = nullptr) {
  QM31 t = ch.draw_felt();
  if (t_out) *t_out = t;
  QM31 t2 = t * t;
  QM31 iv = inv(t2 + M31(1));
  CPoint<QM31> p;
  p.x = (QM31(M31(1)) - t2) * iv;
  p.y = (t + t) * iv;
  return p;
}
// up to three device ranges zeroed by ONE launch (the constraints phase clears two accumulator sets and the slot buffer right in
// front of its kernels: three dependent hipMemsetAsync = three packets on the critical path behind the interaction tree)
struct ZeroRanges { uint4* p[3]; uint64_t n16[3]; };
__global__ void __launch_bounds__(256) k_zero_ranges(ZeroRanges z) {
  for (int r = 0; r < 3; r++)
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < z.n16[r]; i += (uint64_t)gridDim.x * 256) z.p[r][i] = make_uint4(0, 0, 0, 0);
}
static void zero_ranges(void* const p[3], const size_t bytes[3], hipStream_t st) {
  ZeroRanges z;
  uint64_t total = 0;
  for (int r = 0; r < 3; r++) {
    CM_CHECK(((uintptr_t)p[r] & 15) == 0 && (bytes[r] & 15) == 0, "zero_ranges: ranges must be 16-byte aligned");
    z.p[r] = (uint4*)p[r]; z.n16[r] = bytes[r] / 16; total += z.n16[r];
  }
  if (!total) return;
  const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(k_zero_ranges, dim3(blocks), dim3(256), 0, st, z);
  CM_HIP(hipGetLastError());
}
// side streams a fork region spreads its large launches over (A/B: CM_FORK_WIDTH; the join costs one barrier packet per used stream)
static int fork_width(int dflt) {
  const int w = tune(T_FORK_WIDTH);
  return w > 0 && w < dflt ? w : dflt;
}
// =========================================================================================================
// CM_HOST_TRACE=1: host-side time between marks on stderr (where the GPU sits idle waiting for the host)
// (the split of the sampled values, per mille in the chunk evaluated and hashed first: tuning key "oods_split", 780)
struct HostTrace {
  bool on = getenv("CM_HOST_TRACE") != nullptr || getenv("CM_HOST_MARKS") != nullptr;   // MARKS: no synchronising ticks
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[host] %-40s %8.1f us\n", what, std::chrono::duration<double, std::micro>(n - t).count());
    t = n;
  }
};
// One segment proof on one GPU: the phases of prove_cairo_m (prover.rs:23-147) in transcript order.  Each phase enqueues its
// kernels and returns at the next point where the transcript needs a device result; what crosses a phase boundary is a member.
struct SegmentProver {
  const DeviceInput& din;
  const cm_prover_input& in;
  const cm_pcs_config cfg;
  std::unique_ptr<ProofData> out;
  ProofData& pf;
  HostTrace ht;
  struct DtorMark { HostTrace* ht; const char* what; ~DtorMark() { ht->mark(what); } };   // (CM_HOST_MARKS: where the teardown goes)
  DtorMark dm_p{&ht, "~ everything else (P: trees, events)"};
  Prover P;
  DtorMark dm_after_p{&ht, "~ twiddles .. step buffers"};
  hipStream_t st;
  Channel& ch;
  uint32_t clog[air::N_COMPONENTS];          // log2 rows of every component
  uint32_t max_log = 0, comp_log = 0;        // largest component; composition polynomial = max_log + 1
  std::vector<int> by_size, by_size_all;     // launch orders: opcode components / all components by descending size
  ProofTwiddles own_tw;
  std::unique_ptr<Fork> tw_fork;
  ColumnSet pp_evals, tr_evals;              // preprocessed / execution-trace evaluations (trace domain)
  std::vector<size_t> tr0, it0;              // first column of every component in trees 1 / 2
  HostRelations hrel;
  DevBuf drel;                               // DevRelations
  std::vector<size_t> coff;                  // first constraint of every component
  std::vector<QM31> powers;                  // random-coefficient powers, one per constraint
  DevBuf d_powers;
  DevBuf d_step1;                            // device-side transcript step after tree 1: {root[8], nonce[2], n_sent, error, z0[4]}
  DevBuf d_step2;                            // device-side transcript step after tree 2: {channel[16], coefficient[4], root[8]}
  DevBuf d_ctab;                             // composition: every small table of the constraints phase, ONE upload
  CPoint<QM31> oods;
  DtorMark dm_q{&ht, "~ oods buffers, quotients"};
  DevBuf d_oods_table, d_oods_out, d_qblob;  // sampling pointer table, sampled values, DEEP-quotient plan
  size_t o_qjobs = 0, n_qjobs = 0;
  std::vector<uint32_t> q_logs;
  std::vector<ColumnSet> quotients;
  DtorMark dm_fri{&ht, "~ fri"};
  FriPhase fri;
  DtorMark dm_rest{&ht, "~ ojobs, qg, affinity"};
  struct ORef { int t; uint32_t c; bool prev; };
  struct OJob { uint32_t log; bool prev; CPoint<QM31> pt; std::vector<ORef> refs; size_t off = 0, out_off = 0; int chunk = 0; };
  std::vector<OJob> ojobs;                    // sampling jobs, built (and their pointer table uploaded) by oods_prepare()
  size_t n_oods_out = 0;
  bool oods_poll = false;                    // the host watches the landing words of the sampled values instead of waiting for an event
  static void cpu_relax() { __builtin_ia32_pause(); }
  size_t n_oods_out0 = 0, n_flat0 = 0;       // sampled values / flat samples of chunk 0 (the part hashed while chunk 1 is evaluated)
  struct QRef { int t; uint32_t c; };
  struct QEntry { uint32_t col; uint32_t sidx; };   // column of the group, index of its sampled value in d_oods_out
  struct QBatch { CPoint<QM31> pt; std::vector<QEntry> entries; };
  struct QGroup { uint32_t log = 0; std::vector<const uint32_t*> cols; std::vector<QBatch> batches; ColumnSet out;
                  size_t o_cols = 0, o_out = 0, o_ci = 0, o_cc = 0, o_qb = 0, o_sidx = 0, o_ep = 0; };
  std::vector<QGroup> qg;                    // DEEP-quotient size groups
  AffinityScope cpu_scope;   // the calling thread sits next to the GPU for this proof only (pool.hip)
  SegmentProver(const DeviceInput& din_, const cm_pcs_config& cfg_)
      : din(din_), in(din_.meta), cfg(cfg_), out(new ProofData()), pf(*out), st(nullptr), ch(P.ch) {
    pf.config = cfg;
    bind_thread_to_library_device();
    if (g_transcript_log.load()) P.ch.log.p = &pf.transcript;
    P.cfg = cfg;
    P.st = thread_main_stream();
    st = P.st;
    P.start();
  }
  ProofData* run() {
    setup();
    trace_commit();
    interaction();
    composition();
    oods_sampling();
    deep_quotients();
    fri_and_pow();
    decommit();

    kprof_close_run();   // a run of timed launches still open on this thread
    ht.mark("finish: entered");
    P.finish();
    ht.mark("finish: phase events read");
    fork_join_check();
    pf.phase_ms = P.phase_ms;
    pf.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - P.t0).count();
    pf.steps = 0;
    for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) pf.steps += in.n_bundles[i];
    if (pp_cache_enabled()) {
      tl_pp_cache.tree = std::move(P.trees[0]);
      tl_pp_cache.evals = std::move(pp_evals);
      tl_pp_cache.log_blowup = cfg.log_blowup_factor;
      tl_pp_cache.valid = true;
    } else if (tl_pp_cache.valid) {
      tl_pp_cache = PreprocessedCache();  // switched off: give the buffers back to the pool
    }
    // Teardown that can wait: the FRI phase alone owns ~500 pool blocks (20 layers x (4 columns + a tree of ~20 layers)) and
    // giving them back took 80-100 us with the GPU idle between two proofs.  They are parked here and released by this thread's
    // NEXT proof while it waits for tree 1 (or when the thread ends); the next FRI phase finds them back in the pool.
    if (defer_teardown()) {
      std::unique_ptr<Parked>& g = parked();
      g.reset(new Parked());
      g->fri = std::move(fri);
      g->qg = std::move(qg);
      g->ojobs = std::move(ojobs);
      g->quotients = std::move(quotients);
    }
    ht.mark("finish: done");
    return out.release();
  }
  struct Parked { FriPhase fri; std::vector<QGroup> qg; std::vector<OJob> ojobs; std::vector<ColumnSet> quotients; };
  static std::unique_ptr<Parked>& parked() {
    static thread_local std::unique_ptr<Parked> g;
    static thread_local bool hooked = false;
    if (!hooked) {   // cm_pool_trim / cm_shutdown / the pool's out-of-memory retry / prove_sharded release it too (pool.hip)
      hooked = true;
      set_thread_parked_release([] { parked().reset(); });
    }
    return g;
  }
  static bool defer_teardown() { return tune(T_DEFER_TEARDOWN) != 0; }

  // component sizes, twiddles (side stream), transcript setup (prover.rs:33-66)
  void setup() {
    // ---- component log sizes (known from the input lengths) ----
    component_logs(in, clog);
    max_log = 0;
    for (int c = 0; c < air::N_COMPONENTS; c++) max_log = std::max(max_log, clog[c]);
    for (int c = 0; c < air::N_COMPONENTS; c++) CM_CHECK(clog[c] <= 26, "component too large");
    // launch orders: components by descending size (stable)
    for (int c = 0; c < air::N_COMPONENTS; c++) { by_size_all.push_back(c); if (c < air::N_OPCODE_COMPONENTS) by_size.push_back(c); }
    auto bigger = [&](int x, int y) { return clog[x] > clog[y]; };
    std::stable_sort(by_size.begin(), by_size.end(), bigger);
    std::stable_sort(by_size_all.begin(), by_size_all.end(), bigger);
    comp_log = max_log + 1;
    CM_CHECK(cfg.n_queries >= 1 && cfg.n_queries <= 4096, "PcsConfig: n_queries must be in 1..4096");
    CM_CHECK(cfg.pow_bits <= 64, "PcsConfig: pow_bits must be at most 64");
    CM_CHECK(cfg.log_last_layer_degree_bound <= max_log, "PcsConfig: log_last_layer_degree_bound exceeds the largest trace column");
    if (tw_cache_enabled()) P.tw = cached_twiddles(comp_log + cfg.log_blowup_factor, st);
    else {
      tw_fork.reset(new Fork(st));
      own_tw.build(comp_log + cfg.log_blowup_factor, tw_fork->stream(Fork::N - 1));   // joined before the first transform
      P.tw = &own_tw.t;
    }

    // ---- transcript setup (prover.rs:33-36, 62-66) ----
    pf.public_data = din.public_data;
    mix_config_and_public_data(ch, cfg, pf.public_data);
    ht.mark("setup");
    P.tick("setup");

  }

  // tree 0 (preprocessed, on a side stream) and tree 1: execution trace of every component + commitment (prover.rs:70-82)
  void trace_commit() {
    // ---- tree 0: preprocessed trace (prover.rs:70-73) ----
    std::unique_ptr<Fork> pp_fork;
    hipStream_t tree0_stream = nullptr;
    struct DrainOnExit {   // an exception between the fork and the join must not return tree 0's buffers to the pool under its kernels
      hipStream_t& s; bool joined = false;
      ~DrainOnExit() { if (s && !joined) (void)hipStreamSynchronize(s); }
    } tree0_guard{tree0_stream};
    bool build_tree0 = false;
    if (pp_cache_enabled() && tl_pp_cache.valid && tl_pp_cache.log_blowup == cfg.log_blowup_factor) {
      P.trees[0] = std::move(tl_pp_cache.tree);  // both handed back at the end of the proof
      pp_evals = std::move(tl_pp_cache.evals);
      tl_pp_cache.valid = false;
      ch.mix_root(P.trees[0].root);
    } else {
      std::vector<uint32_t> logs(air::PREPROC_LOG, air::PREPROC_LOG + air::N_PREPROC);
      pp_evals.alloc(logs, st);
      // The constant columns are generated on the side stream that builds the twiddles and later commits tree 0 (the same HIP
      // stream: ordered behind one, in front of the other): nothing before the LogUp phase reads them on the main stream, which
      // starts trace generation ~150 us earlier this way.  cm_set_tuning("pp_side", 0) / CM_PP_SIDE=0: on the main stream (A/B).
      const bool pp_side = tune(T_PP_SIDE) != 0;
      hipStream_t pps = st;
      if (pp_side) {
        if (!tw_fork) tw_fork.reset(new Fork(st));   // (twiddle cache on: nothing else is on that stream yet)
        pps = tw_fork->stream(Fork::N - 1);
      }
      launch_preproc_all(pp_evals.ptrs.data(), pps);
      build_tree0 = true;   // enqueued on a side stream right after the trace-generation launches (below)
    }
    ht.mark("preprocessed enqueued");
    P.tick("preprocessed");
    ht.mark("preprocessed gpu done");

    // ---- tree 1: execution trace (prover.rs:77-82; Claim::write_trace) ----
    tr0.assign(air::N_COMPONENTS, 0);
    it0.assign(air::N_COMPONENTS, 0);
    {
      std::vector<uint32_t> logs;
      for (int c = 0; c < air::N_COMPONENTS; c++) {
        tr0[c] = logs.size();
        for (int k = 0; k < air::component_info(c).n_trace; k++) logs.push_back(clog[c]);
      }
      tr_evals.alloc(logs, st);
    }
    DevBuf flag(4);   // range-check / bitwise lookup out of range: read back with the tree-1 root (no round trip of its own)
    uint32_t* flag_host = pinned_words();
    {
      // multiplicity columns = histograms over every lookup of every opcode component (components/mod.rs:139-160)
      CM_HIP(hipMemsetAsync(flag.p, 0, 4, st));
      HistPtrs h;
      h.rc8 = tr_evals.ptrs[tr0[air::C_RC8]]; h.rc16 = tr_evals.ptrs[tr0[air::C_RC16]];
      h.rc20 = tr_evals.ptrs[tr0[air::C_RC20]]; h.bitwise = tr_evals.ptrs[tr0[air::C_BITWISE]];
      h.error_flag = flag.u32();
      if (h.rc16 == h.rc8 + (1u << 8) && h.rc20 == h.rc16 + (1u << 16) && h.bitwise == h.rc20 + (1u << 20)) {
        // the four multiplicity columns are adjacent in the trace arena: one memset
        CM_HIP(hipMemsetAsync(h.rc8, 0, 4 * ((size_t)(1u << 8) + (1u << 16) + (1u << 20) + (1u << 18)), st));
      } else {
        CM_HIP(hipMemsetAsync(h.rc8, 0, 4u << 8, st));
        CM_HIP(hipMemsetAsync(h.rc16, 0, 4u << 16, st));
        CM_HIP(hipMemsetAsync(h.rc20, 0, 4u << 20, st));
        CM_HIP(hipMemsetAsync(h.bitwise, 0, 4u << 18, st));
      }
      // small opcode components (idle ones are 16 padding rows): trace + histogram of all of them in ONE launch
      std::vector<SmallTraceJob> small_trace;
      for (int c = 0; c < air::N_OPCODE_COMPONENTS; c++)
        if (clog[c] <= SMALL_COMPONENT_MAX_LOG && !no_small_batch())
          small_trace.push_back(SmallTraceJob{din.bundles[c].p, (uint32_t)in.n_bundles[c], tr_evals.dev(tr0[c]), clog[c], c});
      DevBuf d_small_trace = upload(small_trace, st);
      // components are independent: fork over side streams (trace then histogram of one component stay ordered)
      KProfRegion kreg("k_trace_gen(region)", st);
      Fork fk(st);
      // large components first: their kernels keep the GPU busy while the host issues the ~40 launches of the idle
      // ones (the host launch rate, not the GPU, bounds this region otherwise); histogram adds commute
      // the batched small components first: ~100 us of pure latency on a handful of CUs, hidden under the large kernels
      // (launched last it ran alone at the end of the region)
      launch_trace_hist_small(d_small_trace.as<SmallTraceJob>(), (uint32_t)small_trace.size(), din.data_accesses.p, h, fk.stream(Fork::N - 2));
      int spos = 0;
      for (int pos = 0; pos < air::N_OPCODE_COMPONENTS; pos++) {
        const int c = by_size[pos];
        if (clog[c] <= SMALL_COMPONENT_MAX_LOG && !no_small_batch()) continue;
        hipStream_t sc = fk.stream(spos == 0 ? Fork::main_or(0) : spos % fork_width(Fork::N - 2));   // by_size: the first one is the largest
        spos++;
        const bool fuse_th = tune(T_TRACE_HIST_FUSE) != 0;   // A/B
        if (fuse_th) launch_opcode_trace_hist(c, din.bundles[c].p, (uint32_t)in.n_bundles[c], din.data_accesses.p, clog[c], tr_evals.dev(tr0[c]), h, sc);
        else {
          launch_opcode_trace(c, din.bundles[c].p, (uint32_t)in.n_bundles[c], din.data_accesses.p, clog[c], tr_evals.dev(tr0[c]), sc);
          launch_hist(c, (const uint32_t* const*)tr_evals.dev(tr0[c]), clog[c], h, sc);
        }
      }
      launch_memory_trace(din.init_mem.p, (uint32_t)in.n_initial_memory, din.fin_mem.p, (uint32_t)in.n_final_memory, in.initial_root,
                          in.final_root, clog[air::C_MEMORY], tr_evals.dev(tr0[air::C_MEMORY]), fk.stream(air::C_MEMORY));
      launch_merkle_trace(din.init_tree.p, (uint32_t)in.n_initial_tree, din.fin_tree.p, (uint32_t)in.n_final_tree, in.initial_root,
                          in.final_root, clog[air::C_MERKLE], tr_evals.dev(tr0[air::C_MERKLE]), fk.stream(air::C_MERKLE));
      launch_clock_update_trace(din.clock_updates.p, (uint32_t)in.n_clock_updates, clog[air::C_CLOCK_UPDATE],
                                tr_evals.dev(tr0[air::C_CLOCK_UPDATE]), fk.stream(air::C_CLOCK_UPDATE));
      launch_poseidon2_trace(din.init_tree.p, (uint32_t)in.n_initial_tree, din.fin_tree.p, (uint32_t)in.n_final_tree,
                             clog[air::C_POSEIDON2], tr_evals.dev(tr0[air::C_POSEIDON2]), fk.stream(air::C_POSEIDON2));
      fk.join();
      kreg.close();
      ht.mark("trace_gen: launched + joined");
      flag_host[0] = 0xffffffffu;
      CM_HIP(hipMemcpyAsync(flag_host, flag.p, 4, hipMemcpyDeviceToHost, st));
    }
    if (tw_fork) tw_fork->join();   // the proof's twiddle tables are complete from here on
    if (build_tree0) {
      // tree 0 (a chain of ~30 small launches) is built on a side stream while the transforms and hashes of tree 1 keep
      // the GPU busy; its root comes back together with the root of tree 1.  Forked from the stream position after the
      // trace-generation launches so that its chain does not queue in front of them.
      // The side streams share four hardware queues with the main and the pipeline stream, and a queue runs its packets in order:
      // on side stream 7 the whole tree sat BEHIND the transforms of tree 1 (same queue as the pipeline stream) and its Merkle
      // chain ran alone after tree 1 had finished (0.17 ms of a nearly idle GPU, and the host learnt root 0 only then).  A stream
      // of the highest priority class has hardware queues of its own.  CM_TREE0_PRIO=0: the side stream (A/B), 1: lowest class.
      // A lone proof only: with several proofs in flight (cm_prove_many) the other proofs' kernels fill the queues anyway, and
      // four high-priority chains cutting into them cost 0.5-1 ms per proof (9.6 -> 10.1-10.8 with four in flight).
      const int t0_prio = tune(T_TREE0_PRIO);
      if (t0_prio != 0 && g_proofs_in_flight.load(std::memory_order_relaxed) <= 1) {
        tree0_stream = thread_priority_stream(t0_prio);
        hipEvent_t e = Prover::pipe_event();
        CM_HIP(hipEventRecord(e, st));                       // trace generation launched, twiddles joined
        CM_HIP(hipStreamWaitEvent(tree0_stream, e, 0));
      } else {
        pp_fork.reset(new Fork(st));
        tree0_stream = pp_fork->stream(Fork::N - 1);   // the side stream waits for THIS point of the main stream (trace generation done)
      }
    }
    P.tick("trace_gen");
    // tree 1 first: its large transforms start as soon as the trace exists and keep the GPU busy while the host issues tree 0's
    // chain of small launches (enqueued first, that chain delayed the first tree-1 kernel by the host time of ~30 launches)
    const bool tree1_first = tune(T_TREE1_FIRST) != 0;   // A/B switch
    if (build_tree0 && !tree1_first && !(tune(T_TREE0_GUEST) != 0 && !pp_cache_enabled())) P.commit_enqueue(P.trees[0], &pp_evals, false, tree0_stream);
    P.trees[1].merkle.pace_ev = Prover::pace_event(1);
    // (round 6) "tree0_guest": the seven preprocessed columns ride in tree 1's size-group launches (Prover::GuestTree) and only tree 0's
    // Merkle launches run on its stream.  Not with the preprocessed cache on: a cached tree must own its device tables.
    const bool tree0_guest = build_tree0 && tune(T_TREE0_GUEST) != 0 && !pp_cache_enabled();
    {
      Prover::GuestTree guest{&P.trees[0], &pp_evals, tree0_stream};
      Prover::CommitPrep cp1 = P.commit_prepare(P.trees[1], &tr_evals, false, st, true, false, P.pipe_stream(), nullptr, tree0_guest ? &guest : nullptr);
      ht.mark("trace_commit: tree 1 prepared");
      P.commit_launch(cp1);
      ht.mark("trace_commit: tree 1 launched");
    }
    if (build_tree0 && tree1_first && !tree0_guest) P.commit_enqueue(P.trees[0], &pp_evals, false, tree0_stream);
    ht.mark("trace_commit: tree 0 enqueued");
    // Root 0 first (transcript order, prover.rs:70-82: root 0, claim, root 1): tree 0 has been running on its side stream next
    // to all of the above; its root is copied on THAT stream and waited for here, while the GPU is still busy with tree 1.
    if (build_tree0) {
      static thread_local hipEvent_t ev_root0 = nullptr;
      if (!ev_root0) { CM_HIP(hipEventCreateWithFlags(&ev_root0, hipEventDisableTiming)); thread_event_owned(ev_root0); }
      hipStream_t ps = tree0_stream;
      CM_HIP(hipMemcpyAsync(pinned_words() + PIN_ROOT0, P.trees[0].merkle.layers[0].p, 32, hipMemcpyDeviceToHost, ps));
      CM_HIP(hipEventRecord(ev_root0, ps));
      if (pp_fork) pp_fork->join();
      else CM_HIP(hipStreamWaitEvent(st, ev_root0, 0));   // what follows on the main stream reads tree 0's LDE
      tree0_guard.joined = true;
      CM_HIP(hipEventSynchronize(ev_root0));
      ht.mark("trace_commit: root 0 arrived");
      memcpy(P.trees[0].root.data(), pinned_words() + PIN_ROOT0, 32);
      ch.mix_root(P.trees[0].root);
    }
    for (int c = 0; c < air::N_COMPONENTS; c++) { pf.claim_log_sizes.push_back(clog[c]); ch.mix_u64(clog[c]); }
    // Everything from root 1 to the relation challenges happens on the device, right behind the tree (k_step_pow_relations:
    // mix_root, interaction proof of work, mix_u64(nonce), Relations::draw + alpha powers): the LogUp kernels start without a
    // host round trip.  The host replays the steps from the copied-back root / nonce when it waits for the claimed sums.
    {
      static_assert(sizeof(DevRelations) == 4 * 4 * (air::N_RELATIONS + air::N_RELATIONS * air::MAX_REL_SIZE), "DevRelations layout");
      uint32_t cw[9];
      memcpy(cw, ch.digest.data(), 32);
      cw[8] = ch.n_sent;
      drel.alloc(sizeof(DevRelations));
      d_step1.alloc(16 * 4);
      uint32_t* rel = drel.u32();
      step_pow_relations(cw, P.trees[1].merkle.layers[0].u32(), INTERACTION_POW_BITS, air::N_RELATIONS, air::MAX_REL_SIZE, rel,
                         rel + 4 * air::N_RELATIONS, d_step1.u32(), st);
      CM_HIP(hipMemcpyAsync(pinned_words() + PIN_STEP1, d_step1.p, 16 * 4, hipMemcpyDeviceToHost, st));
      ht.mark("trace_commit: enqueued");
      parked().reset();   // the previous proof's parked teardown: the host is about to wait ~0.9 ms anyway
      ht.mark("trace_commit: previous proof's teardown");
      P.pace(&P.trees[1].merkle);
      ht.mark("trace_commit: paced (tree 1 at its top)");
    }
    P.tick("trace_commit");

  }

  // interaction PoW, relation draws, tree 2: LogUp columns + commitment enqueue, claimed sums (prover.rs:90-102)
  void interaction() {
    // ---- tree 2: interaction trace (prover.rs:96-102) ----
    hipEvent_t sums_ready = nullptr;
    ColumnSet it_evals;
    // The LogUp tail (claimed sums, the running-sum columns' prefix scans: ~0.15 ms of short memory-bound kernels) runs on a side
    // stream while the tree-2 transforms of every OTHER column start; the running-sum columns go last in their size group.
    // CM_LOGUP_DEFER=0: the tail on the main stream in front of the tree (the round-3 order).
    const bool defer_tail = tune(T_LOGUP_DEFER) != 0;
    std::vector<DevBuf> tail_scratch;   // alive until the host has seen the sums
    Prover::DeferredCols late;
    Prover::CommitPrep tree2_prep;
    bool tree2_prepared = false;
    {
      std::vector<uint32_t> logs;
      for (int c = 0; c < air::N_COMPONENTS; c++) {
        it0[c] = logs.size();
        for (int k = 0; k < air::component_info(c).n_interaction; k++) logs.push_back(clog[c]);
      }
      ht.mark("interaction: entered");
      it_evals.alloc(logs, st);
      ht.mark("interaction: columns allocated");
    }
    {
      tail_scratch.emplace_back(air::N_COMPONENTS * 16);
      uint32_t* const d_sums = tail_scratch.back().u32();
      std::vector<LogupTailJob> jobs(air::N_COMPONENTS);
      // small components (idle opcode components = 16 padding rows, the tiny builtins): ONE launch for all of them
      std::vector<SmallLogupJob> small_jobs;
      uint32_t small_max_log = 0;
      for (int c = 0; c < air::N_COMPONENTS; c++)
        if (clog[c] <= SMALL_COMPONENT_MAX_LOG && !no_small_batch()) {
          small_jobs.push_back(SmallLogupJob{(const uint32_t* const*)tr_evals.dev(tr0[c]), it_evals.dev(it0[c]), clog[c], c});
          small_max_log = std::max(small_max_log, clog[c]);
        }
      DevBuf d_small = upload(small_jobs, st);
      ht.mark("interaction: small jobs uploaded");
      KProfRegion kreg("k_logup(region)", st);
      Fork fk(st);
      // Side streams of this region: every stream the join waits on is one barrier packet at the head of the main queue, ~5.5 us
      // each even when its event has long fired (tools/join_lab.hip: 15 us for three, 42-49 us for seven).  The ~15 LogUp kernels
      // lose nothing on five streams (interaction_gen + interaction_commit 2.66 -> 2.61 ms, profiles/r04n_ab_logup_width.txt); the
      // constraint and quotient regions do (1.0 -> 1.3 ms with four) and keep all eight.  CM_LOGUP_WIDTH: side streams used here.
      const int lw = std::max(1, std::min(tune(T_LOGUP_WIDTH), Fork::N - 1));
      // (which side stream: the streams share a few hardware queues and a queue runs its packets in order — on the stream that
      // shares the MAIN stream's queue this ~90 us latency-bound launch sat in front of the region's largest kernel, round-5
      // timeline.  "logup_small_stream" >= 0 picks the stream, -1 = stream `lw`.)
      const int lss = tune(T_LOGUP_SMALL_STREAM);
      launch_logup_small(d_small.as<SmallLogupJob>(), (uint32_t)small_jobs.size(), small_max_log, (const uint32_t* const*)pp_evals.dev(),
                         drel.as<DevRelations>(), fk.stream(lss >= 0 ? lss : lw));   // first: latency-bound, hidden under the large kernels
      int spos = 0;
      for (int pos = 0; pos < air::N_COMPONENTS; pos++) {   // large components first (see trace generation)
        const int c = by_size_all[pos];
        const air::ComponentInfo& info = air::component_info(c);
        for (int k = 0; k < 4; k++) jobs[c].col[k] = it_evals.ptrs[it0[c] + info.n_interaction - 4 + k];
        jobs[c].log_size = clog[c];
        if (clog[c] <= SMALL_COMPONENT_MAX_LOG && !no_small_batch()) continue;
        launch_logup(c, (const uint32_t* const*)tr_evals.dev(tr0[c]), (const uint32_t* const*)pp_evals.dev(), clog[c],
                     drel.as<DevRelations>(), it_evals.dev(it0[c]), fk.stream(spos == 0 ? Fork::main_or(0) : spos % lw));
        spos++;
      }
      // Tree 2's host-side preparation (size groups, pointer tables, layer buffers, launch plan: ~0.1 ms for 1036 columns) HERE,
      // while the LogUp kernels just launched execute — it used to sit between the LogUp tail and the first transform with the
      // GPU idle (50 us in the round-5 timeline).  Its one host->device copy lands on the main stream behind the region's
      // largest kernel.  CM_COMMIT_PREP_EARLY=0: prepared at the launch (A/B).
      const bool prep_early = tune(T_COMMIT_PREP_EARLY) != 0;
      if (defer_tail) {
        late.late.assign(it_evals.ptrs.size(), 0);
        for (int c = 0; c < air::N_COMPONENTS; c++)
          for (int k = 0; k < 4; k++) late.late[it0[c] + air::component_info(c).n_interaction - 4 + k] = 1;
      }
      if (prep_early) {
        CommittedTree& t = P.trees[2];
        t.coeffs = std::move(it_evals);   // (the kernels above hold the column pointers; the arena just changes its owner)
        t.merkle.pace_ev = Prover::pace_event(2);
        tree2_prep = P.commit_prepare(t, nullptr, true, st, true, /*evals_in_place=*/true, P.pipe_stream(), defer_tail ? &late : nullptr);
        tree2_prepared = true;
      }
      fk.join();
      kreg.close();
      ht.mark("interaction: logup launched + joined");
      hipStream_t sf = st;
      if (defer_tail) {
        sf = thread_side_stream(1);   // (side stream 0 is the pipeline's transform stream)
        hipEvent_t e = Prover::pipe_event();
        CM_HIP(hipEventRecord(e, st));
        CM_HIP(hipStreamWaitEvent(sf, e, 0));
      }
      logup_finalize_all(jobs, d_sums, sf, defer_tail ? &tail_scratch : nullptr);
      static_assert(PIN_SUMS + air::N_COMPONENTS * 4 <= PIN_COEFF, "pinned slot layout");
      const uint32_t* sums = pinned_words() + PIN_SUMS;
      CM_HIP(hipMemcpyAsync((void*)sums, d_sums, air::N_COMPONENTS * 16, hipMemcpyDeviceToHost, sf));
      // the host only waits for THIS copy (an event), after the tree-2 transforms and hashes have been enqueued behind it:
      // no GPU idle time while the host reads and mixes the 34 sums
      static thread_local hipEvent_t ev_sums = nullptr;
      if (!ev_sums) { CM_HIP(hipEventCreateWithFlags(&ev_sums, hipEventDisableTiming)); thread_event_owned(ev_sums); }
      CM_HIP(hipEventRecord(ev_sums, sf));
      sums_ready = ev_sums;
      late.ready = ev_sums;
    }
    ht.mark("interaction: tail enqueued");
    P.tick("interaction_gen");
    // interpolate in place: coeffs alias the evaluation buffer (every size group right in front of its extension)
    if (tree2_prepared) P.commit_launch(tree2_prep);
    else {
      CommittedTree& t = P.trees[2];
      t.coeffs = std::move(it_evals);
      t.merkle.pace_ev = Prover::pace_event(2);
      P.commit_enqueue(t, nullptr, true, st, true, /*evals_in_place=*/true, P.pipe_stream(), defer_tail ? &late : nullptr);
    }
    ht.mark("interaction: tree 2 enqueued");
    {
      CM_HIP(hipEventSynchronize(sums_ready));
      tail_scratch.clear();
      // host replay of the device-side step behind tree 1 (its results were copied back in front of the sums)
      {
        const uint32_t* s1 = pinned_words() + PIN_STEP1;
        CM_CHECK(pinned_words()[PIN_FLAG] == 0, "trace generation: a range-check / bitwise lookup value is out of range");
        memcpy(P.trees[1].root.data(), s1, 32);
        ch.mix_root(P.trees[1].root);
        pf.interaction_pow = (uint64_t)s1[8] | ((uint64_t)s1[9] << 32);
        {
          Channel probe = ch;   // the nonce must satisfy the proof-of-work predicate on the HOST channel
          probe.mix_u64(pf.interaction_pow);
          CM_CHECK(s1[11] == 0 && probe.trailing_zeros() >= INTERACTION_POW_BITS, "interaction proof of work: device step diverged from the host channel");
        }
        ch.mix_u64(pf.interaction_pow);
        DevRelations drel_h;
        draw_relations(ch, hrel, drel_h);
        CM_CHECK(ch.n_sent == s1[10] && memcmp(drel_h.z[0], s1 + 12, 16) == 0, "relations: device transcript diverged from the host channel");
      }
      const uint32_t* sums = pinned_words() + PIN_SUMS;
      for (int c = 0; c < air::N_COMPONENTS; c++) pf.claimed_sums.push_back(QM31::from_u32(sums + 4 * c));
      for (auto& cs : pf.claimed_sums) ch.mix_felts(&cs, 1);
    }
    tr_evals.buf.release();
  }

  // stwo prove: constraint quotients of all components -> composition polynomial, tree 3 enqueued
  void composition() {
    // ---- stwo prove: composition polynomial.  Everything that does not depend on the random coefficient (accumulators,
    // slots, per-component arguments, their uploads and memsets) is prepared and enqueued here, behind the tree-2 kernels;
    // the root is read back, the coefficient drawn and its powers uploaded right before the launches. ----
    size_t total_constraints = 0;
    coff.assign(air::N_COMPONENTS, 0);
    for (int c = 0; c < air::N_COMPONENTS; c++) { coff[c] = total_constraints; total_constraints += air::component_info(c).n_constraints; }
    powers.assign(total_constraints, QM31());
    d_powers.alloc(16 * total_constraints);
    // accumulators: 4 columns per evaluation log.  The top size gets its own ColumnSet (it becomes the coefficient
    // set of tree 3), all smaller sizes share one — two pointer-table uploads and two memsets instead of one pair
    // per size.
    struct AccRef { ColumnSet* set; size_t first; uint32_t* const* dev() const { return set->dev(first); } };
    std::map<uint32_t, AccRef> accs;  // evaluation log -> its 4 accumulator columns
    std::set<uint32_t> acc_interpolated;   // evaluation logs whose accumulator is interpolated inside the constraints region
    std::map<uint32_t, std::vector<int>> cgroups;
    for (int c = 0; c < air::N_COMPONENTS; c++) cgroups[clog[c] + 1].push_back(c);
    ColumnSet acc_top, acc_rest;
    {
      std::vector<uint32_t> rest_logs;
      for (auto& kv : cgroups)
        if (kv.first != comp_log) { accs[kv.first] = AccRef{&acc_rest, rest_logs.size()}; rest_logs.insert(rest_logs.end(), 4, kv.first); }
      CM_CHECK(cgroups.count(comp_log), "composition polynomial log size mismatch");
      accs[comp_log] = AccRef{&acc_top, 0};
      // (pointer tables: in the phase's one upload below; zeroing: one launch together with the slot buffer)
      acc_top.alloc(std::vector<uint32_t>(4, comp_log), st, false);
      if (!rest_logs.empty()) acc_rest.alloc(rest_logs, st, false);
    }
    // The constraints are evaluated on CanonicCoset(log + 1).  With log_blowup_factor = 1 (REGULAR_96_BITS) that is the
    // committed LDE domain and the kernels read the committed columns; with a larger blowup every polynomial is evaluated
    // on its (log + 1) domain separately (Stwo does the same: `poly.evaluate(eval_domain)`), at the cost of one more forward
    // transform per column and its memory.
    CM_CHECK(cfg.log_blowup_factor >= 1 && cfg.log_blowup_factor <= 4, "PcsConfig: log_blowup_factor must be in 1..4");
    ColumnSet cdom[3];   // evaluation-domain copies of trees 0..2 (only when log_blowup_factor > 1)
    if (cfg.log_blowup_factor > 1) {
      for (int t = 0; t < 3; t++) {
        std::vector<uint32_t> logs(P.trees[t].coeffs.logs);
        for (auto& l : logs) l += 1;
        cdom[t].alloc(logs, st);
        std::vector<const uint32_t*> table;
        struct Grp { uint32_t log, n; size_t off; };
        std::vector<Grp> grps;
        for (auto& kv : by_log(P.trees[t].coeffs.logs)) {
          grps.push_back(Grp{kv.first, (uint32_t)kv.second.size(), table.size()});
          for (auto i : kv.second) table.push_back(P.trees[t].coeffs.ptrs[i]);
          for (auto i : kv.second) table.push_back(cdom[t].ptrs[i]);
        }
        DevBuf d_table = upload(table, st);
        const uint32_t** dt = d_table.as<const uint32_t*>();
        for (auto& g : grps) evaluate((const uint32_t* const*)(dt + g.off), (uint32_t* const*)(dt + g.off + g.n), g.n, g.log, g.log + 1, *P.tw, st);
        CM_HIP(hipStreamSynchronize(st));   // d_table is a temporary
      }
    }
    auto cdom_cols = [&](int t, size_t first) -> const uint32_t* const* {
      return (const uint32_t* const*)(cfg.log_blowup_factor > 1 ? cdom[t].dev(first) : P.trees[t].lde.dev(first));
    };
    {
      // Components are independent except for the shared per-size accumulator.  Small sizes (many idle
      // components of 2^4 rows, each a latency-bound launch) get a private zeroed accumulator slot per
      // component and run concurrently on side streams; the slots are summed afterwards (field addition is
      // exact, so the order is irrelevant).  Large sizes keep the shared accumulator and one stream per size.
      constexpr uint32_t SLOT_MAX_LOG = 15;
      struct SlotGroup { uint32_t el; uint32_t n; size_t off_words; size_t tab0; };
      std::vector<SlotGroup> sgroups;
      std::vector<int> slot_of(air::N_COMPONENTS, -1);
      std::vector<uint32_t*> slot_tab;
      size_t slot_words = 0;
      for (auto& kv : cgroups)
        if (kv.second.size() > 1 && kv.first <= SLOT_MAX_LOG) {
          SlotGroup g{kv.first, (uint32_t)kv.second.size(), slot_words, slot_tab.size()};
          for (size_t k = 0; k < kv.second.size(); k++) slot_of[kv.second[k]] = (int)(slot_tab.size() / 4 + k);
          slot_tab.resize(slot_tab.size() + 4 * kv.second.size());
          slot_words += (size_t)4 * kv.second.size() << kv.first;
          sgroups.push_back(g);
        }
      DevBuf slots(((slot_words * 4 + 15) & ~(size_t)15) + 16);
      if (slot_words)
        for (auto& g : sgroups)
          for (uint32_t k = 0; k < 4 * g.n; k++) slot_tab[g.tab0 + k] = slots.u32() + g.off_words + ((size_t)k << g.el);
      {
        void* const zp[3] = {acc_top.buf.p, acc_rest.buf.p ? acc_rest.buf.p : acc_top.buf.p, slots.p};
        const size_t zb[3] = {acc_top.buf.bytes & ~(size_t)15, acc_rest.buf.p ? (acc_rest.buf.bytes & ~(size_t)15) : 0, (slot_words * 4 + 15) & ~(size_t)15};
        zero_ranges(zp, zb, st);
      }
      // ONE device table for the phase: [acc_top pointers | acc_rest pointers | slot table | small-component arguments | their ids];
      // its address is known before the arguments (which point into it) are built
      const size_t o_top = 0, o_rest = 64, o_slot = (o_rest + acc_rest.ptrs.size() * sizeof(void*) + 15) & ~(size_t)15;
      const size_t o_args = (o_slot + slot_tab.size() * sizeof(void*) + 15) & ~(size_t)15;
      const size_t o_cids = o_args + air::N_COMPONENTS * sizeof(ConstraintArgs);
      d_ctab.alloc(o_cids + air::N_COMPONENTS * sizeof(int) + 16);
      uint8_t* const ctab = d_ctab.as<uint8_t>();
      acc_top.d_view = (uint32_t**)(ctab + o_top);
      acc_rest.d_view = (uint32_t**)(ctab + o_rest);
      uint32_t** const d_slot_tab = (uint32_t**)(ctab + o_slot);
      // arguments of every component first: the small ones (<= 2^SMALL_COMPONENT_MAX_LOG rows) go to ONE batched launch
      // whose argument array has to be uploaded before the fork
      std::vector<ConstraintArgs> cargs(air::N_COMPONENTS);
      std::vector<ConstraintArgs> small_args;
      std::vector<int> small_cids;
      uint32_t small_max_log = 0;
      for (auto it = cgroups.rbegin(); it != cgroups.rend(); ++it) {
        for (int c : it->second) {
          const air::ComponentInfo& info = air::component_info(c);
          ConstraintArgs& a = cargs[c];
          a.tr = cdom_cols(1, tr0[c]);
          a.it = cdom_cols(2, it0[c]);
          a.pp = cdom_cols(0, 0);
          a.rels = drel.as<DevRelations>();
          a.coeff = d_powers.u32() + 4 * coff[c];
          a.acc = slot_of[c] >= 0 ? d_slot_tab + 4 * slot_of[c] : accs.at(it->first).dev();
          a.log_size = clog[c];
          a.n_base = info.n_base_constraints;
          (pf.claimed_sums[c] * inv(M31::from_u32(1u << clog[c]))).to_u32(a.cumsum_shift);
          for (uint32_t k = 0; k < 2; k++) {
            CPoint<M31> p = point_at_index(domain_index_at(clog[c] + 1, k));
            a.denom_inv[k] = inv(coset_vanishing_canonic<M31>(clog[c], p)).v;
          }
          // a small component either owns a private slot or is alone in its size group: no accumulator is shared inside the batch
          if (clog[c] <= SMALL_COMPONENT_MAX_LOG && !no_small_batch() && (slot_of[c] >= 0 || it->second.size() == 1)) {
            small_args.push_back(a);
            small_cids.push_back(c);
            small_max_log = std::max(small_max_log, clog[c]);
          }
        }
      }
      {
        std::vector<uint8_t> blob(o_cids + small_cids.size() * sizeof(int));
        memcpy(blob.data() + o_top, acc_top.ptrs.data(), acc_top.ptrs.size() * sizeof(void*));
        if (!acc_rest.ptrs.empty()) memcpy(blob.data() + o_rest, acc_rest.ptrs.data(), acc_rest.ptrs.size() * sizeof(void*));
        if (!slot_tab.empty()) memcpy(blob.data() + o_slot, slot_tab.data(), slot_tab.size() * sizeof(void*));
        if (!small_args.empty()) memcpy(blob.data() + o_args, small_args.data(), small_args.size() * sizeof(ConstraintArgs));
        if (!small_cids.empty()) memcpy(blob.data() + o_cids, small_cids.data(), small_cids.size() * sizeof(int));
        stage_upload(d_ctab.p, blob.data(), blob.size(), st);
      }
      const ConstraintArgs* const d_small_args = (const ConstraintArgs*)(ctab + o_args);
      const int* const d_small_cids = (const int*)(ctab + o_cids);
      // Transcript step on the device (prover.rs:131, stwo prove: mix the interaction root, draw the random coefficient): the
      // channel state goes over in the kernel arguments, k_chan_init_mix_root_draw mixes root 2 and draws the coefficient,
      // k_coeff_powers expands its powers — the constraint kernels start right behind the tree, with no host round trip.  The
      // host replays both steps on its own channel when root 3 comes back (oods_sampling) and checks the coefficients agree.
      {
        uint32_t cw[9];
        memcpy(cw, ch.digest.data(), 32);
        cw[8] = ch.n_sent;
        d_step2.alloc(4 * (16 + 4 + 8));
        uint32_t* d_chan = d_step2.u32();
        chan_init_mix_root_draw(cw, d_chan, P.trees[2].merkle.layers[0].u32(), d_chan + 16, d_chan + 20, st);
        coeff_powers(d_chan + 16, d_powers.u32(), (uint32_t)total_constraints, st);
        static_assert(PIN_COEFF + 4 == PIN_ROOT2, "{coefficient, root 2} come back in one copy");
        CM_HIP(hipMemcpyAsync(pinned_words() + PIN_COEFF, d_chan + 16, 48, hipMemcpyDeviceToHost, st));
      }
      P.tick("interaction_commit");
      for (int t = 0; t < 3; t++) for (auto l : P.trees[t].coeffs.logs) pf.cells += 1ull << l;
      P.pace(&P.trees[2].merkle);
      KProfRegion kreg("k_constraints(region)", st);
      Fork fk(st);
      // side-stream plan of the region (A/B: tuning key "cons_plan" = g0,g1,g2,g3,small,slot0,slot1,slot2 = stream index of the size
      // groups in descending size, of the batched small components and of the slotted ones)
      // ("cons_plan": eight octal digits, default 01237456 = streams {0,1,2,3 | 7 | 4,5,6})
      std::vector<int> cplan(8);
      {
        const int code = tune(T_CONS_PLAN);
        for (int k = 0; k < 8; k++) cplan[k] = (code >> (3 * (7 - k))) & 7;
      }
      launch_constraints_small(d_small_args, d_small_cids, (uint32_t)small_args.size(), small_max_log,
                               fk.stream(cplan[4]));   // first: latency-bound, hidden under the large kernels
      ht.mark("constraints: small batch launched");
      int gi = 0, small_rr = 0;
      static const bool early_interp = getenv("CM_NO_EARLY_ACC_INTERP") == nullptr;   // A/B switch
      // A WIDE component of few rows (poseidon2: 443 columns, 426 constraints on 2^10 evaluation rows) is one long program per row
      // on four blocks: ~0.33 ms of pure latency.  In size order it was enqueued last and ran ALONE behind the large kernels
      // (round-5 timeline: 5570 -> 5905 us of a region that the large groups had left at ~5720).  Launched first it hides under
      // them.  "cons_wide_first" = 0: the plain size order (A/B).
      const bool wide_first = tune(T_CONS_WIDE_FIRST) != 0;
      auto is_wide = [&](int c) { return wide_first && air::component_info(c).n_trace >= 128 && clog[c] <= 14; };
      std::vector<hipStream_t> cstream(air::N_COMPONENTS, nullptr);
      for (auto it = cgroups.rbegin(); it != cgroups.rend(); ++it, ++gi)
        for (int c : it->second) {
          if (std::find(small_cids.begin(), small_cids.end(), c) != small_cids.end()) continue;
          // the components of a size group share its accumulator: one stream for all of them; slotted ones are independent
          cstream[c] = slot_of[c] >= 0 ? fk.stream(cplan[5 + (small_rr++ % 3)]) : fk.stream(gi == 0 ? Fork::main_or(cplan[0]) : cplan[gi % 4]);
          if (is_wide(c)) launch_constraints(c, cargs[c], cstream[c]);
        }
      gi = 0;
      // large groups first (descending size) so the long kernels start early
      for (auto it = cgroups.rbegin(); it != cgroups.rend(); ++it, ++gi) {
        bool one_stream = true;   // every component of the group ran on the group's stream (no slots, no batched small ones)
        for (int c : it->second) {
          if (std::find(small_cids.begin(), small_cids.end(), c) != small_cids.end()) { one_stream = false; continue; }
          if (slot_of[c] >= 0) one_stream = false;
          if (!is_wide(c)) launch_constraints(c, cargs[c], cstream[c]);
        }
        // DomainEvaluationAccumulator::finalize starts here for such a group: its accumulator is interpolated on the same
        // stream right behind its constraint kernels — no second fork/join region for the large accumulators
        if (one_stream && early_interp) {
          interpolate(accs.at(it->first).dev(), 4, it->first, *P.tw, fk.stream(gi == 0 ? Fork::main_or(cplan[0]) : cplan[gi % 4]));
          acc_interpolated.insert(it->first);
        }
        ht.mark("constraints: size group launched");
      }
      fk.join();
      kreg.close();
      for (auto& g : sgroups) sum_slots(accs.at(g.el).dev(), slots.u32() + g.off_words, g.n, g.el, st);
    }
    P.tick("constraints");
    // DomainEvaluationAccumulator::finalize.  Stwo walks the sizes upward: interpolate(vals_l + evaluate_l(cur)).
    // Interpolation is linear and interpolate_l(evaluate_l(cur)) is cur zero-padded (coefficient bases nest),
    // so the same coefficients come from interpolating every accumulator at its OWN size (independent, on side
    // streams) and adding the zero-padded coefficient vectors — field arithmetic is exact, the result is
    // bit-identical, and the extend/add chain over the large domains disappears.
    {
      CommittedTree& t = P.trees[3];
      if (acc_interpolated.empty()) {
        Fork fk(st);
        int k = 0;
        for (auto it = accs.rbegin(); it != accs.rend(); ++it, ++k) interpolate(it->second.dev(), 4, it->first, *P.tw, fk.stream(k));
        fk.join();
      } else {   // what is left are the slotted / batched small sizes: a handful of single-launch transforms, in a row
        for (auto it = accs.rbegin(); it != accs.rend(); ++it)
          if (!acc_interpolated.count(it->first)) interpolate(it->second.dev(), 4, it->first, *P.tw, st);
      }
      {
        AddColumnsSrc as;
        as.n = 0;
        for (auto& kv : accs)
          if (kv.first != comp_log) {
            CM_CHECK(as.n < 28, "composition: too many accumulator sizes");
            as.log[as.n] = kv.first;
            as.src[as.n++] = (const uint32_t* const*)kv.second.dev();
          }
        add_columns_multi(acc_top.dev(), as, 4, st);
      }
      t.coeffs = std::move(acc_top);
      oods_prepare();   // needs tree 3's coefficient pointers, nothing of its commitment
      KProfAloneScope kprof_alone_scope;   // (measurement only: the composition tree has the GPU to itself, kprof.hpp)
      P.commit_enqueue(t, nullptr, true, st, true, false, P.pipe_stream());
    }
  }

  // Sampling jobs = (log size, point): every column at the OODS point, plus the previous-row mask
  // (oods - trace_step(log)) of each component's last LogUp column group.  All jobs share one pointer-table
  // upload, one scratch buffer and ONE device->host copy of the results.  Everything but the points themselves is
  // built and uploaded HERE, in front of the composition tree's kernels in stream order (called by composition() right
  // before that tree is enqueued): the table's host->device copy is off the path root 3 -> evaluation kernels.
  void oods_prepare() {
    ojobs.clear();
    n_oods_out = 0;
    {
      // Two chunks in the transcript's flat order (tree, column, mask): the sampled values of chunk 0 come back first and the
      // host hashes them (channel.mix_felts, ~0.1 ms of sequential Blake2s over ~30 KB) while the GPU evaluates chunk 1 — the
      // split makes the two take about as long (CM_OODS_SPLIT: per mille of the samples in chunk 0; 1000 = one chunk)
      const uint32_t split_pm = (uint32_t)tune(T_OODS_SPLIT);
      std::vector<std::vector<char>> has_prev(4);
      size_t total = 0;
      for (int t = 0; t < 4; t++) { has_prev[t].assign(P.trees[t].coeffs.size(), 0); total += P.trees[t].coeffs.size(); }
      for (int c = 0; c < air::N_COMPONENTS; c++) {
        const int ni = air::component_info(c).n_interaction;
        for (int k = ni - 4; k < ni; k++) { has_prev[2][it0[c] + k] = 1; total++; }
      }
      const size_t split = split_pm >= 1000 ? total : total * split_pm / 1000;
      std::map<std::tuple<int, bool, uint32_t>, std::vector<ORef>> groups;   // (chunk, previous-row mask, log size)
      size_t cum = 0;
      n_flat0 = 0;
      for (int t = 0; t < 4; t++)
        for (uint32_t c = 0; c < P.trees[t].coeffs.size(); c++) {
          const int chunk = cum < split ? 0 : 1;
          const uint32_t log = P.trees[t].coeffs.logs[c];
          if (has_prev[t][c]) groups[{chunk, true, log}].push_back({t, c, true});
          groups[{chunk, false, log}].push_back({t, c, false});
          cum += has_prev[t][c] ? 2 : 1;
          if (chunk == 0) n_flat0 = cum;
        }
      for (auto& kv : groups) {
        OJob j{std::get<2>(kv.first), std::get<1>(kv.first), {}, kv.second};
        j.chunk = std::get<0>(kv.first);
        ojobs.push_back(std::move(j));
      }
    }
    {
      std::vector<const uint32_t*> table;
      for (auto& j : ojobs) {
        j.off = table.size();
        j.out_off = n_oods_out;
        for (auto& r : j.refs) table.push_back(P.trees[r.t].coeffs.ptrs[r.c]);
        n_oods_out += j.refs.size();
        if (j.chunk == 0) n_oods_out0 = n_oods_out;
      }
      d_oods_table = upload(table, st);
      d_oods_out.alloc(n_oods_out * 16);
    }
  }

  // OODS point, mask values of every column (eval_at_point), DEEP-quotient plan built while the kernels run
  void oods_sampling() {
    // where the sampled value of (tree, column) lands in d_oods_out: the DEEP-quotient coefficients are computed on the
    // device straight from there (k_quotient_coeffs)
    std::vector<std::vector<uint32_t>> sidx_cur(4), sidx_prev(4);
    for (int t = 0; t < 4; t++) { sidx_cur[t].assign(P.trees[t].coeffs.size(), 0); sidx_prev[t].assign(P.trees[t].coeffs.size(), 0); }
    for (auto& j : ojobs)
      for (size_t i = 0; i < j.refs.size(); i++)
        (j.refs[i].prev ? sidx_prev : sidx_cur)[j.refs[i].t][j.refs[i].c] = (uint32_t)(j.out_off + i);
    pf.sampled_values.resize(4);
    for (int t = 0; t < 4; t++) pf.sampled_values[t].resize(P.trees[t].coeffs.size());
    // Device-side OODS step (A/B: CM_HOST_OODS=1 restores the host form): the channel state after the coefficient draw is still
    // in HBM (d_step2), so mix_root(root 3) + the draw of CirclePoint::get_random_point run on the device right behind the
    // composition tree, k_oods_maps turns the felt into every job's point and per-bit factors, and the evaluation kernels
    // start without the host having seen root 3 (it used to cost the host replay + the launches: ~60 us of idle GPU).  The host
    // replays the same steps when root 3 arrives and checks the drawn felt.
    static const bool dev_oods = getenv("CM_HOST_OODS") == nullptr;
    oods_poll = dev_oods && tune(T_OODS_POLL) != 0;
    static thread_local hipEvent_t ev_root3 = nullptr, ev_chunk0 = nullptr;
    bool chunk0_event = false;
    if (dev_oods) P.tick("composition_commit");
    const uint32_t* oods_w = nullptr;
    DevBuf d_step3;
    if (dev_oods) {
      d_step3.alloc(4 * (4 + 8));
      uint32_t* d_chan = d_step2.u32();
      chan_mix_root_draw(d_chan, P.trees[3].merkle.layers[0].u32(), d_step3.u32(), d_step3.u32() + 4, st);
      // root 3 and the felt come back HERE in stream order — in front of the evaluation kernels enqueued next
      CM_HIP(hipMemcpyAsync(pinned_words() + PIN_STEP3, d_step3.p, 48, hipMemcpyDeviceToHost, st));   // {felt[4], root 3 [8]}: one copy
      if (!ev_root3) { CM_HIP(hipEventCreateWithFlags(&ev_root3, hipEventDisableTiming)); thread_event_owned(ev_root3); }
      CM_HIP(hipEventRecord(ev_root3, st));
      std::vector<EapJob> ej[2];
      for (auto& j : ojobs) {
        EapJob e{j.log, (uint32_t)j.refs.size(), d_oods_table.as<const uint32_t*>() + j.off, QM31(), QM31(), d_oods_out.u32() + 4 * j.out_off};
        if (j.prev) {
          CPoint<M31> step = point_at_index(subgroup_gen_index(j.log));
          e.has_shift = true; e.shift_x = step.x.v; e.shift_y = (-step.y).v;
        }
        ej[j.chunk].push_back(e);
      }
      // chunk 0, its values to the host, chunk 1, its values (one pinned landing buffer)
      uint8_t* land = (uint8_t*)stage_landing(n_oods_out * 16, st);
      oods_w = (const uint32_t*)land;
      // Polling form (default; A/B: CM_OODS_POLL=0 = copy commands + event / stream synchronisation): the reduction kernel writes
      // every sampled value into the landing buffer itself, next to its copy in HBM — the 12.5 KB copy of chunk 0 went through the
      // SDMA engine and cost 27 us between the two chunks' kernels — and the host does not wait for an event: it fills the landing
      // words with a value no M31 word takes and watches them change.
      const bool host_write = oods_poll && tune(T_OODS_HOST_WRITE) != 0;
      if (oods_poll) memset(land, 0xFF, n_oods_out * 16);
      if (host_write)
        for (int c = 0; c < 2; c++)
          for (auto& e : ej[c]) e.h_out = (uint32_t*)land + (e.d_out - d_oods_out.u32());
      eval_at_point_multi(ej[0], st, d_step3.u32());
      if (n_oods_out0 && !host_write) CM_HIP(hipMemcpyAsync(land, d_oods_out.p, n_oods_out0 * 16, hipMemcpyDeviceToHost, st));
      if (!ej[1].empty()) {
        if (!oods_poll) {
          if (!ev_chunk0) { CM_HIP(hipEventCreateWithFlags(&ev_chunk0, hipEventDisableTiming)); thread_event_owned(ev_chunk0); }
          CM_HIP(hipEventRecord(ev_chunk0, st));
        }
        chunk0_event = true;
        eval_at_point_multi(ej[1], st, d_step3.u32());
        if (!host_write)
          CM_HIP(hipMemcpyAsync(land + n_oods_out0 * 16, d_oods_out.as<uint8_t>() + n_oods_out0 * 16, (n_oods_out - n_oods_out0) * 16, hipMemcpyDeviceToHost, st));
      }
    }
    // ONE round trip for two roots: root 3 is waited for; root 2 and the random coefficient of the device-side step are
    // already in pinned memory.  Host replay in transcript order.
    ht.mark("composition: enqueued, waiting for root 3");
    if (dev_oods) {   // the root comes back behind the tree, NOT behind the evaluation kernels enqueued after it
      CM_HIP(hipEventSynchronize(ev_root3));
      memcpy(P.trees[3].root.data(), pinned_words() + PIN_STEP3 + 4, 32);
    } else {
      P.trees[3].merkle.root(P.trees[3].root.data(), st);
    }
    ht.mark("composition: root 3 arrived");
    {
      memcpy(P.trees[2].root.data(), pinned_words() + PIN_ROOT2, 32);
      ch.mix_root(P.trees[2].root);
      const QM31 rho = ch.draw_felt();
      CM_CHECK(rho == QM31::from_u32(pinned_words() + PIN_COEFF), "composition: device transcript diverged from the host channel");
      QM31 cur(M31(1));
      for (size_t g = powers.size(); g-- > 0;) { powers[g] = cur; cur = cur * rho; }   // host copy: the OODS check needs it
    }
    ch.mix_root(P.trees[3].root);
    ht.mark("composition: host replay of the coefficient step");
    if (!dev_oods) P.tick("composition_commit");

    // host side of compute_fri_quotients for every size group, packed into ONE upload:
    // [column pointers | out pointers | col_index | coef_c | batches] per group, 16-byte aligned
    std::vector<uint8_t> qblob;
    auto qput = [&](const void* ptr, size_t bytes) {
      size_t o = (qblob.size() + 15) & ~(size_t)15;
      qblob.resize(o + bytes);
      if (bytes && ptr) memcpy(qblob.data() + o, ptr, bytes);
      return o;
    };

    ht.mark("(composition commit done)");
    // ---- OODS sampling ----
    {
      QM31 t;
      oods = draw_oods_point(ch, &t);
      if (dev_oods) CM_CHECK(t == QM31::from_u32(pinned_words() + PIN_STEP3), "oods: device transcript diverged from the host channel");
    }
    ht.mark("oods: point drawn");
    std::map<uint32_t, CPoint<QM31>> prev_points;
    {
      std::vector<OJob>& jobs = ojobs;
      const size_t n_out = n_oods_out;
      DevBuf& dout = d_oods_out;
      {
        std::vector<EapJob> ej;
        for (auto& j : jobs) {
          j.pt = oods;
          if (j.prev) {
            CPoint<M31> step = point_at_index(subgroup_gen_index(j.log));
            CPoint<QM31> neg{QM31(step.x), QM31(-step.y)};
            j.pt = cadd(oods, neg);
            prev_points[j.log] = j.pt;
          }
          ej.push_back(EapJob{j.log, (uint32_t)j.refs.size(), d_oods_table.as<const uint32_t*>() + j.off, j.pt.x, j.pt.y,
                              dout.u32() + 4 * j.out_off});
        }
        if (!dev_oods) eval_at_point_multi(ej, st);   // (device form: already running, from the felt of the device-side step)
      }
      const uint32_t* w = dev_oods ? oods_w : (const uint32_t*)stage_download_async(dout.p, n_out * 16, st);   // read after the sync below
      ht.mark("oods: enqueued");
      // ---- while the evaluation kernels run: everything about the sampled values and the DEEP quotients
      // (compute_fri_quotients) that does not depend on the values themselves ----
      for (auto& j : jobs)
        for (auto& r : j.refs) pf.sampled_values[r.t][r.c].push_back(QM31());   // mask order [-1, 0]: sized now, filled below
      size_t n_samples = 0;
      for (auto& t : pf.sampled_values) for (auto& c : t) n_samples += c.size();
      std::map<uint32_t, std::vector<QRef>, std::greater<uint32_t>> qgroups;  // LDE log -> columns (tree-major order)
      for (int t = 0; t < 4; t++)
        for (uint32_t c = 0; c < P.trees[t].lde.size(); c++) qgroups[P.trees[t].lde.logs[c]].push_back({t, c});
      for (auto& kv : qgroups) {
        QGroup g;
        g.log = kv.first;
        for (uint32_t i = 0; i < kv.second.size(); i++) {
          const QRef& r = kv.second[i];
          g.cols.push_back(P.trees[r.t].lde.ptrs[r.c]);
          const size_t ns = pf.sampled_values[r.t][r.c].size();
          for (size_t k = 0; k < ns; k++) {
            // sample points: [oods] or [prev, oods]
            CPoint<QM31> pt = (ns == 2 && k == 0) ? prev_points[P.trees[r.t].coeffs.logs[r.c]] : oods;
            size_t bi = 0;
            for (; bi < g.batches.size(); bi++) if (g.batches[bi].pt.x == pt.x && g.batches[bi].pt.y == pt.y) break;
            if (bi == g.batches.size()) g.batches.push_back(QBatch{pt, {}});
            g.batches[bi].entries.push_back(QEntry{i, (ns == 2 && k == 0) ? sidx_prev[r.t][r.c] : sidx_cur[r.t][r.c]});
          }
        }
        if (framing().sample_batch_sorted)   // ColumnSampleBatch::new_vec as a BTreeMap keyed by point (framing.hpp)
          std::stable_sort(g.batches.begin(), g.batches.end(), [](const QBatch& a, const QBatch& b) { return secure_point_less(a.pt, b.pt); });
        size_t n_entries = 0;
        for (auto& bt : g.batches) n_entries += bt.entries.size();
        g.out.alloc(std::vector<uint32_t>(4, g.log), st, false);
        g.o_cols = qput(g.cols.data(), g.cols.size() * sizeof(void*));
        g.o_out = qput(g.out.ptrs.data(), 4 * sizeof(void*));
        {
          std::vector<uint32_t> ci, si;
          std::vector<const uint32_t*> ep;   // per entry: the column pointer itself
          std::vector<QuotientBatch> qb(g.batches.size());
          for (size_t bi = 0; bi < g.batches.size(); bi++) {
            memset(&qb[bi], 0, sizeof(QuotientBatch));
            qb[bi].begin = (uint32_t)ci.size();
            for (auto& en : g.batches[bi].entries) { ci.push_back(en.col); si.push_back(en.sidx); ep.push_back(g.cols[en.col]); }
            qb[bi].end = (uint32_t)ci.size();
            g.batches[bi].pt.x.to_u32(qb[bi].point);   // words = (Pr.x, Pi.x): QM31 = (a.a, a.b, b.a, b.b)
            g.batches[bi].pt.y.to_u32(qb[bi].point + 4);
          }
          g.o_ci = qput(ci.data(), ci.size() * 4);
          g.o_ep = qput(ep.data(), ep.size() * sizeof(void*));
          g.o_sidx = qput(si.data(), si.size() * 4);
          g.o_cc = qput(nullptr, n_entries * 16);            // filled by k_quotient_coeffs
          g.o_qb = qput(qb.data(), qb.size() * sizeof(QuotientBatch));   // sums / batch coefficient filled on the device
        }
        n_qjobs += g.batches.size();
        qg.push_back(std::move(g));
      }
      {  // the whole plan goes to the device now, behind the OODS kernels; only the random coefficient is still missing
        o_qjobs = qput(nullptr, n_qjobs * sizeof(QuotientCoefJob));
        d_qblob.alloc(qblob.size());
        uint8_t* base = d_qblob.as<uint8_t>();
        QuotientCoefJob* qj = (QuotientCoefJob*)(qblob.data() + o_qjobs);
        size_t k = 0;
        for (auto& g : qg)
          for (size_t bi = 0; bi < g.batches.size(); bi++, k++) {
            qj[k].qb = (QuotientBatch*)(base + g.o_qb) + bi;
            qj[k].coef_c = (uint32_t*)(base + g.o_cc);
            qj[k].sample_idx = (const uint32_t*)(base + g.o_sidx);
          }
        stage_upload(d_qblob.p, qblob.data(), qblob.size(), st);
      }
      ht.mark("oods: overlapped quotient planning");
      // channel.mix_felts(flattened sampled values): chunk 0 is hashed as soon as it has landed, while chunk 1 is evaluated
      Channel::FeltMixer fm;
      ch.mix_felts_begin(fm);
      std::vector<QM31> flat;
      flat.reserve(n_samples);
      auto fill = [&](int chunk) {
        for (auto& j : jobs)
          if (j.chunk == chunk)
            for (size_t i = 0; i < j.refs.size(); i++) {
              auto& sv = pf.sampled_values[j.refs[i].t][j.refs[i].c];
              (j.refs[i].prev ? sv.front() : sv.back()) = QM31::from_u32(&w[4 * (j.out_off + i)]);
            }
      };
      size_t ft = 0, fc = 0, n_mixed = 0;   // cursor of the flat order: (tree, column)
      auto mix_upto = [&](size_t n_flat) {
        flat.clear();
        while (ft < 4 && n_mixed + flat.size() < n_flat) {
          auto& tr = pf.sampled_values[ft];
          if (fc == tr.size()) { ft++; fc = 0; continue; }
          for (auto& sm : tr[fc]) flat.push_back(sm);
          fc++;
        }
        ch.mix_felts_update(fm, flat.data(), flat.size());
        n_mixed += flat.size();
      };
      // all words of [w0, w1) have arrived from the device (every word the copy writes is < 2^31); gives up after ~2 ms of
      // spinning and lets the caller synchronise instead
      auto landed = [&](size_t q0, size_t q1) {
        const volatile uint32_t* v = w;
        for (int spin = 0; spin < 200000; spin++) {
          size_t i = 4 * q1;
          while (i > 4 * q0 && v[i - 1] != 0xFFFFFFFFu) i--;
          if (i == 4 * q0) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
          cpu_relax();
        }
        return false;
      };
      if (chunk0_event) {
        if (!oods_poll) CM_HIP(hipEventSynchronize(ev_chunk0));
        else if (!landed(0, n_oods_out0)) CM_HIP(hipStreamSynchronize(st));
        ht.mark("oods: chunk 0 landed");
        fill(0);
        mix_upto(n_flat0);
        ht.mark("oods: chunk 0 filled + mixed");
      }
      if (!(oods_poll && landed(chunk0_event ? n_oods_out0 : 0, n_out))) CM_HIP(hipStreamSynchronize(st));
      ht.mark("oods: waited for gpu");
      if (!chunk0_event) fill(0);
      fill(1);
      mix_upto(n_samples);
      CM_CHECK(n_mixed == n_samples, "oods: flat sample count");
      ch.mix_felts_end(fm);
      ht.mark("oods: fill + mix_felts");
    }
    P.tick("oods_sampling");
  }

  // compute_fri_quotients: one launch per size group
  void deep_quotients() {
    // ---- DEEP quotients (compute_fri_quotients): the value-dependent coefficients, one upload, the launches ----
    QM31 qcoeff = ch.draw_felt();
    {
      // per-sample coefficients, batch sums and batch coefficients: one small kernel on the sampled values in HBM
      quotient_coeffs((const QuotientCoefJob*)(d_qblob.as<uint8_t>() + o_qjobs), (uint32_t)n_qjobs, d_oods_out.u32(), qcoeff, st);
      const uint8_t* base = d_qblob.as<uint8_t>();
      // one kernel per size group, independent outputs: the small groups (latency-bound, ~140 us in a row) overlap the large
      KProfRegion kregq("k_quotients", st);   // concurrent launches: timed as one interval
      Fork fkq(st);
      int qk = 0;
      std::vector<std::pair<QuotientArgs, double>> qargs;
      for (auto& g : qg) {
        QuotientArgs a;
        a.tw = view(*P.tw); a.log_size = g.log;
        a.cols = (const uint32_t* const*)(base + g.o_cols);
        a.out = (uint32_t* const*)(base + g.o_out);
        a.col_index = (const uint32_t*)(base + g.o_ci);
        a.entry_cols = (const uint32_t* const*)(base + g.o_ep);
        a.coef_c = (const uint32_t*)(base + g.o_cc);
        a.batches = (const QuotientBatch*)(base + g.o_qb);
        a.n_batches = (uint32_t)g.batches.size();
        qargs.push_back({a, (double)g.cols.size()});
        q_logs.push_back(g.log);
        quotients.push_back(std::move(g.out));
      }
      // (round 6) The FRI first-layer tree's leaf layer — 2^log hashes of four words each, 8.4 M compressions at the metric config,
      // 0.21 ms as a launch of its own with the GPU otherwise idle — is written by the quotient kernel of the LARGEST size group:
      // that kernel waits on HBM (4 B per LDE cell in, 4.1 TB/s) and its VALU ports are half idle.  "quot_leaf" = 0: separate launch.
      if (tune(T_QUOT_LEAF) != 0 && !qargs.empty() && (qargs.size() == 1 || qargs[1].first.log_size < qargs[0].first.log_size) &&
          quotient_leaf_serves(qargs[0].first)) {
        fri.first_tree.leaf_prealloc.alloc((size_t)32 << qargs[0].first.log_size);
        qargs[0].first.leaf_hashes = fri.first_tree.leaf_prealloc.u32();
        fri.first_leaf_done = true;
      }
      // the small groups (latency-bound column-slice kernels, ~170 us in a row) go FIRST, together on one side stream, so that
      // they hide under the large groups instead of trailing them (they ended the region ~90 us after the last large kernel);
      // the large groups follow by descending size, one stream each
      for (auto& qa : qargs) if (qa.first.log_size < 14) launch_quotients(qa.first, qa.second, fkq.stream(Fork::N - 1));
      for (auto& qa : qargs)
        if (qa.first.log_size >= 14) { launch_quotients(qa.first, qa.second, fkq.stream(qk == 0 ? Fork::main_or(0) : qk % fork_width(Fork::N - 1))); qk++; }
      fkq.join();
      kregq.close();
    }
    P.tick("quotients");
    ht.mark("quotients: gpu done");

  }

  // FRI commit phase (FriPhase), proof of work, and — device-side tail (tail_device.hpp) — the whole rest of the proof
  DeviceTail tail;
  bool tail_done = false;
  void fri_and_pow() {
    // ---- FRI commit (FriPhase::commit) ----
    fri.commit_enqueue(P, cfg, quotients, q_logs);
    const bool dev_tail = DeviceTail::supported(cfg, q_logs, fri);
    // last layer -> PoW -> queries -> decommitment of every tree, enqueued behind the last fold: no host round trip
    if (dev_tail) tail.enqueue(P, fri, quotients, q_logs);
    check_composition_at_oods(pf, tr0, it0, clog, hrel, powers, coff, oods);   // host-only, overlapped with the kernels enqueued above
    ht.mark("fri: enqueued, oods check done");
    // The host now waits ~2 ms for the FRI layers: give back what nothing enqueued later can need — the trace-domain evaluations,
    // the coefficient columns of the four trees, the small per-phase tables — instead of doing it between two proofs with the GPU
    // idle.  (Pool blocks are reused in stream order on this thread's stream: releasing them under running kernels is safe.)
    if (defer_teardown()) {
      tr_evals = ColumnSet();
      if (!pp_cache_enabled()) pp_evals = ColumnSet();
      for (int t = pp_cache_enabled() ? 1 : 0; t < 4; t++) P.trees[t].coeffs = ColumnSet();
      drel = DevBuf(); d_powers = DevBuf(); d_step1 = DevBuf(); d_step2 = DevBuf(); d_ctab = DevBuf(); d_oods_table = DevBuf(); d_qblob = DevBuf();
      ht.mark("fri: early teardown");
    }
    if (dev_tail) {
      tail.wait_last(st);                                // challenges, roots and the last layer are in pinned memory ...
      ht.mark("tail: last layer landed");
      fri.commit_finish(P, cfg, pf, true);               // ... replayed while the proof of work and the tables run
      ht.mark("tail: commit phase replayed");
      tail.wait_tables(st);
      if (tail.nonce_found()) {
        pf.proof_of_work = tail.nonce();
        ch.mix_u64(pf.proof_of_work);
        CM_CHECK(ch.trailing_zeros() >= cfg.pow_bits, "pow: the device's nonce fails the host's check");
        tail.plan(Queries::draw(ch, cfg.n_queries, q_logs[0]));   // host tables while the gathers write the witnesses
        tail_done = true;
        ht.mark("tail: nonce, queries, host tables");
        return;
      }
      // (no nonce within 16x the expected range, probability e^-16: search on with the host-driven form; the decommitment
      // the device made from nothing is discarded)
      CM_HIP(hipStreamSynchronize(st));
    } else {
      fri.commit_finish(P, cfg, pf, false);
      P.tick("fri_commit");
    }
    pf.proof_of_work = grind_gpu(ch.digest.data(), cfg.pow_bits, st);
    ch.mix_u64(pf.proof_of_work);
    if (!dev_tail) P.tick("pow");
    ht.mark("pow done");

  }

  // queries, one batched gather for every tree, proof assembly
  void decommit() {
    if (tail_done) {   // the witnesses land in pinned memory in proof order: wait for the last gather, copy them
      CM_HIP(hipStreamSynchronize(st));
      ht.mark("tail: gpu done");
      tail.copy(P, fri, pf);
      ht.mark("decommit: copied the device tail's witnesses");
      return;
    }
    // ---- queries + decommitment ----
    Queries queries = Queries::draw(ch, cfg.n_queries, q_logs[0]);
    ht.mark("decommit: queries drawn");
    const bool ticked = tail.enqueued;   // (device tail without a nonce: its phase events are already in the stream)
    std::map<uint32_t, std::vector<uint32_t>> qpos;
    for (auto l : q_logs) qpos[l] = queries.fold(queries.log_domain_size - l).positions;
    ht.mark("decommit: qpos");
    {
      // One batched gather for every tree of the proof: FRI first layer, inner layers, the 4 commitment trees.
      GatherBatch gb;
      {  // ~ n_queries x tree depth x (2 siblings) per tree; growing these vectors dominated the planning time
        const size_t per_tree = (size_t)cfg.n_queries * 2 * (q_logs[0] + 2);
        gb.hash_addrs.reserve(per_tree * (fri.inner.size() + 5));
        gb.word_addrs.reserve((size_t)cfg.n_queries * 8 * (fri.inner.size() + 1));
        gb.runs.reserve((size_t)cfg.n_queries * 4 * (q_logs[0] + 2));
      }
      fri.plan_decommit(queries, qpos, quotients, q_logs, gb);
      DecommitPlan tree_plan[4];
      ht.mark("decommit: fri plans");
      for (int t = 0; t < 4; t++) tree_plan[t] = P.trees[t].merkle.plan_decommit(qpos, gb);
      ht.mark("decommit: tree plans");
      gb.run(st);
      ht.mark("decommit: gather run (upload+kernel+d2h)");
      fri.finish_decommit(gb, pf);
      ht.mark("decommit: finish fri");
      pf.decommitments.resize(4);
      pf.queried_values.resize(4);
      for (int t = 0; t < 4; t++) {
        MerkleTree::finish_decommit(tree_plan[t], gb, pf.queried_values[t], pf.decommitments[t]);
        pf.commitments.push_back(P.trees[t].root);
      }
      ht.mark("decommit: finish trees");
    }
    ht.mark("decommit: finish (locals released)");
    if (!ticked) P.tick("decommit");
  }
};
ProofData* prove(const DeviceInput& din, const cm_pcs_config& cfg) {
  static const bool marks = getenv("CM_HOST_MARKS") != nullptr;
  static thread_local std::chrono::steady_clock::time_point last_return;
  static thread_local bool have_last = false;
  auto now = [] { return std::chrono::steady_clock::now(); };
  if (marks && have_last) fprintf(stderr, "[host] %-40s %8.1f us\n", "(between two proofs: caller)", std::chrono::duration<double, std::micro>(now() - last_return).count());
  const auto t0 = now();
  ProofData* r;
  std::chrono::steady_clock::time_point t1;
  {
    SegmentProver sp(din, cfg);
    if (marks) fprintf(stderr, "[host] %-40s %8.1f us\n", "(prover object constructed)", std::chrono::duration<double, std::micro>(now() - t0).count());
    r = sp.run();
    t1 = now();
  }
  if (marks) fprintf(stderr, "[host] %-40s %8.1f us\n", "(prover object destroyed)", std::chrono::duration<double, std::micro>(now() - t1).count());
  last_return = now(); have_last = true;
  return r;
}

#include "prover_sharded.inc"

}  // namespace cm

// ---- segment pipeline: several independent segment proofs in flight on one GPU ---------------
// Continuation segments are independent proofs (runner/src/vm/mod.rs:184-240).  Every worker is a persistent
// host thread with its own main stream, side streams, device pool and upload ring (all thread-local), so the
// launch gaps, host round trips and latency-bound kernels of one proof overlap with the others' work.
namespace cm {
namespace {
struct ProveWorkers {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> jobs;
  std::vector<std::thread> threads;
  void ensure(unsigned n) {
    std::lock_guard<std::mutex> lk(mu);
    while (threads.size() < n) {
      threads.emplace_back([this] {
        bool placed = false;
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk2(mu);
            cv.wait(lk2, [this] { return !jobs.empty(); });
            job = std::move(jobs.front());
            jobs.pop_front();
          }
          // library-owned thread: it stays next to the GPU (cm_init has run by the time a job arrives)
          if (!placed) { bind_worker_thread_cpus(); placed = true; }
          job();
        }
      });
      threads.back().detach();
    }
  }
  void submit(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(f)); }
    cv.notify_one();
  }
};
ProveWorkers& prove_workers() { static ProveWorkers* w = new ProveWorkers(); return *w; }
}  // namespace
}  // namespace cm

namespace cm {
std::string verify_proof(const ProofData& pf, const cm_pcs_config& expected);                 // verifier.hip
bool proof_from_words(const uint32_t* w, uint64_t n, ProofData& p, std::string& err);
}  // namespace cm

// ================================================================= C ABI
struct cm_proof {   // owns its ProofData: every failure path (parse error, exception mid-proof) releases it with the wrapper
  cm::ProofData* d = nullptr; std::string json, transcript_json; std::vector<uint32_t> words;
  ~cm_proof() { delete d; }
};
struct cm_device_input { cm::DeviceInput* d = nullptr; ~cm_device_input() { delete d; } };
extern "C" int32_t cm_set_last_error(const char* msg);

template <class F>
static int32_t pguard(F&& f) {
  try { f(); return 0; }
  catch (const cm::CmError& e) { cm_set_last_error(e.what()); return e.code ? e.code : 1; }
  catch (const std::exception& e) { cm_set_last_error(e.what()); return 1; }
}
static cm_pcs_config default_cfg() { return cm_pcs_config{16, 1, 0, 80}; }

extern "C" {
int32_t cm_input_upload(const cm_prover_input* input, cm_device_input** out) {
  return pguard([&] { std::unique_ptr<cm_device_input> h(new cm_device_input()); h->d = cm::upload_input(*input); *out = h.release(); });
}
struct cm_host_input { cm::host::ProverInputOwned owned; cm_prover_input view; };  // = host_api.hip
int32_t cm_adapt_segment_device(const cm_runner_segment* seg, cm_device_input** out) {
  return pguard([&] { std::unique_ptr<cm_device_input> h(new cm_device_input()); h->d = cm::adapt_segment_device(*seg); *out = h.release(); });
}
int32_t cm_device_input_download(const cm_device_input* in, cm_host_input** out) {
  return pguard([&] {
    cm_host_input* h = new cm_host_input();
    cm::download_input(*in->d, h->owned);
    h->view = h->owned.view();
    *out = h;
  });
}
int32_t cm_input_free(cm_device_input* h) { delete h; return 0; }
int32_t cm_prove_device(const cm_device_input* input, const cm_pcs_config* config, cm_proof** out) {
  return pguard([&] {
    cm_pcs_config cfg = config ? *config : default_cfg();
    std::unique_ptr<cm_proof> p(new cm_proof());
    p->d = cm::prove(*input->d, cfg);
    *out = p.release();
  });
}
int32_t cm_prove_segment(const cm_prover_input* input, const cm_pcs_config* config, cm_proof** out) {
  cm_device_input* di = nullptr;
  int32_t rc = cm_input_upload(input, &di);
  if (rc) return rc;
  rc = cm_prove_device(di, config, out);
  cm_input_free(di);
  return rc;
}
int32_t cm_shard_plan(const cm_prover_input* input, const cm_pcs_config* config, uint32_t world, int32_t owner_out[CM_N_COMPONENTS],
                      uint64_t* staging_words) {
  return pguard([&] {
    CM_CHECK(world >= 1 && world <= 8 && (world & (world - 1)) == 0, "cm_shard_plan: world must be 1, 2, 4 or 8");
    const cm_pcs_config cfg = config ? *config : default_cfg();
    CM_CHECK(cfg.log_blowup_factor >= 1 && cfg.log_blowup_factor <= 4, "cm_shard_plan: log_blowup_factor");
    uint32_t clog[air::N_COMPONENTS];
    cm::component_logs(*input, clog);
    // the SAME plan cm_prove_sharded will run under this config (components are split only at log_blowup_factor 1)
    const cm::ShardPlan p = cm::make_shard_plan(clog, world, cfg.log_blowup_factor == 1);
    for (int c = 0; c < air::N_COMPONENTS; c++) if (owner_out) owner_out[c] = p.owner[c];
    // (the bound is computed for a blowup of 2; the LDE, and with it every exchange, grows by 2^(B - 1))
    if (staging_words) *staging_words = cm::shard_staging_words(clog, p, world) << (cfg.log_blowup_factor - 1);
  });
}
int32_t cm_shard_plan_columns(const cm_prover_input* input, const cm_pcs_config* config, uint32_t world, int32_t* trace_col_owner,
                              uint32_t* n_trace_cols, int32_t* interaction_col_owner, uint32_t* n_interaction_cols, uint64_t load_cells[8]) {
  return pguard([&] {
    CM_CHECK(world >= 1 && world <= 8 && (world & (world - 1)) == 0, "cm_shard_plan_columns: world must be 1, 2, 4 or 8");
    const cm_pcs_config cfg = config ? *config : default_cfg();
    uint32_t clog[air::N_COMPONENTS];
    cm::component_logs(*input, clog);
    const cm::ShardPlan p = cm::make_shard_plan(clog, world, cfg.log_blowup_factor == 1);
    // n_*_cols: capacity of the array on entry, the number of columns on return; an array that is too small is an error, never
    // an overrun
    if (trace_col_owner) {
      CM_CHECK(n_trace_cols && *n_trace_cols >= p.tr_owner.size(), "cm_shard_plan_columns: trace_col_owner is too small (set *n_trace_cols to its capacity)");
      for (size_t j = 0; j < p.tr_owner.size(); j++) trace_col_owner[j] = p.tr_owner[j];
    }
    if (interaction_col_owner) {
      CM_CHECK(n_interaction_cols && *n_interaction_cols >= p.it_owner.size(),
               "cm_shard_plan_columns: interaction_col_owner is too small (set *n_interaction_cols to its capacity)");
      for (size_t j = 0; j < p.it_owner.size(); j++) interaction_col_owner[j] = p.it_owner[j];
    }
    if (n_trace_cols) *n_trace_cols = (uint32_t)p.tr_owner.size();
    if (n_interaction_cols) *n_interaction_cols = (uint32_t)p.it_owner.size();
    if (load_cells) for (int k = 0; k < 8; k++) load_cells[k] = p.load[k];
  });
}
int32_t cm_prove_sharded(const cm_device_input* input, const cm_pcs_config* config, const cm_comm* comm, cm_proof** out) {
  return pguard([&] {
    // the caller's struct may be older / shorter than this library's: copy what its struct_size covers, the rest stays zero
    CM_CHECK(comm && comm->struct_size >= offsetof(cm_comm, flags) && comm->struct_size <= 4096,
             "cm_prove_sharded: cm_comm.struct_size does not cover the required fields (set it to sizeof(cm_comm))");
    cm_comm cc;
    memset(&cc, 0, sizeof(cc));
    memcpy(&cc, comm, std::min<size_t>(comm->struct_size, sizeof(cm_comm)));
    CM_CHECK(cc.all_gather && cc.all_to_all_v && cc.send_buf && cc.recv_buf, "cm_prove_sharded: incomplete cm_comm");
    cm_pcs_config cfg = config ? *config : default_cfg();
    std::unique_ptr<cm_proof> p(new cm_proof());
    try {
      p->d = cm::prove_sharded(*input->d, cfg, cc);
    } catch (...) {
      if (cc.abort) cc.abort(cc.ctx);   // the peers are heading for a collective this rank will never join
      throw;
    }
    *out = p.release();
  });
}
int32_t cm_prove_many(const cm_device_input* const* inputs, uint32_t n, const cm_pcs_config* config, uint32_t inflight,
                      cm_proof** outs) {
  if (!n) return 0;
  if (inflight < 1) inflight = 1;
  if (inflight > 8) inflight = 8;
  cm_pcs_config cfg = config ? *config : default_cfg();
  cm::ProveWorkers& w = cm::prove_workers();
  w.ensure(inflight);
  struct Shared { std::mutex mu; std::condition_variable cv; uint32_t done = 0, next = 0; int32_t rc = 0; std::string err; } sh;
  for (uint32_t i = 0; i < n; i++) outs[i] = nullptr;
  const uint32_t runners = inflight < n ? inflight : n;
  // `runners` jobs, each pulling segment indices until none is left: exactly that many proofs are in flight
  for (uint32_t r = 0; r < runners; r++) {
    w.submit([&] {
      struct InFlight { InFlight() { cm::g_proofs_in_flight.fetch_add(1); } ~InFlight() { cm::g_proofs_in_flight.fetch_sub(1); } } in_flight;
      for (;;) {
        uint32_t i;
        { std::lock_guard<std::mutex> lk(sh.mu); i = sh.next++; }
        if (i >= n) break;
        int32_t rc = 0;
        std::string err;
        try {
          std::unique_ptr<cm_proof> p(new cm_proof());
          p->d = cm::prove(*inputs[i]->d, cfg);
          outs[i] = p.release();
        } catch (const cm::CmError& e) { rc = e.code ? e.code : 1; err = e.what(); }
        catch (const std::exception& e) { rc = 1; err = e.what(); }
        if (rc) { std::lock_guard<std::mutex> lk(sh.mu); if (!sh.rc) { sh.rc = rc; sh.err = err; } }
      }
      std::lock_guard<std::mutex> lk(sh.mu);
      sh.done++;
      sh.cv.notify_all();
    });
  }
  std::unique_lock<std::mutex> lk(sh.mu);
  sh.cv.wait(lk, [&] { return sh.done == runners; });
  lk.unlock();
  if (sh.rc) { cm_set_last_error(sh.err.c_str()); return sh.rc; }
  return 0;
}
// ---- streaming ingest (prover.rs:23-29 takes a HOST `&mut ProverInput`, the reference's bench clones one per
// iteration, benches/prover_speed_benchmark.rs:65) ---------------------------------------------------------------------------
// The calling thread is the PRODUCER: it turns item i into a device-resident input (upload of a host ProverInput, or the device
// adapter on a runner segment) on its own stream and device pool while up to `inflight` worker threads prove the items before
// it.  At most inflight + 1 device inputs are alive (the ones being proved and the one being produced); a spent input goes back
// to the producer, which releases it into ITS pool — the next upload reuses the same blocks, so a steady stream of equal-sized
// segments performs no hipMalloc / hipFree.  The PCIe copies run on the SDMA engines next to the proofs' kernels.
extern "C++" {
// n_producers: threads that turn items into device inputs (the calling thread + n_producers - 1 library threads).  One is enough
// for plain uploads (6 ms of PCIe per 10 ms proof); the device adapter is ~13 ms of uploads, kernels and five host round trips per
// segment while the GPU is busy proving, so runner segments get two.  An input is released by the producer that made it (its
// device pool owns the blocks).
template <class Produce>
static int32_t prove_streamed(uint32_t n, Produce&& produce, const cm_pcs_config* config, uint32_t inflight, cm_proof** outs,
                              uint32_t n_producers = 1) {
  if (!n) return 0;
  if (inflight < 1) inflight = 1;
  if (inflight > 8) inflight = 8;
  if (n_producers < 1) n_producers = 1;
  if (n_producers > 3) n_producers = 3;
  if (n_producers > n) n_producers = n;
  const cm_pcs_config cfg = config ? *config : default_cfg();
  for (uint32_t i = 0; i < n; i++) outs[i] = nullptr;
  cm::ProveWorkers& w = cm::prove_workers();
  w.ensure(inflight + n_producers - 1);
  struct Job { uint32_t i; cm::DeviceInput* d; uint32_t pid; };
  struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> ready;                         // produced, not yet picked up
    std::deque<cm::DeviceInput*> spent[3];         // proved: back to the producer that made them
    uint32_t outstanding[3] = {0, 0, 0};           // inputs of producer p that exist (being produced, ready, being proved, spent)
    uint32_t alive = 0, next_i = 0, workers_done = 0, producers_done = 0, n_producers = 1;
    uint32_t extra_done = 0;                       // library-thread producers that have left `sh` for good (counted under `mu`)
    bool no_more = false, stop = false;
    int32_t rc = 0;
    std::string err;
  } sh;
  sh.n_producers = n_producers;
  const uint32_t cap = inflight + n_producers;     // device inputs alive at once
  const uint32_t runners = inflight < n ? inflight : n;
  for (uint32_t r = 0; r < runners; r++) {
    w.submit([&sh, &cfg, outs] {
      struct InFlight { InFlight() { cm::g_proofs_in_flight.fetch_add(1); } ~InFlight() { cm::g_proofs_in_flight.fetch_sub(1); } } in_flight;
      for (;;) {
        Job job;
        {
          std::unique_lock<std::mutex> lk(sh.mu);
          sh.cv.wait(lk, [&] { return !sh.ready.empty() || sh.no_more; });
          if (sh.ready.empty()) break;
          job = sh.ready.front();
          sh.ready.pop_front();
        }
        int32_t rc = 0;
        std::string err;
        try {
          std::unique_ptr<cm_proof> p(new cm_proof());
          p->d = cm::prove(*job.d, cfg);
          outs[job.i] = p.release();
        } catch (const cm::CmError& e) { rc = e.code ? e.code : 1; err = e.what(); }
        catch (const std::exception& e) { rc = 1; err = e.what(); }
        catch (...) { rc = 1; err = "unknown error"; }
        std::lock_guard<std::mutex> lk(sh.mu);
        if (rc && !sh.rc) { sh.rc = rc; sh.err = err; }
        sh.spent[job.pid].push_back(job.d);
        sh.cv.notify_all();
      }
      std::lock_guard<std::mutex> lk(sh.mu);
      sh.workers_done++;
      sh.cv.notify_all();
    });
  }
  // one producer: takes the next index while a slot is free, makes the device input, hands it to the workers; frees what comes back
  auto producer = [&sh, &produce, n, cap](uint32_t pid) {
    cm::bind_thread_to_library_device();
    auto release_spent = [&](std::unique_lock<std::mutex>& lk) {   // lock held on entry and exit; the frees run outside of it
      std::deque<cm::DeviceInput*> v;
      v.swap(sh.spent[pid]);
      sh.alive -= (uint32_t)v.size();
      sh.outstanding[pid] -= (uint32_t)v.size();
      lk.unlock();
      for (auto* d : v) delete d;
      lk.lock();
      sh.cv.notify_all();
    };
    for (;;) {
      uint32_t i;
      {
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return !sh.spent[pid].empty() || sh.stop || sh.next_i >= n || sh.alive < cap; });
        if (!sh.spent[pid].empty()) { release_spent(lk); continue; }
        if (sh.stop || sh.next_i >= n) break;
        if (sh.alive >= cap) continue;
        i = sh.next_i++;
        sh.alive++;
        sh.outstanding[pid]++;
      }
      cm::DeviceInput* d = nullptr;
      int32_t rc = 0;
      std::string err;
      try { d = produce(i); }
      catch (const cm::CmError& e) { rc = e.code ? e.code : 1; err = e.what(); }
      catch (const std::exception& e) { rc = 1; err = e.what(); }
      catch (...) { rc = 1; err = "unknown error"; }
      std::lock_guard<std::mutex> lk(sh.mu);
      if (!d) {   // this item cannot be made: report it, stop producing (the items already handed over are still proved)
        if (!sh.rc) { sh.rc = rc ? rc : 1; sh.err = err; }
        sh.stop = true;
        sh.alive--;
        sh.outstanding[pid]--;
      } else sh.ready.push_back(Job{i, d, pid});
      sh.cv.notify_all();
    }
    std::unique_lock<std::mutex> lk(sh.mu);
    if (++sh.producers_done == sh.n_producers) sh.no_more = true;
    sh.cv.notify_all();
    // its inputs come back as the workers finish: release them here, on the thread whose pool owns them
    while (sh.outstanding[pid] > 0) {
      sh.cv.wait(lk, [&] { return !sh.spent[pid].empty(); });
      release_spent(lk);
    }
  };
  // (the completion count is bumped INSIDE the lock, like workers_done: the caller's wait below reads it under the same lock, so it
  // cannot return — and destroy `sh` — between the count and the notify; the job touches nothing of `sh` after the guard)
  for (uint32_t pid = 1; pid < n_producers; pid++)
    w.submit([&producer, &sh, pid] { producer(pid); std::lock_guard<std::mutex> lk(sh.mu); sh.extra_done++; sh.cv.notify_all(); });
  {
    cm::AffinityScope cpu_scope;
    producer(0);
  }
  {
    std::unique_lock<std::mutex> lk(sh.mu);
    sh.cv.wait(lk, [&] { return sh.workers_done == runners && sh.extra_done == n_producers - 1; });
  }
  if (sh.rc) { cm_set_last_error(sh.err.c_str()); return sh.rc; }
  return 0;
}
}  // extern "C++"
int32_t cm_prove_many_host(const cm_prover_input* const* inputs, uint32_t n, const cm_pcs_config* config, uint32_t inflight, cm_proof** outs) {
  return prove_streamed(n, [&](uint32_t i) {
    CM_CHECK(inputs && inputs[i], "cm_prove_many_host: null input");
    return cm::upload_input(*inputs[i], cm::thread_main_stream());
  }, config, inflight, outs);
}
int32_t cm_prove_many_segments(const cm_runner_segment* const* segments, uint32_t n, const cm_pcs_config* config, uint32_t inflight,
                               cm_proof** outs) {
  return prove_streamed(n, [&](uint32_t i) {
    CM_CHECK(segments && segments[i], "cm_prove_many_segments: null segment");
    return cm::adapt_segment_device(*segments[i]);
  }, config, inflight, outs, /*n_producers=*/2);
}
// ---- per-component AIR ops (include/cairom_hip.h): the kernels of the whole-segment prover, one component
// at a time on caller-owned columns --------------------------------------------------------------------------------
extern "C++" {
namespace {
inline hipStream_t S_(cm_stream_t s) { return (hipStream_t)(uintptr_t)s; }
inline uint32_t* H_(cm_handle h) { return (uint32_t*)(uintptr_t)h; }
uint32_t component_log_size(const cm_prover_input& in, int c) {
  using namespace cm;
  if (c < air::N_OPCODE_COMPONENTS) return log_size_for(in.n_bundles[c]);
  switch (c) {
    case air::C_MEMORY: return log_size_for(in.n_initial_memory + in.n_final_memory);
    case air::C_MERKLE: case air::C_POSEIDON2: return log_size_for(in.n_initial_tree + in.n_final_tree);
    case air::C_CLOCK_UPDATE: return log_size_for(in.n_clock_updates);
    case air::C_RC8: return 8; case air::C_RC16: return 16; case air::C_RC20: return 20; case air::C_BITWISE: return 18;
  }
  throw CmError(1, "bad component id");
}
std::vector<uint32_t*> handles(const cm_handle* h, int n) {
  std::vector<uint32_t*> v(n);
  for (int i = 0; i < n; i++) { v[i] = H_(h[i]); if (!v[i]) throw cm::CmError(1, "null column handle"); }
  return v;
}
void check_component(int32_t c) { if (c < 0 || c >= air::N_COMPONENTS) throw cm::CmError(1, "bad component id"); }
}  // namespace
}  // extern "C++"

int32_t cm_component_info(int32_t component, uint32_t* n_trace_cols, uint32_t* n_interaction_cols, uint32_t* n_constraints) {
  return pguard([&] {
    check_component(component);
    const air::ComponentInfo& i = air::component_info(component);
    if (n_trace_cols) *n_trace_cols = (uint32_t)i.n_trace;
    if (n_interaction_cols) *n_interaction_cols = (uint32_t)i.n_interaction;
    if (n_constraints) *n_constraints = (uint32_t)i.n_constraints;
  });
}
int32_t cm_component_log_size(const cm_device_input* input, int32_t component, uint32_t* log_size) {
  return pguard([&] { check_component(component); *log_size = component_log_size(input->d->meta, component); });
}
int32_t cm_trace_write(const cm_device_input* input, int32_t c, const cm_handle* cols, cm_stream_t s) {
  return pguard([&] {
    using namespace cm;
    check_component(c);
    CM_CHECK(c <= air::C_POSEIDON2, "cm_trace_write: the lookup-table components' trace is their multiplicity column (cm_histogram)");
    bind_thread_to_library_device();
    const DeviceInput& d = *input->d;
    const cm_prover_input& in = d.meta;
    const uint32_t lg = component_log_size(in, c);
    hipStream_t st = S_(s);
    DevBuf tab = upload(handles(cols, air::component_info(c).n_trace), st);
    uint32_t* const* dc = tab.as<uint32_t*>();
    if (c < air::N_OPCODE_COMPONENTS) launch_opcode_trace(c, d.bundles[c].p, (uint32_t)in.n_bundles[c], d.data_accesses.p, lg, dc, st);
    else if (c == air::C_MEMORY)
      launch_memory_trace(d.init_mem.p, (uint32_t)in.n_initial_memory, d.fin_mem.p, (uint32_t)in.n_final_memory, in.initial_root,
                          in.final_root, lg, dc, st);
    else if (c == air::C_MERKLE)
      launch_merkle_trace(d.init_tree.p, (uint32_t)in.n_initial_tree, d.fin_tree.p, (uint32_t)in.n_final_tree, in.initial_root,
                          in.final_root, lg, dc, st);
    else if (c == air::C_CLOCK_UPDATE) launch_clock_update_trace(d.clock_updates.p, (uint32_t)in.n_clock_updates, lg, dc, st);
    else launch_poseidon2_trace(d.init_tree.p, (uint32_t)in.n_initial_tree, d.fin_tree.p, (uint32_t)in.n_final_tree, lg, dc, st);
    CM_HIP(hipStreamSynchronize(st));
  });
}
int32_t cm_histogram(int32_t c, const cm_handle* trace_cols, uint32_t log_size, cm_handle rc8, cm_handle rc16, cm_handle rc20,
                     cm_handle bitwise, cm_stream_t s) {
  return pguard([&] {
    using namespace cm;
    CM_CHECK(c >= 0 && c < air::N_OPCODE_COMPONENTS, "cm_histogram: opcode components only");
    bind_thread_to_library_device();
    hipStream_t st = S_(s);
    DevBuf tab = upload(handles(trace_cols, air::component_info(c).n_trace), st), flag(4);
    CM_HIP(hipMemsetAsync(flag.p, 0, 4, st));
    HistPtrs h;
    h.rc8 = H_(rc8); h.rc16 = H_(rc16); h.rc20 = H_(rc20); h.bitwise = H_(bitwise); h.error_flag = flag.u32();
    CM_CHECK(h.rc8 && h.rc16 && h.rc20 && h.bitwise, "cm_histogram: null multiplicity column");
    launch_hist(c, (const uint32_t* const*)tab.as<uint32_t*>(), log_size, h, st);
    uint32_t f = 0;
    CM_HIP(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    CM_CHECK(f == 0, "cm_histogram: a range-check / bitwise lookup value is out of range");
  });
}
int32_t cm_preprocessed_column(int32_t id, cm_handle col, cm_stream_t s) {
  return pguard([&] {
    using namespace cm;
    CM_CHECK(id >= 0 && id < air::N_PREPROC && col, "cm_preprocessed_column: bad id / null column");
    bind_thread_to_library_device();
    launch_preproc(id, air::PREPROC_LOG[id], H_(col), S_(s));
    CM_HIP(hipStreamSynchronize(S_(s)));
  });
}
int32_t cm_interaction_write(int32_t c, const cm_handle* trace_cols, const cm_handle* preprocessed, uint32_t log_size,
                             const cm_relations* relations, const cm_handle* out, uint32_t claimed_sum[4], cm_stream_t s) {
  return pguard([&] {
    using namespace cm;
    check_component(c);
    static_assert(sizeof(cm_relations) == sizeof(DevRelations), "cm_relations must mirror DevRelations");
    bind_thread_to_library_device();
    hipStream_t st = S_(s);
    const air::ComponentInfo& info = air::component_info(c);
    std::vector<uint32_t*> oc = handles(out, info.n_interaction);
    UploadBatch ub;
    uint32_t** d_tr = nullptr; uint32_t** d_pp = nullptr; uint32_t** d_out = nullptr;
    ub.add(handles(trace_cols, info.n_trace), &d_tr);
    ub.add(handles(preprocessed, air::N_PREPROC), &d_pp);
    ub.add(oc, &d_out);
    DevBuf tabs = ub.flush(st), drel(sizeof(DevRelations)), d_sums(16);
    stage_upload(drel.p, relations, sizeof(DevRelations), st);
    launch_logup(c, (const uint32_t* const*)d_tr, (const uint32_t* const*)d_pp, log_size, drel.as<DevRelations>(), d_out, st);
    std::vector<LogupTailJob> jobs(1);
    for (int k = 0; k < 4; k++) jobs[0].col[k] = oc[info.n_interaction - 4 + k];
    jobs[0].log_size = log_size;
    logup_finalize_all(jobs, d_sums.u32(), st);
    CM_HIP(hipMemcpyAsync(claimed_sum, d_sums.p, 16, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
  });
}
int32_t cm_constraints_accumulate(int32_t c, const cm_handle* trace_lde, const cm_handle* interaction_lde,
                                  const cm_handle* preprocessed_lde, uint32_t log_size, const cm_relations* relations,
                                  const uint32_t* coeff_powers, const uint32_t claimed_sum[4], const cm_handle acc[4],
                                  cm_stream_t s) {
  return pguard([&] {
    using namespace cm;
    check_component(c);
    bind_thread_to_library_device();
    hipStream_t st = S_(s);
    const air::ComponentInfo& info = air::component_info(c);
    UploadBatch ub;
    uint32_t** d_tr = nullptr; uint32_t** d_it = nullptr; uint32_t** d_pp = nullptr; uint32_t** d_acc = nullptr;
    uint32_t* d_coeff = nullptr;
    ub.add(handles(trace_lde, info.n_trace), &d_tr);
    ub.add(handles(interaction_lde, info.n_interaction), &d_it);
    ub.add(handles(preprocessed_lde, air::N_PREPROC), &d_pp);
    ub.add(handles(acc, 4), &d_acc);
    ub.add(std::vector<uint32_t>(coeff_powers, coeff_powers + 4 * (size_t)info.n_constraints), &d_coeff);
    DevBuf tabs = ub.flush(st), drel(sizeof(DevRelations));
    stage_upload(drel.p, relations, sizeof(DevRelations), st);
    ConstraintArgs a;
    a.tr = (const uint32_t* const*)d_tr; a.it = (const uint32_t* const*)d_it; a.pp = (const uint32_t* const*)d_pp;
    a.rels = drel.as<DevRelations>();
    a.coeff = d_coeff;
    a.acc = d_acc;
    a.log_size = log_size;
    a.n_base = info.n_base_constraints;
    (QM31::from_u32(claimed_sum) * inv(M31::from_u32(1u << log_size))).to_u32(a.cumsum_shift);
    for (uint32_t k = 0; k < 2; k++) {
      CPoint<M31> pt = point_at_index(domain_index_at(log_size + 1, k));
      a.denom_inv[k] = inv(coset_vanishing_canonic<M31>(log_size, pt)).v;
    }
    launch_constraints(c, a, st);
    CM_HIP(hipStreamSynchronize(st));
  });
}
// verify_cairo_m (crates/prover/src/verifier.rs:17-95): 0 = accepted; status 11 + cm_last_error() = name of the failed check
int32_t cm_verify_proof(const cm_proof* p, const cm_pcs_config* expected) {
  return pguard([&] {
    std::string e = cm::verify_proof(*p->d, expected ? *expected : default_cfg());
    if (!e.empty()) throw cm::CmError(11, "verification failed: " + e);
  });
}
int32_t cm_verify_proof_words(const uint32_t* words, uint64_t n_words, const cm_pcs_config* expected) {
  return pguard([&] {
    cm::ProofData pd;
    std::string e;
    if (!cm::proof_from_words(words, n_words, pd, e)) throw cm::CmError(11, "verification failed: " + e);
    e = cm::verify_proof(pd, expected ? *expected : default_cfg());
    if (!e.empty()) throw cm::CmError(11, "verification failed: " + e);
  });
}
// proof object from its flat word stream (cm_proof_words format): host code, no GPU needed — lets a verifier-side process
// re-serialise a received proof (cm_proof_json) or hand it to cm_verify_proof
int32_t cm_proof_from_words(const uint32_t* words, uint64_t n_words, cm_proof** out) {
  return pguard([&] {
    std::unique_ptr<cm_proof> p(new cm_proof());
    p->d = new cm::ProofData();
    std::string e;
    if (!cm::proof_from_words(words, n_words, *p->d, e)) throw cm::CmError(11, "cm_proof_from_words: " + e);
    *out = p.release();
  });
}
int32_t cm_proof_free(cm_proof* p) { delete p; return 0; }
int32_t cm_proof_json(const cm_proof* p, const char** json_out, size_t* len_out) {
  return pguard([&] {
    cm_proof* q = const_cast<cm_proof*>(p);
    if (q->json.empty()) q->json = cm::proof_to_json(*p->d);
    *json_out = q->json.c_str();
    *len_out = q->json.size();
  });
}
int32_t cm_set_transcript_log(int32_t on) { cm::g_transcript_log.store(on ? 1 : 0); return 0; }
int32_t cm_proof_transcript(const cm_proof* p, const char** json_out, size_t* len_out) {
  return pguard([&] {
    cm_proof* q = const_cast<cm_proof*>(p);
    if (q->transcript_json.empty()) q->transcript_json = cm::transcript_to_json(p->d->transcript);
    *json_out = q->transcript_json.c_str();
    *len_out = q->transcript_json.size();
  });
}
int32_t cm_proof_words(const cm_proof* p, const uint32_t** words_out, uint64_t* n_out) {
  return pguard([&] {
    cm_proof* q = const_cast<cm_proof*>(p);
    if (q->words.empty()) q->words = cm::proof_to_words(*p->d);
    *words_out = q->words.data();
    *n_out = q->words.size();
  });
}
int32_t cm_proof_commitments(const cm_proof* p, uint8_t roots[4][32]) {
  for (int t = 0; t < 4; t++) memcpy(roots[t], p->d->commitments[t].data(), 32);
  return 0;
}
int32_t cm_kprof_enable(int32_t on) {
  cm::KProf::get().reset();
  cm::KProf::get().on = on != 0;
  cm::KProf::get().only.clear();
  return 0;
}
// time only one kernel class (name as reported by cm_kprof_report); NULL / "" = all classes
int32_t cm_set_preprocessed_cache(int32_t on) {
  cm::g_pp_cache.store(on ? 1 : 0, std::memory_order_relaxed);
  return 0;
}
int32_t cm_set_tuning(const char* key, int32_t value) {
  if (!key) return cm_set_last_error("cm_set_tuning: null key");
  for (int k = 0; k < cm::T_COUNT; k++)
    if (strcmp(cm::TUNE_TABLE[k].key, key) == 0) {
      if (value < cm::TUNE_TABLE[k].lo || value > cm::TUNE_TABLE[k].hi) return cm_set_last_error("cm_set_tuning: value out of the key's range");
      cm::tune_values()[k].store(value);
      return 0;
    }
  return cm_set_last_error("cm_set_tuning: unknown key");
}
int32_t cm_set_device_tail(int32_t on) {
  cm::g_device_tail.store(on ? 1 : 0);
  return 0;
}
int32_t cm_tail_list(const uint32_t* positions, uint32_t n_positions, uint32_t log_domain, uint32_t qmask, uint32_t list, uint32_t k,
                     uint32_t* out, uint32_t cap, uint32_t* n_out) {
  return pguard([&] {
    CM_CHECK(log_domain < cm::TAIL_MAX_SHIFTS && k <= log_domain && list <= 2, "cm_tail_list: bad arguments");
    CM_CHECK((positions || n_positions == 0) && n_out && (out || cap == 0), "cm_tail_list: null argument");
    for (uint32_t i = 0; i < n_positions; i++)   // TailTables::build assumes sorted, de-duplicated positions inside the domain
      CM_CHECK((positions[i] >> log_domain) == 0 && (i == 0 || positions[i - 1] < positions[i]), "cm_tail_list: positions must be strictly increasing and below 2^log_domain");
    cm::TailTables tt;
    tt.build(std::vector<uint32_t>(positions, positions + n_positions), log_domain, qmask);
    const uint32_t* v = tt.list(list, k);
    const uint32_t nv = tt.cnt[list][k];
    for (uint32_t i = 0; i < nv && i < cap; i++) out[i] = v[i];
    *n_out = nv;
  });
}
int32_t cm_set_twiddle_cache(int32_t on) {
  cm::g_tw_cache.store(on ? 1 : 0, std::memory_order_relaxed);
  return 0;
}
int32_t cm_kprof_filter(const char* name) {
  cm::KProf::get().only = name ? name : "";
  return 0;
}
// JSON: {"kernel": {"calls": n, "ms": total, "bytes": algorithmic bytes}, ...}
int32_t cm_kprof_report(char* buf, size_t buf_len) {
  (void)hipDeviceSynchronize();
  cm::KProf& k = cm::KProf::get();
  k.flush();
  std::string s = "{";
  bool first = true;
  for (auto& kv : k.agg) {
    if (!first) s += ",";
    first = false;
    s += "\"" + kv.first + "\":{\"calls\":" + std::to_string(kv.second.calls) + ",\"ms\":" + std::to_string(kv.second.ms) +
         ",\"bytes\":" + std::to_string(kv.second.bytes) + ",\"work\":" + std::to_string(kv.second.work) + "}";
  }
  s += "}";
  if (buf && buf_len) { size_t n = s.size() < buf_len - 1 ? s.size() : buf_len - 1; memcpy(buf, s.data(), n); buf[n] = 0; }
  return (int32_t)s.size();
}
int32_t cm_proof_stats(const cm_proof* p, uint64_t* cells, uint64_t* steps, double* phase_ms, uint32_t n_phases) {
  if (cells) *cells = p->d->cells;
  if (steps) *steps = p->d->steps;
  for (uint32_t i = 0; i < n_phases; i++) phase_ms[i] = i < p->d->phase_ms.size() ? p->d->phase_ms[i] : 0.0;
  return (int32_t)p->d->phase_ms.size();
}
}
