// Device-side adapter: `import_from_runner_output` (/root/reference/crates/prover/src/adapter/mod.rs:97-193,
// adapter/memory.rs:271-535) for one runner segment, with the O(steps) part on the GPU.
//
// The reference walks the memory log sequentially through a HashMap<address, (value, clock, multiplicity)>
// to find, for every access, the previous access of the same cell.  Here:
//   1. per step: opcode from the (immutable) program words at `pc`, entry / operand counts  -> exclusive scans
//      give every step its slice of the memory log (checked against the log: first entry address == pc);
//   2. all log entries are sorted by (address, entry index) with one 64-bit radix sort (hipCUB): the
//      predecessor in the sorted order IS the previous access of the cell (or the segment's initial memory);
//   3. clock-update rows (gaps > 2^20 - 1) are counted per entry, scanned and scattered in log order;
//   4. steps are bucketed per (opcode component, opcode) with a stable 11-bit radix sort, bundles and data accesses are
//      written straight into the device-resident ProverInput;
//   5. one record per touched cell (first value, last value, last clock) is compacted for the host, which
//      builds the boundary-memory rows, public multiplicities and the two partial Merkle trees (small:
//      O(touched cells), shared with the host adapter).
// Output order is identical to the host adapter (host_adapter.hpp) — tests compare the two field by field.
#include "../../include/cairom_hip.h"
#include "engine.hpp"
#include "host_adapter.hpp"
#include <hipcub/hipcub.hpp>
#include <stdlib.h>

namespace cm {

struct DeviceInput;  // prover.hip
DeviceInput* make_device_input(const cm_prover_input& meta_host_small, DevBuf (&bundles)[CM_N_OPCODE_COMPONENTS], DevBuf& data_accesses,
                               DevBuf& clock_updates, DevBuf* init_tree_dev, DevBuf* fin_tree_dev);

namespace {

struct OpTable { uint8_t size[64], acc[64], comp[64], valid[64]; };

// per step: info = component << 14 | opcode << 8 | entries << 4 | operand accesses (bits 8..18 are the step sort's key: inside a
// component the reference concatenates the bundles of its opcode variants in the order of `define_opcodes!`
// (components/opcodes/mod.rs:51-58, 223-268), which is ascending opcode id for every group); cnt = entries | accesses << 32,
// scanned in place into the step's offsets in the memory log and in the data-access array
constexpr uint32_t INFO_KEY_LO = 8, INFO_KEY_HI = 19;
__device__ __forceinline__ uint32_t info_comp(uint32_t info) { return info >> 14; }
__device__ __forceinline__ uint32_t info_ne(uint32_t info) { return (info >> 4) & 15u; }
__device__ __forceinline__ uint32_t info_na(uint32_t info) { return info & 15u; }
// the memory at segment start: the locals (address i) and the heap (index i = address MAX_ADDRESS - i), runner/src/vm/mod.rs:205-221
struct InitMem {
  const uint32_t* lo; uint32_t n_lo;
  const uint32_t* hi; uint32_t n_hi;
  __device__ __forceinline__ const uint32_t* cell(uint32_t addr) const {
    if (addr < n_lo) return lo + 4 * (size_t)addr;
    const uint32_t h = host::MAX_ADDRESS - addr;   // (addr > MAX_ADDRESS wraps to a huge index: no cell)
    if (h < n_hi) return hi + 4 * (size_t)h;
    return nullptr;
  }
};
__global__ void k_step_counts(const uint32_t* __restrict__ trace, uint32_t n_steps, const uint32_t* __restrict__ init_mem,
                              uint32_t n_init, OpTable tab, uint32_t* __restrict__ info, unsigned long long* __restrict__ cnt,
                              uint32_t* __restrict__ err) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_steps) return;
  uint32_t pc = trace[2 * t];
  uint32_t op = pc < n_init ? init_mem[4 * (size_t)pc] : 64u;
  if (op >= 64u || !tab.valid[op]) { atomicOr(err, 1u); info[t] = 0; cnt[t] = 0; return; }
  const uint32_t na = tab.acc[op], ne = 1u + (tab.size[op] > 4 ? 1u : 0u) + na;
  info[t] = ((uint32_t)tab.comp[op] << 14) | (op << 8) | (ne << 4) | na;
  cnt[t] = (unsigned long long)ne | ((unsigned long long)na << 32);
}
// per log entry e: its address (the sort key), its own index (the sort payload) and its clock = step + 1; the largest address
// bounds the sort's bit range
__global__ void k_entry_keys(const uint32_t* __restrict__ trace, uint32_t n_steps, const unsigned long long* __restrict__ off,
                             const uint32_t* __restrict__ info, const uint32_t* __restrict__ mem /*5 words each*/, uint32_t n_mem,
                             uint32_t* __restrict__ addr_key, uint32_t* __restrict__ entry_id, uint32_t* __restrict__ entry_clock,
                             uint32_t* __restrict__ err /*[0] flags, [1] largest address*/) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t amax = 0;
  if (t < n_steps) {
    uint32_t e0 = (uint32_t)off[t], n = info_ne(info[t]);
    if (e0 + n > n_mem) { atomicOr(err, 2u); n = 0; }
    if (n && mem[5 * (size_t)e0] != trace[2 * t]) atomicOr(err, 4u);  // first entry of a step is the fetch at pc
    for (uint32_t k = 0; k < n; k++) {
      uint32_t e = e0 + k, a = mem[5 * (size_t)e];
      addr_key[e] = a;
      entry_id[e] = e;
      entry_clock[e] = t + 1;
      amax = a > amax ? a : amax;
    }
  }
  for (int d = 32; d; d >>= 1) { uint32_t o = __shfl_xor(amax, d); amax = o > amax ? o : amax; }
  // (one address: only waves that would raise it go to the atomic unit — 65 K same-address atomics took 0.6 ms)
  if ((threadIdx.x & 63u) == 0 && amax > __atomic_load_n(err + 1, __ATOMIC_RELAXED)) atomicMax(err + 1, amax);
}
// sorted position i (by address; entries of one cell stay in log order: the sort is stable) -> previous access of the same cell.
// link[e] = (previous clock, adjusted by the clock-update rows in between; word 0 of the previous value).  Entries further than
// RC20_LIMIT clocks from their predecessor are rare: only they write cu_count[e] (zeroed by the caller) and their position.
__global__ void k_prev_links(const uint32_t* __restrict__ sorted_addr, const uint32_t* __restrict__ sorted_e, uint32_t n_mem,
                             const uint32_t* __restrict__ mem, const uint32_t* __restrict__ entry_clock,
                             InitMem init, uint2* __restrict__ link,
                             uint32_t* __restrict__ cu_count, uint32_t* __restrict__ cu_pos, uint32_t* __restrict__ head_flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
  const bool live = i < n_mem;
  const uint32_t addr = live ? sorted_addr[i] : 0u, e = live ? sorted_e[i] : 0u;
  // this entry's own clock and value word (two reads at a log-ordered index); the next position takes them from here
  const uint32_t clk = live ? entry_clock[e] : 0u, v0 = live ? mem[5 * (size_t)e + 1] : 0u;
  uint32_t paddr = __shfl_up(addr, 1), pclk = __shfl_up(clk, 1), pv0 = __shfl_up(v0, 1);
  if (lane == 0 && live && i > 0) {
    const uint32_t pe = sorted_e[i - 1];
    paddr = sorted_addr[i - 1]; pclk = entry_clock[pe]; pv0 = mem[5 * (size_t)pe + 1];
  }
  if (!live) return;
  const bool head = i == 0 || paddr != addr;
  head_flag[i] = head ? 1u : 0u;
  if (head) {
    pclk = 0;
    const uint32_t* ic = init.cell(addr);
    pv0 = ic ? ic[0] : v0;
  }
  const uint32_t delta = clk - pclk;
  const uint32_t steps = delta > air::RC20_LIMIT ? delta / air::RC20_LIMIT : 0u;
  if (steps) { cu_count[e] = steps; cu_pos[e] = i; }
  link[e] = make_uint2(pclk + steps * air::RC20_LIMIT, pv0);
}
// clock-update rows in log order (cu_off = exclusive scan of cu_count).  The value of a cell outside the initial memory is the
// value of its FIRST access: the head of its run in the sorted order, found by bisection from the entry's own position.
__global__ void k_clock_updates(const uint32_t* __restrict__ mem, uint32_t n_mem, const uint32_t* __restrict__ cu_count,
                                const uint32_t* __restrict__ cu_off, const uint32_t* __restrict__ cu_pos, const uint2* __restrict__ link,
                                const uint32_t* __restrict__ sorted_addr, const uint32_t* __restrict__ sorted_e,
                                InitMem init, cm_clock_update* __restrict__ out) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_mem) return;
  uint32_t n = cu_count[e];
  if (!n) return;
  uint32_t addr = mem[5 * (size_t)e];
  const uint32_t* iv = init.cell(addr);
  if (!iv) {
    uint32_t lo = 0, hi = cu_pos[e];          // first position whose address is `addr`
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (sorted_addr[mid] < addr) lo = mid + 1; else hi = mid; }
    iv = mem + 5 * (size_t)sorted_e[lo] + 1;
  }
  uint32_t pclk = link[e].x - n * air::RC20_LIMIT;
  for (uint32_t k = 0; k < n; k++) {
    cm_clock_update u;
    u.address = addr; u.prev_clock = pclk;
    for (int j = 0; j < 4; j++) u.value[j] = iv[j];
    out[cu_off[e] + k] = u;
    pclk += air::RC20_LIMIT;
  }
}
__global__ void k_iota(uint32_t* p, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}
// sorted[] is non-decreasing in the key bits (component, opcode): ends[c] = index after the last step of component c
__global__ void k_run_ends(const uint32_t* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ ends) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i + 1 == n || info_comp(sorted[i + 1]) != info_comp(sorted[i])) ends[info_comp(sorted[i])] = i + 1;
}
struct BundleDst { cm_bundle* p[CM_N_OPCODE_COMPONENTS]; uint32_t start[CM_N_OPCODE_COMPONENTS + 1]; };
// position of every step in the component-sorted (stable) order
__global__ void k_step_pos(const uint32_t* __restrict__ sorted_step, uint32_t n_steps, uint32_t* __restrict__ pos) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_steps) pos[sorted_step[i]] = i;
}
// One thread per step, in LOG order: every read (trace, offsets, log entries, links) and the data-access rows are sequential;
// only the 48-byte bundle goes to its component's array (the other way round — threads in bundle order — gathers one useful
// element per cache line from all of those: 0.47 ms against 0.3 at 4.2 M steps).
__global__ void k_bundles(const uint32_t* __restrict__ step_pos, uint32_t n_steps, const uint32_t* __restrict__ trace,
                          const uint32_t* __restrict__ info, const unsigned long long* __restrict__ off,
                          const uint32_t* __restrict__ mem, const uint2* __restrict__ link, BundleDst dst,
                          cm_data_access* __restrict__ accesses, OpTable tab) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_steps) return;
  const uint32_t i = step_pos[t], inf = info[t], c = info_comp(inf);
  const unsigned long long o = off[t];
  const uint32_t e0 = (uint32_t)o, a0 = (uint32_t)(o >> 32), ne = info_ne(inf), na = info_na(inf);
  uint32_t ninst = ne - na;  // 1 or 2 instruction-word entries
  cm_bundle b;
  b.pc = trace[2 * t]; b.fp = trace[2 * t + 1]; b.clock = t + 1; b.inst_prev_clock = link[e0].x;
  const uint32_t* w = mem + 5 * (size_t)e0 + 1;
  const uint32_t sz = tab.size[w[0] & 63u];  // only the instruction's own words are kept (adapter/mod.rs:132-150)
#pragma unroll
  for (int k = 0; k < 4; k++) b.inst[k] = (uint32_t)k < sz ? w[k] : 0u;
  b.inst[4] = sz > 4 ? w[5] : 0u;
  b.inst[5] = sz > 5 ? w[6] : 0u;
  b.span_start = a0; b.span_len = na;
  static_assert(sizeof(cm_bundle) == 48 && sizeof(cm_data_access) == 16, "rows are written as 16-byte words");
  uint4* bo = reinterpret_cast<uint4*>(dst.p[c] + (i - dst.start[c]));   // (pool blocks are 256-byte aligned)
  bo[0] = make_uint4(b.pc, b.fp, b.clock, b.inst_prev_clock);
  bo[1] = make_uint4(b.inst[0], b.inst[1], b.inst[2], b.inst[3]);
  bo[2] = make_uint4(b.inst[4], b.inst[5], b.span_start, b.span_len);
  for (uint32_t k = 0; k < na; k++) {
    uint32_t e = e0 + ninst + k;
    const uint2 l = link[e];
    reinterpret_cast<uint4*>(accesses)[a0 + k] = make_uint4(mem[5 * (size_t)e], l.x, l.y, mem[5 * (size_t)e + 1]);
  }
}
// one record per touched cell: (address, first entry, last entry), compacted by the head flags' exclusive scan
struct CellRec { uint32_t addr, first_entry, last_entry, last_clock; };
__global__ void k_cells(const uint32_t* __restrict__ sorted_addr, const uint32_t* __restrict__ sorted_e, uint32_t n,
                        const uint32_t* __restrict__ head_flag, const uint32_t* __restrict__ head_rank /*exclusive scan of head_flag*/,
                        const uint32_t* __restrict__ entry_clock, CellRec* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t addr = sorted_addr[i];
  bool tail = i + 1 == n || sorted_addr[i + 1] != addr;
  const uint32_t run = head_rank[i] - (head_flag[i] ? 0u : 1u);  // exclusive scan counts this run's head for i > head
  if (head_flag[i]) { out[run].addr = addr; out[run].first_entry = sorted_e[i]; }
  if (tail) { out[run].last_entry = sorted_e[i]; out[run].last_clock = entry_clock[sorted_e[i]]; }
}

template <class F>
void with_temp(F&& f) {  // hipCUB two-phase calls
  size_t bytes = 0;
  f(nullptr, bytes);
  DevBuf tmp(bytes ? bytes : 4);
  f(tmp.p, bytes);
}
inline dim3 grid1(uint32_t n) { return dim3((n + 255) / 256); }

// ---- partial Merkle tree over the boundary memory, on the GPU (adapter/merkle.rs:183-295) ---------------------
// The host builder hashes O(cells * 30) Poseidon2 nodes sequentially: fine for fibonacci (tens of cells), minutes
// for a program that touches 10^6 cells.  Level by level, without host round trips: entries (index, value, mult)
// sorted by index; an entry is the first of its sibling pair iff its parent differs from its predecessor's; an
// exclusive scan of those flags gives every parent its slot; one thread per pair hashes it (absent sibling =
// default hash of that depth) and emits the NodeData row + the parent entry.  Node order = depth 30..1,
// ascending index, exactly like the host builder.
struct TreeState { uint32_t n, node_off; };   // device: entries at the current depth, nodes written so far
__global__ void k_tree_flags(const uint32_t* __restrict__ idx, const TreeState* __restrict__ stt, uint32_t cap, uint32_t* __restrict__ first) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  first[i] = (i < stt->n && (i == 0 || (idx[i] >> 1) != (idx[i - 1] >> 1))) ? 1u : 0u;
}
__global__ void k_tree_level(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val, const uint32_t* __restrict__ mult,
                             const uint32_t* __restrict__ first, const uint32_t* __restrict__ ppos, const TreeState* __restrict__ stt,
                             uint32_t cap, uint32_t depth, uint32_t dflt, cm_merkle_node* __restrict__ nodes,
                             uint32_t* __restrict__ nidx, uint32_t* __restrict__ nval, uint32_t* __restrict__ nmult) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap || i >= stt->n || !first[i]) return;
  const uint32_t index = idx[i];
  uint32_t lv = dflt, lm = 0, rv = dflt, rm = 0;
  if ((index & 1u) == 0) {
    lv = val[i]; lm = mult[i];
    if (i + 1 < stt->n && idx[i + 1] == index + 1) { rv = val[i + 1]; rm = mult[i + 1]; }
  } else {
    rv = val[i]; rm = mult[i];
  }
  const uint32_t ph = host::poseidon2_hash(lv, rv);
  cm_merkle_node nd;
  nd.index = index & ~1u; nd.depth = depth; nd.left_value = lv; nd.right_value = rv; nd.parent_value = ph;
  nd.left_mult = lm; nd.right_mult = rm; nd.parent_mult = 1u;
  const uint32_t slot = ppos[i];
  nodes[stt->node_off + slot] = nd;
  nidx[slot] = index >> 1; nval[slot] = ph; nmult[slot] = 1u;
}
__global__ void k_tree_advance(TreeState* stt, const uint32_t* __restrict__ first, const uint32_t* __restrict__ ppos) {
  if (threadIdx.x || blockIdx.x) return;
  const uint32_t n = stt->n;
  const uint32_t parents = n ? ppos[n - 1] + first[n - 1] : 0u;
  stt->node_off += parents;
  stt->n = parents;
}
// leaves: (address << 2 | i, value_i, mult) sorted by address; returns the root; nodes + count stay on the device
uint32_t build_partial_merkle_tree_device(const std::vector<uint32_t>& idx_h, const std::vector<uint32_t>& val_h,
                                          const std::vector<uint32_t>& mult_h, DevBuf& nodes_out, uint64_t& n_nodes, hipStream_t st) {
  const uint32_t cap = (uint32_t)idx_h.size();
  CM_CHECK(cap > 0, "partial merkle tree: no leaves");
  const std::vector<uint32_t>& dflt = host::poseidon2_default_hashes();
  DevBuf a_idx = upload(idx_h, st), a_val = upload(val_h, st), a_mult = upload(mult_h, st);
  DevBuf b_idx((size_t)cap * 4), b_val((size_t)cap * 4), b_mult((size_t)cap * 4), d_first((size_t)cap * 4), d_ppos((size_t)cap * 4), d_st(sizeof(TreeState));
  // every level has at most as many nodes as the one below: cap * TREE_HEIGHT bounds the total
  nodes_out.alloc((size_t)cap * air::TREE_HEIGHT * sizeof(cm_merkle_node));
  TreeState s0{cap, 0};
  stage_upload(d_st.p, &s0, sizeof(s0), st);
  DevBuf* cur[3] = {&a_idx, &a_val, &a_mult};
  DevBuf* nxt[3] = {&b_idx, &b_val, &b_mult};
  size_t tmp_bytes = 0;
  CM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_first.u32(), d_ppos.u32(), (int)cap, st));
  DevBuf tmp(tmp_bytes ? tmp_bytes : 4);
  for (uint32_t depth = air::TREE_HEIGHT; depth >= 1; depth--) {
    hipLaunchKernelGGL(k_tree_flags, grid1(cap), dim3(256), 0, st, cur[0]->u32(), d_st.as<TreeState>(), cap, d_first.u32());
    CM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, d_first.u32(), d_ppos.u32(), (int)cap, st));
    hipLaunchKernelGGL(k_tree_level, grid1(cap), dim3(256), 0, st, cur[0]->u32(), cur[1]->u32(), cur[2]->u32(), d_first.u32(), d_ppos.u32(),
                       d_st.as<TreeState>(), cap, depth, dflt[depth], nodes_out.as<cm_merkle_node>(), nxt[0]->u32(), nxt[1]->u32(), nxt[2]->u32());
    hipLaunchKernelGGL(k_tree_advance, dim3(1), dim3(64), 0, st, d_st.as<TreeState>(), d_first.u32(), d_ppos.u32());
    for (int k = 0; k < 3; k++) std::swap(cur[k], nxt[k]);
  }
  TreeState fin;
  uint32_t root = 0;
  CM_HIP(hipMemcpyAsync(&fin, d_st.p, sizeof(fin), hipMemcpyDeviceToHost, st));
  CM_HIP(hipMemcpyAsync(&root, cur[1]->p, 4, hipMemcpyDeviceToHost, st));
  CM_HIP(hipGetLastError());
  CM_HIP(hipStreamSynchronize(st));
  CM_CHECK(fin.n == 1, "partial merkle tree: did not converge to one root");
  n_nodes = fin.node_off;
  return root;
}


}  // namespace

// host tail shared with the host adapter: boundary memory rows, public multiplicities, partial Merkle trees
DeviceInput* adapt_segment_device(const cm_runner_segment& seg) {
  bind_thread_to_library_device();
  hipStream_t st = thread_main_stream();
  CM_CHECK(seg.n_trace >= 2, "adapter: empty trace");
  const uint32_t n_steps = (uint32_t)(seg.n_trace - 1), n_mem = (uint32_t)seg.n_memory_trace, n_init = (uint32_t)seg.n_initial_memory;
  CM_CHECK(seg.n_memory_trace < (1ull << 32) && seg.n_trace < (1ull << 32), "adapter: segment too large");
  CM_CHECK(seg.n_memory_trace >= 1, "adapter: empty memory trace");
  // ---- upload the runner output ----
  const uint32_t n_heap = (uint32_t)seg.n_initial_heap;
  CM_CHECK(seg.n_initial_memory + seg.n_initial_heap <= (uint64_t)host::MAX_ADDRESS + 1, "adapter: locals and heap overlap");
  DevBuf d_trace(seg.n_trace * 8), d_mem((size_t)n_mem * 20 + 4), d_init((size_t)n_init * 16 + 4), d_heap((size_t)n_heap * 16 + 4), d_err(8);
  CM_HIP(hipMemcpyAsync(d_trace.p, seg.trace, seg.n_trace * 8, hipMemcpyHostToDevice, st));
  if (n_mem) CM_HIP(hipMemcpyAsync(d_mem.p, seg.memory_trace, (size_t)n_mem * 20, hipMemcpyHostToDevice, st));
  if (n_init) CM_HIP(hipMemcpyAsync(d_init.p, seg.initial_memory, (size_t)n_init * 16, hipMemcpyHostToDevice, st));
  if (n_heap) CM_HIP(hipMemcpyAsync(d_heap.p, seg.initial_heap, (size_t)n_heap * 16, hipMemcpyHostToDevice, st));
  const InitMem init{d_init.u32(), n_init, d_heap.u32(), n_heap};
  OpTable tab;
  memset(&tab, 0, sizeof(tab));
  for (uint32_t op = 0; op < 64; op++) {
    host::OpInfo oi;
    if (host::op_info(op, oi) && air::component_of_opcode(op) >= 0) {
      tab.valid[op] = 1; tab.size[op] = (uint8_t)oi.size_m31; tab.acc[op] = (uint8_t)oi.accesses;
      tab.comp[op] = (uint8_t)air::component_of_opcode(op);
    }
  }
  // ---- 1. per-step counts and offsets (one 64-bit scan: log entries | operand accesses) ----
  CM_HIP(hipMemsetAsync(d_err.p, 0, 8, st));
  DevBuf d_info((size_t)n_steps * 4 + 4), d_off((size_t)n_steps * 8 + 8);
  hipLaunchKernelGGL(k_step_counts, grid1(n_steps), dim3(256), 0, st, d_trace.u32(), n_steps, d_init.u32(), n_init, tab, d_info.u32(),
                     d_off.as<unsigned long long>(), d_err.u32());
  // (the last step's own counts are needed for the totals: keep them before the in-place scan)
  uint32_t* const pin = pinned_words() + PIN_LAST_LAYER;   // no proof runs on this thread while it adapts a segment
  CM_HIP(hipMemcpyAsync(pin + 0, d_off.as<unsigned long long>() + (n_steps - 1), 8, hipMemcpyDeviceToHost, st));
  with_temp([&](void* t, size_t& b) {
    CM_HIP(hipcub::DeviceScan::ExclusiveSum(t, b, d_off.as<unsigned long long>(), d_off.as<unsigned long long>(), (int)n_steps, st));
  });
  // ---- 2. sort the log by address; entries of one cell keep their log order (stable radix sort, 32-bit key + entry index) ----
  uint32_t n_acc = 0, addr_bits = 1;
  DevBuf d_addr((size_t)n_mem * 4 + 4), d_eid((size_t)n_mem * 4 + 4), d_saddr((size_t)n_mem * 4 + 4), d_se((size_t)n_mem * 4 + 4),
      d_eclk((size_t)n_mem * 4 + 4);
  hipLaunchKernelGGL(k_entry_keys, grid1(n_steps), dim3(256), 0, st, d_trace.u32(), n_steps, d_off.as<unsigned long long>(), d_info.u32(),
                     d_mem.u32(), n_mem, d_addr.u32(), d_eid.u32(), d_eclk.u32(), d_err.u32());
  {
    // totals, error flags and the largest address are needed on the host before the sort sizes are trusted
    CM_HIP(hipMemcpyAsync(pin + 2, d_off.as<unsigned long long>() + (n_steps - 1), 8, hipMemcpyDeviceToHost, st));
    CM_HIP(hipMemcpyAsync(pin + 4, d_err.p, 8, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    const uint32_t err = pin[4], amax = pin[5];
    CM_CHECK(!(err & 1u), "adapter: invalid opcode (or an opcode without a prover component)");
    CM_CHECK(!(err & 2u) && pin[0] + pin[2] == n_mem, "adapter: memory trace length does not match the instructions executed");
    CM_CHECK(!(err & 4u), "adapter: a step's first memory entry is not the instruction fetch at pc");
    n_acc = pin[1] + pin[3];
    while (addr_bits < 32 && (amax >> addr_bits)) addr_bits++;
  }
  with_temp([&](void* t, size_t& b) {
    CM_HIP(hipcub::DeviceRadixSort::SortPairs(t, b, d_addr.u32(), d_saddr.u32(), d_eid.u32(), d_se.u32(), (int)n_mem, 0, (int)addr_bits, st));
  });
  // ---- 3. previous accesses, clock updates ----
  DevBuf d_link((size_t)n_mem * 8 + 8), d_cuc((size_t)n_mem * 4 + 4), d_cupos((size_t)n_mem * 4 + 4), d_head((size_t)n_mem * 4 + 4),
      d_cuoff((size_t)n_mem * 4 + 4), d_hrank((size_t)n_mem * 4 + 4);
  CM_HIP(hipMemsetAsync(d_cuc.p, 0, (size_t)n_mem * 4, st));
  hipLaunchKernelGGL(k_prev_links, grid1(n_mem), dim3(256), 0, st, d_saddr.u32(), d_se.u32(), n_mem, d_mem.u32(), d_eclk.u32(), init,
                     d_link.as<uint2>(), d_cuc.u32(), d_cupos.u32(), d_head.u32());
  with_temp([&](void* t, size_t& b) { CM_HIP(hipcub::DeviceScan::ExclusiveSum(t, b, d_cuc.u32(), d_cuoff.u32(), (int)n_mem, st)); });
  with_temp([&](void* t, size_t& b) { CM_HIP(hipcub::DeviceScan::ExclusiveSum(t, b, d_head.u32(), d_hrank.u32(), (int)n_mem, st)); });
  // ---- 4. steps bucketed per opcode component, opcode variants grouped inside (stable 11-bit sort) ----
  DevBuf d_steps((size_t)n_steps * 4 + 4), d_steps_sorted((size_t)n_steps * 4 + 4), d_info_sorted((size_t)n_steps * 4 + 4);
  hipLaunchKernelGGL(k_iota, grid1(n_steps), dim3(256), 0, st, d_steps.u32(), n_steps);
  with_temp([&](void* t, size_t& b) {
    CM_HIP(hipcub::DeviceRadixSort::SortPairs(t, b, d_info.u32(), d_info_sorted.u32(), d_steps.u32(), d_steps_sorted.u32(), (int)n_steps,
                                              (int)INFO_KEY_LO, (int)INFO_KEY_HI, st));
  });
  // component counts: the sorted component array is non-decreasing -> count = upper bound difference
  DevBuf d_counts(32 * 4);
  CM_HIP(hipMemsetAsync(d_counts.p, 0, 32 * 4, st));
  hipLaunchKernelGGL(k_run_ends, grid1(n_steps), dim3(256), 0, st, d_info_sorted.u32(), n_steps, d_counts.u32());
  // ONE host round trip for everything the allocations below need
  uint32_t n_cu = 0, n_cells = 0;
  uint32_t ends[32];
  {
    CM_HIP(hipMemcpyAsync(pin + 8, d_cuoff.u32() + (n_mem - 1), 4, hipMemcpyDeviceToHost, st));
    CM_HIP(hipMemcpyAsync(pin + 9, d_cuc.u32() + (n_mem - 1), 4, hipMemcpyDeviceToHost, st));
    CM_HIP(hipMemcpyAsync(pin + 10, d_hrank.u32() + (n_mem - 1), 4, hipMemcpyDeviceToHost, st));
    CM_HIP(hipMemcpyAsync(pin + 11, d_head.u32() + (n_mem - 1), 4, hipMemcpyDeviceToHost, st));
    CM_HIP(hipMemcpyAsync(pin + 16, d_counts.p, sizeof(ends), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    n_cu = pin[8] + pin[9];
    n_cells = pin[10] + pin[11];
    memcpy(ends, pin + 16, sizeof(ends));
  }
  DevBuf d_cu((size_t)n_cu * sizeof(cm_clock_update) + 4);
  if (n_cu)
    hipLaunchKernelGGL(k_clock_updates, grid1(n_mem), dim3(256), 0, st, d_mem.u32(), n_mem, d_cuc.u32(), d_cuoff.u32(), d_cupos.u32(),
                       d_link.as<uint2>(), d_saddr.u32(), d_se.u32(), init, d_cu.as<cm_clock_update>());
  uint64_t counts[CM_N_OPCODE_COMPONENTS] = {0};
  {
    uint32_t prev_end = 0;  // ends[c] = one past the last step of component c in sorted order (0 if absent)
    for (int c = 0; c < CM_N_OPCODE_COMPONENTS; c++)
      if (ends[c]) { counts[c] = ends[c] - prev_end; prev_end = ends[c]; }
  }
  DevBuf bundles[CM_N_OPCODE_COMPONENTS];
  BundleDst dst;
  uint32_t run = 0;
  for (int c = 0; c < CM_N_OPCODE_COMPONENTS; c++) {
    bundles[c].alloc(counts[c] * sizeof(cm_bundle) + 4);
    dst.p[c] = bundles[c].as<cm_bundle>();
    dst.start[c] = run;
    run += (uint32_t)counts[c];
  }
  dst.start[CM_N_OPCODE_COMPONENTS] = run;
  DevBuf d_acc((size_t)n_acc * sizeof(cm_data_access) + 4);
  hipLaunchKernelGGL(k_step_pos, grid1(n_steps), dim3(256), 0, st, d_steps_sorted.u32(), n_steps, d_steps.u32());   // (the iota is spent)
  hipLaunchKernelGGL(k_bundles, grid1(n_steps), dim3(256), 0, st, d_steps.u32(), n_steps, d_trace.u32(), d_info.u32(),
                     d_off.as<unsigned long long>(), d_mem.u32(), d_link.as<uint2>(), dst, d_acc.as<cm_data_access>(), tab);
  // ---- 5. touched cells -> host: boundary memory, multiplicities, Merkle trees ----
  DevBuf d_cells((size_t)n_cells * sizeof(CellRec) + 16);
  hipLaunchKernelGGL(k_cells, grid1(n_mem), dim3(256), 0, st, d_saddr.u32(), d_se.u32(), n_mem, d_head.u32(), d_hrank.u32(), d_eclk.u32(),
                     d_cells.as<CellRec>());
  std::vector<CellRec> cells(n_cells);
  CM_HIP(hipMemcpyAsync(cells.data(), d_cells.p, (size_t)n_cells * sizeof(CellRec), hipMemcpyDeviceToHost, st));
  CM_HIP(hipGetLastError());
  CM_HIP(hipStreamSynchronize(st));
  std::map<uint32_t, host::MemState> initial_memory, final_memory;
  for (uint32_t a = 0; a < n_init; a++) {
    host::MemState s{{seg.initial_memory[4 * (size_t)a], seg.initial_memory[4 * (size_t)a + 1], seg.initial_memory[4 * (size_t)a + 2],
                      seg.initial_memory[4 * (size_t)a + 3]}, 0u, 0u};
    initial_memory[a] = s;
    final_memory[a] = s;
  }
  for (uint32_t i = 0; i < n_heap; i++) {
    const uint32_t* w = seg.initial_heap + 4 * (size_t)i;
    host::MemState s{{w[0], w[1], w[2], w[3]}, 0u, 0u};
    initial_memory[host::MAX_ADDRESS - i] = s;
    final_memory[host::MAX_ADDRESS - i] = s;
  }
  auto val = [&](uint32_t e) { const uint32_t* w = seg.memory_trace + 5 * (size_t)e + 1; return host::Cell{w[0], w[1], w[2], w[3]}; };
  for (const CellRec& c : cells) {
    auto it = initial_memory.find(c.addr);
    if (it != initial_memory.end()) it->second.mult = 1;
    else initial_memory[c.addr] = host::MemState{val(c.first_entry), 0u, 1u};
    final_memory[c.addr] = host::MemState{val(c.last_entry), c.last_clock, host::M31_NEG1};
  }
  host::ProverInputOwned tail;
  for (int i = 0; i < 2; i++) { tail.program_range[i] = seg.program_range[i]; tail.input_range[i] = seg.input_range[i]; tail.output_range[i] = seg.output_range[i]; }
  // small boundary memories: trees on the host (a few hundred Poseidon2 hashes); large ones: on the GPU
  size_t tree_min = 2048;
  if (const char* e = getenv("CM_ADAPTER_DEVICE_TREE_MIN")) tree_min = (size_t)strtoull(e, nullptr, 10);
  const bool device_trees = initial_memory.size() >= tree_min;
  host::finish_boundary_memory(initial_memory, final_memory, tail, !device_trees);
  DevBuf init_tree_dev, fin_tree_dev;
  uint64_t n_init_tree = 0, n_fin_tree = 0;
  if (device_trees) {
    for (int half = 0; half < 2; half++) {
      std::vector<uint32_t> li, lv, lm;
      host::partial_tree_leaves(half ? final_memory : initial_memory, half == 0, tail.program_range, tail.input_range, tail.output_range, li, lv, lm);
      uint32_t root = build_partial_merkle_tree_device(li, lv, lm, half ? fin_tree_dev : init_tree_dev, half ? n_fin_tree : n_init_tree, st);
      (half ? tail.final_root : tail.initial_root) = root;
    }
  }
  // ---- assemble the device-resident ProverInput ----
  cm_prover_input meta = tail.view();
  meta.initial_pc = seg.trace[0]; meta.initial_fp = seg.trace[1];
  meta.final_pc = seg.trace[2 * (size_t)n_steps]; meta.final_fp = seg.trace[2 * (size_t)n_steps + 1];
  for (int c = 0; c < CM_N_OPCODE_COMPONENTS; c++) { meta.bundles[c] = nullptr; meta.n_bundles[c] = counts[c]; }
  meta.data_accesses = nullptr; meta.n_data_accesses = n_acc;
  meta.clock_updates = nullptr; meta.n_clock_updates = n_cu;
  if (device_trees) { meta.n_initial_tree = n_init_tree; meta.n_final_tree = n_fin_tree; }
  return make_device_input(meta, bundles, d_acc, d_cu, device_trees ? &init_tree_dev : nullptr, device_trees ? &fin_tree_dev : nullptr);
}

}  // namespace cm
