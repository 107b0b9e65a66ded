// Host-only part of the C ABI: synthetic VM + adapter (cm_vm_run, cm_synth_fibonacci, ...).
#include "../../include/cairom_hip.h"
#include "host_adapter.hpp"
#include <string>
#include <string.h>

struct cm_host_input {
  cm::host::ProverInputOwned owned;
  cm_prover_input view;
};

extern "C" {
extern int32_t cm_set_last_error(const char* msg);

static int32_t build(const std::vector<std::vector<uint32_t>>& program, uint32_t entry_pc, const std::vector<uint32_t>& args,
                     uint32_t n_returns, uint64_t max_steps, uint32_t segment_index, cm_host_input** out, uint32_t* n_segments_out) {
  try {
    uint32_t plen = 0;
    std::vector<cm::host::Segment> segs = cm::host::run_program(program, entry_pc, args, n_returns, max_steps, &plen);
    if (n_segments_out) *n_segments_out = (uint32_t)segs.size();
    if (segment_index >= segs.size()) return cm_set_last_error("segment index out of range");
    uint32_t prog[2] = {0, plen}, inp[2] = {plen, plen + (uint32_t)args.size()},
             outp[2] = {plen + (uint32_t)args.size(), plen + (uint32_t)args.size() + n_returns};
    cm_host_input* h = new cm_host_input();
    h->owned = cm::host::import_segment(segs[segment_index], prog, inp, outp);
    h->view = h->owned.view();
    *out = h;
    return 0;
  } catch (const std::exception& e) {
    return cm_set_last_error(e.what());
  }
}

int32_t cm_vm_run(const uint32_t* instr_words, const uint32_t* instr_lens, uint32_t n_instr, uint32_t entry_pc,
                  const uint32_t* args, uint32_t n_args, uint32_t n_returns, uint64_t max_steps, uint32_t segment_index,
                  cm_host_input** out, uint32_t* n_segments_out) {
  std::vector<std::vector<uint32_t>> program;
  size_t off = 0;
  for (uint32_t i = 0; i < n_instr; i++) {
    program.emplace_back(instr_words + off, instr_words + off + instr_lens[i]);
    off += instr_lens[i];
  }
  return build(program, entry_pc, std::vector<uint32_t>(args, args + n_args), n_returns, max_steps, segment_index, out, n_segments_out);
}
int32_t cm_synth_fibonacci(uint32_t n, uint64_t max_steps, uint32_t segment_index, cm_host_input** out) {
  return build(cm::host::fibonacci_loop_program(), 0, {n}, 1, max_steps, segment_index, out, nullptr);
}
struct cm_host_segment {
  cm::host::Segment seg;
  std::vector<uint32_t> trace, mem, init, heap;
  cm_runner_segment view;
};
static int32_t build_segment(const std::vector<std::vector<uint32_t>>& program, uint32_t entry_pc, const std::vector<uint32_t>& args,
                             uint32_t n_returns, uint64_t max_steps, uint32_t segment_index, cm_host_segment** out,
                             uint32_t* n_segments_out) {
  try {
    uint32_t plen = 0;
    std::vector<cm::host::Segment> segs = cm::host::run_program(program, entry_pc, args, n_returns, max_steps, &plen);
    if (n_segments_out) *n_segments_out = (uint32_t)segs.size();
    if (segment_index >= segs.size()) return cm_set_last_error("segment index out of range");
    cm_host_segment* h = new cm_host_segment();
    h->seg = std::move(segs[segment_index]);
    for (auto& t : h->seg.trace) { h->trace.push_back(t[0]); h->trace.push_back(t[1]); }
    for (auto& e : h->seg.memory_trace) { h->mem.push_back(e.addr); for (int k = 0; k < 4; k++) h->mem.push_back(e.value[k]); }
    for (auto& c : h->seg.initial_memory) for (int k = 0; k < 4; k++) h->init.push_back(c[k]);
    for (auto& c : h->seg.initial_heap) for (int k = 0; k < 4; k++) h->heap.push_back(c[k]);
    cm_runner_segment& v = h->view;
    v.initial_heap = h->heap.data(); v.n_initial_heap = h->seg.initial_heap.size();
    v.trace = h->trace.data(); v.n_trace = h->seg.trace.size();
    v.memory_trace = h->mem.data(); v.n_memory_trace = h->seg.memory_trace.size();
    v.initial_memory = h->init.data(); v.n_initial_memory = h->seg.initial_memory.size();
    v.program_range[0] = 0; v.program_range[1] = plen;
    v.input_range[0] = plen; v.input_range[1] = plen + (uint32_t)args.size();
    v.output_range[0] = plen + (uint32_t)args.size(); v.output_range[1] = plen + (uint32_t)args.size() + n_returns;
    *out = h;
    return 0;
  } catch (const std::exception& e) {
    return cm_set_last_error(e.what());
  }
}
int32_t cm_vm_segment(const uint32_t* instr_words, const uint32_t* instr_lens, uint32_t n_instr, uint32_t entry_pc,
                      const uint32_t* args, uint32_t n_args, uint32_t n_returns, uint64_t max_steps, uint32_t segment_index,
                      cm_host_segment** out, uint32_t* n_segments_out) {
  std::vector<std::vector<uint32_t>> program;
  size_t off = 0;
  for (uint32_t i = 0; i < n_instr; i++) {
    program.emplace_back(instr_words + off, instr_words + off + instr_lens[i]);
    off += instr_lens[i];
  }
  return build_segment(program, entry_pc, std::vector<uint32_t>(args, args + n_args), n_returns, max_steps, segment_index, out, n_segments_out);
}
int32_t cm_synth_fibonacci_segment(uint32_t n, uint64_t max_steps, uint32_t segment_index, cm_host_segment** out) {
  return build_segment(cm::host::fibonacci_loop_program(), 0, {n}, 1, max_steps, segment_index, out, nullptr);
}
const cm_runner_segment* cm_host_segment_view(const cm_host_segment* h) { return &h->view; }
int32_t cm_host_segment_free(cm_host_segment* h) { delete h; return 0; }
// ---- runner artifacts (wire formats of the reference) ---------------------------------------
// trace file: per state `fp` then `pc`, little-endian u32 (crates/common/src/execution.rs:28-40; reader
// IoTraceEntry {fp, pc}, crates/prover/src/adapter/io.rs:38-43).  memory trace file: optional
// MemoryTraceMetadata {program_length: u32} header (io.rs:76-80) then per access `address, v0, v1, v2, v3`
// little-endian u32 (execution.rs:52-66; IoMemoryEntry, io.rs:56-60).  The reference does not serialise the
// initial memory or the public ranges yet (adapter/mod.rs:215-237 `unimplemented!`): the caller supplies them.
static void put_le32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get_le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
int32_t cm_segment_serialize_trace(const cm_runner_segment* s, uint8_t* out, uint64_t cap, uint64_t* len) {
  *len = s->n_trace * 8;
  if (!out) return 0;
  if (cap < *len) return cm_set_last_error("cm_segment_serialize_trace: buffer too small");
  for (uint64_t i = 0; i < s->n_trace; i++) {
    put_le32(out + 8 * i, s->trace[2 * i + 1]);      // fp
    put_le32(out + 8 * i + 4, s->trace[2 * i]);      // pc
  }
  return 0;
}
int32_t cm_segment_serialize_memory_trace(const cm_runner_segment* s, int32_t with_header, uint8_t* out, uint64_t cap, uint64_t* len) {
  const uint64_t hdr = with_header ? 4 : 0;
  *len = hdr + s->n_memory_trace * 20;
  if (!out) return 0;
  if (cap < *len) return cm_set_last_error("cm_segment_serialize_memory_trace: buffer too small");
  if (with_header) put_le32(out, s->program_range[1] - s->program_range[0]);
  for (uint64_t i = 0; i < s->n_memory_trace; i++)
    for (int k = 0; k < 5; k++) put_le32(out + hdr + 20 * i + 4 * k, s->memory_trace[5 * i + k]);
  return 0;
}
int32_t cm_segment_from_artifacts(const uint8_t* trace, uint64_t trace_len, const uint8_t* mem, uint64_t mem_len, int32_t mem_has_header,
                                  const uint32_t* initial_memory, uint64_t n_initial_memory, const uint32_t ranges[6],
                                  cm_host_segment** out) {
  if (trace_len % 8) return cm_set_last_error("trace file: length is not a multiple of 8 bytes (fp, pc)");
  const uint64_t hdr = mem_has_header ? 4 : 0;
  if (mem_len < hdr || (mem_len - hdr) % 20) return cm_set_last_error("memory trace file: bad length (header + 20-byte records)");
  if (mem_has_header && get_le32(mem) != ranges[1] - ranges[0]) return cm_set_last_error("memory trace header: program_length does not match the program range");
  cm_host_segment* h = new cm_host_segment();
  for (uint64_t i = 0; i < trace_len / 8; i++) { h->trace.push_back(get_le32(trace + 8 * i + 4)); h->trace.push_back(get_le32(trace + 8 * i)); }
  for (uint64_t i = 0; i < (mem_len - hdr) / 4; i++) h->mem.push_back(get_le32(mem + hdr + 4 * i));
  h->init.assign(initial_memory, initial_memory + 4 * n_initial_memory);
  cm_runner_segment& v = h->view;
  v.trace = h->trace.data(); v.n_trace = trace_len / 8;
  v.memory_trace = h->mem.data(); v.n_memory_trace = (mem_len - hdr) / 20;
  v.initial_memory = h->init.data(); v.n_initial_memory = n_initial_memory;
  v.initial_heap = nullptr; v.n_initial_heap = 0;   // (cm_host_segment_set_initial_heap)
  for (int i = 0; i < 2; i++) { v.program_range[i] = ranges[i]; v.input_range[i] = ranges[2 + i]; v.output_range[i] = ranges[4 + i]; }
  *out = h;
  return 0;
}
int32_t cm_host_segment_set_initial_heap(cm_host_segment* h, const uint32_t* initial_heap, uint64_t n_initial_heap) {
  if (!h) return cm_set_last_error("cm_host_segment_set_initial_heap: null segment");
  if (n_initial_heap && !initial_heap) return cm_set_last_error("cm_host_segment_set_initial_heap: null heap");
  // (bounded before any arithmetic on it: 4 * n and n + n_initial_memory must not wrap)
  if (n_initial_heap > (uint64_t)cm::host::MAX_ADDRESS + 1) return cm_set_last_error("cm_host_segment_set_initial_heap: heap larger than the address space");
  if (n_initial_heap + h->view.n_initial_memory > (uint64_t)cm::host::MAX_ADDRESS + 1) return cm_set_last_error("cm_host_segment_set_initial_heap: locals and heap overlap");
  h->heap.assign(initial_heap, initial_heap + 4 * n_initial_heap);
  h->view.initial_heap = h->heap.data(); h->view.n_initial_heap = n_initial_heap;
  return 0;
}
const cm_prover_input* cm_host_input_view(const cm_host_input* h) { return &h->view; }
uint64_t cm_host_input_steps(const cm_host_input* h) { return h->owned.n_steps; }
int32_t cm_host_input_free(cm_host_input* h) { delete h; return 0; }
// Test hook for the reference's Memory::push known-answer tests (adapter/memory.rs:545-858).
// preload: n_preload cells (addr, v0..v3); script: n entries (addr, v0..v3, clock).
// results: per entry (prev_clock, prev_v0..3); state_out per address asked: see tests/test_adapter.py.
int32_t cm_adapter_memory_script(const uint32_t* preload, uint32_t n_preload, const uint32_t* script, uint32_t n,
                                 uint32_t* results, uint32_t* n_clock_updates, uint32_t* clock_updates_out, uint32_t cu_cap,
                                 const uint32_t* query_addrs, uint32_t n_query, uint32_t* state_out) {
  try {
    cm::host::MemoryTracker mt;
    for (uint32_t i = 0; i < n_preload; i++) {
      cm::host::MemState s{{preload[5 * i + 1], preload[5 * i + 2], preload[5 * i + 3], preload[5 * i + 4]}, 0u, 0u};
      mt.initial_memory[preload[5 * i]] = s;
      mt.final_memory[preload[5 * i]] = s;
    }
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t* e = script + 6 * i;
      cm::host::MemArg a = mt.push(e[0], cm::host::Cell{e[1], e[2], e[3], e[4]}, e[5]);
      results[5 * i] = a.prev_clock;
      for (int k = 0; k < 4; k++) results[5 * i + 1 + k] = a.prev_val[k];
    }
    *n_clock_updates = (uint32_t)mt.clock_updates.size();
    for (uint32_t i = 0; i < mt.clock_updates.size() && i < cu_cap; i++) {
      clock_updates_out[6 * i] = mt.clock_updates[i].address;
      clock_updates_out[6 * i + 1] = mt.clock_updates[i].prev_clock;
      for (int k = 0; k < 4; k++) clock_updates_out[6 * i + 2 + k] = mt.clock_updates[i].value[k];
    }
    // per queried address: present_i, v0..3, clock, mult (initial) then the same for final  (14 words)
    for (uint32_t i = 0; i < n_query; i++) {
      uint32_t* o = state_out + 14 * i;
      for (int half = 0; half < 2; half++) {
        auto& m = half ? mt.final_memory : mt.initial_memory;
        auto it = m.find(query_addrs[i]);
        uint32_t* q = o + 7 * half;
        q[0] = it != m.end();
        if (it != m.end()) { for (int k = 0; k < 4; k++) q[1 + k] = it->second.value[k]; q[5] = it->second.clock; q[6] = it->second.mult; }
        else for (int k = 1; k < 7; k++) q[k] = 0;
      }
    }
    return 0;
  } catch (const std::exception& e) { return cm_set_last_error(e.what()); }
}
// Test hook for the reference's partial-Merkle-tree shape tests (adapter/merkle.rs:302-423): cells = n x (addr, v0..v3);
// nodes_out receives up to cap cm_merkle_node records (8 words each), *n_nodes the total, *root the root (0 when empty).
int32_t cm_adapter_partial_tree(const uint32_t* cells, uint32_t n, int32_t initial, const uint32_t ranges[6], uint32_t* nodes_out,
                                uint64_t cap, uint64_t* n_nodes, uint32_t* root) {
  try {
    std::map<uint32_t, cm::host::MemState> mem;
    for (uint32_t i = 0; i < n; i++)
      mem[cells[5 * i]] = cm::host::MemState{{cells[5 * i + 1], cells[5 * i + 2], cells[5 * i + 3], cells[5 * i + 4]}, 0u, 0u};
    std::vector<cm_merkle_node> nodes;
    *root = cm::host::build_partial_merkle_tree(mem, initial != 0, ranges, ranges + 2, ranges + 4, nodes);
    *n_nodes = nodes.size();
    for (size_t i = 0; i < nodes.size() && i < cap; i++) memcpy((void*)(nodes_out + 8 * i), (const void*)&nodes[i], 32);
    return 0;
  } catch (const std::exception& e) { return cm_set_last_error(e.what()); }
}
int32_t cm_poseidon2_permute(uint32_t state[16]) {
  cm::M31 s[16];
  for (int i = 0; i < 16; i++) s[i] = cm::M31::from_u32(state[i]);
  cm::host::poseidon2_permute(s);
  for (int i = 0; i < 16; i++) state[i] = s[i].v;
  return 0;
}
}
