// Host-side input pipeline of the product: (1) a minimal Cairo-M VM for synthetic traces (the runner is
// out of scope), (2) the adapter `import_from_runner_output`
// (/root/reference/crates/prover/src/adapter/mod.rs:97-193, adapter/memory.rs:271-535,
// adapter/merkle.rs:183-295) restated with deterministic row order (the reference iterates
// HashMaps; this build emits memory rows in ascending address order).
// Everything here is sequential host code — it prepares `cm_prover_input`, the thing the GPU path proves.
#pragma once
#include <stdint.h>
#include <array>
#include <map>
#include <unordered_map>
#include <vector>
#include <string>
#include <stdexcept>
#include <algorithm>
#include "../../include/cairom_hip.h"
#include "field.hpp"
#include "air/air_common.hpp"
#include "air/builtins.hpp"

namespace cm {
namespace host {

struct CmOps {
  using M = cm::M31;
  static CM_HD M mk(uint32_t v) { return cm::M31(v); }
  static CM_HD M inv(M x) { return cm::inv(x); }
};

// ---- Poseidon2-M31 hash (crates/prover/src/poseidon2.rs:25-54) ----------------------------------
CM_HD void poseidon2_permute(cm::M31* s) {
  using namespace air;
  auto mk = [](uint32_t v) { return cm::M31(v); };
  p2_external_matrix(s);
  for (int half = 0; half < 2; half++) {
    if (half == 1)
      for (int r = 0; r < P2_PARTIAL; r++) {
        s[0] = s[0] + cm::M31(p2_int_rc(r));
        cm::M31 x = s[0], x2 = x * x;
        s[0] = x2 * x2 * x;
        p2_internal_matrix(s, mk);
      }
    for (int r = 0; r < P2_FULL / 2; r++) {
      for (int i = 0; i < P2_T; i++) {
        cm::M31 x = s[i] + cm::M31(p2_ext_rc(half * 4 + r, i)), x2 = x * x;
        s[i] = x2 * x2 * x;
      }
      p2_external_matrix(s);
    }
  }
}
CM_HD uint32_t poseidon2_hash(uint32_t l, uint32_t r) {
  cm::M31 s[16];
  s[0] = cm::M31(l);
  s[1] = cm::M31(r);
  poseidon2_permute(s);
  return s[0].v;
}
inline const std::vector<uint32_t>& poseidon2_default_hashes() {
  static std::vector<uint32_t> d = [] {
    std::vector<uint32_t> v(air::TREE_HEIGHT + 1, 0);
    for (int depth = (int)air::TREE_HEIGHT - 1; depth >= 0; depth--) v[depth] = poseidon2_hash(v[depth + 1], v[depth + 1]);
    return v;
  }();
  return d;
}

// ---- instruction table (crates/common/src/instruction.rs:314-577) -------------------------------
struct OpInfo { int size_m31; int accesses; };
inline bool op_info(uint32_t op, OpInfo& o) {
  switch (op) {
    case 0: case 1: case 2: case 3: o = {4, 3}; return true;
    case 4: case 6: case 48: o = {4, 2}; return true;
    case 50: o = {3, 1}; return true;
    case 8: o = {4, 3}; return true;
    case 42: o = {4, 4}; return true;
    case 9: case 43: o = {3, 1}; return true;
    case 10: o = {3, 2}; return true;
    case 11: o = {1, 2}; return true;
    case 12: case 13: o = {2, 0}; return true;
    case 14: o = {3, 1}; return true;
    case 15: case 16: case 17: o = {4, 6}; return true;
    case 18: o = {5, 8}; return true;
    case 19: case 21: o = {5, 4}; return true;
    case 22: o = {6, 6}; return true;
    case 23: o = {4, 2}; return true;
    case 24: case 28: o = {4, 5}; return true;
    case 30: case 34: o = {5, 3}; return true;
    case 36: case 37: case 38: o = {4, 6}; return true;
    case 39: case 40: case 41: o = {5, 4}; return true;
    case 44: o = {4, 3}; return true;
    case 45: o = {4, 4}; return true;
    case 46: o = {2, 1}; return true;
    case 47: o = {2, 2}; return true;
    default: return false;
  }
}

using Cell = std::array<uint32_t, 4>;  // QM31 memory cell
struct MemEntry { uint32_t addr; Cell value; };
struct Segment {
  std::vector<std::array<uint32_t, 2>> trace;  // (pc, fp) per step + the final state
  std::vector<MemEntry> memory_trace;
  std::vector<Cell> initial_memory;            // dense: address i -> cell (the locals: program, frames)
  std::vector<Cell> initial_heap;              // dense from the top: index i -> cell at MAX_ADDRESS - i (runner/src/vm/mod.rs:205-221)
};
constexpr uint32_t MAX_ADDRESS = (1u << 28) - 1;   // crates/common/src/lib.rs MAX_ADDRESS: the heap grows downwards from here

// ---- minimal VM (semantics: crates/runner/src/vm/instructions/{store,jnz,jump,call}.rs; access-log
// order: instruction word(s) first, then operands src0, src1, dst; ret reads fp-1 then fp-2) ----------
struct VM {
  // Memory of crates/runner/src/memory/mod.rs:48-62: `mem` = the locals (address i), `heap` = cells from the top (index i =
  // address MAX_ADDRESS - i).  A read never grows either vector (an untouched cell reads as zero, :186-200); a write beyond both
  // grows the NEARER one (:262-283) — which cells exist matters for the initial memory of a continuation segment.
  std::vector<Cell> mem, heap;
  std::vector<MemEntry> log;
  uint32_t pc = 0, fp = 0, final_pc = 0;
  void ensure(uint32_t a) { if (a >= mem.size()) mem.resize((size_t)a + 1, Cell{0, 0, 0, 0}); }   // (frame set-up only)
  static void check_address(uint32_t a) {
    if (a > MAX_ADDRESS) throw std::runtime_error("vm: address " + std::to_string(a) + " is beyond MAX_ADDRESS");
  }
  Cell get(uint32_t a) const {
    check_address(a);
    if (a < mem.size()) return mem[a];
    const size_t h = (size_t)MAX_ADDRESS - a;
    if (h < heap.size()) return heap[h];
    return Cell{0, 0, 0, 0};
  }
  void put(uint32_t a, const Cell& c) {
    check_address(a);
    const size_t h = (size_t)MAX_ADDRESS - a;
    if (a < mem.size()) { mem[a] = c; return; }
    if (h < heap.size()) { heap[h] = c; return; }
    if ((size_t)a - mem.size() < h - heap.size()) { mem.resize((size_t)a + 1, Cell{0, 0, 0, 0}); mem[a] = c; }
    else { heap.resize(h + 1, Cell{0, 0, 0, 0}); heap[h] = c; }
  }
  uint32_t rd(uint32_t a) {
    const Cell c = get(a);
    if (c[1] | c[2] | c[3]) throw std::runtime_error("vm: the cell at " + std::to_string(a) + " is not a base-field value");
    log.push_back({a, c});
    return c[0];
  }
  void wr(uint32_t a, uint32_t v) { const Cell c{v, 0, 0, 0}; put(a, c); log.push_back({a, c}); }
  static uint32_t add(uint32_t a, uint32_t b) { return (cm::M31(a) + cm::M31(b)).v; }
  static uint32_t sub(uint32_t a, uint32_t b) { return (cm::M31(a) - cm::M31(b)).v; }
  static uint32_t mul(uint32_t a, uint32_t b) { return (cm::M31(a) * cm::M31(b)).v; }
  void step() {
    if (pc >= mem.size()) throw std::runtime_error("vm: instruction fetch from an uninitialised cell at " + std::to_string(pc));
    Cell w0 = mem[pc];
    log.push_back({pc, w0});
    uint32_t op = w0[0];
    OpInfo oi;
    if (!op_info(op, oi)) throw std::runtime_error("vm: invalid opcode " + std::to_string(op));
    uint32_t in[6] = {w0[0], w0[1], w0[2], w0[3], 0, 0};
    uint32_t npc_inc = 1;
    if (oi.size_m31 > 4) {
      if ((size_t)pc + 1 >= mem.size()) throw std::runtime_error("vm: instruction fetch from an uninitialised cell at " + std::to_string(pc + 1));
      Cell w1 = mem[pc + 1];
      log.push_back({pc + 1, w1});
      in[4] = w1[0]; in[5] = w1[1];
      npc_inc = 2;
    }
    uint32_t npc = add(pc, npc_inc), nfp = fp;
    auto lo16 = [](uint32_t v) { return v & 0xffffu; };
    auto rd32 = [&](uint32_t a) { uint32_t l = rd(a), h = rd(add(a, 1)); return (h << 16) | lo16(l); };
    auto wr32 = [&](uint32_t a, uint32_t v) { wr(a, v & 0xffffu); wr(add(a, 1), v >> 16); };
    switch (op) {
      case 0: case 1: case 2: case 3: {
        uint32_t a = rd(add(fp, in[1])), b = rd(add(fp, in[2])), r;
        if (op == 0) r = add(a, b); else if (op == 1) r = sub(a, b); else if (op == 2) r = mul(a, b);
        else r = (cm::M31(a) * cm::inv(cm::M31(b))).v;
        wr(add(fp, in[3]), r);
      } break;
      case 4: wr(add(fp, in[3]), add(rd(add(fp, in[1])), in[2])); break;
      case 6: wr(add(fp, in[3]), mul(rd(add(fp, in[1])), in[2])); break;
      case 48: { uint32_t a = rd(add(fp, in[1])); wr(add(fp, in[3]), a <= in[2] ? 1u : 0u); } break;
      case 50: { uint32_t a = rd(add(fp, in[1])); if (a != in[2]) throw std::runtime_error("vm: assert_eq failed"); } break;
      case 8: { uint32_t base = rd(add(fp, in[1])); uint32_t v = rd(add(base, in[2])); wr(add(fp, in[3]), v); } break;
      case 44: { uint32_t base = rd(add(fp, in[1])); uint32_t v = rd(add(fp, in[3])); wr(add(base, in[2]), v); } break;
      case 42: { uint32_t base = rd(add(fp, in[1])), off = rd(add(fp, in[2])); uint32_t v = rd(add(base, off)); wr(add(fp, in[3]), v); } break;
      case 45: { uint32_t base = rd(add(fp, in[1])), off = rd(add(fp, in[2])); uint32_t v = rd(add(fp, in[3])); wr(add(base, off), v); } break;
      case 9: wr(add(fp, in[2]), in[1]); break;
      case 43: wr(add(fp, in[2]), add(fp, in[1])); break;
      case 10: { wr(add(fp, in[1]), fp); wr(add(add(fp, in[1]), 1), add(pc, 1)); nfp = add(add(fp, in[1]), 2); npc = in[2]; } break;
      case 11: { uint32_t rpc = rd(sub(fp, 1)), rfp = rd(sub(fp, 2)); npc = rpc; nfp = rfp; } break;
      case 12: npc = in[1]; break;
      case 13: npc = add(pc, in[1]); break;
      case 14: { uint32_t c = rd(add(fp, in[1])); if (c != 0) npc = add(pc, in[2]); } break;
      // ---- u32 ops: two 16-bit limbs in consecutive cells (runner/src/memory/mod.rs:22-25)
      case 15: case 16: case 17: case 36: case 37: case 38: {
        uint32_t a = rd32(add(fp, in[1])), b = rd32(add(fp, in[2])), r;
        switch (op) { case 15: r = a + b; break; case 16: r = a - b; break; case 17: r = a * b; break;
                      case 36: r = a & b; break; case 37: r = a | b; break; default: r = a ^ b; }
        wr32(add(fp, in[3]), r);
      } break;
      case 18: {
        uint32_t a = rd32(add(fp, in[1])), b = rd32(add(fp, in[2]));
        uint32_t q = b ? a / b : 0, r = b ? a % b : 0;
        wr32(add(fp, in[3]), q); wr32(add(fp, in[4]), r);
      } break;
      case 19: case 21: case 39: case 40: case 41: {
        uint32_t a = rd32(add(fp, in[1])), b = (in[3] << 16) | lo16(in[2]), r;
        switch (op) { case 19: r = a + b; break; case 21: r = a * b; break; case 39: r = a & b; break;
                      case 40: r = a | b; break; default: r = a ^ b; }
        wr32(add(fp, in[4]), r);
      } break;
      case 22: {
        uint32_t a = rd32(add(fp, in[1])), b = (in[3] << 16) | lo16(in[2]);
        uint32_t q = b ? a / b : 0, r = b ? a % b : 0;
        wr32(add(fp, in[4]), q); wr32(add(fp, in[5]), r);
      } break;
      case 23: wr32(add(fp, in[3]), (in[2] << 16) | lo16(in[1])); break;
      case 24: case 28: {
        uint32_t a = rd32(add(fp, in[1])), b = rd32(add(fp, in[2]));
        wr(add(fp, in[3]), op == 24 ? (a == b) : (a < b));
      } break;
      case 30: case 34: {
        uint32_t a = rd32(add(fp, in[1])), b = (in[3] << 16) | lo16(in[2]);
        wr(add(fp, in[4]), op == 30 ? (a == b) : (a < b));
      } break;
      default: throw std::runtime_error("vm: opcode not supported by the synthetic VM: " + std::to_string(op));
    }
    pc = npc;
    fp = nfp;
  }
};

// program: list of instructions as M31 words (opcode first). Returns the segments (cut at max_steps the
// way crates/runner/src/vm/mod.rs:158-240 does).
inline std::vector<Segment> run_program(const std::vector<std::vector<uint32_t>>& program, uint32_t entry_pc,
                                        const std::vector<uint32_t>& args, uint32_t n_returns, uint64_t max_steps,
                                        uint32_t* program_len_out) {
  VM vm;
  for (auto& ins : program) {
    for (size_t i = 0; i < ins.size(); i += 4) {
      Cell c{0, 0, 0, 0};
      for (size_t k = 0; k < 4 && i + k < ins.size(); k++) c[k] = ins[i + k];
      vm.mem.push_back(c);
    }
  }
  uint32_t plen = (uint32_t)vm.mem.size();
  if (program_len_out) *program_len_out = plen;
  vm.final_pc = plen;
  uint32_t fp_offset = (uint32_t)args.size() + n_returns + 2;
  uint32_t new_fp = plen + fp_offset;
  for (size_t i = 0; i < args.size(); i++) {
    uint32_t a = new_fp - (uint32_t)(args.size() + n_returns + 2 - i);
    vm.ensure(a);
    vm.mem[a] = Cell{args[i], 0, 0, 0};
  }
  vm.pc = entry_pc;
  vm.fp = new_fp;
  vm.ensure(new_fp - 1);
  vm.mem[new_fp - 2] = Cell{new_fp, 0, 0, 0};
  vm.mem[new_fp - 1] = Cell{vm.final_pc, 0, 0, 0};
  std::vector<Segment> segs;
  std::vector<Cell> initial = vm.mem, initial_heap = vm.heap;
  for (;;) {
    Segment s;
    while (vm.pc != vm.final_pc && s.trace.size() < max_steps) {
      s.trace.push_back({vm.pc, vm.fp});
      vm.step();
    }
    s.trace.push_back({vm.pc, vm.fp});
    s.memory_trace.swap(vm.log);
    s.initial_memory = initial;
    s.initial_heap = initial_heap;
    bool done = vm.pc == vm.final_pc;
    segs.push_back(std::move(s));
    if (done) break;
    initial = vm.mem;
    initial_heap = vm.heap;
  }
  return segs;
}

// ---- adapter ------------------------------------------------------------------------------------
struct ProverInputOwned {
  uint32_t initial_pc = 0, initial_fp = 0, final_pc = 0, final_fp = 0;
  std::vector<cm_bundle> bundles[CM_N_OPCODE_COMPONENTS];
  std::vector<cm_data_access> data_accesses;
  std::vector<cm_memory_cell> initial_memory, final_memory;
  std::vector<cm_clock_update> clock_updates;
  std::vector<cm_merkle_node> initial_tree, final_tree;
  uint32_t initial_root = 0, final_root = 0;
  uint32_t program_range[2] = {0, 0}, input_range[2] = {0, 0}, output_range[2] = {0, 0};
  uint64_t n_steps = 0;
  cm_prover_input view() const {
    cm_prover_input v;
    v.initial_pc = initial_pc; v.initial_fp = initial_fp; v.final_pc = final_pc; v.final_fp = final_fp;
    for (int i = 0; i < CM_N_OPCODE_COMPONENTS; i++) { v.bundles[i] = bundles[i].data(); v.n_bundles[i] = bundles[i].size(); }
    v.data_accesses = data_accesses.data(); v.n_data_accesses = data_accesses.size();
    v.initial_memory = initial_memory.data(); v.n_initial_memory = initial_memory.size();
    v.final_memory = final_memory.data(); v.n_final_memory = final_memory.size();
    v.clock_updates = clock_updates.data(); v.n_clock_updates = clock_updates.size();
    v.initial_tree = initial_tree.data(); v.n_initial_tree = initial_tree.size();
    v.final_tree = final_tree.data(); v.n_final_tree = final_tree.size();
    v.initial_root = initial_root; v.final_root = final_root;
    for (int i = 0; i < 2; i++) { v.program_range[i] = program_range[i]; v.input_range[i] = input_range[i]; v.output_range[i] = output_range[i]; }
    return v;
  }
};

struct MemState { Cell value; uint32_t clock; uint32_t mult; };  // (value, clock, multiplicity)
constexpr uint32_t M31_NEG1 = cm::P - 1;

// build_partial_merkle_tree (adapter/merkle.rs:183-295).  memory: address -> state (ordered map gives
// the deterministic leaf order; node order = depth 30..1, ascending index, as in the reference).
inline uint32_t build_partial_merkle_tree(const std::map<uint32_t, MemState>& memory, bool initial,
                                          const uint32_t prog[2], const uint32_t inp[2], const uint32_t outp[2],
                                          std::vector<cm_merkle_node>& nodes) {
  struct MV { uint32_t value, mult; };
  std::map<uint32_t, MV> cur;
  if (memory.empty()) return 0;   // the reference returns (empty tree, None) (adapter/merkle.rs:190-193); callers check `nodes`
  for (auto& kv : memory) {
    uint32_t addr = kv.first;
    bool pub = initial ? ((addr >= prog[0] && addr < prog[1]) || (addr >= inp[0] && addr < inp[1]))
                       : (addr >= outp[0] && addr < outp[1]);
    for (uint32_t i = 0; i < 4; i++) cur[(addr << 2) + i] = MV{kv.second.value[i], pub ? 2u : 1u};
  }
  const std::vector<uint32_t>& dflt = poseidon2_default_hashes();
  for (uint32_t depth = air::TREE_HEIGHT; depth >= 1; depth--) {
    std::map<uint32_t, MV> parent;
    for (auto it = cur.begin(); it != cur.end();) {
      uint32_t index = it->first;
      uint32_t left_index = index & ~1u, right_index = left_index | 1u;
      MV l{dflt[depth], 0}, r{dflt[depth], 0};
      auto li = cur.find(left_index), ri = cur.find(right_index);
      if (li != cur.end()) l = li->second;
      if (ri != cur.end()) r = ri->second;
      uint32_t ph = poseidon2_hash(l.value, r.value);
      nodes.push_back(cm_merkle_node{left_index, depth, l.value, r.value, ph, l.mult, r.mult, 1u});
      parent[index >> 1] = MV{ph, 1u};
      // skip both siblings
      ++it;
      if (it != cur.end() && it->first == right_index && index == left_index) ++it;
    }
    cur.swap(parent);
  }
  return cur.begin()->second.value;
}

// Memory (adapter/memory.rs:186-193) with Memory::push (adapter/memory.rs:470-535)
struct MemArg { uint32_t address; Cell prev_val, value; uint32_t prev_clock, clock; };
struct MemoryTracker {
  std::map<uint32_t, MemState> initial_memory, final_memory;
  std::vector<cm_clock_update> clock_updates;
  MemArg push(uint32_t address, const Cell& value, uint32_t clock) {
    MemState prev;
    auto it = final_memory.find(address);
    if (it == final_memory.end()) {
      prev = MemState{value, 0u, M31_NEG1};
      final_memory[address] = MemState{value, clock, M31_NEG1};
    } else {
      prev = it->second;
      it->second = MemState{value, clock, M31_NEG1};
    }
    uint32_t prev_clk = prev.clock;
    if (prev_clk == 0) {
      auto ii = initial_memory.find(address);
      if (ii != initial_memory.end()) ii->second.mult = 1;
      else initial_memory[address] = MemState{value, 0u, 1u};
    }
    auto init_it = initial_memory.find(address);
    if (clock > prev_clk) {
      uint32_t delta = clock - prev_clk;
      if (delta > air::RC20_LIMIT) {
        if (init_it == initial_memory.end()) throw std::runtime_error("adapter: clock update on a cell without initial entry");
        const MemState& init = init_it->second;
        uint32_t num_steps = delta / air::RC20_LIMIT;
        for (uint32_t k = 0; k < num_steps; k++) {
          clock_updates.push_back(cm_clock_update{address, prev_clk, {init.value[0], init.value[1], init.value[2], init.value[3]}});
          prev_clk += air::RC20_LIMIT;
        }
      }
    }
    return MemArg{address, prev.value, value, prev_clk, clock};
  }
};

// Tail of import_internal shared by the host and the device adapter: public-range multiplicities
// (adapter/memory.rs:427-461), boundary memory rows in ascending address order, the two partial Merkle trees.
// out.program_range / input_range / output_range must be set.
inline void finish_boundary_memory(std::map<uint32_t, MemState>& initial_memory, std::map<uint32_t, MemState>& final_memory,
                                   ProverInputOwned& out, bool build_trees = true) {
  const uint32_t* prog = out.program_range;
  const uint32_t* inp = out.input_range;
  const uint32_t* outp = out.output_range;
  // update_multiplicities (adapter/memory.rs:427-461)
  for (uint32_t addr = prog[0]; addr < prog[1]; addr++) {
    auto i = initial_memory.find(addr); if (i != initial_memory.end()) i->second.mult = 0;
    auto f = final_memory.find(addr); if (f != final_memory.end() && f->second.mult == 0) f->second.mult = M31_NEG1;
  }
  for (uint32_t addr = inp[0]; addr < inp[1]; addr++) {
    auto i = initial_memory.find(addr); if (i != initial_memory.end()) i->second.mult = 0;
    auto f = final_memory.find(addr); if (f != final_memory.end() && f->second.mult == 0) f->second.mult = M31_NEG1;
  }
  for (uint32_t addr = outp[0]; addr < outp[1]; addr++) {
    auto f = final_memory.find(addr); if (f != final_memory.end()) f->second.mult = 0;
    auto i = initial_memory.find(addr); if (i != initial_memory.end()) i->second.mult = 1;
  }
  for (auto& kv : initial_memory)
    out.initial_memory.push_back(cm_memory_cell{kv.first, {kv.second.value[0], kv.second.value[1], kv.second.value[2], kv.second.value[3]}, kv.second.clock, kv.second.mult});
  for (auto& kv : final_memory)
    out.final_memory.push_back(cm_memory_cell{kv.first, {kv.second.value[0], kv.second.value[1], kv.second.value[2], kv.second.value[3]}, kv.second.clock, kv.second.mult});
  if (!build_trees) return;  // the device adapter hashes large trees on the GPU
  out.initial_root = build_partial_merkle_tree(initial_memory, true, prog, inp, outp, out.initial_tree);
  out.final_root = build_partial_merkle_tree(final_memory, false, prog, inp, outp, out.final_tree);
}
// leaves of a partial tree: (address << 2 | i, value_i, multiplicity 2 for public cells else 1), ascending
inline void partial_tree_leaves(const std::map<uint32_t, MemState>& memory, bool initial, const uint32_t prog[2], const uint32_t inp[2],
                                const uint32_t outp[2], std::vector<uint32_t>& idx, std::vector<uint32_t>& val, std::vector<uint32_t>& mult) {
  for (auto& kv : memory) {
    uint32_t addr = kv.first;
    bool pub = initial ? ((addr >= prog[0] && addr < prog[1]) || (addr >= inp[0] && addr < inp[1])) : (addr >= outp[0] && addr < outp[1]);
    for (uint32_t i = 0; i < 4; i++) { idx.push_back((addr << 2) + i); val.push_back(kv.second.value[i]); mult.push_back(pub ? 2u : 1u); }
  }
}

// import_internal (adapter/mod.rs:97-193)
inline ProverInputOwned import_segment(const Segment& seg, const uint32_t prog[2], const uint32_t inp[2],
                                       const uint32_t outp[2]) {
  ProverInputOwned out;
  if (seg.trace.empty()) throw std::runtime_error("adapter: empty trace");
  for (int i = 0; i < 2; i++) { out.program_range[i] = prog[i]; out.input_range[i] = inp[i]; out.output_range[i] = outp[i]; }
  MemoryTracker mt;
  std::map<uint32_t, MemState>& initial_memory = mt.initial_memory;
  std::map<uint32_t, MemState>& final_memory = mt.final_memory;
  for (size_t a = 0; a < seg.initial_memory.size(); a++) {
    MemState s{seg.initial_memory[a], 0u, 0u};
    initial_memory[(uint32_t)a] = s;
    final_memory[(uint32_t)a] = s;
  }
  if (seg.initial_heap.size() > (size_t)MAX_ADDRESS + 1 - seg.initial_memory.size()) throw std::runtime_error("adapter: locals and heap overlap");
  for (size_t i = 0; i < seg.initial_heap.size(); i++) {
    MemState s{seg.initial_heap[i], 0u, 0u};
    initial_memory[MAX_ADDRESS - (uint32_t)i] = s;
    final_memory[MAX_ADDRESS - (uint32_t)i] = s;
  }
  using Arg = MemArg;
  auto push = [&](uint32_t address, const Cell& value, uint32_t clock) -> Arg { return mt.push(address, value, clock); };
  out.initial_pc = seg.trace[0][0];
  out.initial_fp = seg.trace[0][1];
  size_t mi = 0;
  uint32_t clock = 1;
  auto next_mem = [&]() -> const MemEntry& {
    if (mi >= seg.memory_trace.size()) throw std::runtime_error("adapter: unexpected end of memory trace");
    return seg.memory_trace[mi++];
  };
  for (size_t t = 0; t + 1 < seg.trace.size(); t++) {
    const MemEntry& ie = next_mem();
    Arg iarg = push(ie.addr, ie.value, clock);
    uint32_t op = ie.value[0];
    OpInfo oi;
    if (!op_info(op, oi)) throw std::runtime_error("adapter: invalid opcode");
    cm_bundle b;
    b.pc = seg.trace[t][0]; b.fp = seg.trace[t][1]; b.clock = clock; b.inst_prev_clock = iarg.prev_clock;
    for (int k = 0; k < 6; k++) b.inst[k] = 0;
    for (int k = 0; k < 4 && k < oi.size_m31; k++) b.inst[k] = ie.value[k];
    if (oi.size_m31 > 4) {
      const MemEntry& e2 = next_mem();
      push(e2.addr, e2.value, clock);
      b.inst[4] = e2.value[0];
      if (oi.size_m31 > 5) b.inst[5] = e2.value[1];
    }
    b.span_start = (uint32_t)out.data_accesses.size();
    for (int k = 0; k < oi.accesses; k++) {
      const MemEntry& oe = next_mem();
      Arg a = push(oe.addr, oe.value, clock);
      out.data_accesses.push_back(cm_data_access{a.address, a.prev_clock, a.prev_val[0], a.value[0]});
    }
    b.span_len = (uint32_t)out.data_accesses.size() - b.span_start;
    int comp = air::component_of_opcode(op);
    if (comp < 0) throw std::runtime_error("adapter: opcode has no prover component");
    out.bundles[comp].push_back(b);
    clock++;
  }
  // Row order inside a component = the reference's: `states_by_opcodes` keeps one Vec per OPCODE in step order
  // (adapter/mod.rs:118-130) and Claim::write_trace concatenates the variants of a component in `define_opcodes!` order
  // (components/opcodes/mod.rs:51-58, 223-268: ascending opcode id in every group) — e.g. store_fp_fp = all adds, then all
  // subs, muls, divs.  A stable sort by opcode keeps the step order inside each variant.
  for (int c = 0; c < CM_N_OPCODE_COMPONENTS; c++)
    std::stable_sort(out.bundles[c].begin(), out.bundles[c].end(), [](const cm_bundle& x, const cm_bundle& y) { return x.inst[0] < y.inst[0]; });
  out.clock_updates = mt.clock_updates;
  out.n_steps = seg.trace.size() - 1;
  out.final_pc = seg.trace.back()[0];
  out.final_fp = seg.trace.back()[1];
  finish_boundary_memory(initial_memory, final_memory, out);
  return out;
}

// The hand-assembled fibonacci_loop of SURVEY §8d (10 steps/iteration; 10*n + 12 steps total).
inline std::vector<std::vector<uint32_t>> fibonacci_loop_program() {
  const uint32_t P = cm::P;
  auto neg = [&](uint32_t k) { return P - k; };
  return {
      {9, 0, 0},           //  0: a = 0
      {9, 1, 1},           //  1: b = 1
      {9, 0, 2},           //  2: i = 0
      {4, neg(4), 0, 3},   //  3: n' = [fp-4] + 0
      {1, 2, 3, 5},        //  4: t = i - n'
      {14, 5, 3},          //  5: jnz t -> 8
      {9, 0, 5},           //  6: t = 0
      {13, 2},             //  7: jmp rel -> 9
      {9, 1, 5},           //  8: t = 1
      {14, 5, 2},          //  9: jnz t -> 11
      {13, 7},             // 10: jmp rel -> 17
      {0, 0, 1, 6},        // 11: tmp = a + b
      {4, 1, 0, 0},        // 12: a = b + 0
      {4, 6, 0, 1},        // 13: b = tmp + 0
      {4, 2, 1, 7},        // 14: i' = i + 1
      {4, 7, 0, 2},        // 15: i = i' + 0
      {13, neg(12)},       // 16: jmp rel -> 4
      {4, 0, 0, neg(3)},   // 17: ret slot = a + 0
      {11},                // 18: ret
  };
}

}  // namespace host
}  // namespace cm
