// Host-only part of the C ABI: synthetic VM + adapter (cm_vm_run, cm_synth_fibonacci, ...).
#include "../../include/cairom_hip.h"
#include "host_adapter.hpp"
#include <string>

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
const cm_prover_input* cm_host_input_view(const cm_host_input* h) { return &h->view; }
uint64_t cm_host_input_steps(const cm_host_input* h) { return h->owned.n_steps; }
int32_t cm_host_input_free(cm_host_input* h) { delete h; return 0; }
int32_t cm_poseidon2_permute(uint32_t state[16]) {
  cm::M31 s[16];
  for (int i = 0; i < 16; i++) s[i] = cm::M31::from_u32(state[i]);
  cm::host::poseidon2_permute(s);
  for (int i = 0; i < 16; i++) state[i] = s[i].v;
  return 0;
}
}
