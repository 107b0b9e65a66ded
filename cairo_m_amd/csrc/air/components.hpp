// Component registry: ComponentId -> AIR description type, plus static metadata derived by running the
// description through a counting evaluator (the analogue of Stwo's InfoEvaluator).
#pragma once
#include "air_common.hpp"
#include "logup_stream.hpp"
#include "opcodes_felt.hpp"
#include "opcodes_u32.hpp"
#include "builtins.hpp"

namespace air {

// X(component id, description type)
#define AIR_OPCODE_COMPONENTS(X)                      \
  X(C_ASSERT_EQ_FP_IMM, AssertEqFpImm)                \
  X(C_CALL_ABS_IMM, CallAbsImm)                       \
  X(C_JMP_IMM, JmpImm)                                \
  X(C_JNZ_FP_IMM, JnzFpImm)                           \
  X(C_RET, Ret)                                       \
  X(C_STORE_IMM, StoreImm)                            \
  X(C_STORE_FP_FP, StoreFpFp)                         \
  X(C_STORE_FP_IMM, StoreFpImm)                       \
  X(C_DOUBLE_DEREF_FP_IMM, DoubleDerefFpImm)          \
  X(C_DOUBLE_DEREF_FP_FP, DoubleDerefFpFp)            \
  X(C_STORE_FRAME_POINTER, StoreFramePointer)         \
  X(C_U32_STORE_IMM, U32StoreImm)                     \
  X(C_U32_STORE_ADD_FP_IMM, U32StoreAddFpImm)         \
  X(C_U32_STORE_MUL_FP_IMM, U32StoreMulFpImm)         \
  X(C_U32_STORE_DIV_FP_IMM, U32StoreDivFpImm)         \
  X(C_U32_STORE_EQ_FP_FP, U32StoreEqFpFp)             \
  X(C_U32_STORE_EQ_FP_IMM, U32StoreEqFpImm)           \
  X(C_U32_STORE_LT_FP_IMM, U32StoreLtFpImm)           \
  X(C_U32_STORE_LT_FP_FP, U32StoreLtFpFp)             \
  X(C_U32_STORE_ADD_FP_FP, U32StoreAddFpFp)           \
  X(C_U32_STORE_SUB_FP_FP, U32StoreSubFpFp)           \
  X(C_U32_STORE_MUL_FP_FP, U32StoreMulFpFp)           \
  X(C_U32_STORE_DIV_FP_FP, U32StoreDivFpFp)           \
  X(C_U32_STORE_BITWISE_FP_FP, U32StoreBitwiseFpFp)   \
  X(C_U32_STORE_BITWISE_FP_IMM, U32StoreBitwiseFpImm) \
  X(C_STORE_LE_FP_IMM, StoreLeFpImm)

using RangeCheck8C = RangeCheckC<PP_RC8, REL_RC8>;
using RangeCheck16C = RangeCheckC<PP_RC16, REL_RC16>;
using RangeCheck20C = RangeCheckC<PP_RC20, REL_RC20>;

#define AIR_BUILTIN_COMPONENTS(X) \
  X(C_MEMORY, MemoryC)            \
  X(C_MERKLE, MerkleC)            \
  X(C_CLOCK_UPDATE, ClockUpdateC) \
  X(C_POSEIDON2, Poseidon2C)      \
  X(C_RC8, RangeCheck8C)          \
  X(C_RC16, RangeCheck16C)        \
  X(C_RC20, RangeCheck20C)        \
  X(C_BITWISE, BitwiseC)

#define AIR_ALL_COMPONENTS(X) AIR_OPCODE_COMPONENTS(X) AIR_BUILTIN_COMPONENTS(X)

template <int CID> struct ComponentOf;
#define AIR_X(id, T) template <> struct ComponentOf<id> { using type = T; };
AIR_ALL_COMPONENTS(AIR_X)
#undef AIR_X

inline const char* component_name(int cid) {
  switch (cid) {
#define AIR_X(id, T) case id: return #T;
    AIR_ALL_COMPONENTS(AIR_X)
#undef AIR_X
    default: return "?";
  }
}

// ---- counting evaluator --------------------------------------------------------------------------
struct InfoF {};
inline InfoF operator+(InfoF, InfoF) { return {}; }
inline InfoF operator-(InfoF, InfoF) { return {}; }
inline InfoF operator*(InfoF, InfoF) { return {}; }
inline InfoF operator-(InfoF) { return {}; }
struct InfoEval : LogupStream<InfoEval, InfoF, InfoF> {
  int n_trace = 0, n_base_constraints = 0, n_entries = 0, n_batches = 0;
  int n_preproc = 0, preproc_ids[8] = {0};
  int rel_count[N_RELATIONS] = {0};
  InfoF next() { n_trace++; return {}; }
  InfoF preproc(int id) { preproc_ids[n_preproc++] = id; return {}; }
  InfoF c(uint32_t) { return {}; }
  void constraint(InfoF) { n_base_constraints++; }
  InfoF combine(int, const InfoF*, int) { return {}; }
  InfoF ef_from(InfoF) { return {}; }
  void on_entry(int r, InfoF, const InfoF*, int) { n_entries++; rel_count[r]++; }
  void emit_batch(bool, InfoF, InfoF) { n_batches++; }
};

struct ComponentInfo {
  int n_trace;             // tree-1 columns
  int n_interaction;       // tree-2 columns (4 per logup batch)
  int n_base_constraints;  // add_constraint calls
  int n_constraints;       // base + one per logup batch
  int n_preproc;
  int preproc_ids[8];
  int rel_count[N_RELATIONS];
};
template <class C>
inline ComponentInfo make_info() {
  InfoEval e;
  C::eval(e);
  ComponentInfo i;
  i.n_trace = e.n_trace;
  i.n_interaction = 4 * e.n_batches;
  i.n_base_constraints = e.n_base_constraints;
  i.n_constraints = e.n_base_constraints + e.n_batches;
  i.n_preproc = e.n_preproc;
  for (int k = 0; k < 8; k++) i.preproc_ids[k] = e.preproc_ids[k];
  for (int k = 0; k < N_RELATIONS; k++) i.rel_count[k] = e.rel_count[k];
  return i;
}
inline const ComponentInfo& component_info(int cid) {
  static const ComponentInfo infos[N_COMPONENTS] = {
#define AIR_X(id, T) make_info<T>(),
      AIR_ALL_COMPONENTS(AIR_X)
#undef AIR_X
  };
  return infos[cid];
}

}  // namespace air
