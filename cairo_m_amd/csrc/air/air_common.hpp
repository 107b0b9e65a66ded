// AIR description layer shared by every evaluator (HIP domain/logup/histogram kernels in the product,
// CPU assert/quotient/point evaluators in the oracle).  A component is written ONCE as
//   template <class E> void eval(E& e)           — restates the reference's FrameworkEval::evaluate
//   template <class O> void witness(...)         — restates the row closure of Claim::write_trace
// and instantiated by each evaluator, the way Stwo instantiates `evaluate<E: EvalAtRow>`.
// The header is field-agnostic: E::F only needs + - * and e.c(u32); witness code goes through the
// `O` ops policy (O::M, O::mk, O::inv), so product (cm::M31) and oracle (orc::M31) keep separate
// arithmetic implementations.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AIR_HD __host__ __device__ __forceinline__
#else
#define AIR_HD inline
#endif

namespace air {

// ---- relations (crates/prover/src/relations.rs:7-44; draw order components/mod.rs:311-323) -------
enum RelId : int {
  REL_REGISTERS = 0,  // 3
  REL_MEMORY = 1,     // 6
  REL_MERKLE = 2,     // 4
  REL_POSEIDON2 = 3,  // 16
  REL_RC8 = 4,        // 1
  REL_RC16 = 5,       // 1
  REL_RC20 = 6,       // 1
  REL_BITWISE = 7,    // 4
  N_RELATIONS = 8
};
constexpr int REL_SIZE[N_RELATIONS] = {3, 6, 4, 16, 1, 1, 1, 4};
constexpr int MAX_REL_SIZE = 16;

// z and alpha powers of every relation, as 4 x u32 each (layout-only; evaluators convert).
struct RelationsRaw {
  uint32_t z[N_RELATIONS][4];
  uint32_t alpha_pow[N_RELATIONS][MAX_REL_SIZE][4];
};

// ---- components, in `Components::provers()` order (components/mod.rs:420-431 and the macro order of
// components/opcodes/mod.rs:223-268).  The same order is used for tree-1/tree-2 column order,
// Claim::mix_into and the random-coefficient slices. ---------------------------------------------
enum ComponentId : int {
  C_ASSERT_EQ_FP_IMM = 0,
  C_CALL_ABS_IMM,
  C_JMP_IMM,
  C_JNZ_FP_IMM,
  C_RET,
  C_STORE_IMM,
  C_STORE_FP_FP,
  C_STORE_FP_IMM,
  C_DOUBLE_DEREF_FP_IMM,
  C_DOUBLE_DEREF_FP_FP,
  C_STORE_FRAME_POINTER,
  C_U32_STORE_IMM,
  C_U32_STORE_ADD_FP_IMM,
  C_U32_STORE_MUL_FP_IMM,
  C_U32_STORE_DIV_FP_IMM,
  C_U32_STORE_EQ_FP_FP,
  C_U32_STORE_EQ_FP_IMM,
  C_U32_STORE_LT_FP_IMM,
  C_U32_STORE_LT_FP_FP,
  C_U32_STORE_ADD_FP_FP,
  C_U32_STORE_SUB_FP_FP,
  C_U32_STORE_MUL_FP_FP,
  C_U32_STORE_DIV_FP_FP,
  C_U32_STORE_BITWISE_FP_FP,
  C_U32_STORE_BITWISE_FP_IMM,
  C_STORE_LE_FP_IMM,
  C_MEMORY,  // 26
  C_MERKLE,
  C_CLOCK_UPDATE,
  C_POSEIDON2,
  C_RC8,
  C_RC16,
  C_RC20,
  C_BITWISE,
  N_COMPONENTS  // 34
};
constexpr int N_OPCODE_COMPONENTS = 26;

// preprocessed columns, tree-0 order (crates/prover/src/preprocessed/mod.rs:75-82)
enum PreprocId : int { PP_BITWISE_0 = 0, PP_BITWISE_1, PP_BITWISE_2, PP_BITWISE_3, PP_RC8, PP_RC16, PP_RC20, N_PREPROC };
constexpr uint32_t PREPROC_LOG[N_PREPROC] = {18, 18, 18, 18, 8, 16, 20};

// opcode ids (crates/common/src/instruction.rs:314-577)
constexpr uint32_t OP_STORE_ADD_FP_FP = 0, OP_STORE_ADD_FP_IMM = 4, OP_STORE_MUL_FP_IMM = 6, OP_STORE_DOUBLE_DEREF_FP = 8,
                   OP_STORE_IMM = 9, OP_CALL_ABS_IMM = 10, OP_RET = 11, OP_JMP_ABS_IMM = 12, OP_JMP_REL_IMM = 13,
                   OP_JNZ_FP_IMM = 14, OP_U32_STORE_ADD_FP_FP = 15, OP_U32_STORE_SUB_FP_FP = 16,
                   OP_U32_STORE_MUL_FP_FP = 17, OP_U32_STORE_DIV_REM_FP_FP = 18, OP_U32_STORE_ADD_FP_IMM = 19,
                   OP_U32_STORE_MUL_FP_IMM = 21, OP_U32_STORE_DIV_REM_FP_IMM = 22, OP_U32_STORE_IMM = 23,
                   OP_U32_STORE_EQ_FP_FP = 24, OP_U32_STORE_LT_FP_FP = 28, OP_U32_STORE_EQ_FP_IMM = 30,
                   OP_U32_STORE_LT_FP_IMM = 34, OP_U32_STORE_AND_FP_FP = 36, OP_U32_STORE_AND_FP_IMM = 39,
                   OP_STORE_DOUBLE_DEREF_FP_FP = 42, OP_STORE_FRAME_POINTER = 43, OP_STORE_TO_DOUBLE_DEREF_FP_IMM = 44,
                   OP_STORE_TO_DOUBLE_DEREF_FP_FP = 45, OP_STORE_LE_FP_IMM = 48, OP_ASSERT_EQ_FP_IMM = 50;

// opcode -> opcode component (components/opcodes/mod.rs:223-268); -1 = no component
AIR_HD int component_of_opcode(uint32_t op) {
  switch (op) {
    case 50: return C_ASSERT_EQ_FP_IMM;
    case 10: return C_CALL_ABS_IMM;
    case 12: case 13: return C_JMP_IMM;
    case 14: return C_JNZ_FP_IMM;
    case 11: return C_RET;
    case 9: return C_STORE_IMM;
    case 0: case 1: case 2: case 3: return C_STORE_FP_FP;
    case 4: case 6: return C_STORE_FP_IMM;
    case 8: case 44: return C_DOUBLE_DEREF_FP_IMM;
    case 42: case 45: return C_DOUBLE_DEREF_FP_FP;
    case 43: return C_STORE_FRAME_POINTER;
    case 23: return C_U32_STORE_IMM;
    case 19: return C_U32_STORE_ADD_FP_IMM;
    case 21: return C_U32_STORE_MUL_FP_IMM;
    case 22: return C_U32_STORE_DIV_FP_IMM;
    case 24: return C_U32_STORE_EQ_FP_FP;
    case 30: return C_U32_STORE_EQ_FP_IMM;
    case 34: return C_U32_STORE_LT_FP_IMM;
    case 28: return C_U32_STORE_LT_FP_FP;
    case 15: return C_U32_STORE_ADD_FP_FP;
    case 16: return C_U32_STORE_SUB_FP_FP;
    case 17: return C_U32_STORE_MUL_FP_FP;
    case 18: return C_U32_STORE_DIV_FP_FP;
    case 36: case 37: case 38: return C_U32_STORE_BITWISE_FP_FP;
    case 39: case 40: case 41: return C_U32_STORE_BITWISE_FP_IMM;
    case 48: return C_STORE_LE_FP_IMM;
    default: return -1;
  }
}

// ---- witness inputs (C-ABI layouts, include/cairom_hip.h) ---------------------------------------
struct Bundle {
  uint32_t pc, fp, clock, inst_prev_clock;
  uint32_t inst[6];
  uint32_t span_start, span_len;
};
struct Access {
  uint32_t address, prev_clock, prev_value, value;
};
// ExecutionBundle::default(): {pc=fp=clock=0, Ret, prev_clock 0, span (0,0)} (adapter/memory.rs:112-124)
AIR_HD Bundle default_bundle() {
  Bundle b;
  b.pc = b.fp = b.clock = b.inst_prev_clock = 0;
  b.inst[0] = OP_RET;
  b.inst[1] = b.inst[2] = b.inst[3] = b.inst[4] = b.inst[5] = 0;
  b.span_start = 0;
  b.span_len = 0;
  return b;
}
// get_access_field (crates/prover/src/utils/data_accesses.rs:10-28): k-th access of the span or 0
AIR_HD Access access_at(const Bundle& b, const Access* acc, uint32_t k) {
  if (k < b.span_len) return acc[b.span_start + k];
  Access z;
  z.address = z.prev_clock = z.prev_value = z.value = 0;
  return z;
}

constexpr uint32_t RC20_LIMIT = (1u << 20) - 1;  // adapter/memory.rs:15
constexpr uint32_t TREE_HEIGHT = 30;             // adapter/merkle.rs:63
constexpr uint32_t M31_P = 0x7fffffffu;
// compile-time M31 inverse (x^(P-2)) for AIR constants such as (44-8)^-1
constexpr uint32_t m31_mul_const(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % M31_P); }
constexpr uint32_t m31_inv_const(uint32_t x) {
  uint32_t r = 1, b = x % M31_P;
  uint32_t e = M31_P - 2;
  while (e) {
    if (e & 1) r = m31_mul_const(r, b);
    b = m31_mul_const(b, b);
    e >>= 1;
  }
  return r;
}

}  // namespace air
