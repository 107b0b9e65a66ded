// AIR of the u32 opcode components (two 16-bit limbs per u32; 8-bit limbs for mul/div/bitwise).
// Restated from /root/reference/crates/prover/src/components/opcodes/u32_store_*.rs — see the header of
// opcodes_felt.hpp for the eval/witness convention.  Reference quirks are kept verbatim and flagged
// "QUIRK" (they change committed cells / constraint values, so bit-exactness requires them).
#pragma once
#include "air_common.hpp"

namespace air {

constexpr uint32_t TWO16 = 1u << 16, TWO8 = 1u << 8;
constexpr uint32_t MAX_CARRY_0 = 254, MAX_CARRY_1 = 509, MAX_CARRY_2 = 764, MAX_CARRY_3 = 1019,
                   MAX_CARRY_4 = 765, MAX_CARRY_5 = 510, MAX_CARRY_6 = 255;  // u32_store_div_fp_fp.rs:190-196

#define AIR_COMMON5(o, b, enabler) \
  o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock)
#define AIR_EVAL5(e) \
  F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next()

// ------------------------------------------------------------------------------------------------
// u32_store_imm.rs (opcode 23) — 12 columns.  witness :119-215, eval :430-566
struct U32StoreImm {
  static constexpr int N_TRACE = 12;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]);
    o[8] = O::mk(a0.prev_value); o[9] = O::mk(a1.prev_value); o[10] = O::mk(a0.prev_clock); o[11] = O::mk(a1.prev_clock);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_U32_STORE_IMM);
    AIR_EVAL5(e);
    F imm_lo = e.next(), imm_hi = e.next(), dst_off = e.next(), dst_prev_val_lo = e.next(), dst_prev_val_hi = e.next();
    F dst_prev_clock_lo = e.next(), dst_prev_clock_hi = e.next();
    e.constraint(enabler * (one - enabler));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, imm_lo, imm_hi, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, imm_lo, imm_hi, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock_lo, dst_prev_val_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, imm_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dst_prev_clock_hi, dst_prev_val_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, imm_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, imm_lo);
    e.rel(REL_RC16, m1, imm_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock_lo - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock_hi - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_add_fp_fp.rs (15) / u32_store_sub_fp_fp.rs (16) — 22 columns.  add: witness :155-390, eval :540-809
template <bool SUB>
struct U32StoreAddSubFpFp {
  static constexpr int N_TRACE = 22;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    Access a4 = access_at(b, acc, 4), a5 = access_at(b, acc, 5);
    uint32_t c0, c1;
    if (!SUB) {
      c0 = (O::mk(a0.value) + O::mk(a2.value)).v > 0xFFFF;
      c1 = (O::mk(a1.value) + O::mk(a3.value) + O::mk(c0)).v > 0xFFFF;
    } else {
      c0 = a0.value < a2.value;
      c1 = a1.value < (O::mk(a3.value) + O::mk(c0)).v;
    }
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]);
    o[8] = O::mk(a0.value); o[9] = O::mk(a1.value); o[10] = O::mk(a0.prev_clock); o[11] = O::mk(a1.prev_clock);
    o[12] = O::mk(a2.value); o[13] = O::mk(a3.value); o[14] = O::mk(a2.prev_clock); o[15] = O::mk(a3.prev_clock);
    o[16] = O::mk(a4.prev_value); o[17] = O::mk(a5.prev_value); o[18] = O::mk(a4.prev_clock); o[19] = O::mk(a5.prev_clock);
    o[20] = O::mk(c0); o[21] = O::mk(c1);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two16 = e.c(TWO16), opc = e.c(SUB ? OP_U32_STORE_SUB_FP_FP : OP_U32_STORE_ADD_FP_FP);
    AIR_EVAL5(e);
    F src0_off = e.next(), src1_off = e.next(), dst_off = e.next();
    F op0_lo = e.next(), op0_hi = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F op1_lo = e.next(), op1_hi = e.next(), op1_pc_lo = e.next(), op1_pc_hi = e.next();
    F dpv_lo = e.next(), dpv_hi = e.next(), dpc_lo = e.next(), dpc_hi = e.next();
    F c0 = e.next(), c1 = e.next();
    F res_lo = SUB ? (op0_lo + c0 * two16 - op1_lo) : (op0_lo + op1_lo - c0 * two16);
    F res_hi = SUB ? (op0_hi - c0 + c1 * two16 - op1_hi) : (op0_hi + op1_hi + c0 - c1 * two16);
    e.constraint(enabler * (one - enabler));
    e.constraint(c0 * (one - c0));
    e.constraint(c1 * (one - c1));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + src1_off, op1_pc_lo, op1_lo);
    e.rel(REL_MEMORY, enabler, fp + src1_off, clock, op1_lo);
    e.rel(REL_MEMORY, -enabler, fp + src1_off + one, op1_pc_hi, op1_hi);
    e.rel(REL_MEMORY, enabler, fp + src1_off + one, clock, op1_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, res_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, res_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, op0_lo);
    e.rel(REL_RC16, m1, op0_hi);
    e.rel(REL_RC16, m1, op1_lo);
    e.rel(REL_RC16, m1, op1_hi);
    e.rel(REL_RC16, m1, res_lo);
    e.rel(REL_RC16, m1, res_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.finalize_pairs();
  }
};
using U32StoreAddFpFp = U32StoreAddSubFpFp<false>;
using U32StoreSubFpFp = U32StoreAddSubFpFp<true>;

// ------------------------------------------------------------------------------------------------
// u32_store_add_fp_imm.rs (19) — 19 columns.  witness :150-360, eval :500-746
struct U32StoreAddFpImm {
  static constexpr int N_TRACE = 19;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    uint32_t c0 = (O::mk(a0.value) + O::mk(b.inst[2])).v > 0xFFFF;
    uint32_t c1 = (O::mk(a1.value) + O::mk(b.inst[3]) + O::mk(c0)).v > 0xFFFF;
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]); o[8] = O::mk(b.inst[4]);
    o[9] = O::mk(a0.value); o[10] = O::mk(a1.value); o[11] = O::mk(a0.prev_clock); o[12] = O::mk(a1.prev_clock);
    o[13] = O::mk(a2.prev_value); o[14] = O::mk(a3.prev_value); o[15] = O::mk(a2.prev_clock); o[16] = O::mk(a3.prev_clock);
    o[17] = O::mk(c0); o[18] = O::mk(c1);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two16 = e.c(TWO16), opc = e.c(OP_U32_STORE_ADD_FP_IMM);
    AIR_EVAL5(e);
    F src_off = e.next(), imm_lo = e.next(), imm_hi = e.next(), dst_off = e.next();
    F op0_lo = e.next(), op0_hi = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F dpv_lo = e.next(), dpv_hi = e.next(), dpc_lo = e.next(), dpc_hi = e.next(), c0 = e.next(), c1 = e.next();
    F res_lo = op0_lo + imm_lo - c0 * two16;
    F res_hi = op0_hi + imm_hi + c0 - c1 * two16;
    e.constraint(enabler * (one - enabler));
    e.constraint(c0 * (one - c0));
    e.constraint(c1 * (one - c1));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, -enabler, pc + one, inst_prev_clock, dst_off);
    e.rel(REL_MEMORY, enabler, pc + one, clock, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, res_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, res_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, op0_lo);
    e.rel(REL_RC16, m1, op0_hi);
    e.rel(REL_RC16, m1, imm_lo);
    e.rel(REL_RC16, m1, imm_hi);
    e.rel(REL_RC16, m1, res_lo);
    e.rel(REL_RC16, m1, res_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_eq_fp_fp.rs (24) — 22 columns.  witness :150-330, eval :480-746
// QUIRK: dst_off is read from inst_value_4 (u32_store_eq_fp_fp.rs:210), which is 0 for this 4-word instruction.
struct U32StoreEqFpFp {
  static constexpr int N_TRACE = 22;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    Access a4 = access_at(b, acc, 4);
    M one = O::mk(1);
    M diff_lo = O::mk(a2.value) - O::mk(a0.value), diff_hi = O::mk(a3.value) - O::mk(a1.value);
    M inv_lo = diff_lo.v != 0 ? O::inv(diff_lo) : O::mk(0), inv_hi = diff_hi.v != 0 ? O::inv(diff_hi) : O::mk(0);
    M is_eq_lo = one - diff_lo * inv_lo;
    M is_eq_prod = is_eq_lo * (one - diff_hi * inv_hi);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[4]);
    o[8] = O::mk(a0.value); o[9] = O::mk(a1.value); o[10] = O::mk(a0.prev_clock); o[11] = O::mk(a1.prev_clock);
    o[12] = O::mk(a2.value); o[13] = O::mk(a3.value); o[14] = O::mk(a2.prev_clock); o[15] = O::mk(a3.prev_clock);
    o[16] = O::mk(a4.prev_value); o[17] = O::mk(a4.prev_clock);
    o[18] = inv_lo; o[19] = inv_hi; o[20] = is_eq_lo; o[21] = is_eq_prod;
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_U32_STORE_EQ_FP_FP);
    AIR_EVAL5(e);
    F src0_off = e.next(), src1_off = e.next(), dst_off = e.next();
    F op0_lo = e.next(), op0_hi = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F op1_lo = e.next(), op1_hi = e.next(), op1_pc_lo = e.next(), op1_pc_hi = e.next();
    F dst_prev_val = e.next(), dst_prev_clock = e.next();
    F diff_inv_lo = e.next(), diff_inv_hi = e.next(), is_eq_lo = e.next(), is_eq_prod = e.next();
    e.constraint(enabler * (one - enabler));
    F diff_lo = op1_lo - op0_lo, diff_hi = op1_hi - op0_hi;
    e.constraint(diff_lo * (diff_inv_lo * diff_lo - one));
    e.constraint(diff_hi * (diff_inv_hi * diff_hi - one));
    e.constraint(is_eq_lo - (one - diff_lo * diff_inv_lo));
    e.constraint(is_eq_prod - is_eq_lo * (one - diff_hi * diff_inv_hi));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + src1_off, op1_pc_lo, op1_lo);
    e.rel(REL_MEMORY, enabler, fp + src1_off, clock, op1_lo);
    e.rel(REL_MEMORY, -enabler, fp + src1_off + one, op1_pc_hi, op1_hi);
    e.rel(REL_MEMORY, enabler, fp + src1_off + one, clock, op1_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, is_eq_prod);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, op0_lo);
    e.rel(REL_RC16, m1, op0_hi);
    e.rel(REL_RC16, m1, op1_lo);
    e.rel(REL_RC16, m1, op1_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_eq_fp_imm.rs (30) — 16 columns.  witness :140-290, eval :430-632
// QUIRK: the second instruction word is looked up at `pc` (not pc+1) (u32_store_eq_fp_imm.rs eval).
struct U32StoreEqFpImm {
  static constexpr int N_TRACE = 16;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2);
    M two16 = O::mk(TWO16);
    M diff = O::mk(a0.value) + O::mk(a1.value) * two16 - O::mk(b.inst[2]) - O::mk(b.inst[3]) * two16;
    M diff_inv = diff.v != 0 ? O::inv(diff) : O::mk(0);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]); o[8] = O::mk(b.inst[4]);
    o[9] = O::mk(a0.value); o[10] = O::mk(a1.value); o[11] = O::mk(a0.prev_clock); o[12] = O::mk(a1.prev_clock);
    o[13] = O::mk(a2.prev_value); o[14] = O::mk(a2.prev_clock); o[15] = diff_inv;
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two16 = e.c(TWO16), opc = e.c(OP_U32_STORE_EQ_FP_IMM);
    AIR_EVAL5(e);
    F src0_off = e.next(), imm_lo = e.next(), imm_hi = e.next(), dst_off = e.next();
    F op0_lo = e.next(), op0_hi = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F dst_prev_val = e.next(), dst_prev_clock = e.next(), diff_inv = e.next();
    e.constraint(enabler * (one - enabler));
    F diff = op0_lo + op0_hi * two16 - imm_lo - imm_hi * two16;
    e.constraint(diff * (diff_inv * diff - one));
    e.constraint(diff_inv * (diff_inv * diff - one));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, one - diff * diff_inv);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, op0_lo);
    e.rel(REL_RC16, m1, op0_hi);
    e.rel(REL_RC16, m1, imm_lo);
    e.rel(REL_RC16, m1, imm_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_lt_fp_fp.rs (28) — 20 columns.  witness :150-330 (borrow folds enabler :223-237), eval :510-794
struct U32StoreLtFpFp {
  static constexpr int N_TRACE = 20;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    Access a4 = access_at(b, acc, 4);
    uint32_t borrow_lo = a2.value < a0.value + enabler;
    uint32_t borrow_hi = a3.value < a1.value + borrow_lo;
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]);
    o[8] = O::mk(a0.value); o[9] = O::mk(a1.value); o[10] = O::mk(a0.prev_clock); o[11] = O::mk(a1.prev_clock);
    o[12] = O::mk(a2.value); o[13] = O::mk(a3.value); o[14] = O::mk(a2.prev_clock); o[15] = O::mk(a3.prev_clock);
    o[16] = O::mk(a4.prev_value); o[17] = O::mk(a4.prev_clock); o[18] = O::mk(borrow_lo); o[19] = O::mk(borrow_hi);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two16 = e.c(TWO16), opc = e.c(OP_U32_STORE_LT_FP_FP);
    AIR_EVAL5(e);
    F src0_off = e.next(), src1_off = e.next(), dst_off = e.next();
    F op0_lo = e.next(), op0_hi = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F op1_lo = e.next(), op1_hi = e.next(), op1_pc_lo = e.next(), op1_pc_hi = e.next();
    F dst_prev_val = e.next(), dst_prev_clock = e.next(), borrow_lo = e.next(), borrow_hi = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(borrow_lo * (one - borrow_lo));
    e.constraint(borrow_hi * (one - borrow_hi));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + src1_off, op1_pc_lo, op1_lo);
    e.rel(REL_MEMORY, enabler, fp + src1_off, clock, op1_lo);
    e.rel(REL_MEMORY, -enabler, fp + src1_off + one, op1_pc_hi, op1_hi);
    e.rel(REL_MEMORY, enabler, fp + src1_off + one, clock, op1_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, one - borrow_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, op0_lo);
    e.rel(REL_RC16, m1, op0_hi);
    e.rel(REL_RC16, m1, op1_lo);
    e.rel(REL_RC16, m1, op1_hi);
    e.rel(REL_RC16, m1, op1_lo - enabler + borrow_lo * two16 - op0_lo);
    e.rel(REL_RC16, m1, op1_hi - borrow_lo + borrow_hi * two16 - op0_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_lt_fp_imm.rs (34) — 17 columns.  witness :145-300, eval :460-698
struct U32StoreLtFpImm {
  static constexpr int N_TRACE = 17;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2);
    uint32_t borrow_lo = b.inst[2] < a0.value + enabler;
    uint32_t borrow_hi = b.inst[3] < a1.value + borrow_lo;
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]); o[8] = O::mk(b.inst[4]);
    o[9] = O::mk(a0.value); o[10] = O::mk(a1.value); o[11] = O::mk(a0.prev_clock); o[12] = O::mk(a1.prev_clock);
    o[13] = O::mk(a2.prev_value); o[14] = O::mk(a2.prev_clock); o[15] = O::mk(borrow_lo); o[16] = O::mk(borrow_hi);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two16 = e.c(TWO16), opc = e.c(OP_U32_STORE_LT_FP_IMM);
    AIR_EVAL5(e);
    F src_off = e.next(), imm_lo = e.next(), imm_hi = e.next(), dst_off = e.next();
    F op0_lo = e.next(), op0_hi = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F dst_prev_val = e.next(), dst_prev_clock = e.next(), borrow_lo = e.next(), borrow_hi = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(borrow_lo * (one - borrow_lo));
    e.constraint(borrow_hi * (one - borrow_hi));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, -enabler, pc + one, inst_prev_clock, dst_off);
    e.rel(REL_MEMORY, enabler, pc + one, clock, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, one - borrow_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, op0_lo);
    e.rel(REL_RC16, m1, op0_hi);
    e.rel(REL_RC16, m1, imm_lo);
    e.rel(REL_RC16, m1, imm_hi);
    e.rel(REL_RC16, m1, imm_lo - enabler + borrow_lo * two16 - op0_lo);
    e.rel(REL_RC16, m1, imm_hi - borrow_lo + borrow_hi * two16 - op0_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// 8-bit schoolbook product of two u32 given as 4 limbs each, low 4 result limbs + carries
// (u32_store_mul_fp_fp.rs:268-290).  All intermediate values < 2^19, so plain u32 arithmetic is exact.
AIR_HD void mul_limbs_8(const uint32_t a[4], const uint32_t b[4], uint32_t res[4], uint32_t carry[4]) {
  uint32_t s0 = a[0] * b[0];
  carry[0] = s0 >> 8; res[0] = s0 - carry[0] * 256;
  uint32_t s1 = a[0] * b[1] + a[1] * b[0] + carry[0];
  carry[1] = s1 >> 8; res[1] = s1 - carry[1] * 256;
  uint32_t s2 = a[0] * b[2] + a[1] * b[1] + a[2] * b[0] + carry[1];
  carry[2] = s2 >> 8; res[2] = s2 - carry[2] * 256;
  uint32_t s3 = a[0] * b[3] + a[1] * b[2] + a[2] * b[1] + a[3] * b[0] + carry[2];
  carry[3] = s3 >> 8; res[3] = s3 - carry[3] * 256;
}

// ------------------------------------------------------------------------------------------------
// u32_store_mul_fp_fp.rs (17) — 32 columns.  witness :205-420, eval :700-1062
// NB decompose_8 here is (x & 0xFF, x >> 8) (no mask on the high limb) — u32_store_mul_fp_fp.rs:255-259.
struct U32StoreMulFpFp {
  static constexpr int N_TRACE = 32;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    Access a4 = access_at(b, acc, 4), a5 = access_at(b, acc, 5);
    uint32_t x[4] = {a0.value & 0xFF, a0.value >> 8, a1.value & 0xFF, a1.value >> 8};
    uint32_t y[4] = {a2.value & 0xFF, a2.value >> 8, a3.value & 0xFF, a3.value >> 8};
    uint32_t res[4], carry[4];
    mul_limbs_8(x, y, res, carry);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]);
    for (int i = 0; i < 4; i++) o[8 + i] = O::mk(x[i]);
    o[12] = O::mk(a0.prev_clock); o[13] = O::mk(a1.prev_clock);
    for (int i = 0; i < 4; i++) o[14 + i] = O::mk(y[i]);
    o[18] = O::mk(a2.prev_clock); o[19] = O::mk(a3.prev_clock);
    o[20] = O::mk(a4.prev_value); o[21] = O::mk(a5.prev_value); o[22] = O::mk(a4.prev_clock); o[23] = O::mk(a5.prev_clock);
    for (int i = 0; i < 4; i++) { o[24 + i] = O::mk(res[i]); o[28 + i] = O::mk(carry[i]); }
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two8 = e.c(TWO8), opc = e.c(OP_U32_STORE_MUL_FP_FP);
    AIR_EVAL5(e);
    F src0_off = e.next(), src1_off = e.next(), dst_off = e.next();
    F op0_0 = e.next(), op0_1 = e.next(), op0_2 = e.next(), op0_3 = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F op1_0 = e.next(), op1_1 = e.next(), op1_2 = e.next(), op1_3 = e.next(), op1_pc_lo = e.next(), op1_pc_hi = e.next();
    F dpv_lo = e.next(), dpv_hi = e.next(), dpc_lo = e.next(), dpc_hi = e.next();
    F res_0 = e.next(), res_1 = e.next(), res_2 = e.next(), res_3 = e.next();
    F carry_0 = e.next(), carry_1 = e.next(), carry_2 = e.next(), carry_3 = e.next();
    (void)op1_pc_lo; (void)op1_pc_hi;
    e.constraint(enabler * (one - enabler));
    e.constraint(enabler * (res_0 - (op0_0 * op1_0 - carry_0 * two8)));
    e.constraint(enabler * (res_1 - (op0_0 * op1_1 + op0_1 * op1_0 + carry_0 - carry_1 * two8)));
    e.constraint(enabler * (res_2 - (op0_0 * op1_2 + op0_1 * op1_1 + op0_2 * op1_0 + carry_1 - carry_2 * two8)));
    e.constraint(enabler * (res_3 - (op0_0 * op1_3 + op0_1 * op1_2 + op0_2 * op1_1 + op0_3 * op1_0 + carry_2 - carry_3 * two8)));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, op0_0 + op0_1 * two8);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_0 + op0_1 * two8);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, op0_2 + op0_3 * two8);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, op0_2 + op0_3 * two8);
    e.rel(REL_MEMORY, -enabler, fp + src1_off, op1_pc_lo, op1_0 + op1_1 * two8);
    e.rel(REL_MEMORY, enabler, fp + src1_off, clock, op1_0 + op1_1 * two8);
    e.rel(REL_MEMORY, -enabler, fp + src1_off + one, op1_pc_hi, op1_2 + op1_3 * two8);
    e.rel(REL_MEMORY, enabler, fp + src1_off + one, clock, op1_2 + op1_3 * two8);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, res_0 + res_1 * two8);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, res_2 + res_3 * two8);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC8, m1, op0_0); e.rel(REL_RC8, m1, op0_1); e.rel(REL_RC8, m1, op0_2); e.rel(REL_RC8, m1, op0_3);
    e.rel(REL_RC8, m1, op1_0); e.rel(REL_RC8, m1, op1_1); e.rel(REL_RC8, m1, op1_2); e.rel(REL_RC8, m1, op1_3);
    e.rel(REL_RC8, m1, res_0); e.rel(REL_RC8, m1, res_1); e.rel(REL_RC8, m1, res_2); e.rel(REL_RC8, m1, res_3);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_0) - carry_0);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_1) - carry_1);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_2) - carry_2);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_3) - carry_3);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_mul_fp_imm.rs (21) — 29 columns.  witness :195-400, eval :650-989
struct U32StoreMulFpImm {
  static constexpr int N_TRACE = 29;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    uint32_t x[4] = {a0.value & 0xFF, a0.value >> 8, a1.value & 0xFF, a1.value >> 8};
    uint32_t y[4] = {b.inst[2] & 0xFF, b.inst[2] >> 8, b.inst[3] & 0xFF, b.inst[3] >> 8};
    uint32_t res[4], carry[4];
    mul_limbs_8(x, y, res, carry);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[1]);
    for (int i = 0; i < 4; i++) o[6 + i] = O::mk(y[i]);
    o[10] = O::mk(b.inst[4]);
    for (int i = 0; i < 4; i++) o[11 + i] = O::mk(x[i]);
    o[15] = O::mk(a0.prev_clock); o[16] = O::mk(a1.prev_clock);
    o[17] = O::mk(a2.prev_value); o[18] = O::mk(a3.prev_value); o[19] = O::mk(a2.prev_clock); o[20] = O::mk(a3.prev_clock);
    for (int i = 0; i < 4; i++) { o[21 + i] = O::mk(res[i]); o[25 + i] = O::mk(carry[i]); }
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two8 = e.c(TWO8), opc = e.c(OP_U32_STORE_MUL_FP_IMM);
    AIR_EVAL5(e);
    F src_off = e.next(), imm_0 = e.next(), imm_1 = e.next(), imm_2 = e.next(), imm_3 = e.next(), dst_off = e.next();
    F op0_0 = e.next(), op0_1 = e.next(), op0_2 = e.next(), op0_3 = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F dpv_lo = e.next(), dpv_hi = e.next(), dpc_lo = e.next(), dpc_hi = e.next();
    F res_0 = e.next(), res_1 = e.next(), res_2 = e.next(), res_3 = e.next();
    F carry_0 = e.next(), carry_1 = e.next(), carry_2 = e.next(), carry_3 = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(enabler * (res_0 - (op0_0 * imm_0 - carry_0 * two8)));
    e.constraint(enabler * (res_1 - (op0_0 * imm_1 + op0_1 * imm_0 + carry_0 - carry_1 * two8)));
    e.constraint(enabler * (res_2 - (op0_0 * imm_2 + op0_1 * imm_1 + op0_2 * imm_0 + carry_1 - carry_2 * two8)));
    e.constraint(enabler * (res_3 - (op0_0 * imm_3 + op0_1 * imm_2 + op0_2 * imm_1 + op0_3 * imm_0 + carry_2 - carry_3 * two8)));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src_off, imm_0 + imm_1 * two8, imm_2 + imm_3 * two8);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src_off, imm_0 + imm_1 * two8, imm_2 + imm_3 * two8);
    e.rel(REL_MEMORY, -enabler, pc + one, inst_prev_clock, dst_off);
    e.rel(REL_MEMORY, enabler, pc + one, clock, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src_off, op0_pc_lo, op0_0 + op0_1 * two8);
    e.rel(REL_MEMORY, enabler, fp + src_off, clock, op0_0 + op0_1 * two8);
    e.rel(REL_MEMORY, -enabler, fp + src_off + one, op0_pc_hi, op0_2 + op0_3 * two8);
    e.rel(REL_MEMORY, enabler, fp + src_off + one, clock, op0_2 + op0_3 * two8);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, res_0 + res_1 * two8);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, res_2 + res_3 * two8);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC8, m1, op0_0); e.rel(REL_RC8, m1, op0_1); e.rel(REL_RC8, m1, op0_2); e.rel(REL_RC8, m1, op0_3);
    e.rel(REL_RC8, m1, imm_0); e.rel(REL_RC8, m1, imm_1); e.rel(REL_RC8, m1, imm_2); e.rel(REL_RC8, m1, imm_3);
    e.rel(REL_RC8, m1, res_0); e.rel(REL_RC8, m1, res_1); e.rel(REL_RC8, m1, res_2); e.rel(REL_RC8, m1, res_3);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_0) - carry_0);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_1) - carry_1);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_2) - carry_2);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_3) - carry_3);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32 div/rem core (u32_store_div_fp_fp.rs:300-470): q*d + r = n with 8-bit limb products.
struct DivCore {
  uint32_t d[4], q[4], mul_carry[7], prod[8], add_carry[4], sub_borrow[2], r_lo, r_hi;
};
AIR_HD DivCore div_core(uint32_t n_lo, uint32_t n_hi, uint32_t d_lo, uint32_t d_hi) {
  DivCore c;
  c.d[0] = d_lo & 0xFF; c.d[1] = (d_lo >> 8) & 0xFF; c.d[2] = d_hi & 0xFF; c.d[3] = (d_hi >> 8) & 0xFF;
  uint32_t n = n_lo | (n_hi << 16), dv = d_lo | (d_hi << 16);
  uint32_t qv = dv == 0 ? 0 : n / dv, rv = dv == 0 ? 0 : n % dv;  // division by zero -> (0, 0) (:362-364)
  uint32_t q_lo = qv & 0xFFFF, q_hi = qv >> 16;
  c.r_lo = rv & 0xFFFF; c.r_hi = rv >> 16;
  c.q[0] = q_lo & 0xFF; c.q[1] = (q_lo >> 8) & 0xFF; c.q[2] = q_hi & 0xFF; c.q[3] = (q_hi >> 8) & 0xFF;
  const uint32_t* q = c.q; const uint32_t* d = c.d;
  uint32_t raw[7] = {q[0] * d[0],
                     q[0] * d[1] + q[1] * d[0],
                     q[0] * d[2] + q[2] * d[0] + q[1] * d[1],
                     q[0] * d[3] + q[3] * d[0] + q[1] * d[2] + q[2] * d[1],
                     q[1] * d[3] + q[3] * d[1] + q[2] * d[2],
                     q[2] * d[3] + q[3] * d[2],
                     q[3] * d[3]};
  uint32_t carry = 0;
  for (int i = 0; i < 7; i++) {
    uint32_t with = raw[i] + carry;
    c.mul_carry[i] = with >> 8;
    c.prod[i] = with - c.mul_carry[i] * 256;
    carry = c.mul_carry[i];
  }
  c.prod[7] = c.mul_carry[6];
  uint32_t a0 = c.prod[0] + c.prod[1] * 256 + c.r_lo;
  c.add_carry[0] = a0 > 0xFFFF;
  uint32_t a1 = c.prod[2] + c.prod[3] * 256 + c.r_hi + c.add_carry[0];
  c.add_carry[1] = a1 > 0xFFFF;
  uint32_t a2 = c.prod[4] + c.prod[5] * 256 + c.add_carry[1];
  c.add_carry[2] = a2 > 0xFFFF;
  uint32_t a3 = c.prod[6] + c.prod[7] * 256 + c.add_carry[2];
  c.add_carry[3] = a3 > 0xFFFF;
  c.sub_borrow[0] = (d[0] | (d[1] << 8)) < c.r_lo + 1;
  c.sub_borrow[1] = (d[2] | (d[3] << 8)) < c.r_hi + c.sub_borrow[0];
  return c;
}

// u32_store_div_fp_fp.rs (18) — 54 columns (witness :257-560, eval :1000-1513)
// u32_store_div_fp_imm.rs (22) — 51 columns (witness :250-540, eval :950-1440)
template <bool IMM>
struct U32StoreDivT {
  static constexpr int N_TRACE = IMM ? 51 : 54;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1);
    AIR_COMMON5(o, b, enabler);
    int k;
    DivCore c;
    if (!IMM) {
      Access a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3), a4 = access_at(b, acc, 4), a5 = access_at(b, acc, 5);
      Access a6 = access_at(b, acc, 6), a7 = access_at(b, acc, 7);
      c = div_core(a0.value, a1.value, a2.value, a3.value);
      o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]); o[8] = O::mk(b.inst[4]);
      o[9] = O::mk(a0.value); o[10] = O::mk(a1.value); o[11] = O::mk(a0.prev_clock); o[12] = O::mk(a1.prev_clock);
      for (int i = 0; i < 4; i++) o[13 + i] = O::mk(c.d[i]);
      o[17] = O::mk(a2.prev_clock); o[18] = O::mk(a3.prev_clock);
      o[19] = O::mk(a4.prev_value); o[20] = O::mk(a5.prev_value); o[21] = O::mk(a4.prev_clock); o[22] = O::mk(a5.prev_clock);
      o[23] = O::mk(a6.prev_value); o[24] = O::mk(a7.prev_value); o[25] = O::mk(a6.prev_clock); o[26] = O::mk(a7.prev_clock);
      k = 27;
    } else {
      Access a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3), a4 = access_at(b, acc, 4), a5 = access_at(b, acc, 5);
      c = div_core(a0.value, a1.value, b.inst[2], b.inst[3]);
      o[5] = O::mk(b.inst[1]);
      for (int i = 0; i < 4; i++) o[6 + i] = O::mk(c.d[i]);
      o[10] = O::mk(b.inst[4]);
      o[11] = O::mk(a4.address) - O::mk(b.fp);  // dst_rem_off = address(access 4) - fp (u32_store_div_fp_imm.rs)
      o[12] = O::mk(a0.value); o[13] = O::mk(a1.value); o[14] = O::mk(a0.prev_clock); o[15] = O::mk(a1.prev_clock);
      o[16] = O::mk(a2.prev_value); o[17] = O::mk(a3.prev_value); o[18] = O::mk(a2.prev_clock); o[19] = O::mk(a3.prev_clock);
      o[20] = O::mk(a4.prev_value); o[21] = O::mk(a5.prev_value); o[22] = O::mk(a4.prev_clock); o[23] = O::mk(a5.prev_clock);
      k = 24;
    }
    for (int i = 0; i < 4; i++) o[k++] = O::mk(c.q[i]);
    for (int i = 0; i < 7; i++) o[k++] = O::mk(c.mul_carry[i]);
    for (int i = 0; i < 8; i++) o[k++] = O::mk(c.prod[i]);
    for (int i = 0; i < 4; i++) o[k++] = O::mk(c.add_carry[i]);
    o[k++] = O::mk(c.sub_borrow[0]); o[k++] = O::mk(c.sub_borrow[1]);
    o[k++] = O::mk(c.r_lo); o[k++] = O::mk(c.r_hi);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two8 = e.c(TWO8), two16 = e.c(TWO16);
    F opc = e.c(IMM ? OP_U32_STORE_DIV_REM_FP_IMM : OP_U32_STORE_DIV_REM_FP_FP);
    AIR_EVAL5(e);
    F src0_off, src1_off, dst_off, dst_rem_off, n_lo, n_hi, op0_pc_lo, op0_pc_hi, d_0, d_1, d_2, d_3, op1_pc_lo, op1_pc_hi;
    F dpv_lo, dpv_hi, dpc_lo, dpc_hi, rpv_lo, rpv_hi, rpc_lo, rpc_hi;
    if (!IMM) {
      src0_off = e.next(); src1_off = e.next(); dst_off = e.next(); dst_rem_off = e.next();
      n_lo = e.next(); n_hi = e.next(); op0_pc_lo = e.next(); op0_pc_hi = e.next();
      d_0 = e.next(); d_1 = e.next(); d_2 = e.next(); d_3 = e.next(); op1_pc_lo = e.next(); op1_pc_hi = e.next();
    } else {
      src0_off = e.next(); d_0 = e.next(); d_1 = e.next(); d_2 = e.next(); d_3 = e.next();
      dst_off = e.next(); dst_rem_off = e.next();
      n_lo = e.next(); n_hi = e.next(); op0_pc_lo = e.next(); op0_pc_hi = e.next();
    }
    dpv_lo = e.next(); dpv_hi = e.next(); dpc_lo = e.next(); dpc_hi = e.next();
    rpv_lo = e.next(); rpv_hi = e.next(); rpc_lo = e.next(); rpc_hi = e.next();
    F q_0 = e.next(), q_1 = e.next(), q_2 = e.next(), q_3 = e.next();
    F mc0 = e.next(), mc1 = e.next(), mc2 = e.next(), mc3 = e.next(), mc4 = e.next(), mc5 = e.next(), mc6 = e.next();
    F prod_0 = e.next(), prod_1 = e.next(), prod_2 = e.next(), prod_3 = e.next();
    F prod_4 = e.next(), prod_5 = e.next(), prod_6 = e.next(), prod_7 = e.next();
    F ac0 = e.next(), ac1 = e.next(), ac2 = e.next(), ac3 = e.next(), sb0 = e.next(), sb1 = e.next();
    F r_lo = e.next(), r_hi = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(enabler * ac0 * (one - ac0));
    e.constraint(enabler * ac1 * (one - ac1));
    e.constraint(enabler * ac2 * (one - ac2));
    e.constraint(enabler * sb0 * (one - sb0));
    F op1_val_lo = d_0 + d_1 * two8, op1_val_hi = d_2 + d_3 * two8;
    e.constraint(enabler * (q_0 * d_0 - mc0 * two8 - prod_0));
    e.constraint(enabler * (q_0 * d_1 + q_1 * d_0 + mc0 - mc1 * two8 - prod_1));
    e.constraint(enabler * (q_0 * d_2 + q_2 * d_0 + q_1 * d_1 + mc1 - mc2 * two8 - prod_2));
    e.constraint(enabler * (q_0 * d_3 + q_3 * d_0 + q_1 * d_2 + q_2 * d_1 + mc2 - mc3 * two8 - prod_3));
    e.constraint(enabler * (q_1 * d_3 + q_3 * d_1 + q_2 * d_2 + mc3 - mc4 * two8 - prod_4));
    e.constraint(enabler * (q_2 * d_3 + q_3 * d_2 + mc4 - mc5 * two8 - prod_5));
    e.constraint(enabler * (q_3 * d_3 + mc5 - mc6 * two8 - prod_6));
    e.constraint(enabler * (mc6 - prod_7));
    e.constraint(enabler * (n_lo - (prod_0 + prod_1 * two8 + r_lo - ac0 * two16)));
    e.constraint(enabler * (n_hi - (prod_2 + prod_3 * two8 + r_hi + ac0 - ac1 * two16)));
    e.constraint(enabler * (prod_4 + prod_5 * two8 + ac1 - ac2 * two16));
    e.constraint(enabler * (prod_6 + prod_7 * two8 + ac2 - ac3 * two16));
    e.constraint(enabler * ac3);
    F sub_check_lo = d_0 + d_1 * two8 + sb0 * two16 - r_lo - one;
    F sub_check_hi = d_2 + d_3 * two8 + sb1 * two16 - r_hi - sb0;
    e.constraint(enabler * sb1);
    F res_lo = q_0 + q_1 * two8, res_hi = q_2 + q_3 * two8;
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one + one, fp, clock + one);
    if (!IMM) {
      e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, src1_off, dst_off);
      e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, src1_off, dst_off);
      e.rel(REL_MEMORY, -enabler, pc + one, inst_prev_clock, dst_rem_off);
      e.rel(REL_MEMORY, enabler, pc + one, clock, dst_rem_off);
    } else {
      e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, op1_val_lo, op1_val_hi);
      e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, op1_val_lo, op1_val_hi);
      e.rel(REL_MEMORY, -enabler, pc + one, inst_prev_clock, dst_off, dst_rem_off);
      e.rel(REL_MEMORY, enabler, pc + one, clock, dst_off, dst_rem_off);
    }
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, n_lo);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, n_lo);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, n_hi);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, n_hi);
    if (!IMM) {
      e.rel(REL_MEMORY, -enabler, fp + src1_off, op1_pc_lo, op1_val_lo);
      e.rel(REL_MEMORY, enabler, fp + src1_off, clock, op1_val_lo);
      e.rel(REL_MEMORY, -enabler, fp + src1_off + one, op1_pc_hi, op1_val_hi);
      e.rel(REL_MEMORY, enabler, fp + src1_off + one, clock, op1_val_hi);
    }
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, res_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, res_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_rem_off, rpc_lo, rpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_rem_off, clock, r_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_rem_off + one, rpc_hi, rpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_rem_off + one, clock, r_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC8, m1, d_0); e.rel(REL_RC8, m1, d_1); e.rel(REL_RC8, m1, d_2); e.rel(REL_RC8, m1, d_3);
    e.rel(REL_RC8, m1, q_0); e.rel(REL_RC8, m1, q_1); e.rel(REL_RC8, m1, q_2); e.rel(REL_RC8, m1, q_3);
    e.rel(REL_RC8, m1, prod_0); e.rel(REL_RC8, m1, prod_1); e.rel(REL_RC8, m1, prod_2); e.rel(REL_RC8, m1, prod_3);
    e.rel(REL_RC8, m1, prod_4); e.rel(REL_RC8, m1, prod_5); e.rel(REL_RC8, m1, prod_6); e.rel(REL_RC8, m1, prod_7);
    e.rel(REL_RC16, m1, n_lo);
    e.rel(REL_RC16, m1, n_hi);
    e.rel(REL_RC16, m1, r_lo);
    e.rel(REL_RC16, m1, r_hi);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_0) - mc0);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_1) - mc1);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_2) - mc2);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_3) - mc3);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_4) - mc4);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_5) - mc5);
    e.rel(REL_RC16, m1, e.c(MAX_CARRY_6) - mc6);
    e.rel(REL_RC16, m1, sub_check_lo);
    e.rel(REL_RC16, m1, sub_check_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.rel(REL_RC20, m1, clock - rpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - rpc_hi - enabler);
    e.finalize_pairs();
  }
};
using U32StoreDivFpFp = U32StoreDivT<false>;
using U32StoreDivFpImm = U32StoreDivT<true>;

// ------------------------------------------------------------------------------------------------
// u32_store_bitwise_fp_fp.rs (36 And, 37 Or, 38 Xor) — 29 columns.  witness :190-380, eval :560-783
// QUIRK: the enabler constraint is written enabler*(enabler-1) (sign differs from every other component).
struct U32StoreBitwiseFpFp {
  static constexpr int N_TRACE = 29;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    Access a4 = access_at(b, acc, 4), a5 = access_at(b, acc, 5);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[0] == OP_RET ? OP_U32_STORE_AND_FP_FP : b.inst[0]);
    o[6] = O::mk(b.inst[1]); o[7] = O::mk(b.inst[2]); o[8] = O::mk(b.inst[3]);
    o[9] = O::mk(a0.value & 0xFF); o[10] = O::mk((a0.value >> 8) & 0xFF);
    o[11] = O::mk(a1.value & 0xFF); o[12] = O::mk((a1.value >> 8) & 0xFF);
    o[13] = O::mk(a0.prev_clock); o[14] = O::mk(a1.prev_clock);
    o[15] = O::mk(a2.value & 0xFF); o[16] = O::mk((a2.value >> 8) & 0xFF);
    o[17] = O::mk(a3.value & 0xFF); o[18] = O::mk((a3.value >> 8) & 0xFF);
    o[19] = O::mk(a2.prev_clock); o[20] = O::mk(a3.prev_clock);
    o[21] = O::mk(a4.prev_value); o[22] = O::mk(a5.prev_value);
    o[23] = O::mk(a4.value & 0xFF); o[24] = O::mk((a4.value >> 8) & 0xFF);
    o[25] = O::mk(a5.value & 0xFF); o[26] = O::mk((a5.value >> 8) & 0xFF);
    o[27] = O::mk(a4.prev_clock); o[28] = O::mk(a5.prev_clock);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F two8 = e.c(TWO8), one = e.c(1);
    AIR_EVAL5(e);
    F opc = e.next(), src0_off = e.next(), src1_off = e.next(), dst_off = e.next();
    F op0_0 = e.next(), op0_1 = e.next(), op0_2 = e.next(), op0_3 = e.next(), op0_pc_lo = e.next(), op0_pc_hi = e.next();
    F op1_0 = e.next(), op1_1 = e.next(), op1_2 = e.next(), op1_3 = e.next(), op1_pc_lo = e.next(), op1_pc_hi = e.next();
    F dpv_lo = e.next(), dpv_hi = e.next(), dst_0 = e.next(), dst_1 = e.next(), dst_2 = e.next(), dst_3 = e.next();
    F dpc_lo = e.next(), dpc_hi = e.next();
    e.constraint(enabler * (enabler - one));
    F bitwise_op = opc - e.c(OP_U32_STORE_AND_FP_FP);
    F op0_lo = op0_0 + op0_1 * two8, op0_hi = op0_2 + op0_3 * two8;
    F op1_lo = op1_0 + op1_1 * two8, op1_hi = op1_2 + op1_3 * two8;
    F dst_lo = dst_0 + dst_1 * two8, dst_hi = dst_2 + dst_3 * two8;
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, src1_off, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src0_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src0_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + src1_off, op1_pc_lo, op1_lo);
    e.rel(REL_MEMORY, enabler, fp + src1_off, clock, op1_lo);
    e.rel(REL_MEMORY, -enabler, fp + src1_off + one, op1_pc_hi, op1_hi);
    e.rel(REL_MEMORY, enabler, fp + src1_off + one, clock, op1_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, dst_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, dst_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_0, op1_0, dst_0);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_1, op1_1, dst_1);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_2, op1_2, dst_2);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_3, op1_3, dst_3);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op1_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// u32_store_bitwise_fp_imm.rs (39 And, 40 Or, 41 Xor) — 26 columns.  witness :180-360, eval :520-720
struct U32StoreBitwiseFpImm {
  static constexpr int N_TRACE = 26;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    AIR_COMMON5(o, b, enabler);
    o[5] = O::mk(b.inst[0] == OP_RET ? OP_U32_STORE_AND_FP_IMM : b.inst[0]);
    o[6] = O::mk(b.inst[1]);
    o[7] = O::mk(b.inst[2] & 0xFF); o[8] = O::mk((b.inst[2] >> 8) & 0xFF);
    o[9] = O::mk(b.inst[3] & 0xFF); o[10] = O::mk((b.inst[3] >> 8) & 0xFF);
    o[11] = O::mk(b.inst[4]);
    o[12] = O::mk(a0.value & 0xFF); o[13] = O::mk((a0.value >> 8) & 0xFF);
    o[14] = O::mk(a1.value & 0xFF); o[15] = O::mk((a1.value >> 8) & 0xFF);
    o[16] = O::mk(a0.prev_clock); o[17] = O::mk(a1.prev_clock);
    o[18] = O::mk(a2.prev_value); o[19] = O::mk(a3.prev_value);
    o[20] = O::mk(a2.value & 0xFF); o[21] = O::mk((a2.value >> 8) & 0xFF);
    o[22] = O::mk(a3.value & 0xFF); o[23] = O::mk((a3.value >> 8) & 0xFF);
    o[24] = O::mk(a2.prev_clock); o[25] = O::mk(a3.prev_clock);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F two8 = e.c(TWO8), one = e.c(1);
    AIR_EVAL5(e);
    F opc = e.next(), src_off = e.next(), imm_0 = e.next(), imm_1 = e.next(), imm_2 = e.next(), imm_3 = e.next();
    F dst_off = e.next(), op0_0 = e.next(), op0_1 = e.next(), op0_2 = e.next(), op0_3 = e.next();
    F op0_pc_lo = e.next(), op0_pc_hi = e.next(), dpv_lo = e.next(), dpv_hi = e.next();
    F dst_0 = e.next(), dst_1 = e.next(), dst_2 = e.next(), dst_3 = e.next(), dpc_lo = e.next(), dpc_hi = e.next();
    e.constraint(enabler * (enabler - one));
    F bitwise_op = opc - e.c(OP_U32_STORE_AND_FP_IMM);
    F op0_lo = op0_0 + op0_1 * two8, op0_hi = op0_2 + op0_3 * two8;
    F imm_lo = imm_0 + imm_1 * two8, imm_hi = imm_2 + imm_3 * two8;
    F dst_lo = dst_0 + dst_1 * two8, dst_hi = dst_2 + dst_3 * two8;
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src_off, imm_lo, imm_hi);
    e.rel(REL_MEMORY, -enabler, pc + one, inst_prev_clock, dst_off);
    e.rel(REL_MEMORY, enabler, pc + one, clock, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src_off, op0_pc_lo, op0_lo);
    e.rel(REL_MEMORY, enabler, fp + src_off, clock, op0_lo);
    e.rel(REL_MEMORY, -enabler, fp + src_off + one, op0_pc_hi, op0_hi);
    e.rel(REL_MEMORY, enabler, fp + src_off + one, clock, op0_hi);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dpc_lo, dpv_lo);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, dst_lo);
    e.rel(REL_MEMORY, -enabler, fp + dst_off + one, dpc_hi, dpv_hi);
    e.rel(REL_MEMORY, enabler, fp + dst_off + one, clock, dst_hi);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_0, imm_0, dst_0);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_1, imm_1, dst_1);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_2, imm_2, dst_2);
    e.rel(REL_BITWISE, m1, bitwise_op, op0_3, imm_3, dst_3);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_lo - enabler);
    e.rel(REL_RC20, m1, clock - op0_pc_hi - enabler);
    e.rel(REL_RC20, m1, clock - dpc_lo - enabler);
    e.rel(REL_RC20, m1, clock - dpc_hi - enabler);
    e.finalize_pairs();
  }
};

#undef AIR_COMMON5
#undef AIR_EVAL5

}  // namespace air
