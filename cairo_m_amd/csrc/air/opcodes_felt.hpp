// AIR of the felt (M31) opcode components.  Each component restates, from
// /root/reference/crates/prover/src/components/opcodes/<name>.rs:
//   eval<E>     <- `impl FrameworkEval for Eval { fn evaluate }`  (column order = next_trace_mask order,
//                  constraint order = add_constraint order, relation order = add_to_relation order)
//   witness<O>  <- the per-row closure of `Claim::write_trace` (trace cells only; lookup tuples are
//                  re-derived from eval's relation entries, see DESIGN.md "LogUp from the AIR").
#pragma once
#include "air_common.hpp"

namespace air {

// ------------------------------------------------------------------------------------------------
// store_fp_imm.rs  (opcodes 4 StoreAddFpImm, 6 StoreMulFpImm) — 18 columns
// witness: store_fp_imm.rs:147-296, eval: :457-616
struct StoreFpImm {
  static constexpr int N_TRACE = 18;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    uint32_t op = b.inst[0];
    M imm = O::mk(op == OP_RET ? 0u : b.inst[2]);
    M imm_inv = imm.v != 0 ? O::inv(imm) : O::mk(0);
    uint32_t flag = op >= OP_STORE_ADD_FP_IMM ? op - OP_STORE_ADD_FP_IMM : 0;  // saturating_sub
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1);
    M en = O::mk(enabler);
    M src_val = O::mk(a0.value), imm_col = O::mk(b.inst[2]);
    o[0] = en; o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = imm_col; o[7] = O::mk(b.inst[3]);
    o[8] = O::mk(a0.prev_clock); o[9] = src_val; o[10] = imm_inv;
    o[11] = O::mk(a1.prev_clock); o[12] = O::mk(a1.prev_value); o[13] = O::mk(a1.value);
    o[14] = O::mk(flag / 2) * en; o[15] = O::mk(flag % 2) * en;
    o[16] = src_val * imm_col; o[17] = src_val * imm_inv;
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F src_off = e.next(), imm = e.next(), dst_off = e.next(), src_prev_clock = e.next(), src_val = e.next();
    F imm_inv = e.next(), dst_prev_clock = e.next(), dst_prev_val = e.next(), dst_val = e.next();
    F flag0 = e.next(), flag1 = e.next(), prod = e.next(), div = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(flag0 * (one - flag0));
    e.constraint(flag1 * (one - flag1));
    e.constraint(prod - src_val * imm);
    e.constraint(imm * (imm_inv * imm - one));
    e.constraint(imm_inv * (imm_inv * imm - one));
    e.constraint(div - src_val * imm_inv);
    F is_add = (one - flag0) * (one - flag1), is_sub = (one - flag0) * flag1;
    F is_mul = flag0 * (one - flag1), is_div = flag0 * flag1;
    F opcode_id = e.c(OP_STORE_ADD_FP_IMM) + e.c(2) * flag0 + flag1;
    F res = is_add * (src_val + imm) + is_sub * (src_val - imm) + is_mul * prod + is_div * div;
    e.constraint(dst_val - res);
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opcode_id, src_off, imm, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opcode_id, src_off, imm, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src_off, src_prev_clock, src_val);
    e.rel(REL_MEMORY, enabler, fp + src_off, clock, src_val);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, dst_val);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - src_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// store_fp_fp.rs  (opcodes 0..3 StoreAdd/Sub/Mul/DivFpFp) — 20 columns
// witness: store_fp_fp.rs:153-318, eval: :496-681
struct StoreFpFp {
  static constexpr int N_TRACE = 20;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    uint32_t op = b.inst[0];
    uint32_t flag = op == OP_RET ? 0 : op - OP_STORE_ADD_FP_FP;
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2);
    M op0 = O::mk(a0.value), op1 = O::mk(a1.value);
    M op1_inv = op1.v != 0 ? O::inv(op1) : O::mk(0);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(b.inst[3]);
    o[8] = O::mk(a0.prev_clock); o[9] = op0; o[10] = O::mk(a1.prev_clock); o[11] = op1; o[12] = op1_inv;
    o[13] = O::mk(a2.prev_clock); o[14] = O::mk(a2.prev_value); o[15] = O::mk(a2.value);
    o[16] = O::mk(flag / 2); o[17] = O::mk(flag % 2);
    o[18] = op0 * op1; o[19] = op0 * op1_inv;
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F off0 = e.next(), off1 = e.next(), off2 = e.next();
    F op0_prev_clock = e.next(), op0_val = e.next(), op1_prev_clock = e.next(), op1_val = e.next(), op1_inv = e.next();
    F dst_prev_clock = e.next(), dst_prev_val = e.next(), dst_val = e.next();
    F flag0 = e.next(), flag1 = e.next(), prod = e.next(), div = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(flag0 * (one - flag0));
    e.constraint(flag1 * (one - flag1));
    e.constraint(prod - op0_val * op1_val);
    e.constraint(op1_val * (op1_inv * op1_val - one));
    e.constraint(op1_inv * (op1_inv * op1_val - one));
    e.constraint(div - op0_val * op1_inv);
    F is_add = (one - flag0) * (one - flag1), is_sub = (one - flag0) * flag1;
    F is_mul = flag0 * (one - flag1), is_div = flag0 * flag1;
    F opcode_id = e.c(OP_STORE_ADD_FP_FP) + e.c(2) * flag0 + flag1;
    F res = is_add * (op0_val + op1_val) + is_sub * (op0_val - op1_val) + is_mul * prod + is_div * div;
    e.constraint(dst_val - res);
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opcode_id, off0, off1, off2);
    e.rel(REL_MEMORY, enabler, pc, clock, opcode_id, off0, off1, off2);
    e.rel(REL_MEMORY, -enabler, fp + off0, op0_prev_clock, op0_val);
    e.rel(REL_MEMORY, enabler, fp + off0, clock, op0_val);
    e.rel(REL_MEMORY, -enabler, fp + off1, op1_prev_clock, op1_val);
    e.rel(REL_MEMORY, enabler, fp + off1, clock, op1_val);
    e.rel(REL_MEMORY, -enabler, fp + off2, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + off2, clock, dst_val);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op1_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// jnz_fp_imm.rs (opcode 14) — 12 columns.  witness: jnz_fp_imm.rs:121-238, eval: :330-446
struct JnzFpImm {
  static constexpr int N_TRACE = 12;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    Access a0 = access_at(b, acc, 0);
    M op0 = O::mk(a0.value), one = O::mk(1), imm = O::mk(b.inst[2]), pc = O::mk(b.pc);
    M op0_inv = op0.v == 0 ? O::mk(0) : O::inv(op0);
    M taken = O::mk(op0.v == 0 ? 0u : 1u);
    o[0] = O::mk(enabler); o[1] = pc; o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = imm; o[7] = O::mk(a0.prev_clock); o[8] = op0; o[9] = op0_inv; o[10] = taken;
    o[11] = pc + one + taken * (imm - one);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_JNZ_FP_IMM);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F off0 = e.next(), imm = e.next(), op0_prev_clock = e.next(), op0_val = e.next(), op0_val_inv = e.next();
    F taken = e.next(), pc_new = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(enabler * op0_val * (taken - one));
    e.constraint(enabler * (taken - op0_val * op0_val_inv));
    e.constraint(enabler * (pc_new - pc - one - taken * (imm - one)));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc_new, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, off0, imm);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, off0, imm);
    e.rel(REL_MEMORY, -enabler, fp + off0, op0_prev_clock, op0_val);
    e.rel(REL_MEMORY, enabler, fp + off0, clock, op0_val);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// jmp_imm.rs (opcodes 12 JmpAbsImm, 13 JmpRelImm) — 7 columns.  witness: jmp_imm.rs:109-190, eval: :268-345
struct JmpImm {
  static constexpr int N_TRACE = 7;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access*, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    M en = O::mk(enabler);
    o[0] = en; o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]);
    o[6] = en * (O::mk(b.inst[0]) - O::mk(OP_JMP_ABS_IMM));
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F off0 = e.next(), is_rel = e.next();
    F opcode_id = e.c(OP_JMP_ABS_IMM) + is_rel;
    e.constraint(enabler * (one - enabler));
    e.constraint(is_rel * (one - is_rel));
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opcode_id, off0);
    e.rel(REL_MEMORY, enabler, pc, clock, opcode_id, off0);
    e.rel(REL_RC20, e.c(M31_P - 1), clock - inst_prev_clock - enabler);
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, off0 + pc * is_rel, fp, clock + one);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// store_imm.rs (opcode 9) — 9 columns.  witness: store_imm.rs:113-195, eval: :302-419
struct StoreImm {
  static constexpr int N_TRACE = 9;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(a0.prev_clock); o[8] = O::mk(a0.prev_value);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), zero = e.c(0), opc = e.c(OP_STORE_IMM);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F off0 = e.next(), off2 = e.next(), dst_prev_clock = e.next(), dst_prev_val = e.next();
    e.constraint(enabler * (one - enabler));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, off0, off2);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, off0, off2);
    e.rel(REL_MEMORY, -enabler, fp + off2, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + off2, clock, off0, zero, zero, zero);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// ret.rs (opcode 11) — 9 columns.  witness: ret.rs:118-215, eval: :356-486
struct Ret {
  static constexpr int N_TRACE = 9;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1);  // [fp-1], [fp-2]
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(a1.prev_clock); o[6] = O::mk(a1.value); o[7] = O::mk(a0.prev_clock); o[8] = O::mk(a0.value);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two = e.c(2), opc = e.c(OP_RET);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F fp_min_2_prev_clock = e.next(), fp_min_2_val = e.next(), fp_min_1_prev_clock = e.next(), fp_min_1_val = e.next();
    e.constraint(enabler * (one - enabler));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, fp_min_1_val, fp_min_2_val, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc);
    e.rel(REL_MEMORY, enabler, pc, clock, opc);
    e.rel(REL_MEMORY, -enabler, fp - two, fp_min_2_prev_clock, fp_min_2_val);
    e.rel(REL_MEMORY, enabler, fp - two, clock, fp_min_2_val);
    e.rel(REL_MEMORY, -enabler, fp - enabler, fp_min_1_prev_clock, fp_min_1_val);
    e.rel(REL_MEMORY, enabler, fp - enabler, clock, fp_min_1_val);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - fp_min_2_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - fp_min_1_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// call_abs_imm.rs (opcode 10) — 11 columns.  witness: call_abs_imm.rs:121-222, eval: :364-494
struct CallAbsImm {
  static constexpr int N_TRACE = 11;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]);
    o[7] = O::mk(a0.prev_clock); o[8] = O::mk(a0.prev_value); o[9] = O::mk(a1.prev_clock); o[10] = O::mk(a1.prev_value);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_CALL_ABS_IMM);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F off0 = e.next(), off1 = e.next(), op0_prev_clock = e.next(), op0_prev_val = e.next();
    F op0p1_prev_clock = e.next(), op0p1_prev_val = e.next();
    e.constraint(enabler * (one - enabler));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, off1, fp + off0 + one + one, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, off0, off1);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, off0, off1);
    e.rel(REL_MEMORY, -enabler, fp + off0, op0_prev_clock, op0_prev_val);
    e.rel(REL_MEMORY, enabler, fp + off0, clock, fp);
    e.rel(REL_MEMORY, -enabler, fp + off0 + one, op0p1_prev_clock, op0p1_prev_val);
    e.rel(REL_MEMORY, enabler, fp + off0 + one, clock, pc + one);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0p1_prev_clock - enabler);
    e.finalize_pairs();
  }
};


// ------------------------------------------------------------------------------------------------
// assert_eq_fp_imm.rs (opcode 50) — 9 columns.  witness: assert_eq_fp_imm.rs:105-190, eval: :300-417
struct AssertEqFpImm {
  static constexpr int N_TRACE = 9;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(a0.prev_clock); o[8] = O::mk(a0.value);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_ASSERT_EQ_FP_IMM);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F src0_off = e.next(), imm = e.next(), op0_prev_clock = e.next(), op0_val = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(op0_val - imm);
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src0_off, imm);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src0_off, imm);
    e.rel(REL_MEMORY, -enabler, fp + src0_off, op0_prev_clock, op0_val);
    e.rel(REL_MEMORY, enabler, fp + src0_off, clock, op0_val);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - op0_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// store_frame_pointer.rs (opcode 43) — 9 columns.  witness: store_frame_pointer.rs:107-190, eval: :300-417
struct StoreFramePointer {
  static constexpr int N_TRACE = 9;
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(b.inst[2]); o[7] = O::mk(a0.prev_value); o[8] = O::mk(a0.prev_clock);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_STORE_FRAME_POINTER);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F imm = e.next(), dst_off = e.next(), dst_prev_val = e.next(), dst_prev_clock = e.next();
    e.constraint(enabler * (one - enabler));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, imm, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, imm, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, fp + imm);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// double_deref_fp_imm.rs (opcodes 8 StoreDoubleDerefFp, 44 StoreToDoubleDerefFpImm) — 17 columns
// witness: double_deref_fp_imm.rs:139-250, eval: :378-509
struct DoubleDerefFpImm {
  static constexpr int N_TRACE = 17;
  static constexpr uint32_t DELTA_INV = m31_inv_const(OP_STORE_TO_DOUBLE_DEREF_FP_IMM - OP_STORE_DOUBLE_DEREF_FP);
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    M opc = O::mk(b.inst[0] == OP_RET ? OP_STORE_DOUBLE_DEREF_FP : b.inst[0]);
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2);
    M one = O::mk(1), fp = O::mk(b.fp), off1 = O::mk(b.inst[2]), off2 = O::mk(b.inst[3]), val0 = O::mk(a0.value);
    M write_lhs = (opc - O::mk(OP_STORE_DOUBLE_DEREF_FP)) * O::mk(DELTA_INV);
    M addr1 = write_lhs * (fp + off2) + (one - write_lhs) * (val0 + off1);
    M addr2 = write_lhs * (val0 + off1) + (one - write_lhs) * (fp + off2);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = fp; o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = opc; o[6] = O::mk(b.inst[1]); o[7] = off1; o[8] = off2; o[9] = val0; o[10] = O::mk(a0.prev_clock);
    o[11] = addr1; o[12] = O::mk(a1.value); o[13] = O::mk(a1.prev_clock); o[14] = addr2;
    o[15] = O::mk(a2.prev_value); o[16] = O::mk(a2.prev_clock);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F opc = e.next(), off0 = e.next(), off1 = e.next(), off2 = e.next(), val0 = e.next(), prev_clock0 = e.next();
    F addr1 = e.next(), val1 = e.next(), prev_clock1 = e.next(), addr2 = e.next(), prev_val2 = e.next(), prev_clock2 = e.next();
    F write_lhs = (opc - e.c(OP_STORE_DOUBLE_DEREF_FP)) * e.c(DELTA_INV);
    e.constraint(enabler * (one - enabler));
    e.constraint(write_lhs * (one - write_lhs));
    e.constraint(enabler * (addr1 - write_lhs * (fp + off2) - (one - write_lhs) * (val0 + off1)));
    e.constraint(enabler * (addr2 - write_lhs * (val0 + off1) - (one - write_lhs) * (fp + off2)));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, off0, off1, off2);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, off0, off1, off2);
    e.rel(REL_MEMORY, -enabler, fp + off0, prev_clock0, val0);
    e.rel(REL_MEMORY, enabler, fp + off0, clock, val0);
    e.rel(REL_MEMORY, -enabler, addr1, prev_clock1, val1);
    e.rel(REL_MEMORY, enabler, addr1, clock, val1);
    e.rel(REL_MEMORY, -enabler, addr2, prev_clock2, prev_val2);
    e.rel(REL_MEMORY, enabler, addr2, clock, val1);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock0 - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock1 - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock2 - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// double_deref_fp_fp.rs (opcodes 42 StoreDoubleDerefFpFp, 45 StoreToDoubleDerefFpFp) — 19 columns
// witness: double_deref_fp_fp.rs:145-265, eval: :410-562
struct DoubleDerefFpFp {
  static constexpr int N_TRACE = 19;
  static constexpr uint32_t DELTA_INV = m31_inv_const(OP_STORE_TO_DOUBLE_DEREF_FP_FP - OP_STORE_DOUBLE_DEREF_FP_FP);
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    using M = typename O::M;
    M opc = O::mk(b.inst[0] == OP_RET ? OP_STORE_DOUBLE_DEREF_FP_FP : b.inst[0]);
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1), a2 = access_at(b, acc, 2), a3 = access_at(b, acc, 3);
    M one = O::mk(1), fp = O::mk(b.fp), off2 = O::mk(b.inst[3]), val0 = O::mk(a0.value), val1 = O::mk(a1.value);
    M write_lhs = (opc - O::mk(OP_STORE_DOUBLE_DEREF_FP_FP)) * O::mk(DELTA_INV);
    M addr2 = write_lhs * (fp + off2) + (one - write_lhs) * (val0 + val1);
    M addr3 = write_lhs * (val0 + val1) + (one - write_lhs) * (fp + off2);
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = fp; o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = opc; o[6] = O::mk(b.inst[1]); o[7] = O::mk(b.inst[2]); o[8] = off2;
    o[9] = val0; o[10] = O::mk(a0.prev_clock); o[11] = val1; o[12] = O::mk(a1.prev_clock);
    o[13] = addr2; o[14] = O::mk(a2.value); o[15] = O::mk(a2.prev_clock);
    o[16] = addr3; o[17] = O::mk(a3.prev_value); o[18] = O::mk(a3.prev_clock);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F opc = e.next(), off0 = e.next(), off1 = e.next(), off2 = e.next();
    F val0 = e.next(), prev_clock0 = e.next(), val1 = e.next(), prev_clock1 = e.next();
    F addr2 = e.next(), val2 = e.next(), prev_clock2 = e.next(), addr3 = e.next(), prev_val3 = e.next(), prev_clock3 = e.next();
    F write_lhs = (opc - e.c(OP_STORE_DOUBLE_DEREF_FP_FP)) * e.c(DELTA_INV);
    e.constraint(enabler * (one - enabler));
    e.constraint(write_lhs * (one - write_lhs));
    e.constraint(enabler * (addr2 - write_lhs * (fp + off2) - (one - write_lhs) * (val0 + val1)));
    e.constraint(enabler * (addr3 - write_lhs * (val0 + val1) - (one - write_lhs) * (fp + off2)));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, off0, off1, off2);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, off0, off1, off2);
    e.rel(REL_MEMORY, -enabler, fp + off0, prev_clock0, val0);
    e.rel(REL_MEMORY, enabler, fp + off0, clock, val0);
    e.rel(REL_MEMORY, -enabler, fp + off1, prev_clock1, val1);
    e.rel(REL_MEMORY, enabler, fp + off1, clock, val1);
    e.rel(REL_MEMORY, -enabler, addr2, prev_clock2, val2);
    e.rel(REL_MEMORY, enabler, addr2, clock, val2);
    e.rel(REL_MEMORY, -enabler, addr3, prev_clock3, prev_val3);
    e.rel(REL_MEMORY, enabler, addr3, clock, val2);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock0 - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock1 - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock2 - enabler);
    e.rel(REL_RC20, m1, clock - prev_clock3 - enabler);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// store_le_fp_imm.rs (opcode 48) — 22 columns.  witness: store_le_fp_imm.rs:225-400, eval: :520-747
struct StoreLeFpImm {
  static constexpr int N_TRACE = 22;
  static constexpr uint32_t PRIME_OVER_3_HIGH = ((M31_P / 3) >> 16) + 1;  // store_le_fp_imm.rs:132
  static constexpr uint32_t PRIME_OVER_2_HIGH = ((M31_P / 2) >> 16) + 1;  // :133
  template <class O>
  static AIR_HD void witness(const Bundle& b, const Access* acc, uint32_t enabler, typename O::M* o) {
    Access a0 = access_at(b, acc, 0), a1 = access_at(b, acc, 1);
    uint32_t src = a0.value, imm = b.inst[2];
    uint32_t is_le = src <= imm ? 1u : 0u;
    uint32_t a = is_le ? src : imm, bb = is_le ? imm : src;
    // three arcs, stable sort by length (store_le_fp_imm.rs:303-315)
    uint32_t len[3] = {a, bb >= a ? bb - a : 0u, M31_P - 1 - bb};
    uint32_t idx[3] = {0, 1, 2};
    for (int i = 1; i < 3; i++)
      for (int j = i; j > 0 && len[j - 1] > len[j]; j--) {
        uint32_t t = len[j]; len[j] = len[j - 1]; len[j - 1] = t;
        t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t;
      }
    uint32_t exclude = idx[2];
    uint32_t k01 = 0, k02 = 0, k12 = 0;
    if (enabler == 1) { k01 = exclude == 2; k02 = exclude == 1; k12 = exclude == 0; }
    o[0] = O::mk(enabler); o[1] = O::mk(b.pc); o[2] = O::mk(b.fp); o[3] = O::mk(b.clock); o[4] = O::mk(b.inst_prev_clock);
    o[5] = O::mk(b.inst[1]); o[6] = O::mk(imm); o[7] = O::mk(b.inst[3]);
    o[8] = O::mk(src); o[9] = O::mk(a0.prev_clock); o[10] = O::mk(a1.prev_value); o[11] = O::mk(a1.prev_clock);
    o[12] = O::mk(a); o[13] = O::mk(bb); o[14] = O::mk(k01); o[15] = O::mk(k02); o[16] = O::mk(k12);
    o[17] = O::mk(len[0] % PRIME_OVER_3_HIGH); o[18] = O::mk(len[0] / PRIME_OVER_3_HIGH);
    o[19] = O::mk(len[1] % PRIME_OVER_2_HIGH); o[20] = O::mk(len[1] / PRIME_OVER_2_HIGH);
    o[21] = O::mk(is_le);
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), opc = e.c(OP_STORE_LE_FP_IMM), p3 = e.c(PRIME_OVER_3_HIGH), p2 = e.c(PRIME_OVER_2_HIGH);
    F enabler = e.next(), pc = e.next(), fp = e.next(), clock = e.next(), inst_prev_clock = e.next();
    F src_off = e.next(), imm = e.next(), dst_off = e.next(), src_val = e.next(), src_prev_clock = e.next();
    F dst_prev_val = e.next(), dst_prev_clock = e.next(), a = e.next(), b = e.next();
    F keep_0_1 = e.next(), keep_0_2 = e.next(), keep_1_2 = e.next();
    F arc_short_lo = e.next(), arc_short_hi = e.next(), arc_long_lo = e.next(), arc_long_hi = e.next(), is_le = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(keep_0_1 * (one - keep_0_1));
    e.constraint(keep_0_2 * (one - keep_0_2));
    e.constraint(keep_1_2 * (one - keep_1_2));
    e.constraint(enabler * (keep_0_1 + keep_0_2 + keep_1_2 - one));
    e.constraint(is_le * (one - is_le));
    F arc_short = arc_short_lo + arc_short_hi * p3;
    F arc_long = arc_long_lo + arc_long_hi * p2;
    F arc_sum = arc_short + arc_long;
    F arc_prod = arc_short * arc_long;
    e.constraint(keep_0_1 * (arc_sum - (a + b - a)));
    e.constraint(keep_0_1 * (arc_prod - a * (b - a)));
    e.constraint(keep_0_2 * (arc_sum - (a - one - b)));
    e.constraint(keep_0_2 * (arc_prod - a * (-one - b)));
    e.constraint(keep_1_2 * (arc_sum - (b - a - one - b)));
    e.constraint(keep_1_2 * (arc_prod - (b - a) * (-one - b)));
    e.constraint(enabler * (a - is_le * src_val - (one - is_le) * imm));
    e.constraint(enabler * (b - is_le * imm - (one - is_le) * src_val));
    e.rel(REL_REGISTERS, -enabler, pc, fp, clock);
    e.rel(REL_REGISTERS, enabler, pc + one, fp, clock + one);
    e.rel(REL_MEMORY, -enabler, pc, inst_prev_clock, opc, src_off, imm, dst_off);
    e.rel(REL_MEMORY, enabler, pc, clock, opc, src_off, imm, dst_off);
    e.rel(REL_MEMORY, -enabler, fp + src_off, src_prev_clock, src_val);
    e.rel(REL_MEMORY, enabler, fp + src_off, clock, src_val);
    e.rel(REL_MEMORY, -enabler, fp + dst_off, dst_prev_clock, dst_prev_val);
    e.rel(REL_MEMORY, enabler, fp + dst_off, clock, is_le);
    F m1 = e.c(M31_P - 1);
    e.rel(REL_RC16, m1, arc_short_lo);
    e.rel(REL_RC16, m1, arc_short_hi);
    e.rel(REL_RC16, m1, arc_long_lo);
    e.rel(REL_RC16, m1, arc_long_hi);
    e.rel(REL_RC20, m1, clock - inst_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - src_prev_clock - enabler);
    e.rel(REL_RC20, m1, clock - dst_prev_clock - enabler);
    e.finalize_pairs();
  }
};

}  // namespace air
