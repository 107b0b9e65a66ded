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

}  // namespace air
