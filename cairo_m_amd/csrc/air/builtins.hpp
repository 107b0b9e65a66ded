// AIR of the non-opcode components: memory, merkle, clock_update, poseidon2, range_check_{8,16,20},
// bitwise.  Restated from /root/reference/crates/prover/src/components/{memory,merkle,clock_update,
// poseidon2}.rs and crates/prover/src/preprocessed/{range_check/range_check_macro.rs,bitwise.rs}.
#pragma once
#include "air_common.hpp"
#include "poseidon2_consts.hpp"

namespace air {

// ---- input rows (C-ABI layouts) -----------------------------------------------------------------
struct MemoryCell { uint32_t address, value[4], clock, multiplicity; };
struct ClockUpdateRow { uint32_t address, prev_clock, value[4]; };
struct MerkleNode { uint32_t index, depth, left_value, right_value, parent_value, left_mult, right_mult, parent_mult; };

// ------------------------------------------------------------------------------------------------
// memory.rs — 9 columns.  witness: memory.rs:93-195, eval: memory.rs:289-369
struct MemoryC {
  static constexpr int N_TRACE = 9;
  // `cell` is null on padding rows
  template <class O>
  static AIR_HD void witness(const MemoryCell* cell, uint32_t root, uint32_t enabler, typename O::M* o) {
    o[0] = O::mk(enabler);
    if (cell) {
      o[1] = O::mk(cell->address); o[2] = O::mk(cell->clock);
      o[3] = O::mk(cell->value[0]); o[4] = O::mk(cell->value[1]); o[5] = O::mk(cell->value[2]); o[6] = O::mk(cell->value[3]);
      o[7] = O::mk(cell->multiplicity); o[8] = O::mk(root);
    } else {
      for (int i = 1; i < 9; i++) o[i] = O::mk(0);
    }
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), m31_2 = e.c(2), m31_3 = e.c(3), m31_4 = e.c(4), tree_height = e.c(TREE_HEIGHT);
    F enabler = e.next(), address = e.next(), clock = e.next();
    F value0 = e.next(), value1 = e.next(), value2 = e.next(), value3 = e.next();
    F multiplicity = e.next(), root = e.next();
    e.constraint(enabler * (one - enabler));
    e.rel(REL_MEMORY, multiplicity, address, clock, value0, value1, value2, value3);
    e.rel(REL_MERKLE, -enabler, address * m31_4, tree_height, value0, root);
    e.rel(REL_MERKLE, -enabler, address * m31_4 + one, tree_height, value1, root);
    e.rel(REL_MERKLE, -enabler, address * m31_4 + m31_2, tree_height, value2, root);
    e.rel(REL_MERKLE, -enabler, address * m31_4 + m31_3, tree_height, value3, root);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// merkle.rs — 10 columns.  witness: merkle.rs:92-201, eval: merkle.rs:296-379
struct MerkleC {
  static constexpr int N_TRACE = 10;
  template <class O>
  static AIR_HD void witness(const MerkleNode* n, uint32_t root, uint32_t enabler, typename O::M* o) {
    o[0] = O::mk(enabler);
    if (n) {
      o[1] = O::mk(n->index); o[2] = O::mk(n->depth); o[3] = O::mk(n->left_value); o[4] = O::mk(n->right_value);
      o[5] = O::mk(n->parent_value); o[6] = O::mk(n->left_mult); o[7] = O::mk(n->right_mult); o[8] = O::mk(n->parent_mult);
      o[9] = O::mk(root);
    } else {
      for (int i = 1; i < 10; i++) o[i] = O::mk(0);
    }
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F one = e.c(1), two = e.c(2), two_inv = e.c(1u << 30);  // 2^-1 = 2^30 mod (2^31-1)
    F enabler = e.next(), index = e.next(), depth = e.next(), left_value = e.next(), right_value = e.next();
    F parent_value = e.next(), lm = e.next(), rm = e.next(), pm = e.next(), root = e.next();
    e.constraint(enabler * (one - enabler));
    e.constraint(lm * (lm - one) * (lm - one * two));
    e.constraint(rm * (rm - one) * (rm - one * two));
    e.constraint(pm * (pm - one) * (pm - one * two));
    e.rel(REL_MERKLE, lm, index, depth, left_value, root);
    e.rel(REL_MERKLE, rm, index + one, depth, right_value, root);
    e.rel(REL_MERKLE, -pm, index * two_inv, depth - one, parent_value, root);
    e.rel(REL_POSEIDON2, enabler, left_value, right_value);
    e.rel(REL_POSEIDON2, -enabler, parent_value);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// clock_update.rs — 7 columns.  witness: clock_update.rs:77-166, eval: clock_update.rs:213-266
struct ClockUpdateC {
  static constexpr int N_TRACE = 7;
  template <class O>
  static AIR_HD void witness(const ClockUpdateRow* r, uint32_t enabler, typename O::M* o) {
    o[0] = O::mk(enabler);
    if (r) {
      o[1] = O::mk(r->address); o[2] = O::mk(r->prev_clock);
      o[3] = O::mk(r->value[0]); o[4] = O::mk(r->value[1]); o[5] = O::mk(r->value[2]); o[6] = O::mk(r->value[3]);
    } else {
      for (int i = 1; i < 7; i++) o[i] = O::mk(0);
    }
  }
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F enabler = e.next(), address = e.next(), prev_clk = e.next();
    F value0 = e.next(), value1 = e.next(), value2 = e.next(), value3 = e.next();
    F one = e.c(1);
    e.constraint(enabler * (one - enabler));
    e.rel(REL_MEMORY, -enabler, address, prev_clk, value0, value1, value2, value3);
    e.rel(REL_MEMORY, enabler, address, prev_clk + e.c(RC20_LIMIT), value0, value1, value2, value3);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// poseidon2.rs — 443 columns = 1 + 16*(1 + 8*3) + 3*14.
// helpers: apply_m4 :94-109, apply_external_round_matrix :113-138, apply_internal_round_matrix :142-153
constexpr int P2_T = 16, P2_FULL = 8, P2_PARTIAL = 14;

AIR_HD uint32_t p2_ext_rc(int r, int i) { return P2_EXTERNAL_RC[r][i]; }
AIR_HD uint32_t p2_int_rc(int r) { return P2_INTERNAL_RC[r]; }
AIR_HD uint32_t p2_diag(int i) { return P2_INTERNAL_DIAG[i]; }

template <class F>
AIR_HD void p2_apply_m4(F& x0, F& x1, F& x2, F& x3) {
  F t0 = x0 + x1;
  F t02 = t0 + t0;
  F t1 = x2 + x3;
  F t12 = t1 + t1;
  F t2 = x1 + x1 + t1;
  F t3 = x3 + x3 + t0;
  F t4 = t12 + t12 + t3;
  F t5 = t02 + t02 + t2;
  F t6 = t3 + t5;
  F t7 = t2 + t4;
  x0 = t6; x1 = t5; x2 = t7; x3 = t4;
}
template <class F>
AIR_HD void p2_external_matrix(F* s) {
  for (int i = 0; i < 4; i++) p2_apply_m4(s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
  for (int j = 0; j < 4; j++) {
    F t = s[j] + s[j + 4] + s[j + 8] + s[j + 12];
    for (int i = 0; i < 4; i++) s[4 * i + j] = s[4 * i + j] + t;
  }
}
// mk(u32) -> F
template <class F, class MK>
AIR_HD void p2_internal_matrix(F* s, MK mk) {
  F sum = s[0];
  for (int i = 1; i < P2_T; i++) sum = sum + s[i];
  for (int i = 0; i < P2_T; i++) s[i] = s[i] * mk(p2_diag(i)) + sum;
}

struct Poseidon2C {
  static constexpr int N_TRACE = 1 + P2_T * (1 + P2_FULL * 3) + 3 * P2_PARTIAL;  // 443
  // witness: poseidon2.rs:172-325.  `in` = 16-word initial state (zeros on padding rows)
  // `o` is anything indexable with `o[c] = M` (a plain array, or a writer that stores straight into the trace
  // columns: 443 cells per row do not fit in registers, and a local array means 1.7 KiB of scratch per thread)
  template <class O, class Out>
  static AIR_HD void witness(const uint32_t* in, uint32_t enabler, Out o) {
    using M = typename O::M;
    auto mk = [](uint32_t v) { return O::mk(v); };
    int c = 0;
    o[c++] = O::mk(enabler);
    M s[P2_T];
    for (int i = 0; i < P2_T; i++) { s[i] = O::mk(in ? in[i] : 0u); o[c++] = s[i]; }
    p2_external_matrix(s);
    for (int half = 0; half < 2; half++) {
      if (half == 1) {
        for (int r = 0; r < P2_PARTIAL; r++) {
          s[0] = s[0] + O::mk(p2_int_rc(r));
          M init = s[0];
          s[0] = s[0] * s[0]; o[c++] = s[0];
          s[0] = s[0] * s[0]; o[c++] = s[0];
          s[0] = init * s[0]; o[c++] = s[0];
          p2_internal_matrix(s, mk);
        }
      }
      for (int r = 0; r < P2_FULL / 2; r++) {
        M init[P2_T];
        for (int i = 0; i < P2_T; i++) { s[i] = s[i] + O::mk(p2_ext_rc(half * 4 + r, i)); init[i] = s[i]; }
        for (int i = 0; i < P2_T; i++) { s[i] = s[i] * s[i]; o[c++] = s[i]; }
        for (int i = 0; i < P2_T; i++) { s[i] = s[i] * s[i]; o[c++] = s[i]; }
        for (int i = 0; i < P2_T; i++) s[i] = s[i] * init[i];
        p2_external_matrix(s);
        for (int i = 0; i < P2_T; i++) o[c++] = s[i];
      }
    }
  }
  // eval: poseidon2.rs:392-505
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    auto mk = [&e](uint32_t v) { return e.c(v); };
    F enabler = e.next();
    F s[P2_T], initial_state[P2_T];
    for (int i = 0; i < P2_T; i++) { s[i] = e.next(); initial_state[i] = s[i]; }
    p2_external_matrix(s);
    for (int half = 0; half < 2; half++) {
      if (half == 1) {
        for (int r = 0; r < P2_PARTIAL; r++) {
          s[0] = s[0] + e.c(p2_int_rc(r));
          F init = s[0];
          F m = e.next();
          e.constraint(enabler * (s[0] * s[0] - m));
          s[0] = m;
          m = e.next();
          e.constraint(enabler * (s[0] * s[0] - m));
          s[0] = m;
          m = e.next();
          e.constraint(enabler * (init * s[0] - m));
          s[0] = m;
          p2_internal_matrix(s, mk);
        }
      }
      for (int r = 0; r < P2_FULL / 2; r++) {
        F init[P2_T];
        for (int i = 0; i < P2_T; i++) { s[i] = s[i] + e.c(p2_ext_rc(half * 4 + r, i)); init[i] = s[i]; }
        for (int i = 0; i < P2_T; i++) s[i] = s[i] * s[i];
        for (int i = 0; i < P2_T; i++) { F m = e.next(); e.constraint(enabler * (s[i] - m)); s[i] = m; }
        for (int i = 0; i < P2_T; i++) s[i] = s[i] * s[i];
        for (int i = 0; i < P2_T; i++) { F m = e.next(); e.constraint(enabler * (s[i] - m)); s[i] = m; }
        for (int i = 0; i < P2_T; i++) s[i] = s[i] * init[i];
        p2_external_matrix(s);
        for (int i = 0; i < P2_T; i++) { F m = e.next(); e.constraint(enabler * (s[i] - m)); s[i] = m; }
      }
    }
    e.rel_arr(REL_POSEIDON2, -enabler, initial_state, P2_T);
    e.rel(REL_POSEIDON2, enabler, s[0]);
    e.finalize_pairs();
  }
};

// ------------------------------------------------------------------------------------------------
// range_check_macro.rs:165-184 — 1 trace column (multiplicity) + 1 preprocessed column
template <int PP, int REL>
struct RangeCheckC {
  static constexpr int N_TRACE = 1;
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F value = e.preproc(PP);
    F multiplicity = e.next();
    e.rel(REL, multiplicity, value);
    e.finalize_single();
  }
};
// bitwise.rs:217-238 — 1 trace column + 4 preprocessed columns
struct BitwiseC {
  static constexpr int N_TRACE = 1;
  template <class E>
  static AIR_HD void eval(E& e) {
    using F = typename E::F;
    F op = e.preproc(PP_BITWISE_0), in1 = e.preproc(PP_BITWISE_1), in2 = e.preproc(PP_BITWISE_2), res = e.preproc(PP_BITWISE_3);
    F multiplicity = e.next();
    e.rel(REL_BITWISE, multiplicity, op, in1, in2, res);
    e.finalize_single();
  }
};

// preprocessed column value at row i (preprocessed/range_check/mod.rs:62-67, bitwise.rs:283-319)
AIR_HD uint32_t preproc_value(int pp, uint32_t i) {
  if (pp >= PP_RC8) return i;
  if (i >= 3u * 65536u) return 0;
  uint32_t op = i >> 16, in1 = (i >> 8) & 0xff, in2 = i & 0xff;
  switch (pp) {
    case PP_BITWISE_0: return op;
    case PP_BITWISE_1: return in1;
    case PP_BITWISE_2: return in2;
    default: return op == 0 ? (in1 & in2) : op == 1 ? (in1 | in2) : (in1 ^ in2);
  }
}

}  // namespace air
