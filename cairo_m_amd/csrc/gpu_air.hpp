// Device-side evaluators over the shared AIR descriptions (air/*.hpp): one thread = one row.
//   TraceGen   — Claim::write_trace row closures -> column-major stores (coalesced per column)
//   HistEval   — range-check / bitwise multiplicities (global atomics; reference: relaxed AtomicU32,
//                range_check_macro.rs:72-79, bitwise.rs:86-109)
//   LogupEval  — LogUp interaction columns (write_interaction_trace + LogupTraceGenerator::finalize_col)
//   DomainEval — FrameworkComponent::evaluate_constraint_quotients_on_domain (hot loop A)
// Relations / random-coefficient powers are wave-uniform and come through scalar loads.
#pragma once
#include "device_common.hpp"
#include "field.hpp"
#include "air/components.hpp"

namespace cm {

struct DevOps {
  using M = M31;
  static CM_HD M mk(uint32_t v) { return M31(v); }
  static CM_HD M inv(M x) { return cm::inv(x); }
};

struct DevRelations {
  uint32_t z[air::N_RELATIONS][4];
  uint32_t alpha_pow[air::N_RELATIONS][air::MAX_REL_SIZE][4];
};

__device__ __forceinline__ QM31 dev_combine(const DevRelations* __restrict__ rel, int r, const M31* v, int n) {
  QAcc acc;  // sum_i alpha^i * v_i as unreduced 64-bit products (n <= MAX_REL_SIZE, unrolled after inlining)
  for (int i = 0; i < n; i++) acc.add(rel->alpha_pow[r][i], v[i]);
  return acc.value() - QM31::from_u32(rel->z[r]);
}

struct EmptyEF {};
__device__ __forceinline__ EmptyEF operator*(EmptyEF, EmptyEF) { return {}; }
__device__ __forceinline__ EmptyEF operator*(EmptyEF, M31) { return {}; }
__device__ __forceinline__ EmptyEF operator+(EmptyEF, EmptyEF) { return {}; }

struct HistPtrs { uint32_t *rc8, *rc16, *rc20, *bitwise; uint32_t* error_flag; };
constexpr uint32_t HIST_SMALL = 2048;  // per-block LDS bins for the small range-check values
// Values outside the small bins go through a block-private tagged cache (direct-mapped, key = table | index) before
// they touch HBM.  Random keys miss and fall through to a global atomic — those are cheap (27 G/s, tools/atomic_lab.hip)
// — but a HOT key (a constant limb of the program, a frequent bitwise operand pair) would otherwise be one global
// atomic per wave on ONE address, and same-address atomics serialise at ~11 ns each on this part (0.09 G/s).
constexpr uint32_t HIST_CACHE_LOG = 11, HIST_CACHE = 1u << HIST_CACHE_LOG;
constexpr uint32_t HIST_BINS = 2 * HIST_SMALL + 256;           // [rc20 small][rc16 small][rc8]
constexpr uint32_t HIST_LDS_WORDS = HIST_BINS + 2 * HIST_CACHE;  // + [tags][counts]
constexpr uint32_t HIST_EMPTY = 0xffffffffu;
__device__ __forceinline__ void hist_lds_init(uint32_t* lds) {
  for (uint32_t i = threadIdx.x; i < HIST_LDS_WORDS; i += blockDim.x) lds[i] = (i >= HIST_BINS && i < HIST_BINS + HIST_CACHE) ? HIST_EMPTY : 0u;
  __syncthreads();
}
__device__ __forceinline__ void hist_lds_flush(const uint32_t* lds, const HistPtrs& h) {
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < HIST_BINS; i += blockDim.x) {
    uint32_t c = lds[i];
    if (!c) continue;
    if (i < HIST_SMALL) atomicAdd(h.rc20 + i, c);
    else if (i < 2 * HIST_SMALL) atomicAdd(h.rc16 + (i - HIST_SMALL), c);
    else atomicAdd(h.rc8 + (i - 2 * HIST_SMALL), c);
  }
  for (uint32_t i = threadIdx.x; i < HIST_CACHE; i += blockDim.x) {
    const uint32_t key = lds[HIST_BINS + i];
    if (key == HIST_EMPTY) continue;
    uint32_t* t = (key >> 20) == 0 ? h.rc16 : (key >> 20) == 1 ? h.rc20 : h.bitwise;
    atomicAdd(t + (key & 0xfffffu), lds[HIST_BINS + HIST_CACHE + i]);
  }
}

struct HistEval : air::LogupStream<HistEval, M31, EmptyEF> {
  const uint32_t* const* cols;
  uint32_t row;
  int ci = 0;
  HistPtrs h;
  uint32_t* lds;  // [rc20: HIST_SMALL][rc16: HIST_SMALL][rc8: 256][cache tags][cache counts], block-private
  const M31* vals = nullptr;   // the row's trace cells still in registers (k_opcode_trace_hist), else read back from `cols`
  __device__ M31 next() { return vals ? vals[ci++] : M31(cols[ci++][row]); }
  __device__ M31 preproc(int) { return M31(); }
  __device__ M31 c(uint32_t v) { return M31(v); }
  __device__ void constraint(M31) {}
  __device__ EmptyEF combine(int, const M31*, int) { return {}; }
  __device__ EmptyEF ef_from(M31) { return {}; }
  // Lookup values are heavily repeated inside a wave (e.g. clock deltas of a loop body), so plain
  // per-lane atomics serialise on a handful of addresses.  Wave-aggregate first: up to 4 rounds of
  // "leader value -> ballot of equal lanes -> one atomicAdd(popcount)", then per-lane atomics for the rest.
  // n occurrences of (table, idx): block-private cache first, HBM on a tag conflict
  __device__ void add_count(uint32_t* t, uint32_t table_id, uint32_t idx, uint32_t n) {
    const uint32_t key = (table_id << 20) | idx;
    const uint32_t slot = (key * 2654435761u) >> (32 - HIST_CACHE_LOG);
    uint32_t* tags = lds + HIST_BINS;
    const uint32_t old = atomicCAS(tags + slot, HIST_EMPTY, key);
    if (old == HIST_EMPTY || old == key) atomicAdd(tags + HIST_CACHE + slot, n);
    else atomicAdd(t + idx, n);
  }
  __device__ void bump(uint32_t* t, uint32_t table_id, uint32_t idx, uint32_t size) {
    if (idx >= size) { atomicOr(h.error_flag, 1u); return; }
    bool pending = true;
#pragma unroll 1
    for (int round = 0; round < 4; round++) {
      unsigned long long active = __ballot(pending);
      if (!active) return;
      int leader = __ffsll((long long)active) - 1;
      uint32_t lv = (uint32_t)__shfl((int)idx, leader, 64);
      unsigned long long same = __ballot(pending && idx == lv);
      if (pending && idx == lv) {
        if ((int)(threadIdx.x & 63) == leader) add_count(t, table_id, idx, (uint32_t)__popcll(same));
        pending = false;
      }
    }
    if (pending) add_count(t, table_id, idx, 1u);
  }
  __device__ void on_entry(int rel, M31, const M31* v, int) {
    // small values (clock deltas, limbs of small numbers) are counted in block-private LDS bins and
    // flushed once per block; the rest goes to HBM with wave-aggregated atomics
    if (rel == air::REL_RC8) { if (v[0].v < 256u) atomicAdd(lds + 2 * HIST_SMALL + v[0].v, 1u); else atomicOr(h.error_flag, 1u); }
    else if (rel == air::REL_RC16) { if (v[0].v < HIST_SMALL) atomicAdd(lds + HIST_SMALL + v[0].v, 1u); else bump(h.rc16, 0u, v[0].v, 1u << 16); }
    else if (rel == air::REL_RC20) { if (v[0].v < HIST_SMALL) atomicAdd(lds + v[0].v, 1u); else bump(h.rc20, 1u, v[0].v, 1u << 20); }
    else if (rel == air::REL_BITWISE) {
      uint32_t ok = (v[0].v < 3u) & (v[1].v < 256u) & (v[2].v < 256u);
      bump(h.bitwise, 2u, ok ? v[0].v * 65536u + (v[1].v << 8) + v[2].v : 0xffffffffu, 1u << 18);
    }
  }
  __device__ void emit_batch(bool, EmptyEF, EmptyEF) {}
};

struct LogupEval : air::LogupStream<LogupEval, M31, QM31> {
  const uint32_t* const* cols;   // tree-1 trace-domain columns of the component
  const uint32_t* const* pp;     // preprocessed trace-domain columns by PreprocId
  uint32_t* const* out;          // interaction columns (trace domain)
  const DevRelations* rels;
  uint32_t row;
  int ci = 0, batch = 0;
  QM31 prev;
  const uint32_t* trv = nullptr;  // trace cells of this row already in registers (see k_logup)
  __device__ M31 next() { return trv ? M31(trv[ci++]) : M31(CM_GCOL(cols[ci++])[row]); }
  __device__ M31 preproc(int id) { return M31(CM_GCOL(pp[id])[row]); }
  __device__ M31 c(uint32_t v) { return M31(v); }
  __device__ void constraint(M31) {}
  __device__ QM31 combine(int r, const M31* v, int n) { return dev_combine(rels, r, v, n); }
  __device__ QM31 ef_from(M31 m) { return QM31(m); }
  __device__ void on_entry(int, M31, const M31*, int) {}
  // (round 6) ENTRY-WISE fractions.  The interaction column of batch j holds prev + n0/d0 + n1/d1; Stwo forms the pair as
  // (n0 d1 + n1 d0) / (d0 d1) and inverts the product — in the field that is the same element as the sum of the two quotients, so
  // each entry's 1/d is computed on its own: norm_u(d) = d.a^2 - R d.b^2 in CM31, its norm in M31, ONE M31 inversion for a group of
  // up to LOGUP_INV_ENTRIES entries (Montgomery's trick), then conj / norm back up.  24 field multiplications per entry against 71 per
  // PAIR of the batch form (the 20-product d0 d1, the norm of the product, and a 20-product numerator x inverse at the end);
  // the canonical words written are identical.  Every index below is a compile-time constant after inlining (a component's
  // stream is straight-line code), so the buffers live in registers.  CM_LOGUP_BATCH_FORM: the round-3 batch form (A/B).
  // INVARIANT (shared inversion): a zero norm in a group makes inv(all) = 0 and zeroes ALL fractions of the group, where the
  // per-batch form lost one.  A denominator z - sum alpha^i v_i is zero with probability ~2^-124 per entry over the verifier's
  // (z, alpha) — Stwo's own batch inverse panics on it, the reference would not produce a proof either — and a proof made
  // from such a row fails the OODS composition check of this prover (check_composition_at_oods), so it cannot leave the library.
#ifndef CM_LOGUP_BATCH_FORM
#ifndef CM_LOGUP_INV_ENTRIES
#define CM_LOGUP_INV_ENTRIES 8
#endif
  static constexpr int GE = CM_LOGUP_INV_ENTRIES;
  static_assert(GE % 2 == 0, "groups end on a batch boundary");
  QM31 ed[GE];   // denominators, slot 0 = the newest entry (a shift register: constant indices even before `cnt` is promoted)
  M31 em[GE];    // multiplicities
  int cnt = 0;
  __device__ __forceinline__ void rel_arr(int r, M31 mult, const M31* vals, int n) {
    const QM31 den = dev_combine(rels, r, vals, n);
#pragma unroll
    for (int k = GE - 1; k > 0; k--) { ed[k] = ed[k - 1]; em[k] = em[k - 1]; }
    ed[0] = den; em[0] = mult; cnt++;
    if (cnt == GE) flush();
  }
  // an odd entry count leaves a last batch of ONE entry (Stwo's finalize_logup_in_pairs): flush() pairs from the OLDEST entry on, so
  // with an odd count the pairs sit in slots (n-1, n-2) ... (2, 1) and the single entry in slot 0; groups in the middle of a stream
  // are always GE (even) entries
  __device__ __forceinline__ void finalize_pairs() { flush(); }
  __device__ __forceinline__ void finalize_single() { finalize_pairs(); }
  __device__ void emit_batch(bool, QM31, QM31) {}   // (unused: the entries never reach LogupStream's pairing)
  __device__ __forceinline__ void flush() {
    const int n = cnt;
    if (n == 0) return;
    CM31 t[GE];
    M31 nr[GE], pre[GE], ni[GE];
    M31 all;
#pragma unroll
    for (int k = 0; k < GE; k++)
      if (k < n) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CM_LOGUP_NORM_PLAIN)
        {
          // norm_u(d) = a^2 - R b^2 with a = a0 + a1 i, b = b0 + b1 i, R = 2 + i, as ONE unreduced 64-bit sum per coordinate (signs folded
          // into the operands as P - x, doubled terms as doubled operands < 2^32; a lazy fold wherever more than four product-units meet):
          //   re = a0^2 - a1^2 - 2 b0^2 + 2 b1^2 + 2 b0 b1        im = 2 a0 a1 - b0^2 + b1^2 - 4 b0 b1
          // 10 multiply-adds, 2 lazy folds, 2 Mersenne folds — against two lazy CM31 products (8 + 4 folds), mul_R and a modular subtraction
          typedef unsigned long long u64;
          const u64 a0 = ed[k].a.a.v, a1 = ed[k].a.b.v, b0 = ed[k].b.a.v, b1 = ed[k].b.b.v;
          const u64 na1 = P - ed[k].a.b.v, nb0 = P - ed[k].b.a.v;
          const uint32_t b0d = ed[k].b.a.v << 1, b1d = ed[k].b.b.v << 1, a1d = ed[k].a.b.v << 1;
          u64 re = a0 * a0 + na1 * a1 + nb0 * b0d;                 // 1 + 1 + 2 units
          re = m31_fold_lazy(re) + b1 * b1d + b0 * b1d;            // + 2 + 2 units
          u64 im = a0 * a1d + nb0 * b0 + b1 * b1;                  // 2 + 1 + 1 units
          im = m31_fold_lazy(im) + nb0 * b1d + nb0 * b1d;          // + 2 + 2 units  (-4 b0 b1)
          t[k] = CM31(m31_fold64(re), m31_fold64(im));
        }
        nr[k] = m31_fold64((unsigned long long)t[k].a.v * t[k].a.v + (unsigned long long)t[k].b.v * t[k].b.v);
#else
        t[k] = ed[k].a * ed[k].a - mul_R(ed[k].b * ed[k].b);
        nr[k] = t[k].a * t[k].a + t[k].b * t[k].b;
#endif
        pre[k] = k == 0 ? nr[0] : pre[k - 1] * nr[k];
        all = pre[k];
      }
    M31 run = inv(all);
#pragma unroll
    for (int k = GE - 1; k >= 0; k--)
      if (k < n) {
        ni[k] = k == 0 ? run : run * pre[k - 1];
        if (k > 0) run = run * nr[k];
      }
    QM31 fr[GE];   // m_k / d_k
#pragma unroll
    for (int k = 0; k < GE; k++)
      if (k < n) {
        const M31 nm = ni[k] * em[k];                                     // the multiplicity rides on the M31 inverse: one product
        const CM31 ti(t[k].a * nm, -(t[k].b * nm));                       // m / norm_u(d)
        fr[k] = QM31(ed[k].a * ti, -(ed[k].b * ti));                      // m conj_u(d) / norm_u(d)
      }
    // oldest entry first: slots n-1, n-2 make the first batch; an odd n (only in a component's last group) ends on the single slot 0
    const int odd = n & 1;
#pragma unroll
    for (int k = GE - 1; k >= 0; k--)
      if (k < n && ((n - 1 - k) & 1) == 0) {   // k = the older slot of a pair, or the single slot 0 of an odd group
        const bool single = odd && k == 0;
        const QM31 v = single ? prev + fr[0] : prev + fr[k] + fr[k > 0 ? k - 1 : 0];
        CM_GCOL_W(out[4 * batch + 0])[row] = v.a.a.v;
        CM_GCOL_W(out[4 * batch + 1])[row] = v.a.b.v;
        CM_GCOL_W(out[4 * batch + 2])[row] = v.b.a.v;
        CM_GCOL_W(out[4 * batch + 3])[row] = v.b.b.v;
        prev = v;
        batch++;
      }
    cnt = 0;
  }
#else
  // The batches of a row are buffered, up to LOGUP_INV_GROUP at a time, and their denominators inverted TOGETHER: a QM31 inverse
  // is conj / norm with one M31 inversion (37 multiplications by Fermat) at the bottom; Montgomery's trick replaces the group's
  // M31 inversions by one inversion + 3 multiplications each.  Every index below is a compile-time constant after inlining (the
  // stream of a component is straight-line code), so the buffers live in registers.  CM_LOGUP_INV_GROUP=1: one inversion per batch.
#ifndef CM_LOGUP_INV_GROUP
#define CM_LOGUP_INV_GROUP 6
#endif
  // INVARIANT (shared inversion): a zero norm in a group makes inv(all) = 0 and zeroes ALL fractions of the group, where the
  // per-batch form lost one.  A denominator z - sum alpha^i v_i is zero with probability ~2^-124 per entry over the verifier's
  // (z, alpha) — Stwo's own batch inverse panics on it, the reference would not produce a proof either — and a proof made
  // from such a row fails the OODS composition check of this prover (check_composition_at_oods), so it cannot leave the library.
  static constexpr int G = CM_LOGUP_INV_GROUP;
  QM31 bn[G], bd[G];
  int cnt = 0;
  // (the buffer is a shift register — slot 0 = the newest batch — so that every array index is a constant even before the
  // compiler has promoted `cnt`: a `bn[cnt]` store keeps the whole evaluator in scratch memory)
  __device__ __forceinline__ void emit_batch(bool, QM31 num, QM31 den) {
#pragma unroll
    for (int k = G - 1; k > 0; k--) { bn[k] = bn[k - 1]; bd[k] = bd[k - 1]; }
    bn[0] = num; bd[0] = den; cnt++;
    if (cnt == G) flush();
  }
  __device__ __forceinline__ void flush() {
    const int n = cnt;
    if (n == 0) return;
    CM31 d[G];
    M31 nr[G], pre[G], ni[G];
    M31 all;
#pragma unroll
    for (int k = 0; k < G; k++)
      if (k < n) {
        d[k] = bd[k].a * bd[k].a - mul_R(bd[k].b * bd[k].b);
#if defined(__HIP_DEVICE_COMPILE__)
        nr[k] = m31_fold64((unsigned long long)d[k].a.v * d[k].a.v + (unsigned long long)d[k].b.v * d[k].b.v);
#else
        nr[k] = d[k].a * d[k].a + d[k].b * d[k].b;
#endif
        pre[k] = k == 0 ? nr[0] : pre[k - 1] * nr[k];
        all = pre[k];
      }
    M31 t = inv(all);
#pragma unroll
    for (int k = G - 1; k >= 0; k--)
      if (k < n) {
        ni[k] = k == 0 ? t : t * pre[k - 1];
        if (k > 0) t = t * nr[k];
      }
#pragma unroll
    for (int k = G - 1; k >= 0; k--)   // oldest batch first
      if (k < n) {
        const CM31 di(d[k].a * ni[k], -(d[k].b * ni[k]));
        const QM31 v = prev + bn[k] * QM31(bd[k].a * di, -(bd[k].b * di));
        CM_GCOL_W(out[4 * batch + 0])[row] = v.a.a.v;
        CM_GCOL_W(out[4 * batch + 1])[row] = v.a.b.v;
        CM_GCOL_W(out[4 * batch + 2])[row] = v.b.a.v;
        CM_GCOL_W(out[4 * batch + 3])[row] = v.b.b.v;
        prev = v;
        batch++;
      }
    cnt = 0;
  }
#endif
};

struct DomainEval : air::LogupStream<DomainEval, M31, QM31> {
  const uint32_t* const* tr;   // tree-1 LDE columns
  const uint32_t* const* it;   // tree-2 LDE columns
  const uint32_t* const* pp;   // tree-0 LDE columns by PreprocId
  const DevRelations* rels;
  const uint32_t* coeff;       // 4 u32 per constraint
  const uint32_t* const* it_prev = nullptr;   // optional: the four cumulative-sum columns AT THE PREVIOUS ROW, indexed by `row` (sharded halo)
  uint32_t row, prev_row;
  int n_base;
  QM31 cumsum_shift;
  int ci = 0, ii = 0, kb = 0, kl = 0;
  QM31 prev_col, acc;
  QAcc base_acc;  // sum over the add_constraint constraints of rho^k * C_k, accumulated lazily
  const uint32_t* trv = nullptr;  // trace cells of this row already in registers (see k_constraints)
  __device__ M31 next() { return trv ? M31(trv[ci++]) : M31(CM_GCOL(tr[ci++])[row]); }
  __device__ M31 preproc(int id) { return M31(CM_GCOL(pp[id])[row]); }
  __device__ M31 c(uint32_t v) { return M31(v); }
  __device__ void constraint(M31 x) { base_acc.add(coeff + 4 * (kb++), x); }
  __device__ QM31 total() const { return base_acc.value(); }
  // (round 6) rho^k * X for a QM31 constraint value X joins the same lazy accumulator as the base-field constraints (QAcc::add_q):
  // no QM31 product, no reduction until total()
#ifdef CM_CONSTRAINT_Q_PLAIN   /* A/B: the round-5 form */
  __device__ void constraint_q(QM31 x) { acc += QM31::from_u32(coeff + 4 * (n_base + kl++)) * x; }
  __device__ QM31 total_plain() const { return acc + base_acc.value(); }
#else
  __device__ void constraint_q(QM31 x) { base_acc.add_q(coeff + 4 * (n_base + kl++), x); }
#endif
  __device__ QM31 combine(int r, const M31* v, int n) { return dev_combine(rels, r, v, n); }
  __device__ QM31 ef_from(M31 m) { return QM31(m); }
  __device__ void on_entry(int, M31, const M31*, int) {}
  __device__ QM31 mask(uint32_t r) { return QM31(M31(CM_GCOL(it[ii])[r]), M31(CM_GCOL(it[ii + 1])[r]), M31(CM_GCOL(it[ii + 2])[r]), M31(CM_GCOL(it[ii + 3])[r])); }
  __device__ void emit_batch(bool last, QM31 num, QM31 den) {
    if (!last) {
      QM31 cur = mask(row);
      ii += 4;
      QM31 diff = cur - prev_col;
      prev_col = cur;
      constraint_q(diff * den - num);
    } else {
      QM31 pr = it_prev ? QM31(M31(CM_GCOL(it_prev[0])[row]), M31(CM_GCOL(it_prev[1])[row]), M31(CM_GCOL(it_prev[2])[row]), M31(CM_GCOL(it_prev[3])[row]))
                        : mask(prev_row);
      QM31 cur = mask(row);
      ii += 4;
      constraint_q((cur - pr - prev_col + cumsum_shift) * den - num);
    }
  }
};

// (shifted_row: device_common.hpp)

}  // namespace cm
