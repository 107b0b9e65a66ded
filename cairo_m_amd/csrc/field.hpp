// M31 / CM31 / QM31 arithmetic for host and gfx950 device code (product path).
// Values are canonical u32 in [0, P).  Layout of QM31 = ((a,b),(c,d)) as 4 u32, matching the
// reference's use of Stwo's SecureField (crates/prover/src/public_data.rs:146-149).
// All reductions use the Mersenne fold (no division); results are bit-identical to `x mod P`.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CM_HD __host__ __device__ __forceinline__
#else
#define CM_HD inline
#endif

namespace cm {

constexpr uint32_t P = 0x7fffffffu;

struct M31 {
  uint32_t v;
  CM_HD M31() : v(0) {}
  CM_HD explicit M31(uint32_t x) : v(x) {}  // x must already be < P
  static CM_HD M31 reduce(uint64_t x) {     // any x < 2^62
    uint32_t lo = (uint32_t)(x & P), hi = (uint32_t)(x >> 31);
    uint32_t s = lo + (hi & P) + (uint32_t)(x >> 62);
    s = (s & P) + (s >> 31);
    return M31(s >= P ? s - P : s);
  }
  static CM_HD M31 from_u32(uint32_t x) {  // any u32
    uint32_t s = (x & P) + (x >> 31);
    return M31(s >= P ? s - P : s);
  }
  CM_HD bool is_zero() const { return v == 0; }
};
// Conditional subtraction of P.  On gfx950 `min(s, s - P)` costs a 4-cycle VOP3 v_min_u32 after the subtract;
// subtract-with-borrow + select is two 2-cycle VOP2 ops (v_sub_co_u32, v_cndmask_b32) — tools/valu_lab.hip.
CM_HD uint32_t m31_csub(uint32_t s) {  // s < 2P  ->  s mod P
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  const bool borrow = __builtin_usub_overflow(s, P, &r);
  return borrow ? s : r;
#else
  return s >= P ? s - P : s;
#endif
}
CM_HD M31 operator+(M31 a, M31 b) { return M31(m31_csub(a.v + b.v)); }
CM_HD M31 operator-(M31 a, M31 b) {
  uint32_t s = a.v - b.v;
  return M31(a.v < b.v ? s + P : s);
}
CM_HD M31 operator-(M31 a) { return M31(a.v ? P - a.v : 0); }
CM_HD M31 operator*(M31 a, M31 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (2a) * b = 2t: the high word of the 64-bit product IS t >> 31 and (low word >> 1) IS t & P — a shift, the
  // multiply-add and a shift instead of multiply, and, 4-cycle v_alignbit (t = a*b < 2^62, so 2t < 2^63)
  uint64_t t2 = (uint64_t)(a.v << 1) * b.v;
  uint32_t s = ((uint32_t)t2 >> 1) + (uint32_t)(t2 >> 32);
#else
  uint64_t t = (uint64_t)a.v * b.v;
  uint32_t s = (uint32_t)(t & P) + (uint32_t)(t >> 31);
#endif
  return M31(m31_csub(s));
}
// a * w for a twiddle stored DOUBLED (w2 = 2w < 2^32, the form of every entry of the Twiddles tables): the high word of
// a * w2 is (a w) >> 31 and (low word >> 1) is (a w) & P — the shift of the generic product (2a) * w is paid once, when the
// table is built, instead of once per butterfly.
CM_HD M31 mul_tw2(M31 a, uint32_t w2) {
  const uint64_t t2 = (uint64_t)a.v * w2;
  return M31(m31_csub(((uint32_t)t2 >> 1) + (uint32_t)(t2 >> 32)));
}
CM_HD M31& operator+=(M31& a, M31 b) { a = a + b; return a; }
CM_HD M31& operator-=(M31& a, M31 b) { a = a - b; return a; }
CM_HD M31& operator*=(M31& a, M31 b) { a = a * b; return a; }
CM_HD bool operator==(M31 a, M31 b) { return a.v == b.v; }
CM_HD bool operator!=(M31 a, M31 b) { return a.v != b.v; }

template <int N>
CM_HD M31 sqn(M31 x) {
#pragma unroll
  for (int i = 0; i < N; i++) x = x * x;
  return x;
}
// x^(P-2); inverse(0) = 0 (callers that need Stwo's panic check for zero themselves).
CM_HD M31 inv(M31 v) {
  M31 t0 = sqn<2>(v) * v;
  M31 t1 = sqn<1>(t0) * t0;
  M31 t2 = sqn<3>(t1) * t0;
  M31 t3 = sqn<1>(t2) * t0;
  M31 t4 = sqn<8>(t3) * t3;
  M31 t5 = sqn<8>(t4) * t3;
  return sqn<7>(t5) * t2;
}

struct CM31 {
  M31 a, b;
  CM_HD CM31() {}
  CM_HD CM31(M31 a_, M31 b_) : a(a_), b(b_) {}
  CM_HD explicit CM31(M31 a_) : a(a_), b() {}
};
CM_HD CM31 operator+(CM31 x, CM31 y) { return CM31(x.a + y.a, x.b + y.b); }
CM_HD CM31 operator-(CM31 x, CM31 y) { return CM31(x.a - y.a, x.b - y.b); }
CM_HD CM31 operator-(CM31 x) { return CM31(-x.a, -x.b); }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ M31 m31_fold64(unsigned long long s);
#endif
// device form: each coordinate is one unreduced 64-bit sum of two raw products (-x.b as P - x.b) and one fold
CM_HD CM31 operator*(CM31 x, CM31 y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CM_QM31_MUL_PLAIN)
  typedef unsigned long long u64;
  const u64 xa = x.a.v, xb = x.b.v, nb = P - x.b.v;
  return CM31(m31_fold64(xa * y.a.v + nb * y.b.v), m31_fold64(xa * y.b.v + xb * y.a.v));
#else
  return CM31(x.a * y.a - x.b * y.b, x.a * y.b + x.b * y.a);
#endif
}
CM_HD CM31 operator*(CM31 x, M31 y) { const uint32_t y2 = y.v << 1; return CM31(mul_tw2(x.a, y2), mul_tw2(x.b, y2)); }   // one shift for both
CM_HD CM31 inv(CM31 x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CM_QM31_MUL_PLAIN)
  M31 n = inv(m31_fold64((unsigned long long)x.a.v * x.a.v + (unsigned long long)x.b.v * x.b.v));
#else
  M31 n = inv(x.a * x.a + x.b * x.b);
#endif
  return CM31(x.a * n, -(x.b * n));
}
// multiply by R = 2 + i
CM_HD CM31 mul_R(CM31 x) { return CM31(x.a + x.a - x.b, x.a + x.b + x.b); }

struct QM31 {
  CM31 a, b;
  CM_HD QM31() {}
  CM_HD QM31(CM31 a_, CM31 b_) : a(a_), b(b_) {}
  CM_HD explicit QM31(M31 x) : a(x), b() {}
  CM_HD QM31(M31 x0, M31 x1, M31 x2, M31 x3) : a(x0, x1), b(x2, x3) {}
  static CM_HD QM31 from_u32(const uint32_t* w) { return QM31(M31(w[0]), M31(w[1]), M31(w[2]), M31(w[3])); }
  CM_HD void to_u32(uint32_t* w) const { w[0] = a.a.v; w[1] = a.b.v; w[2] = b.a.v; w[3] = b.b.v; }
  CM_HD M31 coord(int i) const { return i == 0 ? a.a : i == 1 ? a.b : i == 2 ? b.a : b.b; }
  CM_HD bool is_zero() const { return (a.a.v | a.b.v | b.a.v | b.b.v) == 0; }
};
CM_HD QM31 operator+(QM31 x, QM31 y) { return QM31(x.a + y.a, x.b + y.b); }
CM_HD QM31 operator-(QM31 x, QM31 y) { return QM31(x.a - y.a, x.b - y.b); }
CM_HD QM31 operator-(QM31 x) { return QM31(-x.a, -x.b); }
#if defined(__HIP_DEVICE_COMPILE__)
// Sum of at most four products of values <= P (< 2^64) -> canonical M31: s = lo31 + 2^31 mid31 + 2^62 top2, 2^31 = 1 (mod P)
__device__ __forceinline__ M31 m31_fold64(unsigned long long s) {
#ifdef CM_FOLD64_OLD
  const uint32_t lo = (uint32_t)s, hi = (uint32_t)(s >> 32);
  uint32_t t = (lo & P) + (__funnelshift_r(lo, hi, 31) & P) + (hi >> 30);   // <= 2P + 3 < 2^32
  t = (t & P) + (t >> 31);                                                    // <= P + 1
  return M31(m31_csub(t));
#else
  // 2^32 = 2 (mod P): u = lo + 2 hi < 3 * 2^32 (one v_lshl_add_u64), then u = u31 + 2^31 * (u >> 31), u >> 31 <= 5
  const unsigned long long u = (unsigned long long)(uint32_t)s + ((s >> 32) << 1);
  const uint32_t ulo = (uint32_t)u, uhi = (uint32_t)(u >> 32);
  const uint32_t t = (ulo & P) + __funnelshift_r(ulo, uhi, 31);             // <= P + 5
  return M31(m31_csub(t));
#endif
}
#endif
// (a + b u)(c + d u), u^2 = 2 + i.  Device form: every output coordinate is ONE unreduced 64-bit sum of raw products
// (v_mad_u64_u32 each) with the signs folded into the operands (P - x for -x, 2y for a doubled term) and a single-instruction
// partial fold (m31_fold_lazy) wherever more than four product-units would overflow; one Mersenne fold per coordinate at the end
// and NO modular additions:
//   r0 = x0y0 - x1y1 + 2(x2y2 - x3y3) - (x2y3 + x3y2)      r2 = x0y2 - x1y3 + x2y0 - x3y1
//   r1 = x0y1 + x1y0 + (x2y2 - x3y3) + 2(x2y3 + x3y2)      r3 = x0y3 + x1y2 + x2y1 + x3y0
// 20 mads + 2 partial + 4 full folds (~230 issue cycles) against 16 mads + 6 full folds + 7 modular adds (~300) of the
// grouped form it replaces; field arithmetic is exact, the canonical result is the same.
CM_HD unsigned long long m31_fold_lazy(unsigned long long x);
CM_HD QM31 operator*(QM31 x, QM31 y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CM_QM31_MUL_PLAIN)
  typedef unsigned long long u64;
  const u64 x0 = x.a.a.v, x1 = x.a.b.v, x2 = x.b.a.v, x3 = x.b.b.v;
  const u64 n1 = P - x.a.b.v, n2 = P - x.b.a.v, n3 = P - x.b.b.v;   // -x1, -x2, -x3 as values in [1, P]
  const uint32_t y0 = y.a.a.v, y1 = y.a.b.v, y2 = y.b.a.v, y3 = y.b.b.v;
  const uint32_t y2d = y2 << 1, y3d = y3 << 1;                        // < 2^32: a product with them counts as two units
  u64 s0 = x0 * y0 + n1 * y1 + x2 * y2d;                               // 4 units
  s0 = m31_fold_lazy(s0) + n3 * y3d + n2 * y3 + n3 * y2;               // + 4 units
  u64 s1 = x0 * y1 + x1 * y0 + x2 * y2 + n3 * y3;                      // 4 units
  s1 = m31_fold_lazy(s1) + x2 * y3d + x3 * y2d;                        // + 4 units
  const u64 s2 = x0 * y2 + n1 * y3 + x2 * y0 + n3 * y1;
  const u64 s3 = x0 * y3 + x1 * y2 + x2 * y1 + x3 * y0;
  return QM31(m31_fold64(s0), m31_fold64(s1), m31_fold64(s2), m31_fold64(s3));
#else
  return QM31(x.a * y.a + mul_R(x.b * y.b), x.a * y.b + x.b * y.a);
#endif
}
CM_HD QM31 operator*(QM31 x, M31 y) {   // the scalar is doubled once (mul_tw2), not every coordinate
  const uint32_t y2 = y.v << 1;
  return QM31(mul_tw2(x.a.a, y2), mul_tw2(x.a.b, y2), mul_tw2(x.b.a, y2), mul_tw2(x.b.b, y2));
}
CM_HD QM31 operator*(M31 y, QM31 x) { return x * y; }
CM_HD QM31 operator+(QM31 x, M31 y) { return QM31(CM31(x.a.a + y, x.a.b), x.b); }
CM_HD QM31 operator-(QM31 x, M31 y) { return QM31(CM31(x.a.a - y, x.a.b), x.b); }
CM_HD QM31 mul_cm31(QM31 x, CM31 y) { return QM31(x.a * y, x.b * y); }
CM_HD QM31& operator+=(QM31& a, QM31 b) { a = a + b; return a; }
CM_HD QM31& operator-=(QM31& a, QM31 b) { a = a - b; return a; }
CM_HD QM31& operator*=(QM31& a, QM31 b) { a = a * b; return a; }
CM_HD bool operator==(QM31 x, QM31 y) {
  return x.a.a.v == y.a.a.v && x.a.b.v == y.a.b.v && x.b.a.v == y.b.a.v && x.b.b.v == y.b.b.v;
}
CM_HD bool operator!=(QM31 x, QM31 y) { return !(x == y); }
CM_HD QM31 conj_u(QM31 x) { return QM31(x.a, -x.b); }  // a - b*u ("complex_conjugate" in Stwo)
CM_HD QM31 inv(QM31 x) {
  CM31 den = x.a * x.a - mul_R(x.b * x.b);
  CM31 di = inv(den);
  return QM31(x.a * di, -(x.b * di));
}
CM_HD QM31 qpow(QM31 x, uint64_t e) {
  QM31 r(M31(1));
  while (e) {
    if (e & 1) r = r * x;
    x = x * x;
    e >>= 1;
  }
  return r;
}

// ---- circle group over a field F (M31 or QM31) ----
template <class F>
struct CPoint {
  F x, y;
};
template <class F>
CM_HD CPoint<F> cadd(CPoint<F> p, CPoint<F> q) {
  return CPoint<F>{p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x};
}
template <class F>
CM_HD CPoint<F> cconj(CPoint<F> p) { return CPoint<F>{p.x, -p.y}; }
// derived `Ord` of CirclePoint<SecureField> (x then y; QM31 as its four words in order): the key of a BTreeMap over sample
// points — framing switch `sample_batch=sorted`
inline bool secure_point_less(const CPoint<QM31>& a, const CPoint<QM31>& b) {
  uint32_t wa[8], wb[8];
  a.x.to_u32(wa); a.y.to_u32(wa + 4);
  b.x.to_u32(wb); b.y.to_u32(wb + 4);
  for (int i = 0; i < 8; i++) if (wa[i] != wb[i]) return wa[i] < wb[i];
  return false;
}
CM_HD M31 double_x(M31 x) { M31 s = x * x; return s + s - M31(1); }
CM_HD QM31 double_x(QM31 x) { QM31 s = x * x; return s + s - M31(1); }

constexpr uint32_t CIRCLE_GEN_X = 2u, CIRCLE_GEN_Y = 1268011823u;

// G^idx (idx mod 2^31) by double-and-add; host or device.
CM_HD CPoint<M31> point_at_index(uint32_t idx) {
  idx &= 0x7fffffffu;
  CPoint<M31> res{M31(1), M31(0)};
  CPoint<M31> cur{M31(CIRCLE_GEN_X), M31(CIRCLE_GEN_Y)};
  while (idx) {
    if (idx & 1) res = cadd(res, cur);
    cur = cadd(cur, cur);
    idx >>= 1;
  }
  return res;
}
CM_HD uint32_t subgroup_gen_index(uint32_t log_size) { return 1u << (31 - log_size); }
// index (exponent of G) of point i (natural order) of CanonicCoset(log).circle_domain()
CM_HD uint32_t domain_index_at(uint32_t log, uint32_t i) {
  uint32_t half = 1u << (log - 1);
  uint32_t init = subgroup_gen_index(log + 1);  // half_odds(log-1).initial = G_{2^(log+1)}
  uint32_t step = subgroup_gen_index(log - 1);
  if (log == 1) step = 0;
  if (i < half) return (init + step * i) & 0x7fffffffu;
  return (0x80000000u - ((init + step * (i - half)) & 0x7fffffffu)) & 0x7fffffffu;
}
CM_HD uint32_t bit_reverse(uint32_t i, uint32_t log) {
  if (log == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(i) >> (32 - log);
#else
  uint32_t r = 0;
  for (uint32_t b = 0; b < log; b++) r |= ((i >> b) & 1u) << (log - 1 - b);
  return r;
#endif
}

// Lazy accumulator for  sum_k c_k * x_k  with c_k in QM31 (4 canonical words) and x_k in M31: the 64-bit
// products of one coordinate are added unreduced — four of them plus a folded remainder fit a u64
// (4 * (2^31-1)^2 + 3 * 2^32 < 2^64) — and folded back below 3 * 2^32 after every fourth term (m31_fold_lazy).  On gfx950 this is
// one v_mad_u64_u32 per coordinate-term instead of a multiply + Mersenne fold + modular add.
// Partial fold of an unreduced 64-bit accumulator: 2^32 = 2 (mod P), so x = lo + 2^32 hi = lo + 2 hi (mod P) < 3 * 2^32.
// Four more products of canonical values fit on top of that (4 (2^31-1)^2 + 3 * 2^32 = 2^64 - 2^32 + 4 < 2^64), so this
// single shift-and-add (one v_lshl_add_u64 on gfx950) is all that is needed between groups of four terms — the two-step
// Mersenne fold ((x & P) + (x >> 31), twice) it replaces was half of the VALU work of the multiply-accumulate loops.
CM_HD unsigned long long m31_fold_lazy(unsigned long long x) {
  return (unsigned long long)(uint32_t)x + ((x >> 32) << 1);
}
struct QAcc {
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  int pending = 0;
  static CM_HD unsigned long long fold(unsigned long long x) { return m31_fold_lazy(x); }
  CM_HD void add(const uint32_t* c4, M31 x) {
    const unsigned long long v = x.v;
    q0 += v * c4[0]; q1 += v * c4[1]; q2 += v * c4[2]; q3 += v * c4[3];
    if (++pending == 4) { q0 = fold(q0); q1 = fold(q1); q2 = fold(q2); q3 = fold(q3); pending = 0; }
  }
  // += c * x for a QM31 coefficient c (4 canonical words) and a QM31 value x, WITHOUT a QM31 product: with the basis
  // e = (1, i, u, iu), c x = sum_j x_j (c e_j) and the coordinates of c e_j are fixed linear forms of c = (a, b, c, d):
  //   c e_0 = (a, b, c, d)   c e_1 = (-b, a, -d, c)   c e_2 = (2c - d, c + 2d, a, b)   c e_3 = (-(c + 2d), 2c - d, -b, a)
  // so every coordinate of the sum takes four raw products (one v_mad_u64_u32 each) on top of the lazily folded accumulator — 16
  // multiply-adds + 4 one-instruction folds instead of a 20-product QM31 multiplication with four Mersenne folds and four modular
  // additions.  The coefficient is wave-uniform in the AIR kernels (a random-coefficient power read through scalar loads), so its five
  // derived words (-b, -d, 2c - d, c + 2d, -(c + 2d)) cost scalar instructions.  Negatives are P - v in [1, P]: x * P = 0 (mod P).
  CM_HD void add_q(const uint32_t* c4, const QM31& x) {
    if (pending) { q0 = fold(q0); q1 = fold(q1); q2 = fold(q2); q3 = fold(q3); pending = 0; }
    const uint32_t a = c4[0], b = c4[1], c = c4[2], d = c4[3];
    const uint32_t nb = P - b, nd = P - d;
    const uint32_t e = (M31(c) + M31(c) - M31(d)).v, f = (M31(c) + M31(d) + M31(d)).v, nf = P - f;
    const unsigned long long x0 = x.a.a.v, x1 = x.a.b.v, x2 = x.b.a.v, x3 = x.b.b.v;
    q0 += x0 * a + x1 * nb + x2 * e + x3 * nf;
    q1 += x0 * b + x1 * a + x2 * f + x3 * e;
    q2 += x0 * c + x1 * nd + x2 * a + x3 * nb;
    q3 += x0 * d + x1 * c + x2 * b + x3 * a;
    q0 = fold(q0); q1 = fold(q1); q2 = fold(q2); q3 = fold(q3);
  }
  CM_HD QM31 value() const {
    return QM31(M31::reduce(fold(q0)), M31::reduce(fold(q1)), M31::reduce(fold(q2)), M31::reduce(fold(q3)));
  }
};

}  // namespace cm
