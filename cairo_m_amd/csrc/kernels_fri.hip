// DEEP/FRI kernels for gfx950.
//   k_quotients      : QuotientOps::accumulate_quotients (hot loop C) — one thread per LDE row,
//                      every committed column of the size group read exactly once, column-major.
//   k_fold_circle / k_fold_line : FriOps::{fold_circle_into_line, fold_line}
// Domain points come from the FFT twiddle tables (no per-row group exponentiation).
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "fri_kernels.hpp"
#include "kprof.hpp"
#include "blake2s_dev.hpp"
#include "framing.hpp"

namespace cm {

// (x, y) of CanonicCoset(l).circle_domain() at bit-reversed storage row r
__device__ __forceinline__ void domain_point_at_row(const TwiddleView& tw, uint32_t l, uint32_t r, M31& x, M31& y) {
  uint32_t h = r >> 1;
  M31 yy(tw.ytw[(1u << (l - 1)) + h] >> 1);   // the tables hold 2w
  y = (r & 1u) ? -yy : yy;
  uint32_t L = tw.R - l;
  M31 xx(tw.xtw[(1u << (tw.R - 1)) - (1u << (tw.R - 1 - L)) + (h >> 1)] >> 1);
  x = (h & 1u) ? -xx : xx;
}

// Threads are arranged as (rows_per_block = 256 / S) x S column slices: every slice sums its share of the
// per-column terms, slices are reduced through LDS, slice 0 finishes the row.  S = 1 for large domains; small
// domains (idle components: 2^5 rows but ~10^3 columns) use S up to 64 so the launch is not one serial loop.
template <int S>
__global__ void __launch_bounds__(256) k_quotients(QuotientArgs a) {
  constexpr int ROWS = 256 / S;
  __shared__ uint32_t red[S > 1 ? 256 * 4 : 4];
  const uint32_t slice = threadIdx.x / ROWS, rl = threadIdx.x % ROWS;
  const uint32_t row = blockIdx.x * ROWS + rl;          // index into the columns (a row slice when the launch is sharded)
  const bool live = row < (a.n_rows ? a.n_rows : (1u << a.log_size));
  M31 px, py;
  if (live) domain_point_at_row(a.tw, a.log_size, a.row0 + row, px, py);
  QM31 acc;
  for (uint32_t b = 0; b < a.n_batches; b++) {
    const QuotientBatch& qb = a.batches[b];
    QM31 num;
    if (S == 1) {
      // Large domains are one row per lane: the per-column term coef_k * f_k(row) is accumulated as raw 64-bit
      // products (4 per coordinate fit in a u64: 4 * (2^31-1)^2 + 2^32 < 2^64) and folded once per group of 4,
      // and 8 column loads are issued before any arithmetic so the lane keeps 8 HBM requests in flight
      // (one load + one QM31 multiply-add at a time ran at ~1 TB/s: latency-bound, not bandwidth-bound).
      if (live) {
        unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;  // < 3 * 2^32 between groups
        auto fold = [](unsigned long long x) -> unsigned long long { return m31_fold_lazy(x); };
        uint32_t k = qb.begin;
        if (a.entry_cols) {   // resolved column pointers: 16 loads in flight
          for (; k + 16 <= qb.end; k += 16) {
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = CM_GCOL(a.entry_cols[k + j])[row];
#pragma unroll
            for (int g = 0; g < 4; g++) {
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const uint32_t* c = a.coef_c + 4 * (k + 4 * g + j);
                const unsigned long long x = v[4 * g + j];
                q0 += x * c[0]; q1 += x * c[1]; q2 += x * c[2]; q3 += x * c[3];
              }
              q0 = fold(q0); q1 = fold(q1); q2 = fold(q2); q3 = fold(q3);
            }
          }
        }
        for (; k + 8 <= qb.end; k += 8) {
          uint32_t v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = CM_GCOL(a.cols[a.col_index[k + j]])[row];
#pragma unroll
          for (int g = 0; g < 2; g++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const uint32_t* c = a.coef_c + 4 * (k + 4 * g + j);
              const unsigned long long x = v[4 * g + j];
              q0 += x * c[0]; q1 += x * c[1]; q2 += x * c[2]; q3 += x * c[3];
            }
            q0 = fold(q0); q1 = fold(q1); q2 = fold(q2); q3 = fold(q3);
          }
        }
        for (; k < qb.end; k++) {
          const uint32_t* c = a.coef_c + 4 * k;
          const unsigned long long x = CM_GCOL(a.cols[a.col_index[k]])[row];
          q0 = fold(q0 + x * c[0]); q1 = fold(q1 + x * c[1]); q2 = fold(q2 + x * c[2]); q3 = fold(q3 + x * c[3]);
        }
        num = QM31(M31::reduce(q0), M31::reduce(q1), M31::reduce(q2), M31::reduce(q3));
      }
    } else if (live) {
      for (uint32_t k = qb.begin + slice; k < qb.end; k += S) {
        M31 v(a.cols[a.col_index[k]][row]);
        num += QM31::from_u32(a.coef_c + 4 * k) * v;
      }
    }
    if (S > 1) {
      __syncthreads();
      num.to_u32(red + 4 * threadIdx.x);
      __syncthreads();
      if (slice == 0) {
        for (int s2 = 1; s2 < S; s2++) num += QM31::from_u32(red + 4 * (s2 * ROWS + rl));
      }
    }
    if (slice == 0 && live) {
      // sum_k (a_k * y + b_k) = A*y + B (A, B summed on the host; field arithmetic is exact)
      num = num - (QM31::from_u32(qb.sum_a) * py + QM31::from_u32(qb.sum_b));
      // denominator (Pr.x - p.x) * Pi.y - (Pr.y - p.y) * Pi.x in CM31
      CM31 prx(M31(qb.point[0]), M31(qb.point[1])), pix(M31(qb.point[2]), M31(qb.point[3]));
      CM31 pry(M31(qb.point[4]), M31(qb.point[5])), piy(M31(qb.point[6]), M31(qb.point[7]));
      CM31 den = (prx - CM31(px)) * piy - (pry - CM31(py)) * pix;
      acc = acc * QM31::from_u32(qb.batch_coeff) + mul_cm31(num, inv(den));
    }
  }
  if (slice == 0 && live) {
    a.out[0][row] = acc.a.a.v; a.out[1][row] = acc.a.b.v; a.out[2][row] = acc.b.a.v; a.out[3][row] = acc.b.b.v;
  }
}

// Large domains, R rows per thread (rows blk * 256 R + k * 256 + tid: every k is one coalesced wave access) and at most two
// sample batches (a size group has the OODS point and, for the LogUp cumulative-sum columns, the previous point):
// the R * n_batches CM31 denominators of a thread are inverted with ONE field inversion (Montgomery's trick: prefix
// products, one inverse, back-substitution) instead of one 37-multiplication exponentiation each — in the one-row kernel the
// inversions were ~40 % of the VALU work of a row.  Same field elements, so the output is bit-identical.
// LEAF: the kernel also writes a.leaf_hashes[row] = hash_node(no children, the row's four words) — the leaf layer of the FRI
// first-layer tree (framing and store pattern of k_fold_leaf / k_merkle_narrow<RFC, false, 4>: the 64 rows of a wave are
// consecutive, their hashes leave through a wave-private LDS window as two wave-contiguous 1 KiB stores).
template <int R, bool LEAF = false, bool RFC = false>
__global__ void __launch_bounds__(256) k_quotients_rows(QuotientArgs a) {
  __shared__ uint4 leaf_stage[LEAF ? 4 : 1][LEAF ? 128 : 1];
  const uint32_t rbase = blockIdx.x * (256u * R) + threadIdx.x;     // n is a multiple of 256 R (the launcher checks)
  const uint32_t nb = a.n_batches;                                   // 1 or 2
  M31 py[R];
  CM31 dinv[2][R];
  {
    CM31 den[2][R];
#pragma unroll
    for (int k = 0; k < R; k++) {
      M31 px;
      domain_point_at_row(a.tw, a.log_size, a.row0 + rbase + 256u * k, px, py[k]);
#pragma unroll
      for (uint32_t b = 0; b < 2; b++) {
        if (b < nb) {
          const QuotientBatch& qb = a.batches[b];
          // (Pr.x - p.x) * Pi.y - (Pr.y - p.y) * Pi.x in CM31
          CM31 prx(M31(qb.point[0]), M31(qb.point[1])), pix(M31(qb.point[2]), M31(qb.point[3]));
          CM31 pry(M31(qb.point[4]), M31(qb.point[5])), piy(M31(qb.point[6]), M31(qb.point[7]));
          den[b][k] = (prx - CM31(px)) * piy - (pry - CM31(py[k])) * pix;
        } else {
          den[b][k] = CM31(M31(1));
        }
      }
    }
    // batch inverse over den[0][0..R), then den[1][0..R) when there are two batches
    constexpr int N = 2 * R;
    CM31 pre[N];   // pre[i] = d_0 * ... * d_i  (d_i = den[i / R][i % R])
    pre[0] = den[0][0];
#pragma unroll
    for (int i = 1; i < N; i++) pre[i] = (i < R || nb == 2) ? pre[i - 1] * den[i / R][i % R] : pre[i - 1];
    CM31 run = inv(pre[N - 1]);
#pragma unroll
    for (int i = N - 1; i >= 1; i--) {
      if (i < R || nb == 2) {
        dinv[i / R][i % R] = run * pre[i - 1];
        run = run * den[i / R][i % R];
      } else {
        dinv[i / R][i % R] = CM31(M31(1));
      }
    }
    dinv[0][0] = run;
  }
  QM31 acc[R];
  for (uint32_t b = 0; b < nb; b++) {
    const QuotientBatch& qb = a.batches[b];
    unsigned long long q[R][4];
#pragma unroll
    for (int k = 0; k < R; k++) { q[k][0] = 0; q[k][1] = 0; q[k][2] = 0; q[k][3] = 0; }
    uint32_t e = qb.begin;
    constexpr int CH = 16 / R;     // columns per step: 16 loads in flight per lane
    for (; e + CH <= qb.end; e += CH) {
      uint32_t v[CH][R];
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const cm_gptr col = CM_GCOL(a.entry_cols[e + j]);
#pragma unroll
        for (int k = 0; k < R; k++) v[j][k] = col[rbase + 256u * k];
      }
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const uint32_t* c = a.coef_c + 4 * (e + j);
        const unsigned long long c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
#pragma unroll
        for (int k = 0; k < R; k++) {
          const unsigned long long x = v[j][k];
          q[k][0] += x * c0; q[k][1] += x * c1; q[k][2] += x * c2; q[k][3] += x * c3;
        }
        if ((j & 3) == 3) {   // four raw products per coordinate fit a u64 next to the lazily folded rest
#pragma unroll
          for (int k = 0; k < R; k++) { q[k][0] = m31_fold_lazy(q[k][0]); q[k][1] = m31_fold_lazy(q[k][1]); q[k][2] = m31_fold_lazy(q[k][2]); q[k][3] = m31_fold_lazy(q[k][3]); }
        }
      }
      if (CH & 3) {
#pragma unroll
        for (int k = 0; k < R; k++) { q[k][0] = m31_fold_lazy(q[k][0]); q[k][1] = m31_fold_lazy(q[k][1]); q[k][2] = m31_fold_lazy(q[k][2]); q[k][3] = m31_fold_lazy(q[k][3]); }
      }
    }
    for (; e < qb.end; e++) {
      const cm_gptr col = CM_GCOL(a.entry_cols[e]);
      const uint32_t* c = a.coef_c + 4 * e;
#pragma unroll
      for (int k = 0; k < R; k++) {
        const unsigned long long x = col[rbase + 256u * k];
        q[k][0] = m31_fold_lazy(q[k][0] + x * c[0]); q[k][1] = m31_fold_lazy(q[k][1] + x * c[1]);
        q[k][2] = m31_fold_lazy(q[k][2] + x * c[2]); q[k][3] = m31_fold_lazy(q[k][3] + x * c[3]);
      }
    }
    const QM31 sum_a = QM31::from_u32(qb.sum_a), sum_b = QM31::from_u32(qb.sum_b), bc = QM31::from_u32(qb.batch_coeff);
#pragma unroll
    for (int k = 0; k < R; k++) {
      QM31 num(M31::reduce(q[k][0]), M31::reduce(q[k][1]), M31::reduce(q[k][2]), M31::reduce(q[k][3]));
      num = num - (sum_a * py[k] + sum_b);
      const CM31 di = b == 0 ? dinv[0][k] : dinv[1][k];
      acc[k] = acc[k] * bc + mul_cm31(num, di);
    }
  }
#pragma unroll
  for (int k = 0; k < R; k++) {
    const uint32_t row = rbase + 256u * k;
    a.out[0][row] = acc[k].a.a.v; a.out[1][row] = acc[k].a.b.v; a.out[2][row] = acc[k].b.a.v; a.out[3][row] = acc[k].b.b.v;
  }
  if (LEAF) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint4* st = leaf_stage[w];
#pragma unroll
    for (int k = 0; k < R; k++) {
      uint32_t z[16];
#pragma unroll
      for (int i = 0; i < 16; i++) z[i] = 0;   // 12 literal zeros: most message additions of the compression fold away
      z[0] = acc[k].a.a.v; z[1] = acc[k].a.b.v; z[2] = acc[k].b.a.v; z[3] = acc[k].b.b.v;
      NodeFrame<RFC> fr(false, 4);
      uint32_t h[8];
      fr.init(h);
      fr.absorb(h, z, 16);
      st[lane * 2 + 0] = make_uint4(h[0], h[1], h[2], h[3]);
      st[lane * 2 + 1] = make_uint4(h[4], h[5], h[6], h[7]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t wave_row0 = blockIdx.x * (256u * R) + 256u * k + 64u * w;   // the wave's 64 consecutive rows of this k
      uint4* o = reinterpret_cast<uint4*>(a.leaf_hashes + (size_t)wave_row0 * 8);
      o[lane] = st[lane];
      o[64 + lane] = st[64 + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// compute_fri_quotients' per-sample coefficients on the device (one block per batch): entry k of a batch gets
// alpha = coeff^(k+1); with v its sampled value and (Px, Py) the batch's point: a = conj(v) - v, c = conj(Py) - Py,
// b = v c - a Py; coef_c = alpha c, sum_a = sum alpha a, sum_b = sum alpha b, batch_coeff = coeff^n.
__global__ void __launch_bounds__(256) k_quotient_coeffs(const QuotientCoefJob* __restrict__ jobs, const uint32_t* __restrict__ samples,
                                                         uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3) {
  const QuotientCoefJob jb = jobs[blockIdx.x];
  const QM31 coeff = QM31(M31(q0), M31(q1), M31(q2), M31(q3));
  const uint32_t begin = jb.qb->begin, end = jb.qb->end;
  const QM31 py = QM31::from_u32(jb.qb->point + 4);
  const QM31 cdiff = conj_u(py) - py;
  QM31 sa, sb;
  // entry k of the batch gets coeff^(k+1): one power per thread, then a stride of coeff^256 (a power per entry made the
  // kernel 41 us of dependent multiplications on 21 blocks)
  QM31 alpha = qpow(coeff, (uint64_t)threadIdx.x + 1);
  const QM31 stride = end - begin > 256 ? qpow(coeff, 256) : QM31();
  for (uint32_t e = begin + threadIdx.x; e < end; e += 256, alpha = alpha * stride) {
    const QM31 v = QM31::from_u32(samples + 4 * (size_t)jb.sample_idx[e]);
    const QM31 a = conj_u(v) - v;
    const QM31 b = v * cdiff - a * py;
    sa += alpha * a;
    sb += alpha * b;
    (alpha * cdiff).to_u32(jb.coef_c + 4 * (size_t)e);
  }
  sa = block_reduce_qm31(sa);
  sb = block_reduce_qm31(sb);
  if (threadIdx.x == 0) {
    sa.to_u32(jb.qb->sum_a);
    sb.to_u32(jb.qb->sum_b);
    qpow(coeff, (uint64_t)(end - begin)).to_u32(jb.qb->batch_coeff);
  }
}

__device__ __forceinline__ QM31 ld4(const uint32_t* const* c, uint32_t i) {
  return QM31(M31(c[0][i]), M31(c[1][i]), M31(c[2][i]), M31(c[3][i]));
}
__device__ __forceinline__ void st4(uint32_t* const* c, uint32_t i, QM31 v) {
  c[0][i] = v.a.a.v; c[1][i] = v.a.b.v; c[2][i] = v.b.a.v; c[3][i] = v.b.b.v;
}
struct Ptr4 { uint32_t* p[4]; };
struct CPtr4 { const uint32_t* p[4]; };

// dst[i] = dst[i] * alpha^2 + (f0 + f1) + alpha * (f0 - f1) / y_i,  (f0, f1) = src[2i], src[2i+1]
__global__ void __launch_bounds__(256) k_fold_circle(Ptr4 dst, CPtr4 src, uint32_t log_n, TwiddleView tw, const uint32_t alpha4_0,
                                                     const uint32_t alpha4_1, const uint32_t alpha4_2, const uint32_t alpha4_3,
                                                     int accumulate, const uint32_t* __restrict__ alpha_dev, uint32_t i0, uint32_t n_out) {
  // (i0, n_out): the launch computes outputs [i0, i0 + n_out) of the fold into arrays that hold just that row range
  // (sharded FRI: dst / src are a rank's slices); the whole layer is i0 = 0, n_out = 2^(log_n - 1)
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  QM31 alpha = alpha_dev ? QM31::from_u32(alpha_dev) : QM31(M31(alpha4_0), M31(alpha4_1), M31(alpha4_2), M31(alpha4_3));
  M31 yinv(tw.iytw[(1u << (log_n - 1)) + i0 + i] >> 1);
  QM31 f0 = ld4(src.p, 2 * i), f1 = ld4(src.p, 2 * i + 1);
  QM31 v = (f0 + f1) + alpha * ((f0 - f1) * yinv);
  if (accumulate) v = ld4(dst.p, i) * (alpha * alpha) + v;
  st4(dst.p, i, v);
}
// out[i] = (f0 + f1) + alpha * (f0 - f1) / x_i on LineDomain(half_odds(log_n))
__global__ void __launch_bounds__(256) k_fold_line(Ptr4 out, CPtr4 src, uint32_t log_n, TwiddleView tw, const uint32_t alpha4_0,
                                                   const uint32_t alpha4_1, const uint32_t alpha4_2, const uint32_t alpha4_3,
                                                   const uint32_t* __restrict__ alpha_dev, uint32_t i0, uint32_t n_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  QM31 alpha = alpha_dev ? QM31::from_u32(alpha_dev) : QM31(M31(alpha4_0), M31(alpha4_1), M31(alpha4_2), M31(alpha4_3));
  uint32_t L = tw.R - (log_n + 1);
  M31 xinv(tw.ixtw[(1u << (tw.R - 1)) - (1u << (tw.R - 1 - L)) + i0 + i] >> 1);
  QM31 f0 = ld4(src.p, 2 * i), f1 = ld4(src.p, 2 * i + 1);
  st4(out.p, i, (f0 + f1) + alpha * ((f0 - f1) * xinv));
}

// fold_line into the next layer and, in the same pass, fold the quotient columns of that size in:
// out[i] = fold_line(src)[i] * alpha_c^2 + fold_circle(circle)[i]   (alpha = this layer's challenge, alpha_c = the first one)
__global__ void __launch_bounds__(256) k_fold_line_circle(Ptr4 out, CPtr4 src, CPtr4 circle, uint32_t log_n, TwiddleView tw,
                                                          const uint32_t* __restrict__ alpha_dev, const uint32_t* __restrict__ alpha_c_dev) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << (log_n - 1))) return;
  const QM31 alpha = QM31::from_u32(alpha_dev), ac = QM31::from_u32(alpha_c_dev);
  uint32_t L = tw.R - (log_n + 1);
  M31 xinv(tw.ixtw[(1u << (tw.R - 1)) - (1u << (tw.R - 1 - L)) + i] >> 1);
  QM31 f0 = ld4(src.p, 2 * i), f1 = ld4(src.p, 2 * i + 1);
  const QM31 line = (f0 + f1) + alpha * ((f0 - f1) * xinv);
  // circle evaluation of log_n folds onto the line of log_n - 1 (k_fold_circle with its log = log_n)
  M31 yinv(tw.iytw[(1u << (log_n - 1)) + i] >> 1);
  QM31 g0 = ld4(circle.p, 2 * i), g1 = ld4(circle.p, 2 * i + 1);
  const QM31 v = (g0 + g1) + ac * ((g0 - g1) * yinv);
  st4(out.p, i, line * (ac * ac) + v);
}

// One pass for a FRI layer of 2^(log_n - 1) values AND the leaf layer of its Merkle tree: the folded value is hashed in the
// registers it was computed in.  The separate leaf launch read the four coordinate columns back (16 B per leaf) behind a fold that
// was a memory-bound launch of its own in front of an issue-bound one; here the fold's loads fly under the previous chunk's
// compression.  MODE 0: fold_line (k_fold_line); MODE 1: fold_line + fold_circle of the quotient columns of that size
// (k_fold_line_circle); MODE 2: fold_circle alone into a blank layer (the first inner layer: k_fold_circle, not accumulating).  Leaf framing and store pattern = k_merkle_narrow<RFC, false, 4> (merkle_kernels.hpp): a wave walks `npw`
// chunks of 64 leaves, the hashes leave through a wave-private LDS window as two wave-contiguous 1 KiB stores.
template <bool RFC, int MODE>
__global__ void __launch_bounds__(256) k_fold_leaf(Ptr4 out, CPtr4 src, CPtr4 circle, uint32_t log_n, TwiddleView tw,
                                                   const uint32_t* __restrict__ alpha_dev, const uint32_t* __restrict__ alpha_c_dev,
                                                   uint32_t* __restrict__ hashes, uint32_t npw, uint32_t row0) {
  // (row0: the launch computes outputs [row0, row0 + grid * 256 * npw) of the layer into arrays — values, sources, hashes — that hold
  // just that row range: a rank's slices in the sharded prover; the twiddles are read at the global row.  Whole layer: row0 = 0)
  __shared__ uint4 stage[4][128];           // per wave: 64 leaves x 32 B
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint4* st = stage[w];
  const uint32_t leaf0 = (blockIdx.x * 4u + w) * 64u * npw;
  __builtin_assume(leaf0 < (1u << 29));
  QM31 alpha;
  if (MODE != 2) alpha = QM31::from_u32(alpha_dev);
  QM31 ac, ac2;
  if (MODE >= 1) { ac = QM31::from_u32(alpha_c_dev); ac2 = ac * ac; }
  const uint32_t L = MODE == 2 ? 0u : tw.R - (log_n + 1);
  const uint32_t* __restrict__ ixt = tw.ixtw + ((1u << (tw.R - 1)) - (1u << (tw.R - 1 - L))) + row0;
  const uint32_t* __restrict__ iyt = tw.iytw + (1u << (log_n - 1)) + row0;
  // (plain scalars: indexed arrays of loads captured by reference went to scratch in k_merkle_narrow)
  uint2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0}, c0, c1, c2, c3;
  uint32_t xw = 0, yw = 0;
#define CM_FOLD_ISSUE(i_)                                                              \
  {                                                                                    \
    if (MODE != 2) {                                                                   \
      s0 = reinterpret_cast<const uint2*>(src.p[0])[i_]; s1 = reinterpret_cast<const uint2*>(src.p[1])[i_];   \
      s2 = reinterpret_cast<const uint2*>(src.p[2])[i_]; s3 = reinterpret_cast<const uint2*>(src.p[3])[i_];   \
      xw = ixt[i_];                                                                    \
    }                                                                                  \
    if (MODE >= 1) {                                                                   \
      c0 = reinterpret_cast<const uint2*>(circle.p[0])[i_]; c1 = reinterpret_cast<const uint2*>(circle.p[1])[i_]; \
      c2 = reinterpret_cast<const uint2*>(circle.p[2])[i_]; c3 = reinterpret_cast<const uint2*>(circle.p[3])[i_]; \
      yw = iyt[i_];                                                                    \
    }                                                                                  \
  }
  CM_FOLD_ISSUE(leaf0 + lane)
  for (uint32_t c = 0; c < npw; c++) {
    const uint32_t i = leaf0 + 64u * c + lane;
    const QM31 f0(M31(s0.x), M31(s1.x), M31(s2.x), M31(s3.x)), f1(M31(s0.y), M31(s1.y), M31(s2.y), M31(s3.y));
    QM31 v;
    if (MODE != 2) v = (f0 + f1) + alpha * ((f0 - f1) * M31(xw >> 1));
    if (MODE >= 1) {
      const QM31 g0(M31(c0.x), M31(c1.x), M31(c2.x), M31(c3.x)), g1(M31(c0.y), M31(c1.y), M31(c2.y), M31(c3.y));
      const QM31 cf = (g0 + g1) + ac * ((g0 - g1) * M31(yw >> 1));
      v = MODE == 2 ? cf : v * ac2 + cf;
    }
    if (c + 1 < npw) CM_FOLD_ISSUE(i + 64u)   // the next chunk's loads fly during this chunk's compression
    st4(out.p, i, v);
    uint32_t z[16];
#pragma unroll
    for (int k = 0; k < 16; k++) z[k] = 0;   // 12 literal zeros: most message additions of the compression fold away
    z[0] = v.a.a.v; z[1] = v.a.b.v; z[2] = v.b.a.v; z[3] = v.b.b.v;
    NodeFrame<RFC> fr(false, 4);
    uint32_t h[8];
    fr.init(h);
    fr.absorb(h, z, 16);
    st[lane * 2 + 0] = make_uint4(h[0], h[1], h[2], h[3]);
    st[lane * 2 + 1] = make_uint4(h[4], h[5], h[6], h[7]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4* o = reinterpret_cast<uint4*>(hashes + (size_t)(leaf0 + 64u * c) * 8);
    o[lane] = st[lane];
    o[64 + lane] = st[64 + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#undef CM_FOLD_ISSUE
}

template <bool RFC>   // framing.hpp switch `hash_node`
__global__ void __launch_bounds__(1024) k_fri_tail(FriTailArgs a) {
  // Tree levels of <= 256 nodes: one node per QUAD of lanes (b2s_compress_quad, ~1.4 us instead of ~2.7 us per
  // dependent compression) with the level being consumed kept in LDS; larger levels: one node per lane through HBM.
  __shared__ uint32_t bufA[256 * 8];
  __shared__ uint32_t bufB[128 * 8];
  __shared__ uint32_t s_x8[8];
  __shared__ uint32_t s_alpha[4];
  uint32_t* buf[2];
  buf[0] = bufA;
  buf[1] = bufB;
  const uint32_t tid = threadIdx.x;
  const uint32_t node = tid >> 2, q = tid & 3u;
  uint32_t ai = a.first_index;
  if (tid < 4) s_alpha[tid] = a.alphas[tid];   // the circle-fold challenge
  __syncthreads();
  for (uint32_t l = a.top_log; l > a.last_log; l--, ai++) {
    const FriTailLayer& L = a.layers[l];
    const uint32_t n = 1u << l;
    if (L.circle[0]) {  // fold_circle_into_line of the quotient columns of log l + 1 (k_fold_circle)
      const QM31 alpha = QM31::from_u32(a.alphas);
      const QM31 alpha2 = alpha * alpha;
      for (uint32_t i = tid; i < n; i += 1024) {
        M31 yinv(a.tw.iytw[n + i] >> 1);
        QM31 f0 = ld4(L.circle, 2 * i), f1 = ld4(L.circle, 2 * i + 1);
        st4(L.cols, i, ld4(L.cols, i) * alpha2 + ((f0 + f1) + alpha * ((f0 - f1) * yinv)));
      }
      __syncthreads();
    }
    // Merkle tree of the 4 coordinate columns (k_merkle_layer framing: one compression per node)
    int cur = 0;          // LDS buffer the next quad-mode level writes
    bool prev_lds = false;  // level k + 1 sits in buf[cur ^ 1]
    for (int k = (int)l; k >= 0; k--) {
      const uint32_t nk = 1u << k;
      const bool leaf = (k == (int)l);
      if (nk <= 256) {
        if (node < nk) {
          uint32_t m[16];
          if (leaf) {
#pragma unroll
            for (int c = 0; c < 16; c++) m[c] = 0;
            m[0] = L.cols[0][node]; m[1] = L.cols[1][node]; m[2] = L.cols[2][node]; m[3] = L.cols[3][node];
          } else {
            const uint32_t* p = prev_lds ? buf[cur ^ 1] + node * 16 : L.merkle[k + 1] + (size_t)node * 16;
#pragma unroll
            for (int c = 0; c < 16; c++) m[c] = p[c];
          }
          NodeFrame<RFC> fr(!leaf, leaf ? 4u : 0u);
          uint32_t h0, h1;
          fr.init_quad(q, h0, h1);
          fr.absorb_quad(h0, h1, m, q, leaf ? 16u : 64u);
          uint32_t* o = L.merkle[k] + (size_t)node * 8;
          o[q] = h0; o[4 + q] = h1;
          buf[cur][node * 8 + q] = h0; buf[cur][node * 8 + 4 + q] = h1;
        }
        prev_lds = true;
        cur ^= 1;
      } else {
        for (uint32_t i = tid; i < nk; i += 1024) {
          uint32_t m[16];
          if (leaf) {
#pragma unroll
            for (int c = 0; c < 16; c++) m[c] = 0;
            m[0] = L.cols[0][i]; m[1] = L.cols[1][i]; m[2] = L.cols[2][i]; m[3] = L.cols[3][i];
          } else {
            const uint4* p = reinterpret_cast<const uint4*>(L.merkle[k + 1] + (size_t)i * 16);
            uint4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
            m[0] = c0.x; m[1] = c0.y; m[2] = c0.z; m[3] = c0.w; m[4] = c1.x; m[5] = c1.y; m[6] = c1.z; m[7] = c1.w;
            m[8] = c2.x; m[9] = c2.y; m[10] = c2.z; m[11] = c2.w; m[12] = c3.x; m[13] = c3.y; m[14] = c3.z; m[15] = c3.w;
          }
          NodeFrame<RFC> fr(!leaf, leaf ? 4u : 0u);
          uint32_t h[8];
          fr.init(h);
          fr.absorb(h, m, leaf ? 16u : 64u);
          uint4* o = reinterpret_cast<uint4*>(L.merkle[k] + (size_t)i * 8);
          o[0] = make_uint4(h[0], h[1], h[2], h[3]);
          o[1] = make_uint4(h[4], h[5], h[6], h[7]);
        }
      }
      __syncthreads();
    }
    // the root (level 0) is in buf[cur ^ 1][0..8): transcript step on the first quad
    if (tid < 4) chan_mix_root_draw_quad(tid, a.chan, buf[cur ^ 1], a.alphas + 4 * ai, a.roots + 8 * ai, s_x8, s_alpha);
    __syncthreads();
    {  // fold_line into the next layer (k_fold_line)
      const QM31 alpha = QM31::from_u32(s_alpha);
      uint32_t* const* dst = a.layers[l - 1].cols;
      const uint32_t Lx = a.tw.R - (l + 1);
      const uint32_t* xt = a.tw.ixtw + (1u << (a.tw.R - 1)) - (1u << (a.tw.R - 1 - Lx));
      for (uint32_t i = tid; i < n / 2; i += 1024) {
        M31 xinv(xt[i] >> 1);
        QM31 f0 = ld4(L.cols, 2 * i), f1 = ld4(L.cols, 2 * i + 1);
        st4(dst, i, (f0 + f1) + alpha * ((f0 - f1) * xinv));
      }
    }
    __syncthreads();
  }
}

// ================================================================= host wrappers
// can the two-rows kernel write the leaf hashes of this group?  (whole domain, the default two-rows form, a layer that is a launch
// of its own in MerkleTree::plan_commit)
// (a row range — the sharded prover's slice — is served like the whole domain: the kernel indexes columns, outputs and leaf hashes
// by the LOCAL row and only takes the domain point at row0 + row; the range must be a power of two of at least 2^merkle_multi_top
// rows, so that the leaf layer of the local subtree is a launch of its own)
bool quotient_leaf_serves(const QuotientArgs& a) {
  const uint32_t n = a.n_rows ? a.n_rows : (1u << a.log_size);
  return (n & (n - 1)) == 0 && n >= (1u << tune(T_MERKLE_MULTI_TOP)) && a.row0 % n == 0 && tune(T_QUOT_ROWS) == 2 && a.entry_cols && a.n_batches >= 1 &&
         a.n_batches <= 2;
}
void launch_quotients(const QuotientArgs& a, double n_cols, hipStream_t st) {
  uint32_t n = a.n_rows ? a.n_rows : (1u << a.log_size);
  KProfScope kp("k_quotients", (4.0 * n_cols + 16.0 + (a.leaf_hashes ? 32.0 : 0.0)) * (double)n, st);
  // two rows per thread with one shared denominator inversion (A/B: CM_QUOT_ROWS=1 restores the one-row kernel)
  const int rows = tune(T_QUOT_ROWS);
  if (a.leaf_hashes) {   // (the caller has checked quotient_leaf_serves)
    CM_CHECK(quotient_leaf_serves(a), "launch_quotients: leaf hashes asked for a launch the row kernel does not serve");
    if (framing().hash_node_rfc) hipLaunchKernelGGL((k_quotients_rows<2, true, true>), dim3(n / 512), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_quotients_rows<2, true, false>), dim3(n / 512), dim3(256), 0, st, a);
  } else if (a.log_size >= 14 && rows == 2 && a.entry_cols && a.n_batches >= 1 && a.n_batches <= 2 && n % 512 == 0)
    hipLaunchKernelGGL(k_quotients_rows<2>, dim3(n / 512), dim3(256), 0, st, a);
  else if (a.log_size >= 14 && rows == 4 && a.entry_cols && a.n_batches >= 1 && a.n_batches <= 2 && n % 1024 == 0)
    hipLaunchKernelGGL(k_quotients_rows<4>, dim3(n / 1024), dim3(256), 0, st, a);
  else if (a.log_size >= 14) hipLaunchKernelGGL(k_quotients<1>, dim3((n + 255) / 256), dim3(256), 0, st, a);
  else if (a.log_size >= 10) hipLaunchKernelGGL(k_quotients<8>, dim3((n + 31) / 32), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_quotients<64>, dim3((n + 3) / 4), dim3(256), 0, st, a);
  CM_HIP(hipGetLastError());
}
void fold_circle_into_line(uint32_t* const dst[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw,
                           const QM31& alpha, bool accumulate, hipStream_t st, const uint32_t* d_alpha) {
  fold_circle_into_line_rows(dst, src, log_n, tw, alpha, accumulate, 0u, 1u << (log_n - 1), st, d_alpha);
}
void fold_circle_into_line_rows(uint32_t* const dst[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw,
                                const QM31& alpha, bool accumulate, uint32_t i0, uint32_t n_out, hipStream_t st, const uint32_t* d_alpha) {
  CM_CHECK(log_n >= 2 && log_n <= tw.R && (uint64_t)i0 + n_out <= ((uint64_t)1 << (log_n - 1)), "fold_circle: bad log size / row range");
  Ptr4 d; CPtr4 s;
  for (int i = 0; i < 4; i++) { d.p[i] = dst[i]; s.p[i] = src[i]; }
  hipLaunchKernelGGL(k_fold_circle, dim3((n_out + 255) / 256), dim3(256), 0, st, d, s, log_n, view(tw), alpha.a.a.v, alpha.a.b.v,
                     alpha.b.a.v, alpha.b.b.v, accumulate ? 1 : 0, d_alpha, i0, n_out);
  CM_HIP(hipGetLastError());
}
void fold_line(uint32_t* const out[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw, const QM31& alpha,
               hipStream_t st, const uint32_t* d_alpha) {
  fold_line_rows(out, src, log_n, tw, alpha, 0u, 1u << (log_n - 1), st, d_alpha);
}
void fold_line_rows(uint32_t* const out[4], const uint32_t* const src[4], uint32_t log_n, const Twiddles& tw, const QM31& alpha,
                    uint32_t i0, uint32_t n_out, hipStream_t st, const uint32_t* d_alpha) {
  CM_CHECK(log_n >= 1 && log_n + 1 <= tw.R && (uint64_t)i0 + n_out <= ((uint64_t)1 << (log_n - 1)), "fold_line: bad log size / row range");
  Ptr4 d; CPtr4 s;
  for (int i = 0; i < 4; i++) { d.p[i] = out[i]; s.p[i] = src[i]; }
  hipLaunchKernelGGL(k_fold_line, dim3((n_out + 255) / 256), dim3(256), 0, st, d, s, log_n, view(tw), alpha.a.a.v, alpha.a.b.v,
                     alpha.b.a.v, alpha.b.b.v, d_alpha, i0, n_out);
  CM_HIP(hipGetLastError());
}
void fold_line_and_circle(uint32_t* const out[4], const uint32_t* const src[4], const uint32_t* const circle[4], uint32_t log_n,
                          const Twiddles& tw, hipStream_t st, const uint32_t* d_alpha, const uint32_t* d_alpha_circle) {
  CM_CHECK(log_n >= 2 && log_n + 1 <= tw.R, "fold_line_and_circle: bad log size");
  Ptr4 d; CPtr4 s, c;
  for (int i = 0; i < 4; i++) { d.p[i] = out[i]; s.p[i] = src[i]; c.p[i] = circle[i]; }
  uint32_t n = 1u << (log_n - 1);
  hipLaunchKernelGGL(k_fold_line_circle, dim3((n + 255) / 256), dim3(256), 0, st, d, s, c, log_n, view(tw), d_alpha, d_alpha_circle);
  CM_HIP(hipGetLastError());
}
// fold_line (circle == nullptr) or fold_line_and_circle, plus the leaf hashes of the tree over `out` (2^(log_n - 1) leaves of
// four coordinate words); true if the launch was made (the layer must be large enough for the wave-per-chunk walk)
bool fold_line_leaf(uint32_t* const out[4], const uint32_t* const src[4], const uint32_t* const* circle, uint32_t log_n,
                    const Twiddles& tw, hipStream_t st, const uint32_t* d_alpha, const uint32_t* d_alpha_circle, uint32_t* d_leaf_hashes,
                    uint32_t row0, uint32_t n_rows) {
  const bool on = tune(T_FRI_FOLD_LEAF) != 0;   // A/B: 0 = fold, then the leaf launch
  const uint32_t n = n_rows ? n_rows : 1u << (log_n - 1);   // (a row range: the sharded prover's slice, power of two, row0 a multiple)
  if (!on || log_n < 15 || log_n + 1 > tw.R || n < (1u << 14) || (n & (n - 1)) != 0 || row0 % n != 0 || (uint64_t)row0 + n > ((uint64_t)1 << (log_n - 1))) return false;
  uint32_t npw = std::min(8u, std::max(1u, n >> 20));
  while (npw > 1 && (n % (256u * npw)) != 0) npw >>= 1;
  Ptr4 d; CPtr4 s, c;
  for (int i = 0; i < 4; i++) { d.p[i] = out[i]; s.p[i] = src[i]; c.p[i] = circle ? circle[i] : nullptr; }
  // algorithmic bytes: two source values (+ two quotient values) in, the folded value and its hash out
  KProfScope kp("k_fold_leaf", ((circle ? 64.0 : 32.0) + 16.0 + 32.0) * (double)n, st, /* Blake2s compressions */ (double)n);
  const dim3 grid(n / (256u * npw));
  const bool rfc = framing().hash_node_rfc;
#define CM_FL(R, M) hipLaunchKernelGGL((k_fold_leaf<R, M>), grid, dim3(256), 0, st, d, s, c, log_n, view(tw), d_alpha, d_alpha_circle, d_leaf_hashes, npw, row0)
  if (circle) { if (rfc) CM_FL(true, 1); else CM_FL(false, 1); }
  else { if (rfc) CM_FL(true, 0); else CM_FL(false, 0); }
#undef CM_FL
  CM_HIP(hipGetLastError());
  return true;
}
// fold_circle_into_line of ONE group of quotient columns (2^log_n circle evaluations) into a blank layer + its leaf hashes
bool fold_circle_leaf(uint32_t* const out[4], const uint32_t* const circle[4], uint32_t log_n, const Twiddles& tw, hipStream_t st,
                      const uint32_t* d_alpha_circle, uint32_t* d_leaf_hashes, uint32_t row0, uint32_t n_rows) {
  const bool on = tune(T_FRI_FOLD_LEAF) != 0;
  const uint32_t n = n_rows ? n_rows : 1u << (log_n - 1);
  if (!on || log_n < 15 || log_n > tw.R || n < (1u << 14) || (n & (n - 1)) != 0 || row0 % n != 0 || (uint64_t)row0 + n > ((uint64_t)1 << (log_n - 1))) return false;
  uint32_t npw = std::min(8u, std::max(1u, n >> 20));
  while (npw > 1 && (n % (256u * npw)) != 0) npw >>= 1;
  Ptr4 d; CPtr4 s, c;
  for (int i = 0; i < 4; i++) { d.p[i] = out[i]; s.p[i] = nullptr; c.p[i] = circle[i]; }
  KProfScope kp("k_fold_leaf", (32.0 + 16.0 + 32.0) * (double)n, st, /* Blake2s compressions */ (double)n);
  const dim3 grid(n / (256u * npw));
  if (framing().hash_node_rfc)
    hipLaunchKernelGGL((k_fold_leaf<true, 2>), grid, dim3(256), 0, st, d, s, c, log_n, view(tw), (const uint32_t*)nullptr, d_alpha_circle, d_leaf_hashes, npw, row0);
  else
    hipLaunchKernelGGL((k_fold_leaf<false, 2>), grid, dim3(256), 0, st, d, s, c, log_n, view(tw), (const uint32_t*)nullptr, d_alpha_circle, d_leaf_hashes, npw, row0);
  CM_HIP(hipGetLastError());
  return true;
}
void quotient_coeffs(const QuotientCoefJob* d_jobs, uint32_t n_jobs, const uint32_t* d_samples, const QM31& coeff, hipStream_t st) {
  if (!n_jobs) return;
  hipLaunchKernelGGL(k_quotient_coeffs, dim3(n_jobs), dim3(256), 0, st, d_jobs, d_samples, coeff.a.a.v, coeff.a.b.v, coeff.b.a.v,
                     coeff.b.b.v);
  CM_HIP(hipGetLastError());
}
void fri_tail(const FriTailArgs& a, hipStream_t st) {
  CM_CHECK(a.top_log <= FRI_TAIL_MAX_LOG && a.last_log < a.top_log, "fri_tail: bad layer range");
  KProfScope kp("k_fri_tail", 0.0, st);
  if (framing().hash_node_rfc) hipLaunchKernelGGL(k_fri_tail<true>, dim3(1), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(k_fri_tail<false>, dim3(1), dim3(1024), 0, st, a);
  CM_HIP(hipGetLastError());
}

}  // namespace cm
