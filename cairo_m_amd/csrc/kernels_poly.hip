// Circle-FFT kernels for gfx950: twiddle precompute, multi-layer LDS-staged butterfly passes
// (interpolate / evaluate / LDE), bit-reversal, and OODS point evaluation.
//
// Replaces (reference call sites): `SimdBackend::precompute_twiddles` (crates/prover/src/prover.rs:56-60),
// `tree_builder.extend_evals` -> PolyOps::interpolate and `.commit` -> PolyOps::evaluate
// (prover.rs:71-73, 80-82, 100-102), PolyOps::eval_at_point (inside stwo `prove`, prover.rs:131).
//
// Data layout: one column = contiguous u32[2^log] in HBM holding canonical M31 values, evaluations in
// bit-reversed order of the canonic circle domain (Stwo `BitReversedOrder`).  Batches of equal-size
// columns are addressed through a device array of column pointers (blockIdx.y = column).
#include <algorithm>
#include <string.h>
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "kprof.hpp"
#include "fft_pass.hpp"

namespace cm {

// ---------------------------------------------------------------- twiddles
// Layout for root log size R (largest domain):
//   xtw  : line-layer twiddles of the root half coset, layer L at offset 2^(R-1) - 2^(R-1-L),
//          2^(R-2-L) entries: x( half_odds(R-1-L).at(bitrev(h)) )
//   ytw  : circle-layer twiddles of EVERY log size n (1..R) at offset 2^(n-1):
//          y( half_odds(n-1).at(bitrev(h)) ), 2^(n-1) entries
// ixtw/iytw hold the element-wise inverses.
// Every twiddle is a coordinate of G^idx with idx a multiple of 2^(30-R): with B = G^(2^(30-R)) (order 2^(R+1)) the point is
// B^e, e < 2^(R+1), and B^e = hi[e >> TW_LO] + lo[e & (2^TW_LO - 1)] from two small tables — one point addition (4 M31
// multiplications) instead of a 31-step double-and-add (~250).  The inverses come from Montgomery batches of 8 per thread
// (one x^(P-2) per 8 elements).  Together ~10x fewer multiplications than the element-wise form: the tables are rebuilt in
// every proof like the reference does (prover.rs:56-60), so this is on the proof's clock.
constexpr uint32_t TW_LO = 13;
// entries per thread = size of a Montgomery batch (one x^(P-2) per batch): tuning key "tw_batch" (2 / 4 / 8 / 16), default 8
__global__ void k_twiddle_point_tables(uint32_t R, uint32_t* __restrict__ tab) {   // tab: lo[2^TW_LO] then hi[...], (x, y) pairs
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t sh = 30 - R, n_lo = 1u << TW_LO, n_hi = (R + 1 > TW_LO) ? 1u << (R + 1 - TW_LO) : 1u;
  if (t >= n_lo + n_hi) return;
  const uint32_t idx = t < n_lo ? (t << sh) : ((t - n_lo) << (sh + TW_LO));
  CPoint<M31> p = point_at_index(idx);
  tab[2 * t] = p.x.v;
  tab[2 * t + 1] = p.y.v;
}
__device__ __forceinline__ CPoint<M31> twiddle_point(const uint32_t* __restrict__ tab, uint32_t R, uint32_t idx) {
  const uint32_t e = (idx & 0x7fffffffu) >> (30 - R);
  const uint32_t lo = e & ((1u << TW_LO) - 1), hi = (1u << TW_LO) + (e >> TW_LO);
  const uint2 a = *reinterpret_cast<const uint2*>(tab + 2 * lo), b = *reinterpret_cast<const uint2*>(tab + 2 * hi);
  return cadd(CPoint<M31>{M31(a.x), M31(a.y)}, CPoint<M31>{M31(b.x), M31(b.y)});
}
// out[k] = v[k], iout[k] = 1 / v[k] for TW_BATCH non-zero values of one thread
template <uint32_t TW_BATCH>
__device__ __forceinline__ void batch_inverse8(const M31 (&v)[TW_BATCH], M31 (&iv)[TW_BATCH], uint32_t n) {
  M31 pre[TW_BATCH];
  M31 acc(1);
  for (uint32_t k = 0; k < n; k++) { pre[k] = acc; acc = acc * v[k]; }
  M31 ia = inv(acc);
  for (uint32_t k = n; k-- > 0;) { iv[k] = ia * pre[k]; ia = ia * v[k]; }
}
template <uint32_t TW_BATCH>
__global__ void __launch_bounds__(256) k_twiddles_x(uint32_t* __restrict__ xtw, uint32_t* __restrict__ ixtw, uint32_t R, const uint32_t* __restrict__ tab) {
  const uint32_t total = (1u << (R - 1)) - 1;
  const uint32_t nthreads = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  M31 v[TW_BATCH], iv[TW_BATCH];
  uint32_t n = 0;
  for (uint32_t k = 0; k < TW_BATCH; k++) {
    const uint32_t t = t0 + k * nthreads;
    if (t >= total) break;
    // find layer L with offset(L) <= t < offset(L+1); offset(L) = 2^(R-1) - 2^(R-1-L)
    uint32_t rem = (1u << (R - 1)) - t;            // in (2^(R-2-L), 2^(R-1-L)]
    uint32_t L = (R - 1) - (32 - __clz(rem - 1));  // 2^(R-1-L) >= rem > 2^(R-2-L)
    if (rem == 1) L = R - 2;                       // last layer has a single entry
    uint32_t off = (1u << (R - 1)) - (1u << (R - 1 - L));
    uint32_t h = t - off;
    uint32_t bits = R - 2 - L;
    uint32_t j = bit_reverse(h, bits);
    uint32_t init = subgroup_gen_index(R + 1 - L);
    uint32_t step = subgroup_gen_index(R - 1 - L);
    v[n++] = twiddle_point(tab, R, init + step * j).x;
  }
  batch_inverse8(v, iv, n);
  for (uint32_t k = 0; k < n; k++) { const uint32_t t = t0 + k * nthreads; xtw[t] = v[k].v << 1; ixtw[t] = iv[k].v << 1; }   // tables hold 2w (mul_tw2)
  if (t0 == 0) { xtw[total] = 0; ixtw[total] = 0; }   // the unused last entry (was a hipMemsetAsync each: four API calls at the very
                                                       // start of a proof, where the host's launch rate is the bound)
}
template <uint32_t TW_BATCH>
__global__ void __launch_bounds__(256) k_twiddles_y(uint32_t* __restrict__ ytw, uint32_t* __restrict__ iytw, uint32_t R, const uint32_t* __restrict__ tab) {
  const uint32_t nthreads = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  M31 v[TW_BATCH], iv[TW_BATCH];
  uint32_t n = 0;
  for (uint32_t k = 0; k < TW_BATCH; k++) {
    const uint32_t t = 1 + t0 + k * nthreads;  // t in [1, 2^R)
    if (t >= (1u << R)) break;
    uint32_t lv = 32 - __clz(t);  // offset 2^(lv-1) <= t < 2^lv
    uint32_t h = t - (1u << (lv - 1));
    uint32_t j = bit_reverse(h, lv - 1);
    uint32_t init = subgroup_gen_index(lv + 1);
    uint32_t step = (lv >= 2) ? subgroup_gen_index(lv - 1) : 0;
    v[n++] = twiddle_point(tab, R, init + step * j).y;
  }
  batch_inverse8(v, iv, n);
  for (uint32_t k = 0; k < n; k++) { const uint32_t t = 1 + t0 + k * nthreads; ytw[t] = v[k].v << 1; iytw[t] = iv[k].v << 1; }
  if (t0 == 0) { ytw[0] = 0; iytw[0] = 0; }   // the unused first entry
}

// ---------------------------------------------------------------- butterfly passes
// One launch applies butterfly layers [lo, hi) of a size-2^n transform to every column.
// Tile = 2^W values of index bits [lo,hi)  x  2^M consecutive low indices (M = 0 when lo == 0).
// INVERSE: ibutterfly (a+b, (a-b)*itw), layers ascending.  Forward: (a+b*tw, a-b*tw), descending.
// in_len: logical input length; reads at index >= in_len return 0 (zero-extension => LDE).
// (FftPassArgs lives in fft_pass.hpp; this generic LDS-sweep kernel serves tiles smaller than 2^11,
// the register-blocked radix-8 kernel in kernels_fft.hip serves full 2^11 tiles.)
template <bool INVERSE>
__global__ void __launch_bounds__(256) k_fft_pass(FftPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
  const uint32_t W = a.hi - a.lo;
  const uint32_t M = a.M;
  const uint32_t tile_log = W + M;
  const uint32_t tile_sz = 1u << tile_log;
  const uint32_t* src = a.src[blockIdx.y];
  uint32_t* dst = a.dst[blockIdx.y];
  // decompose tile id into (high bits above hi, fixed low bits [M, lo))
  const uint32_t low_fixed_bits = a.lo - M;  // 0 when lo == 0 (then M == 0)
  const uint32_t tid = blockIdx.x;
  const uint32_t lowf = tid & ((1u << low_fixed_bits) - 1);
  const uint32_t high = tid >> low_fixed_bits;
  const uint32_t base = (high << a.hi) | (lowf << M);
  // global index of tile element (mid, l):  base | mid << lo | l
  for (uint32_t e = threadIdx.x; e < tile_sz; e += blockDim.x) {
    uint32_t l = e & ((1u << M) - 1), mid = e >> M;
    uint32_t g = base | (mid << a.lo) | l;
    tile[e] = (g < a.in_len) ? src[g] : 0u;
  }
  __syncthreads();
  const uint32_t nbf = tile_sz >> 1;
  for (uint32_t k = 0; k < W; k++) {
    const uint32_t j = INVERSE ? k : (W - 1 - k);  // bit of `mid` being paired
    const uint32_t layer = a.lo + j;
    const uint32_t s = j + M;                      // LDS stride log
    for (uint32_t b = threadIdx.x; b < nbf; b += blockDim.x) {
      uint32_t e0 = ((b >> s) << (s + 1)) | (b & ((1u << s) - 1));
      uint32_t e1 = e0 | (1u << s);
      uint32_t l = e0 & ((1u << M) - 1), mid = e0 >> M;
      uint32_t g0 = base | (mid << a.lo) | l;
      uint32_t h = g0 >> (layer + 1);
      uint32_t tw;
      if (layer == 0) {
        tw = a.ytw[(1u << (a.n - 1)) + h];
      } else {
        uint32_t L = a.R - a.n + layer - 1;
        tw = a.xtw[(1u << (a.R - 1)) - (1u << (a.R - 1 - L)) + h];
      }
      M31 x(tile[e0]), y(tile[e1]);
      if (INVERSE) {
        tile[e0] = (x + y).v;
        tile[e1] = mul_tw2(x - y, tw).v;
      } else {
        M31 yt = mul_tw2(y, tw);
        tile[e0] = (x + yt).v;
        tile[e1] = (x - yt).v;
      }
    }
    __syncthreads();
  }
  const M31 sc(a.scale);
  for (uint32_t e = threadIdx.x; e < tile_sz; e += blockDim.x) {
    uint32_t l = e & ((1u << M) - 1), mid = e >> M;
    uint32_t g = base | (mid << a.lo) | l;
    uint32_t v = tile[e];
    if (a.scale != 1u) v = (sc * M31(v)).v;
    dst[g] = v;
  }
}

// ---------------------------------------------------------------- small columns: interpolate + extend in ONE launch
// A commitment carries dozens of small columns in several sizes (idle components: 2^4 rows; the tiny builtins; rc8 ...).
// As size groups they were two latency-bound launches each — ~25 launches of ~6 us per proof, in a row on the tree's
// stream.  Here one block owns one column whatever its size (<= SMALL_COMMIT_MAX_LOG): evaluations -> LDS -> n inverse
// layers -> coefficients (scaled by 2^-n) -> zero-extended -> n + blowup forward layers -> LDE, all sizes in one grid.
__global__ void __launch_bounds__(256) k_small_commit(const SmallCommitJob* __restrict__ jobs, TwiddleTables tw, uint32_t blowup) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
  const SmallCommitJob jb = jobs[blockIdx.x];
  const uint32_t n = jb.log_n, N = 1u << n, no = n + blowup, NO = 1u << no;
  const bool from_coeffs = jb.src == nullptr;   // src may alias coeffs (interpolation in place): it is read into LDS first
  for (uint32_t e = threadIdx.x; e < N; e += blockDim.x) tile[e] = (from_coeffs ? jb.coeffs : jb.src)[e];
  __syncthreads();
  if (!from_coeffs) {
    for (uint32_t layer = 0; layer < n; layer++) {
      for (uint32_t b = threadIdx.x; b < (N >> 1); b += blockDim.x) {
        const uint32_t e0 = ((b >> layer) << (layer + 1)) | (b & ((1u << layer) - 1)), e1 = e0 | (1u << layer);
        const uint32_t h = e0 >> (layer + 1);
        const uint32_t t = layer == 0 ? tw.iytw[(1u << (n - 1)) + h]
                                      : tw.ixtw[(1u << (tw.R - 1)) - (1u << (tw.R - 1 - (tw.R - n + layer - 1))) + h];
        const M31 x(tile[e0]), y(tile[e1]);
        tile[e0] = (x + y).v;
        tile[e1] = mul_tw2(x - y, t).v;
      }
      __syncthreads();
    }
    const M31 sc(jb.inv_n);
    for (uint32_t e = threadIdx.x; e < N; e += blockDim.x) {
      const uint32_t v = (sc * M31(tile[e])).v;
      tile[e] = v;
      jb.coeffs[e] = v;
    }
  }
  for (uint32_t e = N + threadIdx.x; e < NO; e += blockDim.x) tile[e] = 0u;
  __syncthreads();
  for (uint32_t k = 0; k < no; k++) {
    const uint32_t layer = no - 1 - k;
    for (uint32_t b = threadIdx.x; b < (NO >> 1); b += blockDim.x) {
      const uint32_t e0 = ((b >> layer) << (layer + 1)) | (b & ((1u << layer) - 1)), e1 = e0 | (1u << layer);
      const uint32_t h = e0 >> (layer + 1);
      const uint32_t t = layer == 0 ? tw.ytw[(1u << (no - 1)) + h]
                                    : tw.xtw[(1u << (tw.R - 1)) - (1u << (tw.R - 1 - (tw.R - no + layer - 1))) + h];
      const M31 x(tile[e0]), yt = mul_tw2(M31(tile[e1]), t);
      tile[e0] = (x + yt).v;
      tile[e1] = (x - yt).v;
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < NO; e += blockDim.x) jb.lde[e] = tile[e];
}

// ---------------------------------------------------------------- bit reversal (in place)
__global__ void k_bit_reverse(uint32_t* const* cols, uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  uint32_t j = bit_reverse(i, log_n);
  if (j > i) {
    uint32_t* c = cols[blockIdx.y];
    uint32_t a = c[i], b = c[j];
    c[i] = b;
    c[j] = a;
  }
}

// ---------------------------------------------------------------- eval_at_point
// value = sum_i coeff[i] * prod_{bits k of i} m_k,  m_0 = y, m_1 = x, m_k = pi^{k-1}(x).
// Split i = (hi, lo) with lo = LOW_BITS bits: the per-point table lowtab[lo] (QM31, built by
// k_point_table) is shared by every column of the batch; each block reduces one 2^LOW_BITS chunk
// and multiplies by hightab[hi].  partial[col][chunk] -> k_reduce_partials sums chunks.
constexpr uint32_t EAP_LOW_BITS = 12;

// tab[i] = prod_{bits k of i} maps[first_bit + k], i < 2^nbits.  The <= 32 QM31 factors travel in the
// kernel arguments (512 B), so sampling needs no host->device copy.
struct PointMaps { uint32_t w[32 * 4]; };
__global__ void k_point_table(PointMaps maps, uint32_t first_bit, uint32_t nbits, uint32_t* tab) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << nbits)) return;
  QM31 r(M31(1));
  for (uint32_t k = 0; k < nbits; k++)
    if ((i >> k) & 1u) r = r * QM31::from_u32(maps.w + 4 * (first_bit + k));
  r.to_u32(tab + 4 * i);
}

__global__ void __launch_bounds__(256) k_eval_at_point_partial(const uint32_t* const* coeffs, uint32_t log_n,
                                                               const uint32_t* lowtab, const uint32_t* hightab,
                                                               uint32_t* partial /*[ncols][nchunks][4]*/) {
  const uint32_t low_bits = log_n < EAP_LOW_BITS ? log_n : EAP_LOW_BITS;
  const uint32_t chunk = blockIdx.x, nchunks = gridDim.x;
  const uint32_t* c = coeffs[blockIdx.y] + ((size_t)chunk << low_bits);
  QM31 acc;
  for (uint32_t i = threadIdx.x; i < (1u << low_bits); i += blockDim.x) {
    QM31 t = QM31::from_u32(lowtab + 4 * i);
    acc += t * M31(c[i]);
  }
  acc = block_reduce_qm31(acc);
  if (threadIdx.x == 0) {
    acc = acc * QM31::from_u32(hightab + 4 * chunk);
    acc.to_u32(partial + 4 * ((size_t)blockIdx.y * nchunks + chunk));
  }
}
__global__ void __launch_bounds__(256) k_reduce_partials(const uint32_t* partial, uint32_t nchunks, uint32_t* out) {
  QM31 acc;
  const uint32_t* p = partial + 4 * (size_t)blockIdx.x * nchunks;
  for (uint32_t i = threadIdx.x; i < nchunks; i += blockDim.x) acc += QM31::from_u32(p + 4 * i);
  acc = block_reduce_qm31(acc);
  if (threadIdx.x == 0) acc.to_u32(out + 4 * blockIdx.x);
}


// ---- all sampling jobs of a proof in three launches ------------------------------------------------------
// A proof samples ~20 (log size, point) groups; as separate launches (4 each) the small groups are pure
// launch latency.  Job descriptors + the per-bit point factors travel in ONE upload; blocks find their job
// with a scalar scan over the (<= 64) block-range prefix.
// value = sum_i coeff[i] * low[i_lo] * high[i_hi].  A thread owns 4 low indices and walks EAP2_GROUP chunks
// (= high indices): coeff * high[chunk] — a wave-uniform QM31 from scalar loads — is accumulated as raw 64-bit
// products, and the low factor is applied ONCE per thread at the end.  The inner loop is one coalesced
// coefficient load + 4 multiply-adds: no per-coefficient table traffic (the first version re-read a 16-byte
// table entry from L2 for every 4-byte coefficient and ran at ~2 TB/s).
constexpr uint32_t EAP2_LOW_BITS = 10, EAP2_GROUP = 32;
struct EapJobDev {
  uint32_t log_n, ncols;
  const uint32_t* const* coeffs;
  uint32_t* out;                 // 4 * ncols words
  uint32_t* out_host;            // the same values into pinned host memory (or null)
  uint32_t low_off, high_off;    // word offsets of the two tables in the scratch buffer
  uint32_t partial_off;          // word offset of partial[ncols][ngroups][4]
  uint32_t block_begin;          // first block of this job in k_eval_partial_multi
  uint32_t col_begin;            // first (job, column) index of this job in k_reduce_partials_multi
  uint32_t ngroups;              // chunk groups per column
  uint32_t shift_x, shift_y, has_shift;   // device-side OODS point: the job samples at oods + (shift_x, shift_y) (M31 point)
  uint32_t maps[32 * 4];         // QM31 factor of index bit k
};
// OODS point on the device (CirclePoint::get_random_point: t = draw_felt(); x = (1 - t^2) / (1 + t^2), y = 2t / (1 + t^2)) from
// the felt the device-side transcript step left in HBM, shifted per job, expanded into the per-bit factors [y, x, pi(x), ...]:
// the evaluation kernels start right behind the composition tree, without waiting for the host to see root 3.
__global__ void __launch_bounds__(64) k_oods_maps(EapJobDev* __restrict__ jobs, uint32_t nj, const uint32_t* __restrict__ t4) {
  const uint32_t k = threadIdx.x;
  if (k >= nj) return;
  EapJobDev& jb = jobs[k];
  const QM31 t = QM31::from_u32(t4), t2 = t * t, iv = inv(t2 + M31(1));
  CPoint<QM31> p{(QM31(M31(1)) - t2) * iv, (t + t) * iv};
  if (jb.has_shift) p = cadd(p, CPoint<QM31>{QM31(M31(jb.shift_x)), QM31(M31(jb.shift_y))});
  p.y.to_u32(jb.maps);
  QM31 x = p.x;
  for (uint32_t b = 1; b < jb.log_n; b++) { x.to_u32(jb.maps + 4 * b); x = double_x(x); }
}
__global__ void __launch_bounds__(256) k_point_tables_multi(const EapJobDev* __restrict__ jobs, uint32_t* __restrict__ scratch) {
  const EapJobDev& jb = jobs[blockIdx.y];
  const uint32_t low = jb.log_n < EAP2_LOW_BITS ? jb.log_n : EAP2_LOW_BITS;
  const uint32_t high = jb.log_n - low;
  const bool hi_tab = blockIdx.z == 1;
  const uint32_t nbits = hi_tab ? high : low, first_bit = hi_tab ? low : 0;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << nbits)) return;
  QM31 r(M31(1));
  for (uint32_t k = 0; k < nbits; k++)
    if ((i >> k) & 1u) r = r * QM31::from_u32(jb.maps + 4 * (first_bit + k));
  r.to_u32(scratch + (hi_tab ? jb.high_off : jb.low_off) + 4 * i);
}
__device__ __forceinline__ unsigned long long fold31x2(unsigned long long x) { return m31_fold_lazy(x); }
// first block / first column of every job, passed BY VALUE (kernel arguments sit in scalar registers): finding a
// block's job by walking the 560-byte descriptors in memory cost ~20 dependent scalar loads per block
struct EapStarts { uint32_t v[64]; };
__global__ void __launch_bounds__(256) k_eval_partial_multi(const EapJobDev* __restrict__ jobs, uint32_t njobs, EapStarts starts,
                                                            uint32_t* __restrict__ scratch) {
  uint32_t j = 0;
#pragma unroll 8
  for (uint32_t k = 1; k < 64; k++) j += (k < njobs && starts.v[k] <= blockIdx.x) ? 1u : 0u;
  const EapJobDev& jb = jobs[j];
  const uint32_t low_bits = jb.log_n < EAP2_LOW_BITS ? jb.log_n : EAP2_LOW_BITS;
  const uint32_t nchunks = 1u << (jb.log_n - low_bits);
  const uint32_t b = blockIdx.x - jb.block_begin;
  const uint32_t col = b / jb.ngroups, grp = b - col * jb.ngroups;
  const uint32_t c0 = grp * EAP2_GROUP, c1 = min(c0 + EAP2_GROUP, nchunks);
  const cm_gptr coef = CM_GCOL(jb.coeffs[col]);   // HBM column reached through the job table: global, not flat, loads
  const uint32_t* __restrict__ high = scratch + jb.high_off;
  constexpr uint32_t PER = (1u << EAP2_LOW_BITS) / 256;  // low indices per thread
  unsigned long long q[PER][4];
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) q[k][0] = q[k][1] = q[k][2] = q[k][3] = 0;
  const bool full_low = low_bits == EAP2_LOW_BITS;
  for (uint32_t cb = c0; cb < c1; cb += 4) {
    if (full_low && cb + 4 <= c1) {
      // common case: no bounds tests, so the 16 coefficient loads of this step issue back to back
      uint32_t x[4][PER];
#pragma unroll
      for (uint32_t cc = 0; cc < 4; cc++)
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) x[cc][k] = coef[((size_t)(cb + cc) << EAP2_LOW_BITS) + threadIdx.x + k * 256];
#pragma unroll
      for (uint32_t cc = 0; cc < 4; cc++) {
        const uint32_t c = cb + cc;
        const uint32_t h0 = high[4 * c], h1 = high[4 * c + 1], h2 = high[4 * c + 2], h3 = high[4 * c + 3];  // wave-uniform
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
          const unsigned long long xv = x[cc][k];
          q[k][0] += xv * h0; q[k][1] += xv * h1; q[k][2] += xv * h2; q[k][3] += xv * h3;
        }
      }
    } else {
#pragma unroll
      for (uint32_t cc = 0; cc < 4; cc++) {
        const uint32_t c = cb + cc;
        if (c < c1) {
          const uint32_t h0 = high[4 * c], h1 = high[4 * c + 1], h2 = high[4 * c + 2], h3 = high[4 * c + 3];
#pragma unroll
          for (uint32_t k = 0; k < PER; k++) {
            const uint32_t i = threadIdx.x + k * 256;
            if (i < (1u << low_bits)) {
              const unsigned long long xv = coef[((size_t)c << low_bits) + i];
              q[k][0] += xv * h0; q[k][1] += xv * h1; q[k][2] += xv * h2; q[k][3] += xv * h3;
            }
          }
        }
      }
    }
    // 4 raw products per coordinate fit a u64 on top of a folded remainder (4 * (2^31-1)^2 + 3 * 2^32 < 2^64)
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) { q[k][0] = fold31x2(q[k][0]); q[k][1] = fold31x2(q[k][1]); q[k][2] = fold31x2(q[k][2]); q[k][3] = fold31x2(q[k][3]); }
  }
  QM31 acc;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t i = threadIdx.x + k * 256;
    if (i < (1u << low_bits)) {
      QM31 s(M31::reduce(q[k][0]), M31::reduce(q[k][1]), M31::reduce(q[k][2]), M31::reduce(q[k][3]));
      acc += s * QM31::from_u32(scratch + jb.low_off + 4 * i);
    }
  }
  acc = block_reduce_qm31(acc);
  if (threadIdx.x == 0) acc.to_u32(scratch + jb.partial_off + 4 * ((size_t)col * jb.ngroups + grp));
}
__global__ void __launch_bounds__(256) k_reduce_partials_multi(const EapJobDev* __restrict__ jobs, uint32_t njobs, EapStarts starts,
                                                               const uint32_t* __restrict__ scratch) {
  uint32_t j = 0;
#pragma unroll 8
  for (uint32_t k = 1; k < 64; k++) j += (k < njobs && starts.v[k] <= blockIdx.x) ? 1u : 0u;
  const EapJobDev& jb = jobs[j];
  const uint32_t col = blockIdx.x - jb.col_begin;
  const uint32_t* p = scratch + jb.partial_off + 4 * (size_t)col * jb.ngroups;
  QM31 acc;
  for (uint32_t i = threadIdx.x; i < jb.ngroups; i += blockDim.x) acc += QM31::from_u32(p + 4 * i);
  acc = block_reduce_qm31(acc);
  if (threadIdx.x == 0) {
    acc.to_u32(jb.out + 4 * col);
    if (jb.out_host) *reinterpret_cast<uint4*>(jb.out_host + 4 * col) = make_uint4(acc.a.a.v, acc.a.b.v, acc.b.a.v, acc.b.b.v);
  }
}

// ================================================================= host wrappers
// fills tables whose buffers the caller allocated (nx = 2^(R-1) words for xtw / ixtw, ny = 2^R for ytw / iytw,
// twiddles_scratch_words(R) for scratch)
size_t twiddles_scratch_words(uint32_t R) { return 2 * (((size_t)1 << TW_LO) + ((size_t)1 << (R + 1 > TW_LO ? R + 1 - TW_LO : 0))); }
void twiddles_build(const Twiddles& t, hipStream_t st) {
  const uint32_t R = t.R;
  CM_CHECK(R >= 2 && R <= 28, "twiddles: log size out of range (columns are limited to 2^26 rows)");
  CM_CHECK(t.scratch, "twiddles: no scratch buffer");
  size_t nx = (size_t)1 << (R - 1), ny = (size_t)1 << R;
  const uint32_t n_tab = (uint32_t)(twiddles_scratch_words(R) / 2);
  hipLaunchKernelGGL(k_twiddle_point_tables, dim3((n_tab + 255) / 256), dim3(256), 0, st, R, t.scratch);
  const uint32_t tb = (uint32_t)tune(T_TW_BATCH);
  auto blocks = [tb](size_t n) { return dim3((uint32_t)((n + 256 * tb - 1) / (256 * tb))); };
#define CM_TW(B)                                                                                            \
  hipLaunchKernelGGL(k_twiddles_x<B>, blocks(nx), dim3(256), 0, st, t.xtw, t.ixtw, R, t.scratch);           \
  hipLaunchKernelGGL(k_twiddles_y<B>, blocks(ny), dim3(256), 0, st, t.ytw, t.iytw, R, t.scratch);
  if (tb == 2) { CM_TW(2) } else if (tb == 4) { CM_TW(4) } else if (tb == 16) { CM_TW(16) } else { CM_CHECK(tb == 8, "tw_batch: 2, 4, 8 or 16"); CM_TW(8) }
#undef CM_TW
  CM_HIP(hipGetLastError());
}
Twiddles* twiddles_create(uint32_t R, hipStream_t st) {
  CM_CHECK(R >= 2 && R <= 28, "twiddles: log size out of range (columns are limited to 2^26 rows)");
  Twiddles* t = new Twiddles();
  t->R = R;
  size_t nx = (size_t)1 << (R - 1), ny = (size_t)1 << R;
  CM_HIP(hipMalloc(&t->xtw, nx * 4));
  CM_HIP(hipMalloc(&t->ixtw, nx * 4));
  CM_HIP(hipMalloc(&t->ytw, ny * 4));
  CM_HIP(hipMalloc(&t->iytw, ny * 4));
  CM_HIP(hipMalloc(&t->scratch, twiddles_scratch_words(R) * 4));
  twiddles_build(*t, st);
  return t;
}
void twiddles_destroy(Twiddles* t) {
  if (!t) return;
  (void)hipFree(t->xtw); (void)hipFree(t->ixtw); (void)hipFree(t->ytw); (void)hipFree(t->iytw); (void)hipFree(t->scratch);
  delete t;
}

constexpr uint32_t FFT_TILE_LOG = 11;        // tile of the generic LDS-sweep kernel (transforms below 2^11)
constexpr uint32_t FFT_CONTIG_LOG = 12;      // layers of the contiguous pass: three radix-16 rounds (13 = 5 + 3 + 5 when that saves a sweep)
constexpr uint32_t FFT_MAX_STRIDED_W = 9;    // a strided pass carries up to 9 layers (2^9 x 2^5 tile: 128-B runs, 64 KiB of LDS)

// Plan: layers [0, k0) in one contiguous pass, the rest in as few strided passes of <= 9 layers as possible, balanced.
// Measured on 64 columns x 2^22 (tools/fft_lab.hip, us forward / inverse): contiguous 11 layers 616 / 681, 12 layers
// 544 / 598, 13 layers 604 / 707; strided (2^14 tile) 6 layers 371, 7: 407, 8: 405 / 423, 9: 472 / 496, 10 (64-B runs):
// 618 / 872.  So k0 = 12, except that a transform with exactly 22 layers takes 13 + 9 instead of 12 + 10.
// A 2^21 transform is 12 + 9, a 2^22 one 13 + 9: two sweeps over HBM where the radix-8 plan (11 + 5 + 5 / 11 + 6 + 5) made
// three.  CM_FFT_OLD_PLAN=1 restores the 11-layer contiguous pass and <= 7-layer strided passes (A/B).
static void plan_passes(uint32_t n, std::vector<std::pair<uint32_t, uint32_t>>& passes) {
  static const bool old_plan = getenv("CM_FFT_OLD_PLAN") != nullptr;
  uint32_t contig = old_plan ? 11u : FFT_CONTIG_LOG;
  const uint32_t max_w = old_plan ? 7u : FFT_MAX_STRIDED_W;
  if (!old_plan && n == contig + max_w + 1) contig++;
  uint32_t k0 = n < contig ? n : contig;
  passes.push_back({0, k0});
  uint32_t rest = n - k0;
  if (rest == 0) return;
  uint32_t np = (rest + max_w - 1) / max_w;
  uint32_t lo = k0;
  for (uint32_t i = 0; i < np; i++) {
    uint32_t w = (rest - (lo - k0) + (np - i) - 1) / (np - i);
    passes.push_back({lo, lo + w});
    lo += w;
  }
}
template <bool INV>
static void launch_pass(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n, uint32_t lo,
                        uint32_t hi, uint32_t in_len, uint32_t scale, const Twiddles& tw, hipStream_t st) {
  FftPassArgs a;
  a.src = d_src; a.dst = d_dst;
  a.xtw = INV ? tw.ixtw : tw.xtw;
  a.ytw = INV ? tw.iytw : tw.ytw;
  a.R = tw.R; a.n = n; a.lo = lo; a.hi = hi;
  uint32_t W = hi - lo;
  const uint32_t rb = fft_pass_rb_tile_log(W, lo);
  uint32_t M = 0;
  if (rb) M = rb - W;
  else if (lo > 0) { M = FFT_TILE_LOG - W; if (M > lo) M = lo; }
  a.M = M; a.in_len = in_len; a.scale = scale;
  uint32_t tile_log = W + M;
  uint32_t ntiles = 1u << (n - tile_log);
  size_t lds = (size_t)4 << tile_log;
  // algorithmic bytes of one pass: every element written once; read once unless it is implicit zero padding
  KProfScope kp(INV ? "k_fft_pass<ifft>" : "k_fft_pass<fft>",
                4.0 * ncols * ((double)(1u << n) + (double)(in_len < (1u << n) ? in_len : (1u << n))), st,
                /* butterflies */ (double)ncols * (double)(1u << (n - 1)) * (double)(hi - lo));
  if (rb) launch_fft_pass_rb(INV, a, rb, ntiles, ncols, st);
  else hipLaunchKernelGGL(k_fft_pass<INV>, dim3(ntiles, ncols), dim3(256), lds, st, a);
}
void interpolate_oop(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n, const Twiddles& tw,
                     hipStream_t st) {
  CM_CHECK(n >= 1 && n <= tw.R, "interpolate: log size exceeds twiddle table");
  if (ncols == 0) return;
  std::vector<std::pair<uint32_t, uint32_t>> passes;
  plan_passes(n, passes);
  uint32_t inv_n = inv(M31::from_u32(1u << n)).v;
  for (size_t i = 0; i < passes.size(); i++) {
    bool last = i + 1 == passes.size();
    launch_pass<true>(i == 0 ? d_src : (const uint32_t* const*)d_dst, d_dst, ncols, n, passes[i].first, passes[i].second,
                      1u << n, last ? inv_n : 1u, tw, st);
  }
  CM_HIP(hipGetLastError());
}
void interpolate(uint32_t* const* d_cols, uint32_t ncols, uint32_t n, const Twiddles& tw, hipStream_t st) {
  interpolate_oop(d_cols, d_cols, ncols, n, tw, st);
}
void evaluate(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n_in, uint32_t n_out,
              const Twiddles& tw, hipStream_t st) {
  CM_CHECK(n_out >= n_in && n_out <= tw.R && n_out >= 1, "evaluate: bad log sizes");
  if (ncols == 0) return;
  std::vector<std::pair<uint32_t, uint32_t>> passes;
  plan_passes(n_out, passes);
  for (size_t k = passes.size(); k-- > 0;) {
    bool first = k + 1 == passes.size();
    launch_pass<false>(first ? d_src : (const uint32_t* const*)d_dst, d_dst, ncols, n_out, passes[k].first,
                       passes[k].second, first ? (1u << n_in) : (1u << n_out), 1u, tw, st);
  }
  CM_HIP(hipGetLastError());
}
// interpolate (2^n evaluations -> coefficients) + evaluate on the domain of 2^(n+1): tree_builder.extend_evals with
// log_blowup_factor = 1.  For 2^18 .. 2^21 rows the last pass of the inverse transform and the first pass of the forward one are
// ONE sweep (k_fft_fused_rb, kernels_fft.hip): [0,12) inverse | [12,n) inverse + top layer + [12,n) forward of both halves |
// [0,12) forward.  d_src may equal d_coeffs (in place).  CM_FFT_FUSED=0: the two transforms one after the other (A/B).
bool fft_fused_serves(uint32_t W);
void launch_fft_fused_rb(const FftPassArgs& inv, const FftPassArgs& fwd, uint32_t ntiles, uint32_t ncols, hipStream_t st);
void interpolate_extend(const uint32_t* const* d_src, uint32_t* const* d_coeffs, uint32_t* const* d_lde, uint32_t ncols, uint32_t n,
                        const Twiddles& tw, hipStream_t st) {
  const bool fused_on = tune(T_FFT_FUSED) != 0;
  CM_CHECK(n >= 1 && n + 1 <= tw.R, "interpolate_extend: log size exceeds twiddle table");
  if (ncols == 0) return;
  const uint32_t W = n > FFT_CONTIG_LOG ? n - FFT_CONTIG_LOG : 0;
  std::vector<std::pair<uint32_t, uint32_t>> passes;
  plan_passes(n, passes);
  const bool fused = fused_on && fft_fused_serves(W) && passes.size() == 2 && passes[0].second == FFT_CONTIG_LOG &&
                     fft_pass_rb_tile_log(W, FFT_CONTIG_LOG) == 14;
  if (!fused) {
    interpolate_oop(d_src, d_coeffs, ncols, n, tw, st);
    evaluate((const uint32_t* const*)d_coeffs, d_lde, ncols, n, n + 1, tw, st);
    return;
  }
  launch_pass<true>(d_src, d_coeffs, ncols, n, 0, FFT_CONTIG_LOG, 1u << n, 1u, tw, st);
  {
    FftPassArgs inv_a, fwd_a;
    inv_a.src = (const uint32_t* const*)d_coeffs; inv_a.dst = d_coeffs;
    inv_a.xtw = tw.ixtw; inv_a.ytw = tw.iytw; inv_a.R = tw.R; inv_a.n = n; inv_a.lo = FFT_CONTIG_LOG; inv_a.hi = n;
    inv_a.M = 14 - W; inv_a.in_len = 1u << n; inv_a.scale = inv(M31::from_u32(1u << n)).v;
    fwd_a.src = (const uint32_t* const*)d_coeffs; fwd_a.dst = d_lde;   // (src is not read: the coefficients arrive in registers)
    fwd_a.xtw = tw.xtw; fwd_a.ytw = tw.ytw; fwd_a.R = tw.R; fwd_a.n = n + 1; fwd_a.lo = FFT_CONTIG_LOG; fwd_a.hi = n;
    fwd_a.M = 14 - W; fwd_a.in_len = 2u << n; fwd_a.scale = 1u;
    const double N = (double)(1u << n);
    KProfScope kp("k_fft_fused", 4.0 * ncols * 4.0 * N, st, /* butterflies */ (double)ncols * 1.5 * N * (double)W);
    launch_fft_fused_rb(inv_a, fwd_a, 1u << (n - 14), ncols, st);
  }
  launch_pass<false>((const uint32_t* const*)d_lde, d_lde, ncols, n + 1, 0, FFT_CONTIG_LOG, 2u << n, 1u, tw, st);
  CM_HIP(hipGetLastError());
}
void small_commit(const SmallCommitJob* d_jobs, uint32_t n_jobs, uint32_t max_log, uint32_t blowup, const Twiddles& tw, hipStream_t st) {
  if (!n_jobs) return;
  CM_CHECK(max_log >= 1 && max_log <= SMALL_COMMIT_MAX_LOG && max_log + blowup <= tw.R && max_log + blowup <= 14, "small_commit: bad sizes");
  TwiddleTables t{tw.R, tw.xtw, tw.ixtw, tw.ytw, tw.iytw};
  const size_t lds = (size_t)4 << (max_log + blowup);
  if (lds > 48 * 1024) {
    static const hipError_t once = hipFuncSetAttribute((const void*)k_small_commit, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)once;
  }
  KProfScope kp("k_small_commit", 0.0, st);
  hipLaunchKernelGGL(k_small_commit, dim3(n_jobs), dim3(256), lds, st, d_jobs, t, blowup);
  CM_HIP(hipGetLastError());
}
void bit_reverse_columns(uint32_t* const* d_cols, uint32_t ncols, uint32_t n, hipStream_t st) {
  if (ncols == 0) return;
  uint32_t N = 1u << n;
  hipLaunchKernelGGL(k_bit_reverse, dim3((N + 255) / 256, ncols), dim3(256), 0, st, d_cols, n);
  CM_HIP(hipGetLastError());
}

size_t eval_at_point_scratch_words(uint32_t ncols, uint32_t n) {
  uint32_t low = n < EAP_LOW_BITS ? n : EAP_LOW_BITS;
  uint32_t high = n - low;
  return 32 * 4 + 4 * ((size_t)1 << low) + 4 * ((size_t)1 << high) + 4 * (size_t)ncols * ((size_t)1 << high);
}
void eval_at_point_batch(const uint32_t* const* d_coeffs, uint32_t ncols, uint32_t n, const QM31& px, const QM31& py,
                         uint32_t* d_scratch, uint32_t* d_out, hipStream_t st) {
  if (ncols == 0) return;
  uint32_t low = n < EAP_LOW_BITS ? n : EAP_LOW_BITS;
  uint32_t high = n - low;
  // maps[k]: factor of index bit k
  PointMaps maps;
  for (uint32_t k = 0; k < 32 * 4; k++) maps.w[k] = 0;
  py.to_u32(maps.w);
  QM31 x = px;
  for (uint32_t k = 1; k < n; k++) { x.to_u32(maps.w + 4 * k); x = double_x(x); }
  uint32_t* d_low = d_scratch + 32 * 4;
  uint32_t* d_high = d_low + 4 * ((size_t)1 << low);
  uint32_t* d_partial = d_high + 4 * ((size_t)1 << high);
  hipLaunchKernelGGL(k_point_table, dim3(((1u << low) + 255) / 256), dim3(256), 0, st, maps, 0u, low, d_low);
  hipLaunchKernelGGL(k_point_table, dim3(((1u << high) + 255) / 256), dim3(256), 0, st, maps, low, high, d_high);
  uint32_t nchunks = 1u << high;
  KProfScope kp("k_eval_at_point", 4.0 * ncols * (double)(1u << n), st);
  hipLaunchKernelGGL(k_eval_at_point_partial, dim3(nchunks, ncols), dim3(256), 0, st, d_coeffs, n, d_low, d_high,
                     d_partial);
  hipLaunchKernelGGL(k_reduce_partials, dim3(ncols), dim3(256), 0, st, d_partial, nchunks, d_out);
  CM_HIP(hipGetLastError());
}

void eval_at_point_multi(const std::vector<EapJob>& jobs, hipStream_t st, const uint32_t* d_oods_t) {
  if (jobs.empty()) return;

  std::vector<EapJobDev> dj(jobs.size());
  size_t words = 0;
  uint32_t blocks = 0, cols = 0, max_tab_blocks = 1;
  double bytes = 0;
  for (size_t k = 0; k < jobs.size(); k++) {
    const EapJob& j = jobs[k];
    CM_CHECK(j.ncols > 0 && j.log_n < 32, "eval_at_point_multi: bad job");
    EapJobDev& d = dj[k];
    memset(&d, 0, sizeof(d));
    const uint32_t low = j.log_n < EAP2_LOW_BITS ? j.log_n : EAP2_LOW_BITS, high = j.log_n - low;
    const uint32_t ngroups = ((1u << high) + EAP2_GROUP - 1) / EAP2_GROUP;
    d.log_n = j.log_n; d.ncols = j.ncols; d.coeffs = j.d_coeffs; d.out = j.d_out; d.out_host = j.h_out; d.ngroups = ngroups;
    d.low_off = (uint32_t)words; words += (size_t)4 << low;
    d.high_off = (uint32_t)words; words += (size_t)4 << high;
    d.partial_off = (uint32_t)words; words += (size_t)4 * j.ncols * ngroups;
    CM_CHECK(words < ((size_t)1 << 32), "eval_at_point_multi: scratch too large");
    d.block_begin = blocks; blocks += j.ncols * ngroups;
    d.col_begin = cols; cols += j.ncols;
    if (d_oods_t) {   // the point comes from the device (k_oods_maps): px / py carry the M31 shift of the job, or nothing
      d.has_shift = j.has_shift ? 1u : 0u;
      d.shift_x = j.shift_x; d.shift_y = j.shift_y;
    } else {
      j.py.to_u32(d.maps);
      QM31 x = j.px;
      for (uint32_t b = 1; b < j.log_n; b++) { x.to_u32(d.maps + 4 * b); x = double_x(x); }
    }
    max_tab_blocks = std::max(max_tab_blocks, ((1u << std::max(low, high)) + 255) / 256);
    bytes += 4.0 * j.ncols * (double)((size_t)1 << j.log_n);
  }
  DevBuf d_jobs = upload(dj, st), scratch(words * 4);
  const EapJobDev* djp = d_jobs.as<EapJobDev>();
  const uint32_t nj = (uint32_t)jobs.size();
  CM_CHECK(nj <= 64, "eval_at_point_multi: more than 64 sampling groups");
  KProfScope kp("k_eval_at_point", bytes, st);
  if (d_oods_t) hipLaunchKernelGGL(k_oods_maps, dim3(1), dim3(64), 0, st, d_jobs.as<EapJobDev>(), nj, d_oods_t);
  hipLaunchKernelGGL(k_point_tables_multi, dim3(max_tab_blocks, nj, 2), dim3(256), 0, st, djp, scratch.u32());
  CM_CHECK(nj <= 64, "eval_at_point_multi: more than 64 sampling groups");
  EapStarts bstart, cstart;
  for (uint32_t k = 0; k < 64; k++) { bstart.v[k] = k < nj ? dj[k].block_begin : 0xffffffffu; cstart.v[k] = k < nj ? dj[k].col_begin : 0xffffffffu; }
  hipLaunchKernelGGL(k_eval_partial_multi, dim3(blocks), dim3(256), 0, st, djp, nj, bstart, scratch.u32());
  hipLaunchKernelGGL(k_reduce_partials_multi, dim3(cols), dim3(256), 0, st, djp, nj, cstart, scratch.u32());
  CM_HIP(hipGetLastError());
}

}  // namespace cm
