// Kernels of the device-side proof tail (tail_device.hpp): last FRI layer -> proof of work -> query tables -> decommitment
// gathers.  Reference: the tail of stwo `prove` (crates/prover/src/prover.rs:131): FriProver::commit's last layer,
// GrindOps::grind, Queries::generate, FriProver::decommit, CommitmentSchemeProver::prove_values -> MerkleProver::decommit.
#include <string.h>
#include <algorithm>
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "kprof.hpp"
#include "blake2s_dev.hpp"
#include "framing.hpp"
#include "tail_device.hpp"

namespace cm {

// ---- K1: the last layer ------------------------------------------------------------------------------------------------
// LinePoly of the last layer (in-place line IFFT over the 2^log_n evaluations in bit-reversed order, scaled by 1 / n, the
// first 2^log_keep coefficients kept in LinePoly's bit-reversed order — the host twin is FriPhase::commit_finish), the degree
// check, channel.mix_felts(coefficients).  Also hands the challenges, roots and evaluations of the commit phase to the host
// (pinned memory) and arms the nonce cell of the proof of work.
__global__ void __launch_bounds__(256) k_tail_last(TailLastArgs a) {
  __shared__ uint32_t v[4][TAIL_MAX_LAST];
  __shared__ uint32_t msg[8 + 4 * TAIL_MAX_LAST + 16];
  __shared__ uint32_t s_bad;
  const uint32_t tid = threadIdx.x, n = 1u << a.log_n, keep = 1u << a.log_keep;
  for (uint32_t i = tid; i < a.n_ar_words; i += 256) a.h_ar[i] = a.d_ar[i];
  for (uint32_t i = tid; i < 4 * n; i += 256) {
    const uint32_t c = i >> a.log_n, j = i & (n - 1);
    const uint32_t w = a.last[c][j];
    v[c][j] = w;
    a.h_last[i] = w;
  }
  if (tid == 0) { s_bad = 0; *a.nonce = ~0ull; }
  __syncthreads();
  for (uint32_t l = 0; l < a.log_n; l++) {
    const uint32_t stride = 1u << l, half = n >> 1;
    if (tid < 4 * half) {
      const uint32_t c = tid / half, bf = tid % half;
      const uint32_t h = bf >> l, k = bf & (stride - 1);
      const uint32_t i0 = (h << (l + 1)) + k, i1 = i0 + stride;
      const M31 x(a.xinv[(n - (n >> l)) + h]);
      const M31 p(v[c][i0]), q(v[c][i1]);
      v[c][i0] = (p + q).v;
      v[c][i1] = ((p - q) * x).v;
    }
    __syncthreads();
  }
  // coefficient of degree p sits at position p; scale, check the degree bound, order the kept ones
  for (uint32_t i = tid; i < 4 * n; i += 256) {
    const uint32_t c = i >> a.log_n, j = i & (n - 1);
    const uint32_t w = (M31(v[c][j]) * M31(a.ninv)).v;
    if (j >= keep) { if (w) s_bad = 1; }
    else msg[8 + 4 * (a.log_keep ? __brev(j) >> (32 - a.log_keep) : 0u) + c] = w;   // bit_reverse(j, log_keep)
  }
  if (tid < 8) msg[tid] = a.chan[tid];
  __syncthreads();
  if (tid == 0) {
    // mix_felts: digest = Blake2s256(digest || le32 words of the coefficients)
    const uint32_t words = 8 + 4 * keep, total = 4 * words, nblk = (total + 63) / 64;
    uint32_t h[8] = {0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    for (uint32_t b = 0; b < nblk; b++) {
      uint32_t m[16];
#pragma unroll
      for (int i = 0; i < 16; i++) { const uint32_t wi = 16 * b + i; m[i] = wi < words ? msg[wi] : 0u; }
      const bool lastb = b + 1 == nblk;
      b2s_compress(h, m, lastb ? total : 64 * (b + 1), lastb ? 0xFFFFFFFFu : 0u);
    }
    for (int i = 0; i < 8; i++) a.chan[i] = h[i];
    a.chan[8] = 0;
    a.hdr[6] = s_bad;
  }
  // "the pinned words of this kernel have been written": the host watches this word instead of an event behind the launch (an event
  // record is a barrier packet in front of the proof of work)
  __threadfence_system();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(a.hdr + TAIL_HDR_LAST_DONE, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- K2: proof of work ---------------------------------------------------------------------------------------------------
// GrindOps::grind: the smallest nonce whose mix_u64 hash ends in `bits` zero bits.  The grid covers `span` nonces per sweep
// (every thread one nonce) and sweeps on until a hit below the sweep's first nonce is recorded or `limit` is reached: the expected
// nonce (2^bits) falls into the first sweep or two, so the launch costs one or two compressions per lane, not 16x that.
template <bool U32S>
__global__ void __launch_bounds__(256) k_tail_grind(const uint32_t* __restrict__ chan, uint32_t bits, uint64_t span, uint64_t limit,
                                                    unsigned long long* result) {
  uint32_t dg[8], h[8];
#pragma unroll
  for (int i = 0; i < 8; i++) dg[i] = chan[i];
  for (uint64_t base = 0; base < limit; base += span) {
    if (base && __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base) return;
    const uint64_t nonce = base + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    mix_u64_dev<U32S>(dg, nonce, h);
    uint32_t tz;
    if (h[0]) tz = __ffs(h[0]) - 1;
    else if (h[1]) tz = 32 + __ffs(h[1]) - 1;
    else if (h[2]) tz = 64 + __ffs(h[2]) - 1;
    else if (h[3]) tz = 96 + __ffs(h[3]) - 1;
    else tz = 128;
    if (tz >= bits && nonce < limit) atomicMin(result, (unsigned long long)nonce);   // (limit < span only under the test cap)
  }
}

// ---- K3: queries and tables ------------------------------------------------------------------------------------------------
// first index i with S[i] >= key (S sorted, n elements in LDS)
__device__ __forceinline__ uint32_t lds_lower_bound(const uint32_t* S, uint32_t n, uint64_t key) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint64_t)S[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// is x a member of unique(S >> k)?
__device__ __forceinline__ bool lds_has(const uint32_t* S, uint32_t n, uint32_t x, uint32_t k) {
  const uint32_t i = lds_lower_bound(S, n, (uint64_t)x << k);
  return i < n && (S[i] >> k) == x;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d);
    if (lane >= (uint32_t)d) x += y;
  }
  return x;
}

// exclusive prefix sum over the 1024 threads of the block (two levels of wave scans, three barriers); s_w: 17 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t* s_w, uint32_t lane, uint32_t wave, uint32_t& total) {
  const uint32_t incl = wave_incl_scan(x, lane);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    const uint32_t v = lane < 16 ? s_w[lane] : 0u;
    const uint32_t vi = wave_incl_scan(v, lane);
    if (lane < 16) s_w[lane] = vi - v;
    if (lane == 15) s_w[16] = vi;
  }
  __syncthreads();
  const uint32_t r = s_w[wave] + incl - x;
  total = s_w[16];
  __syncthreads();
  return r;
}

template <bool U32S>
__global__ void __launch_bounds__(1024) k_tail_tables(TailTablesArgs a) {
  __shared__ uint32_t s_a[TAIL_MAX_QUERIES], s_b[TAIL_MAX_QUERIES], s_S[TAIL_MAX_QUERIES];
  __shared__ uint32_t s_cnt[3 * TAIL_MAX_SHIFTS];
  __shared__ uint32_t s_w[17];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t NP = a.nq_pad, nq = a.n_queries, L0 = a.log_domain;
  const unsigned long long nonce = *a.nonce;
  const bool fail = nonce == ~0ull;
  const uint32_t ndraw = (nq + 7) / 8;
  if (tid < 3 * TAIL_MAX_SHIFTS) s_cnt[tid] = 0;
  if (!fail && tid < ndraw) {
    // mix_u64(nonce), then draw_random_bytes with n_sent = tid: Blake2s256(digest || le32(n_sent) || 0^28 || 0x00)
    uint32_t dg[8], h[8], m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) dg[i] = a.chan[i];
    mix_u64_dev<U32S>(dg, nonce, h);
    uint32_t d[8] = {0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
#pragma unroll
    for (int i = 0; i < 8; i++) { m[i] = h[i]; m[8 + i] = 0; }
    m[8] = tid;
    b2s_compress(d, m, 64, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = 0;
    b2s_compress(d, m, 65, 0xFFFFFFFFu);
    const uint32_t mask = (1u << L0) - 1u;   // L0 < 32 (tail_tables)
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint32_t idx = tid * 8 + j; if (idx < nq) s_a[idx] = d[j] & mask; }
  } else if (tid >= 128) {  // the descriptors: pinned host memory -> device, 16 bytes per access, by the waves that do not hash
    const uint4* src = reinterpret_cast<const uint4*>(a.h_desc);
    uint4* dst = reinterpret_cast<uint4*>(a.d_desc);
    const uint32_t n16 = a.n_desc * (uint32_t)(sizeof(TailDesc) / 16);
    for (uint32_t i = tid - 128; i < n16; i += 1024 - 128) dst[i] = src[i];
  }
  __syncthreads();   // (the device copy of the channel is not advanced: nothing on the device reads it after this step)
  // sort by rank (ties by index): nq broadcast reads per thread instead of a barrier per bitonic step
  if (!fail && tid < nq) {
    const uint32_t v = s_a[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nq; j++) { const uint32_t w = s_a[j]; rank += (w < v || (w == v && j < tid)) ? 1u : 0u; }
    s_b[rank] = v;
  }
  __syncthreads();
  uint32_t ns;
  {  // unique
    const uint32_t flag = (!fail && tid < nq && (tid == 0 || s_b[tid] != s_b[tid - 1])) ? 1u : 0u;
    const uint32_t pos = block_excl_scan(flag, s_w, lane, wave, ns);
    if (flag) s_S[pos] = s_b[tid];
    __syncthreads();
  }
  for (uint32_t i = tid; i < ns; i += 1024) a.h_positions[i] = s_S[i];
  // the tables, one wave per shift
  uint32_t* const g_cnt = a.tab;
  uint32_t* const g_U = a.tab + 3 * TAIL_MAX_SHIFTS;
  uint32_t* const g_W = g_U + (size_t)TAIL_MAX_SHIFTS * NP;
  uint32_t* const g_F = g_W + (size_t)TAIL_MAX_SHIFTS * NP;
  for (uint32_t k = wave; k <= L0 && k < TAIL_MAX_SHIFTS; k += 16) {
    const uint32_t l = L0 - k;                       // the layer: 2^l nodes
    const bool exp_l = (a.qmask >> l) & 1u;          // (first FRI tree) columns at this layer: sibling pairs are opened
    const bool exp_c = l + 1 <= L0 && ((a.qmask >> (l + 1)) & 1u);
    uint32_t bU = 0, bW = 0, bF = 0;
    for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool valid = i < ns;
      const uint32_t u = valid ? s_S[i] >> k : 0u;
      const bool isU = valid && (i == 0 || (s_S[i - 1] >> k) != u);
      const uint32_t sib = u ^ 1u;
      const bool sib_here = isU && l >= 1 && lds_has(s_S, ns, sib, k);
      const bool mW = isU && l >= 1 && !sib_here;
      const unsigned long long mu = __ballot(isU), mw = __ballot(mW);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (isU) g_U[(size_t)k * NP + bU + __popcll(mu & below)] = u;
      if (mW) g_W[(size_t)k * NP + bW + __popcll(mw & below)] = sib;
      // F: the hashes of layer l + 1 the walk asks for at layer l
      uint32_t item[3], cF = 0;
      if (isU && k >= 1) {
        uint32_t own = 0xFFFFFFFFu;
        if (!exp_c) {
          if (!lds_has(s_S, ns, 2 * u, k - 1)) own = 2 * u;
          else if (!lds_has(s_S, ns, 2 * u + 1, k - 1)) own = 2 * u + 1;
        }
        const bool sib_only = exp_l && mW;
        if (sib_only && sib < u) { item[cF++] = 2 * sib; item[cF++] = 2 * sib + 1; }
        if (own != 0xFFFFFFFFu) item[cF++] = own;
        if (sib_only && sib > u) { item[cF++] = 2 * sib; item[cF++] = 2 * sib + 1; }
      }
      const uint32_t incl = wave_incl_scan(cF, lane);
      for (uint32_t t = 0; t < cF; t++) g_F[(size_t)k * 4 * NP + bF + incl - cF + t] = item[t];
      bU += (uint32_t)__popcll(mu);
      bW += (uint32_t)__popcll(mw);
      bF += __shfl(incl, 63);
    }
    if (lane == 0) {
      s_cnt[k] = bU; s_cnt[TAIL_MAX_SHIFTS + k] = bW; s_cnt[2 * TAIL_MAX_SHIFTS + k] = bF;
    }
  }
  __syncthreads();
  if (tid < 3 * TAIL_MAX_SHIFTS) g_cnt[tid] = s_cnt[tid];
  // first output word of every descriptor (exclusive scan of the piece sizes, 1024 descriptors at a time)
  uint32_t carry = 0;
  for (uint32_t d0 = 0; d0 < a.n_desc; d0 += 1024) {
    const uint32_t d = d0 + tid;
    uint32_t size = 0;
    if (d < a.n_desc) {
      const TailDesc& D = a.d_desc[d];   // (the block's own copy: visible after the barriers above)
      const uint32_t kind = D.kind, k = D.k;
      const uint32_t cnt = kind == TD_ROWS_U ? s_cnt[k] : kind == TD_HASH_F ? s_cnt[2 * TAIL_MAX_SHIFTS + k] : s_cnt[TAIL_MAX_SHIFTS + k];
      size = cnt * (kind == TD_ROWS_U ? D.width : kind == TD_COORDS_W ? 4u : 8u);
    }
    uint32_t total;
    const uint32_t excl = block_excl_scan(size, s_w, lane, wave, total);
    if (d < a.n_desc) a.d_off[d] = carry + excl;
    carry += total;
  }
  if (tid == 0) {
    a.hdr[0] = fail ? TAIL_NO_NONCE : TAIL_OK;
    a.hdr[1] = (uint32_t)nonce; a.hdr[2] = (uint32_t)(nonce >> 32);
    a.hdr[3] = ns;
    a.hdr[4] = carry;
  }
  __threadfence_system();   // header, positions: visible to the host before the word it watches
  __syncthreads();
  if (tid == 0) __hip_atomic_store(a.hdr + TAIL_HDR_TABLES_DONE, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- K4: the gathers ------------------------------------------------------------------------------------------------------
// Pieces [0, n_small): hashes (8 words per node) or QM31 evaluations (4 words per row) at the nodes of a W / F list, `small_y`
// blocks per piece; pieces [n_small, n_desc): one row of a run of columns per block.  `out` is pinned host memory.
__global__ void __launch_bounds__(256) k_tail_gather(const TailDesc* __restrict__ desc, const uint32_t* __restrict__ off,
                                                     const uint32_t* __restrict__ tab, uint32_t NP, uint32_t n_small, uint32_t n_desc,
                                                     uint32_t small_y, uint32_t nq, uint32_t* __restrict__ out) {
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const uint32_t* const cnt = tab;
  const uint32_t* const U = tab + 3 * TAIL_MAX_SHIFTS;
  const uint32_t* const W = U + (size_t)TAIL_MAX_SHIFTS * NP;
  const uint32_t* const F = W + (size_t)TAIL_MAX_SHIFTS * NP;
  if (b < n_small * small_y) {
    const uint32_t d = b / small_y, y = b - d * small_y;
    const TailDesc D = desc[d];
    const bool isF = D.kind == TD_HASH_F, coords = D.kind == TD_COORDS_W;
    const uint32_t n = cnt[(isF ? 2 * TAIL_MAX_SHIFTS : TAIL_MAX_SHIFTS) + D.k];
    const uint32_t* list = isF ? F + (size_t)D.k * 4 * NP : W + (size_t)D.k * NP;
    const uint32_t sh = coords ? 2 : 3, wpi = 1u << sh, per = 256u >> sh, w = tid & (wpi - 1);
    uint32_t* o = out + off[d];
    const uint32_t* src = coords ? (const uint32_t*)D.p[w] : (const uint32_t*)D.p[0];
    for (uint32_t r = y * per + (tid >> sh); r < n; r += small_y * per) {
      const uint32_t idx = list[r];
      o[r * wpi + w] = coords ? src[idx] : src[(size_t)idx * 8 + w];
    }
  } else {
    const uint32_t b2 = b - n_small * small_y;
    const uint32_t d = n_small + b2 / nq, r = b2 % nq;
    if (d >= n_desc) return;
    const TailDesc D = desc[d];
    if (r >= cnt[D.k]) return;
    const uint32_t row = U[(size_t)D.k * NP + r];
    const uint32_t* const* cols = (const uint32_t* const*)D.p[0];
    uint32_t* o = out + off[d] + (size_t)r * D.width;
    for (uint32_t c = tid; c < D.width; c += 256) o[c] = cols[c][row];
  }
}

// ================================================================= host wrappers
void tail_last_layer(const TailLastArgs& a, hipStream_t st) {
  CM_CHECK((1u << a.log_n) <= TAIL_MAX_LAST && a.log_keep <= a.log_n, "tail_last_layer: last layer too large");
  KProfScope kp("k_tail_last", 0.0, st);
  hipLaunchKernelGGL(k_tail_last, dim3(1), dim3(256), 0, st, a);
  CM_HIP(hipGetLastError());
}
void tail_grind(const uint32_t* d_chan, uint32_t bits, unsigned long long* d_nonce, hipStream_t st) {
  CM_CHECK(bits <= TAIL_MAX_POW_BITS, "tail_grind: pow_bits");
  uint64_t limit = (uint64_t)1 << std::max(bits + 4, 8u);               // 16x the expected nonce: a miss has probability e^-16
  if (const int cap = tune(T_TAIL_GRIND_CAP)) limit = std::min<uint64_t>(limit, (uint64_t)1 << (cap - 1));   // (tests force the fallback)
  const uint64_t span = std::min<uint64_t>(std::max<uint64_t>(limit, 256), (uint64_t)1 << 17);   // 512 blocks: two per CU
  KProfScope kp("k_tail_grind", 0.0, st);
  if (framing().mix_u64_u32s) hipLaunchKernelGGL(k_tail_grind<true>, dim3((uint32_t)(span / 256)), dim3(256), 0, st, d_chan, bits, span, limit, d_nonce);
  else hipLaunchKernelGGL(k_tail_grind<false>, dim3((uint32_t)(span / 256)), dim3(256), 0, st, d_chan, bits, span, limit, d_nonce);
  CM_HIP(hipGetLastError());
}
void tail_tables(const TailTablesArgs& a, hipStream_t st) {
  CM_CHECK(a.n_queries >= 1 && a.n_queries <= TAIL_MAX_QUERIES && a.nq_pad >= a.n_queries && (a.nq_pad & (a.nq_pad - 1)) == 0 &&
               a.nq_pad <= TAIL_MAX_QUERIES && a.log_domain < TAIL_MAX_SHIFTS,
           "tail_tables: query count / domain size");
  KProfScope kp("k_tail_tables", 0.0, st);
  if (framing().mix_u64_u32s) hipLaunchKernelGGL(k_tail_tables<true>, dim3(1), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(k_tail_tables<false>, dim3(1), dim3(1024), 0, st, a);
  CM_HIP(hipGetLastError());
}
void tail_gather(const TailDesc* d_desc, const uint32_t* d_off, const uint32_t* d_tab, uint32_t nq_pad, uint32_t n_small, uint32_t n_desc,
                 uint32_t n_queries, uint32_t* out, hipStream_t st) {
  if (!n_desc) return;
  const uint32_t small_y = (n_queries + 31) / 32;
  const uint32_t blocks = n_small * small_y + (n_desc - n_small) * n_queries;
  KProfScope kp("k_tail_gather", 0.0, st);
  hipLaunchKernelGGL(k_tail_gather, dim3(blocks), dim3(256), 0, st, d_desc, d_off, d_tab, nq_pad, n_small, n_desc, small_y, n_queries, out);
  CM_HIP(hipGetLastError());
}

// host mirror: same definitions; U[k] is unique(S >> k) by one pass, a sibling is a neighbour in U[k], the children of a node
// are a run of U[k - 1]
void TailTables::build(const std::vector<uint32_t>& S, uint32_t log_domain, uint32_t qmask) {
  L0 = log_domain;
  n = std::max<uint32_t>((uint32_t)S.size(), 1u);
  buf.assign((size_t)(L0 + 1) * 5 * n, 0);
  memset(cnt, 0, sizeof(cnt));
  const uint32_t ns = (uint32_t)S.size();
  for (uint32_t k = 0; k <= L0; k++) {
    const uint32_t l = L0 - k;
    const bool exp_l = (qmask >> l) & 1u, exp_c = l + 1 <= L0 && ((qmask >> (l + 1)) & 1u);
    uint32_t *U = list(0, k), *W = list(1, k), *F = list(2, k);
    uint32_t nu = 0, nw = 0, nf = 0;
    for (uint32_t i = 0; i < ns; i++) { const uint32_t u = S[i] >> k; if (!nu || U[nu - 1] != u) U[nu++] = u; }
    const uint32_t* C = k >= 1 ? list(0, k - 1) : nullptr;   // the layer below
    const uint32_t nc = k >= 1 ? cnt[0][k - 1] : 0;
    uint32_t p = 0;
    for (uint32_t r = 0; r < nu; r++) {
      const uint32_t u = U[r], sib = u ^ 1u;
      const bool sib_here = (u & 1u) ? (r > 0 && U[r - 1] == sib) : (r + 1 < nu && U[r + 1] == sib);
      const bool mW = l >= 1 && !sib_here;
      if (mW) W[nw++] = sib;
      if (k >= 1) {
        bool has0 = false, has1 = false;
        if (p < nc && C[p] == 2 * u) { has0 = true; p++; }
        if (p < nc && C[p] == 2 * u + 1) { has1 = true; p++; }
        uint32_t own = 0xFFFFFFFFu;
        if (!exp_c) { if (!has0) own = 2 * u; else if (!has1) own = 2 * u + 1; }
        const bool sib_only = exp_l && mW;
        if (sib_only && sib < u) { F[nf++] = 2 * sib; F[nf++] = 2 * sib + 1; }
        if (own != 0xFFFFFFFFu) F[nf++] = own;
        if (sib_only && sib > u) { F[nf++] = 2 * sib; F[nf++] = 2 * sib + 1; }
      }
    }
    cnt[0][k] = nu; cnt[1][k] = nw; cnt[2][k] = nf;
  }
}

}  // namespace cm
