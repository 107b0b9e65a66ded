// Register-blocked circle-FFT pass for gfx950.
//
// A block of 256 threads owns a 2^11-element tile; every thread keeps 8 elements in registers and applies
// up to 3 butterfly layers (radix-8) before the tile is re-distributed through LDS, so an 11-layer pass
// costs 3 LDS exchanges instead of 11 read-modify-write sweeps.  The kernel is specialised on the number of
// layers W of the pass (tile = 2^W strided values x 2^M contiguous words, M = 11 - W; W = 11 is the
// contiguous pass) so that every index computation constant-folds and the round loop is fully unrolled —
// the pass is ALU-bound on M31 butterflies, not HBM-bound, so instruction count is what matters.
// Global accesses: the contiguous pass stages its HBM side through LDS with 16-byte-per-lane accesses;
// strided passes move 2^M-word runs (M >= 4 => full 64-B segments) with coalesced dwords.
// Replaces the butterfly loops of Stwo's SimdBackend `ifft`/`rfft` (reached from
// tree_builder.extend_evals / commit, crates/prover/src/prover.rs:71-73, 80-82, 100-102).
#include "field.hpp"
#include "device_common.hpp"
#include "fft_pass.hpp"

namespace cm {

constexpr uint32_t TILE_LOG = 11;
// LDS padding: one extra word every 32 to break the power-of-two strides of the exchanges; a thread's 8 accesses of one
// exchange then share a base register and differ by immediate offsets.  (A conflict-free GF(2) swizzle
// i ^ ((i>>5)&7) ^ (((i>>6)&3)<<3) was measured 5 % slower: it needs a v_xor per access, and the pass is VALU-bound —
// tools/fft_lab.hip, tools/lds_lab.hip.)
__device__ __forceinline__ constexpr uint32_t phys(uint32_t i) { return i + (i >> 5); }

template <int W>
struct PassGeom {
  static constexpr uint32_t M = (W == 11) ? 0u : (uint32_t)(TILE_LOG - W);
  static constexpr uint32_t NR = (W + 2) / 3;
  // round r works on local bits [b, b + k)
  static __device__ __forceinline__ constexpr uint32_t b(uint32_t r) { return M + 3 * r; }
  static __device__ __forceinline__ constexpr uint32_t k(uint32_t r) { return (W - 3 * r) < 3 ? (W - 3 * r) : 3; }
};

template <bool INVERSE, int W>
__global__ void __launch_bounds__(256) k_fft_pass_r8(FftPassArgs a) {
  using G = PassGeom<W>;
  constexpr uint32_t M = G::M;
  __shared__ uint32_t tile[(1u << TILE_LOG) + (1u << (TILE_LOG - 5))];
  const uint32_t* __restrict__ src = a.src[blockIdx.y];
  uint32_t* __restrict__ dst = a.dst[blockIdx.y];
  const uint32_t low_fixed_bits = a.lo - M;
  const uint32_t lowf = blockIdx.x & ((1u << low_fixed_bits) - 1);
  const uint32_t high = blockIdx.x >> low_fixed_bits;
  const uint32_t base = (high << a.hi) | (lowf << M);
  const uint32_t t = threadIdx.x;
  const uint32_t lo = (W == 11) ? 0u : a.lo;  // the contiguous pass starts at layer 0: fold the shifts away
  const bool padded = a.in_len < (1u << a.n);   // only the first pass of an LDE reads implicit zeros
  auto gidx = [&](uint32_t li) -> uint32_t { return base | ((li >> M) << lo) | (li & ((1u << M) - 1)); };
  M31 v[8];
#pragma unroll
  for (uint32_t rr = 0; rr < G::NR; rr++) {
    const uint32_t r = INVERSE ? rr : (G::NR - 1 - rr);
    const uint32_t b = G::b(r), k = G::k(r);
    // element e of thread t: j = e & (2^k-1) inside the butterfly group, g = e >> k selects the group; the 11-k remaining
    // bits are rho = (t << (3-k)) | g and the local index is ((rho >> b) << (b+k)) | (j << b) | (rho & (2^b - 1)).
    // The bits of (j, g) and those of t never overlap, so li[e] = li0(t) + c[e] with c[e] a compile-time constant: the
    // eight tile addresses of a round are ONE computed base plus immediate offsets (the LDS padding i + (i >> 5) splits
    // the same way because the low five bits cannot carry), and so are the global indices of the contiguous pass.
    const uint32_t rho0 = t << (3 - k);
    const uint32_t li0 = ((rho0 >> b) << (b + k)) | (rho0 & ((1u << b) - 1));
    const uint32_t pli0 = phys(li0);
    const uint32_t gi0 = gidx(li0);
    uint32_t c[8];
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) {
      const uint32_t j = e & ((1u << k) - 1), g = e >> k;
      c[e] = ((g >> b) << (b + k)) | (j << b) | (g & ((1u << b) - 1));
    }
    auto ptile = [&](uint32_t e) -> uint32_t { return pli0 + c[e] + (c[e] >> 5); };
    auto gel = [&](uint32_t e) -> uint32_t { return gi0 + (((c[e] >> M) << lo) | (c[e] & ((1u << M) - 1))); };
    const bool staged_in = (rr == 0) && INVERSE && M == 0;
    if (rr == 0 && !staged_in) {
      if (padded) {
        // zero-extension without exec-masked loads: out-of-range lanes read element 0 (one cached line) and the
        // value is replaced by a select, so the 8 loads still issue back to back
#pragma unroll
        for (uint32_t e = 0; e < 8; e++) {
          const uint32_t gi = gel(e);
          const bool in = gi < a.in_len;
          const uint32_t x = src[in ? gi : 0u];
          v[e] = M31(in ? x : 0u);
        }
      } else {  // no per-element bounds test: 8 loads issue back to back instead of 8 exec-masked branches
#pragma unroll
        for (uint32_t e = 0; e < 8; e++) { const uint32_t gi = gel(e); __builtin_assume(gi < (1u << 29)); v[e] = M31(src[gi]); }
      }
    } else {
      if (staged_in) {
        // contiguous tile, first round works on index bits 0..2: stage through LDS with coalesced 16-byte loads
#pragma unroll
        for (uint32_t it = 0; it < 2; it++) {
          uint32_t w0 = (it * 256 + t) * 4;
          uint4 q = make_uint4(0, 0, 0, 0);
          if (!padded || base + w0 < a.in_len) q = *reinterpret_cast<const uint4*>(src + base + w0);
          tile[phys(w0)] = q.x; tile[phys(w0 + 1)] = q.y; tile[phys(w0 + 2)] = q.z; tile[phys(w0 + 3)] = q.w;
        }
        __syncthreads();
      }
#ifndef CM_FFT_ABL_NO_LDS
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) v[e] = M31(tile[ptile(e)]);
#endif
    }
    // butterflies: k layers on bits [b, b+k).  h(e) = h0 + (j >> (s+1)) with h0 from the j = 0 element of the group
#pragma unroll
    for (uint32_t ss = 0; ss < k; ss++) {
      const uint32_t s = INVERSE ? ss : (k - 1 - ss);
      const uint32_t layer = lo + (b - M) + s;
      const uint32_t* __restrict__ twp;
      if (W == 11 && b + s == 0) twp = a.ytw + (1u << (a.n - 1));
      else {
        const uint32_t L = a.R - a.n + layer - 1;
        twp = a.xtw + ((1u << (a.R - 1)) - (1u << (a.R - 1 - L)));
      }
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) {
        if ((e >> s) & 1u) continue;
        const uint32_t e1 = e | (1u << s);
        const uint32_t g0 = e & ~((1u << k) - 1);  // j = 0 element of this group
        const uint32_t j = e & ((1u << k) - 1);
        uint32_t h = (gel(g0) >> (layer + 1)) + (j >> (s + 1));
        __builtin_assume(h < (1u << 29));  // byte offset fits 32 bits: saddr + 32-bit voffset addressing
        // The 64 lanes of a wave differ in 6 consecutive bits of rho = (t << (3-k)) | g, i.e. local index bits below
        // 9 - k; a round on bits [b, b+k) with 9 - k <= b therefore pairs elements whose twiddle index is the same
        // in every lane: fetch it with a scalar load (no per-lane address arithmetic, no vector memory op).
        if (9 - k <= b) h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
#ifdef CM_FFT_ABL_NO_TW  /* tools/fft_lab: cost of the twiddle fetch (results are wrong) */
        M31 w(h | 3u);
#else
        M31 w(twp[h]);
#endif
        M31 x = v[e], y = v[e1];
        if (INVERSE) { v[e] = x + y; v[e1] = (x - y) * w; }
        else { M31 yt = y * w; v[e] = x + yt; v[e1] = x - yt; }
      }
    }
    const bool staged_out = (rr + 1 == G::NR) && !INVERSE && M == 0;
    if (rr + 1 == G::NR && !staged_out) {
      const M31 sc(a.scale);
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) {
        M31 o = v[e];
        if (INVERSE && a.scale != 1u) o = o * sc;
        dst[gel(e)] = o.v;
      }
    } else {
#ifndef CM_FFT_ABL_NO_LDS  /* tools/fft_lab: cost of the LDS exchanges (results are wrong) */
      __syncthreads();  // previous round's readers are done with the tile
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) tile[ptile(e)] = v[e].v;
      __syncthreads();
#endif
      if (staged_out) {
        // forward transform, last round holds 8 consecutive words per lane: store 16 B per lane from LDS
#pragma unroll
        for (uint32_t it = 0; it < 2; it++) {
          uint32_t w0 = (it * 256 + t) * 4;
          uint4 q = make_uint4(tile[phys(w0)], tile[phys(w0 + 1)], tile[phys(w0 + 2)], tile[phys(w0 + 3)]);
          *reinterpret_cast<uint4*>(dst + base + w0) = q;
        }
      }
    }
  }
}

template <bool INV>
static void launch_w(const FftPassArgs& a, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  const uint32_t W = a.hi - a.lo;
  dim3 grid(ntiles, ncols), block(256);
  switch (W) {
    case 1: hipLaunchKernelGGL((k_fft_pass_r8<INV, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_fft_pass_r8<INV, 2>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((k_fft_pass_r8<INV, 3>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((k_fft_pass_r8<INV, 4>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((k_fft_pass_r8<INV, 5>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((k_fft_pass_r8<INV, 6>), grid, block, 0, st, a); break;
    case 7: hipLaunchKernelGGL((k_fft_pass_r8<INV, 7>), grid, block, 0, st, a); break;
    case 11: hipLaunchKernelGGL((k_fft_pass_r8<INV, 11>), grid, block, 0, st, a); break;
    default: break;
  }
}
bool fft_pass_r8_supported(uint32_t W, uint32_t M, uint32_t lo) {
  if (W == 11) return lo == 0 && M == 0;
  return W >= 1 && W <= 7 && M == TILE_LOG - W && lo >= M;
}
void launch_fft_pass_r8(bool inverse, const FftPassArgs& a, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  if (inverse) launch_w<true>(a, ntiles, ncols, st);
  else launch_w<false>(a, ntiles, ncols, st);
}

}  // namespace cm
