// Register-blocked circle-FFT pass for gfx950.
//
// A block of 256 threads owns a 2^11-element tile; every thread keeps 8 elements in registers and applies
// up to 3 butterfly layers (radix-8) before the tile is re-distributed through LDS, so an 11-layer pass
// costs 3 LDS exchanges instead of 11 read-modify-write sweeps (the first and last rounds go straight
// from/to HBM).  Global accesses: contiguous pass = 32 B per lane (8 consecutive words) on one side and
// coalesced dwords on the other; strided passes move 2^M-word runs (M >= 4 => full 64-B segments).
// Replaces the butterfly loops of Stwo's SimdBackend `ifft`/`rfft` (reached from
// tree_builder.extend_evals / commit, crates/prover/src/prover.rs:71-73, 80-82, 100-102).
#include "field.hpp"
#include "device_common.hpp"
#include "fft_pass.hpp"

namespace cm {

constexpr uint32_t TILE_LOG = 11;
// LDS padding: one extra word every 32 to break the power-of-two strides of the exchanges
__device__ __forceinline__ uint32_t phys(uint32_t i) { return i + (i >> 5); }


__device__ __forceinline__ uint32_t tile_gidx(const FftPassArgs& a, uint32_t base, uint32_t li) {
  return base | ((li >> a.M) << a.lo) | (li & ((1u << a.M) - 1));
}
template <bool INVERSE, int K>
__device__ __forceinline__ void butterflies(M31 (&v)[8], const uint32_t (&li)[8], const FftPassArgs& a, uint32_t b, uint32_t base) {
#pragma unroll
  for (int ss = 0; ss < K; ss++) {
    const int s = INVERSE ? ss : (K - 1 - ss);
    const uint32_t layer = a.lo + (b - a.M) + s;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if ((e >> s) & 1) continue;
      const int e1 = e | (1 << s);
      uint32_t h = tile_gidx(a, base, li[e]) >> (layer + 1);
      uint32_t tw;
      if (layer == 0) tw = a.ytw[(1u << (a.n - 1)) + h];
      else {
        uint32_t L = a.R - a.n + layer - 1;
        tw = a.xtw[(1u << (a.R - 1)) - (1u << (a.R - 1 - L)) + h];
      }
      M31 x = v[e], y = v[e1], w(tw);
      if (INVERSE) { v[e] = x + y; v[e1] = (x - y) * w; }
      else { M31 yt = y * w; v[e] = x + yt; v[e1] = x - yt; }
    }
  }
}

template <bool INVERSE>
__global__ void __launch_bounds__(256) k_fft_pass_r8(FftPassArgs a) {
  __shared__ uint32_t tile[(1u << TILE_LOG) + (1u << (TILE_LOG - 5))];
  const uint32_t W = a.hi - a.lo, M = a.M;
  const uint32_t* __restrict__ src = a.src[blockIdx.y];
  uint32_t* __restrict__ dst = a.dst[blockIdx.y];
  const uint32_t low_fixed_bits = a.lo - M;
  const uint32_t lowf = blockIdx.x & ((1u << low_fixed_bits) - 1);
  const uint32_t high = blockIdx.x >> low_fixed_bits;
  const uint32_t base = (high << a.hi) | (lowf << M);
  const uint32_t t = threadIdx.x;
  // local index (11 bits) -> global index
  auto gidx = [&](uint32_t li) { return base | ((li >> M) << a.lo) | (li & ((1u << M) - 1)); };
  // rounds over the active local bits [M, 11): chunks of 3 (last chunk may be 1 or 2)
  const uint32_t nrounds = (W + 2) / 3;
  M31 v[8];
  for (uint32_t rr = 0; rr < nrounds; rr++) {
    const uint32_t r = INVERSE ? rr : (nrounds - 1 - rr);
    const uint32_t b = M + 3 * r;                       // first local bit of the round
    const uint32_t k = (W - 3 * r) < 3 ? (W - 3 * r) : 3;  // layers in this round
    // element e of thread t: j = e & (2^k-1) inside the butterfly group, g = e >> k selects the group
    uint32_t li[8];
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) {
      uint32_t j = e & ((1u << k) - 1), g = e >> k;
      uint32_t rho = (t << (3 - k)) | g;               // the 11-k remaining bits
      li[e] = ((rho >> b) << (b + k)) | (j << b) | (rho & ((1u << b) - 1));
    }
    if (rr == 0 && !(INVERSE && M == 0)) {
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) {
        uint32_t gi = gidx(li[e]);
        v[e] = M31(gi < a.in_len ? src[gi] : 0u);
      }
    } else {
      if (rr == 0) {
        // contiguous tile, first round works on index bits 0..2: a direct load would be 8 dwords per lane at a
        // 32-byte lane stride.  Stage the tile through LDS with fully coalesced 16-byte loads instead.
#pragma unroll
        for (uint32_t it = 0; it < 2; it++) {
          uint32_t w0 = (it * 256 + t) * 4;
          uint4 q = make_uint4(0, 0, 0, 0);
          if (base + w0 < a.in_len) q = *reinterpret_cast<const uint4*>(src + base + w0);
          tile[phys(w0)] = q.x; tile[phys(w0 + 1)] = q.y; tile[phys(w0 + 2)] = q.z; tile[phys(w0 + 3)] = q.w;
        }
        __syncthreads();
      }
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) v[e] = M31(tile[phys(li[e])]);
    }
    // butterflies (compile-time layer count so that v[] stays in registers)
    if (k == 3) butterflies<INVERSE, 3>(v, li, a, b, base);
    else if (k == 2) butterflies<INVERSE, 2>(v, li, a, b, base);
    else butterflies<INVERSE, 1>(v, li, a, b, base);
    if (rr + 1 == nrounds && !(!INVERSE && M == 0)) {
      const M31 sc(a.scale);
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) {
        M31 o = v[e];
        if (a.scale != 1u) o = o * sc;
        dst[gidx(li[e])] = o.v;
      }
    } else if (rr + 1 == nrounds) {
      // forward transform, last round holds 8 consecutive words per lane: stage through LDS, store 16 B per lane
      __syncthreads();
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) tile[phys(li[e])] = v[e].v;
      __syncthreads();
#pragma unroll
      for (uint32_t it = 0; it < 2; it++) {
        uint32_t w0 = (it * 256 + t) * 4;
        uint4 q = make_uint4(tile[phys(w0)], tile[phys(w0 + 1)], tile[phys(w0 + 2)], tile[phys(w0 + 3)]);
        *reinterpret_cast<uint4*>(dst + base + w0) = q;
      }
    } else {
      __syncthreads();  // previous round's readers are done with the tile
#pragma unroll
      for (uint32_t e = 0; e < 8; e++) tile[phys(li[e])] = v[e].v;
      __syncthreads();
    }
  }
}

void launch_fft_pass_r8(bool inverse, const FftPassArgs& a, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  if (inverse) hipLaunchKernelGGL(k_fft_pass_r8<true>, dim3(ntiles, ncols), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_fft_pass_r8<false>, dim3(ntiles, ncols), dim3(256), 0, st, a);
}

}  // namespace cm
