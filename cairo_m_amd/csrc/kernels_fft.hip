// Register-blocked circle-FFT passes for gfx950.
//
// One launch applies W butterfly layers to a batch of columns.  A block owns a tile of 2^TL elements — 2^W values of the
// strided index bits [lo, hi) times 2^M consecutive words (M = TL - W; the contiguous pass has lo = 0, M = 0) — and
// every thread keeps 2^E of them in registers, applying up to E layers (radix 2^E) before the tile is re-distributed
// through LDS.  Three shapes are instantiated:
//
//   TL = 11, E = 3   256 threads, 8 KiB tile     strided passes of 1..7 layers (M >= 4: 64-B runs and longer)
//   TL = 12, E = 4   256 threads, 16 KiB tile    the contiguous pass: 12 layers in three radix-16 rounds
//   TL = 13, E = 5   256 threads, 32 KiB tile    contiguous pass of 13 layers (5 + 3 + 5) for 22-layer transforms
//   TL = 14, E = 4   1024 threads, 64 KiB tile   strided passes of 6..10 layers (two blocks per CU in the 160 KiB LDS)
//
// so that a 2^21 / 2^22 transform is TWO sweeps over HBM (12 + 9 / 13 + 9 layers) instead of three (11 + 5 + 5 /
// 11 + 6 + 5) and a layer costs fewer LDS exchanges (a pass of 9-13 layers has 3 rounds; the radix-8 form needed 4).
// The kernels are ALU-bound on the M31 butterflies (profiles/*_pmc_sq.json: VALU-active share x resident waves ~ 1), so
// both the instruction count per layer and the number of sweeps matter.  Everything is specialised on (W, TL, E) so the
// index arithmetic constant-folds and the rounds unroll.
// Global accesses: the contiguous pass stages its HBM side through LDS with 16-byte-per-lane accesses; strided passes move
// 2^M-word runs with coalesced dwords; the rounds are ordered so that a partial round (fewer than E layers) is never the
// one facing HBM when the pass has three rounds.
// Replaces the butterfly loops of Stwo's SimdBackend `ifft` / `rfft` (reached from tree_builder.extend_evals / commit,
// crates/prover/src/prover.rs:71-73, 80-82, 100-102).
#include "field.hpp"
#include "device_common.hpp"
#include "fft_pass.hpp"
#include "engine.hpp"

namespace cm {

// LDS padding: one extra word every 32 to break the power-of-two strides of the exchanges; a thread's accesses of one
// exchange then share a base register and differ by immediate offsets.  (A conflict-free GF(2) swizzle was measured 5 %
// slower: it needs a v_xor per access, and the pass is VALU-bound — tools/fft_lab.hip, tools/lds_lab.hip.)
__device__ __forceinline__ constexpr uint32_t phys(uint32_t i) { return i + (i >> 5); }

template <int W, int TL, int E>
struct PassGeom {
  static constexpr uint32_t M = (W == TL) ? 0u : (uint32_t)(TL - W);
  static constexpr uint32_t NR = (W + E - 1) / E;
  static constexpr uint32_t REM = W - E * (NR - 1);   // layers of the one partial round (= E when E divides W)
  // round r works on local bits [b(r), b(r) + k(r)).  With three or more rounds the partial one sits in the middle, so
  // the rounds that load from / store to HBM have every lane on consecutive words.
  static __device__ __forceinline__ constexpr uint32_t k(uint32_t r) {
    if (NR >= 3) return r == 1 ? REM : (uint32_t)E;
    return r == NR - 1 ? REM : (uint32_t)E;
  }
  static __device__ __forceinline__ constexpr uint32_t b(uint32_t r) {
    uint32_t s = M;
    for (uint32_t i = 0; i < r; i++) s += k(i);
    return s;
  }
};

// The pass itself, for tile `bx` of the pass `a` describes on columns src -> dst.  PRELOADED: v[] already holds the thread's 2^E
// elements of the first round (the fused kernel below hands the coefficients over in registers); on return v[] holds what the
// last round stored (non-staged stores only).
template <bool INVERSE, int W, int TL, int E, bool PRELOADED>
__device__ __forceinline__ void fft_pass_rb_body(const FftPassArgs& a, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                 const uint32_t bx, M31 (&v)[1 << E], uint32_t* tile) {
  using G = PassGeom<W, TL, E>;
  constexpr uint32_t M = G::M;
  constexpr uint32_t NE = 1u << E;            // elements per thread
  constexpr uint32_t NT = 1u << (TL - E);     // threads per block
  static_assert(!PRELOADED || M > 0, "only strided passes take their input in registers");
  // raw buffer resources over the two columns (stride 0, no bounds: offsets stay below 2^(n+2) <= 2^30 bytes)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0xffffffffu, 0x00020000);
  const uint32_t low_fixed_bits = a.lo - M;
  const uint32_t lowf = bx & ((1u << low_fixed_bits) - 1);
  const uint32_t high = bx >> low_fixed_bits;
  const uint32_t base = (high << a.hi) | (lowf << M);
  const uint32_t t = threadIdx.x;
  const uint32_t lo = (M == 0) ? 0u : a.lo;   // the contiguous pass starts at layer 0: fold the shifts away
  const bool padded = a.in_len < (1u << a.n);   // only the first pass of an LDE reads implicit zeros
  auto gidx = [&](uint32_t li) -> uint32_t { return base | ((li >> M) << lo) | (li & ((1u << M) - 1)); };
#pragma clang loop unroll(full)
  for (uint32_t rr = 0; rr < G::NR; rr++) {
    const uint32_t r = INVERSE ? rr : (G::NR - 1 - rr);
    const uint32_t b = G::b(r), k = G::k(r);
    // element e of thread t: j = e & (2^k-1) inside the butterfly group, g = e >> k selects the group; the TL-k remaining
    // bits are rho = (t << (E-k)) | g and the local index is ((rho >> b) << (b+k)) | (j << b) | (rho & (2^b - 1)).
    // The bits of (j, g) and those of t never overlap, so li[e] = li0(t) + c[e] with c[e] a compile-time constant: the
    // tile addresses of a round are ONE computed base plus immediate offsets (the LDS padding i + (i >> 5) splits the same
    // way because the low five bits cannot carry), and so are the global indices of the contiguous pass.
    const uint32_t rho0 = t << (E - k);
    const uint32_t li0 = ((rho0 >> b) << (b + k)) | (rho0 & ((1u << b) - 1));
    const uint32_t pli0 = phys(li0);
    const uint32_t gi0 = gidx(li0);
    uint32_t c[NE];
#pragma clang loop unroll(full)
    for (uint32_t e = 0; e < NE; e++) {
      const uint32_t j = e & ((1u << k) - 1), g = e >> k;
      c[e] = ((g >> b) << (b + k)) | (j << b) | (g & ((1u << b) - 1));
    }
    auto ptile = [&](uint32_t e) -> uint32_t { return pli0 + c[e] + (c[e] >> 5); };
    // global index of element e = gi0 (per lane) + cgl(e) (the same in every lane: a compile-time constant shifted by the
    // runtime `lo`).  The HBM accesses of a round are BUFFER instructions: one per-lane byte offset shared by all 2^E accesses
    // (voffset) plus a scalar offset per element (soffset) — `buffer_load_dword v, v_off, s[rsrc], s_e offen` — instead of a
    // v_lshl_add_u32 + v_lshl_add_u64 pair per access (two VOP3, one of them 64-bit: 64 of the ~1000 VALU instructions of a
    // strided 9-layer pass; flat `global_` addressing has no scalar-offset operand and the compiler re-associates a uniform
    // base pointer back into per-lane 64-bit adds).
    auto cgl = [&](uint32_t e) -> uint32_t { return ((c[e] >> M) << lo) | (c[e] & ((1u << M) - 1)); };
    auto gel = [&](uint32_t e) -> uint32_t { return gi0 + cgl(e); };
    const uint32_t gb0 = gi0 << 2;   // byte offset of the lane inside the column
    const bool staged_in = (rr == 0) && INVERSE && M == 0;
    if (rr == 0 && PRELOADED) {
      // (the caller's registers)
    } else if (rr == 0 && !staged_in) {
      if (padded && !INVERSE && a.hi == a.n && a.in_len == (1u << (a.n - 1))) {
        // extension by two (every LDE of the prover): the upper half of the input is implicit zero padding and the top layer,
        // applied first, turns every (x, 0) into (x, x) without a twiddle (below) — so only the elements of the LOWER half are
        // loaded (all of them in range: no bounds test, no select) and the rest is filled in by that layer
#pragma clang loop unroll(full)
        for (uint32_t e = 0; e < NE; e++)
          v[e] = ((e >> (k - 1)) & 1u) ? M31(0u) : M31((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)gb0, (int)(cgl(e) << 2), 0));
      } else if (padded) {
        // zero-extension without exec-masked loads: out-of-range lanes read element 0 (one cached line) and the
        // value is replaced by a select, so the loads still issue back to back
#pragma clang loop unroll(full)
        for (uint32_t e = 0; e < NE; e++) {
          const uint32_t gi = gel(e);
          const bool in = gi < a.in_len;
          const uint32_t x = src[in ? gi : 0u];
          v[e] = M31(in ? x : 0u);
        }
      } else {  // no per-element bounds test: the loads issue back to back instead of as exec-masked branches
#pragma clang loop unroll(full)
        for (uint32_t e = 0; e < NE; e++) v[e] = M31((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)gb0, (int)(cgl(e) << 2), 0));
      }
    } else {
      if (staged_in) {
        // contiguous tile, first round works on the lowest index bits: stage through LDS with coalesced 16-byte loads
#pragma clang loop unroll(full)
        for (uint32_t it = 0; it < (1u << (TL - 2)) / NT; it++) {
          uint32_t w0 = (it * NT + t) * 4;
          uint4 q = make_uint4(0, 0, 0, 0);
          if (!padded || base + w0 < a.in_len) q = *reinterpret_cast<const uint4*>(src + base + w0);
          tile[phys(w0)] = q.x; tile[phys(w0 + 1)] = q.y; tile[phys(w0 + 2)] = q.z; tile[phys(w0 + 3)] = q.w;
        }
        __syncthreads();
      }
#ifndef CM_FFT_ABL_NO_LDS
#pragma clang loop unroll(full)
      for (uint32_t e = 0; e < NE; e++) v[e] = M31(tile[ptile(e)]);
#endif
    }
    // butterflies: k layers on bits [b, b+k).  h(e) = h0 + (j >> (s+1)) with h0 from the j = 0 element of the group
#pragma clang loop unroll(full)
    for (uint32_t ss = 0; ss < k; ss++) {
      const uint32_t s = INVERSE ? ss : (k - 1 - ss);
      const uint32_t layer = lo + (b - M) + s;
      const uint32_t* __restrict__ twp;
      if (M == 0 && b + s == 0) twp = a.ytw + (1u << (a.n - 1));
      else {
        const uint32_t L = a.R - a.n + layer - 1;
        twp = a.xtw + ((1u << (a.R - 1)) - (1u << (a.R - 1 - L)));
      }
      if (!INVERSE && rr == 0 && ss == 0) {
        // first layer a forward transform applies = the top layer of this pass.  When it is the top layer of the TRANSFORM
        // and the upper half of the input is implicit zero padding (an LDE: in_len <= 2^(n-1)), every butterfly is
        // (x, 0) -> (x, x): no twiddle, no multiplication — one layer in ~22 of a 2^21 / 2^22 extension
        if (a.hi == a.n && a.in_len <= (1u << (a.n - 1))) {
#pragma clang loop unroll(full)
          for (uint32_t e = 0; e < NE; e++)
            if (!((e >> s) & 1u)) v[e | (1u << s)] = v[e];
          continue;
        }
      }
#pragma clang loop unroll(full)
      for (uint32_t e = 0; e < NE; e++) {
        if ((e >> s) & 1u) continue;
        const uint32_t e1 = e | (1u << s);
        const uint32_t g0 = e & ~((1u << k) - 1);  // j = 0 element of this group
        const uint32_t j = e & ((1u << k) - 1);
        uint32_t h = (gel(g0) >> (layer + 1)) + (j >> (s + 1));
        __builtin_assume(h < (1u << 29));  // byte offset fits 32 bits: saddr + 32-bit voffset addressing
        // The 64 lanes of a wave differ in 6 consecutive bits of rho = (t << (E-k)) | g, i.e. local index bits below
        // E - k + 6; a round on bits [b, b+k) with E - k + 6 <= b therefore pairs elements whose twiddle index is the same
        // in every lane: fetch it with a scalar load (no per-lane address arithmetic, no vector memory op).
        if (E - k + 6 <= b) h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
#ifdef CM_FFT_ABL_NO_TW  /* tools/fft_lab: cost of the twiddle fetch (results are wrong) */
        const uint32_t w = h | 3u;
#else
        const uint32_t w = twp[h];   // 2 * twiddle (mul_tw2)
#endif
        M31 x = v[e], y = v[e1];
        if (INVERSE) { v[e] = x + y; v[e1] = mul_tw2(x - y, w); }
        else { M31 yt = mul_tw2(y, w); v[e] = x + yt; v[e1] = x - yt; }
      }
    }
    const bool staged_out = (rr + 1 == G::NR) && !INVERSE && M == 0;
    if (rr + 1 == G::NR && !staged_out) {
      const M31 sc(a.scale);
#pragma clang loop unroll(full)
      for (uint32_t e = 0; e < NE; e++) {
        M31 o = v[e];
        if (INVERSE && a.scale != 1u) o = sc * o;   // the doubled operand of M31 operator* is the loop-invariant one
        v[e] = o;
        __builtin_amdgcn_raw_buffer_store_b32((int)o.v, rdst, (int)gb0, (int)(cgl(e) << 2), 0);
      }
    } else {
#ifndef CM_FFT_ABL_NO_LDS  /* tools/fft_lab: cost of the LDS exchanges (results are wrong) */
      __syncthreads();  // previous round's readers are done with the tile
#pragma clang loop unroll(full)
      for (uint32_t e = 0; e < NE; e++) tile[ptile(e)] = v[e].v;
      __syncthreads();
#endif
      if (staged_out) {
        // forward transform, last round holds consecutive words per lane: store 16 B per lane from LDS
#pragma clang loop unroll(full)
        for (uint32_t it = 0; it < (1u << (TL - 2)) / NT; it++) {
          uint32_t w0 = (it * NT + t) * 4;
          uint4 q = make_uint4(tile[phys(w0)], tile[phys(w0 + 1)], tile[phys(w0 + 2)], tile[phys(w0 + 3)]);
          *reinterpret_cast<uint4*>(dst + base + w0) = q;
        }
      }
    }
  }
}

template <bool INVERSE, int W, int TL, int E>
__global__ void __launch_bounds__(1 << (TL - E)) k_fft_pass_rb(FftPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile[];   // 2^TL + 2^(TL-5) words
  M31 v[1 << E];
  fft_pass_rb_body<INVERSE, W, TL, E, false>(a, a.src[blockIdx.y], a.dst[blockIdx.y], blockIdx.x, v, tile);
}

// ---- the last pass of an interpolation and the first pass of the extension by two that follows it, in ONE sweep ------------
// tree_builder.extend_evals = interpolate + evaluate on the double domain (prover.rs:71-73, 80-82, 100-102).  The inverse
// transform of 2^n ends with the strided layers [12, n); the forward transform of 2^(n+1) begins with its top layer — (x, 0) ->
// (x, x) on the zero-padded coefficients, no arithmetic — and the layers [12, n) of BOTH halves: the same tile of coefficients,
// in the same register layout the inverse pass ends with.  So a block finishes the inverse pass, stores the coefficients (the
// OODS sampling reads them), keeps them in registers and runs the forward layers of the lower and of the upper half on them.
// HBM: N words read, N + 2N written, against N + N and N + 2N of the two separate passes: one read of the coefficients less,
// a fifth of what the pair moved (these strided passes run at the HBM stream rate: removing 8 % of their VALU instructions
// changed nothing, profiles/r04*).  `inv` is the pass [12, n) of the inverse transform (in place on the coefficient columns),
// `fwd` the pass [12, n) of the 2^(n+1) forward transform with dst = the extended columns (src unused).
struct FftFusedArgs { FftPassArgs inv, fwd; };
template <int W, int E = 4>
__global__ void __launch_bounds__(1 << (14 - E)) k_fft_fused_rb(FftFusedArgs a) {
  constexpr int TL = 14;
  extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
  M31 v[1 << E], cf[1 << E];
  fft_pass_rb_body<true, W, TL, E, false>(a.inv, a.inv.src[blockIdx.y], a.inv.dst[blockIdx.y], blockIdx.x, v, tile);
#pragma clang loop unroll(full)
  for (uint32_t e = 0; e < (1u << E); e++) cf[e] = v[e];
  const uint32_t half_shift = a.fwd.lo - PassGeom<W, TL, E>::M;   // tiles per half = 2^(lo - M)
  uint32_t* __restrict__ lde = a.fwd.dst[blockIdx.y];
  fft_pass_rb_body<false, W, TL, E, true>(a.fwd, nullptr, lde, blockIdx.x, v, tile);
#pragma clang loop unroll(full)
  for (uint32_t e = 0; e < (1u << E); e++) v[e] = cf[e];
  fft_pass_rb_body<false, W, TL, E, true>(a.fwd, nullptr, lde, (1u << half_shift) | blockIdx.x, v, tile);
}
template <int W, int E = 4>
static void launch_fused_one(const FftFusedArgs& a, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  constexpr size_t lds0 = ((size_t)4 << 14) + ((size_t)4 << 9);
  // "fft_half_occ" (A/B, bit 0: the 2^14-tile kernels): ask for more LDS than the tile needs so that ONE block fits a CU instead of two —
  // 4 waves per SIMD are left for the Merkle kernels the commitment pipeline runs next to the transforms
  const size_t lds = (tune(T_FFT_HALF_OCC) & 1) ? (size_t)84 * 1024 : lds0;
  static const hipError_t once = hipFuncSetAttribute((const void*)k_fft_fused_rb<W, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
  (void)once;
  hipLaunchKernelGGL((k_fft_fused_rb<W, E>), dim3(ntiles, ncols), dim3(1u << (14 - E)), lds, st, a);
}
bool fft_fused_serves(uint32_t W) { return W >= 6 && W <= 9; }
void launch_fft_fused_rb(const FftPassArgs& inv, const FftPassArgs& fwd, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  FftFusedArgs a{inv, fwd};
  switch (inv.hi - inv.lo) {
    case 6: launch_fused_one<6>(a, ntiles, ncols, st); break;
    case 7: launch_fused_one<7>(a, ntiles, ncols, st); break;
    case 8: launch_fused_one<8>(a, ntiles, ncols, st); break;
    case 9: launch_fused_one<9>(a, ntiles, ncols, st); break;
    default: break;
  }
}

template <bool INV, int W, int TL, int E>
static void launch_one(const FftPassArgs& a, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  constexpr size_t lds0 = ((size_t)4 << TL) + ((size_t)4 << (TL - 5));
  // "fft_half_occ" (A/B): bit 0 = one 2^14-tile block per CU instead of two, bit 1 = four 2^12-tile blocks per CU instead of eight
  const int ho = tune(T_FFT_HALF_OCC);
  const size_t lds = (TL == 14 && (ho & 1)) ? (size_t)84 * 1024 : (TL == 12 && (ho & 2)) ? (size_t)33 * 1024 : lds0;
  if (lds0 > 48 * 1024) {   // above the default dynamic-LDS limit: raise it once per instantiation
    static const hipError_t once =
        hipFuncSetAttribute((const void*)k_fft_pass_rb<INV, W, TL, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    (void)once;
  }
  hipLaunchKernelGGL((k_fft_pass_rb<INV, W, TL, E>), dim3(ntiles, ncols), dim3(1u << (TL - E)), lds, st, a);
}

template <bool INV>
static void launch_w(const FftPassArgs& a, uint32_t tile_log, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  const uint32_t W = a.hi - a.lo;
  if (tile_log == 11) {
    switch (W) {
      case 1: launch_one<INV, 1, 11, 3>(a, ntiles, ncols, st); break;
      case 2: launch_one<INV, 2, 11, 3>(a, ntiles, ncols, st); break;
      case 3: launch_one<INV, 3, 11, 3>(a, ntiles, ncols, st); break;
      case 4: launch_one<INV, 4, 11, 3>(a, ntiles, ncols, st); break;
      case 5: launch_one<INV, 5, 11, 3>(a, ntiles, ncols, st); break;
      case 6: launch_one<INV, 6, 11, 3>(a, ntiles, ncols, st); break;
      case 7: launch_one<INV, 7, 11, 3>(a, ntiles, ncols, st); break;
      case 11: launch_one<INV, 11, 11, 3>(a, ntiles, ncols, st); break;
      default: break;
    }
  } else if (tile_log == 12) {
    launch_one<INV, 12, 12, 4>(a, ntiles, ncols, st);
  } else if (tile_log == 13) {
    launch_one<INV, 13, 13, 5>(a, ntiles, ncols, st);
  } else if (tile_log == 14) {
    switch (W) {
      case 6: launch_one<INV, 6, 14, 4>(a, ntiles, ncols, st); break;
      case 7: launch_one<INV, 7, 14, 4>(a, ntiles, ncols, st); break;
      case 8: launch_one<INV, 8, 14, 4>(a, ntiles, ncols, st); break;
      case 9: launch_one<INV, 9, 14, 4>(a, ntiles, ncols, st); break;
      case 10: launch_one<INV, 10, 14, 4>(a, ntiles, ncols, st); break;
      default: break;
    }
  }
}
// tile log the register-blocked kernels serve a pass with, 0 = none (the generic LDS-sweep kernel takes it)
// Strided passes: the longer the contiguous runs (2^M words), the closer the pass gets to the HBM stream rate — measured on
// 64 columns x 2^22 (tools/fft_lab.hip): 64-B runs (M = 4) 3.5 TB/s, 128-B runs 4.4 TB/s, 256-B runs 5.0 TB/s read + write.
// So passes of 6 layers and more use the 2^14 tile (M = 8..4), shorter ones the 2^11 tile (M >= 6).
uint32_t fft_pass_rb_tile_log(uint32_t W, uint32_t lo) {
  if (lo == 0) return (W >= 11 && W <= 13) ? W : 0u;
  if (W >= 6 && W <= 10 && lo >= 14 - W) return 14u;
  if (W >= 1 && W <= 7 && lo >= 11 - W) return 11u;
  return 0u;
}
void launch_fft_pass_rb(bool inverse, const FftPassArgs& a, uint32_t tile_log, uint32_t ntiles, uint32_t ncols, hipStream_t st) {
  if (inverse) launch_w<true>(a, tile_log, ntiles, ncols, st);
  else launch_w<false>(a, tile_log, ntiles, ncols, st);
}

}  // namespace cm
