// Shared argument block of the circle-FFT butterfly passes.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace cm {

// One launch applies butterfly layers [lo, hi) of a size-2^n transform to every column of a batch.
// Tile = 2^W values of index bits [lo,hi)  x  2^M consecutive low indices (M = 0 when lo == 0), W + M = 11, 12 or 14.
// INVERSE: ibutterfly (a+b, (a-b)*itw), layers ascending.  Forward: (a+b*tw, a-b*tw), descending.
// in_len: logical input length; reads at index >= in_len return 0 (zero-extension => LDE).
struct FftPassArgs {
  const uint32_t* const* src;
  uint32_t* const* dst;
  const uint32_t* xtw;  // (i)xtw table
  const uint32_t* ytw;  // (i)ytw table
  uint32_t R;           // root log of the twiddle tables
  uint32_t n;           // transform log size
  uint32_t lo, hi;      // layer range
  uint32_t M;           // log of contiguous low run per tile
  uint32_t in_len;      // logical input length per column
  uint32_t scale;       // multiply outputs by this (1 = none); used for 1/N on the last inverse pass
};

// register-blocked passes (kernels_fft.hip): tile log (11 / 12 / 14) serving a pass of W layers starting at layer lo,
// 0 = not served (the generic LDS-sweep kernel of kernels_poly.hip takes it).  M = tile_log - W (0 for lo == 0).
uint32_t fft_pass_rb_tile_log(uint32_t W, uint32_t lo);
void launch_fft_pass_rb(bool inverse, const FftPassArgs& a, uint32_t tile_log, uint32_t ntiles, uint32_t ncols, hipStream_t st);

}  // namespace cm
