// FFT pass lab (development tool): times k_fft_pass_rb on 64 columns of 2^22 words and, when built with
// -DCM_FFT_ABL_NO_TW / -DCM_FFT_ABL_NO_LDS, the same kernel without twiddle fetches / LDS exchanges (wrong results,
// timing only) to see what a butterfly's ~55 cycles are made of.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cairo_m_amd/csrc [-DCM_FFT_ABL_...] tools/fft_lab.hip -o tools/fft_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "kernels_fft.hip"
using namespace cm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main() {
  const uint32_t n = 22, ncols = 64, R = 24;
  uint32_t *buf, *xtw, *ytw;
  CK(hipMalloc(&buf, (size_t)ncols * 4 << n));
  CK(hipMalloc(&xtw, (size_t)4 << R)); CK(hipMalloc(&ytw, (size_t)4 << R));
  CK(hipMemset(buf, 1, (size_t)ncols * 4 << n)); CK(hipMemset(xtw, 1, (size_t)4 << R)); CK(hipMemset(ytw, 1, (size_t)4 << R));
  std::vector<uint32_t*> ptrs(ncols);
  for (uint32_t c = 0; c < ncols; c++) ptrs[c] = buf + ((size_t)c << n);
  uint32_t** d_ptrs;
  CK(hipMalloc(&d_ptrs, ncols * 8));
  CK(hipMemcpy(d_ptrs, ptrs.data(), ncols * 8, hipMemcpyHostToDevice));
  struct Case { const char* name; bool inv; uint32_t lo, hi, tl; } cases[] = {
      {"fwd  contiguous 11 (TL11 r8)", false, 0, 11, 11}, {"fwd  strided 5 (TL11 M=6)", false, 11, 16, 11}, {"fwd  strided 6 (TL11 M=5)", false, 16, 22, 11},
      {"inv  contiguous 11 (TL11 r8)", true, 0, 11, 11}, {"inv  strided 5 (TL11 M=6)", true, 11, 16, 11},
      {"fwd  contiguous 12 (TL12 r16)", false, 0, 12, 12}, {"inv  contiguous 12 (TL12 r16)", true, 0, 12, 12},
      {"fwd  contiguous 13 (TL13 r32)", false, 0, 13, 13}, {"inv  contiguous 13 (TL13 r32)", true, 0, 13, 13},
      {"fwd  strided 9 (TL14 M=5)", false, 13, 22, 14}, {"inv  strided 9 (TL14 M=5)", true, 13, 22, 14},
      {"fwd  strided 8 (TL14 M=6)", false, 13, 21, 14}, {"inv  strided 8 (TL14 M=6)", true, 13, 21, 14},
      {"fwd  strided 7 (TL14 M=7)", false, 13, 20, 14}, {"fwd  strided 6 (TL14 M=8)", false, 13, 19, 14},
      {"fwd  strided 7 (TL11 M=4)", false, 13, 20, 11}, {"fwd  strided 6 (TL11 M=5)", false, 13, 19, 11}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& c : cases) {
    FftPassArgs a;
    a.src = (const uint32_t* const*)d_ptrs; a.dst = d_ptrs; a.xtw = xtw; a.ytw = ytw; a.R = R; a.n = n; a.lo = c.lo; a.hi = c.hi;
    uint32_t W = c.hi - c.lo;
    const uint32_t rb = c.tl;
    a.M = rb - W; a.in_len = 1u << n; a.scale = 1;
    uint32_t ntiles = 1u << (n - rb);
    launch_fft_pass_rb(c.inv, a, rb, ntiles, ncols, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; i++) launch_fft_pass_rb(c.inv, a, rb, ntiles, ncols, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double elems = (double)ncols * (1u << n), bf = elems / 2 * W;
    printf("%-30s %8.1f us  %6.2f TB/s (r+w)  %6.1f cycles/wave-butterfly\n", c.name, ms * 1e3, elems * 8 / (ms * 1e-3) / 1e12,
           (ms * 1e-3) * 1024 * 1.9e9 / (bf / 64));
  }
  return 0;
}
