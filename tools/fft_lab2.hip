// FFT lab 2 (development tool, round 6): radix-32 register blocking (E = 5, 512 threads on the 2^14 tile) against the shipped radix-16
// (E = 4, 1024 threads) for the strided passes of 9 / 10 layers — with E = 4 they are THREE rounds (4 + 1 + 4 / 4 + 2 + 4: the middle
// round of one or two layers still pays a full LDS exchange), with E = 5 two (5 + 4 / 5 + 5) — and for the fused kernel that ends
// the inverse transform of 2^21 rows and starts the extension (k_fft_fused_rb<9>).  Timing only (tables are memset): 64 columns.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cairo_m_amd/csrc tools/fft_lab2.hip -o tools/fft_lab2 -Lcairo_m_amd -l:libcairom_hip.so -Wl,-rpath,'$ORIGIN/../cairo_m_amd'
//   (kernels_fft.hip reads a tuning switch of the library: link against it)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "kernels_fft.hip"
using namespace cm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F>
static float time_us(F f, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3f / reps;
}
template <bool INV, int W, int TL, int E>
static void lp(const FftPassArgs& a, uint32_t ntiles, uint32_t ncols) {
  constexpr size_t lds = ((size_t)4 << TL) + ((size_t)4 << (TL - 5));
  static const hipError_t once = hipFuncSetAttribute((const void*)k_fft_pass_rb<INV, W, TL, E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)once;
  hipLaunchKernelGGL((k_fft_pass_rb<INV, W, TL, E>), dim3(ntiles, ncols), dim3(1u << (TL - E)), lds, 0, a);
}
int main() {
  const uint32_t R = 24, ncols = 64;
  uint32_t *buf, *buf2, *xtw, *ytw;
  CK(hipMalloc(&buf, (size_t)ncols * 4 << 22)); CK(hipMalloc(&buf2, (size_t)ncols * 4 << 22));
  CK(hipMalloc(&xtw, (size_t)4 << R)); CK(hipMalloc(&ytw, (size_t)4 << R));
  CK(hipMemset(buf, 1, (size_t)ncols * 4 << 22)); CK(hipMemset(buf2, 1, (size_t)ncols * 4 << 22));
  CK(hipMemset(xtw, 1, (size_t)4 << R)); CK(hipMemset(ytw, 1, (size_t)4 << R));
  auto table = [&](uint32_t* base, uint32_t n) {
    std::vector<uint32_t*> p(ncols);
    for (uint32_t c = 0; c < ncols; c++) p[c] = base + ((size_t)c << n);
    uint32_t** d; CK(hipMalloc(&d, ncols * 8)); CK(hipMemcpy(d, p.data(), ncols * 8, hipMemcpyHostToDevice));
    return d;
  };
  auto report = [&](const char* name, float us, uint32_t n, uint32_t W, double bytes_per_elem, double bf_per_elem_layer = 0.5) {
    const double elems = (double)ncols * (double)(1u << n);
    printf("%-46s %8.1f us  %6.2f TB/s  %6.2f T butterflies/s\n", name, us, elems * bytes_per_elem / (us * 1e-6) / 1e12,
           elems * bf_per_elem_layer * W / (us * 1e-6) / 1e12);
  };
  {   // strided passes on 2^22
    const uint32_t n = 22;
    uint32_t** d = table(buf, n);
    auto args = [&](uint32_t lo, uint32_t hi, uint32_t tl) {
      FftPassArgs a; a.src = (const uint32_t* const*)d; a.dst = d; a.xtw = xtw; a.ytw = ytw; a.R = R; a.n = n; a.lo = lo; a.hi = hi;
      a.M = tl - (hi - lo); a.in_len = 1u << n; a.scale = 1; return a;
    };
    const uint32_t nt14 = 1u << (n - 14), nt15 = 1u << (n - 15);
    { auto a = args(13, 22, 14); report("fwd strided 9  TL14 E4 (4+1+4, 1024 thr)", time_us([&] { lp<false, 9, 14, 4>(a, nt14, ncols); }), n, 9, 8);
      report("fwd strided 9  TL14 E5 (5+4, 512 thr)", time_us([&] { lp<false, 9, 14, 5>(a, nt14, ncols); }), n, 9, 8);
      report("inv strided 9  TL14 E4", time_us([&] { lp<true, 9, 14, 4>(a, nt14, ncols); }), n, 9, 8);
      report("inv strided 9  TL14 E5", time_us([&] { lp<true, 9, 14, 5>(a, nt14, ncols); }), n, 9, 8); }
    { auto a = args(12, 22, 14); report("fwd strided 10 TL14 E4 (4+2+4)", time_us([&] { lp<false, 10, 14, 4>(a, nt14, ncols); }), n, 10, 8);
      report("fwd strided 10 TL14 E5 (5+5)", time_us([&] { lp<false, 10, 14, 5>(a, nt14, ncols); }), n, 10, 8);
      report("inv strided 10 TL14 E4", time_us([&] { lp<true, 10, 14, 4>(a, nt14, ncols); }), n, 10, 8);
      report("inv strided 10 TL14 E5", time_us([&] { lp<true, 10, 14, 5>(a, nt14, ncols); }), n, 10, 8); }
    { auto a = args(12, 22, 15); report("fwd strided 10 TL15 E5 (5+5, 1024 thr, M=5)", time_us([&] { lp<false, 10, 15, 5>(a, nt15, ncols); }), n, 10, 8);
      report("inv strided 10 TL15 E5", time_us([&] { lp<true, 10, 15, 5>(a, nt15, ncols); }), n, 10, 8); }
    { auto a = args(13, 21, 14); report("fwd strided 8  TL14 E4 (4+4)", time_us([&] { lp<false, 8, 14, 4>(a, nt14, ncols); }), n, 8, 8);
      report("inv strided 8  TL14 E4", time_us([&] { lp<true, 8, 14, 4>(a, nt14, ncols); }), n, 8, 8); }
    { auto a = args(0, 12, 12); a.M = 0; report("fwd contiguous 12 TL12 E4", time_us([&] { lp<false, 12, 12, 4>(a, 1u << (n - 12), ncols); }), n, 12, 8);
      report("inv contiguous 12 TL12 E4", time_us([&] { lp<true, 12, 12, 4>(a, 1u << (n - 12), ncols); }), n, 12, 8); }
  }
  for (uint32_t n : {21u, 20u}) {   // the fused kernel: inverse [12, n) of 2^n in place + forward [12, n) of both halves of 2^(n+1)
    uint32_t** dco = table(buf, n);
    uint32_t** dld = table(buf2, n + 1 <= 22 ? n + 1 : 22);
    FftFusedArgs f;
    const uint32_t W = n - 12;
    f.inv.src = (const uint32_t* const*)dco; f.inv.dst = dco; f.inv.xtw = xtw; f.inv.ytw = ytw; f.inv.R = R; f.inv.n = n; f.inv.lo = 12; f.inv.hi = n;
    f.inv.M = 14 - W; f.inv.in_len = 1u << n; f.inv.scale = 3;
    f.fwd = f.inv; f.fwd.dst = dld; f.fwd.n = n + 1; f.fwd.in_len = 2u << n; f.fwd.scale = 1;
    const uint32_t nt = 1u << (n - 14);
    char name[96];
    if (n == 21) {
      snprintf(name, sizeof name, "fused 2^%u W=%u E4 (1024 thr)", n, W);
      report(name, time_us([&] { launch_fused_one<9, 4>(f, nt, ncols, 0); }), n, W, 16, 1.5);
      snprintf(name, sizeof name, "fused 2^%u W=%u E5 (512 thr)", n, W);
      report(name, time_us([&] { launch_fused_one<9, 5>(f, nt, ncols, 0); }), n, W, 16, 1.5);
    } else {
      snprintf(name, sizeof name, "fused 2^%u W=%u E4 (1024 thr)", n, W);
      report(name, time_us([&] { launch_fused_one<8, 4>(f, nt, ncols, 0); }), n, W, 16, 1.5);
      snprintf(name, sizeof name, "fused 2^%u W=%u E5 (512 thr)", n, W);
      report(name, time_us([&] { launch_fused_one<8, 5>(f, nt, ncols, 0); }), n, W, 16, 1.5);
    }
  }
  return 0;
}
