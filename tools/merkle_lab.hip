// Merkle-layer kernel lab (development tool, not part of the product): times the shipped k_merkle_layer
// against experimental variants on the layer shapes the fibonacci 2^22 proof produces, and checks that
// every variant writes byte-identical hashes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cairo_m_amd/csrc tools/merkle_lab.hip -o tools/merkle_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "merkle_kernels.hpp"

using namespace cm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---- variant 1: software prefetch of the next 16-column chunk ------------------------------------------
template <bool STAGE>
__global__ void __launch_bounds__(256) k_layer_pf(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                  const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                  uint32_t* __restrict__ out) {
  __shared__ uint4 stage[STAGE ? 256 * 4 : 1];
  const uint32_t tid = threadIdx.x;
  const uint32_t n = 1u << log_size;
  const uint32_t blk0 = blockIdx.x * 256;
  const uint32_t i = blk0 + tid;
  uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t m[16], nx[16];
  if (n_cols) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) nx[k] = (k < n_cols) ? cols[k][i] : 0u;
  }
  if (prev) {
    if (STAGE) {
      const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)blk0 * 16);
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        uint32_t q = k * 256 + tid, node = q >> 2, part = q & 3;
        stage[node * 4 + ((part + (node >> 2)) & 3)] = p[q];
      }
      __syncthreads();
      uint4 a = stage[tid * 4 + ((0 + (tid >> 2)) & 3)], b = stage[tid * 4 + ((1 + (tid >> 2)) & 3)];
      uint4 c = stage[tid * 4 + ((2 + (tid >> 2)) & 3)], d = stage[tid * 4 + ((3 + (tid >> 2)) & 3)];
      m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
      m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
    } else {
      const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)i * 16);
      uint4 a = p[0], b = p[1], c = p[2], d = p[3];
      m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
      m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
    }
    b2s_compress(h, m);
  }
  for (uint32_t c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = nx[k];
    if (c0 + 16 < n_cols) {
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) nx[k] = (c0 + 16 + k < n_cols) ? cols[c0 + 16 + k][i] : 0u;
    }
    b2s_compress(h, m);
  }
  if (STAGE) {
    __syncthreads();
    stage[tid * 2 + 0] = make_uint4(h[0], h[1], h[2], h[3]);
    stage[tid * 2 + 1] = make_uint4(h[4], h[5], h[6], h[7]);
    __syncthreads();
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)blk0 * 8);
    o[tid] = stage[tid];
    o[256 + tid] = stage[256 + tid];
  } else {
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
  }
}

// ---- variant 2: two nodes per thread (two independent compression chains interleaved by the scheduler) ----
__global__ void __launch_bounds__(256) k_layer_x2(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                  const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                  uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;  // nodes 2t, 2t+1
  uint32_t h0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t m0[16], m1[16];
  if (prev) {
    const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)t * 32);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], hh = p[7];
    m0[0] = a.x; m0[1] = a.y; m0[2] = a.z; m0[3] = a.w; m0[4] = b.x; m0[5] = b.y; m0[6] = b.z; m0[7] = b.w;
    m0[8] = c.x; m0[9] = c.y; m0[10] = c.z; m0[11] = c.w; m0[12] = d.x; m0[13] = d.y; m0[14] = d.z; m0[15] = d.w;
    m1[0] = e.x; m1[1] = e.y; m1[2] = e.z; m1[3] = e.w; m1[4] = f.x; m1[5] = f.y; m1[6] = f.z; m1[7] = f.w;
    m1[8] = g.x; m1[9] = g.y; m1[10] = g.z; m1[11] = g.w; m1[12] = hh.x; m1[13] = hh.y; m1[14] = hh.z; m1[15] = hh.w;
    b2s_compress(h0, m0);
    b2s_compress(h1, m1);
  }
  for (uint32_t c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      uint2 v = (c0 + k < n_cols) ? reinterpret_cast<const uint2*>(cols[c0 + k])[t] : make_uint2(0, 0);
      m0[k] = v.x; m1[k] = v.y;
    }
    b2s_compress(h0, m0);
    b2s_compress(h1, m1);
  }
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)t * 16);
  o[0] = make_uint4(h0[0], h0[1], h0[2], h0[3]);
  o[1] = make_uint4(h0[4], h0[5], h0[6], h0[7]);
  o[2] = make_uint4(h1[0], h1[1], h1[2], h1[3]);
  o[3] = make_uint4(h1[4], h1[5], h1[6], h1[7]);
}

// ---- variant 3: wide-and-short layers (few nodes, hundreds of columns): prefetch inside one thread ---------
__global__ void __launch_bounds__(64) k_layer_wide(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                   const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                   uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= (1u << log_size)) return;
  uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t m[16], nx[16];
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) nx[k] = (k < n_cols) ? cols[k][i] : 0u;
  if (prev) {
    const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)i * 16);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
    m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
    b2s_compress(h, m);
  }
  for (uint32_t c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = nx[k];
    if (c0 + 16 < n_cols) {
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) nx[k] = (c0 + 16 + k < n_cols) ? cols[c0 + 16 + k][i] : 0u;
    }
    b2s_compress(h, m);
  }
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  o[0] = make_uint4(h[0], h[1], h[2], h[3]);
  o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i * 2654435761u + seed;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = x & 0x7fffffffu;
}

struct Shape { const char* name; uint32_t log, ncols; bool prev; };

template <class F>
static float time_it(F&& launch, int iters = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; i++) launch();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main() {
  Shape shapes[] = {
      {"trace leaf   2^22 x 42 cols      ", 22, 42, false},
      {"trace L-1    2^21 x 76 cols +prev", 21, 76, true},
      {"inter leaf   2^22 x 24 cols      ", 22, 24, false},
      {"fri leaf     2^23 x 4 cols       ", 23, 4, false},
      {"inner        2^22 children only  ", 22, 0, true},
      {"inner+4      2^22 x 4 cols +prev ", 22, 4, true},
      {"inner        2^20 children only  ", 20, 0, true},
      {"wide         2^5  x 1040 cols+prev", 5, 1040, true},
      {"wide         2^10 x 443 cols+prev", 10, 443, true},
      {"wide         2^7  x 200 cols+prev", 7, 200, true},
  };
  for (const Shape& s : shapes) {
    const size_t n = (size_t)1 << s.log;
    uint32_t* colbuf = nullptr;
    CK(hipMalloc(&colbuf, (s.ncols ? s.ncols : 1) * n * 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((s.ncols * n + 255) / 256)), dim3(256), 0, 0, colbuf, s.ncols * n, 12345u);
    std::vector<const uint32_t*> ptrs(s.ncols ? s.ncols : 1);
    for (uint32_t c = 0; c < s.ncols; c++) ptrs[c] = colbuf + c * n;
    const uint32_t** d_ptrs = nullptr;
    CK(hipMalloc(&d_ptrs, ptrs.size() * 8));
    CK(hipMemcpy(d_ptrs, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice));
    uint32_t* prev = nullptr;
    if (s.prev) {
      CK(hipMalloc(&prev, n * 64));
      hipLaunchKernelGGL(k_fill, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, prev, n * 16, 777u);
    }
    uint32_t *out0 = nullptr, *out1 = nullptr;
    CK(hipMalloc(&out0, n * 32));
    CK(hipMalloc(&out1, n * 32));
    CK(hipDeviceSynchronize());
    const uint64_t compr = n * ((s.prev ? 1 : 0) + (s.ncols + 15) / 16);
    const double bytes = (4.0 * s.ncols + (s.prev ? 64 : 0) + 32) * (double)n;
    std::vector<uint32_t> ref(n * 8), got(n * 8);
    auto report = [&](const char* vname, float ms, uint32_t* outp) {
      CK(hipMemcpy(got.data(), outp, n * 32, hipMemcpyDeviceToHost));
      bool same = memcmp(got.data(), ref.data(), n * 32) == 0;
      printf("  %-22s %9.1f us  %6.1f Gcompr/s  %7.1f GB/s  %s\n", vname, ms * 1e3, compr / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9,
             same ? "ok" : "MISMATCH");
    };
    printf("%s  (%llu compressions)\n", s.name, (unsigned long long)compr);
    const unsigned g256 = (unsigned)((n + 255) / 256);
    {
      float ms = time_it([&] { hipLaunchKernelGGL(k_merkle_layer, dim3(g256), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out0); });
      CK(hipMemcpy(ref.data(), out0, n * 32, hipMemcpyDeviceToHost));
      report("v0 shipped", ms, out0);
    }
    if (n >= 256) {
      CK(hipMemset(out1, 0, n * 32));
      float ms = time_it([&] { hipLaunchKernelGGL(k_layer_pf<true>, dim3(g256), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out1); });
      report("v1 prefetch+stage", ms, out1);
      CK(hipMemset(out1, 0, n * 32));
      ms = time_it([&] { hipLaunchKernelGGL(k_layer_pf<false>, dim3(g256), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out1); });
      report("v1 prefetch direct", ms, out1);
    }
    if (n >= 512) {
      CK(hipMemset(out1, 0, n * 32));
      float ms = time_it([&] { hipLaunchKernelGGL(k_layer_x2, dim3((unsigned)(n / 512)), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out1); });
      report("v2 two nodes/thread", ms, out1);
    }
    {
      CK(hipMemset(out1, 0, n * 32));
      float ms = time_it([&] { hipLaunchKernelGGL(k_layer_wide, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, s.log, prev, d_ptrs, s.ncols, out1); });
      report("v3 wave-blocks+prefetch", ms, out1);
    }
    CK(hipFree(colbuf)); CK(hipFree(d_ptrs)); if (prev) CK(hipFree(prev)); CK(hipFree(out0)); CK(hipFree(out1));
  }
  return 0;
}
