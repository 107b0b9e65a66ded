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


// ---- variant 4: one node per QUAD of lanes (wide-and-short layers are one sequential Blake2s chain per node:
// spreading the 4 columns of the 4x4 state over 4 lanes shortens the chain ~3x).  Lane q of a quad holds state
// column q: a = v[q], b = v[4+q], c = v[8+q], d = v[12+q]; the diagonal step rotates b, c, d by 1, 2, 3 lanes
// with DPP quad_perm moves.  Every lane keeps all 16 message words (same addresses inside the quad: one fetch).
#define QP(p0, p1, p2, p3) ((p0) | ((p1) << 2) | ((p2) << 4) | ((p3) << 6))
#define quad_rot(x, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ uint32_t sel4(uint32_t q, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
  uint32_t lo = (q & 1u) ? x1 : x0, hi = (q & 1u) ? x3 : x2;
  return (q & 2u) ? hi : lo;
}
#define QG(x, y)                                \
  a = a + b + (x); d = rotr(d ^ a, 16);         \
  c = c + d;       b = rotr(b ^ c, 12);         \
  a = a + b + (y); d = rotr(d ^ a, 8);          \
  c = c + d;       b = rotr(b ^ c, 7);
#define QROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                      \
  QG(sel4(q, m[s0], m[s2], m[s4], m[s6]), sel4(q, m[s1], m[s3], m[s5], m[s7]))                            \
  b = quad_rot(b, QP(1, 2, 3, 0)); c = quad_rot(c, QP(2, 3, 0, 1)); d = quad_rot(d, QP(3, 0, 1, 2));      \
  QG(sel4(q, m[s8], m[s10], m[s12], m[s14]), sel4(q, m[s9], m[s11], m[s13], m[s15]))                      \
  b = quad_rot(b, QP(3, 0, 1, 2)); c = quad_rot(c, QP(2, 3, 0, 1)); d = quad_rot(d, QP(1, 2, 3, 0));
// h0 = h[q], h1 = h[4 + q] of the node's chaining value
__device__ __forceinline__ void b2s_compress_quad(uint32_t& h0, uint32_t& h1, const uint32_t (&m)[16], uint32_t q) {
  uint32_t a = h0, b = h1;
  uint32_t c = sel4(q, 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
  uint32_t d = sel4(q, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u);
  QROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  QROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  QROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  QROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  QROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  QROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  QROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  QROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  QROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  QROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  h0 ^= a ^ c;
  h1 ^= b ^ d;
}
__global__ void __launch_bounds__(256) k_layer_quad(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                    const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                    uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = t >> 2, q = t & 3u;
  if (i >= (1u << log_size)) return;  // whole quads drop out together
  uint32_t h0 = 0, h1 = 0;
  uint32_t m[16];
  if (prev) {
    const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)i * 16);
    uint4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
    m[0] = x0.x; m[1] = x0.y; m[2] = x0.z; m[3] = x0.w; m[4] = x1.x; m[5] = x1.y; m[6] = x1.z; m[7] = x1.w;
    m[8] = x2.x; m[9] = x2.y; m[10] = x2.z; m[11] = x2.w; m[12] = x3.x; m[13] = x3.y; m[14] = x3.z; m[15] = x3.w;
    b2s_compress_quad(h0, h1, m, q);
  }
  for (uint32_t c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < n_cols) ? cols[c0 + k][i] : 0u;
    b2s_compress_quad(h0, h1, m, q);
  }
  out[(size_t)i * 8 + q] = h0;
  out[(size_t)i * 8 + 4 + q] = h1;
}

__global__ void __launch_bounds__(256) k_layer_quad_pf(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                       const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                       uint32_t* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = t >> 2, q = t & 3u;
  if (i >= (1u << log_size)) return;
  uint32_t h0 = 0, h1 = 0;
  uint32_t m[16], nx[16];
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) nx[k] = (k < n_cols) ? cols[k][i] : 0u;
  if (prev) {
    const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)i * 16);
    uint4 x0 = p[0], x1 = p[1], x2 = p[2], x3 = p[3];
    m[0] = x0.x; m[1] = x0.y; m[2] = x0.z; m[3] = x0.w; m[4] = x1.x; m[5] = x1.y; m[6] = x1.z; m[7] = x1.w;
    m[8] = x2.x; m[9] = x2.y; m[10] = x2.z; m[11] = x2.w; m[12] = x3.x; m[13] = x3.y; m[14] = x3.z; m[15] = x3.w;
    b2s_compress_quad(h0, h1, m, q);
  }
  for (uint32_t c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = nx[k];
    if (c0 + 16 < n_cols) {
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) nx[k] = (c0 + 16 + k < n_cols) ? cols[c0 + 16 + k][i] : 0u;
    }
    b2s_compress_quad(h0, h1, m, q);
  }
  out[(size_t)i * 8 + q] = h0;
  out[(size_t)i * 8 + 4 + q] = h1;
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
      float ms = time_it([&] { hipLaunchKernelGGL(k_merkle_layer<false>, dim3(g256), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out0); });
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
    {
      CK(hipMemset(out1, 0, n * 32));
      float ms = time_it([&] { hipLaunchKernelGGL(k_layer_quad, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out1); });
      report("v4 quad-lane Blake2s", ms, out1);
    }
    {
      CK(hipMemset(out1, 0, n * 32));
      float ms = time_it([&] { hipLaunchKernelGGL(k_layer_quad_pf, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, 0, s.log, prev, d_ptrs, s.ncols, out1); });
      report("v5 quad + prefetch", ms, out1);
    }
    CK(hipFree(colbuf)); CK(hipFree(d_ptrs)); if (prev) CK(hipFree(prev)); CK(hipFree(out0)); CK(hipFree(out1));
  }
  return 0;
}
