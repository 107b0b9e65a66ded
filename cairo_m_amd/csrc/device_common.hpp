// Shared device helpers (wave64 reductions) and host-side HIP error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <stdexcept>
#include "field.hpp"

namespace cm {

// Column pointers come out of device tables as generic pointers; the columns live in HBM, so say so: global loads / stores
// instead of flat ones (flat instructions also occupy the LDS path and its counter).
typedef const __attribute__((address_space(1))) uint32_t* cm_gptr;
typedef __attribute__((address_space(1))) uint32_t* cm_gptr_w;
#define CM_GCOL(p) ((cm::cm_gptr)(p))
#define CM_GCOL_W(p) ((cm::cm_gptr_w)(p))

// ---- wave64 / block reductions over M31 / QM31 sums -------------------------------------------
__device__ __forceinline__ M31 wave_reduce_m31(M31 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = v + M31((uint32_t)__shfl_down((int)v.v, off, 64));
  return v;
}
__device__ __forceinline__ QM31 wave_reduce_qm31(QM31 q) {
  return QM31(wave_reduce_m31(q.a.a), wave_reduce_m31(q.a.b), wave_reduce_m31(q.b.a), wave_reduce_m31(q.b.b));
}
// result valid in thread 0; blockDim.x must be a multiple of 64 and <= 1024
__device__ __forceinline__ QM31 block_reduce_qm31(QM31 q) {
  __shared__ uint32_t red[16 * 4];
  q = wave_reduce_qm31(q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) q.to_u32(red + 4 * w);
  __syncthreads();
  QM31 r;
  if (threadIdx.x == 0) {
    r = QM31::from_u32(red);
    for (int i = 1; i < nw; i++) r += QM31::from_u32(red + 4 * i);
  }
  return r;
}

// Row `offset` trace-steps away on bit-reversed storage of log n (trace domain log = trace_log <= n).
CM_HD uint32_t shifted_row(uint32_t r, uint32_t n, uint32_t trace_log, int offset) {
  uint32_t i = bit_reverse(r, n);
  uint32_t half = 1u << (n - 1);
  uint32_t mod_mask = (n + 1 >= 32) ? 0xffffffffu : ((1u << (n + 1)) - 1);
  uint32_t e = i < half ? (1u + 4u * i) : (0u - (1u + 4u * (i - half)));
  e += (uint32_t)offset * (1u << (n + 1 - trace_log));
  e &= mod_mask;
  uint32_t j = ((e & 3u) == 1u) ? (e - 1u) / 4u : half + (((mod_mask + 1u) - e - 1u) & mod_mask) / 4u;
  return bit_reverse(j, n);
}


// ---- host error handling -------------------------------------------------------------------------
struct CmError : std::runtime_error {
  int code;
  CmError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define CM_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      throw ::cm::CmError(2, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + \
                                 std::to_string(__LINE__));                                \
  } while (0)
#define CM_CHECK(cond, msg)                                 \
  do {                                                      \
    if (!(cond)) throw ::cm::CmError(1, std::string(msg)); \
  } while (0)

}  // namespace cm
