// Internal C++ interface of the HIP engine: host-side launch wrappers shared by the C ABI
// (capi.hip) and the prover driver (prover.hip).  All pointer-array arguments named d_* are
// DEVICE arrays of device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "field.hpp"
#include "device_common.hpp"

namespace cm {

struct Twiddles {
  uint32_t R = 0;  // tables cover CanonicCoset(R).circle_domain() and every smaller canonic domain
  uint32_t *xtw = nullptr, *ixtw = nullptr, *ytw = nullptr, *iytw = nullptr;
};
Twiddles* twiddles_create(uint32_t R, hipStream_t st);
void twiddles_destroy(Twiddles* t);

// in-place IFFT of ncols columns of 2^n (bit-reversed evals -> coefficients)
void interpolate(uint32_t* const* d_cols, uint32_t ncols, uint32_t n, const Twiddles& tw, hipStream_t st);
void interpolate_oop(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n, const Twiddles& tw,
                     hipStream_t st);
// coefficients (2^n_in each) -> evaluations on the canonic domain of log n_out (n_out >= n_in)
void evaluate(const uint32_t* const* d_src, uint32_t* const* d_dst, uint32_t ncols, uint32_t n_in, uint32_t n_out,
              const Twiddles& tw, hipStream_t st);
void bit_reverse_columns(uint32_t* const* d_cols, uint32_t ncols, uint32_t n, hipStream_t st);

// eval_at_point for a batch of equal-size coefficient columns at ONE point.
// d_scratch: >= 4*(2^10 + 2^max(0,n-10)) + 4*ncols*2^max(0,n-10) + 32*4 u32.  d_out: 4*ncols u32 (device).
size_t eval_at_point_scratch_words(uint32_t ncols, uint32_t n);
void eval_at_point_batch(const uint32_t* const* d_coeffs, uint32_t ncols, uint32_t n, const QM31& px, const QM31& py,
                         uint32_t* d_scratch, uint32_t* d_out, hipStream_t st);

// Merkle
void merkle_layer(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                  uint32_t* d_out, hipStream_t st);
uint64_t grind_gpu(const uint8_t digest[32], uint32_t bits, hipStream_t st);
void gather_hashes(const uint32_t* const* d_layers, const uint32_t* d_layer_idx, const uint32_t* d_node_idx, uint32_t n,
                   uint32_t* d_out, hipStream_t st);
void gather_values(const uint32_t* const* d_cols, const uint32_t* d_col_idx, const uint32_t* d_row_idx, uint32_t n,
                   uint32_t* d_out, hipStream_t st);

// simple RAII device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() {}
  explicit DevBuf(size_t b) { alloc(b); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t b) {
    release();
    if (b == 0) b = 4;
    CM_HIP(hipMalloc(&p, b));
    bytes = b;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  uint32_t* u32() const { return (uint32_t*)p; }
  template <class T> T* as() const { return (T*)p; }
};
template <class T>
inline DevBuf upload(const std::vector<T>& v, hipStream_t st) {
  DevBuf b(v.size() * sizeof(T));
  (void)st;
  if (!v.empty()) CM_HIP(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return b;
}

}  // namespace cm
