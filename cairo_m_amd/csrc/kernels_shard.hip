// Data-movement kernels of the intra-proof sharded prover (prover_sharded.inc): packing row slices of columns for the
// all-to-all that turns component (column) ownership into row-range ownership, and the element-wise reductions that
// follow an exchange.  Pure HBM streaming, coalesced dwords / 16-byte words.
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "shard_kernels.hpp"

namespace cm {

// one block column (blockIdx.y) per segment, grid-stride over its words
__global__ void __launch_bounds__(256) k_copy_segments(const CopySeg* __restrict__ segs) {
  const CopySeg s = segs[blockIdx.y];
  const uint32_t* __restrict__ src = s.src;
  uint32_t* __restrict__ dst = s.dst;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.words; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// out[i] = sum over k < n_copies of in[k * stride + i]   (plain u32 adds: histogram counts) or the same sum mod P
template <bool MODULAR>
__global__ void __launch_bounds__(256) k_sum_copies(const uint32_t* __restrict__ in, uint32_t n_copies, uint64_t stride, uint64_t words,
                                                    uint32_t* __restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x) {
    if (MODULAR) {
      M31 acc(in[i]);
      for (uint32_t k = 1; k < n_copies; k++) acc = acc + M31(in[k * stride + i]);
      out[i] = acc.v;
    } else {
      uint32_t acc = in[i];
      for (uint32_t k = 1; k < n_copies; k++) acc += in[k * stride + i];
      out[i] = acc;
    }
  }
}

__global__ void __launch_bounds__(256) k_pack_parity(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t half_len, uint32_t parity) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < half_len) dst[i] = src[2 * i + parity];
}
__global__ void __launch_bounds__(256) k_halo_build(const uint32_t* __restrict__ even_half, const uint32_t* __restrict__ odd_half, uint32_t even_src,
                                                    uint32_t odd_src, uint32_t row0, uint32_t len, uint32_t n, uint32_t trace_log, uint32_t log_ranks,
                                                    uint32_t* __restrict__ out, uint32_t* __restrict__ err) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= len) return;
  const uint32_t pr = shifted_row(row0 + q, n, trace_log, -1);          // global position of the previous row
  const uint32_t slice_log = n - log_ranks;
  const uint32_t owner = pr >> slice_log, qp = pr & ((1u << slice_log) - 1);
  const bool odd = (q & 1u) != 0;
  if (owner != (odd ? odd_src : even_src) || (qp & 1u) != (q & 1u)) { *err = 1; return; }
  out[q] = (odd ? odd_half : even_half)[qp >> 1];
}
void pack_parity(const uint32_t* d_src, uint32_t* d_dst, uint32_t half_len, uint32_t parity, hipStream_t st) {
  if (!half_len) return;
  hipLaunchKernelGGL(k_pack_parity, dim3((half_len + 255) / 256), dim3(256), 0, st, d_src, d_dst, half_len, parity);
  CM_HIP(hipGetLastError());
}
void halo_build(const uint32_t* d_even_half, const uint32_t* d_odd_half, uint32_t even_src, uint32_t odd_src, uint32_t row0, uint32_t len,
                uint32_t n, uint32_t trace_log, uint32_t log_ranks, uint32_t* d_out, uint32_t* d_err, hipStream_t st) {
  if (!len) return;
  hipLaunchKernelGGL(k_halo_build, dim3((len + 255) / 256), dim3(256), 0, st, d_even_half, d_odd_half, even_src, odd_src, row0, len, n, trace_log,
                     log_ranks, d_out, d_err);
  CM_HIP(hipGetLastError());
}

void copy_segments(const std::vector<CopySeg>& segs, hipStream_t st) {
  if (segs.empty()) return;
  uint64_t mx = 0;
  for (auto& s : segs) mx = s.words > mx ? s.words : mx;
  if (!mx) return;
  DevBuf d = upload(segs, st);
  uint32_t bx = (uint32_t)((mx + 1023) / 1024);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(k_copy_segments, dim3(bx, (uint32_t)segs.size()), dim3(256), 0, st, d.as<CopySeg>());
  CM_HIP(hipGetLastError());
  // (no host synchronisation: the segment table goes back to the calling thread's device pool, whose blocks are reused in stream
  // order on that thread's streams — the rule the single-GPU prover's early teardown relies on; round 6: this call used to stop the
  // stream ~10 times per sharded proof)
}
void sum_copies(const uint32_t* d_in, uint32_t n_copies, uint64_t stride, uint64_t words, uint32_t* d_out, bool modular, hipStream_t st) {
  if (!words) return;
  uint32_t blocks = (uint32_t)((words + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (modular) hipLaunchKernelGGL(k_sum_copies<true>, dim3(blocks), dim3(256), 0, st, d_in, n_copies, stride, words, d_out);
  else hipLaunchKernelGGL(k_sum_copies<false>, dim3(blocks), dim3(256), 0, st, d_in, n_copies, stride, words, d_out);
  CM_HIP(hipGetLastError());
}

}  // namespace cm
