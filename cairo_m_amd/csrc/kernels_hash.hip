// Blake2s kernels for gfx950: mixed-degree Merkle layer hashing and proof-of-work grinding.
//
// Replaces (reference call sites): MerkleOps<Blake2sMerkleHasher>::commit_on_layer reached from
// `tree_builder.commit(channel)` (crates/prover/src/prover.rs:73, 82, 102) and
// `SimdBackend::grind` (prover.rs:90 and inside stwo `prove`, prover.rs:131).
//
// One thread = one tree node.  Column values of a node are read column-major: lane i reads
// cols[c][i], so a wave's 64 lanes read 256 contiguous bytes per column (coalesced); column
// pointers are wave-uniform (scalar loads).  Hash framing: state = 0; optional F(state, left||right);
// then one F per 16 column words (zero padded); t = f = 0 (see DESIGN.md "Merkle node framing").
#include <string.h>
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "kprof.hpp"
#include "blake2s_dev.hpp"
#include "merkle_kernels.hpp"

namespace cm {

// Proof of work: smallest nonce in [base, base + n) with trailing_zeros(F(digest, [lo,hi,0..])[0..16B]) >= bits.
// result initialised to ~0ull; atomicMin keeps the smallest hit.
struct GrindDigest { uint32_t w[8]; };   // the channel digest travels as a kernel argument: no host->device copy
__global__ void __launch_bounds__(256) k_grind(GrindDigest digest, uint32_t bits, uint64_t base, unsigned long long* result) {
  const uint64_t nonce = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h[8];
#pragma unroll
  for (int k = 0; k < 8; k++) h[k] = digest.w[k];
  uint32_t m[16] = {0};
  m[0] = (uint32_t)nonce;
  m[1] = (uint32_t)(nonce >> 32);
  b2s_compress(h, m);
  // trailing zeros of the first 16 bytes as LE u128
  uint32_t tz;
  if (h[0]) tz = __ffs(h[0]) - 1;
  else if (h[1]) tz = 32 + __ffs(h[1]) - 1;
  else if (h[2]) tz = 64 + __ffs(h[2]) - 1;
  else if (h[3]) tz = 96 + __ffs(h[3]) - 1;
  else tz = 128;
  if (tz >= bits) atomicMin(result, (unsigned long long)nonce);
}

// chan = {digest[8], n_sent}.  Blake2sChannel::mix_root then draw_felt (host twin: host_channel.hpp).
__global__ void k_chan_mix_root_draw(uint32_t* chan, const uint32_t* root, uint32_t* felt_out, uint32_t* root_log) {
  __shared__ uint32_t x8[8], felt[4];
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;   // one quad of lanes: half the latency of one thread
  chan_mix_root_draw_quad(threadIdx.x, chan, root, felt_out, root_log, x8, felt);
}

// Decommitment gather: out[q * width + w] = addrs[q][w]  (width 1 = column values, 8 = 32-byte hashes).
__global__ void k_gather_words(const uint32_t* const* addrs, uint32_t n, uint32_t width, uint32_t* out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  uint32_t q = t / width, w = t - q * width;
  out[t] = addrs[q][w];
}


__global__ void __launch_bounds__(64) k_gather_runs(const RowRun* __restrict__ runs, uint32_t* __restrict__ out) {
  const RowRun r = runs[blockIdx.x];
  for (uint32_t c = threadIdx.x; c < r.n_cols; c += blockDim.x) out[r.out_off + c] = r.d_cols[c][r.row];
}

// ================================================================= host wrappers
void merkle_layer(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                  uint32_t* d_out, hipStream_t st) {
  uint32_t n = 1u << log_size;
  // algorithmic bytes: column values once + 64 B of child hashes in, 32 B out per node
  KProfScope kp("k_merkle_layer", (4.0 * ncols + (d_prev ? 64.0 : 0.0) + 32.0) * (double)n, st,
                /* Blake2s compressions */ (double)n * ((d_prev ? 1.0 : 0.0) + (double)((ncols + 15) / 16)));
  hipLaunchKernelGGL(k_merkle_layer, dim3((n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
  CM_HIP(hipGetLastError());
}
void merkle_layer_quad(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                       uint32_t* d_out, hipStream_t st) {
  uint32_t n = 1u << log_size;
  KProfScope kp("k_merkle_layer_quad", (4.0 * ncols + (d_prev ? 64.0 : 0.0) + 32.0) * (double)n, st);
  hipLaunchKernelGGL(k_merkle_layer_quad, dim3((4 * n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
  CM_HIP(hipGetLastError());
}
void merkle_multi(const MerkleMultiArgs& a, double alg_bytes, hipStream_t st) {
  KProfScope kp("k_merkle_multi", alg_bytes, st);
  hipLaunchKernelGGL(k_merkle_multi, dim3((1u << a.top_log) / 256), dim3(256), 0, st, a);
  CM_HIP(hipGetLastError());
}
void chan_mix_root_draw(uint32_t* d_chan, const uint32_t* d_root, uint32_t* d_felt_out, uint32_t* d_root_log, hipStream_t st) {
  hipLaunchKernelGGL(k_chan_mix_root_draw, dim3(1), dim3(64), 0, st, d_chan, d_root, d_felt_out, d_root_log);
  CM_HIP(hipGetLastError());
}
void gather_runs(const RowRun* d_runs, uint32_t n_runs, uint32_t* d_out, hipStream_t st) {
  if (!n_runs) return;
  hipLaunchKernelGGL(k_gather_runs, dim3(n_runs), dim3(64), 0, st, d_runs, d_out);
  CM_HIP(hipGetLastError());
}
// ticket counters of k_merkle_top: a zeroed ring per host thread (trees can be in flight on several streams of one
// prover thread; every launch leaves its counter at zero again)
static uint32_t* next_ticket(hipStream_t st) {
  static thread_local uint32_t* ring = nullptr;
  static thread_local uint32_t pos = 0;
  constexpr uint32_t N = 256;
  if (!ring) {
    CM_HIP(hipMalloc((void**)&ring, N * 4));
    // hipMemset runs on the NULL stream and may return before it has executed; the prover's streams are non-blocking
    // (no implicit ordering with the NULL stream), so zero the ring on the launching stream and wait — once per thread.
    // (Recycled device memory is not zero: with the plain hipMemset a second prover thread drew garbage tickets.)
    CM_HIP(hipMemsetAsync(ring, 0, N * 4, st));
    CM_HIP(hipStreamSynchronize(st));
  }
  return ring + (pos++ % N);
}
void merkle_top(MerkleTopArgs& a, hipStream_t st) {
  CM_CHECK(a.top_log >= 9 && a.top_log <= MERKLE_TOP_MAX_LOG, "merkle_top: bad layer range");
  a.ticket = next_ticket(st);
  KProfScope kp("k_merkle_top", 0.0, st);
  hipLaunchKernelGGL(k_merkle_top, dim3(1u << (a.top_log - 8)), dim3(256), 0, st, a);
  CM_HIP(hipGetLastError());
}
void merkle_tail(const MerkleTailArgs& a, hipStream_t st) {
  KProfScope kp("k_merkle_tail", 0.0, st);
  hipLaunchKernelGGL(k_merkle_tail, dim3(1), dim3(1024), 0, st, a);
  CM_HIP(hipGetLastError());
}
uint64_t grind_gpu(const uint8_t digest[32], uint32_t bits, hipStream_t st) {
  DevBuf d_res(8);
  GrindDigest dg;
  memcpy(dg.w, digest, 32);
  unsigned long long* res = (unsigned long long*)(pinned_words() + PIN_NONCE);   // pinned: a truly asynchronous 8-byte read-back
  // expected nonce ~ 2^bits: start with 8x that and grow, instead of always hashing 2^22 candidates
  uint64_t batch = 1ull << (bits + 3 < 12 ? 12 : (bits + 3 > 22 ? 22 : bits + 3));
  for (uint64_t base = 0;; base += batch, batch = batch < (1ull << 22) ? batch * 2 : batch) {
    CM_HIP(hipMemsetAsync(d_res.p, 0xFF, 8, st));   // ~0ull: atomicMin keeps the smallest nonce
    hipLaunchKernelGGL(k_grind, dim3(batch / 256), dim3(256), 0, st, dg, bits, base, (unsigned long long*)d_res.p);
    CM_HIP(hipGetLastError());
    CM_HIP(hipMemcpyAsync(res, d_res.p, 8, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    if (*res != ~0ull) return *res;
    CM_CHECK(base < (1ull << 40), "grind: no nonce found");
  }
}
void gather_words(const uint32_t* const* d_addrs, uint32_t n, uint32_t width, uint32_t* d_out, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_gather_words, dim3((n * width + 255) / 256), dim3(256), 0, st, d_addrs, n, width, d_out);
  CM_HIP(hipGetLastError());
}

}  // namespace cm
