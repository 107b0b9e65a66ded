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
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "kprof.hpp"
#include "blake2s_dev.hpp"

namespace cm {

// hashes[i] = hash_node(children (prev[2i], prev[2i+1]) if prev != null, cols[*][i])
__global__ void __launch_bounds__(256) k_merkle_layer(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                      const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                      uint32_t* __restrict__ out) {
  // The 64 B of child hashes per node (and the 32 B result) are moved with wave-contiguous 16-byte
  // accesses and re-distributed through LDS: a direct per-lane read would touch every cache line of the
  // wave's 4 KiB window four times.  Slot rotation by (node >> 2) keeps the 128-bit LDS reads conflict-free.
  __shared__ uint4 stage[256 * 4];
  const uint32_t tid = threadIdx.x;
  const uint32_t n = 1u << log_size;
  const uint32_t blk0 = blockIdx.x * 256;
  const uint32_t i = blk0 + tid;
  const bool full_block = blk0 + 256 <= n;
  uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t m[16];
  if (prev) {
    if (full_block) {
      const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)blk0 * 16);
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        uint32_t q = k * 256 + tid, node = q >> 2, part = q & 3;
        stage[node * 4 + ((part + (node >> 2)) & 3)] = p[q];
      }
      __syncthreads();
      uint4 a = stage[tid * 4 + ((0 + (tid >> 2)) & 3)], b = stage[tid * 4 + ((1 + (tid >> 2)) & 3)];
      uint4 c = stage[tid * 4 + ((2 + (tid >> 2)) & 3)], d = stage[tid * 4 + ((3 + (tid >> 2)) & 3)];
      m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
      m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
      m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
      m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
      b2s_compress(h, m);
    } else if (i < n) {
      const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)i * 16);
      uint4 a = p[0], b = p[1], c = p[2], d = p[3];
      m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
      m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
      m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
      m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
      b2s_compress(h, m);
    }
  }
  if (i < n) {
    for (uint32_t c0 = 0; c0 < n_cols; c0 += 16) {
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < n_cols) ? cols[c0 + k][i] : 0u;
      b2s_compress(h, m);
    }
  }
  if (full_block) {
    __syncthreads();
    stage[tid * 2 + 0] = make_uint4(h[0], h[1], h[2], h[3]);
    stage[tid * 2 + 1] = make_uint4(h[4], h[5], h[6], h[7]);
    __syncthreads();
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)blk0 * 8);
    o[tid] = stage[tid];
    o[256 + tid] = stage[256 + tid];
  } else if (i < n) {
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
  }
}

// K consecutive layers (top_log, top_log-1, ..., top_log-K+1) in one launch.  A block hashes 256 nodes of the
// top layer, keeps them in LDS, then 128 parents, 64 grand-parents, ...  Every layer is written to HBM (the
// decommitment gathers need it) but intermediate layers are never re-read from HBM, and a 2^22 tree needs
// 5 launches instead of 16.  Mid-size layers are launch-latency-bound as separate kernels.
__global__ void __launch_bounds__(256) k_merkle_multi(MerkleMultiArgs a) {
  __shared__ uint32_t bufA[256 * 8];
  __shared__ uint32_t bufB[128 * 8];
  uint32_t* buf[2];
  buf[0] = bufA;
  buf[1] = bufB;
  const uint32_t tid = threadIdx.x;
  int cur = 0;
  uint32_t active = 256;  // nodes of the current level handled by this block
#pragma unroll 1
  for (uint32_t lv = 0; lv < a.n_levels; lv++, active >>= 1) {
    const uint32_t log = a.top_log - lv;
    const uint32_t node0 = blockIdx.x * active;  // first node of this block at this level
    if (tid < active) {
      const uint32_t i = node0 + tid;
      uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t m[16];
      if (lv > 0 || a.prev) {
        const uint32_t* p = (lv == 0) ? a.prev + (size_t)i * 16 : buf[cur ^ 1] + tid * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = p[k];
        b2s_compress(h, m);
      }
      const uint32_t c_begin = a.col_begin[lv], c_end = a.col_end[lv];
      for (uint32_t c0 = c_begin; c0 < c_end; c0 += 16) {
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < c_end) ? a.cols[c0 + k][i] : 0u;
        b2s_compress(h, m);
      }
      uint4* o = reinterpret_cast<uint4*>(a.layers[lv] + (size_t)i * 8);
      o[0] = make_uint4(h[0], h[1], h[2], h[3]);
      o[1] = make_uint4(h[4], h[5], h[6], h[7]);
#pragma unroll
      for (int k = 0; k < 8; k++) buf[cur][tid * 8 + k] = h[k];
    }
    (void)log;
    __syncthreads();
    cur ^= 1;
  }
}

// All layers 2^top_log .. 2^0 of a tree in ONE launch (one 1024-thread block): the small layers are pure
// launch/dependency latency as separate kernels (26 trees x 10 layers per proof).  Hashes of the layer being
// consumed stay in LDS; every layer is still written to HBM for the decommitment gathers.
__global__ void __launch_bounds__(1024) k_merkle_tail(MerkleTailArgs a) {
  __shared__ uint32_t bufA[1024 * 8];
  __shared__ uint32_t bufB[512 * 8];
  uint32_t* buf[2];  // layer top -> A (<= 1024 nodes), top-1 -> B (<= 512), top-2 -> A, ...
  buf[0] = bufA;
  buf[1] = bufB;
  const uint32_t tid = threadIdx.x;
  int cur = 0;
  for (int l = (int)a.top_log; l >= 0; l--) {
    const uint32_t n = 1u << l;
    if (tid < n) {
      uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint32_t m[16];
      const bool from_global = (l == (int)a.top_log);
      if (!from_global || a.prev) {
        const uint32_t* p = from_global ? a.prev + (size_t)tid * 16 : buf[cur ^ 1] + tid * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = p[k];
        b2s_compress(h, m);
      }
      const uint32_t c_begin = a.col_begin[l], c_end = a.col_end[l];
      for (uint32_t c0 = c_begin; c0 < c_end; c0 += 16) {
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < c_end) ? a.cols[c0 + k][tid] : 0u;
        b2s_compress(h, m);
      }
      uint32_t* o = a.layers[l] + (size_t)tid * 8;
#pragma unroll
      for (int k = 0; k < 8; k++) { o[k] = h[k]; buf[cur][tid * 8 + k] = h[k]; }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// Proof of work: smallest nonce in [base, base + n) with trailing_zeros(F(digest, [lo,hi,0..])[0..16B]) >= bits.
// result initialised to ~0ull; atomicMin keeps the smallest hit.
__global__ void __launch_bounds__(256) k_grind(const uint32_t* __restrict__ digest, uint32_t bits, uint64_t base,
                                               unsigned long long* result) {
  const uint64_t nonce = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h[8];
#pragma unroll
  for (int k = 0; k < 8; k++) h[k] = digest[k];
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
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  chan_mix_root_draw_dev(chan, root, felt_out, root_log);
}

// Decommitment gather: out[q * width + w] = addrs[q][w]  (width 1 = column values, 8 = 32-byte hashes).
__global__ void k_gather_words(const uint32_t* const* addrs, uint32_t n, uint32_t width, uint32_t* out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * width) return;
  uint32_t q = t / width, w = t - q * width;
  out[t] = addrs[q][w];
}


// ================================================================= host wrappers
void merkle_layer(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                  uint32_t* d_out, hipStream_t st) {
  uint32_t n = 1u << log_size;
  // algorithmic bytes: column values once + 64 B of child hashes in, 32 B out per node
  KProfScope kp("k_merkle_layer", (4.0 * ncols + (d_prev ? 64.0 : 0.0) + 32.0) * (double)n, st);
  hipLaunchKernelGGL(k_merkle_layer, dim3((n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
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
void merkle_tail(const MerkleTailArgs& a, hipStream_t st) {
  KProfScope kp("k_merkle_tail", 0.0, st);
  hipLaunchKernelGGL(k_merkle_tail, dim3(1), dim3(1024), 0, st, a);
  CM_HIP(hipGetLastError());
}
uint64_t grind_gpu(const uint8_t digest[32], uint32_t bits, hipStream_t st) {
  DevBuf d_digest(32), d_res(8);
  CM_HIP(hipMemcpyAsync(d_digest.p, digest, 32, hipMemcpyHostToDevice, st));
  // expected nonce ~ 2^bits: start with 8x that and grow, instead of always hashing 2^22 candidates
  uint64_t batch = 1ull << (bits + 3 < 12 ? 12 : (bits + 3 > 22 ? 22 : bits + 3));
  for (uint64_t base = 0;; base += batch, batch = batch < (1ull << 22) ? batch * 2 : batch) {
    unsigned long long init = ~0ull;
    CM_HIP(hipMemcpyAsync(d_res.p, &init, 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_grind, dim3(batch / 256), dim3(256), 0, st, d_digest.u32(), bits, base,
                       (unsigned long long*)d_res.p);
    CM_HIP(hipGetLastError());
    unsigned long long res;
    CM_HIP(hipMemcpyAsync(&res, d_res.p, 8, hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    if (res != ~0ull) return res;
    CM_CHECK(base < (1ull << 40), "grind: no nonce found");
  }
}
void gather_words(const uint32_t* const* d_addrs, uint32_t n, uint32_t width, uint32_t* d_out, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_gather_words, dim3((n * width + 255) / 256), dim3(256), 0, st, d_addrs, n, width, d_out);
  CM_HIP(hipGetLastError());
}

}  // namespace cm
