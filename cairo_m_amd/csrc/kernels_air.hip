// Per-component AIR kernels for gfx950 (one thread per row, columns streamed column-major):
//   k_opcode_trace / k_builtin_trace : Claim::write_trace (crates/prover/src/components/mod.rs:106-194)
//   k_hist                           : range_check_N / bitwise multiplicities (range_check_macro.rs:62-112)
//   k_logup                          : InteractionClaim::write_interaction_trace (components/mod.rs:198-281)
//   k_constraints                    : FrameworkComponent::evaluate_constraint_quotients_on_domain
// plus the LogUp tail (claimed sum, cumsum shift, prefix sum in coset order = LogupTraceGenerator::
// finalize_last) and preprocessed-column generation (preprocessed/mod.rs:36-38).
#include "gpu_air.hpp"
#include "engine.hpp"
#include "air_kernels.hpp"

namespace cm {

template <class C>
__global__ void __launch_bounds__(256) k_opcode_trace(const air::Bundle* __restrict__ bundles, uint32_t n,
                                                      const air::Access* __restrict__ acc, uint32_t log_size,
                                                      uint32_t* const* __restrict__ cols) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  air::Bundle b = r < n ? bundles[r] : air::default_bundle();
  M31 out[C::N_TRACE];
  C::template witness<DevOps>(b, acc, r < n ? 1u : 0u, out);
#pragma unroll
  for (int c = 0; c < C::N_TRACE; c++) cols[c][r] = out[c].v;
}

// builtin components: kind-specific row sources
__global__ void __launch_bounds__(256) k_memory_trace(const air::MemoryCell* __restrict__ init, uint32_t ni,
                                                      const air::MemoryCell* __restrict__ fin, uint32_t nf, uint32_t root_i,
                                                      uint32_t root_f, uint32_t log_size, uint32_t* const* __restrict__ cols) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  M31 out[air::MemoryC::N_TRACE];
  const air::MemoryCell* cell = r < ni ? init + r : (r < ni + nf ? fin + (r - ni) : nullptr);
  air::MemoryC::witness<DevOps>(cell, r < ni ? root_i : root_f, r < ni + nf ? 1u : 0u, out);
  for (int c = 0; c < air::MemoryC::N_TRACE; c++) cols[c][r] = out[c].v;
}
__global__ void __launch_bounds__(256) k_merkle_trace(const air::MerkleNode* __restrict__ init, uint32_t ni,
                                                      const air::MerkleNode* __restrict__ fin, uint32_t nf, uint32_t root_i,
                                                      uint32_t root_f, uint32_t log_size, uint32_t* const* __restrict__ cols) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  M31 out[air::MerkleC::N_TRACE];
  const air::MerkleNode* n = r < ni ? init + r : (r < ni + nf ? fin + (r - ni) : nullptr);
  air::MerkleC::witness<DevOps>(n, r < ni ? root_i : root_f, r < ni + nf ? 1u : 0u, out);
  for (int c = 0; c < air::MerkleC::N_TRACE; c++) cols[c][r] = out[c].v;
}
__global__ void __launch_bounds__(256) k_clock_update_trace(const air::ClockUpdateRow* __restrict__ rows, uint32_t n,
                                                            uint32_t log_size, uint32_t* const* __restrict__ cols) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  M31 out[air::ClockUpdateC::N_TRACE];
  air::ClockUpdateC::witness<DevOps>(r < n ? rows + r : nullptr, r < n ? 1u : 0u, out);
  for (int c = 0; c < air::ClockUpdateC::N_TRACE; c++) cols[c][r] = out[c].v;
}
// poseidon2: one thread per row; 443 cells staged in registers/scratch then streamed out.
__global__ void __launch_bounds__(64) k_poseidon2_trace(const air::MerkleNode* __restrict__ init, uint32_t ni,
                                                        const air::MerkleNode* __restrict__ fin, uint32_t nf,
                                                        uint32_t log_size, uint32_t* const* __restrict__ cols) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  M31 out[air::Poseidon2C::N_TRACE];
  uint32_t st[16];
  for (int i = 0; i < 16; i++) st[i] = 0;
  bool live = r < ni + nf;
  if (live) {
    const air::MerkleNode* n = r < ni ? init + r : fin + (r - ni);
    st[0] = n->left_value;
    st[1] = n->right_value;
  }
  air::Poseidon2C::witness<DevOps>(live ? st : nullptr, live ? 1u : 0u, out);
  for (int c = 0; c < air::Poseidon2C::N_TRACE; c++) cols[c][r] = out[c].v;
}

template <class C>
__global__ void __launch_bounds__(256) k_hist(const uint32_t* const* __restrict__ cols, uint32_t log_size, HistPtrs h) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  HistEval e;
  e.cols = cols; e.row = r; e.h = h;
  C::eval(e);
}

template <class C>
__global__ void __launch_bounds__(256) k_logup(const uint32_t* const* __restrict__ cols, const uint32_t* const* __restrict__ pp,
                                               uint32_t log_size, const DevRelations* __restrict__ rels,
                                               uint32_t* const* __restrict__ out) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (1u << log_size)) return;
  LogupEval e;
  e.cols = cols; e.pp = pp; e.out = out; e.rels = rels; e.row = r;
  C::eval(e);
}

template <class C>
__global__ void __launch_bounds__(256) k_constraints(ConstraintArgs a) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t en = a.log_size + 1;
  if (r >= (1u << en)) return;
  DomainEval e;
  e.tr = a.tr; e.it = a.it; e.pp = a.pp; e.rels = a.rels; e.coeff = a.coeff;
  e.row = r; e.prev_row = shifted_row(r, en, a.log_size, -1);
  e.n_base = a.n_base;
  e.cumsum_shift = QM31::from_u32(a.cumsum_shift);
  C::eval(e);
  QM31 v = e.acc * M31(a.denom_inv[r >> a.log_size]);
  a.acc[0][r] = (M31(a.acc[0][r]) + v.a.a).v;
  a.acc[1][r] = (M31(a.acc[1][r]) + v.a.b).v;
  a.acc[2][r] = (M31(a.acc[2][r]) + v.b.a).v;
  a.acc[3][r] = (M31(a.acc[3][r]) + v.b.b).v;
}

// ---- LogUp tail ------------------------------------------------------------------------------------
// partial sums of the last 4 interaction columns -> partial[block][4]
__global__ void __launch_bounds__(256) k_qsum_partial(const uint32_t* const* __restrict__ cols4, uint32_t n, uint32_t* partial) {
  QM31 acc;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc += QM31(M31(cols4[0][i]), M31(cols4[1][i]), M31(cols4[2][i]), M31(cols4[3][i]));
  acc = block_reduce_qm31(acc);
  if (threadIdx.x == 0) acc.to_u32(partial + 4 * blockIdx.x);
}
__global__ void __launch_bounds__(256) k_qsum_final(const uint32_t* partial, uint32_t nparts, uint32_t* out) {
  QM31 acc;
  for (uint32_t i = threadIdx.x; i < nparts; i += blockDim.x) acc += QM31::from_u32(partial + 4 * i);
  acc = block_reduce_qm31(acc);
  if (threadIdx.x == 0) acc.to_u32(out);
}
__device__ __forceinline__ uint32_t coset_order_row(uint32_t k, uint32_t log_size) {
  uint32_t ci = (k & 1u) ? (((2u << log_size) - k) >> 1) : (k >> 1);
  return bit_reverse(ci, log_size);
}
constexpr uint32_t SCAN_BLOCK = 1024;  // elements per block (256 threads x 4)
// phase 1: per-block inclusive scan of (value - shift) in coset order -> tmp (linear coset order), block totals
__global__ void __launch_bounds__(256) k_scan_local(const uint32_t* const* __restrict__ cols4, uint32_t log_size,
                                                    const uint32_t* __restrict__ shift4, uint32_t* __restrict__ tmp,
                                                    uint32_t* __restrict__ block_tot) {
  __shared__ uint32_t sh[256 * 4];
  const uint32_t n = 1u << log_size;
  const QM31 shift = QM31::from_u32(shift4);
  const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
  QM31 v[4];
  QM31 run;
  for (int j = 0; j < 4; j++) {
    uint32_t k = base + j;
    if (k < n) {
      uint32_t r = coset_order_row(k, log_size);
      run += QM31(M31(cols4[0][r]), M31(cols4[1][r]), M31(cols4[2][r]), M31(cols4[3][r])) - shift;
    }
    v[j] = run;
  }
  run.to_u32(sh + 4 * threadIdx.x);
  __syncthreads();
  // Hillis-Steele over the 256 thread totals
  for (uint32_t off = 1; off < 256; off <<= 1) {
    QM31 add;
    bool has = threadIdx.x >= off;
    if (has) add = QM31::from_u32(sh + 4 * (threadIdx.x - off));
    __syncthreads();
    if (has) (QM31::from_u32(sh + 4 * threadIdx.x) + add).to_u32(sh + 4 * threadIdx.x);
    __syncthreads();
  }
  QM31 prefix;
  if (threadIdx.x > 0) prefix = QM31::from_u32(sh + 4 * (threadIdx.x - 1));
  for (int j = 0; j < 4; j++) {
    uint32_t k = base + j;
    if (k < n) (v[j] + prefix).to_u32(tmp + 4 * (size_t)k);
  }
  if (threadIdx.x == 255) {
    uint32_t t[4];
    for (int j = 0; j < 4; j++) t[j] = sh[4 * 255 + j];
    for (int j = 0; j < 4; j++) block_tot[4 * blockIdx.x + j] = t[j];
  }
}
// phase 2: exclusive scan of block totals (single block, sequential chunks)
__global__ void __launch_bounds__(256) k_scan_blocks(uint32_t* block_tot, uint32_t nblocks) {
  __shared__ uint32_t sh[256 * 4];
  uint32_t per = (nblocks + 255) / 256;
  uint32_t b0 = threadIdx.x * per;
  QM31 run;
  for (uint32_t i = b0; i < b0 + per && i < nblocks; i++) run += QM31::from_u32(block_tot + 4 * i);
  run.to_u32(sh + 4 * threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    QM31 acc;
    for (int t = 0; t < 256; t++) {
      QM31 x = QM31::from_u32(sh + 4 * t);
      acc.to_u32(sh + 4 * t);
      acc += x;
    }
  }
  __syncthreads();
  QM31 acc = QM31::from_u32(sh + 4 * threadIdx.x);
  for (uint32_t i = b0; i < b0 + per && i < nblocks; i++) {
    QM31 x = QM31::from_u32(block_tot + 4 * i);
    acc.to_u32(block_tot + 4 * i);
    acc += x;
  }
}
// phase 3: add block prefix and scatter back to bit-reversed circle-domain storage
__global__ void __launch_bounds__(256) k_scan_scatter(uint32_t* const* __restrict__ cols4, uint32_t log_size,
                                                      const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ block_pre) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (1u << log_size)) return;
  QM31 v = QM31::from_u32(tmp + 4 * (size_t)k) + QM31::from_u32(block_pre + 4 * (k / SCAN_BLOCK));
  uint32_t r = coset_order_row(k, log_size);
  cols4[0][r] = v.a.a.v; cols4[1][r] = v.a.b.v; cols4[2][r] = v.b.a.v; cols4[3][r] = v.b.b.v;
}

__global__ void k_preproc(int pp_id, uint32_t log_size, uint32_t* col) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (1u << log_size)) col[i] = air::preproc_value(pp_id, i);
}
__global__ void k_add_columns(uint32_t* const* dst, const uint32_t* const* src, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[blockIdx.y][i] = (M31(dst[blockIdx.y][i]) + M31(src[blockIdx.y][i])).v;
}

// ================================================================= host wrappers
static inline dim3 grid_for(uint32_t n, uint32_t bs = 256) { return dim3((n + bs - 1) / bs); }

void launch_opcode_trace(int cid, const void* bundles, uint32_t n, const void* acc, uint32_t log_size, uint32_t* const* d_cols,
                         hipStream_t st) {
  switch (cid) {
#define CM_X(id, T)                                                                                                   \
  case air::id:                                                                                                       \
    hipLaunchKernelGGL(k_opcode_trace<air::T>, grid_for(1u << log_size), dim3(256), 0, st, (const air::Bundle*)bundles, n, \
                       (const air::Access*)acc, log_size, d_cols);                                                   \
    break;
    AIR_OPCODE_COMPONENTS(CM_X)
#undef CM_X
    default: CM_CHECK(false, "launch_opcode_trace: not an opcode component");
  }
  CM_HIP(hipGetLastError());
}
void launch_memory_trace(const void* init, uint32_t ni, const void* fin, uint32_t nf, uint32_t root_i, uint32_t root_f,
                         uint32_t log_size, uint32_t* const* d_cols, hipStream_t st) {
  hipLaunchKernelGGL(k_memory_trace, grid_for(1u << log_size), dim3(256), 0, st, (const air::MemoryCell*)init, ni,
                     (const air::MemoryCell*)fin, nf, root_i, root_f, log_size, d_cols);
  CM_HIP(hipGetLastError());
}
void launch_merkle_trace(const void* init, uint32_t ni, const void* fin, uint32_t nf, uint32_t root_i, uint32_t root_f,
                         uint32_t log_size, uint32_t* const* d_cols, hipStream_t st) {
  hipLaunchKernelGGL(k_merkle_trace, grid_for(1u << log_size), dim3(256), 0, st, (const air::MerkleNode*)init, ni,
                     (const air::MerkleNode*)fin, nf, root_i, root_f, log_size, d_cols);
  CM_HIP(hipGetLastError());
}
void launch_clock_update_trace(const void* rows, uint32_t n, uint32_t log_size, uint32_t* const* d_cols, hipStream_t st) {
  hipLaunchKernelGGL(k_clock_update_trace, grid_for(1u << log_size), dim3(256), 0, st, (const air::ClockUpdateRow*)rows, n,
                     log_size, d_cols);
  CM_HIP(hipGetLastError());
}
void launch_poseidon2_trace(const void* init, uint32_t ni, const void* fin, uint32_t nf, uint32_t log_size,
                            uint32_t* const* d_cols, hipStream_t st) {
  hipLaunchKernelGGL(k_poseidon2_trace, grid_for(1u << log_size, 64), dim3(64), 0, st, (const air::MerkleNode*)init, ni,
                     (const air::MerkleNode*)fin, nf, log_size, d_cols);
  CM_HIP(hipGetLastError());
}
void launch_hist(int cid, const uint32_t* const* d_cols, uint32_t log_size, const HistPtrs& h, hipStream_t st) {
  switch (cid) {
#define CM_X(id, T) \
  case air::id: hipLaunchKernelGGL(k_hist<air::T>, grid_for(1u << log_size), dim3(256), 0, st, d_cols, log_size, h); break;
    AIR_OPCODE_COMPONENTS(CM_X)
#undef CM_X
    default: CM_CHECK(false, "launch_hist: not an opcode component");
  }
  CM_HIP(hipGetLastError());
}
void launch_logup(int cid, const uint32_t* const* d_cols, const uint32_t* const* d_pp, uint32_t log_size,
                  const DevRelations* d_rels, uint32_t* const* d_out, hipStream_t st) {
  switch (cid) {
#define CM_X(id, T)                                                                                                  \
  case air::id:                                                                                                      \
    hipLaunchKernelGGL(k_logup<air::T>, grid_for(1u << log_size), dim3(256), 0, st, d_cols, d_pp, log_size, d_rels, d_out); \
    break;
    AIR_ALL_COMPONENTS(CM_X)
#undef CM_X
  }
  CM_HIP(hipGetLastError());
}
void launch_constraints(int cid, const ConstraintArgs& a, hipStream_t st) {
  switch (cid) {
#define CM_X(id, T) \
  case air::id: hipLaunchKernelGGL(k_constraints<air::T>, grid_for(2u << a.log_size), dim3(256), 0, st, a); break;
    AIR_ALL_COMPONENTS(CM_X)
#undef CM_X
  }
  CM_HIP(hipGetLastError());
}

size_t logup_finalize_scratch_words(uint32_t log_size) {
  size_t n = (size_t)1 << log_size;
  size_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  return 4 * n + 4 * nblocks + 4 * 1024 + 8;
}
// claimed_sum -> d_claimed (4 u32, device); columns rewritten in place with the shifted prefix sum.
void logup_finalize_last(uint32_t* const* d_cols4, uint32_t log_size, uint32_t* d_scratch, uint32_t* h_claimed_sum,
                         hipStream_t st) {
  uint32_t n = 1u << log_size;
  uint32_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  uint32_t* d_tmp = d_scratch;
  uint32_t* d_btot = d_tmp + 4 * (size_t)n;
  uint32_t* d_part = d_btot + 4 * (size_t)nblocks;
  uint32_t* d_sum = d_part + 4 * 1024;
  uint32_t* d_shift = d_sum + 4;
  uint32_t nparts = n / 256 < 1 ? 1 : (n / 256 > 1024 ? 1024 : n / 256);
  hipLaunchKernelGGL(k_qsum_partial, dim3(nparts), dim3(256), 0, st, (const uint32_t* const*)d_cols4, n, d_part);
  hipLaunchKernelGGL(k_qsum_final, dim3(1), dim3(256), 0, st, d_part, nparts, d_sum);
  CM_HIP(hipMemcpyAsync(h_claimed_sum, d_sum, 16, hipMemcpyDeviceToHost, st));
  CM_HIP(hipStreamSynchronize(st));
  QM31 shift = QM31::from_u32(h_claimed_sum) * inv(M31::from_u32(n));
  uint32_t sh[4];
  shift.to_u32(sh);
  CM_HIP(hipMemcpyAsync(d_shift, sh, 16, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_scan_local, dim3(nblocks), dim3(256), 0, st, (const uint32_t* const*)d_cols4, log_size, d_shift, d_tmp,
                     d_btot);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, st, d_btot, nblocks);
  hipLaunchKernelGGL(k_scan_scatter, grid_for(n), dim3(256), 0, st, d_cols4, log_size, d_tmp, d_btot);
  CM_HIP(hipGetLastError());
}
void launch_preproc(int pp_id, uint32_t log_size, uint32_t* d_col, hipStream_t st) {
  hipLaunchKernelGGL(k_preproc, grid_for(1u << log_size), dim3(256), 0, st, pp_id, log_size, d_col);
  CM_HIP(hipGetLastError());
}
void add_columns(uint32_t* const* d_dst, const uint32_t* const* d_src, uint32_t ncols, uint32_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_add_columns, dim3((n + 255) / 256, ncols), dim3(256), 0, st, d_dst, d_src, n);
  CM_HIP(hipGetLastError());
}

}  // namespace cm
