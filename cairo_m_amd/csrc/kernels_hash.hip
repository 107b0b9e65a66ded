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
#include <algorithm>
#include "field.hpp"
#include "device_common.hpp"
#include "engine.hpp"
#include "kprof.hpp"
#include "blake2s_dev.hpp"
#include "merkle_kernels.hpp"
#include "framing.hpp"

namespace cm {

// Proof of work: smallest nonce in [base, base + n) with trailing_zeros(F(digest, [lo,hi,0..])[0..16B]) >= bits.
// result initialised to ~0ull; atomicMin keeps the smallest hit.
struct GrindDigest { uint32_t w[8]; };   // the channel digest travels as a kernel argument: no host->device copy
template <bool U32S>
__global__ void __launch_bounds__(256) k_grind(GrindDigest digest, uint32_t bits, uint64_t base, unsigned long long* result) {
  const uint64_t nonce = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h[8];
  mix_u64_dev<U32S>(digest.w, nonce, h);
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

// Sharded FRI layer on the stream (prover_sharded.inc, round 6): the N sub-roots the all-gather left in `sub` (rank-major, 8 words
// each) -> the top log2 N levels of the tree (children-only nodes: Stwo's raw compression from the zero state, the host twin is
// hash_pair_host) -> root -> mix_root + draw_felt on the device copy of the channel.  `sub_log` keeps the sub-roots for the host,
// which rebuilds the top levels for the decommitment and replays the transcript step once per proof instead of once per layer.
__global__ void __launch_bounds__(64) k_shard_top_step(const uint32_t* __restrict__ sub, uint32_t n, uint32_t* chan, uint32_t* felt_out,
                                                       uint32_t* root_log, uint32_t* sub_log) {
  __shared__ uint32_t nodes[2][8 * 8];
  __shared__ uint32_t x8[8], felt[4];
  const uint32_t t = threadIdx.x;
  for (uint32_t i = t; i < 8 * n; i += 64) { const uint32_t w = sub[i]; nodes[0][i] = w; sub_log[i] = w; }
  __syncthreads();
  uint32_t cur = 0;
  for (uint32_t m = n; m > 1; m >>= 1) {
    if (t < m / 2) {
      uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, msg[16];
      for (int k = 0; k < 16; k++) msg[k] = nodes[cur][16 * t + k];
      b2s_compress(h, msg);
      for (int k = 0; k < 8; k++) nodes[cur ^ 1][8 * t + k] = h[k];
    }
    __syncthreads();
    cur ^= 1;
  }
  if (t < 4) chan_mix_root_draw_quad(t, chan, nodes[cur], felt_out, root_log, x8, felt);
}

// The same top levels WITHOUT a transcript step (sharded trees 1 / 2 / 3, round 6): root -> `root_out` (8 words), the sub-roots ->
// `sub_log`; the transcript kernels of the single-GPU prover (k_step_pow_relations, k_chan_init_mix_root_draw, k_chan_mix_root_draw)
// then read the root from device memory exactly as they read a whole tree's layer 0.
__global__ void __launch_bounds__(64) k_shard_top(const uint32_t* __restrict__ sub, uint32_t n, uint32_t* __restrict__ root_out, uint32_t* __restrict__ sub_log) {
  __shared__ uint32_t nodes[2][8 * 8];
  const uint32_t t = threadIdx.x;
  for (uint32_t i = t; i < 8 * n; i += 64) { const uint32_t w = sub[i]; nodes[0][i] = w; sub_log[i] = w; }
  __syncthreads();
  uint32_t cur = 0;
  for (uint32_t m = n; m > 1; m >>= 1) {
    if (t < m / 2) {
      uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, msg[16];
      for (int k = 0; k < 16; k++) msg[k] = nodes[cur][16 * t + k];
      b2s_compress(h, msg);
      for (int k = 0; k < 8; k++) nodes[cur ^ 1][8 * t + k] = h[k];
    }
    __syncthreads();
    cur ^= 1;
  }
  if (t < 8) root_out[t] = nodes[cur][t];
}

// powers[g] = rho^(n - 1 - g): the random-coefficient powers of the composition polynomial, one thread per power (square and
// multiply), from the coefficient the device-side transcript step (k_chan_mix_root_draw) left in device memory
__global__ void __launch_bounds__(256) k_coeff_powers(const uint32_t* __restrict__ rho4, uint32_t* __restrict__ powers, uint32_t n) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  QM31 r = QM31::from_u32(rho4), acc{M31(1)};
  for (uint32_t e = n - 1 - g; e; e >>= 1) {
    if (e & 1u) acc = acc * r;
    r = r * r;
  }
  acc.to_u32(powers + 4 * g);
}

// the same step with the channel state handed over in the kernel arguments (no upload in front of it)
struct ChanWords { uint32_t w[9]; };
__global__ void k_chan_init_mix_root_draw(ChanWords init, uint32_t* chan, const uint32_t* root, uint32_t* felt_out, uint32_t* root_log) {
  __shared__ uint32_t x8[8], felt[4];
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;
  for (int i = 0; i < 9; i++) chan[i] = init.w[i];   // every lane of the quad stores all nine words, then reads back its own stores
  chan_mix_root_draw_quad(threadIdx.x, chan, root, felt_out, root_log, x8, felt);
}

// Device-side transcript step behind the trace commitment (prover.rs:82-94): mix_root(root 1), the interaction proof of work
// (smallest nonce whose F(digest, nonce) has `pow_bits` trailing zero bits — SimdBackend::grind's predicate), mix_u64(nonce) and
// Relations::draw (one draw_felts(2) per relation: z, alpha) with the alpha powers the AIR kernels read.  ONE wave: the 64
// lanes test 64 nonces at a time, and lane k hashes the draw with counter n_sent = k (a draw is retried with the next
// counter when a word is >= 2P, so the i-th VALID counter belongs to relation i).  The host replays the same steps on its
// own channel later, from `out` = {root[8], nonce[2], n_sent, error, z of relation 0 [4]}, and checks they agree.
template <bool U32S>
__global__ void __launch_bounds__(64) k_step_pow_relations(ChanWords init, const uint32_t* __restrict__ root, uint32_t pow_bits, uint32_t n_rel,
                                                           uint32_t max_rel, uint32_t* __restrict__ rel_z, uint32_t* __restrict__ rel_pow,
                                                           uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x;
  const uint32_t iv0 = 0x6A09E667u ^ 0x01010020u;
  uint32_t h[8], m[16];
  // mix_root: digest = Blake2s256(digest || root): one final 64-byte block
  for (int i = 0; i < 8; i++) { m[i] = init.w[i]; m[8 + i] = root[i]; }
  h[0] = iv0; h[1] = 0xBB67AE85u; h[2] = 0x3C6EF372u; h[3] = 0xA54FF53Au; h[4] = 0x510E527Fu; h[5] = 0x9B05688Cu; h[6] = 0x1F83D9ABu; h[7] = 0x5BE0CD19u;
  b2s_compress(h, m, 64, 0xFFFFFFFFu);
  // proof of work over mix_u64 (default framing: the raw compression F(digest, [lo, hi, 0...], t = 0, f = 0))
  uint64_t nonce = 0;
  for (uint64_t base = 0;; base += 64) {
    uint32_t g[8];
    const uint64_t cand = base + lane;
    mix_u64_dev<U32S>(h, cand, g);
    uint32_t tz;
    if (g[0]) tz = __ffs(g[0]) - 1;
    else if (g[1]) tz = 32 + __ffs(g[1]) - 1;
    else if (g[2]) tz = 64 + __ffs(g[2]) - 1;
    else if (g[3]) tz = 96 + __ffs(g[3]) - 1;
    else tz = 128;
    const unsigned long long hit = __ballot(tz >= pow_bits);
    if (hit) { nonce = base + (uint64_t)(__ffsll((long long)hit) - 1); break; }
  }
  // mix_u64(nonce)
  {
    uint32_t g[8];
    mix_u64_dev<U32S>(h, nonce, g);
    for (int i = 0; i < 8; i++) h[i] = g[i];
  }
  // draw_random_bytes with n_sent = lane: Blake2s256(digest || le32(n_sent) || 0^28 || 0x00) — 65 bytes, two blocks
  uint32_t d[8];
  d[0] = iv0; d[1] = 0xBB67AE85u; d[2] = 0x3C6EF372u; d[3] = 0xA54FF53Au; d[4] = 0x510E527Fu; d[5] = 0x9B05688Cu; d[6] = 0x1F83D9ABu; d[7] = 0x5BE0CD19u;
  for (int i = 0; i < 8; i++) { m[i] = h[i]; m[8 + i] = 0; }
  m[8] = lane;
  b2s_compress(d, m, 64, 0);
  for (int i = 0; i < 16; i++) m[i] = 0;
  b2s_compress(d, m, 65, 0xFFFFFFFFu);
  bool valid = true;
  for (int i = 0; i < 8; i++) valid = valid && d[i] < 2u * P;
  const unsigned long long vmask = __ballot(valid);
  const uint32_t rank = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull));
  if (valid && rank < n_rel) {
    const QM31 z(M31::from_u32(d[0]), M31::from_u32(d[1]), M31::from_u32(d[2]), M31::from_u32(d[3]));
    const QM31 alpha(M31::from_u32(d[4]), M31::from_u32(d[5]), M31::from_u32(d[6]), M31::from_u32(d[7]));
    z.to_u32(rel_z + 4 * rank);
    QM31 cur{M31(1)};
    for (uint32_t i = 0; i < max_rel; i++) { cur.to_u32(rel_pow + 4 * (rank * max_rel + i)); cur = cur * alpha; }
    if (rank == 0) z.to_u32(out + 12);
    if (rank == n_rel - 1) out[10] = lane + 1;   // the channel's n_sent after the draws
  }
  if (lane == 0) {
    for (int i = 0; i < 8; i++) out[i] = root[i];
    out[8] = (uint32_t)nonce; out[9] = (uint32_t)(nonce >> 32);
    out[11] = (uint32_t)__popcll(vmask) < n_rel ? 1u : 0u;   // fewer than n_rel valid draws among 64 counters: cannot happen in practice
  }
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
  KProfExt kp("k_merkle_layer", (4.0 * ncols + (d_prev ? 64.0 : 0.0) + 32.0) * (double)n, st,
              /* Blake2s compressions */ (double)n * ((d_prev ? 1.0 : 0.0) + (double)((ncols + 15) / 16)));
  // narrow layers (no columns or one SecureColumn): a wave walks several 64-node chunks with the next chunk's loads in flight
  // (k_merkle_narrow).  Chunks per wave: as many as still leave >= 2 waves per wave slot of the chip (256 CUs x 32 slots).
  // A/B: CM_MERKLE_NPW=0 restores k_merkle_layer for these layers, 1 / 2 / 4 / 8 force a chunk count.
  const int npw_env = tune(T_MERKLE_NPW);
  if (npw_env != 0 && (ncols == 0 || ncols == 4) && (d_prev || ncols) && log_size >= 14) {
    uint32_t npw = npw_env > 0 ? (uint32_t)npw_env : std::min(8u, std::max(1u, n >> 20));
    while (npw > 1 && (n % (256u * npw)) != 0) npw >>= 1;
    const dim3 grid(n / (256u * npw));
    const bool rfc = framing().hash_node_rfc;
#define CM_NARROW(R, P, C) CM_KPROF_LAUNCH(kp, (k_merkle_narrow<R, P, C>), grid, dim3(256), 0, st, d_prev, d_cols, d_out, npw)
    if (d_prev && ncols) { if (rfc) CM_NARROW(true, true, 4); else CM_NARROW(false, true, 4); }
    else if (d_prev) { if (rfc) CM_NARROW(true, true, 0); else CM_NARROW(false, true, 0); }
    else { if (rfc) CM_NARROW(true, false, 4); else CM_NARROW(false, false, 4); }
#undef CM_NARROW
    CM_HIP(hipGetLastError());
    return;
  }
  if (framing().hash_node_rfc) CM_KPROF_LAUNCH(kp, k_merkle_layer<true>, dim3((n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
  else CM_KPROF_LAUNCH(kp, k_merkle_layer<false>, dim3((n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
  CM_HIP(hipGetLastError());
}
void merkle_layer_quad(uint32_t log_size, const uint32_t* d_prev, const uint32_t* const* d_cols, uint32_t ncols,
                       uint32_t* d_out, hipStream_t st) {
  uint32_t n = 1u << log_size;
  KProfScope kp("k_merkle_layer_quad", (4.0 * ncols + (d_prev ? 64.0 : 0.0) + 32.0) * (double)n, st);
  if (framing().hash_node_rfc) hipLaunchKernelGGL(k_merkle_layer_quad<true>, dim3((4 * n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
  else hipLaunchKernelGGL(k_merkle_layer_quad<false>, dim3((4 * n + 255) / 256), dim3(256), 0, st, log_size, d_prev, d_cols, ncols, d_out);
  CM_HIP(hipGetLastError());
}
void merkle_multi(const MerkleMultiArgs& a, double alg_bytes, hipStream_t st) {
  KProfScope kp("k_merkle_multi", alg_bytes, st);
  if (framing().hash_node_rfc) hipLaunchKernelGGL(k_merkle_multi<true>, dim3((1u << a.top_log) / 256), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_merkle_multi<false>, dim3((1u << a.top_log) / 256), dim3(256), 0, st, a);
  CM_HIP(hipGetLastError());
}
void chan_mix_root_draw(uint32_t* d_chan, const uint32_t* d_root, uint32_t* d_felt_out, uint32_t* d_root_log, hipStream_t st) {
  hipLaunchKernelGGL(k_chan_mix_root_draw, dim3(1), dim3(64), 0, st, d_chan, d_root, d_felt_out, d_root_log);
  CM_HIP(hipGetLastError());
}
void shard_top_step(const uint32_t* d_sub, uint32_t n, uint32_t* d_chan, uint32_t* d_felt_out, uint32_t* d_root_log, uint32_t* d_sub_log, hipStream_t st) {
  CM_CHECK(n >= 1 && n <= 8 && (n & (n - 1)) == 0, "shard_top_step: rank count");
  hipLaunchKernelGGL(k_shard_top_step, dim3(1), dim3(64), 0, st, d_sub, n, d_chan, d_felt_out, d_root_log, d_sub_log);
  CM_HIP(hipGetLastError());
}
void shard_top(const uint32_t* d_sub, uint32_t n, uint32_t* d_root_out, uint32_t* d_sub_log, hipStream_t st) {
  CM_CHECK(n >= 1 && n <= 8 && (n & (n - 1)) == 0, "shard_top: rank count");
  hipLaunchKernelGGL(k_shard_top, dim3(1), dim3(64), 0, st, d_sub, n, d_root_out, d_sub_log);
  CM_HIP(hipGetLastError());
}
void chan_init_mix_root_draw(const uint32_t init9[9], uint32_t* d_chan, const uint32_t* d_root, uint32_t* d_felt_out, uint32_t* d_root_log,
                             hipStream_t st) {
  ChanWords cw;
  memcpy(cw.w, init9, sizeof(cw.w));
  hipLaunchKernelGGL(k_chan_init_mix_root_draw, dim3(1), dim3(64), 0, st, cw, d_chan, d_root, d_felt_out, d_root_log);
  CM_HIP(hipGetLastError());
}
void step_pow_relations(const uint32_t init9[9], const uint32_t* d_root, uint32_t pow_bits, uint32_t n_rel, uint32_t max_rel, uint32_t* d_rel_z,
                        uint32_t* d_rel_pow, uint32_t* d_out16, hipStream_t st) {
  ChanWords cw;
  memcpy(cw.w, init9, sizeof(cw.w));
  CM_CHECK(n_rel >= 1 && n_rel <= 32, "step_pow_relations: relation count");
  if (framing().mix_u64_u32s) hipLaunchKernelGGL(k_step_pow_relations<true>, dim3(1), dim3(64), 0, st, cw, d_root, pow_bits, n_rel, max_rel, d_rel_z, d_rel_pow, d_out16);
  else hipLaunchKernelGGL(k_step_pow_relations<false>, dim3(1), dim3(64), 0, st, cw, d_root, pow_bits, n_rel, max_rel, d_rel_z, d_rel_pow, d_out16);
  CM_HIP(hipGetLastError());
}
void coeff_powers(const uint32_t* d_rho, uint32_t* d_powers, uint32_t n, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_coeff_powers, dim3((n + 255) / 256), dim3(256), 0, st, d_rho, d_powers, n);
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
    uint32_t* own = ring;
    at_thread_exit([own] { (void)hipFree(own); });
  }
  return ring + (pos++ % N);
}
void merkle_top(MerkleTopArgs& a, hipStream_t st) {
  CM_CHECK(a.top_log >= 9 && a.top_log <= MERKLE_TOP_MAX_LOG, "merkle_top: bad layer range");
  a.ticket = next_ticket(st);
  KProfScope kp("k_merkle_top", 0.0, st);
  const dim3 grid(1u << (a.top_log - 8));
  CM_CHECK(a.x.fold_mode <= 2 && (a.x.fold_mode == 0 || (a.prev == nullptr && a.x.fold_src[0] && a.x.fold_dst[0] && a.x.alpha && a.x.ixt)),
           "merkle_top: incomplete fold description");
#define CM_TOP(R, F) hipLaunchKernelGGL((k_merkle_top<R, F>), grid, dim3(256), 0, st, a)
  if (framing().hash_node_rfc) { if (a.x.fold_mode == 2) CM_TOP(true, 2); else if (a.x.fold_mode == 1) CM_TOP(true, 1); else CM_TOP(true, 0); }
  else { if (a.x.fold_mode == 2) CM_TOP(false, 2); else if (a.x.fold_mode == 1) CM_TOP(false, 1); else CM_TOP(false, 0); }
#undef CM_TOP
  CM_HIP(hipGetLastError());
}
void merkle_tail(const MerkleTailArgs& a, hipStream_t st) {
  KProfScope kp("k_merkle_tail", 0.0, st);
  if (framing().hash_node_rfc) hipLaunchKernelGGL(k_merkle_tail<true>, dim3(1), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(k_merkle_tail<false>, dim3(1), dim3(1024), 0, st, a);
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
    if (framing().mix_u64_u32s) hipLaunchKernelGGL(k_grind<true>, dim3(batch / 256), dim3(256), 0, st, dg, bits, base, (unsigned long long*)d_res.p);
    else hipLaunchKernelGGL(k_grind<false>, dim3(batch / 256), dim3(256), 0, st, dg, bits, base, (unsigned long long*)d_res.p);
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
