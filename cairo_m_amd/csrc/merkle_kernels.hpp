// Merkle layer kernels for gfx950 (Blake2s, Stwo MerkleOps<Blake2sMerkleHasher>::commit_on_layer framing).
// Kept in a header so that a lab (tools/*_lab.hip) can time variants against exactly the shipped kernels.
#pragma once
#include "blake2s_dev.hpp"
#include "engine.hpp"
#include "device_common.hpp"

namespace cm {

// hashes[i] = hash_node(children (prev[2i], prev[2i+1]) if prev != null, cols[*][i])
template <bool RFC>
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
  __builtin_assume(i < (1u << 29));  // byte offsets fit 32 bits: scalar column base + 32-bit lane offset addressing
  const bool full_block = blk0 + 256 <= n;
  NodeFrame<RFC> fr(prev != nullptr, n_cols);
  uint32_t h[8];
  fr.init(h);
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
      fr.absorb(h, m, 64);
    } else if (i < n) {
      const uint4* p = reinterpret_cast<const uint4*>(prev + (size_t)i * 16);
      uint4 a = p[0], b = p[1], c = p[2], d = p[3];
      m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
      m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
      m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
      m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
      fr.absorb(h, m, 64);
    }
  }
  if (i < n) {
    __builtin_assume(i < (1u << 29));
    uint32_t c0 = 0;
    for (; c0 + 16 <= n_cols; c0 += 16) {  // full chunks: 16 loads issue back to back, no per-column bounds branches
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) m[k] = CM_GCOL(cols[c0 + k])[i];
      fr.absorb(h, m, 64);
    }
    if (n_cols - c0 == 4) {
      // SecureColumn leaves (every FRI layer, the composition tree): 4 live message words, 12 literal zeros — the compiler
      // drops the `+ 0` of 120 of the 160 message additions of this compression (v_add3 -> v_add: -8 % issue cycles)
      uint32_t z[16] = {0};
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) z[k] = CM_GCOL(cols[c0 + k])[i];
      fr.absorb(h, z, 16);
    } else if (c0 < n_cols) {
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < n_cols) ? CM_GCOL(cols[c0 + k])[i] : 0u;
      fr.absorb(h, m, 4u * (n_cols - c0));
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

// ---- narrow layers: one or two compressions per node ---------------------------------------------------------------------
// SecureColumn trees (the composition tree, every FRI layer) have 4 columns on their leaf layer and none below: a node is ONE
// compression (children only, or 4 column words) or two (children + 4 words).  With one node per thread and one short-lived
// wave per 64 nodes, k_merkle_layer runs those layers at ~33 G compressions/s against ~50 G/s on wide layers (rocprofv3
// timeline, profiles/r03*): a wave lives ~16 us of which ~10 us is its share of the SIMD's VALU — the rest (wave launch, the
// exposed load latency, four block barriers around the LDS staging) cannot be hidden by other waves because 8 per SIMD is the
// occupancy limit.  Here a WAVE walks `npw` consecutive 64-node chunks: the coalesced loads of chunk c + 1 are in flight while
// chunk c is compressed, and the children / result staging goes through a wave-private LDS window (in-order DS ops of one wave:
// no block barrier at all).  NC = 0 (children only) or 4 (SecureColumn words); PREV = the layer has children.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool RFC, bool PREV, int NC>
__global__ void __launch_bounds__(256) k_merkle_narrow(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols,
                                                       uint32_t* __restrict__ out, uint32_t npw) {
  static_assert(NC == 0 || NC == 4, "narrow layers carry no columns or one SecureColumn");
  static_assert(PREV || NC == 4, "a node hashes something");
  __shared__ uint4 stage[4][256];           // per wave: 64 nodes x 64 B
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint4* st = stage[w];
  const uint32_t node0 = (blockIdx.x * 4u + w) * 64u * npw;   // the wave's first node; chunk c = nodes [node0 + 64 c, + 64)
  __builtin_assume(node0 < (1u << 29));
  cm_gptr col[NC ? NC : 1];
#pragma unroll
  for (int k = 0; k < NC; k++) col[k] = CM_GCOL(cols[k]);
  uint4 raw0, raw1, raw2, raw3;
  uint32_t zc0 = 0, zc1 = 0, zc2 = 0, zc3 = 0;
  // (plain scalars, no arrays / lambdas: an indexed `raw[j]` captured by reference went to scratch memory)
#define CM_NARROW_ISSUE(nb_)                                                                   \
  {                                                                                            \
    if (PREV) {                                                                                \
      const uint4* p_ = reinterpret_cast<const uint4*>(prev + (size_t)(nb_) * 16);             \
      raw0 = p_[lane]; raw1 = p_[64 + lane]; raw2 = p_[128 + lane]; raw3 = p_[192 + lane];     \
    }                                                                                          \
    if (NC) { zc0 = col[0][(nb_) + lane]; zc1 = col[1][(nb_) + lane]; zc2 = col[2][(nb_) + lane]; zc3 = col[3][(nb_) + lane]; } \
  }
  CM_NARROW_ISSUE(node0)
  for (uint32_t c = 0; c < npw; c++) {
    const uint32_t nb = node0 + 64u * c;
    uint32_t m[16], z[16];
    if (PREV) {
      // uint4 number q = 64 j + lane of the window belongs to node q >> 2, part q & 3; the slot rotation by (node >> 2) keeps
      // both the 128-bit writes and the per-node reads conflict-free (same scheme as k_merkle_layer)
#define CM_NARROW_PUT(j_, r_)                                                   \
  {                                                                             \
    const uint32_t q_ = (j_) * 64 + lane, node_ = q_ >> 2, part_ = q_ & 3;      \
    st[node_ * 4 + ((part_ + (node_ >> 2)) & 3)] = r_;                          \
  }
      CM_NARROW_PUT(0, raw0) CM_NARROW_PUT(1, raw1) CM_NARROW_PUT(2, raw2) CM_NARROW_PUT(3, raw3)
#undef CM_NARROW_PUT
      wave_lds_sync();
      const uint4 a = st[lane * 4 + ((0 + (lane >> 2)) & 3)], b = st[lane * 4 + ((1 + (lane >> 2)) & 3)];
      const uint4 cc = st[lane * 4 + ((2 + (lane >> 2)) & 3)], d = st[lane * 4 + ((3 + (lane >> 2)) & 3)];
      m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
      m[8] = cc.x; m[9] = cc.y; m[10] = cc.z; m[11] = cc.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
      wave_lds_sync();
    }
    if (NC) {
#pragma unroll
      for (int k = 0; k < 16; k++) z[k] = 0;     // 12 literal zeros: most message additions of this compression fold away
      z[0] = zc0; z[1] = zc1; z[2] = zc2; z[3] = zc3;
    }
    if (c + 1 < npw) CM_NARROW_ISSUE(nb + 64u)   // the next chunk's loads fly during this chunk's compressions
    NodeFrame<RFC> fr(PREV, NC);
    uint32_t h[8];
    fr.init(h);
    if (PREV) fr.absorb(h, m, 64);
    if (NC) fr.absorb(h, z, 4u * NC);
    // result: 64 x 32 B through the wave's LDS window, stored with two wave-contiguous 1 KiB instructions
    st[lane * 2 + 0] = make_uint4(h[0], h[1], h[2], h[3]);
    st[lane * 2 + 1] = make_uint4(h[4], h[5], h[6], h[7]);
    wave_lds_sync();
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)nb * 8);
    o[lane] = st[lane];
    o[64 + lane] = st[64 + lane];
    wave_lds_sync();
  }
#undef CM_NARROW_ISSUE
}

// hash_node of one tree node, one thread per node (h is initialised here) / one quad of lanes per node
template <bool RFC>
__device__ __forceinline__ void merkle_node_thread(const uint32_t* children, const uint32_t* const* __restrict__ cols, uint32_t c_begin,
                                                   uint32_t c_end, uint32_t i, uint32_t (&h)[8]) {
  NodeFrame<RFC> fr(children != nullptr, c_end - c_begin);
  fr.init(h);
  uint32_t m[16];
  if (children) {
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = children[k];
    fr.absorb(h, m, 64);
  }
  uint32_t c0 = c_begin;
  for (; c0 + 16 <= c_end; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = CM_GCOL(cols[c0 + k])[i];
    fr.absorb(h, m, 64);
  }
  if (c0 < c_end) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < c_end) ? CM_GCOL(cols[c0 + k])[i] : 0u;
    fr.absorb(h, m, 4u * (c_end - c0));
  }
}
template <bool RFC>
__device__ __forceinline__ void merkle_node_quad(const uint32_t* children, const uint32_t* const* __restrict__ cols, uint32_t c_begin,
                                                 uint32_t c_end, uint32_t i, uint32_t q, uint32_t& h0, uint32_t& h1) {
  NodeFrame<RFC> fr(children != nullptr, c_end - c_begin);
  fr.init_quad(q, h0, h1);
  uint32_t m[16];
  if (children) {
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = children[k];
    fr.absorb_quad(h0, h1, m, q, 64);
  }
  uint32_t c0 = c_begin;
  for (; c0 + 16 <= c_end; c0 += 16) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = CM_GCOL(cols[c0 + k])[i];
    fr.absorb_quad(h0, h1, m, q, 64);
  }
  if (c0 < c_end) {
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < c_end) ? CM_GCOL(cols[c0 + k])[i] : 0u;
    fr.absorb_quad(h0, h1, m, q, 4u * (c_end - c0));
  }
}

// ---- wide layers: hundreds of columns on a few nodes (the 2^5-row layer of the interaction tree carries the ~1000 interaction
// columns of the idle components: 66 chained compressions per node) ------------------------------------------------------------
// The chain of a node is sequential, and a lone wave runs one quad-lane compression in ~0.7 us (tools/chain_lab.hip) — but with
// the message words fetched inside the chain every 16-column block also waited for a pointer load and a column load from HBM
// (~1.4 us per block, twice the hashing).  Here the whole BLOCK streams the layer's columns through LDS, one group of NT x R words
// ahead of the hashing quads: the column pointers are copied to LDS once, group g + 1 is loaded into registers before the quads
// hash group g and written to the other half of a double buffer afterwards, so no memory latency is left on the chain.
constexpr uint32_t WIDE_PTR_CAP = 2048;   // columns whose pointers fit the LDS table (16 KiB)
constexpr uint32_t WIDE_MIN_COLS = 48;    // below this a layer pays its one or two loads directly
template <int NT, int R>
struct WideLds {
  unsigned long long ptrs[WIDE_PTR_CAP];
  uint32_t buf[2][NT * R];   // group-local column-major: word (c, node) of a group at (c << log_n) + node
};
template <int NT, int R>
__device__ __forceinline__ bool wide_layer_ok(uint32_t ncols, uint32_t log_n) {
  return ncols >= WIDE_MIN_COLS && ncols <= WIDE_PTR_CAP && (16u << log_n) <= (uint32_t)(NT * R) && (4u << log_n) <= (uint32_t)NT;
}
// Every thread of the block calls this (barriers inside); the quad of lanes (node_local, q) of an `active` thread owns node
// node0 + node_local of the layer and returns its hash words h[q], h[4 + q].  `children` = the node's 16 child words or null.
template <bool RFC, int NT, int R>
__device__ __forceinline__ void merkle_wide_quad(WideLds<NT, R>& w, const uint32_t* children, bool active,
                                                 const uint32_t* const* __restrict__ cols, uint32_t c_begin, uint32_t c_end, uint32_t node0,
                                                 uint32_t log_n, uint32_t node_local, uint32_t q, uint32_t tid, uint32_t& h0, uint32_t& h1) {
  const uint32_t ncols = c_end - c_begin;
  for (uint32_t c = tid; c < ncols; c += NT) w.ptrs[c] = (unsigned long long)cols[c_begin + c];
  __syncthreads();
  const uint32_t gcols = (uint32_t)(NT * R) >> log_n;                 // columns of a group: a multiple of 16
  const uint32_t chunks = (ncols + 15u) >> 4, gchunks = gcols >> 4, ngroups = (chunks + gchunks - 1) / gchunks;
  const uint32_t nmask = (1u << log_n) - 1u;
  uint32_t v[R];
#define CM_WIDE_ISSUE(g)                                                                        \
  _Pragma("unroll") for (uint32_t k = 0; k < (uint32_t)R; k++) {                                \
    const uint32_t e = tid + k * NT, cl = (g) * gcols + (e >> log_n);                           \
    v[k] = cl < ncols ? CM_GCOL((const uint32_t*)w.ptrs[cl < ncols ? cl : 0u])[node0 + (e & nmask)] : 0u; \
  }
#define CM_WIDE_PUT(g)                                                                          \
  _Pragma("unroll") for (uint32_t k = 0; k < (uint32_t)R; k++) w.buf[(g) & 1u][tid + k * NT] = v[k];
  CM_WIDE_ISSUE(0u)
  CM_WIDE_PUT(0u)
  __syncthreads();
  NodeFrame<RFC> fr(children != nullptr, ncols);
  fr.init_quad(q, h0, h1);
  uint32_t m[16];
  if (active && children) {
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = children[k];
    fr.absorb_quad(h0, h1, m, q, 64);
  }
#pragma unroll 1
  for (uint32_t g = 0; g < ngroups; g++) {
    const bool more = g + 1 < ngroups;
    if (more) { CM_WIDE_ISSUE(g + 1u) }   // in flight while the quads hash group g
    if (active) {
      const uint32_t* b = w.buf[g & 1u];
      const uint32_t j_end = min(gchunks, chunks - g * gchunks);
#pragma unroll 1
      for (uint32_t j = 0; j < j_end; j++) {
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) m[k] = b[((j * 16u + k) << log_n) + node_local];   // columns past the end were staged as zeros
        const uint32_t c0 = (g * gchunks + j) * 16u;
        fr.absorb_quad(h0, h1, m, q, 4u * min(16u, ncols - c0));
      }
    }
    if (more) { CM_WIDE_PUT(g + 1u) }
    __syncthreads();
  }
#undef CM_WIDE_ISSUE
#undef CM_WIDE_PUT
}

// K consecutive layers (top_log, top_log-1, ..., top_log-K+1) in one launch.  A block hashes 256 nodes of the
// top layer, keeps them in LDS, then 128 parents, 64 grand-parents, ...  Every layer is written to HBM (the
// decommitment gathers need it) but intermediate layers are never re-read from HBM, and a 2^22 tree needs
// 5 launches instead of 16.  Mid-size layers are launch-latency-bound as separate kernels.
template <bool RFC>
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
      const uint32_t c_begin = a.col_begin[lv], c_end = a.col_end[lv];
      NodeFrame<RFC> fr(lv > 0 || a.prev, c_end - c_begin);
      uint32_t h[8];
      fr.init(h);
      uint32_t m[16];
      if (lv > 0 || a.prev) {
        const uint32_t* p = (lv == 0) ? a.prev + (size_t)i * 16 : buf[cur ^ 1] + tid * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = p[k];
        fr.absorb(h, m, 64);
      }
      uint32_t c0 = c_begin;
      for (; c0 + 16 <= c_end; c0 += 16) {
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) m[k] = CM_GCOL(a.cols[c0 + k])[i];
        fr.absorb(h, m, 64);
      }
      if (c0 < c_end) {
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) m[k] = (c0 + k < c_end) ? CM_GCOL(a.cols[c0 + k])[i] : 0u;
        fr.absorb(h, m, 4u * (c_end - c0));
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
template <bool RFC>
__global__ void __launch_bounds__(1024) k_merkle_tail(MerkleTailArgs a) {
  // One node per QUAD of lanes (b2s_compress_quad): these layers have <= 256 nodes, so lanes are plentiful and
  // the sequential compression chain per node (up to ~70 compressions for the 2^5-row idle components) is all
  // that matters.
  __shared__ uint32_t bufA[256 * 8];
  __shared__ uint32_t bufB[128 * 8];
  uint32_t* buf[2];  // layer top -> A (<= 256 nodes), top-1 -> B (<= 128), top-2 -> A, ...
  buf[0] = bufA;
  buf[1] = bufB;
  __shared__ WideLds<1024, 4> wide;
  const uint32_t node = threadIdx.x >> 2, q = threadIdx.x & 3u;
  int cur = 0;
  for (int l = (int)a.top_log; l >= 0; l--) {
    const uint32_t n = 1u << l;
    const bool from_global = (l == (int)a.top_log);
    const uint32_t* ch = (!from_global || a.prev) ? (from_global ? a.prev + (size_t)node * 16 : buf[cur ^ 1] + node * 16) : nullptr;
    uint32_t h0, h1;
    const bool is_wide = wide_layer_ok<1024, 4>(a.col_end[l] - a.col_begin[l], (uint32_t)l);   // block-uniform
    if (is_wide) merkle_wide_quad<RFC, 1024, 4>(wide, ch, node < n, a.cols, a.col_begin[l], a.col_end[l], 0u, (uint32_t)l, node, q, threadIdx.x, h0, h1);
    if (node < n) {
      if (!is_wide) merkle_node_quad<RFC>(ch, a.cols, a.col_begin[l], a.col_end[l], node, q, h0, h1);
      uint32_t* o = a.layers[l] + (size_t)node * 8;
      o[q] = h0; o[4 + q] = h1;
      buf[cur][node * 8 + q] = h0; buf[cur][node * 8 + 4 + q] = h1;
    }
    __syncthreads();
    cur ^= 1;
  }
}

// Layers 2^top_log .. 2^0 in one launch (see MerkleTopArgs).  Phase 1: 9 levels per block (256 nodes -> 1), one node
// per lane while >= 128 nodes are active, one node per quad of lanes below (half the dependent latency).  Phase 2: the
// block that draws the last ticket reads the 2^(top_log-8) nodes the blocks produced and finishes like k_merkle_tail.
template <bool RFC, int FOLD = 0>
__global__ void __launch_bounds__(256) k_merkle_top(MerkleTopArgs a) {
  __shared__ uint32_t bufA[256 * 8];
  __shared__ uint32_t bufB[128 * 8];
  __shared__ uint32_t s_last;
  __shared__ uint32_t s_x8[8], s_felt[4];
  __shared__ WideLds<256, 16> wide;   // phase 2 only
  uint32_t* buf[2];
  buf[0] = bufA;
  buf[1] = bufB;
  const uint32_t tid = threadIdx.x;
  int cur = 0;
  // ---- phase 1: layers top_log .. top_log - 8 of this block's 256-node slice ----
  uint32_t active = 256;
#pragma unroll 1
  for (uint32_t lv = 0; lv <= 8; lv++, active >>= 1) {
    const uint32_t l = a.top_log - lv;
    const uint32_t node0 = blockIdx.x * active;
    const uint32_t c_begin = a.col_begin[l], c_end = a.col_end[l];
    if (active >= 128) {
      if (tid < active) {
        const uint32_t i = node0 + tid;
        uint32_t h[8];
        const uint32_t* ch = lv > 0 ? buf[cur ^ 1] + tid * 16 : (a.prev ? a.prev + (size_t)i * 16 : nullptr);
        if (FOLD && lv == 0) {
          // the FRI layer itself: fold_line of the layer above (+ fold_circle of the quotient columns of this size), stored and
          // hashed as the leaf (k_fold_line / k_fold_line_circle + the leaf framing of k_merkle_layer: one 16-byte block)
          const uint2 s0 = reinterpret_cast<const uint2*>(a.x.fold_src[0])[i], s1 = reinterpret_cast<const uint2*>(a.x.fold_src[1])[i];
          const uint2 s2 = reinterpret_cast<const uint2*>(a.x.fold_src[2])[i], s3 = reinterpret_cast<const uint2*>(a.x.fold_src[3])[i];
          const QM31 alpha = QM31::from_u32(a.x.alpha);
          const QM31 f0(M31(s0.x), M31(s1.x), M31(s2.x), M31(s3.x)), f1(M31(s0.y), M31(s1.y), M31(s2.y), M31(s3.y));
          QM31 v = (f0 + f1) + alpha * ((f0 - f1) * M31(a.x.ixt[i] >> 1));
          if (FOLD == 2) {
            const uint2 c0 = reinterpret_cast<const uint2*>(a.x.fold_circ[0])[i], c1 = reinterpret_cast<const uint2*>(a.x.fold_circ[1])[i];
            const uint2 c2 = reinterpret_cast<const uint2*>(a.x.fold_circ[2])[i], c3 = reinterpret_cast<const uint2*>(a.x.fold_circ[3])[i];
            const QM31 ac = QM31::from_u32(a.x.alpha_c);
            const QM31 g0(M31(c0.x), M31(c1.x), M31(c2.x), M31(c3.x)), g1(M31(c0.y), M31(c1.y), M31(c2.y), M31(c3.y));
            v = v * (ac * ac) + ((g0 + g1) + ac * ((g0 - g1) * M31(a.x.iyt[i] >> 1)));
          }
          a.x.fold_dst[0][i] = v.a.a.v; a.x.fold_dst[1][i] = v.a.b.v; a.x.fold_dst[2][i] = v.b.a.v; a.x.fold_dst[3][i] = v.b.b.v;
          uint32_t z[16];
#pragma unroll
          for (int k = 0; k < 16; k++) z[k] = 0;
          z[0] = v.a.a.v; z[1] = v.a.b.v; z[2] = v.b.a.v; z[3] = v.b.b.v;
          NodeFrame<RFC> fr(false, 4);
          fr.init(h);
          fr.absorb(h, z, 16);
        } else
        merkle_node_thread<RFC>(ch, a.cols, c_begin, c_end, i, h);
        uint4* o = reinterpret_cast<uint4*>(a.layers[l] + (size_t)i * 8);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
#pragma unroll
        for (int k = 0; k < 8; k++) buf[cur][tid * 8 + k] = h[k];
      }
    } else {
      const uint32_t node = tid >> 2, q = tid & 3u;
      if (node < active) {
        const uint32_t i = node0 + node;
        uint32_t h0, h1;
        merkle_node_quad<RFC>(buf[cur ^ 1] + node * 16, a.cols, c_begin, c_end, i, q, h0, h1);
        uint32_t* o = a.layers[l] + (size_t)i * 8;
        o[q] = h0; o[4 + q] = h1;
        buf[cur][node * 8 + q] = h0; buf[cur][node * 8 + 4 + q] = h1;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  // ---- ticket: the last block to arrive owns the rest of the tree ----
  __threadfence();   // every lane's part of this block's layer (top_log - 8) node is visible device-wide ...
  __syncthreads();   // ... before lane 0 draws the ticket
  if (tid == 0) {
    const uint32_t t = atomicAdd(a.ticket, 1u);
    s_last = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();   // acquire: the other blocks' nodes
  if (tid == 0) *a.ticket = 0;   // ready for the next launch that uses this counter
  // ---- phase 2: layers top_log - 9 .. 0, one node per quad of lanes (<= 128 nodes) ----
  const int base_log = (int)a.top_log - 8;   // layer of gridDim.x nodes, in HBM
  const uint32_t node = tid >> 2, q = tid & 3u;
  cur = 0;
  for (int l = base_log - 1; l >= 0; l--) {
    const uint32_t n = 1u << l;
    const bool from_global = (l == base_log - 1);
    if (wide_layer_ok<256, 16>(a.col_end[l] - a.col_begin[l], (uint32_t)l)) {   // block-uniform; n <= 64 here
      const uint32_t* ch = from_global ? a.layers[base_log] + (size_t)node * 16 : buf[cur ^ 1] + node * 16;
      uint32_t h0, h1;
      merkle_wide_quad<RFC, 256, 16>(wide, ch, node < n, a.cols, a.col_begin[l], a.col_end[l], 0u, (uint32_t)l, node, q, tid, h0, h1);
      if (node < n) {
        uint32_t* o = a.layers[l] + (size_t)node * 8;
        o[q] = h0; o[4 + q] = h1;
        buf[cur][node * 8 + q] = h0; buf[cur][node * 8 + 4 + q] = h1;
      }
    } else {
      // 256 threads = 64 quads: loop when the layer has more nodes
      for (uint32_t nd = node; nd < n; nd += 64) {
        const uint32_t* ch = from_global ? a.layers[base_log] + (size_t)nd * 16 : buf[cur ^ 1] + nd * 16;
        uint32_t h0, h1;
        merkle_node_quad<RFC>(ch, a.cols, a.col_begin[l], a.col_end[l], nd, q, h0, h1);
        uint32_t* o = a.layers[l] + (size_t)nd * 8;
        o[q] = h0; o[4 + q] = h1;
        buf[cur][nd * 8 + q] = h0; buf[cur][nd * 8 + 4 + q] = h1;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  // the transcript step behind the tree (FRI layer trees): the root sits in buf[cur ^ 1][0..8)
  if (a.x.chan && tid < 4) chan_mix_root_draw_quad(tid, a.x.chan, buf[cur ^ 1], a.x.felt_out, a.x.root_log, s_x8, s_felt);
}

// One layer, one node per quad of lanes: for mid-size layers that carry hundreds of columns (poseidon2: 443
// columns at 2^10 rows) the chain length per node dominates, not the node count.
template <bool RFC>
__global__ void __launch_bounds__(256) k_merkle_layer_quad(uint32_t log_size, const uint32_t* __restrict__ prev,
                                                           const uint32_t* const* __restrict__ cols, uint32_t n_cols,
                                                           uint32_t* __restrict__ out) {
  __shared__ WideLds<256, 16> wide;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = t >> 2, q = t & 3u;
  const uint32_t n = 1u << log_size;
  uint32_t h0, h1;
  if (log_size >= 6 && wide_layer_ok<256, 16>(n_cols, 6u)) {   // full blocks of 64 nodes: the block streams its columns through LDS
    merkle_wide_quad<RFC, 256, 16>(wide, prev ? prev + (size_t)i * 16 : nullptr, true, cols, 0u, n_cols, blockIdx.x * 64u, 6u, threadIdx.x >> 2, q,
                                   threadIdx.x, h0, h1);
  } else {
    if (i >= n) return;  // whole quads drop out together
    merkle_node_quad<RFC>(prev ? prev + (size_t)i * 16 : nullptr, cols, 0u, n_cols, i, q, h0, h1);
  }
  out[(size_t)i * 8 + q] = h0;
  out[(size_t)i * 8 + 4 + q] = h1;
}

}  // namespace cm
