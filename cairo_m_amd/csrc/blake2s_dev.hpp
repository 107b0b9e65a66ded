// Device-side Blake2s for gfx950: the fully unrolled register compression used by the Merkle / grind
// kernels (t = f = 0 form, Stwo's Merkle node framing), a compact generic RFC 7693 compression, and the
// two transcript steps of the FRI commit phase (Blake2sChannel::mix_root + draw_felt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field.hpp"

namespace cm {

static __device__ __constant__ uint32_t B2S_IV_D[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                                0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __funnelshift_r(x, x, n); }

// rotr(d ^ a, 16) as two SDWA xors (each writes one 16-bit half of the result from the opposite halves of the operands:
// 2 + 2 issue cycles) instead of v_xor (2) + v_alignbit (4, a VOP3) — tools/valu_lab.hip rates; 80 of the 320 rotations of a
// compression.  Measured in one GPU session against the plain build: k_merkle_layer 3.08 -> 2.93 ms per proof (-5 %).
// CM_B2S_SDWA=0: plain form.
#ifndef CM_B2S_SDWA
#define CM_B2S_SDWA 1
#endif
__device__ __forceinline__ uint32_t xor_rotr16(uint32_t d, uint32_t a) {
#if CM_B2S_SDWA
  uint32_t t;
  // gfx940+ dst_sel forwarding hazard: a VALU reading a VGPR right behind a partial (dst_sel) write of it needs one wait state,
  // and the assembler does not insert it inside inline asm
  asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\t"
      "s_nop 0\n\t"
      "v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t"
      "s_nop 0"
      : "=&v"(t)
      : "v"(d), "v"(a));
  return t;
#else
  return rotr(d ^ a, 16);
#endif
}

#define CM_G(a, b, c, d, x, y)                 \
  a = a + b + (x); d = xor_rotr16(d, a);       \
  c = c + d;       b = rotr(b ^ c, 12);        \
  a = a + b + (y); d = rotr(d ^ a, 8);         \
  c = c + d;       b = rotr(b ^ c, 7);

#define CM_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  CM_G(v0, v4, v8, v12, m[s0], m[s1])                                                  \
  CM_G(v1, v5, v9, v13, m[s2], m[s3])                                                  \
  CM_G(v2, v6, v10, v14, m[s4], m[s5])                                                 \
  CM_G(v3, v7, v11, v15, m[s6], m[s7])                                                 \
  CM_G(v0, v5, v10, v15, m[s8], m[s9])                                                 \
  CM_G(v1, v6, v11, v12, m[s10], m[s11])                                               \
  CM_G(v2, v7, v8, v13, m[s12], m[s13])                                                \
  CM_G(v3, v4, v9, v14, m[s14], m[s15])

// h <- F(h, m, t0=0, t1=0, f0=0, f1=0)
__device__ __forceinline__ void b2s_compress(uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t t = 0, uint32_t f0 = 0) {
  uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
  uint32_t v12 = 0x510E527Fu ^ t, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu ^ f0, v15 = 0x5BE0CD19u;
  CM_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  CM_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  CM_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  CM_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  CM_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  CM_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  CM_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  CM_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  CM_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  CM_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// mix_u64 of the channel whose digest is `dg` (framing.hpp switch `mix_u64`): U32S = false -> the raw compression
// F(digest, [lo, hi, 0...], t = 0, f = 0); U32S = true -> Blake2s256(digest || le32(lo) || le32(hi)), one final 40-byte block
template <bool U32S>
__device__ __forceinline__ void mix_u64_dev(const uint32_t (&dg)[8], uint64_t v, uint32_t (&h)[8]) {
  uint32_t m[16] = {0};
  if (U32S) {
#pragma unroll
    for (int k = 0; k < 8; k++) m[k] = dg[k];
    m[8] = (uint32_t)v; m[9] = (uint32_t)(v >> 32);
    h[0] = 0x6A09E667u ^ 0x01010020u; h[1] = 0xBB67AE85u; h[2] = 0x3C6EF372u; h[3] = 0xA54FF53Au;
    h[4] = 0x510E527Fu; h[5] = 0x9B05688Cu; h[6] = 0x1F83D9ABu; h[7] = 0x5BE0CD19u;
    b2s_compress(h, m, 40, 0xFFFFFFFFu);
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = dg[k];
    m[0] = (uint32_t)v; m[1] = (uint32_t)(v >> 32);
    b2s_compress(h, m);
  }
}

// ---- quad-lane compression: one Blake2s state spread over 4 adjacent lanes ----------------------------------
// Wide-and-short Merkle layers (few nodes, hundreds of columns) and the top levels of every tree are one
// sequential compression chain per node; with one node per lane a lone wave needs ~2.7 us per compression.
// Lane q of a quad holds column q of the 4x4 state (a = v[q], b = v[4+q], c = v[8+q], d = v[12+q]); the
// diagonal step rotates b, c, d by 1, 2, 3 lanes with DPP quad_perm moves.  Every lane keeps all 16 message
// words (the 4 lanes of a quad load the same addresses: one fetch) and selects its two words per G with
// lane-id selects.  ~1.4 us per compression (tools/chain_lab.hip), bit-identical to b2s_compress.
#define CM_QP(p0, p1, p2, p3) ((p0) | ((p1) << 2) | ((p2) << 4) | ((p3) << 6))
#define CM_QUAD_ROT(x, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(x), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ uint32_t b2s_sel4(uint32_t q, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
  uint32_t lo = (q & 1u) ? x1 : x0, hi = (q & 1u) ? x3 : x2;
  return (q & 2u) ? hi : lo;
}
#define CM_QG(x, y)   /* latency-bound chains: the plain form (the SDWA pair carries two wait states) */ \
  a = a + b + (x); d = rotr(d ^ a, 16);         \
  c = c + d;       b = rotr(b ^ c, 12);         \
  a = a + b + (y); d = rotr(d ^ a, 8);          \
  c = c + d;       b = rotr(b ^ c, 7);
#define CM_QROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                           \
  CM_QG(b2s_sel4(q, m[s0], m[s2], m[s4], m[s6]), b2s_sel4(q, m[s1], m[s3], m[s5], m[s7]))                         \
  b = CM_QUAD_ROT(b, CM_QP(1, 2, 3, 0)); c = CM_QUAD_ROT(c, CM_QP(2, 3, 0, 1)); d = CM_QUAD_ROT(d, CM_QP(3, 0, 1, 2)); \
  CM_QG(b2s_sel4(q, m[s8], m[s10], m[s12], m[s14]), b2s_sel4(q, m[s9], m[s11], m[s13], m[s15]))                   \
  b = CM_QUAD_ROT(b, CM_QP(3, 0, 1, 2)); c = CM_QUAD_ROT(c, CM_QP(2, 3, 0, 1)); d = CM_QUAD_ROT(d, CM_QP(1, 2, 3, 0));
// h0 = h[q], h1 = h[4 + q] of the node's chaining value; q = lane & 3; all 4 lanes of the quad must be active
__device__ __forceinline__ void b2s_compress_quad(uint32_t& h0, uint32_t& h1, const uint32_t (&m)[16], uint32_t q, uint32_t t = 0,
                                                  uint32_t f0 = 0) {
  uint32_t a = h0, b = h1;
  uint32_t c = b2s_sel4(q, 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
  uint32_t d = b2s_sel4(q, 0x510E527Fu ^ t, 0x9B05688Cu, 0x1F83D9ABu ^ f0, 0x5BE0CD19u);
  CM_QROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  CM_QROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  CM_QROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  CM_QROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  CM_QROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  CM_QROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  CM_QROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  CM_QROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  CM_QROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  CM_QROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  h0 ^= a ^ c;
  h1 ^= b ^ d;
}

// ---- Merkle node framing (framing.hpp switch `hash_node`) ------------------------------------------------------------
// RFC = false: Stwo's raw compression chain from the zero state, t = f = 0 (the default; every kernel below compiles to
// exactly the code it had before the switch existed).  RFC = true: RFC 7693 Blake2s-256 of the same byte string
// left || right || le32(column values): parameter block in h[0], running byte counter, final flag on the last block.
template <bool RFC>
struct NodeFrame {
  uint32_t total, done;
  __device__ __forceinline__ NodeFrame(bool has_children, uint32_t n_cols)
      : total(RFC ? (has_children ? 64u : 0u) + 4u * n_cols : 0u), done(0) {}
  __device__ __forceinline__ void init(uint32_t (&h)[8]) const {
    if (RFC) {
      h[0] = 0x6A09E667u ^ 0x01010020u; h[1] = 0xBB67AE85u; h[2] = 0x3C6EF372u; h[3] = 0xA54FF53Au;
      h[4] = 0x510E527Fu; h[5] = 0x9B05688Cu; h[6] = 0x1F83D9ABu; h[7] = 0x5BE0CD19u;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) h[k] = 0;
    }
  }
  __device__ __forceinline__ void init_quad(uint32_t q, uint32_t& h0, uint32_t& h1) const {
    if (RFC) {
      h0 = b2s_sel4(q, 0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
      h1 = b2s_sel4(q, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u);
    } else {
      h0 = 0; h1 = 0;
    }
  }
  // one 64-byte block carrying `bytes` message bytes (zero padded)
  __device__ __forceinline__ void absorb(uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t bytes) {
    if (RFC) { done += bytes; b2s_compress(h, m, done, done == total ? 0xFFFFFFFFu : 0u); }
    else b2s_compress(h, m);
  }
  __device__ __forceinline__ void absorb_quad(uint32_t& h0, uint32_t& h1, const uint32_t (&m)[16], uint32_t q, uint32_t bytes) {
    if (RFC) { done += bytes; b2s_compress_quad(h0, h1, m, q, done, done == total ? 0xFFFFFFFFu : 0u); }
    else b2s_compress_quad(h0, h1, m, q);
  }
};

// ---- device-side Fiat-Shamir steps for the FRI commit phase -----------------------------------------------
// Between two FRI layers the transcript only does mix_root(layer root) and draw_felt() (the folding
// challenge).  Doing those two hashes on the device keeps the whole commit phase on the stream:
// no device->host root copy, host hash, and relaunch per layer.  The host replays the same steps on its own
// channel afterwards (from the recorded roots) and checks the challenges agree.
// chan = {digest[8], n_sent}.  Blake2sChannel::mix_root then draw_felt (host twin: host_channel.hpp):
//   mix_root: digest = Blake2s256(digest || root)  (64 bytes = one final block, t = 64)
//   draw_felt: Blake2s256(digest || le32(n_sent) || 0^28 || 0x00) (65 bytes), retried until all 8 words < 2P.
// Runs on ONE QUAD of lanes (lanes 0..3 of a wave, all four active, q = lane): half the latency of a one-thread form.
// `root` may live in LDS; `x8` is an 8-word LDS scratch; the challenge also lands in lds_felt[0..4).
static __device__ __noinline__ void chan_mix_root_draw_quad(uint32_t q, uint32_t* chan, const uint32_t* root, uint32_t* felt_out,
                                                            uint32_t* root_log, volatile uint32_t* x8, volatile uint32_t* lds_felt) {
  uint32_t m[16];
  for (int i = 0; i < 8; i++) { m[i] = chan[i]; m[8 + i] = root[i]; }
  root_log[q] = root[q]; root_log[4 + q] = root[4 + q];
  const uint32_t iv0 = b2s_sel4(q, 0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
  const uint32_t iv1 = b2s_sel4(q, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u);
  uint32_t h0 = iv0, h1 = iv1;
  b2s_compress_quad(h0, h1, m, q, 64, 0xFFFFFFFFu);      // mix_root
  chan[q] = h0; chan[4 + q] = h1;
  x8[q] = h0; x8[4 + q] = h1;                              // the 4 lanes run in lockstep; LDS ops of a wave stay in order
  for (int i = 0; i < 8; i++) m[i] = x8[i];
  uint32_t n_sent = 0;
  for (;;) {                                               // draw_felt
    for (int i = 8; i < 16; i++) m[i] = 0;
    m[8] = n_sent++;
    uint32_t d0 = iv0, d1 = iv1;
    b2s_compress_quad(d0, d1, m, q, 64, 0);
    uint32_t z[16];
    for (int i = 0; i < 16; i++) z[i] = 0;
    b2s_compress_quad(d0, d1, z, q, 65, 0xFFFFFFFFu);
    const bool ok = d0 < 2u * P && d1 < 2u * P;
    if ((__ballot(ok) & 0xFull) != 0xFull) continue;
    const uint32_t f = M31::from_u32(d0).v;
    felt_out[q] = f;
    lds_felt[q] = f;
    break;
  }
  if (q == 0) chan[8] = n_sent;
}

}  // namespace cm
