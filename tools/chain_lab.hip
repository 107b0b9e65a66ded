// Lone-wave latency lab (development tool, not part of the product): what bounds the Blake2s dependency chains of the tree
// tops / wide layers / FRI layer chains — the dependent-issue latency of a VALU op, the issue rate of a single wave, or the
// instruction count of the quad-lane compression.  One wave on one CU, timed with s_memtime (100 MHz constant clock) and
// wall_clock64.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cairo_m_amd/csrc tools/chain_lab.hip -o tools/chain_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "blake2s_dev.hpp"
using namespace cm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// ---- raw issue experiments: 64 x 16 = 1024 instructions per loop trip ----
__global__ void k_dep_add(uint32_t* out, uint32_t trips) {
  uint32_t a = threadIdx.x, b = out[0];
  for (uint32_t t = 0; t < trips; t++) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
    REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
  out[threadIdx.x] = a;
}
__global__ void k_ind_add(uint32_t* out, uint32_t trips) {
  uint32_t a = threadIdx.x, c = a + 1, d = a + 2, e = a + 3, b = out[0];
  for (uint32_t t = 0; t < trips; t++) {
    REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
  }
  out[threadIdx.x] = a + c + d + e;
}
__global__ void k_dep_xorrot(uint32_t* out, uint32_t trips) {   // xor -> alignbit chain (the body of G)
  uint32_t a = threadIdx.x, b = out[0];
  for (uint32_t t = 0; t < trips; t++) {
    REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_alignbit_b32 %0, %0, %0, 12\n v_add_u32 %0, %0, %1\n v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b));)
  }
  out[threadIdx.x] = a;
}
__global__ void k_dep_dpp(uint32_t* out, uint32_t trips) {   // add with a DPP-rotated operand behind a VALU write of it
  uint32_t a = threadIdx.x, b = out[0];
  for (uint32_t t = 0; t < trips; t++) {
    REP64(asm volatile("v_xor_b32 %0, %0, %1\n s_nop 1\n v_add_u32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
  }
  out[threadIdx.x] = a;
}

// ---- compression chains ----
// MODE 0: shipped quad form; 1: quad form with lane-uniform message words (no selects: lower bound of any schedule trick);
// 2: one node per lane (b2s_compress)
template <int MODE>
__global__ void k_chain(uint32_t* out, uint32_t n) {
  const uint32_t q = threadIdx.x & 3u;
  uint32_t m[16];
  for (int k = 0; k < 16; k++) m[k] = out[64 + k] + k;
  if (MODE == 2) {
    uint32_t h[8];
    for (int k = 0; k < 8; k++) h[k] = threadIdx.x + k;
    for (uint32_t i = 0; i < n; i++) { b2s_compress(h, m); for (int k = 0; k < 16; k++) m[k] ^= h[0]; }
    out[threadIdx.x] = h[0] ^ h[5];
  } else {
    uint32_t h0 = threadIdx.x, h1 = threadIdx.x * 3;
    for (uint32_t i = 0; i < n; i++) {
      if (MODE == 0) b2s_compress_quad(h0, h1, m, q);
      else {
        uint32_t a = h0, b = h1, c = 0x6A09E667u + q, d = 0x510E527Fu + q;
#define CM_QROUND_U(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                 \
  CM_QG(m[s0], m[s1])                                                                                                     \
  b = CM_QUAD_ROT(b, CM_QP(1, 2, 3, 0)); c = CM_QUAD_ROT(c, CM_QP(2, 3, 0, 1)); d = CM_QUAD_ROT(d, CM_QP(3, 0, 1, 2));  \
  CM_QG(m[s8], m[s9])                                                                                                     \
  b = CM_QUAD_ROT(b, CM_QP(3, 0, 1, 2)); c = CM_QUAD_ROT(c, CM_QP(2, 3, 0, 1)); d = CM_QUAD_ROT(d, CM_QP(1, 2, 3, 0));
        CM_QROUND_U(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
        CM_QROUND_U(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
        CM_QROUND_U(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
        CM_QROUND_U(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
        CM_QROUND_U(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
        CM_QROUND_U(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
        CM_QROUND_U(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
        CM_QROUND_U(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
        CM_QROUND_U(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
        CM_QROUND_U(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
        h0 ^= a ^ c; h1 ^= b ^ d;
      }
      for (int k = 0; k < 16; k++) m[k] ^= h0;   // every word changes: the selects cannot be hoisted out of the chain
    }
    out[threadIdx.x] = h0 ^ h1;
  }
}

// MODE 3 body: the diagonal rotations folded into the first use of b, c, d (VOP2 DPP operands) instead of six v_mov_dpp per round
#define CM_ROTI(x, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xf, 0xf, true))
#define CM_QG_F(x, y, pb, pc, pd)                                     \
  a = a + CM_ROTI(b, pb) + (x); d = rotr(CM_ROTI(d, pd) ^ a, 16);     \
  c = CM_ROTI(c, pc) + d;       b = rotr(CM_ROTI(b, pb) ^ c, 12);     \
  a = a + b + (y); d = rotr(d ^ a, 8);                                \
  c = c + d;       b = rotr(b ^ c, 7);
#define CM_QROUND_F(first, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                  \
  if (first) { CM_QG(b2s_sel4(q, m[s0], m[s2], m[s4], m[s6]), b2s_sel4(q, m[s1], m[s3], m[s5], m[s7])) }                          \
  else { CM_QG_F(b2s_sel4(q, m[s0], m[s2], m[s4], m[s6]), b2s_sel4(q, m[s1], m[s3], m[s5], m[s7]), CM_QP(3, 0, 1, 2), CM_QP(2, 3, 0, 1), CM_QP(1, 2, 3, 0)) } \
  CM_QG_F(b2s_sel4(q, m[s8], m[s10], m[s12], m[s14]), b2s_sel4(q, m[s9], m[s11], m[s13], m[s15]), CM_QP(1, 2, 3, 0), CM_QP(2, 3, 0, 1), CM_QP(3, 0, 1, 2))
__device__ __forceinline__ void b2s_compress_quad_f(uint32_t& h0, uint32_t& h1, const uint32_t (&m)[16], uint32_t q) {
  uint32_t a = h0, b = h1;
  uint32_t c = b2s_sel4(q, 0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au);
  uint32_t d = b2s_sel4(q, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u);
  CM_QROUND_F(true, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  CM_QROUND_F(false, 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  CM_QROUND_F(false, 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  CM_QROUND_F(false, 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  CM_QROUND_F(false, 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  CM_QROUND_F(false, 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  CM_QROUND_F(false, 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  CM_QROUND_F(false, 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  CM_QROUND_F(false, 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  CM_QROUND_F(false, 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  // the state is still in the diagonal frame: lane q holds b of column q + 1, c of q + 2, d of q + 3
  b = CM_ROTI(b, CM_QP(3, 0, 1, 2)); c = CM_ROTI(c, CM_QP(2, 3, 0, 1)); d = CM_ROTI(d, CM_QP(1, 2, 3, 0));
  h0 ^= a ^ c;
  h1 ^= b ^ d;
}
template <bool check>
__global__ void k_chain_f(uint32_t* out, uint32_t n) {
  const uint32_t q = threadIdx.x & 3u;
  uint32_t m[16];
  for (int k = 0; k < 16; k++) m[k] = out[64 + k] + k;
  uint32_t h0 = threadIdx.x, h1 = threadIdx.x * 3, g0 = h0, g1 = h1, bad = 0;
  for (uint32_t i = 0; i < n; i++) {
    b2s_compress_quad_f(h0, h1, m, q);
    if (check) { b2s_compress_quad(g0, g1, m, q); bad |= (g0 ^ h0) | (g1 ^ h1); }
    for (int k = 0; k < 16; k++) m[k] ^= h0;
  }
  out[threadIdx.x] = h0 ^ h1;
  if (check) out[128 + threadIdx.x] = bad;
}

// ---- per-lane compression with the four independent G functions of a half-round issued in LOCKSTEP, one kind of instruction at a
// time (tools/valu_lab.hip: add3 / xor / alignbit / add sequences run 10 % faster grouped by kind across independent chains than
// one dependent chain after the other, even at 8 waves per SIMD) ----
#define SB() __builtin_amdgcn_sched_barrier(0)
#define SD_HI(t, d, a) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "=&v"(t) : "v"(d), "v"(a))
#define SD_LO(t, d, a) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(t) : "v"(d), "v"(a))
#ifndef CM_IL_VARIANT
#define CM_IL_VARIANT 1
#endif
#if CM_IL_VARIANT == 1   /* groups of four by kind, rotation by 16 as SDWA halves */
#define CM_R16x4(a0, d0, a1, d1, a2, d2, a3, d3)                                                                                   \
  { uint32_t t0, t1, t2, t3;                                                                                                         \
    SD_HI(t0, d0, a0); SD_HI(t1, d1, a1); SD_HI(t2, d2, a2); SD_HI(t3, d3, a3); SB();                                                \
    SD_LO(t0, d0, a0); SD_LO(t1, d1, a1); SD_LO(t2, d2, a2); SD_LO(t3, d3, a3); SB();                                                \
    d0 = t0; d1 = t1; d2 = t2; d3 = t3; }
#define CM_XR4(b0, c0, b1, c1, b2, c2, b3, c3, n)                                                                                   \
  b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3; SB();                                                                                      \
  b0 = rotr(b0, n); b1 = rotr(b1, n); b2 = rotr(b2, n); b3 = rotr(b3, n); SB();
#elif CM_IL_VARIANT == 2   /* groups of four by kind, plain xor + alignbit everywhere */
#define CM_XR4(b0, c0, b1, c1, b2, c2, b3, c3, n)                                                                                   \
  b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3; SB();                                                                                      \
  b0 = rotr(b0, n); b1 = rotr(b1, n); b2 = rotr(b2, n); b3 = rotr(b3, n); SB();
#define CM_R16x4(a0, d0, a1, d1, a2, d2, a3, d3) CM_XR4(d0, a0, d1, a1, d2, a2, d3, a3, 16)
#elif CM_IL_VARIANT == 3   /* xor and rotate of the four chains alternating (xor0 rot0 xor1 rot1 ...), SDWA for 16 */
#define CM_R16x4(a0, d0, a1, d1, a2, d2, a3, d3)                                                                                   \
  { uint32_t t0, t1, t2, t3;                                                                                                         \
    SD_HI(t0, d0, a0); SD_HI(t1, d1, a1); SD_HI(t2, d2, a2); SD_HI(t3, d3, a3); SB();                                                \
    SD_LO(t0, d0, a0); SD_LO(t1, d1, a1); SD_LO(t2, d2, a2); SD_LO(t3, d3, a3); SB();                                                \
    d0 = t0; d1 = t1; d2 = t2; d3 = t3; }
#define CM_XR4(b0, c0, b1, c1, b2, c2, b3, c3, n)                                                                                   \
  b0 ^= c0; SB(); b1 ^= c1; SB(); b0 = rotr(b0, n); SB(); b2 ^= c2; SB(); b1 = rotr(b1, n); SB(); b3 ^= c3; SB(); b2 = rotr(b2, n); SB(); b3 = rotr(b3, n); SB();
#elif CM_IL_VARIANT == 4   /* as 1, with the VOP2 groups of neighbouring steps merged (c += d next to the following b ^= c) */
#define CM_R16x4(a0, d0, a1, d1, a2, d2, a3, d3)                                                                                   \
  { uint32_t t0, t1, t2, t3;                                                                                                         \
    SD_HI(t0, d0, a0); SD_HI(t1, d1, a1); SD_HI(t2, d2, a2); SD_HI(t3, d3, a3); SB();                                                \
    SD_LO(t0, d0, a0); SD_LO(t1, d1, a1); SD_LO(t2, d2, a2); SD_LO(t3, d3, a3); SB();                                                \
    d0 = t0; d1 = t1; d2 = t2; d3 = t3; }
#define CM_XR4(b0, c0, b1, c1, b2, c2, b3, c3, n)                                                                                   \
  b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3;                                                                                            \
  b0 = rotr(b0, n); b1 = rotr(b1, n); b2 = rotr(b2, n); b3 = rotr(b3, n); SB();
#endif
#define CM_G4(a0, b0, c0, d0, x0, y0, a1, b1, c1, d1, x1, y1, a2, b2, c2, d2, x2, y2, a3, b3, c3, d3, x3, y3)                      \
  a0 = a0 + b0 + (x0); a1 = a1 + b1 + (x1); a2 = a2 + b2 + (x2); a3 = a3 + b3 + (x3); SB();                                          \
  CM_R16x4(a0, d0, a1, d1, a2, d2, a3, d3)                                                                                           \
  c0 += d0; c1 += d1; c2 += d2; c3 += d3; SB();                                                                                      \
  CM_XR4(b0, c0, b1, c1, b2, c2, b3, c3, 12)                                                                                         \
  a0 = a0 + b0 + (y0); a1 = a1 + b1 + (y1); a2 = a2 + b2 + (y2); a3 = a3 + b3 + (y3); SB();                                          \
  CM_XR4(d0, a0, d1, a1, d2, a2, d3, a3, 8)                                                                                          \
  c0 += d0; c1 += d1; c2 += d2; c3 += d3; SB();                                                                                      \
  CM_XR4(b0, c0, b1, c1, b2, c2, b3, c3, 7)
#define CM_ROUND4(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                              \
  CM_G4(v0, v4, v8, v12, m[s0], m[s1], v1, v5, v9, v13, m[s2], m[s3], v2, v6, v10, v14, m[s4], m[s5], v3, v7, v11, v15, m[s6], m[s7]) \
  CM_G4(v0, v5, v10, v15, m[s8], m[s9], v1, v6, v11, v12, m[s10], m[s11], v2, v7, v8, v13, m[s12], m[s13], v3, v4, v9, v14, m[s14], m[s15])
__device__ __forceinline__ void b2s_compress_il(uint32_t (&h)[8], const uint32_t (&m)[16]) {
  uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
  uint32_t v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au;
  uint32_t v12 = 0x510E527Fu, v13 = 0x9B05688Cu, v14 = 0x1F83D9ABu, v15 = 0x5BE0CD19u;
  CM_ROUND4(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  CM_ROUND4(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
  CM_ROUND4(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
  CM_ROUND4(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
  CM_ROUND4(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
  CM_ROUND4(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
  CM_ROUND4(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
  CM_ROUND4(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
  CM_ROUND4(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
  CM_ROUND4(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
  h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
  h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}
template <bool CHECK>
__global__ void k_chain_il(uint32_t* out, uint32_t n) {
  uint32_t m[16];
  for (int k = 0; k < 16; k++) m[k] = out[64 + k] + k;
  uint32_t h[8], g[8], bad = 0;
  for (int k = 0; k < 8; k++) g[k] = h[k] = threadIdx.x + k;
  for (uint32_t i = 0; i < n; i++) {
    b2s_compress_il(h, m);
    if (CHECK) { b2s_compress(g, m); for (int k = 0; k < 8; k++) bad |= g[k] ^ h[k]; }
    for (int k = 0; k < 16; k++) m[k] ^= h[0];
  }
  out[threadIdx.x] = h[0] ^ h[5];
  if (CHECK) out[192 + threadIdx.x] = bad;
}

template <typename F>
static float time_ms(F f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}
int main() {
  uint32_t* d;
  CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  const uint32_t trips = 2000;
  for (uint32_t threads : {64u, 256u, 512u, 1024u}) {   // 1, 1, 2, 4 waves per SIMD of one CU
    float a = time_ms([&] { hipLaunchKernelGGL(k_dep_add, dim3(1), dim3(threads), 0, 0, d, trips); });
    float b = time_ms([&] { hipLaunchKernelGGL(k_ind_add, dim3(1), dim3(threads), 0, 0, d, trips); });
    float c = time_ms([&] { hipLaunchKernelGGL(k_dep_xorrot, dim3(1), dim3(threads), 0, 0, d, trips); });
    float e = time_ms([&] { hipLaunchKernelGGL(k_dep_dpp, dim3(1), dim3(threads), 0, 0, d, trips); });
    printf("threads %4u  ns per instruction of ONE wave: dependent v_add %.2f  4 independent v_add %.2f  xor/alignbit/add/add3 chain %.2f  "
           "xor,nop,add_dpp,add,add chain %.2f (per 4 VALU + nop)\n",
           threads, a * 1e6 / (trips * 256.0), b * 1e6 / (trips * 256.0), c * 1e6 / (trips * 256.0), e * 1e6 / (trips * 64.0));
  }
  const uint32_t n = 4000;
  for (uint32_t threads : {64u, 256u, 1024u}) {
    float q0 = time_ms([&] { hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(threads), 0, 0, d, n); });
    float q1 = time_ms([&] { hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(threads), 0, 0, d, n); });
    float t2 = time_ms([&] { hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(threads), 0, 0, d, n); });
    float f3 = time_ms([&] { hipLaunchKernelGGL(k_chain_f<false>, dim3(1), dim3(threads), 0, 0, d, n); });
    printf("threads %4u  us per chained compression: quad %.3f  quad without selects %.3f  one lane per node %.3f  quad with DPP operands %.3f\n", threads,
           q0 * 1e3 / n, q1 * 1e3 / n, t2 * 1e3 / n, f3 * 1e3 / n);
  }
  // whole-GPU rate of the per-lane compression with everything in registers (no loads, no stores): the ceiling the Merkle layer
  // kernels can approach — 8 / 4 / 2 waves per SIMD
  for (uint32_t blocks : {256u * 8u, 256u * 4u, 256u * 2u}) {
    const uint32_t nn = 400;
    float tt = time_ms([&] { hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(256), 0, 0, d, nn); });
    printf("%u blocks x 256 threads: %.3e compressions/s (register-only per-lane chain, 16 extra xors per compression)\n", blocks,
           (double)blocks * 256.0 * nn / (tt * 1e-3));
  }
  for (uint32_t blocks : {256u * 8u, 256u * 4u, 256u * 2u}) {
    const uint32_t nn = 400;
    float tt = time_ms([&] { hipLaunchKernelGGL(k_chain_il<false>, dim3(blocks), dim3(256), 0, 0, d, nn); });
    printf("%u blocks x 256 threads: %.3e compressions/s (the four G functions of a half-round in lockstep, grouped by instruction kind)\n", blocks,
           (double)blocks * 256.0 * nn / (tt * 1e-3));
  }
  {
    hipLaunchKernelGGL(k_chain_il<true>, dim3(1), dim3(64), 0, 0, d, 50u);
    std::vector<uint32_t> hh(1024);
    CK(hipMemcpy(hh.data(), d, 4096, hipMemcpyDeviceToHost));
    uint32_t bad2 = 0;
    for (int i = 0; i < 64; i++) bad2 |= hh[192 + i];
    printf("lockstep form equals b2s_compress: %s\n", bad2 ? "NO" : "yes");
  }
  hipLaunchKernelGGL(k_chain_f<true>, dim3(1), dim3(64), 0, 0, d, 100u);
  std::vector<uint32_t> host(1024);
  CK(hipMemcpy(host.data(), d, 4096, hipMemcpyDeviceToHost));
  uint32_t bad = 0;
  for (int i = 0; i < 64; i++) bad |= host[128 + i];
  printf("DPP-operand form equals the shipped quad form: %s\n", bad ? "NO" : "yes");
  return 0;
}
