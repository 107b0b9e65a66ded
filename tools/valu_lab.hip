// Integer-VALU throughput probe for gfx950 (development tool): the hot kernels of this library are integer
// (Blake2s, M31 arithmetic), and the public guides only tabulate floating-point / MFMA rates.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/valu_lab.hip -o tools/valu_lab
// Each kernel runs N_ITER x 64 instructions of one opcode on 8 independent accumulators per lane
// (enough ILP to hide the dependent-issue latency), all CUs saturated with 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int N_ITER = 256;

#define BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)
#define REP8(X) X X X X X X X X

#define DEFKERNEL(NAME, ASM_LINE)                                                              \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t s0, uint32_t s1) {      \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = s0 ^ threadIdx.x, c = s1 + blockIdx.x;                                       \
    for (int it = 0; it < N_ITER; it++) {                                                     \
      REP8(ASM_LINE(a0) ASM_LINE(a1) ASM_LINE(a2) ASM_LINE(a3) ASM_LINE(a4) ASM_LINE(a5) ASM_LINE(a6) ASM_LINE(a7)) \
    }                                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
  }

#define L_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_ADD(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_ADD3(a) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define L_ALIGNBIT(a) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a));
#define L_ALIGNBYTE(a) asm volatile("v_alignbyte_b32 %0, %0, %0, 1" : "+v"(a));
#define L_PERM(a) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a) : "v"(c));
#define L_LSHR(a) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(a));
#define L_LSHLOR(a) asm volatile("v_lshl_or_b32 %0, %0, 25, %1" : "+v"(a) : "v"(b));
#define L_MULLO(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_MULHI(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_MUL24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_MAD24(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define L_MIN(a) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_SUB(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define L_BFE(a) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(a));
#define L_ANDOR(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define L_FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define L_CNDMASK(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : );

#define L_XOR_SDWA(a) asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a) : "v"(b), "v"(c));
#define L_XOR_SDWA_PAD(a) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a) : "v"(b));
#define L_OR_SDWA(a) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a) : "v"(b));
#define L_AND_LIT(a) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(a));
#define L_AND_SGPR(a) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "s"(s0));
#define L_ADD_LIT(a) asm volatile("v_add_u32 %0, 0x7fffffff, %0" : "+v"(a));
#define L_ADD_INLINE(a) asm volatile("v_add_u32 %0, 17, %0" : "+v"(a));
#define L_SUBREV_LIT(a) asm volatile("v_subrev_co_u32 %0, vcc, 0x7fffffff, %0" : "+v"(a) : : "vcc");
#define L_SUBREV_SGPR(a) asm volatile("v_subrev_co_u32 %0, vcc, %1, %0" : "+v"(a) : "s"(s0) : "vcc");
// v_cndmask with a live VCC (written once by a VALU compare before the loop) and the e64 form with an SGPR-pair mask
__global__ void __launch_bounds__(256) k_cndmask_vcc(uint32_t* out, uint32_t s0, uint32_t s1) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t b = s0 ^ threadIdx.x, c = s1 + blockIdx.x;
  asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_nop 4" : : "v"(b), "v"(c) : "vcc");
#define L_CND_VCC(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : );
  for (int it = 0; it < N_ITER; it++) {
    REP8(L_CND_VCC(a0) L_CND_VCC(a1) L_CND_VCC(a2) L_CND_VCC(a3) L_CND_VCC(a4) L_CND_VCC(a5) L_CND_VCC(a6) L_CND_VCC(a7))
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(256) k_cndmask_e64(uint32_t* out, uint32_t s0, uint32_t s1) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t b = s0 ^ threadIdx.x;
  unsigned long long m = ((unsigned long long)s0 << 32) | s1;
#define L_CND_E64(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m));
  for (int it = 0; it < N_ITER; it++) {
    REP8(L_CND_E64(a0) L_CND_E64(a1) L_CND_E64(a2) L_CND_E64(a3) L_CND_E64(a4) L_CND_E64(a5) L_CND_E64(a6) L_CND_E64(a7))
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// the two conditional-subtract forms of M31 arithmetic, as dependent pairs on 8 accumulators
#define L_CSUB_MIN(a) asm volatile("v_subrev_u32 %1, 0x7fffffff, %0\n v_min_u32 %0, %0, %1" : "+v"(a), "=&v"(tmp));
#define L_CSUB_CND(a) asm volatile("v_subrev_co_u32 %1, vcc, 0x7fffffff, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a), "=&v"(tmp) : : "vcc");
__global__ void __launch_bounds__(256) k_csub_min(uint32_t* out, uint32_t s0, uint32_t s1) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, tmp;
  for (int it = 0; it < N_ITER / 2; it++) {
    REP8(L_CSUB_MIN(a0) L_CSUB_MIN(a1) L_CSUB_MIN(a2) L_CSUB_MIN(a3) L_CSUB_MIN(a4) L_CSUB_MIN(a5) L_CSUB_MIN(a6) L_CSUB_MIN(a7))
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void __launch_bounds__(256) k_csub_cnd(uint32_t* out, uint32_t s0, uint32_t s1) {
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, tmp;
  for (int it = 0; it < N_ITER / 2; it++) {
    REP8(L_CSUB_CND(a0) L_CSUB_CND(a1) L_CSUB_CND(a2) L_CSUB_CND(a3) L_CSUB_CND(a4) L_CSUB_CND(a5) L_CSUB_CND(a6) L_CSUB_CND(a7))
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
DEFKERNEL(k_and_lit, L_AND_LIT)
DEFKERNEL(k_and_sgpr, L_AND_SGPR)
DEFKERNEL(k_add_lit, L_ADD_LIT)
DEFKERNEL(k_add_inline, L_ADD_INLINE)
DEFKERNEL(k_subrev_lit, L_SUBREV_LIT)
DEFKERNEL(k_subrev_sgpr, L_SUBREV_SGPR)
DEFKERNEL(k_xor_sdwa, L_XOR_SDWA)
DEFKERNEL(k_xor_sdwa_pad, L_XOR_SDWA_PAD)
DEFKERNEL(k_or_sdwa, L_OR_SDWA)
DEFKERNEL(k_xor, L_XOR)
DEFKERNEL(k_add, L_ADD)
DEFKERNEL(k_add3, L_ADD3)
DEFKERNEL(k_alignbit, L_ALIGNBIT)
DEFKERNEL(k_alignbyte, L_ALIGNBYTE)
DEFKERNEL(k_perm, L_PERM)
DEFKERNEL(k_lshr, L_LSHR)
DEFKERNEL(k_lshlor, L_LSHLOR)
DEFKERNEL(k_mullo, L_MULLO)
DEFKERNEL(k_mulhi, L_MULHI)
DEFKERNEL(k_mul24, L_MUL24)
DEFKERNEL(k_mad24, L_MAD24)
DEFKERNEL(k_min, L_MIN)
DEFKERNEL(k_and, L_AND)
DEFKERNEL(k_sub, L_SUB)
DEFKERNEL(k_bfe, L_BFE)
DEFKERNEL(k_andor, L_ANDOR)
DEFKERNEL(k_fma, L_FMA)
DEFKERNEL(k_cndmask, L_CNDMASK)

// 64-bit multiply-add: v_mad_u64_u32 (full 64-bit product + 64-bit addend)
__global__ void __launch_bounds__(256) k_mad64(uint32_t* out, uint32_t s0, uint32_t s1) {
  unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  uint32_t b = s0 ^ threadIdx.x, c = s1 + blockIdx.x;
#define L_MAD64(a) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "vcc");
  for (int it = 0; it < N_ITER; it++) {
    REP8(L_MAD64(a0) L_MAD64(a1) L_MAD64(a2) L_MAD64(a3) L_MAD64(a4) L_MAD64(a5) L_MAD64(a6) L_MAD64(a7))
  }
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}

typedef void (*kern_t)(uint32_t*, uint32_t, uint32_t);
struct K { const char* name; kern_t fn; };

// ---- op MIXES: do the single-op rates add up when VOP2 and VOP3 alternate?  (They do not: the Blake2s compression looping on
// registers reaches 3.85e10/s where the sum of its ops' single rates gives 4.9e10 — tools/chain_lab.hip.) ----
#define DEFMIX(NAME, BODY, OPS)                                                                \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t s0, uint32_t s1) {      \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = s0 ^ threadIdx.x, c = s1 + blockIdx.x;                                       \
    for (int it = 0; it < N_ITER; it++) { REP8(BODY) }                                        \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
  }                                                                                           \
  static const int NAME##_ops = OPS;
#define ALL8(L) L(a0) L(a1) L(a2) L(a3) L(a4) L(a5) L(a6) L(a7)
#define PAIR_XA(a) L_XOR(a) L_ALIGNBIT(a)
// xor and alignbit alternate every instruction (each pair dependent, 8 independent accumulators)
DEFMIX(k_mix_xa_alt, ALL8(PAIR_XA), 16)
// the same ops in groups of 8 of one kind
DEFMIX(k_mix_xa_grp, ALL8(L_XOR) ALL8(L_ALIGNBIT), 16)
// one G-like sequence per accumulator: add3, xor, alignbit, add, xor, alignbit (the Blake2s ratio), alternating kinds ...
#define SEQ_G(a) L_ADD3(a) L_XOR(a) L_ALIGNBIT(a) L_ADD(a) L_XOR(a) L_ALIGNBIT(a)
DEFMIX(k_mix_g_alt, ALL8(SEQ_G), 48)
// ... and grouped by kind across the 8 accumulators
DEFMIX(k_mix_g_grp, ALL8(L_ADD3) ALL8(L_XOR) ALL8(L_ALIGNBIT) ALL8(L_ADD) ALL8(L_XOR) ALL8(L_ALIGNBIT), 48)
// groups of 4 (what the four independent G functions of a Blake2s half-round allow)
#define ALL4A(L) L(a0) L(a1) L(a2) L(a3)
#define ALL4B(L) L(a4) L(a5) L(a6) L(a7)
DEFMIX(k_mix_g_grp4, ALL4A(L_ADD3) ALL4A(L_XOR) ALL4A(L_ALIGNBIT) ALL4A(L_ADD) ALL4A(L_XOR) ALL4A(L_ALIGNBIT)
                     ALL4B(L_ADD3) ALL4B(L_XOR) ALL4B(L_ALIGNBIT) ALL4B(L_ADD) ALL4B(L_XOR) ALL4B(L_ALIGNBIT), 48)

int main() {
  K ks[] = {{"v_cndmask_b32 (vcc set by v_cmp)", k_cndmask_vcc}, {"v_cndmask_b32_e64 (sgpr mask)", k_cndmask_e64},
            {"csub: sub + min (pair = 1 op)", k_csub_min}, {"csub: sub_co + cndmask (pair = 1 op)", k_csub_cnd},
            {"v_and_b32 literal", k_and_lit}, {"v_and_b32 sgpr", k_and_sgpr}, {"v_add_u32 literal", k_add_lit}, {"v_add_u32 inline const", k_add_inline},
            {"v_subrev_co_u32 literal", k_subrev_lit}, {"v_subrev_co_u32 sgpr", k_subrev_sgpr},
            {"v_xor_b32_sdwa dst WORD_1 preserve", k_xor_sdwa}, {"v_xor_b32_sdwa dst WORD_1 pad", k_xor_sdwa_pad},
            {"v_or_b32_sdwa src0 WORD_1", k_or_sdwa}, {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_alignbit_b32", k_alignbit},
            {"v_alignbyte_b32", k_alignbyte}, {"v_perm_b32", k_perm}, {"v_lshrrev_b32", k_lshr}, {"v_lshl_or_b32", k_lshlor},
            {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24},
            {"v_mad_u32_u24", k_mad24}, {"v_min_u32", k_min}, {"v_and_b32", k_and}, {"v_sub_u32", k_sub},
            {"v_bfe_u32", k_bfe}, {"v_and_or_b32", k_andor}, {"v_fma_f32", k_fma}, {"v_cndmask_b32", k_cndmask},
            {"v_mad_u64_u32", k_mad64}};
  const int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  uint32_t* out = nullptr;
  CK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    double lane_ops = 4.0 * blocks * 256.0 * N_ITER * 64.0;
    double tops = lane_ops / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs, 64 lanes per wave instruction
    double cyc = (2.4e9 * 1024.0) / (tops * 1e12 / 64.0);
    printf("%-36s %7.2f T lane-ops/s   %5.2f cycles / wave-instruction / SIMD (at 2.4 GHz)\n", k.name, tops, cyc);
  }
  struct KM { const char* name; void (*fn)(uint32_t*, uint32_t, uint32_t); int ops; } ms_[] = {
      {"mix xor,alignbit alternating", k_mix_xa_alt, k_mix_xa_alt_ops}, {"mix 8 xor then 8 alignbit", k_mix_xa_grp, k_mix_xa_grp_ops},
      {"mix add3,xor,align,add,xor,align per chain", k_mix_g_alt, k_mix_g_alt_ops}, {"same ops grouped by kind x8", k_mix_g_grp, k_mix_g_grp_ops},
      {"same ops grouped by kind x4", k_mix_g_grp4, k_mix_g_grp4_ops}};
  for (auto& k : ms_) {
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    double lane_ops = 4.0 * blocks * 256.0 * N_ITER * 8.0 * k.ops;
    double tops = lane_ops / (ms * 1e-3) / 1e12;
    printf("%-44s %7.2f T lane-ops/s   %5.3f ns / wave-instruction / SIMD\n", k.name, tops, 1024.0 * 64.0 / (tops * 1e12) * 1e9);
  }
  return 0;
}
