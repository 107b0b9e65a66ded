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

int main() {
  K ks[] = {{"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_alignbit_b32", k_alignbit},
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
    printf("%-18s %7.2f T lane-ops/s   %5.2f cycles / wave-instruction / SIMD (at 2.4 GHz)\n", k.name, tops, cyc);
  }
  return 0;
}
