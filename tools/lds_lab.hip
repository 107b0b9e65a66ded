// LDS / cross-lane throughput probe for gfx950 (development tool, companion of valu_lab): how many cycles a CU needs per
// wave-wide ds_read/ds_write of each width (conflict-free addresses), per ds_bpermute, DPP move and permlane swap.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lds_lab.hip -o tools/lds_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int N_ITER = 256;
#define REP8(X) X X X X X X X X

__global__ void __launch_bounds__(256) k_read32(uint32_t* out, uint32_t stride) {
  __shared__ uint32_t lds[4096];
  for (uint32_t i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  uint32_t addr = ((threadIdx.x * stride) & 1023u) * 4, acc = 0;
  for (int it = 0; it < N_ITER; it++) {
    uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4096\n ds_read_b32 %2, %8 offset:8192\n ds_read_b32 %3, %8 offset:12288\n"
                 "ds_read_b32 %4, %8 offset:128\n ds_read_b32 %5, %8 offset:4224\n ds_read_b32 %6, %8 offset:8320\n ds_read_b32 %7, %8 offset:12416\n"
                 "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
    acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_write32(uint32_t* out, uint32_t stride) {
  __shared__ uint32_t lds[4096];
  uint32_t addr = ((threadIdx.x * stride) & 1023u) * 4, v = threadIdx.x;
  for (int it = 0; it < N_ITER; it++) {
    asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:4096\n ds_write_b32 %0, %1 offset:8192\n ds_write_b32 %0, %1 offset:12288\n"
                 "ds_write_b32 %0, %1 offset:128\n ds_write_b32 %0, %1 offset:4224\n ds_write_b32 %0, %1 offset:8320\n ds_write_b32 %0, %1 offset:12416\n"
                 "s_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(v) : "memory");
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_read64(uint32_t* out, uint32_t stride) {
  __shared__ uint32_t lds[4096];
  for (uint32_t i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  uint32_t addr = ((threadIdx.x * stride) & 511u) * 8, acc = 0;
  for (int it = 0; it < N_ITER; it++) {
    unsigned long long r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:4096\n ds_read_b64 %2, %8 offset:8192\n ds_read_b64 %3, %8 offset:12288\n"
                 "ds_read_b64 %4, %8 offset:128\n ds_read_b64 %5, %8 offset:4224\n ds_read_b64 %6, %8 offset:8320\n ds_read_b64 %7, %8 offset:12416\n"
                 "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
    acc ^= (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_read128(uint32_t* out, uint32_t stride) {
  __shared__ uint32_t lds[4096];
  for (uint32_t i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  uint32_t addr = ((threadIdx.x * stride) & 255u) * 16, acc = 0;
  for (int it = 0; it < N_ITER; it++) {
    u4 r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                 "ds_read_b128 %4, %8 offset:128\n ds_read_b128 %5, %8 offset:4224\n ds_read_b128 %6, %8 offset:8320\n ds_read_b128 %7, %8 offset:12416\n"
                 "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
    u4 x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    acc ^= x.x ^ x.y ^ x.z ^ x.w;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_bpermute(uint32_t* out, uint32_t stride) {
  uint32_t addr = ((threadIdx.x * stride) & 63u) * 4;
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int it = 0; it < N_ITER; it++) {
    asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                 "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n"
                 "s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
#define DEFX(NAME, LINE)                                                                        \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t stride) {                \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = stride ^ threadIdx.x;                                                         \
    for (int it = 0; it < N_ITER; it++) { LINE(a0, a1) LINE(a2, a3) LINE(a4, a5) LINE(a6, a7) LINE(a1, a2) LINE(a3, a4) LINE(a5, a6) LINE(a7, a0) } \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
  }
#define L_DPP_QUAD(x, y) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
#define L_DPP_ROR(x, y) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
#define L_CND_DPP(x, y) asm volatile("s_nop 1\n v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
#define L_SWAP32(x, y) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
#define L_SWAP16(x, y) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
DEFX(k_dpp_quad, L_DPP_QUAD)
DEFX(k_dpp_ror, L_DPP_ROR)
DEFX(k_cnd_dpp, L_CND_DPP)
DEFX(k_swap32, L_SWAP32)
DEFX(k_swap16, L_SWAP16)

typedef void (*kern_t)(uint32_t*, uint32_t);
struct K { const char* name; kern_t fn; uint32_t stride; double ops_per_iter; };
int main() {
  K ks[] = {{"ds_read_b32 stride 1", k_read32, 1, 8}, {"ds_read_b32 stride 2 (2-way)", k_read32, 2, 8}, {"ds_read_b32 stride 4 (4-way)", k_read32, 4, 8},
            {"ds_write_b32 stride 1", k_write32, 1, 8}, {"ds_write_b32 stride 4 (4-way)", k_write32, 4, 8},
            {"ds_read_b64 stride 1", k_read64, 1, 8}, {"ds_read_b128 stride 1", k_read128, 1, 8}, {"ds_bpermute_b32", k_bpermute, 5, 8},
            {"v_mov_b32_dpp quad_perm (+s_nop 1)", k_dpp_quad, 1, 8}, {"v_mov_b32_dpp row_ror:8 (+s_nop 1)", k_dpp_ror, 1, 8},
            {"v_cndmask_b32_dpp (+s_nop 1)", k_cnd_dpp, 1, 8}, {"v_permlane32_swap_b32", k_swap32, 1, 8}, {"v_permlane16_swap_b32", k_swap16, 1, 8}};
  const int blocks = 256 * 8;
  uint32_t* out = nullptr;
  CK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, k.stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, k.stride);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    double wave_ops_per_cu = 4.0 * (blocks / 256.0) * 4.0 * N_ITER * k.ops_per_iter;  // launches x blocks/CU x waves/block
    printf("%-38s %6.2f cycles / wave-op / CU   (%5.2f per SIMD) at 1.9 GHz\n", k.name, ms * 1e-3 * 1.9e9 / wave_ops_per_cu,
           ms * 1e-3 * 1.9e9 / (wave_ops_per_cu / 4));
  }
  return 0;
}
