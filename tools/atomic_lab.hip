// Global-atomic throughput probe for MI355X (development tool): the histogram kernels (range-check / bitwise
// multiplicities) issue one atomicAdd per lookup into 2^16..2^20-entry tables.  Compares agent-scope atomics (coherent
// across the 8 XCDs: performed memory-side) with workgroup-scope atomics into a PER-XCD replica of the table (performed
// in the XCD's own L2), random and same-address patterns.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/atomic_lab.hip -o tools/atomic_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID[3:0]
template <int MODE>  // 0: agent scope, one table; 1: workgroup scope, replica per XCD
__global__ void __launch_bounds__(256) k_atomics(uint32_t* table, uint32_t mask, uint32_t per_thread, uint32_t same, uint32_t* xcc_seen) {
  uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  uint32_t* t = table;
  if (MODE == 1) {
    const uint32_t xcc = xcc_id();
    t = table + (size_t)xcc * (mask + 1);
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1u << xcc);
  }
  for (uint32_t i = 0; i < per_thread; i++) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const uint32_t idx = same ? (i & 7u) : (x & mask);
    if (MODE == 0) __hip_atomic_fetch_add(t + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(t + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ void k_sum(const uint32_t* table, uint32_t n, uint32_t reps, unsigned long long* out) {
  unsigned long long s = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n * reps; i += gridDim.x * 256) s += table[i];
  atomicAdd(out, s);
}
int main() {
  const uint32_t logs[] = {16, 18, 20};
  uint32_t *table, *seen;
  unsigned long long* total;
  CK(hipMalloc(&table, (size_t)8 * 4 << 20));
  CK(hipMalloc(&seen, 4));
  CK(hipMalloc(&total, 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const uint32_t blocks = 1024, per_thread = 64;
  for (uint32_t lg : logs)
    for (uint32_t same = 0; same < 2; same++)
      for (int mode = 0; mode < 2; mode++) {
        const uint32_t mask = (1u << lg) - 1;
        CK(hipMemset(table, 0, (size_t)8 * 4 << 20)); CK(hipMemset(seen, 0, 4)); CK(hipMemset(total, 0, 8));
        CK(hipEventRecord(a, 0));
        if (mode == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, same, seen);
        else hipLaunchKernelGGL(k_atomics<1>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, same, seen);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, 0, table, mask + 1, mode ? 8u : 1u, total);
        unsigned long long tot = 0; uint32_t sn = 0;
        CK(hipMemcpy(&tot, total, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&sn, seen, 4, hipMemcpyDeviceToHost));
        const double n = (double)blocks * 256 * per_thread;
        printf("table 2^%u %-12s %-28s %8.1f us  %7.2f G atomics/s  sum %s (xcc mask %#x)\n", lg, same ? "8 addresses" : "random", mode ? "workgroup scope, XCD replica" : "agent scope",
               ms * 1e3, n / (ms * 1e-3) / 1e9, tot == (unsigned long long)n ? "ok" : "WRONG", sn);
      }
  return 0;
}
