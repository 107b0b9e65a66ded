// Cost of a fork/join over k side streams on gfx950 (cross-stream event waits), and of a device->host->device round trip.
// Build: hipcc -O2 --offload-arch=gfx950 tools/forkjoin_lab.hip -o gpurun_out/forkjoin_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(uint32_t* p, uint32_t iters) {
  uint32_t x = threadIdx.x;
  for (uint32_t i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
  if (x == 0xdeadbeef) p[0] = x;
}
int main() {
  const int N = 8;
  hipStream_t main_s, s[N];
  hipEvent_t fork_ev, done[N], t0, t1;
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  for (int i = 0; i < N; i++) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
  CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  uint32_t* d; CK(hipMalloc(&d, 4096));
  uint32_t* h; CK(hipHostMalloc((void**)&h, 4096, hipHostMallocDefault));
  const uint32_t W = 1000;   // ~ tens of us per kernel
  for (int k = 0; k <= N; k++) {
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
      // a long kernel first so the host is well ahead of the GPU when the fork/join packets are processed
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, 2000000u);
      CK(hipEventRecord(t0, main_s));
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, W);
      if (k > 0) {
        CK(hipEventRecord(fork_ev, main_s));
        for (int i = 0; i < k; i++) { CK(hipStreamWaitEvent(s[i], fork_ev, 0)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], d, W); }
        for (int i = 0; i < k; i++) { CK(hipEventRecord(done[i], s[i])); CK(hipStreamWaitEvent(main_s, done[i], 0)); }
      }
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, W);
      CK(hipEventRecord(t1, main_s));
      CK(hipStreamSynchronize(main_s));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (r >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    printf("fork/join over %d streams: avg %.1f us  best %.1f us  (k=0: two kernels back to back)\n", k, sum / reps * 1000, best * 1000);
  }
  // the same join when the side streams finished long before the main stream reaches it (main carries a kernel 10x longer)
  for (int k = 0; k <= N; k++) {
    float sum = 0; const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, 2000000u);
      CK(hipEventRecord(t0, main_s));
      if (k > 0) {
        CK(hipEventRecord(fork_ev, main_s));
        for (int i = 0; i < k; i++) { CK(hipStreamWaitEvent(s[i], fork_ev, 0)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], d, W); }
      }
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, 10 * W);   // the big kernel stays on the main stream
      for (int i = 0; i < k; i++) { CK(hipEventRecord(done[i], s[i])); CK(hipStreamWaitEvent(main_s, done[i], 0)); }
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, W);
      CK(hipEventRecord(t1, main_s));
      CK(hipStreamSynchronize(main_s));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (r >= 3) sum += ms;
    }
    printf("big kernel on main, %d early side streams joined after it: avg %.1f us\n", k, sum / reps * 1000);
  }
  // round trips: kernel -> D2H 32 B -> host -> H2D 64 B -> kernel
  for (int variant = 0; variant < 3; variant++) {
    float sum = 0; const int reps = 20;
    for (int r = 0; r < reps + 3; r++) {
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, 2000000u);
      CK(hipEventRecord(t0, main_s));
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, W);
      if (variant >= 1) { CK(hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, main_s)); CK(hipStreamSynchronize(main_s)); }
      if (variant >= 2) { h[16] = h[0] + 1; CK(hipMemcpyAsync(d + 64, h + 16, 64, hipMemcpyHostToDevice, main_s)); }
      hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, main_s, d, W);
      CK(hipEventRecord(t1, main_s));
      CK(hipStreamSynchronize(main_s));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (r >= 3) sum += ms;
    }
    printf("round trip variant %d (%s): avg %.1f us\n", variant, variant == 0 ? "no host" : variant == 1 ? "D2H + sync + launch" : "D2H + sync + H2D + launch", sum / reps * 1000);
  }
  return 0;
}
