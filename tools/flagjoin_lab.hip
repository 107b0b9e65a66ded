// Alternatives to the event join of a fork region on gfx950 (development tool; companion of tools/join_lab.hip):
//   hipcc --offload-arch=gfx950 -O2 tools/flagjoin_lab.hip -o /tmp/flagjoin_lab && timeout 120 /tmp/flagjoin_lab
// Region: main + S side streams each run one spin kernel, then a marker kernel on main that must start after all of them.
//   events   : hipEventRecord on every side stream + hipStreamWaitEvent on main (what Fork::join does)
//   wv+coll  : hipStreamWriteValue32(side, &flag[i], epoch) + ONE collector kernel on main that polls the S flags
//   kf+coll  : a 1-thread kernel per side stream stores flag[i] = epoch + the collector kernel
//   kf+waitv : a 1-thread kernel per side stream does atomicAdd(counter) + hipStreamWaitValue32(main, counter >= target)
// Reports (start of the marker kernel) - (end of the last region kernel) in microseconds, median of `reps`.
// Every poll loop gives up after ~0.5 s of device clock and reports it, so a wrong assumption cannot hang the box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long* t_end) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) *t_end = wall_clock64();
}
__global__ void mark(unsigned long long* t_start) { if (threadIdx.x == 0) *t_start = wall_clock64(); }
__global__ void k_flag(uint32_t* flag, uint32_t v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_count(uint32_t* counter) { atomicAdd(counter, 1u); }
__global__ void collect(const uint32_t* flags, int n, uint32_t epoch, unsigned long long limit_ticks, uint32_t* timed_out) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
    if (wall_clock64() - t0 > limit_ticks) { *timed_out = 1; return; }
    __builtin_amdgcn_s_sleep(2);
  }
}
int main() {
  const int reps = 31;
  unsigned long long* t;
  CK(hipHostMalloc((void**)&t, 16 * 8));
  hipStream_t main_s, side[7];
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const double us_per_tick = 1e3 / rate_khz;
  auto us2t = [&](double us) { return (unsigned long long)(us / us_per_tick); };
  uint32_t *flags_dev, *flags_sig, *counter_sig, *timed_out;
  CK(hipMalloc((void**)&flags_dev, 64));
  CK(hipMemset(flags_dev, 0, 64));
  if (hipExtMallocWithFlags((void**)&flags_sig, 64, hipMallocSignalMemory) != hipSuccess) { printf("no signal memory: using device memory\n"); flags_sig = nullptr; }
  uint32_t* sig8[8] = {nullptr};   // signal memory is 8 bytes per allocation on ROCm
  for (int i = 0; i < 8; i++) if (hipExtMallocWithFlags((void**)&sig8[i], 8, hipMallocSignalMemory) != hipSuccess) sig8[i] = nullptr;
  counter_sig = sig8[7];
  CK(hipHostMalloc((void**)&timed_out, 4));
  *timed_out = 0;
  hipEvent_t fork_ev, done[7];
  CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
  for (auto& e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  uint32_t epoch = 0, target = 0;
  if (counter_sig) CK(hipMemset(counter_sig, 0, 8));
  for (int i = 0; i < 7; i++) if (sig8[i]) CK(hipMemset(sig8[i], 0, 8));
  CK(hipDeviceSynchronize());
  struct Cfg { const char* name; int mode; int S; double main_us, side_us; };
  std::vector<Cfg> cfgs;
  for (int S : {1, 3, 7})
    for (int last = 0; last < 2; last++)
      for (int mode = 0; mode < 4; mode++) {
        static const char* mn[4] = {"events  ", "wv+coll ", "kf+coll ", "kf+waitv"};
        static char names[64][96];
        static int ni = 0;
        snprintf(names[ni], 96, "%s %d sides, %s finishes last", mn[mode], S, last ? "a SIDE" : "MAIN");
        cfgs.push_back({names[ni++], mode, S, last ? 150.0 : 250.0, last ? 250.0 : 150.0});
      }
  for (auto& c : cfgs) {
    if (c.mode == 1 && !sig8[0]) { printf("%-48s skipped (no signal memory)\n", c.name); continue; }
    if (c.mode == 3 && !counter_sig) { printf("%-48s skipped (no signal memory)\n", c.name); continue; }
    std::vector<double> gaps;
    for (int r = 0; r < reps; r++) {
      epoch++;
      CK(hipEventRecord(fork_ev, main_s));
      for (int i = 0; i < c.S; i++) CK(hipStreamWaitEvent(side[i], fork_ev, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, main_s, us2t(c.main_us), t + 0);
      for (int i = 0; i < c.S; i++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, side[i], us2t(c.side_us + 3 * i), t + 1 + i);
      if (c.mode == 0) {
        for (int i = 0; i < c.S; i++) { CK(hipEventRecord(done[i], side[i])); CK(hipStreamWaitEvent(main_s, done[i], 0)); }
      } else if (c.mode == 1) {
        // one flag word per side stream in signal memory; the collector reads them through a device table of pointers? keep it
        // simple: the collector polls flags_dev, and a 0-thread-cost write-value targets flags_dev as well when allowed
        for (int i = 0; i < c.S; i++) CK(hipStreamWriteValue32(side[i], flags_dev + i, epoch, 0));
        hipLaunchKernelGGL(collect, dim3(1), dim3(64), 0, main_s, flags_dev, c.S, epoch, us2t(5e5), timed_out);
      } else if (c.mode == 2) {
        for (int i = 0; i < c.S; i++) hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, side[i], flags_dev + i, epoch);
        hipLaunchKernelGGL(collect, dim3(1), dim3(64), 0, main_s, flags_dev, c.S, epoch, us2t(5e5), timed_out);
      } else {
        for (int i = 0; i < c.S; i++) hipLaunchKernelGGL(k_count, dim3(1), dim3(1), 0, side[i], counter_sig);
        target += c.S;
        CK(hipStreamWaitValue32(main_s, counter_sig, target, hipStreamWaitValueGte, 0xFFFFFFFFu));
      }
      hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, main_s, t + 8);
      CK(hipStreamSynchronize(main_s));
      for (int i = 0; i < c.S; i++) CK(hipStreamSynchronize(side[i]));
      unsigned long long last_end = t[0];
      for (int i = 0; i < c.S; i++) last_end = std::max(last_end, t[1 + i]);
      gaps.push_back(((double)t[8] - (double)last_end) * us_per_tick);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%-48s median %7.1f us   min %7.1f   max %7.1f%s\n", c.name, gaps[reps / 2], gaps[0], gaps[reps - 1], *timed_out ? "   (a poll TIMED OUT)" : "");
    *timed_out = 0;
  }
  // ---- the FORK side: main runs kernel A, then S side streams each start one kernel that must come after A -------------------
  //   events : hipEventRecord(main) behind A + hipStreamWaitEvent on every side stream (what Fork::Fork / Fork::stream do)
  //   flags  : a one-thread kernel behind A on main stores the epoch; every side stream first runs a one-wave collector that polls it
  // Reports (start of the LAST side kernel to start) - (end of A), median.
  printf("\n-- fork: side kernels behind a kernel on main --\n");
  for (int S : {1, 3, 7})
    for (int mode = 0; mode < 2; mode++) {
      std::vector<double> gaps;
      for (int r = 0; r < reps; r++) {
        epoch++;
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, main_s, us2t(200), t + 0);
        if (mode == 0) {
          CK(hipEventRecord(fork_ev, main_s));
          for (int i = 0; i < S; i++) CK(hipStreamWaitEvent(side[i], fork_ev, 0));
        } else {
          hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, main_s, flags_dev + 8, epoch);
          for (int i = 0; i < S; i++) hipLaunchKernelGGL(collect, dim3(1), dim3(64), 0, side[i], flags_dev + 8, 1, epoch, us2t(5e5), timed_out);
        }
        for (int i = 0; i < S; i++) hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, side[i], t + 1 + i);
        CK(hipStreamSynchronize(main_s));
        for (int i = 0; i < S; i++) CK(hipStreamSynchronize(side[i]));
        unsigned long long last_start = 0;
        for (int i = 0; i < S; i++) last_start = std::max(last_start, t[1 + i]);
        gaps.push_back(((double)last_start - (double)t[0]) * us_per_tick);
      }
      std::sort(gaps.begin(), gaps.end());
      printf("%s %d sides                                  median %7.1f us   min %7.1f   max %7.1f%s\n", mode ? "flags " : "events", S, gaps[reps / 2], gaps[0],
             gaps[reps - 1], *timed_out ? "   (a poll TIMED OUT)" : "");
      *timed_out = 0;
    }
  return 0;
}
