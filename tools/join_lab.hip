// Cost of a fork/join over HIP streams on gfx950, measured on the device clock (development tool):
//   hipcc --offload-arch=gfx950 -O2 tools/join_lab.hip -o /tmp/join_lab && /tmp/join_lab
// Region: main + S side streams each run one spin kernel; join = record on every side stream, main waits; then a marker
// kernel on main.  Reports (start of the marker kernel) - (end of the last region kernel) in microseconds.
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
int main(int argc, char** argv) {
  const int reps = 30;
  unsigned long long* t;   // [0..7] ends, [8] start of marker
  CK(hipHostMalloc((void**)&t, 16 * 8));
  hipStream_t main_s, side[7];
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const double us_per_tick = 1e3 / rate_khz;
  auto us2t = [&](double us) { return (unsigned long long)(us / us_per_tick); };
  struct Cfg { const char* name; int S; double main_us, side_us; unsigned flags; bool upload_after; int chain = 0; int slow = 0; };   // chain: sides wait on each other, main on the last one; slow = index of the slow side stream
  const unsigned DT = hipEventDisableTiming;
  std::vector<Cfg> cfgs = {
      {"no fork: two kernels back to back on main", 0, 200, 0, DT, false},
      {"1 side, SIDE finishes last", 1, 150, 250, DT, false},
      {"1 side, MAIN finishes last", 1, 250, 150, DT, false},
      {"3 sides, side last", 3, 150, 250, DT, false},
      {"3 sides, main last", 3, 250, 150, DT, false},
      {"7 sides, side last", 7, 150, 250, DT, false},
      {"7 sides, main last", 7, 250, 150, DT, false},
      {"3 sides, side last, release-to-device events", 3, 150, 250, DT | hipEventReleaseToDevice, false},
      {"3 sides, main last, release-to-device events", 3, 250, 150, DT | hipEventReleaseToDevice, false},
      {"3 sides, main last, default-flag events", 3, 250, 150, 0, false},
      {"3 sides, main last, then a pinned H2D copy before the marker", 3, 250, 150, DT, true},
      {"no fork, then a pinned H2D copy before the marker", 0, 200, 0, DT, true},
      {"7 sides, NO kernel on main, slow side waited FIRST", 7, 0, 250, DT, false, 0, 0},
      {"7 sides, NO kernel on main, slow side waited LAST", 7, 0, 250, DT, false, 0, 6},
      {"7 sides, NO kernel on main, two slow sides (waited last)", 7, 0, 250, DT, false, 3, 6},
      {"3 sides, NO kernel on main, slow side waited LAST", 3, 0, 250, DT, false, 0, 2},
      {"7 sides CHAINED, main last", 7, 250, 150, DT, false, 1, 0},
      {"7 sides CHAINED, first side of the chain last", 7, 150, 250, DT, false, 1, 0},
      {"7 sides CHAINED, middle side of the chain last", 7, 150, 250, DT, false, 1, 3},
      {"7 sides CHAINED, end of the chain last", 7, 150, 250, DT, false, 1, 6},
      {"3 sides CHAINED, main last", 3, 250, 150, DT, false, 1, 0},
      {"3 sides CHAINED, end of the chain last", 3, 150, 250, DT, false, 1, 2},
      {"7 sides in two chains (4 + 3), main last", 7, 250, 150, DT, false, 2, 0},
      {"7 sides in two chains (4 + 3), a chain end last", 7, 150, 250, DT, false, 2, 3},
  };
  uint32_t* pinned; uint32_t* dev;
  CK(hipHostMalloc((void**)&pinned, 4096));
  CK(hipMalloc((void**)&dev, 4096));
  for (auto& c : cfgs) {
    hipEvent_t fork_ev, done[7];
    CK(hipEventCreateWithFlags(&fork_ev, c.flags));
    for (auto& e : done) CK(hipEventCreateWithFlags(&e, c.flags));
    std::vector<double> gaps;
    for (int r = 0; r < reps + 3; r++) {
      CK(hipEventRecord(fork_ev, main_s));
      for (int i = 0; i < c.S; i++) {
        CK(hipStreamWaitEvent(side[i], fork_ev, 0));
        const bool slow = i == c.slow || (c.chain == 3 && i == c.slow - 1);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, side[i], us2t(slow ? c.side_us : c.side_us * 0.6), t + 1 + i);
      }
      if (c.main_us > 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, main_s, us2t(c.main_us), t + 0);
      else t[0] = 0;
      if (c.chain == 0 || c.chain == 3) {
        for (int i = 0; i < c.S; i++) { CK(hipEventRecord(done[i], side[i])); CK(hipStreamWaitEvent(main_s, done[i], 0)); }
      } else {
        const int split = c.chain == 2 ? 4 : c.S;   // chain ends: split - 1 and S - 1
        for (int i = 0; i < c.S; i++) {
          if (i != 0 && i != split) CK(hipStreamWaitEvent(side[i], done[i - 1], 0));
          CK(hipEventRecord(done[i], side[i]));
          if (i == split - 1 || i == c.S - 1) CK(hipStreamWaitEvent(main_s, done[i], 0));
        }
      }
      if (c.upload_after) CK(hipMemcpyAsync(dev, pinned, 512, hipMemcpyHostToDevice, main_s));
      hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, main_s, t + 8);
      CK(hipStreamSynchronize(main_s));
      unsigned long long last = t[0];
      for (int i = 0; i < c.S; i++) last = std::max(last, t[1 + i]);
      if (r >= 3) gaps.push_back(((double)t[8] - (double)last) * us_per_tick);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%-62s gap median %6.1f us  min %6.1f  max %6.1f\n", c.name, gaps[gaps.size() / 2], gaps.front(), gaps.back());
    CK(hipEventDestroy(fork_ev));
    for (auto& e : done) CK(hipEventDestroy(e));
  }
  return 0;
}
