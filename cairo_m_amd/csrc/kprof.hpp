// Optional in-library kernel timing with HIP events on the launching stream (used by bench.py for the
// `roofline` object: average launch duration of each kernel class over the timed region, plus the
// algorithmic bytes the launch moves).  Disabled by default: zero events, zero overhead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace cm {

struct KProf {
  struct Rec { const char* name; double bytes; hipEvent_t a, b; };
  struct Agg { uint64_t calls = 0; double ms = 0, bytes = 0; };
  bool on = false;
  std::mutex mu;
  std::vector<Rec> recs;
  std::map<std::string, Agg> agg;
  static KProf& get() { static KProf k; return k; }
  void flush() {  // resolve pending event pairs (caller has synchronised the stream/device)
    std::lock_guard<std::mutex> lk(mu);
    for (auto& r : recs) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
        Agg& g = agg[r.name];
        g.calls++; g.ms += ms; g.bytes += r.bytes;
      }
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
  void reset() { flush(); std::lock_guard<std::mutex> lk(mu); agg.clear(); }
};

struct KProfScope {
  bool active;
  KProf::Rec r;
  hipStream_t st;
  KProfScope(const char* name, double bytes, hipStream_t s) : active(KProf::get().on), st(s) {
    if (!active) return;
    r.name = name; r.bytes = bytes;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, st);
  }
  ~KProfScope() {
    if (!active) return;
    (void)hipEventRecord(r.b, st);
    KProf& k = KProf::get();
    std::lock_guard<std::mutex> lk(k.mu);
    k.recs.push_back(r);
  }
};

}  // namespace cm
