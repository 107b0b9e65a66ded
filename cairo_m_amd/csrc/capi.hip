// extern "C" boundary of libcairom_hip.so (declared in include/cairom_hip.h).
// Thin wrappers: translate opaque handles into device pointers, upload pointer arrays, call the
// engine, map C++ exceptions to status codes + a thread-local error string.
#include "../../include/cairom_hip.h"
#include "engine.hpp"
#include "merkle_tree.hpp"
#include "framing.hpp"
#include <string.h>
#include <stdio.h>
#include <string>
#include <atomic>
#include <mutex>

using namespace cm;

// ---- process-wide framing switches (framing.hpp) -----------------------------------------------------------------------
namespace cm {
namespace {
std::mutex g_framing_mu;
Framing g_framing;
std::atomic<bool> g_framing_init{false};
int g_framing_users = 0;   // provers / verifiers alive (guarded by g_framing_mu)
}  // namespace
const Framing& framing() {
  if (!g_framing_init) {
    std::lock_guard<std::mutex> lk(g_framing_mu);
    if (!g_framing_init) {
      Framing f;
      if (const char* e = getenv("CM_FRAMING")) {
        const std::string err = Framing::parse(e, f);
        // a proof made under a framing the caller did not ask for is a wrong proof: refuse to run (every entry point reports it)
        if (!err.empty()) throw CmError(1, "CM_FRAMING: " + err);
      }
      g_framing = f;
      g_framing_init = true;
    }
  }
  return g_framing;
}
std::string set_framing(const char* spec) {
  Framing f;
  const std::string err = Framing::parse(spec, f);
  if (!err.empty()) return err;
  try { (void)framing(); } catch (const CmError&) { /* a malformed CM_FRAMING is replaced by this explicit setting */ }
  std::lock_guard<std::mutex> lk(g_framing_mu);
  if (g_framing_users > 0) return "framing: a proof or a verification is in flight; the setting can only change between them";
  g_framing = f;
  g_framing_init = true;
  return "";
}
FramingUse::FramingUse() {
  (void)framing();
  std::lock_guard<std::mutex> lk(g_framing_mu);
  g_framing_users++;
}
FramingUse::~FramingUse() {
  std::lock_guard<std::mutex> lk(g_framing_mu);
  g_framing_users--;
}
}  // namespace cm

namespace {
thread_local std::string g_last_error;
std::mutex g_init_mu;
bool g_inited = false;

inline hipStream_t S(cm_stream_t s) { return (hipStream_t)(uintptr_t)s; }
inline uint32_t* P32(cm_handle h) { return (uint32_t*)(uintptr_t)h; }

template <class F>
int32_t guard(F&& f) {
  try {
    bind_thread_to_library_device();  // hipSetDevice is per host thread
    f();
    return 0;
  } catch (const CmError& e) {
    g_last_error = e.what();
    return e.code ? e.code : 1;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  } catch (...) {
    g_last_error = "unknown error";
    return 1;
  }
}
std::vector<uint32_t*> ptrs(const cm_handle* hs, uint32_t n) {
  std::vector<uint32_t*> v(n);
  for (uint32_t i = 0; i < n; i++) v[i] = P32(hs[i]);
  return v;
}
}  // namespace

extern "C" {

int32_t cm_set_last_error(const char* msg) {
  g_last_error = msg;
  return 1;
}

int32_t cm_last_error(char* buf, size_t buf_len) {
  if (!buf || !buf_len) return (int32_t)g_last_error.size();
  size_t n = g_last_error.size() < buf_len - 1 ? g_last_error.size() : buf_len - 1;
  memcpy(buf, g_last_error.data(), n);
  buf[n] = 0;
  return (int32_t)n;
}

int32_t cm_init(int32_t device) {
  return guard([&] {
    std::lock_guard<std::mutex> lk(g_init_mu);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
      throw CmError(3, "cm_init: no HIP device available (libcairom_hip has no CPU fallback)");
    CM_CHECK(device >= 0 && device < ndev, "cm_init: device index out of range");
    CM_HIP(hipSetDevice(device));
    set_library_device(device);
    g_inited = true;
  });
}
int32_t cm_shutdown(void) {
  return guard([&] {
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (g_inited) {
      CM_HIP(hipDeviceSynchronize());
      pool_trim();
    }
    g_inited = false;
  });
}
int32_t cm_pool_trim(void) {
  return guard([&] {
    bind_thread_to_library_device();
    CM_HIP(hipStreamSynchronize(thread_main_stream()));
    pool_trim();
  });
}
int32_t cm_device_mem_info(uint64_t* free_bytes, uint64_t* total_bytes) {
  return guard([&] {
    CM_CHECK(free_bytes && total_bytes, "cm_device_mem_info: null output");
    bind_thread_to_library_device();
    size_t f = 0, t = 0;
    CM_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
  });
}
int32_t cm_stream_create(cm_stream_t* out) {
  return guard([&] {
    hipStream_t st;
    CM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out = (cm_stream_t)(uintptr_t)st;
  });
}
int32_t cm_stream_destroy(cm_stream_t s) {
  return guard([&] {
    if (!s) return;
    stage_forget_stream(S(s));        // its copies out of this thread's upload ring finish first (pool.hip)
    CM_HIP(hipStreamDestroy(S(s)));
  });
}
int32_t cm_set_cpu_affinity(int32_t mode) {
  return guard([&] {
    CM_CHECK(mode >= 0 && mode <= 2, "cm_set_cpu_affinity: mode must be 0 (never), 1 (scoped to the proving calls) or 2 (sticky)");
    set_cpu_affinity_mode(mode);
  });
}
int32_t cm_get_cpu_affinity(void) { return cpu_affinity_mode(); }
int32_t cm_set_framing(const char* spec) {
  return guard([&] {
    const std::string err = set_framing(spec);
    if (!err.empty()) throw CmError(1, err);
  });
}
int32_t cm_get_framing(char* buf, size_t buf_len) {
  std::string d;
  try { d = framing().describe(); }
  catch (const std::exception& e) { g_last_error = e.what(); return -1; }   // a malformed CM_FRAMING: nothing is in force
  if (!buf || !buf_len) return (int32_t)d.size();
  const size_t n = d.size() < buf_len - 1 ? d.size() : buf_len - 1;
  memcpy(buf, d.data(), n);
  buf[n] = 0;
  return (int32_t)n;
}
int32_t cm_stream_sync(cm_stream_t s) {
  return guard([&] { CM_HIP(hipStreamSynchronize(S(s))); });
}

int32_t cm_col_alloc(uint64_t n_u32, cm_handle* out) {
  return guard([&] {
    void* p = nullptr;
    CM_HIP(hipMalloc(&p, (n_u32 ? n_u32 : 1) * 4));
    *out = (cm_handle)(uintptr_t)p;
  });
}
int32_t cm_col_free(cm_handle h) {
  return guard([&] { if (h) CM_HIP(hipFree(P32(h))); });
}
int32_t cm_col_h2d(cm_handle h, const uint32_t* src, uint64_t n, cm_stream_t s) {
  return guard([&] {
    CM_HIP(hipMemcpyAsync(P32(h), src, n * 4, hipMemcpyHostToDevice, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_col_d2h(cm_handle h, uint32_t* dst, uint64_t n, cm_stream_t s) {
  return guard([&] {
    CM_HIP(hipMemcpyAsync(dst, P32(h), n * 4, hipMemcpyDeviceToHost, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
// Column<T>::{at, set, clone} of a Rust-side backend: element ranges and device-to-device copies
int32_t cm_col_read(cm_handle h, uint64_t offset_u32, uint32_t* dst, uint64_t n, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(h && (dst || !n), "cm_col_read: null column / destination");
    CM_HIP(hipMemcpyAsync(dst, P32(h) + offset_u32, n * 4, hipMemcpyDeviceToHost, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_col_write(cm_handle h, uint64_t offset_u32, const uint32_t* src, uint64_t n, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(h && (src || !n), "cm_col_write: null column / source");
    CM_HIP(hipMemcpyAsync(P32(h) + offset_u32, src, n * 4, hipMemcpyHostToDevice, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_col_copy(cm_handle dst, cm_handle src, uint64_t n, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(dst && src, "cm_col_copy: null column");
    CM_HIP(hipMemcpyAsync(P32(dst), P32(src), n * 4, hipMemcpyDeviceToDevice, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_bit_reverse(const cm_handle* cols, uint32_t n_cols, uint32_t log_n, cm_stream_t s) {
  return guard([&] {
    DevBuf d = upload(ptrs(cols, n_cols), S(s));
    bit_reverse_columns(d.as<uint32_t*>(), n_cols, log_n, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}

int32_t cm_twiddles_precompute(uint32_t log_size, cm_handle* tw_out) {
  return guard([&] {
    Twiddles* t = twiddles_create(log_size, 0);
    CM_HIP(hipStreamSynchronize(0));
    *tw_out = (cm_handle)(uintptr_t)t;
  });
}
int32_t cm_twiddles_free(cm_handle tw) {
  return guard([&] { twiddles_destroy((Twiddles*)(uintptr_t)tw); });
}
int32_t cm_interpolate(const cm_handle* cols, uint32_t n_cols, uint32_t log_n, cm_handle tw, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_interpolate: null twiddles");
    DevBuf d = upload(ptrs(cols, n_cols), S(s));
    interpolate(d.as<uint32_t*>(), n_cols, log_n, *(Twiddles*)(uintptr_t)tw, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_evaluate(const cm_handle* coeffs, uint32_t n_cols, uint32_t log_n, uint32_t log_out, cm_handle tw,
                    const cm_handle* out, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_evaluate: null twiddles");
    DevBuf ds = upload(ptrs(coeffs, n_cols), S(s));
    DevBuf dd = upload(ptrs(out, n_cols), S(s));
    evaluate(ds.as<const uint32_t*>(), dd.as<uint32_t*>(), n_cols, log_n, log_out, *(Twiddles*)(uintptr_t)tw, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_interpolate_extend(const cm_handle* evals, const cm_handle* coeffs, const cm_handle* lde, uint32_t n_cols, uint32_t log_n,
                              cm_handle tw, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_interpolate_extend: null twiddles");
    DevBuf de = upload(ptrs(evals, n_cols), S(s));
    DevBuf dc = upload(ptrs(coeffs, n_cols), S(s));
    DevBuf dl = upload(ptrs(lde, n_cols), S(s));
    interpolate_extend(de.as<const uint32_t*>(), dc.as<uint32_t*>(), dl.as<uint32_t*>(), n_cols, log_n, *(Twiddles*)(uintptr_t)tw, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_eval_at_point(const cm_handle* coeffs, uint32_t n_cols, uint32_t log_n, const uint32_t pt_xy[8],
                         uint32_t* out, cm_stream_t s) {
  return guard([&] {
    DevBuf ds = upload(ptrs(coeffs, n_cols), S(s));
    DevBuf scratch(eval_at_point_scratch_words(n_cols, log_n) * 4);
    DevBuf dout((size_t)n_cols * 16);
    eval_at_point_batch(ds.as<const uint32_t*>(), n_cols, log_n, QM31::from_u32(pt_xy), QM31::from_u32(pt_xy + 4),
                        scratch.u32(), dout.u32(), S(s));
    CM_HIP(hipMemcpyAsync(out, dout.p, (size_t)n_cols * 16, hipMemcpyDeviceToHost, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}

int32_t cm_merkle_commit_layer(uint32_t log_size, cm_handle prev_layer, const cm_handle* cols, uint32_t n_cols,
                               cm_handle out_hashes, cm_stream_t s) {
  return guard([&] {
    DevBuf dc = upload(ptrs(cols, n_cols), S(s));
    merkle_layer(log_size, P32(prev_layer), dc.as<const uint32_t*>(), n_cols, P32(out_hashes), S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_merkle_commit(const cm_handle* cols, const uint32_t* col_logs, uint32_t n_cols, uint8_t root[32],
                         cm_stream_t s) {
  return guard([&] {
    std::vector<const uint32_t*> c(n_cols);
    std::vector<uint32_t> logs(col_logs, col_logs + n_cols);
    for (uint32_t i = 0; i < n_cols; i++) c[i] = P32(cols[i]);
    MerkleTree t;
    t.commit(c, logs, S(s));
    t.root(root, S(s));
  });
}
int32_t cm_grind(const uint8_t digest[32], uint32_t pow_bits, uint64_t* nonce_out, cm_stream_t s) {
  return guard([&] { *nonce_out = grind_gpu(digest, pow_bits, S(s)); });
}

}  // extern "C"
