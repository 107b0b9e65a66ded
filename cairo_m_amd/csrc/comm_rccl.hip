// In-library collectives for the sharded prover: cm_comm implemented on RCCL (xGMI inside a node), enqueued on the PROVER'S
// stream — no host synchronisation around an exchange and no Python in the data path (north_star: "RCCL all-gather over xGMI
// only at the Merkle-root and FRI-fold boundaries"; VERDICT r02 missing #6).
//
//   rank 0:      cm_rccl_unique_id(id)                      -> 128 bytes, handed to every rank by the launcher (env, file, MPI, ...)
//   every rank:  cm_rccl_comm_create(id, rank, world, staging_words, &c);   cm_prove_sharded(input, cfg, cm_rccl_comm_view(c), &proof)
//
// The two collectives the prover asks for:
//   all_gather(words_per_rank)        = ncclAllGather(send_buf, recv_buf)                     (sub-roots, sums, samples, queries)
//   all_to_all_v(send[], recv[])      = ncclGroupStart; ncclSend / ncclRecv per peer; ncclGroupEnd   (column owner -> row owner)
// xGMI is point to point: the grouped send/recv pairs of one all-to-all use all seven links of a GPU at once, and no ring is
// involved (an all-to-all of a tree's LDE is 1.6 GB / N per rank at the metric size).
//
// librccl is loaded with dlopen on first use — the copy next to the HIP runtime in use, else one the process has mapped, else the
// system's: the library has no link-time dependency on RCCL and single-GPU users never load it.
#include <dlfcn.h>
#include <string.h>
#include <memory>
#include <mutex>
#include <string>
#include "../../include/cairom_hip.h"
#include "engine.hpp"

extern "C" int32_t cm_set_last_error(const char* msg);

namespace cm {
namespace {

// the slice of rccl.h this file needs (ABI-stable C interface of NCCL / RCCL)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;        // ncclSuccess = 0
constexpr int kNcclUint32 = 3;   // ncclDataType_t::ncclUint32 (rccl.h)
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // FIRST the librccl that sits next to the HIP runtime THIS library runs on (dladdr of a runtime entry point): a process can
    // hold two ROCm stacks — the system's and the one a PyTorch wheel bundles — and whichever libamdhip64 was mapped first serves
    // both; an RCCL built for the other one fails in ncclCommInitRank ("unhandled cuda error": seen with the wheel's RCCL 2.26 on
    // the system's HIP 7.2 after a late `import torch`).  An absolute path maps that copy even when another librccl.so.1 is loaded.
    {
      Dl_info di;
      if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
          dir.resize(slash + 1);
          for (const char* n : {"librccl.so.1", "librccl.so"})
            if ((r.h = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL))) break;
        }
      }
    }
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (!r.h) for (const char* n : names) if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;    // a copy the process already has
    if (!r.h) for (const char* n : names) if ((r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.h) return;
    auto sym = [&](const char* s) { return dlsym(r.h, s); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  if (!r.h || !r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd)
    throw CmError(4, "librccl.so could not be loaded (the sharded prover's in-library collectives need RCCL)");
  return r;
}
void nccl_ck(ncclResult_t rc, const char* what) {
  if (rc != 0) {
    Rccl& r = rccl();
    throw CmError(5, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
  }
}

}  // namespace
}  // namespace cm

struct cm_rccl_comm {
  cm::ncclComm_t comm = nullptr;
  cm_comm view{};
  cm::DevBuf send, recv;
  hipStream_t stream = nullptr;   // the prover's stream (cm_comm::set_stream), or the thread's main stream
  std::string last_error;
};

namespace {
template <class F>
int32_t cguard(cm_rccl_comm* c, F&& f) {
  try { f(); return 0; }
  catch (const std::exception& e) {
    if (c) c->last_error = e.what();
    cm_set_last_error(e.what());
    return 1;
  }
}
int32_t rccl_set_stream(void* ctx, cm_stream_t s) {
  ((cm_rccl_comm*)ctx)->stream = (hipStream_t)(uintptr_t)s;
  return 0;
}
int32_t rccl_all_gather(void* ctx, uint64_t words_per_rank) {
  cm_rccl_comm* c = (cm_rccl_comm*)ctx;
  return cguard(c, [&] {
    if (!words_per_rank) return;
    cm::nccl_ck(cm::rccl().AllGather(c->view.send_buf, c->view.recv_buf, (size_t)words_per_rank, cm::kNcclUint32, c->comm, c->stream),
                "ncclAllGather");
  });
}
int32_t rccl_all_to_all_v(void* ctx, const uint64_t* send_words, const uint64_t* recv_words) {
  cm_rccl_comm* c = (cm_rccl_comm*)ctx;
  return cguard(c, [&] {
    cm::Rccl& r = cm::rccl();
    const uint32_t n = c->view.world, me = c->view.rank;
    uint64_t so = 0, ro = 0, my_s = 0, my_r = 0;
    for (uint32_t k = 0; k < me; k++) { my_s += send_words[k]; my_r += recv_words[k]; }
    // the rank's own block never leaves the device
    if (send_words[me])
      CM_HIP(hipMemcpyAsync(c->view.recv_buf + my_r, c->view.send_buf + my_s, send_words[me] * 4, hipMemcpyDeviceToDevice, c->stream));
    cm::nccl_ck(r.GroupStart(), "ncclGroupStart");
    // The group is closed on EVERY path: a Send / Recv that fails in the middle must not leave this thread inside an open NCCL
    // group (every later call on it — the ncclCommAbort of cm_comm::abort included — would be deferred).  The first error is
    // remembered and reported once the group is closed.
    cm::ncclResult_t first_err = (cm::ncclResult_t)0;
    const char* first_what = nullptr;
    for (uint32_t k = 0; k < n && !first_err; k++) {
      if (k != me && send_words[k]) {
        const cm::ncclResult_t e = r.Send(c->view.send_buf + so, (size_t)send_words[k], cm::kNcclUint32, (int)k, c->comm, c->stream);
        if (e) { first_err = e; first_what = "ncclSend"; break; }
      }
      if (k != me && recv_words[k]) {
        const cm::ncclResult_t e = r.Recv(c->view.recv_buf + ro, (size_t)recv_words[k], cm::kNcclUint32, (int)k, c->comm, c->stream);
        if (e) { first_err = e; first_what = "ncclRecv"; break; }
      }
      so += send_words[k];
      ro += recv_words[k];
    }
    const cm::ncclResult_t end_err = r.GroupEnd();
    if (first_err) cm::nccl_ck(first_err, first_what);
    cm::nccl_ck(end_err, "ncclGroupEnd");
  });
}
// cm_comm::abort: this rank failed in the middle of a sharded proof — tear the communicator down so that the peers' pending and
// next collectives return an error instead of waiting for it (the handle is dead afterwards; cm_rccl_comm_destroy skips it)
void rccl_abort(void* ctx) {
  cm_rccl_comm* c = (cm_rccl_comm*)ctx;
  if (c && c->comm && cm::rccl().CommAbort) { (void)cm::rccl().CommAbort(c->comm); c->comm = nullptr; }
}
}  // namespace

extern "C" {
int32_t cm_rccl_unique_id(uint8_t id_out[128]) {
  return cguard(nullptr, [&] {
    cm::ncclUniqueId id;
    cm::nccl_ck(cm::rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, id.internal, 128);
  });
}
int32_t cm_rccl_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, uint64_t staging_words, cm_rccl_comm** out) {
  return cguard(nullptr, [&] {
    CM_CHECK(world >= 1 && world <= 8 && (world & (world - 1)) == 0 && rank < world, "cm_rccl_comm_create: world must be 1, 2, 4 or 8");
    cm::bind_thread_to_library_device();
    std::unique_ptr<cm_rccl_comm> c(new cm_rccl_comm());
    cm::ncclUniqueId uid;
    memcpy(uid.internal, id, 128);
    cm::nccl_ck(cm::rccl().CommInitRank(&c->comm, (int)world, uid, (int)rank), "ncclCommInitRank");
    c->send.alloc(staging_words * 4);
    c->recv.alloc(staging_words * 4);
    c->stream = cm::thread_main_stream();
    c->view.struct_size = (uint32_t)sizeof(cm_comm);
    c->view.rank = rank; c->view.world = world; c->view.ctx = c.get();
    c->view.send_buf = c->send.u32(); c->view.recv_buf = c->recv.u32(); c->view.buf_words = staging_words;
    c->view.all_to_all_v = rccl_all_to_all_v;
    c->view.all_gather = rccl_all_gather;
    c->view.flags = CM_COMM_STREAM_ORDERED;
    c->view.set_stream = rccl_set_stream;
    c->view.abort = rccl_abort;
    *out = c.release();
  });
}
const cm_comm* cm_rccl_comm_view(const cm_rccl_comm* c) { return c ? &c->view : nullptr; }
int32_t cm_rccl_comm_destroy(cm_rccl_comm* c) {
  if (!c) return 0;
  int32_t rc = cguard(c, [&] {
    if (c->stream) CM_HIP(hipStreamSynchronize(c->stream));
    if (c->comm && cm::rccl().CommDestroy) cm::nccl_ck(cm::rccl().CommDestroy(c->comm), "ncclCommDestroy");
  });
  delete c;
  return rc;
}
}  // extern "C"
