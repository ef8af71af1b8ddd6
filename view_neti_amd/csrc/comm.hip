// The data-parallel exchange of the train step behind the C ABI: ONE all-reduce(sum) of the flat f32 mapper-gradient
// bucket per optimisation step, issued on the caller's stream straight into RCCL (ring / tree over xGMI).
//
// Replaces what the reference gets from accelerate: `accelerator.prepare(text_encoder, ...)` wraps the text encoder in
// DDP and `accelerator.backward(loss)` all-reduces its parameter gradients (training/coach.py:97-99, 211-218).  Here the
// trainable state is one flat bucket, so the whole exchange is one collective of n floats (434 KB at
// D = 768; in learnable_mode 3 the active scene's segment + the view mapper, packed contiguously by the caller).
//
// librccl is NOT a link-time dependency: it is resolved with dlopen on first use (a process that already holds a copy —
// PyTorch ships its own librccl.so.1 — reuses that image), so single-GPU users never touch it and a missing library is an
// error message, not a load failure of libvneti_hip.so.  No fallback: without RCCL the entry points fail.
#include <dlfcn.h>

#include <mutex>

#include "common.h"
#include "../../include/vneti.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x: opaque 128-byte id, enum values below)
struct VnUniqueId { char internal[128]; };
typedef void* VnComm;
typedef int (*fn_get_uid)(VnUniqueId*);
typedef int (*fn_init_rank)(VnComm*, int, VnUniqueId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, VnComm, hipStream_t);
typedef int (*fn_destroy)(VnComm);
typedef const char* (*fn_errstr)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
  void* handle = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
char g_load_err[256] = "symbols missing";  // why load_rccl() failed: dlerror() read ONCE, right after the failing call

void keep_dlerror(const char* what) {
  const char* e = dlerror();  // clears the state: a second call would return NULL
  snprintf(g_load_err, sizeof g_load_err, "%s: %s", what, e ? e : "no dlerror text");
}

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {  // an image the process already holds first (PyTorch's), then the system's
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  for (int i = 0; !g_rccl.handle && i < 3; ++i) {
    g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.handle) keep_dlerror(names[i]);
  }
  if (!g_rccl.handle) return;
  g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.destroy = (fn_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.errstr = (fn_errstr)dlsym(g_rccl.handle, "ncclGetErrorString");
  if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.allreduce || !g_rccl.destroy)
    snprintf(g_load_err, sizeof g_load_err, "ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy not all exported");
}

int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.handle || !g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.allreduce || !g_rccl.destroy) {
    vneti_set_error("RCCL (librccl.so.1) could not be loaded: %s", g_load_err);
    return VNETI_EUNSUP;
  }
  return VNETI_OK;
}

int rccl_fail(const char* what, int rc) {
  vneti_set_error("%s failed: %s (ncclResult %d)", what, g_rccl.errstr ? g_rccl.errstr(rc) : "?", rc);
  return VNETI_EHIP;
}

}  // namespace

extern "C" int vneti_comm_unique_id(void* id128) {
  VN_REQUIRE(id128 != nullptr, "comm_unique_id: null buffer");
  if (int rc = need_rccl()) return rc;
  VnUniqueId id;
  if (int rc = g_rccl.get_uid(&id)) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return VNETI_OK;
}

extern "C" int vneti_comm_init(const void* id128, int rank, int world, void** comm) {
  VN_REQUIRE(id128 && comm, "comm_init: null pointer");
  VN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
  if (int rc = need_rccl()) return rc;
  VnUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  VnComm c = nullptr;
  if (int rc = g_rccl.init_rank(&c, world, id, rank)) return rccl_fail("ncclCommInitRank", rc);
  *comm = c;
  return VNETI_OK;
}

extern "C" int vneti_allreduce_flat(void* comm, float* buf, long long n, void* stream) {
  VN_REQUIRE(comm && buf && n > 0, "allreduce_flat: null communicator / buffer or n = %lld", n);
  if (int rc = need_rccl()) return rc;
  if (int rc = g_rccl.allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, (VnComm)comm, (hipStream_t)stream))
    return rccl_fail("ncclAllReduce", rc);
  return VNETI_OK;
}

extern "C" int vneti_comm_destroy(void* comm) {
  if (!comm) return VNETI_OK;
  if (int rc = need_rccl()) return rc;
  if (int rc = g_rccl.destroy((VnComm)comm)) return rccl_fail("ncclCommDestroy", rc);
  return VNETI_OK;
}
