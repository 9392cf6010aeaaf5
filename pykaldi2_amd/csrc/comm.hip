// Gradient exchange behind the C ABI: RCCL all-reduce over xGMI (gfx950 library).
//
// Replaces the Horovod / NCCL half of the reference's training loop (hvd.DistributedOptimizer at reference
// bin/train_chain.py:141-145, bin/train_ce.py:127-131, bin/train_se.py:130-134): one process per GPU, one communicator
// per process, in-place sum all-reduce of a slice ("bucket") of the flat gradient buffer on the stream the caller
// names -- the compute stream, or a side stream the caller fences with events.  Only the 128-byte unique id travels
// through the launcher's own channel (torch.distributed in pykaldi2_amd/hvd.py, MPI in the reference).
//
// librccl is bound at RUN time with dlopen/dlsym (no link-time dependency: the CPU-only test suite and single-GPU
// users never touch it).  A copy already mapped into the process (the one bundled with PyTorch) is preferred, so a
// process never holds two RCCL instances.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "common.h"

struct pk2_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

namespace pk2 {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char path[256] = {0};
};
static RcclApi g_rccl;

static int rccl_load() {
  if (g_rccl.handle) return PK2_OK;
  const char* env = getenv("PK2_RCCL_LIB");
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  if (env && *env) {
    h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    if (h) snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", env);
  }
  for (const char* n : names) {     // already in the process (PyTorch's bundled copy)?
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) snprintf(g_rccl.path, sizeof(g_rccl.path), "%s (already loaded)", n);
  }
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", n);
  }
  if (!h) {
    h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) snprintf(g_rccl.path, sizeof(g_rccl.path), "/opt/rocm/lib/librccl.so");
  }
  if (!h) { set_error("comm: librccl not found (%s); set PK2_RCCL_LIB", dlerror()); return PK2_ERR_IO; }
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(dlsym(h, "ncclGroupStart"));
  g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    set_error("comm: %s lacks the nccl entry points", g_rccl.path);
    return PK2_ERR_IO;
  }
  g_rccl.handle = h;
  return PK2_OK;
}

static int rccl_fail(const char* what, ncclResult_t r) {
  set_error("comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return PK2_ERR_HIP;
}

}  // namespace pk2

using namespace pk2;

extern "C" int32_t pk2_comm_unique_id_bytes(void) { return (int32_t)NCCL_UNIQUE_ID_BYTES; }

extern "C" int pk2_comm_unique_id(void* id_out) {
  PK2_REQUIRE(id_out, "comm: null pointer");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
  return PK2_OK;
}

extern "C" int pk2_comm_init(int32_t rank, int32_t world, const void* unique_id, pk2_comm** out) {
  PK2_REQUIRE(unique_id && out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad args");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
  auto* c = new pk2_comm();
  c->rank = rank; c->world = world;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);   // collective over the ranks; uses the current device
  if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
  *out = c;
  return PK2_OK;
}

extern "C" int pk2_allreduce_bucket(pk2_comm* comm, float* buf, int64_t count, void* stream) {
  PK2_REQUIRE(comm && comm->comm && buf && count >= 0, "allreduce_bucket: bad args");
  if (count == 0) return PK2_OK;
  ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, comm->comm, static_cast<hipStream_t>(stream));
  if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
  return PK2_OK;
}

extern "C" int pk2_allreduce_guarded(pk2_comm* comm, float* buf, int64_t count, float* guard_slot, void* stream) {
  PK2_REQUIRE(comm && comm->comm && guard_slot && count >= 0 && (buf || count == 0), "allreduce_guarded: bad args");      // (count = 0: the slot travels alone)
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool group = g_rccl.GroupStart && g_rccl.GroupEnd && count > 0;
  ncclResult_t r = ncclSuccess;
  if (group && (r = g_rccl.GroupStart()) != ncclSuccess) return rccl_fail("ncclGroupStart", r);
  if (count > 0) r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, comm->comm, st);
  ncclResult_t r2 = g_rccl.AllReduce(guard_slot, guard_slot, 1, ncclFloat32, ncclMax, comm->comm, st);
  ncclResult_t r3 = group ? g_rccl.GroupEnd() : ncclSuccess;
  if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
  if (r2 != ncclSuccess) return rccl_fail("ncclAllReduce (guard slot)", r2);
  if (r3 != ncclSuccess) return rccl_fail("ncclGroupEnd", r3);
  return PK2_OK;
}

extern "C" int pk2_comm_info(const pk2_comm* comm, int32_t* rank, int32_t* world, char* lib_path, int32_t lib_path_bytes) {
  PK2_REQUIRE(comm, "comm_info: null handle");
  if (rank) *rank = comm->rank;
  if (world) *world = comm->world;
  if (lib_path && lib_path_bytes > 0) snprintf(lib_path, (size_t)lib_path_bytes, "%s", g_rccl.path);
  return PK2_OK;
}

extern "C" int pk2_comm_destroy(pk2_comm* comm) {
  if (!comm) return PK2_OK;
  if (comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->comm);
  delete comm;
  return PK2_OK;
}
