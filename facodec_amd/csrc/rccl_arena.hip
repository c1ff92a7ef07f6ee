// fac_allreduce_arena: the gradient exchange of the training step as an explicit RCCL collective over a flat fp32 arena
// (the bucketed all-reduce(mean) DistributedDataParallel runs behind train.py:110-111,289,361; SURVEY 8b
// lists it in the minimum ABI set).  The default exchange goes through torch.distributed (backend "nccl" IS RCCL on ROCm); this
// entry point is the same collective without torch in the signature -- plain device pointer, element count, communicator handle,
// explicit stream -- for callers that own their streams (overlap control: the collective runs on whatever stream they name) or
// have no torch.distributed process group at all.
//
// RCCL is bound at RUN time: librccl.so.1 is taken from the process if it is already loaded (PyTorch-ROCm ships its own copy and
// two RCCL instances in one process must not be mixed), else dlopen'ed.  libfacodec_hip.so itself has no link-time dependency on
// RCCL: single-GPU users never load it.
#include <dlfcn.h>
#include <string.h>
#include <mutex>

#include "common.h"

namespace fac {

// the few declarations of rccl.h this file needs (stable NCCL ABI: ncclResult_t / ncclDataType_t / ncclRedOp_t are ints)
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
constexpr int kNcclFloat = 7, kNcclSum = 0, kNcclAvg = 4;

struct RcclApi {
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

static RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy the process already uses (torch's), if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
  });
  return &api;
}

static int rccl_fail(const char* what, int rc) {
  RcclApi* r = rccl();
  set_error("%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
  return FAC_ERR_LAUNCH;
}

}  // namespace fac

extern "C" int fac_rccl_available(void) { return fac::rccl()->ok ? 1 : 0; }

extern "C" int fac_rccl_unique_id(void* id128) {
  using namespace fac;
  FAC_REQUIRE(id128 != nullptr, "rccl_unique_id: null pointer");
  FAC_REQUIRE(rccl()->ok, "rccl_unique_id: librccl.so.1 could not be loaded");
  RcclUniqueId id;
  const int rc = rccl()->GetUniqueId(&id);
  if (rc != 0) return rccl_fail("rccl_unique_id", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return FAC_OK;
}

extern "C" int fac_rccl_comm_init(void** comm, const void* id128, int nranks, int rank) {
  using namespace fac;
  FAC_REQUIRE(comm && id128 && nranks > 0 && rank >= 0 && rank < nranks, "rccl_comm_init: bad arguments");
  FAC_REQUIRE(rccl()->ok, "rccl_comm_init: librccl.so.1 could not be loaded");
  RcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  RcclComm c = nullptr;
  const int rc = rccl()->CommInitRank(&c, nranks, id, rank);      // collective over the ranks; binds the CURRENT device
  if (rc != 0) return rccl_fail("rccl_comm_init", rc);
  *comm = c;
  return FAC_OK;
}

extern "C" int fac_rccl_comm_destroy(void* comm) {
  using namespace fac;
  if (comm == nullptr) return FAC_OK;
  FAC_REQUIRE(rccl()->ok, "rccl_comm_destroy: librccl.so.1 could not be loaded");
  const int rc = rccl()->CommDestroy(comm);
  return rc == 0 ? FAC_OK : rccl_fail("rccl_comm_destroy", rc);
}

extern "C" int fac_allreduce_arena(void* comm, float* arena, int64_t count, int average, fac_stream_t stream) {
  using namespace fac;
  FAC_REQUIRE(comm && arena && count > 0, "allreduce_arena: bad arguments");
  FAC_REQUIRE(rccl()->ok, "allreduce_arena: librccl.so.1 could not be loaded");
  const int rc = rccl()->AllReduce(arena, arena, (size_t)count, kNcclFloat, average ? kNcclAvg : kNcclSum, comm, (hipStream_t)stream);
  return rc == 0 ? FAC_OK : rccl_fail("allreduce_arena", rc);
}
