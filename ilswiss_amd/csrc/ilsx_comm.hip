// ilsx_comm.hip — the one exchange step of the path: a single run split over G GPUs all-reduces its flat
// gradient arena between backward and the optimiser step.  The reference has no counterpart (run_experiment.py:57-78 only
// knows independent processes); the collective is RCCL's ncclAllReduce, enqueued on the ctx's OWN stream so that
// backward -> all-reduce -> Adam is stream-ordered with no host synchronisation in between.
//
// librccl.so.1 is resolved with dlopen at the first ilsx_comm_* call: a single-GPU user of libilsx.so has no RCCL
// dependency, and inside a torch process the already-loaded librccl (same SONAME) is the one that gets used.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "host_common.h"

namespace {
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.so) return ILSX_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) {
    so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (so) break;
  }
  if (!so) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name)                                                                   \
  do {                                                                                     \
    *(void**)(&g_rccl.field) = dlsym(so, name);                                            \
    if (!g_rccl.field) { dlclose(so); ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "RCCL symbol %s missing", name); } \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.so = so;
  return ILSX_OK;
}
}  // namespace

#define NCCLCHK(expr)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (expr);                                                                      \
    if (r_ != ncclSuccess) ILSX_FAIL(ILSX_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)

static_assert(ILSX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int ilsx_comm_unique_id(uint8_t* id_host) {
  if (!id_host) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_comm_unique_id: NULL argument");
  ILSX_TRY(rccl_load());
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id_host, id.internal, ILSX_COMM_ID_BYTES);
  return ILSX_OK;
}

extern "C" int ilsx_comm_init(ilsx_ctx* c, const uint8_t* id_host, int n_ranks, int rank) {
  if (!c || !id_host || n_ranks < 1 || rank < 0 || rank >= n_ranks) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_comm_init: bad argument");
  if (c->comm) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_comm_init: this ctx already has a communicator");
  ILSX_TRY(rccl_load());
  HIPCHK(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(id.internal, id_host, ILSX_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  NCCLCHK(g_rccl.CommInitRank(&comm, n_ranks, id, rank));
  c->comm = comm; c->comm_n = n_ranks; c->comm_rank = rank;
  return ILSX_OK;
}

extern "C" int ilsx_comm_destroy(ilsx_ctx* c) {
  if (!c || !c->comm) return ILSX_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  g_rccl.CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr; c->comm_n = 0; c->comm_rank = 0;
  return ILSX_OK;
}

extern "C" int ilsx_comm_info(const ilsx_ctx* c, int* n_ranks, int* rank) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
  if (n_ranks) *n_ranks = c->comm ? c->comm_n : 0;
  if (rank) *rank = c->comm ? c->comm_rank : 0;
  return ILSX_OK;
}

int comm_allreduce_sum(ilsx_ctx* c, float* buf, size_t n) {
  if (!c->comm) ILSX_FAIL(ILSX_ERR_STATE, "no communicator on this ctx (ilsx_comm_init)");
  NCCLCHK(g_rccl.AllReduce(buf, buf, n, ncclFloat32, ncclSum, (ncclComm_t)c->comm, c->stream));
  return ILSX_OK;
}

extern "C" int ilsx_comm_allreduce_sum(ilsx_ctx* c, float* dev_buf, size_t n) {
  if (!c || (!dev_buf && n)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_comm_allreduce_sum: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (n == 0) return ILSX_OK;
  return comm_allreduce_sum(c, dev_buf, n);
}
