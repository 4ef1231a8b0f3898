// comm.hip -- the data-parallel exchange step behind the C ABI (SURVEY.md §8(b),(e)).
// The reference's only parallelism on this path is the all-reduce Mesh-TensorFlow inserts for `layout: batch_dim:data`
// (src/model_fns.py:81-82,189; VAE: CrossShardOptimizer, src/model_fns_tf.py:61).  Here: one process per GPU, one RCCL
// communicator per process, SUM all-reduce of flat fp32 gradient buckets in place, enqueued on the HIP stream the caller
// passes (the engine's side stream, ordered after the bucket's last weight-gradient kernel by an event).
// RCCL is bound at run time (dlopen): the host process normally already maps one librccl (PyTorch-ROCm ships its own),
// and two copies of the library in one address space must not be mixed.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

typedef void* rcclComm_t;
typedef struct { char internal[128]; } rcclUniqueId;   // NCCL_UNIQUE_ID_BYTES
enum { RCCL_FLOAT32 = 7, RCCL_SUM = 0 };

static struct {
  void* lib;
  int (*GetUniqueId)(rcclUniqueId*);
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int);
  int (*CommDestroy)(rcclComm_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t);
  const char* (*GetErrorString)(int);
} g_rccl;

#define RCCL_CALL(expr, what)                                                                          \
  do {                                                                                                 \
    int r__ = (expr);                                                                                  \
    if (r__ != 0) {                                                                                    \
      dmi_set_error("%s: RCCL error %d (%s)", what, r__, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); \
      return DMI_ERR_LAUNCH;                                                                           \
    }                                                                                                  \
  } while (0)

// Bind librccl.  path == NULL / "": the default search ("librccl.so.1", then "librccl.so").  Idempotent.
extern "C" int dmi_comm_load(const char* path) {
  if (g_rccl.lib) return DMI_OK;
  void* h = nullptr;
  if (path && path[0]) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  DMI_REQUIRE(h != nullptr, "comm_load: cannot load librccl (%s)", dlerror());
  g_rccl.GetUniqueId = (int (*)(rcclUniqueId*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(rcclComm_t))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(h, "ncclBroadcast");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  DMI_REQUIRE(g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Broadcast,
              "comm_load: librccl lacks a required symbol");
  g_rccl.lib = h;
  return DMI_OK;
}

extern "C" int dmi_comm_unique_id_bytes(void) { return (int)sizeof(rcclUniqueId); }
// rank 0: fills id_out (HOST, dmi_comm_unique_id_bytes() bytes); the caller ships the bytes to every rank out of band
extern "C" int dmi_comm_unique_id(void* id_out) {
  DMI_REQUIRE(id_out, "comm_unique_id: null pointer");
  int rc = dmi_comm_load(nullptr);
  if (rc) return rc;
  rcclUniqueId id;
  RCCL_CALL(g_rccl.GetUniqueId(&id), "comm_unique_id");
  memcpy(id_out, &id, sizeof(id));
  return DMI_OK;
}
// collective over all ranks (blocking): the calling thread's current HIP device becomes this rank's GPU
extern "C" int dmi_comm_init(void** comm_out, int nranks, int rank, const void* unique_id) {
  DMI_REQUIRE(comm_out && unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
  int rc = dmi_comm_load(nullptr);
  if (rc) return rc;
  rcclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  rcclComm_t c = nullptr;
  RCCL_CALL(g_rccl.CommInitRank(&c, nranks, id, rank), "comm_init");
  *comm_out = c;
  return DMI_OK;
}
extern "C" int dmi_comm_destroy(void* comm) {
  if (!comm || !g_rccl.lib) return DMI_OK;
  RCCL_CALL(g_rccl.CommDestroy((rcclComm_t)comm), "comm_destroy");
  return DMI_OK;
}
// g[0..n) <- sum over ranks of g[0..n)   (fp32, in place), enqueued on `stream`
extern "C" int dmi_allreduce_bucket(void* comm, float* g, int64_t n, void* stream) {
  DMI_REQUIRE(comm && g && n > 0 && g_rccl.lib, "allreduce_bucket: bad arguments / communicator not initialised");
  RCCL_CALL(g_rccl.AllReduce(g, g, (size_t)n, RCCL_FLOAT32, RCCL_SUM, (rcclComm_t)comm, (hipStream_t)stream), "allreduce_bucket");
  return DMI_OK;
}
// buf[0..n) of `root` -> every rank (initial weights / restored state), enqueued on `stream`
extern "C" int dmi_comm_broadcast_f32(void* comm, float* buf, int64_t n, int root, void* stream) {
  DMI_REQUIRE(comm && buf && n > 0 && g_rccl.lib, "comm_broadcast: bad arguments / communicator not initialised");
  RCCL_CALL(g_rccl.Broadcast(buf, buf, (size_t)n, RCCL_FLOAT32, root, (rcclComm_t)comm, (hipStream_t)stream), "comm_broadcast");
  return DMI_OK;
}
