// Chain sharding over the GPUs of one node: the one collective of the HMC
// path, straight on RCCL.
//
// Chains are independent; the only coupling in the reference is through the
// global adaptation statistics -- tf.reduce_mean(acceptance_rate) feeding
// dual averaging (zhusuan/hmc.py:377) and the step-size search (:326), and the
// chain-axis means of the EWMV mass estimator (:138,143).  Both travel in ONE
// ncclAllReduce(sum) of 2 (+ 2*D) doubles per transition, enqueued on the
// compute stream between two transition kernels (include/zshmc.h,
// zshmc_adapt_link).  Messages are <= 16 KiB, i.e. latency-bound on xGMI: no
// bucketing, no second stream -- the next kernel needs the sum anyway.
//
// librccl.so is opened at run time: the copy already mapped into the process
// (torch loads one) if there is one, else the ROCm one.  A single-GPU user
// never touches it.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace zshmc {

// the slice of nccl.h this file uses (RCCL keeps NCCL's ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
enum { kNcclSuccess = 0 };
enum { kNcclFloat64 = 8 };  // ncclDataType_t::ncclDouble
enum { kNcclSum = 0 };      // ncclRedOp_t::ncclSum

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t,
                   hipStream_t) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

// why the library could not be set up (filled once, by rccl_load)
static char g_rccl_why[256] = "";

static bool rccl_load(Rccl* lib) {
  const char* names[] = {getenv("ZSHMC_RCCL_PATH"), "librccl.so.1",
                         "librccl.so", "/opt/rocm/lib/librccl.so.1",
                         "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  // the copy the process already has (torch's), so that there is one RCCL
  for (const char* n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    const char* e = dlerror();  // (one call: dlerror clears what it returns)
    snprintf(g_rccl_why, sizeof(g_rccl_why), "%s", e ? e : "dlopen failed");
    return false;
  }
#define ZS_SYM(field, name)                                              \
  lib->field = reinterpret_cast<decltype(lib->field)>(dlsym(h, name));   \
  if (!lib->field) {                                                     \
    snprintf(g_rccl_why, sizeof(g_rccl_why), "symbol %s missing", name); \
    dlclose(h);                                                          \
    return false;                                                        \
  }
  ZS_SYM(GetUniqueId, "ncclGetUniqueId")
  ZS_SYM(CommInitRank, "ncclCommInitRank")
  ZS_SYM(AllReduce, "ncclAllReduce")
  ZS_SYM(CommDestroy, "ncclCommDestroy")
  ZS_SYM(GetErrorString, "ncclGetErrorString")
#undef ZS_SYM
  lib->handle = h;
  return true;
}

static Rccl* rccl() {
  static Rccl lib;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] { ok = rccl_load(&lib); });
  return ok ? &lib : nullptr;
}

struct Comm {
  ncclComm_t comm;
  int rank, world;
};

static int check_nccl(Rccl* r, int rc, const char* what) {
  if (rc == kNcclSuccess) return ZSHMC_OK;
  set_error("%s: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "?");
  return ZSHMC_ERR_COMM;
}

#define ZS_NEED_RCCL(r)                                                      \
  Rccl* r = rccl();                                                          \
  if (!r) {                                                                  \
    set_error("librccl.so could not be set up (set ZSHMC_RCCL_PATH): %s",    \
              g_rccl_why);                                                   \
    return ZSHMC_ERR_COMM;                                                   \
  }

}  // namespace zshmc

using namespace zshmc;

static_assert(sizeof(ncclUniqueId) == ZSHMC_COMM_ID_BYTES, "ncclUniqueId size");

extern "C" int zshmc_comm_unique_id(void* id_host) {
  ZS_REQUIRE(id_host, "zshmc_comm_unique_id: null buffer");
  ZS_NEED_RCCL(r)
  ncclUniqueId id;
  const int rc = check_nccl(r, r->GetUniqueId(&id), "ncclGetUniqueId");
  if (rc == ZSHMC_OK) memcpy(id_host, &id, sizeof(id));
  return rc;
}

extern "C" int zshmc_comm_create(const void* id_host, int rank, int world_size,
                                 void** comm_out) {
  ZS_REQUIRE(id_host && comm_out, "zshmc_comm_create: null pointer");
  ZS_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size,
             "zshmc_comm_create: rank %d of %d", rank, world_size);
  ZS_NEED_RCCL(r)
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  const int rc = check_nccl(r, r->CommInitRank(&c, world_size, id, rank),
                            "ncclCommInitRank");
  if (rc != ZSHMC_OK) return rc;
  *comm_out = new Comm{c, rank, world_size};
  return ZSHMC_OK;
}

extern "C" int zshmc_comm_all_reduce_sum(void* comm, double* buf, int64_t count,
                                         void* stream) {
  ZS_REQUIRE(comm && buf && count > 0, "zshmc_comm_all_reduce_sum: bad argument");
  ZS_NEED_RCCL(r)
  Comm* c = static_cast<Comm*>(comm);
  return check_nccl(
      r,
      r->AllReduce(buf, buf, (size_t)count, kNcclFloat64, kNcclSum, c->comm,
                   reinterpret_cast<hipStream_t>(stream)),
      "ncclAllReduce");
}

extern "C" int zshmc_comm_world_size(void* comm) {
  return comm ? static_cast<Comm*>(comm)->world : 0;
}

extern "C" int zshmc_comm_destroy(void* comm) {
  if (!comm) return ZSHMC_OK;
  ZS_NEED_RCCL(r)
  Comm* c = static_cast<Comm*>(comm);
  const int rc = check_nccl(r, r->CommDestroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}
