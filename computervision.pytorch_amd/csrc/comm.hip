// Data-parallel gradient exchange behind the C ABI (cvx_allreduce_grads(handle, ncclComm_t, hipStream_t)).
//
// RCCL is reached through dlopen (librccl.so of the ROCm install, or the copy PyTorch ships: whichever the process finds): the library
// has no link-time dependency on it, single-GPU users never load it.  One communicator per process (one process per GPU); the
// unique id travels over whatever the caller has (torch.distributed's store, MPI, a file): cvx_comm_unique_id on rank 0, the 128
// bytes to every rank, cvx_comm_create everywhere.
//
// cvx_engine_backward_exchange is the whole data-parallel backward pass in ONE call: the op ranges of `buckets` run back to back on
// the engine's stream; after each range the communication stream waits for it (events), folds the range's weight-gradient slabs and
// SUM-all-reduces the range's slice of the flat gradient arena -- no host code between the ranges, so the exchange of bucket k
// overlaps the kernels of bucket k + 1 however slow the host language is.  (The Python loop it replaces: train.py,
// backward_with_overlapped_exchange -- kept as the gloo test double.)  The mean's 1/world stays folded into the optimiser step.
#include <dlfcn.h>
#include <string.h>

#include "../../include/cvx_engine.h"
#include "cvx_common.h"

namespace {
// the handful of RCCL entry points used, with RCCL's own signatures (rccl.h: ncclResult_t = int, ncclUniqueId = 128 bytes)
struct UniqueId {
  char internal[128];
};
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream);
typedef const char* (*fn_get_error_string)(int);
constexpr int kNcclFloat = 7, kNcclSum = 0;  // ncclFloat32, ncclSum (rccl.h enums)

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_get_error_string err = nullptr;
} g_rccl;

int load_rccl() {
  if (g_rccl.lib) return 0;
  // The copy that is ALREADY in the process first (torch bundles one under its SONAME and has loaded it before this library is used for an
  // exchange): a second, different RCCL in one process would be driven through the hand-declared ABI above against a version it was not
  // written for.  RTLD_NOLOAD probes return NULL when the name is not loaded yet; only then is a copy loaded by name.
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) break;
  if (!h && dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = dlopen(nullptr, RTLD_NOW);  // linked in under another name: resolve through the global scope
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  CVX_CHECK(h, std::string("RCCL not found (dlopen librccl.so): ") + (dlerror() ? dlerror() : ""));
  {  // the entry points below were declared by hand against the NCCL 2.x ABI (by-value 128-byte unique id, enum values of nccl.h 2.x)
    typedef int (*fn_get_version)(int*);
    fn_get_version gv = (fn_get_version)dlsym(h, "ncclGetVersion");
    int ver = 0;
    if (gv && gv(&ver) == 0) CVX_CHECK(ver / 10000 == 2 || (ver >= 2000 && ver < 3000), "RCCL: library version " + std::to_string(ver) + " is not the 2.x ABI this binding was written for");
  }
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  g_rccl.err = (fn_get_error_string)dlsym(h, "ncclGetErrorString");
  CVX_CHECK(g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce, "RCCL: missing entry points");
  g_rccl.lib = h;
  return 0;
}
#define CVX_NCCL(call)                                                                                              \
  do {                                                                                                              \
    int r__ = (call);                                                                                               \
    if (r__ != 0) CVX_FAIL(std::string(#call) + " -> " + (g_rccl.err ? g_rccl.err(r__) : std::to_string(r__)));     \
  } while (0)
}  // namespace

extern "C" int cvx_comm_unique_id(void* out128) {
  CVX_CHECK(out128, "null argument");
  CVX_TRY(load_rccl());
  CVX_NCCL(g_rccl.get_unique_id(reinterpret_cast<UniqueId*>(out128)));
  return 0;
}

extern "C" int cvx_comm_create(void** comm, const void* unique_id128, int32_t rank, int32_t world, int32_t device) {
  CVX_CHECK(comm && unique_id128 && world >= 1 && rank >= 0 && rank < world, "bad arguments");
  CVX_TRY(load_rccl());
  CVX_HIP(hipSetDevice(device));
  UniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  CVX_NCCL(g_rccl.comm_init_rank(comm, world, id, rank));
  return 0;
}

extern "C" int cvx_comm_destroy(void* comm) {
  if (!comm) return 0;
  CVX_TRY(load_rccl());
  CVX_NCCL(g_rccl.comm_destroy(comm));
  return 0;
}

extern "C" int cvx_allreduce_f32(float* data, int64_t count, void* comm, void* hip_stream) {
  CVX_CHECK(data && comm && count > 0, "bad arguments");
  CVX_TRY(load_rccl());
  CVX_NCCL(g_rccl.all_reduce(data, data, (size_t)count, kNcclFloat, kNcclSum, comm, (hipStream_t)hip_stream));
  return 0;
}

// cvx_allreduce_grads / cvx_engine_backward_exchange live in engine.hip (they need the engine's arenas): these are their RCCL halves
int cvx_comm_allreduce_slice(float* base, long long p0, long long p1, void* comm, hipStream_t stream) {
  if (p1 <= p0) return 0;
  return cvx_allreduce_f32(base + p0, p1 - p0, comm, (void*)stream);
}
