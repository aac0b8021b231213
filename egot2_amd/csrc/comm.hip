// Gradient exchange under the C ABI: egx_comm_* / egx_allreduce = RCCL (ncclAllReduce over xGMI) behind plain
// pointers, for callers that do not go through torch.distributed. One communicator per process (= per GPU); the 128-byte unique id
// travels between the ranks by whatever the caller has (torch.distributed broadcast, MPI, a file). The reference gets its exchange from
// Lightning DDP's reducer (HOI/scripts/multitask/run.py:41-50): one all-reduce (sum, then 1 / n) over the flat gradient buffer.
//
// RCCL is resolved at run time (dlopen): inside a PyTorch process the copy torch already loaded is reused (RTLD_NOLOAD first — two
// RCCL instances in one process would each grab the xGMI / IPC resources), elsewhere librccl.so.1 from the ROCm installation, or
// EGX_RCCL_LIB. The library itself has no link-time dependency on RCCL: a single-GPU user never loads it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "../../include/egot2x.h"
#include "common.h"

namespace egx {
namespace {
// the handful of RCCL declarations used (rccl.h: ncclResult_t = int, ncclSuccess = 0; ncclUniqueId = 128 opaque bytes)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0, ncclAvg = 4 };
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char where[256] = {0};
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char* env = getenv("EGX_RCCL_LIB");
    const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)          // pass 0: a copy this process has already loaded (torch's)
        for (const char* n : names) {
            if (!n) continue;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { strncpy(g_rccl.where, n, sizeof(g_rccl.where) - 1); break; }
        }
    if (!h) return;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce) g_rccl.h = h;
}
const Rccl* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.h ? &g_rccl : nullptr;
}
const char* nccl_err(const Rccl* r, int rc) { return r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"; }
}  // namespace
}  // namespace egx

using namespace egx;

struct egx_comm { ncclComm_t comm; int rank, world; };

extern "C" {

int egx_comm_unique_id(void* id128) {
    EGX_CHECK(id128, "egx_comm_unique_id: null output");
    const Rccl* r = rccl();
    EGX_CHECK(r, "RCCL not found (librccl.so; set EGX_RCCL_LIB)");
    ncclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    EGX_CHECK(rc == 0, "ncclGetUniqueId: %s", nccl_err(r, rc));
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int egx_comm_create(const void* id128, int rank, int world, egx_comm** out) {
    EGX_CHECK(id128 && out && world >= 1 && rank >= 0 && rank < world, "egx_comm_create: bad arguments (rank %d of %d)", rank, world);
    const Rccl* r = rccl();
    EGX_CHECK(r, "RCCL not found (librccl.so; set EGX_RCCL_LIB)");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = r->CommInitRank(&c, world, id, rank);      // binds to the calling thread's current HIP device
    EGX_CHECK(rc == 0 && c, "ncclCommInitRank(rank %d of %d): %s", rank, world, nccl_err(r, rc));
    egx_comm* h = new egx_comm{c, rank, world};
    *out = h;
    return 0;
}

int egx_comm_size(const egx_comm* comm) {
    if (!comm) return -1;
    const Rccl* r = rccl();
    int n = comm->world;
    if (r && r->CommCount && r->CommCount(comm->comm, &n) != 0) return -1;      // what RCCL itself says, not what the caller passed
    return n;
}

int egx_allreduce(egx_comm* comm, void* buf, size_t n, int dtype, int average, void* stream) {
    EGX_CHECK(comm && comm->comm, "egx_allreduce: null communicator");
    EGX_CHECK(dtype == 0 || dtype == 1, "egx_allreduce: dtype %d (0 = fp32, 1 = bf16)", dtype);
    if (n == 0) return 0;
    EGX_CHECK(buf, "egx_allreduce: null buffer");
    const Rccl* r = rccl();
    EGX_CHECK(r, "RCCL not found");
    const int rc = r->AllReduce(buf, buf, n, dtype == 0 ? ncclFloat32 : ncclBfloat16, average ? ncclAvg : ncclSum, comm->comm, (hipStream_t)stream);
    EGX_CHECK(rc == 0, "ncclAllReduce(%zu elements): %s", n, nccl_err(r, rc));
    return 0;
}

int egx_comm_destroy(egx_comm* comm) {
    if (!comm) return 0;
    const Rccl* r = rccl();
    int rc = 0;
    if (r && comm->comm) rc = r->CommDestroy(comm->comm);
    delete comm;
    EGX_CHECK(rc == 0, "ncclCommDestroy: %s", r ? nccl_err(r, rc) : "?");
    return 0;
}

const char* egx_comm_library(void) { return rccl() ? g_rccl.where : ""; }

}  // extern "C"
