// tkz_comm.cpp -- the multi-GPU exchange of the batch encode path, in the C ABI (include/tkz.h, "multi-GPU").
//
// The reference is single-process and has no distributed code; the path shards by contiguous document
// ranges, one process per GPU, tables replicated, token ids never leave the GPU that produced them.  The ONLY exchange is one
// all-gather of {n_docs, n_bytes, n_tokens} (24 bytes per rank) per batch, from which every rank derives the global
// document / token base of its shard.  It is issued here directly on RCCL (ncclAllGather over xGMI), so that the C# / C++
// hosts the boundary is written for can shard without any Python or torch in the process.
//
// RCCL is bound at run time (dlopen of librccl.so.1), not at link time, and its header is not needed at build time either (the
// handful of types and prototypes used are declared below, as RCCL's stable C API defines them): a single-GPU host needs no
// RCCL installed to build or to run libtkz, and in a process that already holds a copy (PyTorch bundles one under the same
// soname) the loader hands back that very copy, so there is never a second RCCL -- or a second HIP runtime -- in the process.
// A missing library fails loudly at tkz_comm_create / tkz_comm_unique_id (TKZ_E_UNSUPPORTED), never silently.
#include <dlfcn.h>
#ifdef TKZ_HOSTEMU
#include "hip_emu.h"      // (the CPU-emulated build of the tests: the same code against the emulator's HIP and tests/hostemu/fake_rccl.cpp)
#else
#include <hip/hip_runtime.h>
#endif

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/tkz.h"

namespace tkz { tkz_status set_error(tkz_status s, const std::string& msg); }

namespace {

// the part of RCCL's C API (rccl.h; identical to NCCL's) this file calls
struct ncclUniqueId { char internal[128]; };
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;                  // enum ncclResult_t: ncclSuccess = 0
constexpr ncclResult_t ncclSuccess = 0;
typedef int ncclDataType_t;                // enum ncclDataType_t: ncclInt64 = 4
constexpr ncclDataType_t ncclInt64 = 4;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
        }
        if (!R.handle) { const char* e = dlerror(); R.error = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?"); return; }
        bool ok = true;
        auto sym = [&](const char* name) -> void* { void* p = dlsym(R.handle, name); if (!p) { ok = false; R.error = std::string("librccl lacks ") + name; } return p; };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
        R.CommCount = reinterpret_cast<decltype(R.CommCount)>(sym("ncclCommCount"));
        R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(sym("ncclCommUserRank"));
        R.GetVersion = reinterpret_cast<decltype(R.GetVersion)>(sym("ncclGetVersion"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { dlclose(R.handle); R.handle = nullptr; }
    });
    return &R;
}

static_assert(sizeof(ncclUniqueId) == TKZ_COMM_ID_BYTES, "tkz.h's id size is RCCL's");

struct DeviceGuard {       // the caller's current HIP device is restored on every exit path
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct tkz_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    int64_t* d_mine = nullptr;      // 3 int64
    int64_t* d_table = nullptr;     // world * 3 int64
    std::string backend;
};

#define RCCL_TRY(expr)                                                                                                   \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess) return tkz::set_error(TKZ_E_DEVICE, std::string(#expr) + ": " + R->GetErrorString(r_)); \
    } while (0)
#define HIP_TRY2(expr)                                                                                              \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return tkz::set_error(TKZ_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

tkz_status tkz_comm_unique_id(uint8_t* id128) {
    if (!id128) return tkz::set_error(TKZ_E_ARG, "null id buffer");
    Rccl* R = rccl();
    if (!R->handle) return tkz::set_error(TKZ_E_UNSUPPORTED, R->error);
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return TKZ_OK;
}

tkz_status tkz_comm_create(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, tkz_comm** out) {
    if (!out || !id128) return tkz::set_error(TKZ_E_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return tkz::set_error(TKZ_E_ARG, "rank / world out of range");
    Rccl* R = rccl();
    if (!R->handle) return tkz::set_error(TKZ_E_UNSUPPORTED, R->error);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tkz::set_error(TKZ_E_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return tkz::set_error(TKZ_E_ARG, "device index out of range");
    DeviceGuard guard(device);
    tkz_comm* c = new tkz_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return tkz::set_error(TKZ_E_DEVICE, std::string("ncclCommInitRank: ") + R->GetErrorString(r)); }
    int n = 0, me = -1, ver = 0;
    (void)R->CommCount(c->comm, &n); (void)R->CommUserRank(c->comm, &me); (void)R->GetVersion(&ver);
    if (n != world || me != rank) { R->CommDestroy(c->comm); delete c; return tkz::set_error(TKZ_E_DEVICE, "communicator does not report the requested rank / world size"); }
    char buf[64];
    snprintf(buf, sizeof buf, "rccl %d.%d.%d", ver / 10000, (ver / 100) % 100, ver % 100);
    c->backend = buf;
    if (hipMalloc((void**)&c->d_mine, 3 * sizeof(int64_t)) != hipSuccess || hipMalloc((void**)&c->d_table, (size_t)world * 3 * sizeof(int64_t)) != hipSuccess) {
        tkz_comm_destroy(c);
        return tkz::set_error(TKZ_E_DEVICE, "hipMalloc of the count buffers failed");
    }
    *out = c;
    return TKZ_OK;
}

void tkz_comm_destroy(tkz_comm* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->d_mine) (void)hipFree(c->d_mine);
    if (c->d_table) (void)hipFree(c->d_table);
    Rccl* R = rccl();
    if (c->comm && R->handle) (void)R->CommDestroy(c->comm);
    delete c;
}

int32_t tkz_comm_world(const tkz_comm* c) { return c ? c->world : 0; }
int32_t tkz_comm_rank(const tkz_comm* c) { return c ? c->rank : -1; }
const char* tkz_comm_backend(const tkz_comm* c) { return c ? c->backend.c_str() : ""; }

tkz_status tkz_comm_allgather_counts_device(tkz_comm* c, const int64_t* d_mine, int64_t* d_table, void* hip_stream) {
    if (!c || !d_mine || !d_table) return tkz::set_error(TKZ_E_ARG, "null argument");
    Rccl* R = rccl();
    DeviceGuard guard(c->device);
    RCCL_TRY(R->AllGather(d_mine, d_table, 3, ncclInt64, c->comm, static_cast<hipStream_t>(hip_stream)));
    return TKZ_OK;
}

tkz_status tkz_comm_allgather_counts(tkz_comm* c, int64_t n_docs, int64_t n_bytes, int64_t n_tokens, int64_t* table) {
    if (!c || !table) return tkz::set_error(TKZ_E_ARG, "null argument");
    Rccl* R = rccl();
    DeviceGuard guard(c->device);
    const int64_t mine[3] = {n_docs, n_bytes, n_tokens};
    HIP_TRY2(hipMemcpy(c->d_mine, mine, sizeof mine, hipMemcpyHostToDevice));
    RCCL_TRY(R->AllGather(c->d_mine, c->d_table, 3, ncclInt64, c->comm, (hipStream_t) nullptr));
    HIP_TRY2(hipMemcpy(table, c->d_table, (size_t)c->world * 3 * sizeof(int64_t), hipMemcpyDeviceToHost));   // (synchronises with the null stream)
    return TKZ_OK;
}

}  // extern "C"
