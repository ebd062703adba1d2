// hip_emu.h -- TEST INFRASTRUCTURE ONLY.  A small fiber-based SIMT emulator + the subset of the HIP
// runtime API that tokenizer_amd/csrc uses, so that the real kernel sources (tkz_kernels.hip) and the
// real C ABI (tkz_api.cpp) can be compiled with g++ -DTKZ_HOSTEMU and executed on a CPU by the
// `not gpu` tests.  One workgroup runs at a time; its threads are fibers scheduled round-robin, and
// the wave64 collectives / workgroup barrier are rendezvous points between fibers (a collective
// reached by only part of a wave deadlocks and is reported -- kernels must call them convergently,
// which is also what the hardware wants).  Never linked into libtkz.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>

#define TKZ_DEV inline
#define TKZ_HD inline
#define TKZ_KERNEL(bounds) static
#define TKZ_KERNEL_OCC(bounds, waves_per_simd) static
#define TKZ_SHARED static
#define __host__
#define __device__
#define __forceinline__ inline

// ---- HIP runtime subset ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; size_t totalGlobalMem; int multiProcessorCount; size_t sharedMemPerBlock; };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);     // (memory from hipHostMalloc: host; anything else: an error, as for pageable memory)
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned flags);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* st);
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocPortable = 1 };
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags = 0);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };

namespace hipemu {
void launch(int64_t grid, int block, const std::function<void()>& body);
uint64_t* wave_exchange(uint64_t v);   // all live lanes of the calling wave rendezvous; returns the 64 deposited values
void block_barrier();
extern int g_tid, g_nthreads;
extern int64_t g_bid, g_nblocks;
}  // namespace hipemu

#define TKZ_LAUNCH(kernel, grid, block, stream, ...) \
    hipemu::launch((int64_t)(grid), (int)(block), [=]() { kernel(__VA_ARGS__); })

inline uint4 tkz_load16_nt(const void* p) { return *reinterpret_cast<const uint4*>(p); }
inline uint32_t tkz_load_nt(const uint32_t* p) { return *p; }
inline uint32_t tkz_atomic_load_agent(const uint32_t* p) { return *p; }
inline int32_t tkz_load_nt(const int32_t* p) { return *p; }
inline void tkz_store_nt(int32_t* p, int32_t v) { *p = v; }
inline void tkz_store16_nt(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

namespace simt {
inline int tid() { return hipemu::g_tid; }
inline int lane() { return hipemu::g_tid & 63; }
inline int wave() { return hipemu::g_tid >> 6; }
inline uint64_t uniform64(uint64_t v) { return v; }
inline int64_t bid() { return hipemu::g_bid; }
inline int64_t nblocks() { return hipemu::g_nblocks; }
inline int nthreads() { return hipemu::g_nthreads; }
inline void sync() { hipemu::block_barrier(); }
inline uint64_t ballot(bool p) {
    uint64_t* v = hipemu::wave_exchange(p ? 1 : 0);
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) m |= (v[l] & 1ull) << l;
    return m;
}
inline int shfl(int v, int src) { return (int)(uint32_t)hipemu::wave_exchange((uint32_t)v)[src & 63]; }
inline uint32_t shflu(uint32_t v, int src) { return (uint32_t)hipemu::wave_exchange(v)[src & 63]; }
inline int shfl_up(int v, int d) {
    int l = lane();
    uint64_t* a = hipemu::wave_exchange((uint32_t)v);
    return l - d >= 0 ? (int)(uint32_t)a[l - d] : v;
}
inline int shfl_xor(int v, int m) { return (int)(uint32_t)hipemu::wave_exchange((uint32_t)v)[(lane() ^ m) & 63]; }
inline int first_lane(int v) {   // value of the lowest live lane
    uint64_t* a = hipemu::wave_exchange(((uint64_t)1 << 32) | (uint32_t)v);
    for (int l = 0; l < 64; ++l) if (a[l] >> 32) return (int)(uint32_t)a[l];
    return v;
}
inline int scan_inclusive(int v) {
    const int l = lane();
    uint64_t* a = hipemu::wave_exchange((uint32_t)v);
    int x = 0;
    for (int i = 0; i <= l; ++i) x += (int)(uint32_t)a[i];
    return x;
}
inline uint32_t wave_min_u32(uint32_t v) {
    uint64_t* a = hipemu::wave_exchange(v);
    uint32_t m = 0xFFFFFFFFu;
    for (int i = 0; i < 64; ++i) m = (uint32_t)a[i] < m ? (uint32_t)a[i] : m;
    return m;
}
inline int last_lane(int v) { return (int)(uint32_t)hipemu::wave_exchange((uint32_t)v)[63]; }
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u)); }
inline int atomic_add(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomic_or(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
inline unsigned atomic_and(unsigned* p, unsigned v) { unsigned o = *p; *p = o & v; return o; }
inline unsigned atomic_max(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline unsigned atomic_cas(unsigned* p, unsigned expect, unsigned v) { unsigned o = *p; if (o == expect) *p = v; return o; }
inline void fence() {}
inline unsigned long long atomic_or64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o | v; return o; }
inline unsigned long long atomic_add64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline long long clock() { return 0; }
inline unsigned long long atomic_max64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomic_min64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
}  // namespace simt
