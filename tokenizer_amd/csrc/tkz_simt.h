// tkz_simt.h -- the wave64 / workgroup primitives the kernels are written against.
//
// Product build (hipcc, gfx950): thin inline wrappers over the CDNA4 intrinsics (64-lane ballot,
// ds_bpermute shuffles, s_barrier, bit ops).  Everything is written for wavefront = 64.
//
// Test build (-DTKZ_HOSTEMU, g++, tests/hostemu/): the same names are provided by a fiber-based
// SIMT emulator so the *actual kernel sources* can be executed on a CPU by the `not gpu` tests
// before GPU time is spent.  That build is test infrastructure only: it is never linked into
// libtkz.so and nothing in tokenizer_amd/ loads it.
#pragma once
#include <stdint.h>

#ifdef TKZ_HOSTEMU
#include "hip_emu.h"   // tests/hostemu/hip_emu.h
#else
#include <hip/hip_runtime.h>

#define TKZ_DEV __device__ __forceinline__
#define TKZ_HD __host__ __device__ __forceinline__
#define TKZ_KERNEL(bounds) __global__ __launch_bounds__(bounds)
#define TKZ_KERNEL_OCC(bounds, waves_per_simd) __global__ __launch_bounds__(bounds, waves_per_simd)
#define TKZ_SHARED __shared__
#define TKZ_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, (stream), __VA_ARGS__)

namespace simt {
TKZ_DEV int tid() { return (int)threadIdx.x; }
TKZ_DEV int lane() { return (int)(threadIdx.x & 63); }
// (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all 64 lanes; without it every
//  value derived from the wave index -- row numbers, mask words -- is kept in VGPRs and computed on the VALU)
TKZ_DEV int wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
TKZ_DEV uint64_t uniform64(uint64_t v) {   // v is known to be wave-uniform: move it to scalar registers
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
TKZ_DEV int64_t bid() { return (int64_t)blockIdx.x; }
TKZ_DEV int64_t nblocks() { return (int64_t)gridDim.x; }
TKZ_DEV int nthreads() { return (int)blockDim.x; }
TKZ_DEV void sync() { __syncthreads(); }
TKZ_DEV uint64_t ballot(bool p) { return __ballot(p ? 1 : 0); }
TKZ_DEV int shfl(int v, int src) { return __shfl(v, src, 64); }
TKZ_DEV uint32_t shflu(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
TKZ_DEV int shfl_up(int v, int d) { return __shfl_up(v, (unsigned)d, 64); }
TKZ_DEV int shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
TKZ_DEV int first_lane(int v) { return __builtin_amdgcn_readfirstlane(v); }
TKZ_DEV int atomic_add(int* p, int v) { return atomicAdd(p, v); }
TKZ_DEV unsigned atomic_or(unsigned* p, unsigned v) { return atomicOr(p, v); }
TKZ_DEV unsigned atomic_and(unsigned* p, unsigned v) { return atomicAnd(p, v); }
TKZ_DEV unsigned atomic_max(unsigned* p, unsigned v) { return atomicMax(p, v); }
TKZ_DEV unsigned atomic_cas(unsigned* p, unsigned expect, unsigned v) { return atomicCAS(p, expect, v); }
TKZ_DEV void fence() { __threadfence(); }
TKZ_DEV unsigned long long atomic_or64(unsigned long long* p, unsigned long long v) { return atomicOr(p, v); }
TKZ_DEV unsigned long long atomic_add64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
TKZ_DEV unsigned long long atomic_min64(unsigned long long* p, unsigned long long v) { return atomicMin(p, v); }
TKZ_DEV unsigned long long atomic_max64(unsigned long long* p, unsigned long long v) { return atomicMax(p, v); }
TKZ_DEV long long clock() { return (long long)__builtin_readcyclecounter(); }
// inclusive prefix sum over the 64 lanes on the DPP crossbar (row shifts inside the rows of 16, then the two row broadcasts): six
// data-parallel moves, no LDS, no ballots
template <int CTRL> TKZ_DEV int dpp_mov(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }
TKZ_DEV int scan_inclusive(int v) {
    const int l = lane();
    int x = v, t;
    t = dpp_mov<0x111>(x); if ((l & 15) >= 1) x += t;      // row_shr:1
    t = dpp_mov<0x112>(x); if ((l & 15) >= 2) x += t;      // row_shr:2
    t = dpp_mov<0x114>(x); if ((l & 15) >= 4) x += t;      // row_shr:4
    t = dpp_mov<0x118>(x); if ((l & 15) >= 8) x += t;      // row_shr:8
    t = dpp_mov<0x142>(x); if ((l & 31) >= 16) x += t;     // row_bcast:15
    t = dpp_mov<0x143>(x); if (l >= 32) x += t;            // row_bcast:31
    return x;
}
TKZ_DEV int last_lane(int v) { return __builtin_amdgcn_readlane(v, 63); }
// minimum over the 64 lanes (every lane gets it), on the same DPP moves as the scan: no LDS, no ballots
TKZ_DEV uint32_t wave_min_u32(uint32_t v) {
    const int l = lane();
    uint32_t x = v, t;
    t = (uint32_t)dpp_mov<0x111>((int)x); if ((l & 15) >= 1) x = t < x ? t : x;
    t = (uint32_t)dpp_mov<0x112>((int)x); if ((l & 15) >= 2) x = t < x ? t : x;
    t = (uint32_t)dpp_mov<0x114>((int)x); if ((l & 15) >= 4) x = t < x ? t : x;
    t = (uint32_t)dpp_mov<0x118>((int)x); if ((l & 15) >= 8) x = t < x ? t : x;
    t = (uint32_t)dpp_mov<0x142>((int)x); if ((l & 31) >= 16) x = t < x ? t : x;
    t = (uint32_t)dpp_mov<0x143>((int)x); if (l >= 32) x = t < x ? t : x;
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// v_alignbit_b32: the low dword of ({hi, lo} >> (sh & 31)) -- an unaligned dword out of two aligned ones in ONE instruction
TKZ_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
}  // namespace simt
// Streaming accesses (touched once: the corpus, the per-piece records, the ids): non-temporal, so that they do not evict the
// vocabulary tables every gather of the encode kernels wants to find in L2.
typedef uint32_t tkz_u32x4 __attribute__((ext_vector_type(4)));
TKZ_DEV uint4 tkz_load16_nt(const void* p) {
    const tkz_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const tkz_u32x4*>(p));
    uint4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
TKZ_DEV uint32_t tkz_load_nt(const uint32_t* p) { return __builtin_nontemporal_load(p); }
// a word other workgroups -- on other XCDs, whose L2 is not this one's -- update with device-scope atomics: read where those atomics are performed
TKZ_DEV uint32_t tkz_atomic_load_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TKZ_DEV int32_t tkz_load_nt(const int32_t* p) { return __builtin_nontemporal_load(p); }
TKZ_DEV void tkz_store_nt(int32_t* p, int32_t v) { __builtin_nontemporal_store(v, p); }
TKZ_DEV void tkz_store16_nt(void* p, uint4 v) {
    tkz_u32x4 x; x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    __builtin_nontemporal_store(x, reinterpret_cast<tkz_u32x4*>(p));
}
#endif

// ---- bit helpers shared by host and device code ------------------------------------------------
TKZ_HD int tkz_popc64(uint64_t x) { return __builtin_popcountll(x); }
TKZ_HD int tkz_popc32(uint32_t x) { return __builtin_popcount(x); }
// (the count-zero builtins are undefined for 0 and the GPU and the host disagree about what they return then: the CPU-emulated
//  build of the tests stops on such a call instead of letting it pass by luck)
#ifdef TKZ_HOSTEMU
#include <cstdio>
#include <cstdlib>
#define TKZ_NONZERO(x, what) do { if (!(x)) { fprintf(stderr, "tkz: %s(0) is undefined\n", what); abort(); } } while (0)
#else
#define TKZ_NONZERO(x, what) ((void)0)
#endif
TKZ_HD int tkz_ctz64(uint64_t x) { TKZ_NONZERO(x, "tkz_ctz64"); return __builtin_ctzll(x); }   // x != 0
TKZ_HD int tkz_ctz32(uint32_t x) { TKZ_NONZERO(x, "tkz_ctz32"); return __builtin_ctz(x); }     // x != 0
TKZ_HD int tkz_msb64(uint64_t x) { TKZ_NONZERO(x, "tkz_msb64"); return 63 - __builtin_clzll(x); }   // x != 0
TKZ_HD int tkz_msb32(uint32_t x) { TKZ_NONZERO(x, "tkz_msb32"); return 31 - __builtin_clz(x); }     // x != 0
TKZ_HD int tkz_ctz64z(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }                         // any x: 64 for 0
TKZ_HD uint64_t tkz_brev64(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
}
// bits [0, n) set, n in [0, 64]
TKZ_HD uint64_t tkz_lowmask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
