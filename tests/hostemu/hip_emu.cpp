// hip_emu.cpp -- TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <chrono>
#include <map>
#include <mutex>

#if !defined(__x86_64__)
#error "the SIMT emulator's context switch is written for x86-64"
#endif

// void tkz_emu_switch(void** save_sp, void* load_sp): save callee-saved state, swap stacks.
asm(".text\n.globl tkz_emu_switch\n.type tkz_emu_switch,@function\n"
    "tkz_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");
extern "C" void tkz_emu_switch(void** save_sp, void* load_sp);

namespace hipemu {
int g_tid, g_nthreads;
int64_t g_bid, g_nblocks;

namespace {
constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 512 << 10;
struct Fiber { void* sp; char* stack; bool done; };
Fiber g_f[kMaxThreads];
void* g_main_sp;
const std::function<void()>* g_body;
uint64_t g_progress;

struct WaveState { uint64_t vals[2][64]; uint64_t contrib; int arrived, alive; uint32_t gen; };
WaveState g_w[kMaxThreads / 64];
struct BlockState { int arrived, alive; uint32_t gen; } g_b;

inline void yield_to_main() { tkz_emu_switch(&g_f[g_tid].sp, g_main_sp); }

void wave_complete(WaveState& w) {
    for (int l = 0; l < 64; ++l) if (!((w.contrib >> l) & 1)) w.vals[w.gen & 1][l] = 0;
    w.contrib = 0; w.arrived = 0; ++w.gen; ++g_progress;
}
void block_complete() { g_b.arrived = 0; ++g_b.gen; ++g_progress; }

void fiber_exit() {
    WaveState& w = g_w[g_tid >> 6];
    --w.alive;
    if (w.alive > 0 && w.arrived == w.alive) wave_complete(w);
    --g_b.alive;
    if (g_b.alive > 0 && g_b.arrived == g_b.alive) block_complete();
    g_f[g_tid].done = true; ++g_progress;
    yield_to_main();
    abort();
}
void trampoline() { (*g_body)(); fiber_exit(); }
}  // namespace

uint64_t* wave_exchange(uint64_t v) {
    WaveState& w = g_w[g_tid >> 6];
    const uint32_t g = w.gen; const int l = g_tid & 63;
    w.vals[g & 1][l] = v; w.contrib |= 1ull << l; ++w.arrived;
    if (w.arrived == w.alive) wave_complete(w);
    else while (w.gen == g) yield_to_main();
    return w.vals[g & 1];
}
void block_barrier() {
    const uint32_t g = g_b.gen;
    ++g_b.arrived;
    if (g_b.arrived == g_b.alive) block_complete();
    else while (g_b.gen == g) yield_to_main();
}

void launch(int64_t grid, int block, const std::function<void()>& body) {
    // (one launch at a time: the fibers and the block state are globals -- callers on several host threads take turns, as kernels of one queue do)
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    if (block <= 0 || block > kMaxThreads) { fprintf(stderr, "hipemu: bad block size %d\n", block); abort(); }
    if (grid <= 0) { fprintf(stderr, "hipemu: a launch with a grid of %lld workgroups (an invalid configuration on HIP: the real runtime fails the call)\n", (long long)grid); abort(); }
    for (int t = 0; t < block; ++t)
        if (!g_f[t].stack) {
            g_f[t].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (g_f[t].stack == MAP_FAILED) { perror("mmap"); abort(); }
        }
    g_body = &body; g_nthreads = block; g_nblocks = grid;
    for (int64_t b = 0; b < grid; ++b) {
        g_bid = b;
        for (int t = 0; t < block; ++t) {
            uint64_t* top = (uint64_t*)(g_f[t].stack + kStack);
            top -= 1; top[0] = 0;                             // fake return address: rsp = 8 (mod 16) at trampoline entry
            *--top = (uint64_t)(uintptr_t)&trampoline;        // popped by `ret`
            for (int i = 0; i < 6; ++i) *--top = 0;           // rbp rbx r12 r13 r14 r15
            g_f[t].sp = top; g_f[t].done = false;
        }
        for (int w = 0; w * 64 < block; ++w) {
            memset(&g_w[w], 0, sizeof(WaveState));
            g_w[w].alive = block - w * 64 < 64 ? block - w * 64 : 64;
        }
        g_b.arrived = 0; g_b.alive = block; g_b.gen = 0;
        int remaining = block, stalls = 0;
        while (remaining) {
            const uint64_t before = g_progress;
            for (int t = 0; t < block; ++t) {
                if (g_f[t].done) continue;
                g_tid = t;
                tkz_emu_switch(&g_main_sp, g_f[t].sp);
                if (g_f[t].done) --remaining;
            }
            if (g_progress == before) {
                if (++stalls > 2) {
                    fprintf(stderr, "hipemu: DEADLOCK in block %lld: a collective/barrier was not reached by all live threads\n", (long long)b);
                    for (int w = 0; w * 64 < block; ++w)
                        fprintf(stderr, "  wave %d: arrived %d of %d alive (gen %u)\n", w, g_w[w].arrived, g_w[w].alive, g_w[w].gen);
                    fprintf(stderr, "  block barrier: arrived %d of %d alive\n", g_b.arrived, g_b.alive);
                    abort();
                }
            } else stalls = 0;
        }
    }
}
}  // namespace hipemu

// ---- HIP runtime subset ----
struct hipemu_event { std::chrono::steady_clock::time_point t; };
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p); strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "gfx950:emulated");
    p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = 1; p->sharedMemPerBlock = (size_t)160 << 10; return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
    size_t m = (n + 255) & ~(size_t)255; if (!m) m = 256;
    void* q = aligned_alloc(256, m);
    if (!q) return hipErrorOutOfMemory;
    memset(q, 0xCD, m);   // device memory is not zero-initialised
    *p = q; return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
namespace { std::mutex g_pin_mu; std::map<uintptr_t, size_t> g_pinned; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    const hipError_t r = hipMalloc(p, n);
    if (r == hipSuccess) { std::lock_guard<std::mutex> lock(g_pin_mu); g_pinned[reinterpret_cast<uintptr_t>(*p)] = n; }
    return r;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::lock_guard<std::mutex> lock(g_pin_mu);
    const uintptr_t x = reinterpret_cast<uintptr_t>(p);
    auto it = g_pinned.upper_bound(x);
    if (it != g_pinned.begin()) { --it; if (x < it->first + it->second) { a->type = hipMemoryTypeHost; return hipSuccess; } }
    return hipErrorInvalidValue;
}
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipHostFree(void* p) { { std::lock_guard<std::mutex> lock(g_pin_mu); g_pinned.erase(reinterpret_cast<uintptr_t>(p)); } free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* st) { *st = nullptr; return hipSuccess; }      // (everything is synchronous here)
// (a created-with-flags stream is a distinct non-null handle, so that code choosing a path on "is there a second stream" takes it)
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned flags, int) { return hipStreamCreateWithFlags(st, flags); }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { if (least) *least = 0; if (greatest) *greatest = 0; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { static int dummy; *st = reinterpret_cast<hipStream_t>(&dummy); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}

// development aid: which rule of the o200k multi-byte block scanner refused a block (tkz_pretok.h)
extern "C" { long long tkz_o2_refusals[32] = {0}; }
