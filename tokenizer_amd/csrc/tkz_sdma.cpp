// tkz_sdma.cpp -- see tkz_sdma.h.  The handful of HSA declarations this needs are restated here (public ABI of ROCr: hsa.h / hsa_ext_amd.h).
#include "tkz_sdma.h"

#ifdef TKZ_HOSTEMU
// (the CPU emulation of the `not gpu` tests has no copy engines: the chunk pipeline keeps to its emulated hipMemcpyAsync)
namespace tkz {
bool sdma_available(int) { return false; }
bool sdma_signal_create(SdmaSignal*) { return false; }
void sdma_signal_destroy(SdmaSignal*) {}
void sdma_signal_arm(SdmaSignal, int64_t) {}
bool sdma_copy_d2h(int, void*, const void*, size_t, SdmaSignal) { return false; }
bool sdma_signal_wait(SdmaSignal) { return true; }
}  // namespace tkz
#else

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace tkz {
namespace {

struct hsa_agent_t { uint64_t handle; };
struct hsa_signal_t { uint64_t handle; };
typedef int hsa_status_t;
typedef int64_t hsa_signal_value_t;
constexpr int kHsaAgentInfoDevice = 17;                 // HSA_AGENT_INFO_DEVICE
constexpr int kHsaDeviceCpu = 0, kHsaDeviceGpu = 1;     // hsa_device_type_t
constexpr int kHsaAmdAgentInfoBdfid = 0xA006;           // HSA_AMD_AGENT_INFO_BDFID: bus << 8 | device << 3 | function
constexpr int kHsaAmdAgentInfoDomain = 0xA00F;          // HSA_AMD_AGENT_INFO_DOMAIN
constexpr int kHsaConditionLt = 2, kHsaWaitBlocked = 0; // HSA_SIGNAL_CONDITION_LT, HSA_WAIT_STATE_BLOCKED

struct Hsa {
    hsa_status_t (*init)() = nullptr;
    hsa_status_t (*iterate_agents)(hsa_status_t (*)(hsa_agent_t, void*), void*) = nullptr;
    hsa_status_t (*agent_get_info)(hsa_agent_t, int, void*) = nullptr;
    hsa_status_t (*signal_create)(hsa_signal_value_t, uint32_t, const hsa_agent_t*, hsa_signal_t*) = nullptr;
    hsa_status_t (*signal_destroy)(hsa_signal_t) = nullptr;
    void (*signal_store_relaxed)(hsa_signal_t, hsa_signal_value_t) = nullptr;
    hsa_signal_value_t (*signal_wait_scacquire)(hsa_signal_t, int, hsa_signal_value_t, uint64_t, int) = nullptr;
    hsa_status_t (*copy_on_engine)(void*, hsa_agent_t, const void*, hsa_agent_t, size_t, uint32_t, const hsa_signal_t*, hsa_signal_t, int, bool) = nullptr;
    hsa_status_t (*engine_status)(hsa_agent_t, hsa_agent_t, uint32_t*) = nullptr;
    bool bound = false;
    int engine = -2;                       // $TKZ_D2H_ENGINE: the engine's bit number; -1: off; unset (-2): chosen per device (agent_of)
    hsa_agent_t cpu{0};
    struct Gpu { hsa_agent_t agent; uint32_t bdf, domain; };
    std::vector<Gpu> gpus;
    std::vector<int> dev_state;            // per HIP device: 0 not looked at, 1 usable, -1 not
    std::vector<hsa_agent_t> dev_agent;
    std::vector<int> dev_engine;
    std::mutex mu;
};
Hsa g;

hsa_status_t on_agent(hsa_agent_t a, void*) {
    int type = -1;
    if (g.agent_get_info(a, kHsaAgentInfoDevice, &type) != 0) return 0;
    if (type == kHsaDeviceCpu && !g.cpu.handle) g.cpu = a;
    if (type == kHsaDeviceGpu) {
        Hsa::Gpu u{a, 0, 0};
        (void)g.agent_get_info(a, kHsaAmdAgentInfoBdfid, &u.bdf);
        (void)g.agent_get_info(a, kHsaAmdAgentInfoDomain, &u.domain);
        g.gpus.push_back(u);
    }
    return 0;
}

// (under g.mu)
bool bind() {
    static bool tried = false;
    if (tried) return g.bound;
    tried = true;
    const char* ev = getenv("TKZ_D2H_ENGINE");
    if (ev && *ev) g.engine = atoi(ev);
    if (g.engine == -1 || g.engine < -2 || g.engine > 15) return false;
    void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
#define TKZ_SYM(field, name) if (!(*reinterpret_cast<void**>(&g.field) = dlsym(h, name))) return false
    TKZ_SYM(init, "hsa_init");
    TKZ_SYM(iterate_agents, "hsa_iterate_agents");
    TKZ_SYM(agent_get_info, "hsa_agent_get_info");
    TKZ_SYM(signal_create, "hsa_signal_create");
    TKZ_SYM(signal_destroy, "hsa_signal_destroy");
    TKZ_SYM(signal_store_relaxed, "hsa_signal_store_relaxed");
    TKZ_SYM(signal_wait_scacquire, "hsa_signal_wait_scacquire");
    TKZ_SYM(copy_on_engine, "hsa_amd_memory_async_copy_on_engine");
    TKZ_SYM(engine_status, "hsa_amd_memory_copy_engine_status");
#undef TKZ_SYM
    // (the HIP runtime has initialised HSA already: this takes one more reference, never given back -- the runtime stays up as long as the process)
    if (g.init() != 0) return false;
    if (g.iterate_agents(on_agent, nullptr) != 0 || !g.cpu.handle || g.gpus.empty()) return false;
    g.bound = true;
    return true;
}

// the HSA agent of a HIP device: by PCI address (HIP_VISIBLE_DEVICES renumbers HIP's devices, not HSA's agents)
bool agent_of(int dev, hsa_agent_t* out, int* engine = nullptr) {
    std::lock_guard<std::mutex> lk(g.mu);
    if (!bind() || dev < 0) return false;
    if ((size_t)dev >= g.dev_state.size()) { g.dev_state.resize((size_t)dev + 1, 0); g.dev_agent.resize((size_t)dev + 1, hsa_agent_t{0}); g.dev_engine.resize((size_t)dev + 1, 0); }
    if (g.dev_state[(size_t)dev] == 0) {
        g.dev_state[(size_t)dev] = -1;
        int bus = -1, device = -1, domain = -1;
        if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev) == hipSuccess && hipDeviceGetAttribute(&device, hipDeviceAttributePciDeviceId, dev) == hipSuccess &&
            hipDeviceGetAttribute(&domain, hipDeviceAttributePciDomainID, dev) == hipSuccess) {
            for (const Hsa::Gpu& u : g.gpus) {
                if ((int)(u.bdf >> 8) == bus && (int)((u.bdf >> 3) & 31u) == device && (int)u.domain == domain) {
                    uint32_t mask = 0;
                    // WHICH engine.  The HIP runtime's uploads take the LOWEST engine that is free when they are issued (engine 0 in a process that uses no
                    // other; engine 1 in tools/sdma_probe.hip, whose own first copy held engine 0), and a download on the same engine queues behind them: 64 MB
                    // of page-locked text as 4 chunks took 2.89 ms with the downloads on engine 0, 2.26-2.28 on engines 1, 2 or 3 (profiles/r06/sdma_engines.txt).
                    // On this chip engines 0..3 move 56 GB/s to host memory and 4..7 12.7 (the xGMI ones): the HIGHEST of the first four it reports.
                    int eng = g.engine;
                    if (g.engine_status(g.cpu, u.agent, &mask) != 0) break;
                    if (eng == -2) { eng = -1; for (int b = 3; b >= 0; --b) if ((mask >> b) & 1u) { eng = b; break; } }
                    if (eng >= 0 && ((mask >> eng) & 1u)) { g.dev_agent[(size_t)dev] = u.agent; g.dev_engine[(size_t)dev] = eng; g.dev_state[(size_t)dev] = 1; }
                    break;
                }
            }
        }
        (void)hipGetLastError();
    }
    if (g.dev_state[(size_t)dev] != 1) return false;
    *out = g.dev_agent[(size_t)dev];
    if (engine) *engine = g.dev_engine[(size_t)dev];
    return true;
}

}  // namespace

bool sdma_available(int hip_device) { hsa_agent_t a; return agent_of(hip_device, &a); }

bool sdma_signal_create(SdmaSignal* s) {
    if (!g.bound) return false;
    hsa_signal_t sig{0};
    if (g.signal_create(0, 0, nullptr, &sig) != 0 || !sig.handle) return false;
    s->handle = sig.handle;
    return true;
}

void sdma_signal_destroy(SdmaSignal* s) {
    if (s->handle && g.bound) (void)g.signal_destroy(hsa_signal_t{s->handle});
    s->handle = 0;
}

void sdma_signal_arm(SdmaSignal s, int64_t n) { g.signal_store_relaxed(hsa_signal_t{s.handle}, n); }

bool sdma_copy_d2h(int hip_device, void* host_dst, const void* dev_src, size_t bytes, SdmaSignal s) {
    hsa_agent_t gpu;
    int engine = 0;
    if (!s.handle || !bytes || !agent_of(hip_device, &gpu, &engine)) return false;
    return g.copy_on_engine(host_dst, g.cpu, dev_src, gpu, bytes, 0, nullptr, hsa_signal_t{s.handle}, 1 << engine, false) == 0;
}

bool sdma_signal_wait(SdmaSignal s) {
    if (!s.handle) return true;
    for (;;) {
        const hsa_signal_value_t v = g.signal_wait_scacquire(hsa_signal_t{s.handle}, kHsaConditionLt, 1, UINT64_MAX, kHsaWaitBlocked);
        if (v < 1) return v == 0;
    }
}

}  // namespace tkz
#endif
