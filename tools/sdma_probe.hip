// Can a download (device -> page-locked host memory) run on an SDMA engine BESIDE kernels on this box?  The runtime's hipMemcpyAsync D2H is a blit
// kernel (__amd_rocclr_copyBuffer) that the next chunk's kernels wait for (profiles/r05/midtrace_64.txt).  This probe issues the same copy through
// the HSA runtime underneath HIP (hsa_amd_memory_async_copy[_on_engine]) and times it alone and beside a store-heavy kernel.
//   hipcc --offload-arch=gfx950 -O2 tools/sdma_probe.hip -o tools/_build/sdma_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void stream_store(uint4* dst, const uint4* src, long long n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            uint4 v = src[i]; v.x += r; dst[i] = v;
        }
}
static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MB = 1 << 20, nbytes = 32 * MB, big = 1024 * MB;
    char *d_src, *h_dst, *d_a, *d_b;
    hipMalloc(&d_src, nbytes); hipHostMalloc(&h_dst, nbytes, 0); hipMalloc(&d_a, big); hipMalloc(&d_b, big);
    hipMemset(d_src, 7, nbytes); hipMemset(d_a, 1, big); hipMemset(d_b, 2, big); memset(h_dst, 0, nbytes);
    hipStream_t sk, sc; hipStreamCreateWithFlags(&sk, hipStreamNonBlocking); hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    hipDeviceSynchronize();
    if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 1; }
    hsa_iterate_agents(on_agent, nullptr);
    if (!have_gpu || !have_cpu) { printf("agents not found\n"); return 1; }
    uint32_t eng = 0; hsa_status_t es = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &eng);
    printf("copy engines free for GPU -> host: status %d mask 0x%x\n", (int)es, eng);
    hsa_signal_t sig; hsa_signal_create(1, 0, nullptr, &sig);
    auto kernel = [&](int reps) { hipLaunchKernelGGL(stream_store, dim3(8192), dim3(256), 0, sk, (uint4*)d_a, (const uint4*)d_b, (long long)(big / 16), reps); };
    auto hsa_copy = [&](int engine_bit, size_t off, size_t len) -> hsa_status_t {
        hsa_signal_store_relaxed(sig, 1);
        if (engine_bit < 0) return hsa_amd_memory_async_copy(h_dst + off, g_cpu, d_src + off, g_gpu, len, 0, nullptr, sig);
        return hsa_amd_memory_async_copy_on_engine(h_dst + off, g_cpu, d_src + off, g_gpu, len, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << engine_bit), false);
    };
    auto hsa_wait = [&] { while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {} };
    for (int rep = 0; rep < 3; ++rep) {
        double t0, t1;
        hipDeviceSynchronize(); t0 = now_us(); kernel(1); hipDeviceSynchronize(); t1 = now_us(); const double k_alone = t1 - t0;
        hipDeviceSynchronize(); t0 = now_us(); hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, sc); hipStreamSynchronize(sc); t1 = now_us(); const double hip_alone = t1 - t0;
        hipDeviceSynchronize(); t0 = now_us(); kernel(1); hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, sc); hipStreamSynchronize(sc); const double tc = now_us(); hipDeviceSynchronize(); t1 = now_us();
        printf("rep %d: kernel alone %.0f us; hipMemcpyAsync D2H 32 MB alone %.0f us (%.1f GB/s); both: copy done at %.0f, all done at %.0f\n", rep, k_alone, hip_alone, nbytes / hip_alone / 1e3, tc - t0, t1 - t0);
        memset(h_dst, 0, 64);
        hipDeviceSynchronize(); t0 = now_us(); hsa_status_t s = hsa_copy(-1, 0, nbytes); if (s != HSA_STATUS_SUCCESS) { printf("hsa copy failed %d\n", (int)s); return 1; } hsa_wait(); t1 = now_us(); const double hsa_alone = t1 - t0;
        printf("        hsa_amd_memory_async_copy alone %.0f us (%.1f GB/s), first byte %d\n", hsa_alone, nbytes / hsa_alone / 1e3, (int)h_dst[0]);
        hipDeviceSynchronize(); t0 = now_us(); kernel(1); hsa_copy(-1, 0, nbytes); hsa_wait(); const double tc2 = now_us(); hipDeviceSynchronize(); t1 = now_us();
        printf("        kernel + hsa copy: copy done at %.0f, all done at %.0f\n", tc2 - t0, t1 - t0);
        // kernel queued first AND a second kernel behind the copy (as the chunk loop does): copy issued, then kernel
        hipDeviceSynchronize(); t0 = now_us(); hsa_copy(-1, 0, nbytes); kernel(1); hsa_wait(); const double tc3 = now_us(); hipDeviceSynchronize(); t1 = now_us();
        printf("        hsa copy + kernel behind it: copy done at %.0f, all done at %.0f\n", tc3 - t0, t1 - t0);
        for (int b = 0; b < 8; ++b) if (eng & (1u << b)) {
            hipDeviceSynchronize(); t0 = now_us(); s = hsa_copy(b, 0, nbytes); if (s != HSA_STATUS_SUCCESS) { printf("        engine %d: status %d\n", b, (int)s); continue; } hsa_wait(); t1 = now_us();
            const double alone = t1 - t0;
            hipDeviceSynchronize(); t0 = now_us(); kernel(1); hsa_copy(b, 0, nbytes); hsa_wait(); const double tc4 = now_us(); hipDeviceSynchronize(); t1 = now_us();
            printf("        engine %d: alone %.0f us (%.1f GB/s); beside the kernel: copy done at %.0f, all done at %.0f\n", b, alone, nbytes / alone / 1e3, tc4 - t0, t1 - t0);
        }
        // small copies: the fixed cost of an HSA copy (8 KB, 1 MB)
        for (size_t len : {size_t(8) << 10, size_t(1) << 20, size_t(8) << 20}) {
            hipDeviceSynchronize(); t0 = now_us(); hsa_copy(-1, 0, len); hsa_wait(); t1 = now_us();
            const double a = t1 - t0;
            hipDeviceSynchronize(); t0 = now_us(); hipMemcpyAsync(h_dst, d_src, len, hipMemcpyDeviceToHost, sc); hipStreamSynchronize(sc); t1 = now_us();
            printf("        %zu KB: hsa %.1f us, hipMemcpyAsync %.1f us\n", len >> 10, a, t1 - t0);
        }
    }
    // H2D for completeness
    {
        hipDeviceSynchronize(); double t0 = now_us(); hipMemcpyAsync(d_src, h_dst, nbytes, hipMemcpyHostToDevice, sc); hipStreamSynchronize(sc); double t1 = now_us();
        printf("hipMemcpyAsync H2D 32 MB alone %.0f us (%.1f GB/s)\n", t1 - t0, nbytes / (t1 - t0) / 1e3);
        hipDeviceSynchronize(); t0 = now_us(); kernel(1); hipMemcpyAsync(d_src, h_dst, nbytes, hipMemcpyHostToDevice, sc); hipStreamSynchronize(sc); const double tc = now_us(); hipDeviceSynchronize(); t1 = now_us();
        printf("kernel + H2D: copy done at %.0f, all done at %.0f\n", tc - t0, t1 - t0);
        // both directions at once through HSA + HIP
        hipDeviceSynchronize(); t0 = now_us(); hsa_copy(-1, 0, nbytes); hipMemcpyAsync(d_a, h_dst + 0, nbytes, hipMemcpyHostToDevice, sc); hsa_wait(); const double tc2 = now_us(); hipStreamSynchronize(sc); t1 = now_us();
        printf("hsa D2H + hip H2D of 32 MB each at once: D2H done at %.0f, H2D done at %.0f\n", tc2 - t0, t1 - t0);
    }
    // which engine does a download have to be on for the runtime's upload to run BESIDE it (full duplex)?
    {
        hsa_signal_t sig2; hsa_signal_create(1, 0, nullptr, &sig2);
        for (int b = 0; b < 4; ++b) {
            hipDeviceSynchronize(); double t0 = now_us(); hsa_copy(b, 0, nbytes); hipMemcpyAsync(d_a, h_dst, nbytes, hipMemcpyHostToDevice, sc); hsa_wait(); const double tc = now_us(); hipStreamSynchronize(sc); double t1 = now_us();
            printf("hsa D2H on engine %d + hip H2D: D2H done at %.0f, H2D done at %.0f\n", b, tc - t0, t1 - t0);
            hipDeviceSynchronize(); t0 = now_us(); hipMemcpyAsync(d_a, h_dst, nbytes, hipMemcpyHostToDevice, sc); hsa_copy(b, 0, nbytes); hsa_wait(); const double tc2 = now_us(); hipStreamSynchronize(sc); t1 = now_us();
            printf("hip H2D first + hsa D2H on engine %d: D2H done at %.0f, H2D done at %.0f\n", b, tc2 - t0, t1 - t0);
        }
        char* h_src2; hipHostMalloc(&h_src2, nbytes, 0); memset(h_src2, 3, nbytes);
        for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) if (b != c) {
            hipDeviceSynchronize(); double t0 = now_us();
            hsa_signal_store_relaxed(sig2, 1);
            hsa_status_t s1 = hsa_amd_memory_async_copy_on_engine(d_a, g_gpu, h_src2, g_cpu, nbytes, 0, nullptr, sig2, (hsa_amd_sdma_engine_id_t)(1u << b), false);
            hsa_status_t s2 = hsa_copy(c, 0, nbytes);
            hsa_wait(); const double tc = now_us();
            while (hsa_signal_wait_scacquire(sig2, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
            double t1 = now_us();
            printf("hsa H2D on engine %d (status %d) + hsa D2H on engine %d (status %d): D2H done at %.0f, H2D done at %.0f\n", b, (int)s1, c, (int)s2, tc - t0, t1 - t0);
        }
        // and with the store kernel running as well
        hipDeviceSynchronize(); double t0 = now_us(); kernel(2); hsa_copy(1, 0, nbytes); hipMemcpyAsync(d_src + 0, h_src2, nbytes, hipMemcpyHostToDevice, sc); hsa_wait(); const double tc = now_us(); hipStreamSynchronize(sc); const double th = now_us(); hipDeviceSynchronize(); double t1 = now_us();
        printf("kernel x2 + hsa D2H engine 1 + hip H2D: D2H done at %.0f, H2D done at %.0f, all at %.0f\n", tc - t0, th - t0, t1 - t0);
    }
    return 0;
}
