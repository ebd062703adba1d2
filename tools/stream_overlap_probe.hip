// Do kernels of two streams run side by side on this box?  Two streams, each with ONE kernel of `wgs` workgroups that spin for ~`ms` milliseconds.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_overlap_probe.hip -o tools/_build/stream_overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    long long n = 0;
    while (wall_clock64() - t0 < ticks) ++n;
    if (n == -1) *sink = 1;
}
static double run(hipStream_t a, hipStream_t b, int wgs_a, int wgs_b, long long ticks, int* sink, bool both) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(wgs_a), dim3(256), 0, a, ticks, sink);
    if (both) hipLaunchKernelGGL(spin, dim3(wgs_b), dim3(256), 0, b, ticks, sink);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
    const long long ticks = (long long)rate * 5;                                            // 5 ms
    hipStream_t s[4];
    hipStreamCreate(&s[0]); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    int least = 0, greatest = 0; hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithPriority(&s[2], hipStreamNonBlocking, greatest); hipStreamCreateWithFlags(&s[3], hipStreamNonBlocking);
    printf("wall clock %d kHz, priorities least %d greatest %d\n", rate, least, greatest);
    run(s[0], s[1], 64, 64, ticks, sink, true);
    const char* names[4] = {"blocking", "nonblocking", "priority", "nonblocking2"};
    for (int wg : {64, 1024, 8192}) {
        printf("one kernel of %d workgroups alone: %.2f ms\n", wg, run(s[0], s[1], wg, wg, ticks, sink, false));
        for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j)
            printf("  %5d + 64 workgroups on %s + %s: %.2f ms\n", wg, names[i], names[j], run(s[i], s[j], wg, 64, ticks, sink, true));
    }
    // with an event dependency, as launch_encode forks: a kernel on s0, then s1 waits for an event recorded behind it and runs beside a second kernel on s0
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[1], ticks / 5, sink);
    hipEventRecord(ev, s[1]); hipStreamWaitEvent(s[3], ev, 0);
    hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, s[1], ticks, sink);
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[3], ticks, sink);
    hipEventRecord(ev, s[3]); hipStreamWaitEvent(s[1], ev, 0);
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s[1], ticks / 5, sink);
    hipDeviceSynchronize();
    printf("fork/join by events: 1 + (8192-workgroup kernel || 64-workgroup kernel, 5 ms each) + 1 ms: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
