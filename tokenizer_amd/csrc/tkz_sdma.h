// tkz_sdma.h -- downloads (device -> page-locked host memory) on a copy engine of their own.
//
// The HIP runtime's hipMemcpyAsync D2H of the chunk pipeline (tkz_api.cpp, encode_host) turned out to be a blit KERNEL that the next chunk's kernels
// wait for, and -- when it does go to a copy engine -- the SAME engine the runtime's uploads use (tools/sdma_probe.hip: a 32 MB download beside a
// 32 MB upload ends at 0.6 / 1.2 ms through the runtime, at 0.70 / 0.70 ms on two engines).  The HSA runtime underneath HIP takes the engine as an
// argument (hsa_amd_memory_async_copy_on_engine): bound at run time from libhsa-runtime64.so.1, which libamdhip64 has loaded anyway -- no header and no
// link-time dependency, like RCCL in tkz_comm.cpp.  Everything here is optional: when the library, an entry point or an agent is missing, or
// TKZ_D2H_ENGINE=-1 (0..15 names an engine), available() is false and the caller keeps to hipMemcpyAsync.
#pragma once
#include <cstddef>
#include <cstdint>

namespace tkz {

struct SdmaSignal { uint64_t handle = 0; };

// Can downloads from HIP device `hip_device` go this way?  (the first call binds the library and finds the agents; thread-safe)
bool sdma_available(int hip_device);
bool sdma_signal_create(SdmaSignal* s);
void sdma_signal_destroy(SdmaSignal* s);
// `n` copies are about to be issued against the signal: it reads n until they complete (each takes one off)
void sdma_signal_arm(SdmaSignal s, int64_t n);
// dev_src[0, bytes) -> host_dst, asynchronous.  The source must be complete and visible (the caller has synchronised the stream that wrote it).
// false: nothing was issued (the signal is as it was) -- use the runtime's copy for this one
bool sdma_copy_d2h(int hip_device, void* host_dst, const void* dev_src, size_t bytes, SdmaSignal s);
// until every armed copy has completed; false on a copy engine error (the signal went negative)
bool sdma_signal_wait(SdmaSignal s);

}  // namespace tkz
