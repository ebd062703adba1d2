// tkz_kernels.h -- host-callable launchers of the HIP kernels (tkz_kernels.hip) and the device
// workspace they operate on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tkz_tables.h"

namespace tkz {

constexpr int kTile = 4096;        // bytes of corpus per workgroup
constexpr int kShortMax = 16;      // pieces up to this many bytes are merged one-per-lane
constexpr int kLongLdsMax = 4096;  // pieces up to this many bytes are merged by one workgroup in LDS

// error bits accumulated on the device (int32 flag word)
enum : int32_t { kErrUtf8 = 1, kErrKeyNotFound = 2, kErrOffsets = 4, kErrQueue = 8 };

struct Workspace {
    // sized from total_bytes / n_docs of the largest batch seen
    uint64_t* docbits = nullptr;     // 1 bit / byte : document starts (+ virtual start at `total`)
    uint64_t* bitmap = nullptr;      // 1 bit / byte : piece starts     (Regex.Matches result)
    uint64_t* validbits = nullptr;   // 1 bit / byte : tmp[i] holds a token
    int32_t* tmp = nullptr;          // 4 B  / byte : token of piece at s lives at tmp[s+k], else -1
    int32_t* tile_count = nullptr;   // tokens whose tmp position falls in the tile
    int64_t* tile_base = nullptr;    // exclusive scan of tile_count
    uint16_t* row_prefix = nullptr;  // per 64-byte row: tokens before the row inside its tile
    int64_t* long_q = nullptr;       // starts of pieces longer than kShortMax
    int64_t* giant_q = nullptr;      // (start,len) of pieces longer than kLongLdsMax
    int32_t* counters = nullptr;     // [0]=err flags [1]=long count [2]=giant count [3]=pad ; int64 total at +4
    int64_t cap_bytes = 0, cap_docs = 0;
    int64_t long_cap = 0, giant_cap = 0;
    int64_t bytes_allocated = 0;
    // giant-piece scratch, allocated on demand
    int32_t* giant_scratch = nullptr; int64_t giant_scratch_elems = 0;
};

struct LaunchTimes;   // profiling hook (see tkz_api.cpp)
typedef void (*KernelHook)(void* ctx, int kernel_id, int phase /*0 before,1 after*/, hipStream_t s);

struct Launch {
    hipStream_t stream;
    KernelHook hook; void* hook_ctx;
};

void launch_docmark(const Launch& L, const int64_t* d_offs, int64_t n_items, int64_t total, uint64_t* bits,
                    int32_t* counters, bool check_bounds);
void launch_pretok(const Launch& L, const TkzTables& T, const uint8_t* d_bytes, int64_t total, const Workspace& w);
void launch_encode(const Launch& L, const TkzTables& T, const uint8_t* d_bytes, int64_t total, const Workspace& w);
void launch_long(const Launch& L, const TkzTables& T, const uint8_t* d_bytes, int64_t total, const Workspace& w,
                 int64_t n_long);
void launch_giant(const Launch& L, const TkzTables& T, const uint8_t* d_bytes, int64_t total, const Workspace& w,
                  int64_t n_giant);
void launch_scan(const Launch& L, const Workspace& w, int64_t ntiles);
void launch_compact(const Launch& L, const Workspace& w, int64_t total, int32_t* d_out, int64_t out_cap, int64_t ntiles);
void launch_docoffs(const Launch& L, const Workspace& w, const int64_t* d_offs, int64_t n_docs, int64_t total,
                    int64_t* d_out_offs);
void launch_corpus(hipStream_t s, int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int min_len, int max_len,
                   int64_t* d_offs, uint8_t* d_bytes, int64_t cap_bytes, int64_t* d_total);

}  // namespace tkz
