// tkz_shard.cpp -- token shard files in the C ABI (include/tkz.h, "token shard files").
//
// The reference returns a bare List<int> per text and has no batch or on-disk format, so there is nothing to be compatible
// with: a shard file is the packed (ids int32[], offsets int64[]) result of one rank's EncodeBatch written as is, so that a
// 100 M-document job streams every rank's result to its own file and a reader memory-maps it.  Layout (little-endian):
//
//   bytes 0..63   "TKZSHRD1" | u32 version = 1 | u32 id_bytes = 4 | i64 n_docs | i64 n_tokens | i64 doc_base | i64 token_base | 16 x 0
//   then          offsets int64[n_docs + 1]   token range of document d inside this shard (offsets[0] = 0)
//   then          ids     int32[n_tokens]
//
// doc_base / token_base come from the one all-gather of counts (tkz_shard_bases), so the files of different ranks concatenate
// into the global result with no further exchange.  tokenizer_amd/shardfile.py reads and writes the same layout.
#include "tkz_simt.h"     // the HIP runtime (or, in the tests' CPU build, its emulation)

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/tkz.h"

namespace tkz { tkz_status set_error(tkz_status s, const std::string& msg); }

namespace {

struct Header {
    char magic[8]; uint32_t version, id_bytes; int64_t n_docs, n_tokens, doc_base, token_base; char pad[16];
};
static_assert(sizeof(Header) == 64, "shard header is 64 bytes");

Header make_header(int64_t n_docs, int64_t n_tokens, int64_t doc_base, int64_t token_base) {
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "TKZSHRD1", 8);
    h.version = 1; h.id_bytes = 4; h.n_docs = n_docs; h.n_tokens = n_tokens; h.doc_base = doc_base; h.token_base = token_base;
    return h;
}

// running check of an offsets stream: starts at 0, never decreases, ends at n_tokens
struct OffsetCheck {
    int64_t prev = 0; bool first = true, ok = true;
    void feed(const int64_t* p, int64_t n) {
        for (int64_t i = 0; i < n; ++i) {
            if (first) { ok = ok && p[i] == 0; first = false; }
            else ok = ok && p[i] >= prev;
            prev = p[i];
        }
    }
};

struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
};

constexpr size_t kChunk = size_t(32) << 20;

}  // namespace

extern "C" {

tkz_status tkz_shard_write(const char* path, const int32_t* ids, int64_t n_tokens, const int64_t* offsets, int64_t n_docs,
                           int64_t doc_base, int64_t token_base) {
    if (!path || !offsets || n_docs < 0 || n_tokens < 0 || (n_tokens > 0 && !ids)) return tkz::set_error(TKZ_E_ARG, "bad shard arguments");
    OffsetCheck chk;
    chk.feed(offsets, n_docs + 1);
    if (!chk.ok || chk.prev != n_tokens) return tkz::set_error(TKZ_E_ARG, "offsets must start at 0, be non-decreasing and end at the number of ids");
    File F;
    F.f = fopen(path, "wb");
    if (!F.f) return tkz::set_error(TKZ_E_ARG, std::string("cannot open ") + path + " for writing");
    const Header h = make_header(n_docs, n_tokens, doc_base, token_base);
    bool ok = fwrite(&h, sizeof h, 1, F.f) == 1;
    ok = ok && fwrite(offsets, 8, (size_t)(n_docs + 1), F.f) == (size_t)(n_docs + 1);
    ok = ok && (n_tokens == 0 || fwrite(ids, 4, (size_t)n_tokens, F.f) == (size_t)n_tokens);
    ok = ok && fflush(F.f) == 0;
    if (!ok) return tkz::set_error(TKZ_E_DEVICE, std::string("short write to ") + path);
    return TKZ_OK;
}

tkz_status tkz_shard_write_device(const char* path, int32_t device, const int32_t* d_ids, int64_t n_tokens, const int64_t* d_offsets,
                                  int64_t n_docs, int64_t doc_base, int64_t token_base) {
    if (!path || !d_offsets || n_docs < 0 || n_tokens < 0 || (n_tokens > 0 && !d_ids)) return tkz::set_error(TKZ_E_ARG, "bad shard arguments");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return tkz::set_error(TKZ_E_NO_DEVICE, "hipSetDevice failed");
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    // two page-locked chunks: the copy of chunk k+1 (its own stream) runs while chunk k is written to the file
    void* stage[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    auto cleanup = [&] { for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipHostFree(stage[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); } if (st) (void)hipStreamDestroy(st); };
    bool hip_ok = hipStreamCreate(&st) == hipSuccess;
    for (int i = 0; i < 2 && hip_ok; ++i) hip_ok = hipHostMalloc(&stage[i], kChunk, 0) == hipSuccess && hipEventCreate(&ev[i]) == hipSuccess;
    if (!hip_ok) { cleanup(); return tkz::set_error(TKZ_E_OUT_OF_MEMORY, "page-locked staging for the shard writer could not be allocated"); }
    File F;
    F.f = fopen(path, "wb");
    if (!F.f) { cleanup(); return tkz::set_error(TKZ_E_ARG, std::string("cannot open ") + path + " for writing"); }
    const Header h = make_header(n_docs, n_tokens, doc_base, token_base);
    bool ok = fwrite(&h, sizeof h, 1, F.f) == 1;
    OffsetCheck chk;
    // the two arrays as one stream of (device pointer, bytes) segments cut into chunks
    struct Seg { const char* p; size_t bytes; bool offsets; } segs[2] = {{reinterpret_cast<const char*>(d_offsets), (size_t)(n_docs + 1) * 8, true},
                                                                         {reinterpret_cast<const char*>(d_ids), (size_t)n_tokens * 4, false}};
    struct Chunk { const char* p; size_t n; bool offsets; };
    auto chunk_at = [&](size_t k, Chunk* c) -> bool {      // k-th chunk of the stream
        for (const Seg& s : segs) {
            const size_t nc = (s.bytes + kChunk - 1) / kChunk;
            if (k < nc) { c->p = s.p + k * kChunk; c->n = s.bytes - k * kChunk < kChunk ? s.bytes - k * kChunk : kChunk; c->offsets = s.offsets; return true; }
            k -= nc;
        }
        return false;
    };
    Chunk cur, nxt;
    bool have = chunk_at(0, &cur);
    if (have) hip_ok = hipMemcpyAsync(stage[0], cur.p, cur.n, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[0], st) == hipSuccess;
    for (size_t k = 0; have && ok && hip_ok; ++k) {
        const bool more = chunk_at(k + 1, &nxt);
        if (more) hip_ok = hipMemcpyAsync(stage[(k + 1) & 1], nxt.p, nxt.n, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[(k + 1) & 1], st) == hipSuccess;
        hip_ok = hip_ok && hipEventSynchronize(ev[k & 1]) == hipSuccess;
        if (!hip_ok) break;
        if (cur.offsets) chk.feed(static_cast<const int64_t*>(stage[k & 1]), (int64_t)(cur.n / 8));
        ok = fwrite(stage[k & 1], 1, cur.n, F.f) == cur.n;
        have = more; cur = nxt;
    }
    ok = ok && fflush(F.f) == 0;
    (void)hipStreamSynchronize(st);
    cleanup();
    if (!hip_ok) return tkz::set_error(TKZ_E_DEVICE, "device-to-host copy failed while writing the shard");
    if (!ok) return tkz::set_error(TKZ_E_DEVICE, std::string("short write to ") + path);
    if (!chk.ok || chk.prev != n_tokens) { fclose(F.f); F.f = nullptr; remove(path); return tkz::set_error(TKZ_E_ARG, "offsets must start at 0, be non-decreasing and end at the number of ids"); }
    return TKZ_OK;
}

tkz_status tkz_shard_read_header(const char* path, int64_t* n_docs, int64_t* n_tokens, int64_t* doc_base, int64_t* token_base) {
    if (!path) return tkz::set_error(TKZ_E_ARG, "null path");
    File F;
    F.f = fopen(path, "rb");
    if (!F.f) return tkz::set_error(TKZ_E_ARG, std::string("cannot open ") + path);
    Header h;
    if (fread(&h, sizeof h, 1, F.f) != 1) return tkz::set_error(TKZ_E_FORMAT, "truncated shard header");
    if (memcmp(h.magic, "TKZSHRD1", 8) != 0 || h.version != 1 || h.id_bytes != 4 || h.n_docs < 0 || h.n_tokens < 0)
        return tkz::set_error(TKZ_E_FORMAT, "not a token shard file (magic / version)");
    if (n_docs) *n_docs = h.n_docs;
    if (n_tokens) *n_tokens = h.n_tokens;
    if (doc_base) *doc_base = h.doc_base;
    if (token_base) *token_base = h.token_base;
    return TKZ_OK;
}

}  // extern "C"
