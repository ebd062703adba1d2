// tkz_api.cpp -- the C ABI of libtkz (include/tkz.h): vocabulary objects, device table upload,
// workspace management and the launch sequence of one encode batch.  HIP only: there is no CPU path.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/tkz.h"
#include "tkz_bpe.h"
#include "tkz_corpus.h"
#include "tkz_kernels.h"
#include "tkz_pretok.h"
#include "tkz_sdma.h"
#include "tkz_vocab.h"

namespace {

thread_local std::string g_err;
#ifdef TKZ_DEVPROF
unsigned long long* g_devprof = nullptr;   // development builds only (make DEVPROF=1, env TKZ_DEV_ABLATE bit 4)
#endif

tkz_status fail(tkz_status s, const std::string& msg) { g_err = msg; return s; }

// The caller's current HIP device is left as it was found: every entry point that needs the encoder's device switches to it
// for the duration of the call only (a host may drive several GPUs, or run torch with another current device).
struct DeviceScope {
    int prev = -1;
    hipError_t enter(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur == dev) return hipSuccess;
        const hipError_t r = hipSetDevice(dev);
        if (r == hipSuccess) prev = cur;
        return r;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(TKZ_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// a grow-only device buffer
// TKZ_LOG_SLOW_MS=<n> in the environment: a batch call that takes longer than n ms on the host says so on stderr, with the time it spent in
// hipMalloc/hipFree (a fresh encoder's first large batch sizes its workspace: gigabytes from the driver, which on some boxes takes seconds)
thread_local int64_t g_alloc_ns = 0;
thread_local int g_alloc_calls = 0;
struct AllocClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocClock() { g_alloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++g_alloc_calls; }
};
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n, int64_t* accounted) {
        if (n <= cap) return hipSuccess;
        AllocClock clock;
        size_t want = std::max(n, cap + cap / 2);
        want = (want + 255) & ~size_t(255);
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess && want > n) { want = (n + 255) & ~size_t(255); e = hipMalloc(&q, want); }
        if (e != hipSuccess) return e;
        if (p) (void)hipFree(p);
        if (getenv("TKZ_LOG_ALLOC")) fprintf(stderr, "tkz alloc: %zu -> %zu bytes (asked %zu)\n", cap, want, n);
        *accounted += (int64_t)want - (int64_t)cap;
        p = q; cap = want;
        return hipSuccess;
    }
    void release() { if (p) { AllocClock clock; (void)hipFree(p); } p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// a part of another buffer (not owned)
struct DevView {
    void* p = nullptr;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct CounterBlock {          // mirrors the device block
    int32_t err; int32_t mneed /* longest miss list of a sub-tile (kErrMissCap) */; int32_t mhigh /* longest list above kMissCapMin that fitted */;
    int32_t over64 /* sub-tiles with more than 64 list entries (k_list_stats; grown lists only) */;
    int64_t grand;
    unsigned long long pool_head;
    int64_t ndocstarts;
    unsigned long long heavy_count;
    int64_t npieces;
    unsigned long long giant_ticket;       // k_giant_merge's work counter
    unsigned long long xcount, xcount2;    // (adjacent: launch_pretok_rows) blocks the o200k ASCII scanner left over, blocks the multi-byte one left over as well
    unsigned long long coop_count, coop_ticket;   // k_list_stats -> k_merge_coop: queued long misses of more than kLanePiece bytes, the next one to be taken
    unsigned long long long_log_count;     // a learning batch: records k_merge_long wanted to log (EncodeParams::long_log)
    int64_t lq_total;                      // long misses in the class queue (the scan of EncodeParams::lq_cnt)
    unsigned long long miss_short, miss_long;   // pieces of the batch that missed the key tables as a whole (k_list_stats: the sums of mcount)
};

}  // namespace

namespace tkz { tkz_status set_error(tkz_status s, const std::string& msg) { return fail(s, msg); } }   // (tkz_comm.cpp, tkz_decode.cpp)

struct tkz_vocab { tkz::Vocab v; };

// Everything one in-flight call needs besides the (read-only) tables: kernel workspace, staging for the host-buffer entry
// points, streams and profiling events.  An encoder keeps a pool of these; every entry point leases one for the duration of the
// call, so host threads sharing one encoder run concurrently, each on its own workspace and streams ("one encoder
// usable from many host threads") -- the reference's instance is likewise safe to share (its only shared mutable state, the LRU
// memo, is locked: LRUCache.cs:61,99).
struct Workspace {
    // kernel workspace
    DevBuf w_gq, w_gcnt, w_xq, w_startbits, w_tmp, w_dense, w_tcount, w_prank, w_pcount, w_pbase, w_tbase, w_bsum, w_doctok, w_dcount, w_dbase, w_pool;
    // what every batch starts from as zeros -- the counter block, the document-start bitmap, the per-sub-tile flags -- lives in ONE buffer, zeroed by ONE
    // memset (three launches were three dependent launches: a batch of a few megabytes is made of little else)
    DevBuf w_zero; size_t zero_bytes = 0;
    DevView w_counters, w_docbits, w_heavyq;
    DevBuf w_mlist, w_mquad, w_mcount, w_pextra, w_coopq, w_lqcnt, w_lqbase, w_lq;
    bool learn_window_start = false;       // this learning batch opens a window: the hit counters and the log start from zero
    DevBuf w_counts3;                      // {n_docs, n_bytes, n_tokens} of the batch this workspace is running (tkz_pending_counts_device)
    bool sized = false;                    // a batch has run to its end here: the lists and the record buffer have seen real text (encode_device: the sizing attempt)
    int32_t mcap = tkz::kMissCapMin;       // entries of a sub-tile's miss list; grows (once, to what the batch needed) when a sub-tile overflows it
    bool place128 = false;                 // a recent batch of this workspace had more than a fifth of its sub-tiles above 64 list entries: k_place<128>
    int low_lists = 0, low_place = 0;      // consecutive batches that would have done with shorter lists / with k_place<64> (hysteresis: kLowBatches)
    bool learning = false;                 // this workspace's batch counts memo hits per slot (TkzTables::memo_hits): the encoder promotes the hottest entries when it ends
    // staging for the host-buffer entry points (two sets: chunk k+1 is uploaded while chunk k is encoded and chunk k-1 downloaded)
    DevBuf s_bytes[2], s_offs[2], s_out[3], s_outoffs[3];      // staging of the host-buffer entry points: two input sets, three output sets (encode_host)
    // the UTF-16 batch entry point: code units, their document marks, per-tile / per-group lengths (two sets: the units of chunk k+1 are uploaded and
    // measured while chunk k is encoded), the UTF-8 batch they become
    struct U16Stage { DevBuf units, offs, docbits, grp, tsum, tbase, bsum, counters, boffs; struct Host { int32_t err; int32_t pad; int64_t grand; }* h = nullptr; } u16[2];
    DevBuf u_bytes[2];
    // Decode
    DevBuf d_grp, d_tsum, d_tbase, d_bsum, d_counters, d_ids, d_idoffs, d_out, d_outoffs;
    // piece-granular entry point: piece byte offsets, token offsets, first piece of every document
    DevBuf p_boffs, p_toffs, p_docp;
    CounterBlock* h_counters = nullptr;   // pinned
    // the single-launch path for small batches (k_small): input, output and status in ONE page-locked block the device reads and writes directly
    uint8_t* h_small = nullptr;
    bool fork_token = false;               // this workspace's call holds the process's one permission to use the side streams (g_fork_in_flight)
    int64_t forked_batches = 0;            // batches that ran the long pieces' kernels beside k_merge_short (tkz_encoder_side_by_side_batches)
    int64_t last_coop = 0;                 // ... and how many of its long misses were pieces of more than 128 bytes (a wavefront each: k_merge_coop)
    int64_t last_lq_total = -1;            // entries of the class queue of the long misses in the workspace's last batch on the batch path (-1: none yet)
    hipStream_t st_side = nullptr, st_side2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;   // large batches: k_merge_long_q and k_merge_coop run beside k_merge_short (launch_encode)
    hipStream_t st_small = nullptr;        // (non-blocking: a small call never waits for another thread's batch on the legacy default stream)
    std::atomic<int64_t> small_calls{0}, small_fallbacks{0};   // (read by tkz_encoder_small_path_calls from other threads)
    int64_t small_clocks[16] = {};         // the phase stamps of the last single-launch call, copied out after its synchronisation
    hipStream_t st_compute = nullptr, st_in = nullptr, st_out = nullptr;   // the host-buffer entry points: kernels / uploads / downloads
    hipEvent_t ev_in[2] = {}, ev_done[2] = {}, ev_out[3] = {};
    tkz::SdmaSignal sig_out[3], sig_outoffs[3];   // downloads on a copy engine of their own (tkz_sdma.h): the completion signal of each staging set
    int sdma_state = 0;                    // 0 not looked at, 1 in use, -1 not available: the runtime's hipMemcpyAsync
    std::atomic<int64_t> engine_downloads{0};      // copies of results that went by copy engine (tkz_encoder_engine_downloads)
    int64_t bytes_allocated = 0;
    bool busy = false;
    // profiling
    hipEvent_t ev[tkz::K_COUNT][2] = {};
    bool ev_used[tkz::K_COUNT] = {};
    double ms[tkz::K_COUNT] = {};
    int64_t launches[tkz::K_COUNT] = {};
    void release_all() {
        DevBuf* bufs[] = {&w_counts3, &w_mlist, &w_mquad, &w_mcount, &w_pextra, &w_coopq, &w_lqcnt, &w_lqbase, &w_lq, &w_gq, &w_gcnt, &w_xq, &w_zero, &w_startbits, &w_tmp, &w_dense, &w_tcount, &w_prank, &w_pcount, &w_pbase, &w_tbase, &w_bsum,
                          &w_doctok, &w_dcount, &w_dbase, &w_pool, &s_bytes[0], &s_bytes[1], &s_offs[0], &s_offs[1], &s_out[0], &s_out[1], &s_out[2],
                          &s_outoffs[0], &s_outoffs[1], &s_outoffs[2], &u_bytes[0], &u_bytes[1],
                          &u16[0].units, &u16[0].offs, &u16[0].docbits, &u16[0].grp, &u16[0].tsum, &u16[0].tbase, &u16[0].bsum, &u16[0].counters, &u16[0].boffs,
                          &u16[1].units, &u16[1].offs, &u16[1].docbits, &u16[1].grp, &u16[1].tsum, &u16[1].tbase, &u16[1].bsum, &u16[1].counters, &u16[1].boffs,
                          &d_grp, &d_tsum, &d_tbase, &d_bsum, &d_counters, &d_ids, &d_idoffs, &d_out, &d_outoffs, &p_boffs, &p_toffs, &p_docp};
        for (DevBuf* b : bufs) b->release();
        if (h_counters) (void)hipHostFree(h_counters);
        for (int q = 0; q < 2; ++q) if (u16[q].h) (void)hipHostFree(u16[q].h);
        if (h_small) (void)hipHostFree(h_small);
        if (st_small) (void)hipStreamDestroy(st_small);
        for (hipStream_t st : {st_side, st_side2}) if (st) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : {ev_fork, ev_join, ev_join2}) if (ev) (void)hipEventDestroy(ev);
        for (int k = 0; k < tkz::K_COUNT; ++k) for (int q = 0; q < 2; ++q) if (ev[k][q]) (void)hipEventDestroy(ev[k][q]);
        for (int q = 0; q < 2; ++q) { if (ev_in[q]) (void)hipEventDestroy(ev_in[q]); if (ev_done[q]) (void)hipEventDestroy(ev_done[q]); }
        for (int q = 0; q < 3; ++q) if (ev_out[q]) (void)hipEventDestroy(ev_out[q]);
        if (st_compute) (void)hipStreamDestroy(st_compute);
        if (st_in) (void)hipStreamDestroy(st_in);
        if (st_out) (void)hipStreamDestroy(st_out);
        for (int q = 0; q < 3; ++q) { tkz::sdma_signal_destroy(&sig_out[q]); tkz::sdma_signal_destroy(&sig_outoffs[q]); }
    }
};

constexpr int64_t kPromoSecondBytes = int64_t(1) << 30;      // the second learning round follows the first window by this much text (then 2, 4, 8 ... times: encode_device)

struct tkz_encoder {
    int device = 0;
    int pattern = 0;
    int max_key_len = 0;
    bool pretok_seq = false;
    bool profiling = false;
    std::mutex mu;                         // the workspace pool, the decoder table
    std::vector<Workspace*> pool;
    // device tables (read-only once built)
    DevBuf t_short, t_mid, t_long, t_blob, t_pair, t_byte, t_bpair, t_bmp, t_counts3, t_memo;
    uint32_t memo_slots = 0;               // the piece memo (tkz_tables.h); TKZ_OPT_PIECE_MEMO switches its use
    TkzTables T{};
    // Decode: id -> bytes (vocabulary keys + registered special tokens), rebuilt when the special tokens change
    DevBuf t_decoff, t_decblob, t_decids;
    TkzDecodeTable D{};
    std::vector<std::pair<int32_t, std::string>> dec_vocab, dec_special;   // host copies (id, bytes)
    int64_t bytes_allocated = 0;           // tables
    std::atomic<int64_t> last_xcount{0}, last_xcount2{0};   // tkz_encoder_pretok_leftovers
    bool small_ok = false;                 // the device's LDS per workgroup holds k_small's (kSmallLdsBytesNeeded)
    bool piece_stats = false;              // TKZ_OPT_PIECE_STATS
    int64_t latency_bytes = [] { const char* v = getenv("TKZ_LATENCY_BYTES"); return v ? (int64_t)atoll(v) : int64_t(16) << 20; }();   // TKZ_OPT_LATENCY_BYTES
    bool case_equiv = false;               // TKZ_OPT_CASE_EQUIVALENCE: `'` + U+017F is a contraction under cl100k (a .NET >= 7 host)
    size_t bmp_image_bytes = 0;            // the class table image on the device (t_bmp)
    DevBuf t_stats;                        // its device block (EncodeParams::stats)
    int64_t stat_batches = 0, stat_giants = 0;   // ... and what the host adds per batch (under mu)
    int pending = 0;                       // tkz_pending handles outstanding (under mu)
    bool destroyed = false;                // tkz_encoder_destroy was called while handles were outstanding: the last _end frees the encoder
    // ---- promoted pieces (tkz_tables.h): hot memo entries moved into the SHORT / MID tables themselves.  All under mu. ----
    int promo_mode = 1;                    // TKZ_OPT_PROMOTE: 0 never on its own, 1 automatic (default)
    int promo_rounds = 0;                  // automatic promotions so far
    int64_t promo_min_bytes = int64_t(8) << 20;   // a batch of at least this many bytes may be a learning batch (TKZ_OPT_PROMOTE_MIN_BYTES)
    size_t promo_cap = 65536;              // promoted pieces the key tables hold at most (TKZ_OPT_PROMOTE_CAP)
    bool learning = false;                 // a batch that counts memo hits is in flight (one at a time)
    int64_t bytes_seen = 0, bytes_at_promo = 0;   // bytes the batch path has encoded; ... when the last promotion happened
    DevBuf t_memo_hits, t_promo;           // the hit counters of a learning batch; the token quads of the promoted pieces
    DevBuf t_long_log;                     // ... and its log of merged pieces of 17..28 bytes (EncodeParams::long_log)
    int64_t long_log_n = 0;                // records the last learning batch left there (set when it ended)
    std::vector<DevBuf> retired;           // table images replaced while other calls may still have been probing them: freed when no call of the encoder is in flight (Lease)
    // ---- the cache ADAPTS (round 6; TKZ_OPT_ADAPT, all under mu).  The reference's LRUCache evicts and refills for ever (LRUCache.cs:79-121); here the share
    // of pieces that miss the key tables as a whole is followed from batch to batch (k_list_stats sums the miss lists: no extra kernel), and when it leaves
    // the level it had after the last promotion the encoder LEARNS AGAIN: promotions dropped, memo emptied, the next batches count hits, the hottest pieces of
    // the text as it is NOW are promoted (pieces that stopped hitting are simply not chosen again).
    int adapt = 1;                         // TKZ_OPT_ADAPT
    // (how much text the miss share is averaged over / has to settle for after a promotion, and how soon after one the encoder may learn again; the
    //  environment variables are the tests' handle on them: their batches are kilobytes)
    int64_t adapt_settle_bytes = [] { const char* v = getenv("TKZ_ADAPT_SETTLE_BYTES"); return v && atoll(v) > 0 ? (int64_t)atoll(v) : int64_t(64) << 20; }();
    int64_t adapt_round_bytes = [] { const char* v = getenv("TKZ_ADAPT_ROUND_BYTES"); return v && atoll(v) > 0 ? (int64_t)atoll(v) : kPromoSecondBytes; }();     // (the tests' handle on the rounds' spacing)
    int64_t adapt_min_bytes = [] { const char* v = getenv("TKZ_ADAPT_MIN_BYTES"); return v && atoll(v) > 0 ? (int64_t)atoll(v) : int64_t(256) << 20; }();
    double ew_miss = 0, base_miss = 0;     // miss share of the recent batches (weighted by their bytes, 64 MB time constant); ... as it settled after the last promotion
    bool ew_valid = false, base_valid = false;
    double win_miss = 0, win_pieces = 0;   // misses and pieces of the learning window so far ...
    double window_miss = 0;                // ... and the miss share of the window the LAST install was learnt in (promotions only lower it on unchanged text)
    bool window_valid = false;
    double last_window_miss = 0;           // the same, kept for the NEXT window to be compared with
    bool last_window_valid = false;
    int64_t bytes_at_install = 0;          // bytes_seen when the key tables were last replaced
    int64_t learn_bytes = 0;               // bytes of the learning window so far (batches smaller than promo_min_bytes add up to one)
    bool memo_clear_pending = false;       // the memo is emptied before the next learning window starts (only while no other call is in flight)
    int64_t n_promotions = 0, n_relearns = 0;
    std::vector<tkz::KeyItem> promo_items; // promoted piece -> promo code, in order of promotion
    std::unordered_set<std::string> promo_keys;
    std::vector<uint32_t> promo_quads;     // 4 tokens per promoted piece (host copy of t_promo)
    uint32_t short_slots_n = 0, mid_slots_n = 0;
    // an automatic promotion runs behind the batch that gathered its statistics (the copy of the memo back to the host, the choice and the rebuilt key
    // tables are ~50 ms of host work: not something the call that happened to be the learning batch should wait for).  `learning` stays set until it
    // is done, so there is one at a time; joined by join_promotion()
    std::thread promo_thread;
    std::mutex promo_join_mu;
};

namespace {

// ONE batch at a time in the whole process runs kernels on side streams behind events (launch_encode's forked form): two such batches at once are six streams with
// waits on one another's events, on a runtime that maps streams onto a handful of hardware queues -- a call that finds the permission taken keeps the serial form
std::atomic<int> g_fork_in_flight{0};

// a workspace of the encoder's pool for the duration of one call
struct Lease {
    tkz_encoder* e; Workspace* ws = nullptr;
    explicit Lease(tkz_encoder* enc) : e(enc) {
        std::lock_guard<std::mutex> lock(e->mu);
        for (Workspace* w : e->pool) if (!w->busy) { ws = w; break; }
        if (!ws) { ws = new Workspace(); e->pool.push_back(ws); }
        ws->busy = true;
    }
    ~Lease() {
        std::lock_guard<std::mutex> lock(e->mu);
        if (ws->fork_token) { ws->fork_token = false; g_fork_in_flight.store(0); }
        ws->busy = false;
        // table images a promotion replaced: every call takes its copy of the table descriptor while it holds a workspace, so with no workspace leased
        // nothing can be probing them any more
        if (!e->retired.empty()) {
            for (Workspace* w : e->pool) if (w->busy) return;
            for (DevBuf& b : e->retired) { e->bytes_allocated -= (int64_t)b.cap; b.release(); }
            e->retired.clear();
        }
    }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
};

void prof_hook(void* ctx, int id, int phase, hipStream_t s) {
    Workspace* e = static_cast<Workspace*>(ctx);
    if (!e->ev[id][phase]) (void)hipEventCreate(&e->ev[id][phase]);
    (void)hipEventRecord(e->ev[id][phase], s);
    if (phase == 1) e->ev_used[id] = true;
}
void prof_collect(Workspace* e) {
    for (int k = 0; k < tkz::K_COUNT; ++k) {
        if (!e->ev_used[k]) continue;
        float t = 0;
        if (hipEventElapsedTime(&t, e->ev[k][0], e->ev[k][1]) == hipSuccess) { e->ms[k] += t; e->launches[k] += 1; }
        e->ev_used[k] = false;
    }
}

template <class T>
hipError_t upload(DevBuf& b, const std::vector<T>& v, int64_t* acc) {
    hipError_t e = b.ensure(std::max<size_t>(16, v.size() * sizeof(T)), acc);
    if (e != hipSuccess) return e;
    return hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

// (re)builds the device decoder table from the vocabulary keys and the registered special tokens.  The reference looks an id
// up in Decoder first, then in SpecialTokensDecoder (TikTokenizer.cs:591-598): a special token never shadows a vocabulary id.
tkz_status build_decode_table(tkz_encoder* e) {
    std::vector<std::pair<int32_t, const std::string*>> ent;
    ent.reserve(e->dec_vocab.size() + e->dec_special.size());
    for (const auto& kv : e->dec_vocab) ent.emplace_back(kv.first, &kv.second);
    const size_t nv = ent.size();
    for (const auto& kv : e->dec_special) ent.emplace_back(kv.first, &kv.second);
    // stable by id: of two entries with one id the vocabulary's (first) wins
    std::vector<size_t> order(ent.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ent[a].first < ent[b].first; });
    (void)nv;
    int64_t max_id = -1;
    for (const auto& kv : ent) max_id = std::max<int64_t>(max_id, kv.first);
    const bool dense = max_id < (int64_t(1) << 22);
    std::vector<uint32_t> off;
    std::vector<int32_t> ids;
    std::vector<uint8_t> blob;
    if (dense) {
        off.assign((size_t)(max_id + 2), 0);
        std::vector<const std::string*> by_id((size_t)(max_id + 1), nullptr);
        for (size_t k : order) if (ent[k].first >= 0 && !by_id[(size_t)ent[k].first]) by_id[(size_t)ent[k].first] = ent[k].second;
        for (int64_t i = 0; i <= max_id; ++i) {
            off[(size_t)i] = (uint32_t)blob.size();
            if (by_id[(size_t)i]) blob.insert(blob.end(), by_id[(size_t)i]->begin(), by_id[(size_t)i]->end());
        }
        off[(size_t)(max_id + 1)] = (uint32_t)blob.size();
    } else {
        int64_t last = INT64_MIN;
        for (size_t k : order) {
            if (ent[k].first == last) continue;
            last = ent[k].first;
            ids.push_back(ent[k].first);
            off.push_back((uint32_t)blob.size());
            blob.insert(blob.end(), ent[k].second->begin(), ent[k].second->end());
        }
        off.push_back((uint32_t)blob.size());
    }
    blob.resize(blob.size() + 16, 0);
    int64_t* acc = &e->bytes_allocated;
    hipError_t h = upload(e->t_decoff, off, acc);
    if (h == hipSuccess) h = upload(e->t_decblob, blob, acc);
    if (h == hipSuccess && !dense) h = upload(e->t_decids, ids, acc);
    if (h != hipSuccess) return fail(TKZ_E_DEVICE, std::string("decoder table upload: ") + hipGetErrorString(h));
    e->D.off = e->t_decoff.as<uint32_t>(); e->D.blob = e->t_decblob.as<uint8_t>(); e->D.ids = dense ? nullptr : e->t_decids.as<int32_t>();
    e->D.n = dense ? max_id + 1 : (int64_t)ids.size(); e->D.dense = dense ? 1 : 0;
    return TKZ_OK;
}

// ---- promoted pieces ------------------------------------------------------------------------------------------------------------------------
// The reference's LRUCache (TikTokenizer.cs:254,270) answers a piece it has seen before without running BytePairEncode again; the device memo does
// the same inside k_merge_short -- at the price of a list entry, a quad, a 32-byte slot and an answer per missed piece and batch.  Under a
// vocabulary that has not seen the text 7 of 8 short misses are such hits and that kernel is 40 % of the step.  A memo answer never changes, so the
// hottest ones are PROMOTED: the host reads the memo (and, after a learning batch, the sampled hit count of every slot) back, adds the pieces to the
// SHORT / MID key tables with a promo code in place of a rank (tkz_tables.h), and uploads the new images; k_probe then finds such a piece like any
// key, the merge kernels never see it, k_place gathers its <= 4 tokens.  Results are the same ids by construction (the memo's answers are exact and a
// slot is read back under the same validity rule the kernels use); the tests compare a promoted encoder with the oracle.
constexpr int64_t kLongLogCap = 65536;             // records of merged 17..28-byte pieces a learning batch may log (EncodeParams::long_log)
constexpr int kPromoAutoRounds = 2;                // automatic promotions: the first batch of >= kPromoMinBytes, and one more after kPromoSecondBytes more

// (re)builds the SHORT / MID images from the vocabulary's keys + the promoted pieces and publishes them; `retire`: other calls may be probing the
// current images (they are kept until the encoder is destroyed), else they are freed
struct KeyTablesImage { DevBuf nt, np; size_t short_bytes = 0, n_short_slots = 0, n_mid_slots = 0, n_promo = 0; uint32_t sseed = 0, mseed = 0; };
// the images on the device, built from the vocabulary's keys + `promo_items` (copies: no lock is held here, the build takes tens of milliseconds)
tkz_status build_key_tables_image(tkz_encoder* e, const std::vector<tkz::KeyItem>& promo_items, const std::vector<uint32_t>& promo_quads, KeyTablesImage* img) {
    std::vector<tkz::KeyItem> items;
    items.reserve(e->dec_vocab.size() + promo_items.size());
    {   // (dec_vocab's vocabulary part never changes after tkz_encoder_create)
        std::vector<size_t> order(e->dec_vocab.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return e->dec_vocab[a].first < e->dec_vocab[b].first; });      // rank order: most frequent first
        for (size_t i : order) { const std::string& k = e->dec_vocab[i].second; if (!k.empty() && k.size() <= TKZ_MID_KEY_MAX) items.push_back(tkz::KeyItem{k, (uint32_t)e->dec_vocab[i].first}); }
    }
    for (const tkz::KeyItem& it : promo_items) items.push_back(it);
    std::vector<TkzShortSlot> ss; std::vector<TkzMidSlot> ms;
    tkz::build_key_tables(items, &ss, &img->sseed, &ms, &img->mseed);
    const size_t short_bytes = ss.size() * sizeof(TkzShortSlot), mid_bytes = ms.size() * sizeof(TkzMidSlot);
    int64_t acc = 0;
    // (uploads on a stream of their own: on the null stream they would queue behind the batches the host has in flight there)
    hipStream_t cs = nullptr;
    hipError_t h = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (h == hipSuccess) h = img->nt.ensure(std::max<size_t>(64, short_bytes + mid_bytes), &acc);
    if (h == hipSuccess && short_bytes) h = hipMemcpyAsync(img->nt.p, ss.data(), short_bytes, hipMemcpyHostToDevice, cs);
    if (h == hipSuccess && mid_bytes) h = hipMemcpyAsync(static_cast<char*>(img->nt.p) + short_bytes, ms.data(), mid_bytes, hipMemcpyHostToDevice, cs);
    if (h == hipSuccess) h = img->np.ensure(std::max<size_t>(64, promo_quads.size() * 4), &acc);
    if (h == hipSuccess && !promo_quads.empty()) h = hipMemcpyAsync(img->np.p, promo_quads.data(), promo_quads.size() * 4, hipMemcpyHostToDevice, cs);
    if (h == hipSuccess) h = hipStreamSynchronize(cs);
    if (cs) (void)hipStreamDestroy(cs);
    if (h != hipSuccess) { img->nt.release(); img->np.release(); return fail(TKZ_E_DEVICE, std::string("promoted tables: ") + hipGetErrorString(h)); }
    img->short_bytes = short_bytes; img->n_short_slots = ss.size(); img->n_mid_slots = ms.size(); img->n_promo = promo_items.size();
    return TKZ_OK;
}
// ... and put in place (e->mu held).  `retire`: other calls may be probing the current images (kept until the encoder is destroyed), else they are freed
void install_key_tables(tkz_encoder* e, KeyTablesImage& img, bool retire) {
    int64_t* acc = &e->bytes_allocated;
    *acc += (int64_t)(img.nt.cap + img.np.cap);
    if (retire) { e->retired.push_back(e->t_short); e->retired.push_back(e->t_promo); }
    else { *acc -= (int64_t)(e->t_short.cap + e->t_promo.cap); e->t_short.release(); e->t_promo.release(); }
    e->t_short = img.nt; e->t_promo = img.np;
    e->T.short_slots = e->t_short.as<TkzShortSlot>(); e->T.short_nb = (uint32_t)(img.n_short_slots / 2); e->T.short_seed = img.sseed;
    e->T.mid_slots = reinterpret_cast<const TkzMidSlot*>(e->t_short.as<char>() + img.short_bytes); e->T.mid_ns = (uint32_t)img.n_mid_slots; e->T.mid_seed = img.mseed;
    e->T.promo = img.n_promo ? e->t_promo.as<uint4>() : nullptr; e->T.promo_n = (uint32_t)img.n_promo;
    e->short_slots_n = (uint32_t)img.n_short_slots; e->mid_slots_n = (uint32_t)img.n_mid_slots;
}

// waits for the automatic promotion in the background, if any (never with e->mu held: the promotion takes it)
void join_promotion(tkz_encoder* e) {
    std::lock_guard<std::mutex> lock(e->promo_join_mu);
    if (e->promo_thread.joinable()) e->promo_thread.join();
}
// The memo (and the hit counters of a learning batch, or null: every valid entry counts alike) is read back and its hottest entries are promoted.
// Called with no lock held; takes e->mu for the bookkeeping and the publication.  *added: entries promoted by this call.
tkz_status promote_from_memo(tkz_encoder* e, bool use_hits, bool retire, int64_t* added) {
    using tkz::kLongLogDwords; using tkz::kLongLogMaxLen;
    if (added) *added = 0;
    if (!e->memo_slots || e->T.max_rank >= (int32_t)kPromoFlag) return TKZ_OK;      // (a promo code must not look like a rank)
    std::vector<TkzMemoSlot> memo(e->memo_slots);
    std::vector<uint32_t> hits;
    // (a stream of its own: this may be the background thread of an automatic promotion, and a copy on the null stream would wait for -- and hold up --
    //  whatever the host has queued there)
    hipStream_t cs = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } stream_guard{cs};
    HIP_TRY(hipMemcpyAsync(memo.data(), e->t_memo.p, memo.size() * sizeof(TkzMemoSlot), hipMemcpyDeviceToHost, cs));
    if (use_hits) { hits.resize(e->memo_slots); HIP_TRY(hipMemcpyAsync(hits.data(), e->t_memo_hits.p, hits.size() * 4, hipMemcpyDeviceToHost, cs)); }
    HIP_TRY(hipStreamSynchronize(cs));
    // ... and the pieces of 17..28 bytes k_merge_long logged during the learning batch (each with its <= 4 tokens): those that were logged at least
    // twice -- real source text is full of them: "\n" + 19 spaces, by the hundred thousand -- go into the MID key table the same way
    std::vector<uint32_t> llog;
    if (use_hits && e->t_long_log.p) {      // (the log's record count lives behind the records: it runs on through the batches of a learning window)
        unsigned long long n = 0;
        HIP_TRY(hipMemcpyAsync(&n, e->t_long_log.as<char>() + (size_t)kLongLogCap * kLongLogDwords * 4, 8, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipStreamSynchronize(cs));
        e->long_log_n = (int64_t)std::min<unsigned long long>(n, (unsigned long long)kLongLogCap);
    }
    if (use_hits && e->long_log_n > 0) {
        llog.resize((size_t)e->long_log_n * kLongLogDwords);
        HIP_TRY(hipMemcpyAsync(llog.data(), e->t_long_log.p, llog.size() * 4, hipMemcpyDeviceToHost, cs));
        HIP_TRY(hipStreamSynchronize(cs));
    }
    std::vector<tkz::KeyItem> items_copy;
    std::vector<uint32_t> quads_copy;
    int64_t n_new = 0;
    {
    std::lock_guard<std::mutex> lock(e->mu);
    const size_t cap = e->promo_cap;
    if (e->promo_items.size() >= cap) return TKZ_OK;
    if (!llog.empty()) {
        struct LongCand { uint32_t count; uint32_t rec; };
        std::unordered_map<std::string, LongCand> seen;
        for (int64_t r = 0; r < e->long_log_n; ++r) {
            const uint32_t* rec = &llog[(size_t)r * kLongLogDwords];
            const uint32_t len = rec[7] & 0xFFu, cnt = (rec[7] >> 8) & 0xFFu;
            if (len <= 16 || len > (uint32_t)kLongLogMaxLen || cnt < 1 || cnt > 4) continue;
            std::string key(len, '\0');
            for (uint32_t b = 0; b < len; ++b) key[b] = (char)((rec[b >> 2] >> (8 * (b & 3))) & 0xFFu);
            auto it = seen.find(key);
            if (it == seen.end()) seen.emplace(key, LongCand{1u, (uint32_t)r}); else ++it->second.count;
        }
        std::vector<std::pair<uint32_t, const std::string*>> order;
        for (const auto& kv : seen) if (kv.second.count >= 2 && !e->promo_keys.count(kv.first)) order.emplace_back(kv.second.count, &kv.first);
        std::sort(order.begin(), order.end(), [](const std::pair<uint32_t, const std::string*>& a, const std::pair<uint32_t, const std::string*>& b) { return a.first != b.first ? a.first > b.first : *a.second < *b.second; });
        const size_t long_room = std::min(cap - e->promo_items.size(), std::max<size_t>(cap / 4, 1));
        for (size_t i = 0; i < order.size() && (size_t)n_new < long_room; ++i) {
            const std::string& key = *order[i].second;
            const uint32_t* rec = &llog[(size_t)seen[key].rec * kLongLogDwords];
            const uint32_t cnt = (rec[7] >> 8) & 0xFFu;
            e->promo_keys.insert(key);
            const uint32_t index = (uint32_t)e->promo_items.size();
            e->promo_items.push_back(tkz::KeyItem{key, kPromoFlag | ((cnt - 1u) << kPromoCntShift) | index});
            for (uint32_t t = 0; t < 4; ++t) e->promo_quads.push_back(t < cnt ? (rec[8 + t] & 0x07FFFFFFu) : 0u);
            ++n_new;
        }
    }
    struct Cand { uint32_t hits, slot; };
    std::vector<Cand> cand;
    uint32_t n_valid = 0;
    for (uint32_t i = 0; i < e->memo_slots; ++i) {
        const uint32_t* v = memo[i].v;
        // the kernels' validity rule: the valid tag in every value word, not the BUSY mark; and a complete key: no zero byte inside its length, nothing
        // but zero bytes beyond it (other calls may be inserting while this copy was taken)
        if (!((v[0] & v[1] & v[2] & v[3]) & kMemoValid) || v[0] == kMemoBusy) continue;
        ++n_valid;
        if (use_hits && hits[i] == 0) continue;
        const uint32_t len = ((v[1] >> 27) & 15u) + 1u;
        bool ok = true;
        for (uint32_t b = 0; b < 16 && ok; ++b) { const uint32_t byte = (memo[i].k[b >> 2] >> (8 * (b & 3))) & 0xFFu; ok = b < len ? byte != 0 : byte == 0; }
        if (!ok) continue;
        cand.push_back(Cand{use_hits ? hits[i] : 1u, i});
    }
    // A memo three quarters full takes hardly any new piece (an entry is never replaced; a bucket has two ways): text with many one-off pieces fills it within a
    // gigabyte, and whatever the text turns into afterwards finds it closed -- the reference's LRUCache would have evicted (LRUCache.cs:79-88).  The next learning
    // window therefore starts on an EMPTY memo (memo_clear_pending: cleared under the lock while no other call is in flight, as after a drift): the hot pieces are
    // back within the first megabytes of that batch, the one-off ones are gone.  (tools/adapt_probe.py ... 1: source text behind 3 GB of synthetic text, the change
    // inside the first promotion's build: 95 GB/s and 2 k new pieces a round with the full memo, against a fresh encoder's 118 and 9 k.)
    if (use_hits && e->adapt && (uint64_t)n_valid * 4 >= (uint64_t)e->memo_slots * 3) e->memo_clear_pending = true;
    std::stable_sort(cand.begin(), cand.end(), [](const Cand& a, const Cand& b) { return a.hits > b.hits; });
    for (const Cand& c : cand) {
        if (e->promo_items.size() >= cap) break;
        const TkzMemoSlot& m = memo[c.slot];
        const uint32_t len = ((m.v[1] >> 27) & 15u) + 1u, cnt = ((m.v[0] >> 29) & 3u) + 1u;
        std::string key(len, '\0');
        for (uint32_t b = 0; b < len; ++b) key[b] = (char)((m.k[b >> 2] >> (8 * (b & 3))) & 0xFFu);
        if (!e->promo_keys.insert(key).second) continue;                        // (promoted before: a stale memo entry)
        const uint32_t index = (uint32_t)e->promo_items.size();
        e->promo_items.push_back(tkz::KeyItem{key, kPromoFlag | ((cnt - 1u) << kPromoCntShift) | index});
        for (int t = 0; t < 4; ++t) e->promo_quads.push_back((uint32_t)t < cnt ? (m.v[t] & 0x07FFFFFFu) : 0u);
        ++n_new;
    }
    if (added) *added = n_new;
    if (!n_new) return TKZ_OK;
    items_copy = e->promo_items; quads_copy = e->promo_quads;
    }
    // the key tables with the promoted pieces in them: built and uploaded WITHOUT the lock (tens of milliseconds), then put in place under it (only one
    // promotion runs at a time -- the encoder's one learning slot, or an idle encoder --, so the list has not changed in between)
    KeyTablesImage img;
    const tkz_status bs = build_key_tables_image(e, items_copy, quads_copy, &img);
    if (bs != TKZ_OK) return bs;
    std::lock_guard<std::mutex> lock(e->mu);
    install_key_tables(e, img, retire);
    return TKZ_OK;
}

// every promotion dropped: the key tables as the vocabulary alone gives them.  `retire`: see install_key_tables.  No lock held on entry.
tkz_status drop_promotions(tkz_encoder* e, bool retire) {
    {
        std::lock_guard<std::mutex> lock(e->mu);
        if (e->promo_items.empty()) return TKZ_OK;
    }
    KeyTablesImage img;
    const tkz_status st = build_key_tables_image(e, {}, {}, &img);
    if (st != TKZ_OK) return st;
    std::lock_guard<std::mutex> lock(e->mu);
    e->promo_items.clear(); e->promo_keys.clear(); e->promo_quads.clear();
    install_key_tables(e, img, retire);
    return TKZ_OK;
}

// TKZ_OPT_ADAPT (e->mu held): a batch that was not a learning batch has ended; `misses` of its `pieces` regex matches were not found in the key tables
// as a whole (vocabulary + promoted pieces).  Returns true when the encoder should learn again: the share of such pieces, averaged over the recent
// batches by their bytes, has left the level at which it settled after the last promotion by more than a quarter (and a percentage point) either way
// -- text whose pieces the promotions no longer answer, or text that a fresh encoder would answer better.  Not within adapt_min_bytes (256 MB) of the last
// change of the tables: a re-learn costs one window at the speed of an encoder without promotions and two table builds on the host.
bool adapt_after_batch(tkz_encoder* e, int64_t total, double misses, double pieces) {
    if (!e->adapt || e->promo_mode != 1 || pieces < 1) return false;
    const int64_t kAdaptSettleBytes = e->adapt_settle_bytes, kAdaptMinBytes = e->adapt_min_bytes;
    const double rate = misses / pieces, w = std::min(1.0, (double)total / (double)kAdaptSettleBytes);
    e->ew_miss = e->ew_valid ? e->ew_miss + (rate - e->ew_miss) * w : rate;
    e->ew_valid = true;
    if (e->learning || e->promo_rounds < 1) return false;
    const int64_t since = e->bytes_seen - e->bytes_at_install;
    if (!e->base_valid) {
        if (since >= kAdaptSettleBytes) {
            e->base_miss = e->ew_miss; e->base_valid = true;
            // A change of text that falls between a learning window and the install of what it learnt -- a promotion is built on a host thread, tens of
            // milliseconds, gigabytes of text at this rate -- has no settled level to leave: the level settles on the new text.  But an install only ever ADDS
            // pieces, so on the text it was learnt from the level it leaves is at or below the WINDOW's own miss share; one that is a quarter (and a percentage
            // point) ABOVE it means those promotions answer another text: a drift.  (bench.py's drift leg, 3 GB of synthetic text in 15 ms and real text
            //  behind it: the first install landed ten real batches later, no drift was ever seen, and the encoder ran at 0.77 of a fresh one.)
            const bool drifted = e->window_valid && e->base_miss > e->window_miss * 1.25 + 0.01;
            e->window_valid = false;
            if (drifted) return true;
        }
        return false;
    }
    if (since < kAdaptMinBytes) return false;
    if (e->promo_items.size() * 10 >= e->promo_cap * 9) return true;     // the list is (nearly) full of what the rounds have added: start over from the text as it is now
    return e->ew_miss > e->base_miss * 1.25 + 0.01 || e->ew_miss < e->base_miss * 0.75 - 0.01;
}

// workspace of one batch of `total` bytes / n_docs documents (grow-only buffers: nothing happens once they are large enough)
tkz_status prepare_workspace(Workspace* ws, int64_t total, int64_t n_docs, bool bitmap_only, bool pieces) {
    using namespace tkz;
    const int64_t nwords = total / 64 + 1;
    const int64_t ntiles = (total + kSub - 1) / kSub;
    const int64_t nblk = (ntiles + kScanBlock - 1) / kScanBlock;
    int64_t* acc = &ws->bytes_allocated;
    {   // [counter block, 256 B] [document-start bits, nwords + 8 words] [a flag per sub-tile, ntiles + 64 bytes]: zeroed together
        static_assert(sizeof(CounterBlock) <= 256, "the counter block's place in the zero region");
        const size_t off_bits = 256, off_flags = off_bits + (size_t)(nwords + 8) * 8, end = off_flags + (size_t)ntiles + 64;
        HIP_TRY(ws->w_zero.ensure(end, acc));
        ws->w_counters.p = ws->w_zero.p; ws->w_docbits.p = ws->w_zero.as<char>() + off_bits; ws->w_heavyq.p = ws->w_zero.as<char>() + off_flags;
        ws->zero_bytes = bitmap_only ? off_flags : end;
    }
    HIP_TRY(ws->w_startbits.ensure((size_t)(nwords + 8) * 8, acc));
    HIP_TRY(ws->w_counts3.ensure(32, acc));
    if (!bitmap_only) {
        HIP_TRY(ws->w_tmp.ensure((size_t)(total + 64) * 4, acc));
        HIP_TRY(ws->w_dense.ensure((size_t)(ntiles / kMergeGroup + 1) * kDenseCap * 4, acc));
        HIP_TRY(ws->w_tcount.ensure((size_t)ntiles * 4, acc));
        HIP_TRY(ws->w_pcount.ensure((size_t)ntiles * 4, acc));
        HIP_TRY(ws->w_pbase.ensure((size_t)ntiles * 8, acc));
        // one 4-byte record per piece.  The number of pieces is known only after the pre-tokenizer has run (English/code text: a
        // piece per ~4.5 bytes; the bound is a piece per byte): the buffer starts at a piece per 3 bytes, k_probe refuses to write
        // past it, and the batch is redone once with the exact size if that was not enough
        // (+ 8 per sub-tile on average: every sub-tile's records start on a 64-byte line of their own)
        HIP_TRY(ws->w_prank.ensure((size_t)(total / 3 + 8 * ntiles + 4096) * 4, acc));
        HIP_TRY(ws->w_mlist.ensure((size_t)ntiles * (size_t)ws->mcap * 4, acc));
        HIP_TRY(ws->w_mquad.ensure((size_t)ntiles * (size_t)ws->mcap * 16, acc));
        HIP_TRY(ws->w_mcount.ensure((size_t)ntiles * 4, acc));
        HIP_TRY(ws->w_pextra.ensure((size_t)ntiles * 4, acc));
        HIP_TRY(ws->w_coopq.ensure((size_t)(total / kLanePiece + 64) * 8, acc));
        {   // the class queue of the long misses (a long miss is 17 bytes at least): counts and bases per (class, chunk of 64 sub-tiles), 8 bytes an entry
            const int64_t nchunks = (ntiles + 63) / 64;
            HIP_TRY(ws->w_lqcnt.ensure((size_t)nchunks * 16 * 4, acc));
            HIP_TRY(ws->w_lqbase.ensure((size_t)(nchunks * 16 + 1) * 8, acc));
            HIP_TRY(ws->w_lq.ensure((size_t)(total / (kShortMax + 1) + 64) * 8, acc));
        }
        HIP_TRY(ws->w_tbase.ensure((size_t)(ntiles + 1) * 8, acc));
        HIP_TRY(ws->w_bsum.ensure((size_t)(nblk + 1) * 8, acc));
        HIP_TRY(ws->w_doctok.ensure((size_t)((pieces ? total : n_docs) + 2) * 4, acc));    // (piece mode: one entry per piece)
        HIP_TRY(ws->w_dcount.ensure((size_t)ntiles * 4, acc));
        HIP_TRY(ws->w_dbase.ensure((size_t)ntiles * 8, acc));
        HIP_TRY(ws->w_gq.ensure((size_t)(total / kArenaPiece + 2) * 24, acc));     // {position, length} per giant piece + the order they are taken in
        HIP_TRY(ws->w_gcnt.ensure((size_t)ntiles * 4, acc));
        if (!ws->w_pool.p) HIP_TRY(ws->w_pool.ensure((size_t)std::min<int64_t>(24 * total + 4096, int64_t(64) << 20), acc));
    }
    return TKZ_OK;
}

// where the piece-granular entry point wants its arrays (all on the device)
struct PiecesOut { int64_t* piece_boffs; int64_t* piece_toffs; int64_t* doc_piece; int64_t piece_cap; int64_t n_pieces; };

// the batch on the device; when pretok == false every "document" is taken as one piece
struct SlowCallLog {
    int64_t total; int attempts = 0; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    SlowCallLog(int64_t n) : total(n) { g_alloc_ns = 0; g_alloc_calls = 0; }
    ~SlowCallLog() {
        const char* v = getenv("TKZ_LOG_SLOW_MS");
        const long limit_ms = v ? atol(v) : -1L;
        if (limit_ms < 0) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > (double)limit_ms)
            { fprintf(stderr, "tkz: batch call of %lld bytes: %.1f ms on the host, %.1f ms of it in %d hipMalloc/hipFree calls, %d attempt(s)\n", (long long)total, ms, (double)g_alloc_ns * 1e-6, g_alloc_calls, attempts); fflush(stderr); }
    }
};
enum { kCallWhole = 0, kCallBegin = 1, kCallEnd = 2 };
// the caller's page-locked text and offsets as the device sees them: encode_device fetches them itself (k_ingest) into d_bytes / d_offs
struct IngestSrc { const uint8_t* h_bytes; const int64_t* h_offs; };
tkz_status encode_device(tkz_encoder* e, Workspace* ws, const uint8_t* d_bytes, const int64_t* d_offs, int64_t n_docs, int64_t total,
                         int32_t* d_out, int64_t out_cap, int64_t* d_out_offs, hipStream_t stream, bool pretok,
                         uint64_t* d_bitmap_only, int64_t* total_tokens, PiecesOut* po = nullptr, int phase = kCallWhole, int64_t* d_counts3 = nullptr,
                         const IngestSrc* ingest = nullptr) {
    // phase: kCallWhole -- enqueue, wait, evaluate (and again if a buffer had to grow); kCallBegin -- enqueue the first attempt and
    // return; kCallEnd -- wait for that attempt, evaluate, and carry on as kCallWhole does (tkz_encode_batch_device_begin / _end)
    using namespace tkz;
    if (n_docs < 0 || total < 0 || out_cap < 0) return fail(TKZ_E_ARG, "negative size");
    if (total_tokens) *total_tokens = 0;
    if (n_docs == 0 && total != 0) return fail(TKZ_E_ARG, "bytes without documents");
    const int64_t nwords = total / 64 + 1;
    if (total == 0) {
        if (phase != kCallEnd) {                                  // (kCallEnd: enqueued when it began)
            int64_t* acc0 = &ws->bytes_allocated;
            HIP_TRY(ws->w_counts3.ensure(32, acc0));
            { tkz::Launch L0{stream, nullptr, ws}; tkz::launch_counts3(L0, n_docs, 0, nullptr, e->t_counts3.as<int64_t>(), ws->w_counts3.as<int64_t>(), d_counts3); }
            if (d_out_offs) HIP_TRY(hipMemsetAsync(d_out_offs, 0, (size_t)(n_docs + 1) * sizeof(int64_t), stream));
            if (d_bitmap_only) { const uint64_t one = 1; HIP_TRY(hipMemcpyAsync(d_bitmap_only, &one, 8, hipMemcpyHostToDevice, stream)); }
            if (phase == kCallBegin) return TKZ_OK;               // (_begin returns without waiting)
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return TKZ_OK;
    }
    const int64_t ntiles = (total + kSub - 1) / kSub;       // sub-tiles: one wavefront each
    SlowCallLog slow_log(total);
    { const tkz_status ps = prepare_workspace(ws, total, n_docs, d_bitmap_only != nullptr, po != nullptr); if (ps != TKZ_OK) return ps; }
    int64_t* acc = &ws->bytes_allocated;
    if (!ws->h_counters) HIP_TRY(hipHostMalloc((void**)&ws->h_counters, sizeof(CounterBlock), 0));

    // a learning batch that ends any other way than with its promotion gives the encoder's one learning slot back
    struct LearnGuard {
        tkz_encoder* e; Workspace* ws; bool keep = false;
        ~LearnGuard() { if (!keep && ws->learning) { ws->learning = false; std::lock_guard<std::mutex> lock(e->mu); e->learning = false; } }
    } learn_guard{e, ws};
    // An attempt that has to be run again (lists, records or scratch to grow) leaves the document marks and the piece-start bitmap as they are: the
    // next one starts behind the pre-tokenizer.
    bool marks_ready = false;
    for (int attempt = 0; attempt < 5; ++attempt) {
        slow_log.attempts = attempt + 1;
        bool pieces_over = false;
        // THE SIZING ATTEMPT: a fresh workspace's first large batch probes a sixteenth of its sub-tiles first.  How long the miss lists must be (and, from the
        // scan, how many records there are) is known after ~1 ms instead of after a whole attempt up to k_probe that is thrown away: a fresh encoder's first
        // batch of text its vocabulary has not seen (5.1 GB, held-out vocabulary) went from 100 ms to ~50.  The full attempt follows on the same bitmaps; a
        // sample that under-estimates leaves the ordinary retry.
        int64_t kSizingMinSub = (int64_t(64) << 20) / kSub;                     // (64 MB of text; TKZ_SIZING_MIN_SUB: the tests' handle on it)
        if (attempt == 0 && !ws->sized) { const char* v = getenv("TKZ_SIZING_MIN_SUB"); if (v && atoll(v) > 0) kSizingMinSub = atoll(v); }
        const bool sizing = attempt == 0 && phase == kCallWhole && !ws->sized && pretok && !d_bitmap_only && !po && (total + kSub - 1) / kSub >= kSizingMinSub;
        const bool marks_reused = marks_ready;
        Launch L{stream, e->profiling ? prof_hook : nullptr, ws};
        // a large batch: two more streams, for the two kernels of the long pieces that last as long as their slowest wavefront (launch_encode runs them beside k_merge_short;
        // without the streams -- the runtime refused one, or an event -- they follow it as they always did)
        static const bool kNoFork = getenv("TKZ_NO_FORK") != nullptr;          // (development: A/B of the forked form)
        static const int kSideLong = [] { const char* v = getenv("TKZ_SIDE_LONG_GRID"); return v ? atoi(v) : 2048; }();
        static const int kSideCoop = [] { const char* v = getenv("TKZ_SIDE_COOP_GRID"); return v ? atoi(v) : 1024; }();
        // WHEN: the workspace's previous batch left a class queue short enough for its kernel to be a matter of latency -- at most kForkMaxLong entries, ~4 batches of 64 for
        // each of the 4,096 wavefronts the chip holds --; with a long queue (mixed text: 17 M entries in 1 GB) the queue kernel is the step's largest and needs the whole chip
        // (measured with the tail grids: 17.0 -> 19.8 ms), and on the bench text (1.3 M) the two forms are equal (20.6 / 20.8 ms).
        static const int64_t kForkMaxLong = [] { const char* v = getenv("TKZ_FORK_MAX_LONG"); return v ? (int64_t)atoll(v) : int64_t(1) << 20; }();
        // (round 6, last session) ... or up to twice that many when one in 2,048 of them is a piece of more than 128 bytes: such a queue ends in a tail of slow wavefronts
        // whatever its length.  436 MB of source text (1.18 M long misses, 1,467 of them of 129+ bytes): 94.5 -> 104.3 GB/s side by side; the bench text (1.31-1.43 M, none
        // above 128 bytes) loses 0.1-0.3 ms that way and stays serial.  profiles/r06/variants_side_by_side2.txt
        static const int64_t kForkMaxLongTail = [] { const char* v = getenv("TKZ_FORK_MAX_LONG_TAIL"); return v ? (int64_t)atoll(v) : int64_t(1) << 21; }();
        static const int64_t kForkTailShare = [] { const char* v = getenv("TKZ_FORK_TAIL_SHARE"); return v && atoll(v) > 0 ? (int64_t)atoll(v) : int64_t(2048); }();
        static const bool kTraceFork = getenv("TKZ_TRACE_FORK") != nullptr;
        if (kTraceFork && attempt == 0 && total > e->latency_bytes) fprintf(stderr, "[tkz fork] bytes %lld: the batch before left %lld long misses, %lld of more than 128 bytes\n", (long long)total, (long long)ws->last_lq_total, (long long)ws->last_coop);
        const bool short_queue = ws->last_lq_total >= 0 && (ws->last_lq_total <= kForkMaxLong || (ws->last_lq_total <= kForkMaxLongTail && ws->last_coop * kForkTailShare >= ws->last_lq_total));
        if (total > e->latency_bytes && !kNoFork && short_queue) {
            if (!ws->fork_token) { int none = 0; ws->fork_token = g_fork_in_flight.compare_exchange_strong(none, 1); }      // (given back when the call ends: ~Lease)
            bool ok = ws->fork_token;
            for (hipStream_t* st : {&ws->st_side, &ws->st_side2}) if (!*st && hipStreamCreateWithFlags(st, hipStreamNonBlocking) != hipSuccess) { *st = nullptr; ok = false; }
            for (hipEvent_t* ev : {&ws->ev_fork, &ws->ev_join, &ws->ev_join2}) if (!*ev && hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { *ev = nullptr; ok = false; }
            if (ok) { L.side = ws->st_side; L.side2 = ws->st_side2; L.ev_fork = ws->ev_fork; L.ev_join = ws->ev_join; L.ev_join2 = ws->ev_join2; L.side_long_grid = kSideLong; L.side_coop_grid = kSideCoop; }
            else (void)hipGetLastError();
        }
        int32_t* counters = ws->w_counters.as<int32_t>();
        if (!(phase == kCallEnd && attempt == 0)) {              // (kCallEnd: the first attempt is in flight already)
        // the tables as they are NOW, one consistent copy for the whole attempt (a promotion at the end of another call's batch replaces the
        // SHORT / MID images and the promo array together: k_probe and k_place of one attempt must see the same generation)
        TkzTables T;
        {
            std::lock_guard<std::mutex> lock(e->mu);
            // LEARNING: the first batches of documents on the batch path (and once more, kPromoSecondBytes later; and again whenever the text has drifted:
            // adapt_after_batch below) count the memo's hits per slot.  A learning WINDOW is promo_min_bytes of text: one large batch, or -- TKZ_OPT_ADAPT --
            // as many smaller ones as it takes (a caller whose batches are 1 MB learns too).
            // (TKZ_OPT_ADAPT: the rounds go on -- a gigabyte after the first window began, then two, four, eight ... gigabytes after the one before: text that
            //  changed without moving the miss share, or right behind a promotion, is still learnt, a round costs one batch that counts hits and ~50 ms of a
            //  host thread; pieces are only ever ADDED by a round -- what empties the list is a drift, or the list reaching its cap: adapt_after_batch)
            const int64_t round_gap = e->adapt_round_bytes << std::min(std::max(e->promo_rounds - 1, 0), 20);
            if (attempt == 0 && !ws->learning && pretok && !d_bitmap_only && e->promo_mode == 1 && !e->learning && (e->adapt || e->promo_rounds < kPromoAutoRounds) &&
                e->T.memo_n != 0 && e->T.max_rank < (int32_t)kPromoFlag && (e->adapt || total >= e->promo_min_bytes) && e->promo_items.size() < e->promo_cap &&
                (e->promo_rounds == 0 || e->bytes_seen - e->bytes_at_promo >= round_gap)) {
                // (the gigabyte to the second round counts from the START of the first learning window: a job of 5 GB batches learns in its first two)
                // The memo is emptied for a window that follows a drift -- it is full of the old text's pieces and takes no new ones --, and that only while
                // no OTHER call is in flight: an entry never changes once it is valid (tkz_tables.h), which is what makes a hit exact, so the table is
                // cleared under nobody's feet, synchronously, with the encoder's lock held (16 MB: microseconds, once per drift)
                bool quiet = true;
                if (e->memo_clear_pending) for (Workspace* w : e->pool) if (w != ws && w->busy) quiet = false;
                if (quiet && e->t_memo_hits.ensure((size_t)e->memo_slots * 4, &e->bytes_allocated) == hipSuccess &&
                    e->t_long_log.ensure((size_t)kLongLogCap * kLongLogDwords * 4 + 64, &e->bytes_allocated) == hipSuccess) {
                    bool ok = true;
                    if (e->memo_clear_pending) {
                        ok = hipMemsetAsync(e->t_memo.p, 0, size_t(e->memo_slots) * sizeof(TkzMemoSlot), stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
                        if (ok) e->memo_clear_pending = false;
                    }
                    if (ok) {
                        ws->learn_window_start = e->learn_bytes == 0;
                        if (ws->learn_window_start) e->bytes_at_promo = e->bytes_seen;
                        e->learning = true; ws->learning = true;
                    }
                }
            }
            T = e->T;
        }
        if (ws->learning) {
            T.memo_hits = e->t_memo_hits.as<uint32_t>();
            T.memo_hits_sparse = total >= (int64_t(64) << 20) ? 1u : 0u;       // (below 64 MB every hit is counted: a few million atomics at most)
            if (ws->learn_window_start && attempt == 0) {                       // (a window's later batches add to its counters and its log)
                HIP_TRY(hipMemsetAsync(T.memo_hits, 0, (size_t)e->memo_slots * 4, stream));
                HIP_TRY(hipMemsetAsync(e->t_long_log.as<char>() + (size_t)kLongLogCap * kLongLogDwords * 4, 0, 64, stream));
            }
        }
        int64_t* grand = reinterpret_cast<int64_t*>(ws->w_counters.as<char>() + offsetof(CounterBlock, grand));
        unsigned long long* pool_head = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, pool_head));
        uint64_t* docbits = ws->w_docbits.as<uint64_t>();
        uint64_t* startbits = ws->w_startbits.as<uint64_t>();
        if (marks_reused) {      // (the counters and the sub-tile flags only)
            HIP_TRY(hipMemsetAsync(ws->w_zero.p, 0, 256, stream));
            if (!d_bitmap_only) HIP_TRY(hipMemsetAsync(ws->w_heavyq.p, 0, (size_t)(ws->w_zero.as<char>() + ws->zero_bytes - ws->w_heavyq.as<char>()), stream));
        } else {
        if (ingest) launch_ingest(L, ingest->h_bytes, total, const_cast<uint8_t*>(d_bytes), ingest->h_offs, n_docs + 1, const_cast<int64_t*>(d_offs), ws->w_zero.p, (int64_t)ws->zero_bytes);
        // counters, document-start bits, sub-tile flags.  (A chunk of a host batch clears them with a kernel of its own, not a fill command: the runtime's fill is
        //  a blit kernel that queued behind its D2H blit of the chunk before -- the 16 MB call's second chunk started when the first one's download ended.)
        else if (ws->zero_bytes <= (size_t(8) << 20)) launch_ingest(L, nullptr, 0, nullptr, nullptr, 0, nullptr, ws->w_zero.p, (int64_t)ws->zero_bytes);
        else HIP_TRY(hipMemsetAsync(ws->w_zero.p, 0, ws->zero_bytes, stream));
        launch_docmark(L, d_offs, n_docs, total, docbits, counters);
        if (!pretok) {
            HIP_TRY(hipMemcpyAsync(startbits, docbits, (size_t)nwords * 8, hipMemcpyDeviceToDevice, stream));
        } else if (e->pretok_seq) {
            HIP_TRY(hipMemcpyAsync(startbits, docbits, (size_t)nwords * 8, hipMemcpyDeviceToDevice, stream));
            launch_pretok_seq(L, e->pattern, d_bytes, d_offs, n_docs, total, startbits, T.bmp_class, counters);
        } else {
            HIP_TRY(ws->w_xq.ensure((size_t)(nwords / kRowsPerWave + 4) * 16, acc));     // two queues (launch_pretok_rows)
            launch_pretok_rows(L, e->pattern, d_bytes, d_offs, n_docs, total, docbits, startbits, nwords, T.bmp_class, counters,
                               ws->w_xq.as<int64_t>(), reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, xcount)));
        }
        if (pretok && e->case_equiv && e->pattern == TKZ_PATTERN_CL100K) launch_case_equiv_fix(L, d_bytes, total, docbits, startbits);
        }
        if (d_bitmap_only) {
            HIP_TRY(hipMemcpyAsync(d_bitmap_only, startbits, (size_t)nwords * 8, hipMemcpyDeviceToDevice, stream));
        } else {
            EncodeParams P{};
            P.bytes = d_bytes; P.total = total; P.startbits = startbits; P.docbits = docbits; P.nwords = nwords;
            P.offs = d_offs; P.n_docs = n_docs;
            P.tmp = ws->w_tmp.as<int32_t>(); P.dense = ws->w_dense.as<int32_t>(); P.tile_count = ws->w_tcount.as<int32_t>();
            P.prank = ws->w_prank.as<int32_t>(); P.prank_cap = (int64_t)(ws->w_prank.cap / 4); P.pcount = ws->w_pcount.as<int32_t>(); P.pbase = ws->w_pbase.as<int64_t>();
            P.mlist = ws->w_mlist.as<uint32_t>(); P.mquad = ws->w_mquad.as<uint4>(); P.mcap = ws->mcap; P.mcount = ws->w_mcount.as<uint32_t>();
            P.docord_base = ws->w_dbase.as<int64_t>(); P.doc_tok = ws->w_doctok.as<int32_t>(); P.counters = counters;
            P.giant_q = ws->w_gq.as<int64_t>(); P.giant_cap = total / kArenaPiece + 1; P.giant_cnt = ws->w_gcnt.as<int32_t>();
            P.giant_count = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, heavy_count));
            P.giant_ticket = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, giant_ticket));
            P.heavy_flag = ws->w_heavyq.as<uint8_t>(); P.nsub = ntiles;
            // (a batch of at most 16 MB waits for the slowest wavefront of every kernel, not for throughput: TKZ_OPT_LATENCY_BYTES, launch_encode.  Handing its
            //  pieces of 33..128 bytes to k_merge_coop as well was tried: 115 us in that kernel for what the lanes do in 6 -- a wavefront takes ~30 us a piece)
            P.lane_piece = kLanePiece;
            P.latency = total <= e->latency_bytes ? 1 : 0;
            // (... and in a small batch, whose three merge stages run as ONE launch: k_merge_latency.  $TKZ_NO_LATENCY_FUSE: the three launches, development A/B)
            static const bool kNoLatencyFuse = getenv("TKZ_NO_LATENCY_FUSE") != nullptr;
            P.tc_atomic = (L.side || (P.latency && !kNoLatencyFuse)) ? 1 : 0;
            if (L.side && !sizing) ++ws->forked_batches;
            P.coop_cap = total / P.lane_piece + 64;
            HIP_TRY(ws->w_coopq.ensure((size_t)P.coop_cap * 8, acc));
            P.coop_q = ws->w_coopq.as<uint64_t>();
            P.coop_count = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, coop_count));
            P.coop_ticket = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, coop_ticket));
            P.pool = ws->w_pool.as<int32_t>(); P.pool_head = pool_head; P.pool_cap = (int64_t)(ws->w_pool.cap / 4);
            P.lq_cnt = ws->w_lqcnt.as<int32_t>(); P.lq_base = ws->w_lqbase.as<int64_t>(); P.lq = ws->w_lq.as<uint64_t>(); P.lq_cap = total / (kShortMax + 1) + 64;
            P.lq_total = reinterpret_cast<int64_t*>(ws->w_counters.as<char>() + offsetof(CounterBlock, lq_total)); P.lq_bsum = ws->w_bsum.as<int64_t>();
            P.miss_sums = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, miss_short));
            P.ablate = 0; P.devprof = nullptr;
            P.stats = e->piece_stats ? e->t_stats.as<unsigned long long>() : nullptr;
            // (statistics: an attempt that has to be run again -- lists or records to grow -- must not be counted twice: the block as it was before
            //  this attempt waits behind it and is put back on a retry)
            if (P.stats) HIP_TRY(hipMemcpyAsync(e->t_stats.as<char>() + 64, P.stats, 64, hipMemcpyDeviceToDevice, stream));
            P.place128 = ws->place128 ? 1 : 0;
            P.promo = T.promo; P.pextra = T.promo ? ws->w_pextra.as<int32_t>() : nullptr;
            if (ws->learning) {
                P.long_log = e->t_long_log.as<uint32_t>(); P.long_log_cap = (int32_t)kLongLogCap; P.long_log_sparse = T.memo_hits_sparse ? 1 : 0;
                P.long_log_count = reinterpret_cast<unsigned long long*>(e->t_long_log.as<char>() + (size_t)kLongLogCap * kLongLogDwords * 4);   // (the encoder's: it runs on through the batches of a window)
            }
#ifdef TKZ_DEVPROF
            { const char* ab = getenv("TKZ_DEV_ABLATE"); P.ablate = ab ? atoi(ab) : 0; }
            if (P.ablate & 16) {
                if (!g_devprof) { HIP_TRY(hipMalloc((void**)&g_devprof, 64 * 8)); }
                HIP_TRY(hipMemsetAsync(g_devprof, 0, 64 * 8, stream));
            }
            P.devprof = g_devprof;
#endif
            int64_t* ndocstarts = reinterpret_cast<int64_t*>(ws->w_counters.as<char>() + offsetof(CounterBlock, ndocstarts));
            // piece granularity: the piece-start bitmap takes the place of the document bitmap from here on, so that the encode
            // kernels record the token position of every PIECE start and k_docoffs yields the token range of every piece
            const uint64_t* markbits = po ? startbits : docbits;
            P.docbits = markbits;
            int64_t* npieces = reinterpret_cast<int64_t*>(ws->w_counters.as<char>() + offsetof(CounterBlock, npieces));
            // the marks (document starts; at piece granularity the piece starts) and the piece starts of every sub-tile, counted in ONE pass over
            // the two bitmaps and scanned by ONE launch (the piece counts rounded up to whole record lines: where a sub-tile's records live in `prank`)
            launch_doccount2(L, markbits, startbits, nwords, total, ntiles, ws->w_dcount.as<int32_t>(), ws->w_pcount.as<int32_t>());
            launch_scan2(L, ntiles, ws->w_bsum.as<int64_t>(), ws->w_dcount.as<int32_t>(), ws->w_dbase.as<int64_t>(), ndocstarts, 1,
                         ws->w_pcount.as<int32_t>(), ws->w_pbase.as<int64_t>(), npieces, kRecordLine, -1);
            if (po) {
                int64_t np = 0;
                HIP_TRY(hipMemcpyAsync(&ws->h_counters->ndocstarts, ndocstarts, 8, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                np = ws->h_counters->ndocstarts;                  // piece starts below `total` (a document start is one)
                po->n_pieces = np;
                // piece arrays too small: the launch sequence still runs to its end (without the piece arrays), so that the caller
                // learns BOTH required sizes from this one call (tkz.h: *n_pieces and *needed_ids on TKZ_E_CAPACITY)
                pieces_over = np > po->piece_cap;
                if (!pieces_over) launch_piece_index(L, startbits, nwords, total, ntiles, ws->w_dbase.as<int64_t>(), np, po->piece_boffs, d_offs, n_docs, po->doc_piece);
            }
            if (sizing) {
                launch_probe_sample(L, T, P, std::min<int64_t>(ntiles, std::max<int64_t>(ntiles / 16, kSizingMinSub / 16)));
            } else {
            launch_encode(L, T, P, ntiles);
            if (P.stats) launch_miss_stats(L, P, ntiles);
            launch_scan2(L, ntiles, ws->w_bsum.as<int64_t>(), P.tile_count, ws->w_tbase.as<int64_t>(), grand, 1, nullptr, nullptr, nullptr, 1, K_SCAN);
            launch_place(L, P, ws->w_tbase.as<int64_t>(), ntiles, d_out, out_cap);
            if (po) {
                if (!pieces_over) launch_docoffs(L, po->piece_boffs, po->n_pieces, total, ws->w_tbase.as<int64_t>(), markbits, P.docord_base, P.doc_tok, grand, po->piece_toffs);
                launch_counts3(L, n_docs, total, grand, e->t_counts3.as<int64_t>(), ws->w_counts3.as<int64_t>(), d_counts3);
            } else      // (the batch's {n_docs, n_bytes, n_tokens} blocks by the same launch)
                launch_docoffs(L, d_offs, n_docs, total, ws->w_tbase.as<int64_t>(), docbits, P.docord_base, P.doc_tok, grand, d_out_offs,
                               n_docs, e->t_counts3.as<int64_t>(), ws->w_counts3.as<int64_t>(), d_counts3);
            }
        }
        HIP_TRY(hipMemcpyAsync(ws->h_counters, counters, sizeof(CounterBlock), hipMemcpyDeviceToHost, stream));
        }
        if (phase == kCallBegin) { learn_guard.keep = true; return TKZ_OK; }
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipGetLastError());
        if (e->profiling) prof_collect(ws);
#ifdef TKZ_DEVPROF
        if (g_devprof && getenv("TKZ_DEV_ABLATE") && (atoi(getenv("TKZ_DEV_ABLATE")) & 16) && !d_bitmap_only) {
            unsigned long long h[64];
            HIP_TRY(hipMemcpy(h, g_devprof, sizeof h, hipMemcpyDeviceToHost));
            const double w = h[0] ? (double)h[0] : 1.0;
            fprintf(stderr, "[tkz devprof] k_probe waves %llu  clock ticks/wave: total %.0f  load+compact %.0f  short batches %.0f  mid batches %.0f | mid pieces/wave %.1f pieces/wave %.1f\n",
                    h[0], h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w);
            if (h[51]) fprintf(stderr, "[tkz devprof] k_probe lane-cycle table, per sub-tile: load+compact %.0f ticks (64 lanes) | 13+-byte pre-pass %.0f ticks, %.1f pieces in %.2f passes of 64 (lane use %.3f) | main loop: first bucket %.0f ticks at lane use %.3f (%.2f batches of 64, %.1f pieces), second bucket %.0f ticks for %.1f lanes (%.3f of the lanes of the iterations that run it: %.2f of %.2f iterations), records and lists %.0f ticks at lane use %.3f\n",
                               h[2] / w, h[4] / w, h[5] / w, (double)((h[5] + 63 * h[0]) / 64) / w, h[5] ? (double)h[5] / (64.0 * (double)((h[5] + 63 * h[0]) / 64)) : 0.0,
                               h[48] / w, (double)h[52] / (64.0 * (double)h[51]), h[51] / w, h[52] / w, h[49] / w, h[53] / w, h[54] ? (double)h[53] / (128.0 * (double)h[54]) : 0.0, h[54] / w, (double)((h[51] + 1) / 2) / w,
                               h[50] / w, (double)h[52] / (64.0 * (double)h[51]));
            if (h[8]) fprintf(stderr, "[tkz devprof] k_giant_merge pieces %llu  clock ticks/piece: rounds in global memory %.0f  the tail %.0f | bytes/piece %.0f tokens/piece %.0f | slowest piece %llu ticks | global rounds/piece %.2f | parts/piece when the tail took over %.0f\n",
                              h[8], (double)h[9] / h[8], (double)h[10] / h[8], (double)h[11] / h[8], (double)h[12] / h[8], h[13], (double)h[14] / h[8], (double)h[7] / h[8]);
            if (h[8]) fprintf(stderr, "[tkz devprof] slowest giant piece: %llu bytes -> %llu tokens, global rounds %llu (%llu ticks), bytes first/middle/last %02llx %02llx %02llx\n",
                              h[24], h[27], h[25], h[28], h[29] & 255, (h[29] >> 8) & 255, (h[29] >> 16) & 255);
            if (h[32]) fprintf(stderr, "[tkz devprof] k_merge_long waves %llu units %llu  ticks/wave %.0f | of all ticks: sort %.3f batch formation %.3f bytes %.3f first level %.3f merges %.3f emission %.3f | fast batches %llu lanes/batch %.1f steps/batch %.1f merges/lane %.2f lane use in the merge loop %.3f ticks/step %.0f\n",
                               h[32], h[44], (double)h[33] / h[32], (double)h[34] / h[33], (double)h[35] / h[33], (double)h[36] / h[33], (double)h[37] / h[33], (double)h[38] / h[33], (double)h[39] / h[33],
                               h[40], (double)h[41] / (h[40] ? h[40] : 1), (double)h[42] / (h[40] ? h[40] : 1), (double)h[43] / (h[41] ? h[41] : 1), (double)h[43] / (64.0 * (h[42] ? h[42] : 1)), (double)h[38] / (h[42] ? h[42] : 1));
            if (h[16]) fprintf(stderr, "[tkz devprof] tail: batches %llu merges %llu (%.2f a batch) proposals/batch %.1f | rounds for chains of equal pairs %llu | ticks/batch %.0f | longest tail: %llu batches, %llu ticks\n",
                               h[16], h[17], (double)h[17] / h[16], (double)h[18] / h[16], h[19], (double)h[22] / h[16], h[20], h[23]);
        }
#endif
        if (!marks_reused) { e->last_xcount = (int64_t)ws->h_counters->xcount; e->last_xcount2 = (int64_t)ws->h_counters->xcount2; }
        const int32_t err = ws->h_counters->err;
        if (err & kErrOffsets) return fail(TKZ_E_ARG, "document offsets must start at 0, be non-decreasing and end at the byte count");
        if (err & kErrUtf8) return fail(TKZ_E_INVALID_UTF8, "input is not well-formed UTF-8 (or a document boundary falls inside a character)");
        if (err & kErrTooLong) return fail(TKZ_E_UNSUPPORTED, "a single piece longer than 2^30 bytes");
        marks_ready = true;                // (what is wrong from here on is the size of a buffer)
        if (!d_bitmap_only) ws->sized = true;
        if (!d_bitmap_only && !sizing && total > e->latency_bytes) ws->last_coop = (int64_t)ws->h_counters->coop_count;
        if (!d_bitmap_only && !sizing && total > e->latency_bytes) ws->last_lq_total = ws->h_counters->lq_total;      // (the next batch's form of the long pieces' kernels: above)
        if (sizing) {
            // k_place's form for THIS batch from the sample (it is otherwise chosen from the batch before: a fresh encoder's first miss-heavy batch ran
            // k_place<64> with most sub-tiles on its general path, 11.7 ms against 7)
            const int64_t nsample = std::min<int64_t>(ntiles, std::max<int64_t>(ntiles / 16, kSizingMinSub / 16));
            if ((int64_t)ws->h_counters->over64 * 5 > nsample) { ws->place128 = true; ws->low_place = 0; }
        }
        if ((err & kErrPool) && attempt < 4) {
            // scratch for the giant pieces was too small.  pool_head keeps counting past the capacity, so it holds the exact need
            // (6 int32 per byte of every giant piece of the batch): size the pool for that -- not for the whole batch -- and rerun
            const size_t need = (size_t)ws->h_counters->pool_head * 4 + 4096;
            if (ws->w_pool.ensure(need, acc) != hipSuccess)
                return fail(TKZ_E_OUT_OF_MEMORY, "scratch for the pieces longer than 1024 bytes: " + std::to_string(need) + " bytes could not be allocated");
            if (e->piece_stats && e->t_stats.p && !d_bitmap_only) HIP_TRY(hipMemcpyAsync(e->t_stats.p, e->t_stats.as<char>() + 64, 64, hipMemcpyDeviceToDevice, stream));
            continue;
        }
        if (err & kErrPool) return fail(TKZ_E_OUT_OF_MEMORY, "long-piece scratch exhausted");
        if ((err & kErrMissCap) && attempt < 4) {
            // a sub-tile missed more pieces than its list holds (text where nearly every piece misses the vocabulary): the longest list
            // any sub-tile needed is known now -- longer lists for this workspace from here on, and the batch again
            int32_t want = kMissCapMin;
            while (want < ws->h_counters->mneed && want < kMissCapMax) want *= 2;
            if (want <= ws->mcap) return fail(TKZ_E_DEVICE, "miss list overflow");
            if (ws->w_mlist.ensure((size_t)ntiles * (size_t)want * 4, acc) != hipSuccess || ws->w_mquad.ensure((size_t)ntiles * (size_t)want * 16, acc) != hipSuccess)
                return fail(TKZ_E_OUT_OF_MEMORY, "miss lists: " + std::to_string((size_t)ntiles * (size_t)want * 20) + " bytes could not be allocated");
            ws->mcap = want;
            if (e->piece_stats && e->t_stats.p && !d_bitmap_only) HIP_TRY(hipMemcpyAsync(e->t_stats.p, e->t_stats.as<char>() + 64, 64, hipMemcpyDeviceToDevice, stream));
            continue;
        }
        if (err & kErrMissCap) return fail(TKZ_E_DEVICE, "miss list overflow");
        if ((err & kErrCapacity) && attempt < 4) {          // more pieces than the record buffer was sized for: the exact count is known now
            const size_t need = ((size_t)ws->h_counters->npieces + 4096) * 4;
            if (ws->w_prank.ensure(need, acc) != hipSuccess)
                return fail(TKZ_E_OUT_OF_MEMORY, "piece records: " + std::to_string(need) + " bytes could not be allocated");
            if (e->piece_stats && e->t_stats.p && !d_bitmap_only) HIP_TRY(hipMemcpyAsync(e->t_stats.p, e->t_stats.as<char>() + 64, 64, hipMemcpyDeviceToDevice, stream));
            continue;
        }
        if (err & kErrCapacity) return fail(TKZ_E_DEVICE, "piece record buffer overflow");
        if (sizing) {       // (the sample fitted the lists as they are; the records are counted exactly by the scan)
            const size_t need = ((size_t)ws->h_counters->npieces + 4096) * 4;
            if (need > ws->w_prank.cap && ws->w_prank.ensure(need, acc) != hipSuccess)
                return fail(TKZ_E_OUT_OF_MEMORY, "piece records: " + std::to_string(need) + " bytes could not be allocated");
            if (e->piece_stats && e->t_stats.p) HIP_TRY(hipMemcpyAsync(e->t_stats.p, e->t_stats.as<char>() + 64, 64, hipMemcpyDeviceToDevice, stream));
            continue;
        }
        if (err & kErrKeyNotFound) return fail(TKZ_E_KEY_NOT_FOUND, "a byte of the input is not in the vocabulary (KeyNotFoundException in the reference)");
        if (!d_bitmap_only && e->piece_stats) {
            std::lock_guard<std::mutex> lock(e->mu);
            ++e->stat_batches; e->stat_giants += (int64_t)ws->h_counters->heavy_count;
        }
        // Growing is immediate, shrinking waits for kLowBatches consecutive batches that would have done with less (the round-4 advisor: a
        // workspace that alternates miss-heavy and ordinary batches -- or the 48 MB chunks of one host call that does -- must not overflow,
        // re-run, free and re-allocate on every other batch).
        constexpr int kLowBatches = 3;
        if (!d_bitmap_only) {
            const bool heavy = (int64_t)ws->h_counters->over64 * 5 > ntiles;      // (more than a fifth of the sub-tiles: the next batches' k_place)
            if (heavy) { ws->place128 = true; ws->low_place = 0; }
            else if (ws->place128 && ++ws->low_place >= kLowBatches) { ws->place128 = false; ws->low_place = 0; }
        }
        if (!d_bitmap_only && ws->mcap > kMissCapMin) {
            // lists that were grown for an earlier batch (text where nearly every piece misses) and that the last kLowBatches batches filled to
            // less than half: half as long from here on (one step at a time), and the buffers given back when they are far larger than such
            // batches need (the lists are ntiles * mcap * 20 bytes: 1.25 B per input byte at 64 entries, 20 B at 1024)
            int32_t want = kMissCapMin;
            while (want < ws->h_counters->mhigh) want *= 2;
            if (want >= ws->mcap) ws->low_lists = 0;
            else if (++ws->low_lists >= kLowBatches) {
                ws->low_lists = 0;
                ws->mcap = std::max(want, ws->mcap / 2);
                if (ws->w_mquad.cap > (size_t)ntiles * (size_t)ws->mcap * 16 * 4) {
                    *acc -= (int64_t)(ws->w_mquad.cap + ws->w_mlist.cap);
                    ws->w_mquad.release(); ws->w_mlist.release();
                }
            }
        }
        if (!d_bitmap_only && pretok) {
            // the batch is done: if it completes a learning window, the hottest entries are promoted now (the copy of the memo back to the host and the
            // rebuilt key tables cost tens of milliseconds: on a thread, behind the batch); else the share of pieces that missed the key tables is
            // compared with what it was after the last promotion (adapt_after_batch)
            bool promote = false, relearn = false;
            {
                std::lock_guard<std::mutex> lock(e->mu);
                e->bytes_seen += total;
                if (ws->learning) {
                    e->learn_bytes += total;
                    e->win_miss += (double)(ws->h_counters->miss_short + ws->h_counters->miss_long); e->win_pieces += (double)ws->h_counters->npieces;
                    promote = !e->adapt || e->learn_bytes >= e->promo_min_bytes;
                    if (!promote) { ws->learning = false; e->learning = false; }      // (the window goes on with the next batch)
                    else if (e->adapt && e->last_window_valid && e->win_pieces >= 1 && e->win_miss / e->win_pieces > e->last_window_miss * 1.25 + 0.01) {
                        // A window whose miss share is a quarter (and a point) ABOVE the window's before it -- although that one's promotions have been installed since,
                        // and promotions only lower the share on unchanged text -- was counted on ANOTHER text, with a memo full of the old one's pieces (it takes no
                        // new entry into a full bucket): what it found is a fraction of what a fresh encoder finds (2 k against 9 k pieces on the source text behind 3 GB
                        // of synthetic text).  A drift: start over -- this window's counts are dropped with the promotions, the memo is emptied, the next batch begins a window.
                        promote = false; relearn = true; ws->learning = false;
                        e->learn_bytes = 0; e->win_miss = e->win_pieces = 0;
                    }
                } else relearn = adapt_after_batch(e, total, (double)(ws->h_counters->miss_short + ws->h_counters->miss_long), (double)ws->h_counters->npieces);
                if (relearn) e->learning = true;                                     // (nothing learns while the promotions are being dropped)
            }
            if (promote || relearn) {
                if (promote) {
                    std::lock_guard<std::mutex> lock(e->mu);
                    ws->learning = false; e->learn_bytes = 0;
                    e->window_valid = e->win_pieces >= 1; e->window_miss = e->window_valid ? e->win_miss / e->win_pieces : 0; e->win_miss = e->win_pieces = 0;
                    e->last_window_valid = e->window_valid; e->last_window_miss = e->window_miss;
                }
                // (the workspace is this call's no longer once it returns; the counters, the log and the memo are the encoder's, and no other batch writes the
                //  first two while e->learning is set)
                join_promotion(e);                 // (the previous one ended before this batch could be armed: this only reaps the thread)
                auto work = [e, promote] {
                    DeviceScope scope;
                    const bool dev = scope.enter(e->device) == hipSuccess;
                    if (promote) {
                        int64_t added = 0;
                        size_t held;
                        { std::lock_guard<std::mutex> lock(e->mu); held = e->promo_items.size(); }
                        if (dev) (void)promote_from_memo(e, true, true, &added);   // (a failure leaves the tables as they were)
                        std::lock_guard<std::mutex> lock(e->mu);
                        e->learning = false; ++e->promo_rounds; ++e->n_promotions;
                        // A round that found much it did not know -- more than a tenth of what the list held -- is a young encoder, or text that CHANGED without the miss
                        // share having had a settled level to leave (the change fell between two installs): the next round then follows a gigabyte later, not
                        // 2^rounds gigabytes.  (bench.py's drift leg, synthetic -> real text: the steps beyond 2 GB ran at 0.81 of an encoder that only ever saw
                        // the real text, whose second round comes after 1 GB while this one's was 4 GB away.)
                        if (e->adapt && e->promo_rounds > 1 && (size_t)added * 10 > held) e->promo_rounds = 1;
                        e->bytes_at_install = e->bytes_seen; e->ew_valid = e->base_valid = false;
                    } else {
                        if (dev) (void)drop_promotions(e, true);
                        std::lock_guard<std::mutex> lock(e->mu);
                        e->learning = false; e->promo_rounds = 0; e->learn_bytes = 0; e->memo_clear_pending = true; ++e->n_relearns;
                        e->bytes_at_install = e->bytes_at_promo = e->bytes_seen; e->ew_valid = e->base_valid = e->window_valid = e->last_window_valid = false; e->win_miss = e->win_pieces = 0;
                    }
                };
                bool started = false;
                {
                    std::lock_guard<std::mutex> jl(e->promo_join_mu);
                    try { e->promo_thread = std::thread(work); started = true; } catch (...) {}      // (no thread to be had: built here, as before round 5)
                }
                if (!started) { const std::string keep_msg = g_err; work(); g_err = keep_msg; }
            }
        }
        if (!d_bitmap_only) {
            if (total_tokens) *total_tokens = ws->h_counters->grand;
            if (pieces_over) return fail(TKZ_E_CAPACITY, "piece arrays too small");
            if (ws->h_counters->grand > out_cap) return fail(TKZ_E_CAPACITY, "output capacity too small");
        }
        return TKZ_OK;
    }
    return fail(TKZ_E_DEVICE, "unreachable");
}

// ---- the single-launch path (k_small) --------------------------------------------------------------------------------------------------
// ITokenizer.Encode(text) on a prompt is microseconds in the reference (TikTokenizer.cs:178-207); the batch path above costs ~25 kernel
// launches, four copy commands and two synchronisations whatever the size: ~160 us for 64 bytes.  A batch of at most kSmallMaxBytes bytes
// in at most kSmallMaxDocs documents of at most kSmallMaxDoc bytes each goes through ONE launch instead: the caller's bytes and offsets
// are memcpy'd into a page-locked block, k_small (one workgroup, all phases) reads them from there and writes ids, offsets and status
// back into it, one stream synchronisation, memcpy out.  Returns TKZ_OK with *handled = false when the kernel hands the batch back (a
// piece of more than 1024 bytes, an error to be diagnosed, lists or buffers to be grown): the caller then takes the batch path.
constexpr size_t kSmallOffBytes = 0, kSmallOffOffs = tkz::kSmallMaxBytes + 64, kSmallOffIds = kSmallOffOffs + (tkz::kSmallMaxDocs + 1) * 8,
                 kSmallOffOut = kSmallOffIds + tkz::kSmallMaxBytes * 4, kSmallOffRes = kSmallOffOut + (tkz::kSmallMaxDocs + 1) * 8, kSmallBlock = kSmallOffRes + 256;
bool small_eligible(const tkz_encoder* e, const int64_t* offs, int64_t n_docs, int64_t total) {
    if (!e->small_ok || e->profiling || e->pretok_seq || (e->case_equiv && e->pattern == TKZ_PATTERN_CL100K) || total <= 0 || total > tkz::kSmallMaxBytes || n_docs < 1 || n_docs > tkz::kSmallMaxDocs) return false;
    const bool o200k = e->pattern == TKZ_PATTERN_O200K || e->pattern == TKZ_PATTERN_O200K_DOTNET;
    if (o200k && total > tkz::kSmallMaxBytesO200k) return false;
    if (o200k)                                    // (split by the sequential matcher there, one lane per document)
        for (int64_t d = 0; d < n_docs; ++d) { const int64_t len = offs[d + 1] - offs[d]; if (len < 0 || len > tkz::kSmallMaxDoc) return false; }
    return true;
}
tkz_status encode_small(tkz_encoder* e, Workspace* ws, const uint8_t* bytes, const int64_t* offs, int64_t n_docs, int64_t total,
                        int32_t* out_ids, int64_t out_cap, int64_t* out_offsets, int64_t* needed, bool* handled) {
    using namespace tkz;
    *handled = false;
    { const tkz_status ps = prepare_workspace(ws, total, n_docs, false, false); if (ps != TKZ_OK) return ps; }
    int64_t* acc = &ws->bytes_allocated;
    HIP_TRY(ws->s_bytes[0].ensure((size_t)kSmallMaxBytes + 64, acc));
    HIP_TRY(ws->s_offs[0].ensure((size_t)(kSmallMaxDocs + 1) * 8, acc));
    if (!ws->h_small) HIP_TRY(hipHostMalloc((void**)&ws->h_small, kSmallBlock, 0));
    if (!ws->st_small) HIP_TRY(hipStreamCreateWithFlags(&ws->st_small, hipStreamNonBlocking));
    uint8_t* H = ws->h_small;
    memcpy(H + kSmallOffBytes, bytes, (size_t)total);
    memcpy(H + kSmallOffOffs, offs, (size_t)(n_docs + 1) * 8);
    int64_t* h_res = reinterpret_cast<int64_t*>(H + kSmallOffRes);
    h_res[0] = -1; h_res[1] = 0; h_res[2] = 0;
    const int64_t ntiles = (total + kSub - 1) / kSub;
    EncodeParams P{};
    P.bytes = ws->s_bytes[0].as<uint8_t>(); P.total = total; P.startbits = ws->w_startbits.as<uint64_t>(); P.docbits = ws->w_docbits.as<uint64_t>(); P.nwords = total / 64 + 1;
    P.offs = ws->s_offs[0].as<int64_t>(); P.n_docs = n_docs;
    P.tmp = ws->w_tmp.as<int32_t>(); P.dense = ws->w_dense.as<int32_t>(); P.tile_count = ws->w_tcount.as<int32_t>();
    P.prank = ws->w_prank.as<int32_t>(); P.prank_cap = (int64_t)(ws->w_prank.cap / 4); P.pcount = ws->w_pcount.as<int32_t>(); P.pbase = ws->w_pbase.as<int64_t>();
    P.mlist = ws->w_mlist.as<uint32_t>(); P.mquad = ws->w_mquad.as<uint4>(); P.mcap = ws->mcap; P.mcount = ws->w_mcount.as<uint32_t>();
    P.docord_base = ws->w_dbase.as<int64_t>(); P.doc_tok = ws->w_doctok.as<int32_t>(); P.counters = ws->w_counters.as<int32_t>();
    P.giant_q = ws->w_gq.as<int64_t>(); P.giant_cap = total / kArenaPiece + 1; P.giant_cnt = ws->w_gcnt.as<int32_t>();
    P.giant_count = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, heavy_count));
    P.giant_ticket = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, giant_ticket));
    P.heavy_flag = ws->w_heavyq.as<uint8_t>(); P.nsub = ntiles;
    P.pool = ws->w_pool.as<int32_t>(); P.pool_head = reinterpret_cast<unsigned long long*>(ws->w_counters.as<char>() + offsetof(CounterBlock, pool_head)); P.pool_cap = (int64_t)(ws->w_pool.cap / 4);
    P.ablate = 0; P.devprof = nullptr; P.stats = nullptr; P.place128 = 0; P.promo = nullptr; P.pextra = nullptr; P.lane_piece = kSmallLanePiece; P.latency = 0;
    SmallArgs A{};
    A.h_bytes = H + kSmallOffBytes; A.h_offs = reinterpret_cast<const int64_t*>(H + kSmallOffOffs);
    A.out = reinterpret_cast<int32_t*>(H + kSmallOffIds); A.out_cap = std::min<int64_t>(out_cap, kSmallMaxBytes); A.out_offs = reinterpret_cast<int64_t*>(H + kSmallOffOut);
    A.h_result = h_res;
    A.docbits = ws->w_docbits.as<uint64_t>(); A.startbits = ws->w_startbits.as<uint64_t>();
    A.pcount = ws->w_pcount.as<int32_t>(); A.pbase = ws->w_pbase.as<int64_t>(); A.docord_base = ws->w_dbase.as<int64_t>(); A.tile_base = ws->w_tbase.as<int64_t>();
    A.counter_words = (int32_t)(sizeof(CounterBlock) / 4);
    A.counts3[0] = e->t_counts3.as<int64_t>(); A.counts3[1] = ws->w_counts3.as<int64_t>();
    TkzTables T;
    { std::lock_guard<std::mutex> lock(e->mu); T = e->T; }
    P.promo = T.promo; P.pextra = T.promo ? ws->w_pextra.as<int32_t>() : nullptr;
    Launch L{ws->st_small, nullptr, ws};
    launch_small(L, T, P, A);
    HIP_TRY(hipStreamSynchronize(ws->st_small));
    HIP_TRY(hipGetLastError());
    ws->small_calls.fetch_add(1, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lock(e->mu); memcpy(ws->small_clocks, h_res + 4, sizeof ws->small_clocks); }
    if (h_res[0] != 0) { ws->small_fallbacks.fetch_add(1, std::memory_order_relaxed); return TKZ_OK; }          // (handled stays false)
    const int64_t tokens = h_res[2];
    if (needed) *needed = tokens;
    *handled = true;
    if (tokens > out_cap) return fail(TKZ_E_CAPACITY, "output capacity too small");
    if (tokens) memcpy(out_ids, H + kSmallOffIds, (size_t)tokens * 4);
    memcpy(out_offsets, H + kSmallOffOut, (size_t)(n_docs + 1) * 8);
    return TKZ_OK;
}

tkz_status check_encoder(tkz_encoder* e, DeviceScope& scope) {
    if (!e) return fail(TKZ_E_ARG, "null encoder");
    hipError_t r = scope.enter(e->device);
    if (r != hipSuccess) return fail(TKZ_E_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(r));
    return TKZ_OK;
}

hipError_t ensure_streams(Workspace* ws) {
    hipError_t r = hipSuccess;
    if (!ws->st_compute) r = hipStreamCreate(&ws->st_compute);
    if (r == hipSuccess && !ws->st_in) r = hipStreamCreate(&ws->st_in);
    if (r == hipSuccess && !ws->st_out) r = hipStreamCreate(&ws->st_out);
    for (int q = 0; q < 2 && r == hipSuccess; ++q) if (!ws->ev_in[q]) r = hipEventCreate(&ws->ev_in[q]);
    for (int q = 0; q < 3 && r == hipSuccess; ++q) if (!ws->ev_out[q]) r = hipEventCreate(&ws->ev_out[q]);
    return r;
}

// Is p page-locked host memory the device can address (tkz_host_alloc, hipHostMalloc, a torch pinned tensor)?  *dev: its device-side address.
bool pinned_host(const void* p, void** dev) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // (pageable memory: an error by design)
    if (a.type != hipMemoryTypeHost) return false;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) != hipSuccess || !d) { (void)hipGetLastError(); return false; }
    *dev = d;
    return true;
}

// host buffers -> staging -> device path -> back, for documents given as UTF-8 bytes (`bytes`) or as UTF-16 code units (`units`: uploaded as they are,
// Encoding.UTF8.GetBytes -- TikTokenizer.cs:261 -- runs on the device; offsets in units then).
//  * at most 128 KiB of UTF-8: the single-launch kernel (encode_small);
//  * a batch whose upload is below 1.5 chunks (a chunk: 16 MB from page-locked buffers, 32 MB from pageable ones; at least 8 MB): one launch sequence.  From
//    page-locked caller buffers the inputs are copied asynchronously and -- up to 8 MB of text -- the ids and offsets are written by the kernels STRAIGHT into
//    the caller's memory (k_place / k_docoffs store whole lines over PCIe): no download commands, one synchronisation;
//  * larger: document ranges of a chunk each, pipelined -- the upload of chunk k+2 (and, for UTF-16, its length pass), the launch sequences of chunks k+1 and
//    k+2 (enqueued ahead, on two workspaces) and the download of chunk k (page-locked results: on a copy engine of its own, tkz_sdma.h) run at the same time.
//    512 MB of page-locked text: 22 -> 37 GB/s, 64 MB: 22 -> 29.5 (profiles/r06/host_batches_ab.txt); what bounds it now is the download of the ids at the
//    link's duplex rate.
tkz_status encode_host(tkz_encoder* e, const uint8_t* bytes, const uint16_t* units, const int64_t* offs, int64_t n_docs, int32_t* out_ids,
                       int64_t out_cap, int64_t* out_offsets, int64_t* needed, bool pretok, uint64_t* bitmap) {
    using namespace tkz;
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    const bool u16 = units != nullptr;
    if (n_docs < 0 || !offs || (n_docs > 0 && !bytes && !units && offs[n_docs] > 0)) return fail(TKZ_E_ARG, "null buffer");
    if (offs[0] != 0) return fail(TKZ_E_ARG, "doc_offsets[0] must be 0");
    const int64_t total = offs[n_docs];                      // bytes, or code units
    if (total < 0) return fail(TKZ_E_ARG, u16 ? "negative unit count" : "negative byte count");
    if (needed) *needed = 0;
    if (u16 && total == 0) {
        for (int64_t d = 0; d <= n_docs; ++d) { if (offs[d] != 0) return fail(TKZ_E_ARG, "document offsets must start at 0, be non-decreasing and end at the unit count"); out_offsets[d] = 0; }
        return TKZ_OK;
    }
    Lease lease(e);
    Workspace* ws = lease.ws;
    int64_t* acc = &ws->bytes_allocated;
    const int64_t unit = u16 ? 2 : 1;
    // ($TKZ_HOST_CHUNK_BYTES: test knob, so that the CPU-emulated tests can exercise the pipeline on kilobytes)
    // (the single-launch path first: at most 128 KiB, a fraction of a chunk -- and none of the questions below are asked of a 64-byte prompt)
    if (!u16 && pretok && !bitmap && small_eligible(e, offs, n_docs, total)) {
        bool handled = false;
        st = encode_small(e, ws, bytes, offs, n_docs, total, out_ids, out_cap, out_offsets, needed, &handled);
        if (st != TKZ_OK || handled) return st;
    }
    // page-locked caller buffers?  (a single chunk then needs no staging for its results; the copies of every path are asynchronous)
    void *dv_in = nullptr, *dv_offs = nullptr, *dv_ids = nullptr, *dv_ooffs = nullptr;
    const bool pin_in = pinned_host(u16 ? (const void*)units : (const void*)bytes, &dv_in) && pinned_host(offs, &dv_offs);
    const bool pin_out = !bitmap && pinned_host(out_offsets, &dv_ooffs) && (out_cap == 0 || pinned_host(out_ids, &dv_ids));
    // (16 MB of upload a chunk.  Until round 6 a chunk's kernels were launched when the chunk before had drained, every chunk paid its launch sequence's
    //  floor of ~0.4 ms and 32 MB chunks were the optimum; with two launch sequences enqueued ahead and the downloads on a copy engine of their own the
    //  floor is hidden: profiles/r06/host_batches_ab.txt.  Pageable buffers keep 32 MB: the runtime stages their copies itself, synchronously, a cost per copy)
    static const int64_t kChunkEnv = [] { const char* v = getenv("TKZ_HOST_CHUNK_BYTES"); const long long n = v ? atoll(v) : 0; return n > 0 ? (int64_t)n : (int64_t)0; }();
    const int64_t kChunkBytes = kChunkEnv ? kChunkEnv : (pin_in && pin_out ? int64_t(16) << 20 : int64_t(32) << 20);
    static const int64_t kChunkMin = [] { const char* v = getenv("TKZ_HOST_CHUNK_MIN"); const long long n = v ? atoll(v) : 0; return n > 0 ? (int64_t)n : (int64_t(8) << 20); }();
    const int64_t up_bytes = total * unit;
    // (from 12 MB up a batch is two chunks at least, of 8 MB or more: the second chunk's upload runs beside the first one's kernels)
    const int64_t chunk_bytes = std::min(kChunkBytes, std::max(std::min(kChunkBytes, kChunkMin), up_bytes / 2));
    int64_t nchunks = (bitmap || !pretok || 2 * up_bytes < 3 * chunk_bytes) ? 1 : std::min<int64_t>(1024, std::max<int64_t>(2, (up_bytes + chunk_bytes / 2) / chunk_bytes));
    // chunk boundaries on documents: chunk k = documents [cut[k], cut[k+1]).  Offsets that are not monotone cannot be cut: the
    // whole batch then goes as one chunk and the device reports them (k_docmark)
    std::vector<int64_t> cut((size_t)nchunks + 1, 0);
    cut[(size_t)nchunks] = n_docs;
    for (int64_t k = 1; k < nchunks; ++k) {
        const int64_t want = total / nchunks * k;
        cut[(size_t)k] = std::lower_bound(offs, offs + n_docs, want) - offs;
        if (cut[(size_t)k] < cut[(size_t)k - 1] || offs[cut[(size_t)k]] < offs[cut[(size_t)k - 1]]) { nchunks = 1; break; }
    }
    if (nchunks == 1) { cut.assign(2, 0); cut[1] = n_docs; }
    if (!u16 && nchunks == 1 && !(pin_in && pin_out && pretok && !bitmap && total > 0)) {
        // ---- one chunk, ordinary (pageable) buffers: blocking copies either side of the launch sequence ----
        HIP_TRY(ws->s_bytes[0].ensure((size_t)total + 64, acc));
        HIP_TRY(ws->s_offs[0].ensure((size_t)(n_docs + 1) * 8, acc));
        const int64_t cap = bitmap ? 0 : std::min<int64_t>(out_cap, total);   // tokens <= bytes: more capacity is never used
        if (!bitmap) {
            HIP_TRY(ws->s_out[0].ensure((size_t)std::max<int64_t>(cap, 1) * 4, acc));
            HIP_TRY(ws->s_outoffs[0].ensure((size_t)(n_docs + 1) * 8, acc));
        } else {
            HIP_TRY(ws->s_out[0].ensure((size_t)(total / 64 + 1) * 8, acc));
        }
        if (total) HIP_TRY(hipMemcpy(ws->s_bytes[0].p, bytes, (size_t)total, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ws->s_offs[0].p, offs, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice));
        int64_t tokens = 0;
        st = encode_device(e, ws, ws->s_bytes[0].as<uint8_t>(), ws->s_offs[0].as<int64_t>(), n_docs, total, ws->s_out[0].as<int32_t>(), cap,
                           ws->s_outoffs[0].as<int64_t>(), nullptr, pretok, bitmap ? ws->s_out[0].as<uint64_t>() : nullptr, &tokens);
        if (needed) *needed = tokens;
        if (st != TKZ_OK) return st;
        if (bitmap) {
            HIP_TRY(hipMemcpy(bitmap, ws->s_out[0].p, (size_t)(total / 64 + 1) * 8, hipMemcpyDeviceToHost));
            return TKZ_OK;
        }
        if (tokens) HIP_TRY(hipMemcpy(out_ids, ws->s_out[0].p, (size_t)tokens * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out_offsets, ws->s_outoffs[0].p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost));
        return TKZ_OK;
    }
    // ---- chunks on three streams (a single chunk is the loop's one iteration) ----
    HIP_TRY(ensure_streams(ws));
    // the kernels write the caller's page-locked ids and offsets themselves -- up to 8 MB of text: beyond that a DMA download beats k_place's stores over PCIe
    // (measured, round 5: 16 MB 1.08 ms direct, 1.00 ms staged)
    const bool direct_out = nchunks == 1 && pin_out && up_bytes <= (int64_t(8) << 20) * unit;
    // TWO chunks' launch sequences are enqueued ahead (round 6).  Until then a chunk's kernels were launched when the chunk before had drained -- the host
    // needs its token count to place the download --, so the device idled for the ~25 launches of every chunk (tools/gpu_job_sdma.sh: 64 MB as 8 chunks took
    // 1.5 ms longer than as 2, ~250 us a chunk, whatever the download did).  Now chunk k + 2 is begun (encode_device, kCallBegin: enqueue and return) the
    // moment chunk k has ended (kCallEnd: wait, evaluate, retry if a list has to grow), on the workspace and the stream chunk k has just left: odd and even chunks
    // alternate between two leased workspaces, two input staging sets and -- since the download of chunk k is only ISSUED when k has ended -- three output sets.
    std::unique_ptr<Lease> lease_b;
    Workspace* W[2] = {ws, ws};
    if (nchunks > 1) {
        lease_b.reset(new Lease(e));
        W[1] = lease_b->ws;
        if (!W[1]->st_compute) HIP_TRY(hipStreamCreate(&W[1]->st_compute));
    }
    const int nout = nchunks > 2 ? 3 : (int)nchunks;
    int64_t max_units = 0, max_docs = 0;
    for (int64_t k = 0; k < nchunks; ++k) {
        max_units = std::max(max_units, offs[cut[(size_t)k + 1]] - offs[cut[(size_t)k]]);
        max_docs = std::max(max_docs, cut[(size_t)k + 1] - cut[(size_t)k]);
    }
    for (int q = 0; q < (nchunks > 1 ? 2 : 1); ++q) {
        if (u16) {
            Workspace::U16Stage& U = ws->u16[q];
            const int64_t nw = max_units / 64 + 1, nt = u16_tiles(max_units), nblk = (nt + kScanBlock - 1) / kScanBlock;
            HIP_TRY(U.units.ensure((size_t)(max_units + 64) * 2, acc));
            HIP_TRY(U.offs.ensure((size_t)(max_docs + 1) * 8, acc));
            HIP_TRY(U.docbits.ensure((size_t)(nw + 8) * 8, acc));
            HIP_TRY(U.grp.ensure((size_t)nt * 64 * 4, acc));
            HIP_TRY(U.tsum.ensure((size_t)nt * 4, acc));
            HIP_TRY(U.tbase.ensure((size_t)nt * 8, acc));
            HIP_TRY(U.bsum.ensure((size_t)(nblk + 1) * 8, acc));
            HIP_TRY(U.counters.ensure(64, acc));
            HIP_TRY(U.boffs.ensure((size_t)(max_docs + 1) * 8, acc));
            if (!U.h) HIP_TRY(hipHostMalloc((void**)&U.h, 64, 0));
        } else {
            HIP_TRY(ws->s_bytes[q].ensure((size_t)max_units + 64, acc));
            HIP_TRY(ws->s_offs[q].ensure((size_t)(max_docs + 1) * 8, acc));
        }
    }
    if (!direct_out) for (int o = 0; o < nout; ++o) {
        HIP_TRY(ws->s_out[o].ensure((size_t)std::max<int64_t>(std::min<int64_t>(out_cap, (u16 ? 3 : 1) * max_units), 1) * 4, acc));      // (a token is at least one byte, a code unit at most three)
        HIP_TRY(ws->s_outoffs[o].ensure((size_t)(max_docs + 1) * 8, acc));
    }
    // the input of chunk k, on its way to the device (stream st_in); for UTF-16 also its document marks, the UTF-8 length of every unit and their scan
    auto stage_in = [&](int64_t k) -> tkz_status {
        const int q = (int)(k & 1);
        const int64_t d0 = cut[(size_t)k], d1 = cut[(size_t)k + 1], u0 = offs[d0], nu = offs[d1] - u0, nd = d1 - d0;
        Launch L{ws->st_in, nullptr, ws};
        if (u16) {
            Workspace::U16Stage& U = ws->u16[q];
            if (nu) HIP_TRY(hipMemcpyAsync(U.units.p, units + u0, (size_t)nu * 2, hipMemcpyHostToDevice, ws->st_in));
            HIP_TRY(hipMemcpyAsync(U.offs.p, offs + d0, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, ws->st_in));
            if (u0) launch_rebase(L, U.offs.as<int64_t>(), nd + 1, u0);
            const int64_t nw = nu / 64 + 1, nt = u16_tiles(nu);
            HIP_TRY(hipMemsetAsync(U.counters.p, 0, 64, ws->st_in));
            HIP_TRY(hipMemsetAsync(U.docbits.p, 0, (size_t)(nw + 8) * 8, ws->st_in));
            launch_docmark(L, U.offs.as<int64_t>(), nd, nu, U.docbits.as<uint64_t>(), U.counters.as<int32_t>());
            launch_u16_len(L, U.units.as<uint16_t>(), nu, U.docbits.as<uint64_t>(), nt, U.grp.as<int32_t>(), U.tsum.as<int32_t>());
            launch_scan(L, U.tsum.as<int32_t>(), nt, U.bsum.as<int64_t>(), U.tbase.as<int64_t>(), reinterpret_cast<int64_t*>(U.counters.as<char>() + 8), -1);
            HIP_TRY(hipMemcpyAsync(U.h, U.counters.p, 16, hipMemcpyDeviceToHost, ws->st_in));
        } else {
            if (nu) HIP_TRY(hipMemcpyAsync(ws->s_bytes[q].p, bytes + u0, (size_t)nu, hipMemcpyHostToDevice, ws->st_in));
            HIP_TRY(hipMemcpyAsync(ws->s_offs[q].p, offs + d0, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, ws->st_in));
            if (u0) launch_rebase(L, ws->s_offs[q].as<int64_t>(), nd + 1, u0);
        }
        HIP_TRY(hipEventRecord(ws->ev_in[q], ws->st_in));
        return TKZ_OK;
    };
    std::vector<int64_t> tok_base((size_t)nchunks + 1, 0);
    bool over = false;                                       // out_cap exceeded: the remaining chunks are only counted
    // downloads by copy engine: page-locked results whose device-side address is their host address (hipHostMalloc, tkz_host_alloc, torch's pinned tensors --
    // not memory registered after the fact, which the HSA runtime knows under another address)
    if (ws->sdma_state == 0) {
        bool ok = sdma_available(e->device);
        for (int o = 0; o < 3 && ok; ++o) ok = sdma_signal_create(&ws->sig_out[o]) && sdma_signal_create(&ws->sig_outoffs[o]);
        ws->sdma_state = ok ? 1 : -1;
    }
    bool sdma_out = ws->sdma_state == 1 && !direct_out && pin_out && dv_ooffs == (void*)out_offsets && (out_cap == 0 || dv_ids == (void*)out_ids);
    bool sig_pending[3] = {false, false, false}, sigo_pending[3] = {false, false, false}, ev_pending[3] = {false, false, false};
    auto wait_engine = [&](int o) {          // the copies of output set o that went by engine have arrived
        bool ok = true;
        if (sig_pending[o]) { sig_pending[o] = false; ok = sdma_signal_wait(ws->sig_out[o]) && ok; }
        if (sigo_pending[o]) { sigo_pending[o] = false; ok = sdma_signal_wait(ws->sig_outoffs[o]) && ok; }
        return ok;
    };
    tkz_status first_err = TKZ_OK;
    std::string first_msg;
    auto note = [&](tkz_status s) { if (first_err == TKZ_OK) { first_err = s; first_msg = g_err; } return s; };
    // (one small chunk on page-locked buffers: no upload stream at all -- the launch sequence starts with k_ingest, which fetches the text itself)
    const bool ingest_in = direct_out && !u16 && pin_in && total > 0 && (reinterpret_cast<uintptr_t>(dv_in) & 15) == 0;
    const IngestSrc ingest_src{static_cast<const uint8_t*>(dv_in), static_cast<const int64_t*>(dv_offs)};
    // what encode_device was given for the chunk in flight on W[q] (kCallEnd is handed the same)
    struct InFlight { const uint8_t* cb; const int64_t* co; int64_t cbytes, nd, cap; int32_t* dst_ids; int64_t* dst_offs; } fl[2] = {};
    // (inside the lambdas a failed runtime call is RECORDED, never returned from the function: the upload of a later chunk may still be reading the caller's text and the
    //  download of an earlier one writing the caller's ids, and the caller is free to release both the moment it sees the error; the drain behind the loop always runs first)
#define CHUNK_TRY(expr) { const hipError_t e_ = (expr); if (e_ != hipSuccess) return note(fail(TKZ_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); }
    // enqueue chunk k's launch sequence behind its upload (one chunk: run it whole -- phase kCallWhole)
    auto begin_chunk = [&](int64_t k, int phase, int64_t* tokens) -> tkz_status {
        const int q = (int)(k & 1), o = (int)(k % nout);
        Workspace* w = W[q];
        const int64_t d0 = cut[(size_t)k], d1 = cut[(size_t)k + 1], nu = offs[d1] - offs[d0], nd = d1 - d0;
        InFlight& F = fl[q];
        F.nd = nd;
        if (u16) {
            // the UTF-8 size of the chunk is known once its length pass is through (the host needs it: the launch shapes of the encode path)
            Workspace::U16Stage& U = ws->u16[q];
            CHUNK_TRY(hipEventSynchronize(ws->ev_in[q]));
            if (U.h->err & kErrOffsets) return note(fail(TKZ_E_ARG, "document offsets must start at 0, be non-decreasing and end at the unit count"));
            F.cbytes = U.h->grand;
            CHUNK_TRY(ws->u_bytes[q].ensure((size_t)F.cbytes + 64, acc));
            Launch L{w->st_compute, nullptr, w};
            launch_u16_write(L, U.units.as<uint16_t>(), nu, U.docbits.as<uint64_t>(), u16_tiles(nu), U.tbase.as<int64_t>(), ws->u_bytes[q].as<uint8_t>(),
                             U.offs.as<int64_t>(), nd, U.grp.as<int32_t>(), reinterpret_cast<int64_t*>(U.counters.as<char>() + 8), U.boffs.as<int64_t>());
            F.cb = ws->u_bytes[q].as<uint8_t>(); F.co = U.boffs.as<int64_t>();
        } else {
            if (!ingest_in) CHUNK_TRY(hipStreamWaitEvent(w->st_compute, ws->ev_in[q], 0));
            F.cb = ws->s_bytes[q].as<uint8_t>(); F.co = ws->s_offs[q].as<int64_t>(); F.cbytes = nu;
        }
        // the download of chunk k - nout has left this chunk's output set
        if (!wait_engine(o)) return note(fail(TKZ_E_DEVICE, "a copy engine reported an error for a download"));
        if (ev_pending[o]) { ev_pending[o] = false; CHUNK_TRY(hipStreamWaitEvent(w->st_compute, ws->ev_out[o], 0)); }
        F.dst_ids = direct_out ? static_cast<int32_t*>(dv_ids) : ws->s_out[o].as<int32_t>();
        F.dst_offs = direct_out ? static_cast<int64_t*>(dv_ooffs) : ws->s_outoffs[o].as<int64_t>();
        // (how much of out_cap the chunks before leave is not known yet when a chunk is begun: the staging set holds a chunk's ids whatever their number, and the
        //  sum is checked when the chunk ends)
        F.cap = over ? 0 : std::min<int64_t>(out_cap, F.cbytes);
        return encode_device(e, w, F.cb, F.co, F.nd, F.cbytes, F.dst_ids, F.cap, F.dst_offs, w->st_compute, true, nullptr, tokens, nullptr, phase, nullptr,
                             ingest_in ? &ingest_src : nullptr);
    };
    auto end_chunk = [&](int64_t k, int64_t* tokens) -> tkz_status {          // (returns when the chunk's stream has drained)
        const InFlight& F = fl[k & 1];
        Workspace* w = W[k & 1];
        return encode_device(e, w, F.cb, F.co, F.nd, F.cbytes, F.dst_ids, F.cap, F.dst_offs, w->st_compute, true, nullptr, tokens, nullptr, kCallEnd, nullptr, nullptr);
    };
    // The download of chunk k.  The runtime's D2H copy of page-locked memory is a blit KERNEL and other kernels make no progress beside it (traced, round 5; a
    // small-grid download kernel of our own stalled their first stores until its PCIe writes had drained).  Page-locked results therefore leave on a COPY ENGINE
    // of their own, named through the HSA runtime (tkz_sdma.h; tools/sdma_probe.hip: 56 GB/s beside a store-heavy kernel, and beside the runtime's upload
    // when the engines differ).  The kernels of the chunk have completed (end_chunk returned), which is all such a copy waits for.
    auto download = [&](int64_t k, int64_t tokens) -> tkz_status {
        const int o = (int)(k % nout);
        const int64_t d0 = cut[(size_t)k], nd = cut[(size_t)k + 1] - d0;
        bool ids_sent = tokens == 0, offs_sent = false;
        const size_t nb_ids = (size_t)tokens * 4, nb_offs = (size_t)(nd + 1) * 8;
        // (an engine that refuses a copy is not asked again: that copy and the rest of the call go through the runtime)
        auto by_engine = [&](void* dst, const void* src, size_t nb, SdmaSignal sg, bool* pending) {
            if (!sdma_out) return false;
            sdma_signal_arm(sg, 1);
            if (sdma_copy_d2h(e->device, dst, src, nb, sg)) { *pending = true; ws->engine_downloads.fetch_add(1, std::memory_order_relaxed); return true; }
            sdma_signal_arm(sg, 0); sdma_out = false; ws->sdma_state = -1;
            return false;
        };
        if (tokens) ids_sent = by_engine(out_ids + tok_base[(size_t)k], ws->s_out[o].p, nb_ids, ws->sig_out[o], &sig_pending[o]);
        offs_sent = by_engine(out_offsets + d0, ws->s_outoffs[o].p, nb_offs, ws->sig_outoffs[o], &sigo_pending[o]);
        if (!ids_sent || !offs_sent) {
            if (!ids_sent) CHUNK_TRY(hipMemcpyAsync(out_ids + tok_base[(size_t)k], ws->s_out[o].p, nb_ids, hipMemcpyDeviceToHost, ws->st_out));
            if (!offs_sent) CHUNK_TRY(hipMemcpyAsync(out_offsets + d0, ws->s_outoffs[o].p, nb_offs, hipMemcpyDeviceToHost, ws->st_out));
            CHUNK_TRY(hipEventRecord(ws->ev_out[o], ws->st_out));
            ev_pending[o] = true;
        }
        return TKZ_OK;
    };
    int64_t begun = 0, ended = 0;            // chunks [ended, begun) are in flight
    // ($TKZ_TRACE_HOST: the host's side of the pipeline on stderr -- microseconds since the call began at which each step RETURNED; development)
    static const bool kTraceHost = getenv("TKZ_TRACE_HOST") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    std::vector<std::pair<std::string, double>> stamps;
    auto stamp = [&](const char* what, int64_t k) {
        if (kTraceHost) stamps.emplace_back(std::string(what) + " " + std::to_string(k), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count());
    };
    auto finish = [&](int64_t k, tkz_status cs, int64_t tokens) -> bool {      // chunk k has ended with status cs: its place in the output, its download; false: stop
        tok_base[(size_t)k + 1] = tok_base[(size_t)k] + tokens;
        if (cs == TKZ_E_CAPACITY || (cs == TKZ_OK && tok_base[(size_t)k + 1] > out_cap)) { over = true; return true; }
        if (cs != TKZ_OK) { note(cs); return false; }
        if (!over && !direct_out && download(k, tokens) != TKZ_OK) return false;
        return true;
    };
    if (nchunks == 1) {
        int64_t tokens = 0;
        const tkz_status ss = ingest_in ? TKZ_OK : stage_in(0);
        if (ss == TKZ_OK) { const tkz_status cs = begin_chunk(0, kCallWhole, &tokens); if (first_err == TKZ_OK) (void)finish(0, cs, tokens); }
        else note(ss);
    } else {
        // (upload 1 is issued BEHIND launch sequence 0, although that starts it ~130 us late: an upload stream's k_rebase kernel waits in a hardware queue for its copy, and
        //  the launch sequence of chunk 0 enqueued behind it -- the runtime maps its streams onto a few hardware queues -- waited with it: 16 MB 884 -> 1,040 us)
        for (int64_t k = 0; k < 2 && first_err == TKZ_OK; ++k) {
            { const tkz_status ss = stage_in(k); if (ss != TKZ_OK) { note(ss); break; } }
            const tkz_status bs = begin_chunk(k, kCallBegin, nullptr);
            if (bs != TKZ_OK) { note(bs); break; }
            ++begun;
            stamp("begun", k);
        }
        for (int64_t k = 0; k < nchunks && first_err == TKZ_OK; ++k) {
            int64_t tokens = 0;
            const tkz_status cs = end_chunk(k, &tokens);
            ++ended;
            stamp("ended", k);
            if (!finish(k, cs, tokens)) break;
            stamp("download issued", k);
            if (k + 2 < nchunks) {         // (its input set and its workspace are chunk k's: free now)
                tkz_status bs = stage_in(k + 2);
                if (bs == TKZ_OK) bs = begin_chunk(k + 2, kCallBegin, nullptr);
                if (bs != TKZ_OK) { note(bs); break; }
                ++begun;
                stamp("begun", k + 2);
            }
        }
        // (a chunk that was begun is always ended: its workspace may hold the encoder's learning slot, and its kernels write the staging sets)
        for (; ended < begun; ++ended) { const std::string keep = g_err; int64_t t = 0; (void)end_chunk(ended, &t); g_err = keep; }
    }
#undef CHUNK_TRY
    (void)hipStreamSynchronize(ws->st_in);      // (an error of the runtime here is an error of the copies above: reported by them or by the next call)
    (void)hipStreamSynchronize(ws->st_out);
    for (int o = 0; o < 3; ++o)
        if (!wait_engine(o)) note(fail(TKZ_E_DEVICE, "a copy engine reported an error for a download"));
    if (kTraceHost && nchunks > 1) {
        stamp("downloads arrived", nchunks);
        std::string line = "[tkz host trace] " + std::to_string(nchunks) + " chunks:";
        for (const auto& sp : stamps) { char b[96]; snprintf(b, sizeof b, " %s @%.0f", sp.first.c_str(), sp.second); line += b; }
        fprintf(stderr, "%s\n", line.c_str());
    }
    if (first_err != TKZ_OK) return fail(first_err, first_msg);
    if (needed) *needed = tok_base[(size_t)nchunks];
    if (over) return fail(TKZ_E_CAPACITY, "output capacity too small");
    // the offsets came back relative to their chunk: add the chunk's token base (chunk 0 needs nothing; the shared boundary entry
    // of two chunks was written by the later one as 0 and gets that chunk's base, which is what the earlier chunk's last entry was)
    for (int64_t k = 1; k < nchunks; ++k) {
        const int64_t tb = tok_base[(size_t)k];
        for (int64_t d = cut[(size_t)k]; d < cut[(size_t)k + 1]; ++d) out_offsets[d] += tb;
    }
    out_offsets[n_docs] = tok_base[(size_t)nchunks];
    return TKZ_OK;
}

const char* const kRegexP1 = "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
const char* const kRegexCl100k =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";
#define TKZ_O2_SUFFIX "(?:'s|'S|'t|'T|'re|'RE|'Re|'eR|'ve|'VE|'vE|'Ve|'m|'M|'ll|'lL|'Ll|'LL|'d|'D)?"
const char* const kRegexO200k =
    "[^\r\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+" TKZ_O2_SUFFIX
    "|[^\r\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*" TKZ_O2_SUFFIX
    "|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";

}  // namespace

extern "C" {

const char* tkz_last_error(void) { return g_err.c_str(); }

tkz_status tkz_vocab_from_tiktoken(const uint8_t* file, size_t n, tkz_vocab** out) {
    if (!out || (!file && n)) return fail(TKZ_E_ARG, "null argument");
    *out = nullptr;
    tkz_vocab* v = new tkz_vocab();
    std::string msg;
    int r = tkz::parse_tiktoken(file, n, &v->v, &msg);
    if (r == TKZ_OK) r = tkz::build_tables(&v->v, &msg);
    if (r != TKZ_OK) { delete v; return fail((tkz_status)r, msg); }
    *out = v;
    return TKZ_OK;
}
void tkz_vocab_destroy(tkz_vocab* v) { delete v; }
int64_t tkz_vocab_size(const tkz_vocab* v) { return v ? (int64_t)v->v.keys.size() : 0; }
int32_t tkz_vocab_max_key_len(const tkz_vocab* v) { return v ? v->v.max_key_len : 0; }
int64_t tkz_vocab_pair_table_entries(const tkz_vocab* v) { return v ? v->v.pair_entries : 0; }
int64_t tkz_vocab_table_bytes(const tkz_vocab* v, int32_t which) {
    if (!v) return 0;
    const tkz::Vocab& V = v->v;
    const int64_t b[5] = {(int64_t)(V.short_slots.size() * sizeof(TkzShortSlot)), (int64_t)(V.mid_slots.size() * sizeof(TkzMidSlot)),
                          (int64_t)(V.long_slots.size() * sizeof(TkzLongSlot) + V.long_blob.size()), (int64_t)(V.pair_slots.size() * sizeof(TkzPairSlot)),
                          (int64_t)((V.byte_rank.size() + V.bytepair_rank.size()) * sizeof(int32_t))};
    if (which == -1) return b[0] + b[1] + b[2] + b[3] + b[4];
    return which >= 0 && which < 5 ? b[which] : 0;
}
int32_t tkz_vocab_rank(const tkz_vocab* v, const uint8_t* key, int32_t len) {
    if (!v || len < 0 || (!key && len)) return -1;
    int32_t r;
    return v->v.lookup(std::string(reinterpret_cast<const char*>(key), (size_t)len), &r) ? r : -1;
}

void tkz_unicode_classes(uint32_t first, int32_t n, uint8_t* out) {
    const uint8_t* ucd = tkz::bmp_class_table().data();
    for (int32_t i = 0; out && i < n; ++i) out[i] = tkz_supp_class(ucd, first + (uint32_t)i);
}

tkz_status tkz_encoder_unicode_classes(tkz_encoder* e, uint32_t first, int32_t n, uint8_t* out) {
    // the table image as the DEVICE holds it (downloaded, then read as the kernels read it): a check of the upload, not of the host copy
    if (!out || n < 0) return fail(TKZ_E_ARG, "bad argument");
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    std::vector<uint8_t> img(e->bmp_image_bytes);
    HIP_TRY(hipMemcpy(img.data(), e->t_bmp.p, img.size(), hipMemcpyDeviceToHost));
    for (int32_t i = 0; i < n; ++i) out[i] = tkz_supp_class(img.data(), first + (uint32_t)i);
    return TKZ_OK;
}

tkz_status tkz_encoder_set_unicode_classes(tkz_encoder* e, const uint8_t* classes, int64_t n_code_points) {
    // The split regexes are whatever the HOST's regex engine makes of \p{L}, \p{N}, \s: the reference compiles them with the running process's
    // System.Text.RegularExpressions (TikTokenizer.cs:77), whose Unicode data is the runtime's (13.0 under net6.0, 15.0 under .NET 8).  A host hands its
    // own classification over here; classes == NULL puts the built-in Unicode 13.0 table back.
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (classes && n_code_points != 0x10000 && n_code_points != 0x110000) return fail(TKZ_E_ARG, "n_code_points must be 65536 (the BMP: code units) or 1114112 (every code point)");
    std::vector<uint8_t> img = tkz::bmp_class_table();                 // [TKZ_UCD_DIRECT direct classes][n][n x {first, last | class << 24}]
    if (classes) {
        for (int64_t cp = 0; cp < n_code_points; ++cp) if (classes[cp] > 8) return fail(TKZ_E_ARG, "a class code above 8 (0 other, 1 Lu, 2 Ll, 3 Lt, 4 Lm, 5 Lo, 6 M, 7 N, 8 white space)");
        // (the ASCII range keeps its classes: the scanners' fast paths know them; and a surrogate code unit is never a letter, digit or space)
        const int64_t direct = std::min<int64_t>(n_code_points, (int64_t)TKZ_UCD_DIRECT);
        for (int64_t cp = 128; cp < direct; ++cp) img[(size_t)cp] = (cp >= 0xD800 && cp <= 0xDFFF) ? 0 : classes[cp];
        if (n_code_points > (int64_t)TKZ_UCD_DIRECT) {
            std::vector<uint32_t> hi;
            for (int64_t cp = TKZ_UCD_DIRECT; cp < n_code_points;) {
                if (!classes[cp]) { ++cp; continue; }
                int64_t end = cp;
                while (end + 1 < n_code_points && classes[end + 1] == classes[cp]) ++end;
                hi.push_back((uint32_t)cp); hi.push_back((uint32_t)end | ((uint32_t)classes[cp] << 24));
                cp = end + 1;
            }
            if (hi.size() / 2 > 4096) return fail(TKZ_E_UNSUPPORTED, "more than 4096 classified ranges above U+3FFFF");
            const uint32_t n = (uint32_t)(hi.size() / 2);
            hi.insert(hi.begin(), n);
            img.resize(TKZ_UCD_DIRECT + hi.size() * 4);
            memcpy(img.data() + TKZ_UCD_DIRECT, hi.data(), hi.size() * 4);
        }
    }
    std::lock_guard<std::mutex> lock(e->mu);
    for (Workspace* w : e->pool) if (w->busy) return fail(TKZ_E_ARG, "the class table can only be replaced while no call of this encoder is in flight");
    if (hipDeviceSynchronize() != hipSuccess) return fail(TKZ_E_DEVICE, "hipDeviceSynchronize");
    const hipError_t h = upload(e->t_bmp, img, &e->bytes_allocated);
    if (h != hipSuccess) return fail(TKZ_E_DEVICE, std::string("class table upload: ") + hipGetErrorString(h));
    e->bmp_image_bytes = img.size();
    e->T.bmp_class = e->t_bmp.as<uint8_t>();
    return TKZ_OK;
}

tkz_status tkz_pattern_from_regex_engine(const char* regex_utf8, int32_t engine, int32_t* pattern_out) {
    if (!regex_utf8 || !pattern_out) return fail(TKZ_E_ARG, "null argument");
    if (engine != TKZ_ENGINE_DOTNET && engine != TKZ_ENGINE_ECMASCRIPT) return fail(TKZ_E_ARG, "unknown regex engine id");
    int32_t p = 0;
    if (!strcmp(regex_utf8, kRegexP1)) p = TKZ_PATTERN_P1;
    else if (!strcmp(regex_utf8, kRegexCl100k)) p = TKZ_PATTERN_CL100K;
    else if (!strcmp(regex_utf8, kRegexO200k)) p = engine == TKZ_ENGINE_ECMASCRIPT ? TKZ_PATTERN_O200K : TKZ_PATTERN_O200K_DOTNET;
    else return fail(TKZ_E_UNSUPPORTED, "only the three split patterns the reference defines are implemented (pattern 1, cl100k_base, o200k_base)");
    if (engine == TKZ_ENGINE_ECMASCRIPT && p != TKZ_PATTERN_O200K)
        return fail(TKZ_E_UNSUPPORTED, "pattern 1 and cl100k_base are implemented with the semantics of the C# reference's engine only (UTF-16 code units, .NET \\s)");
    *pattern_out = p;
    return TKZ_OK;
}
tkz_status tkz_pattern_from_regex(const char* regex_utf8, int32_t* pattern_out) {
    return tkz_pattern_from_regex_engine(regex_utf8, TKZ_ENGINE_DOTNET, pattern_out);
}

tkz_status tkz_encoder_create(const tkz_vocab* v, int32_t pattern, int32_t device, tkz_encoder** out) {
    if (!out || !v) return fail(TKZ_E_ARG, "null argument");
    *out = nullptr;
    if (pattern < TKZ_PATTERN_P1 || pattern > TKZ_PATTERN_O200K_DOTNET) return fail(TKZ_E_UNSUPPORTED, "unknown pattern id");
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev <= 0) return fail(TKZ_E_NO_DEVICE, "no HIP device available (libtkz has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TKZ_E_ARG, "device index out of range");
    DeviceScope scope;
    HIP_TRY(scope.enter(device));
    tkz_encoder* e = new tkz_encoder();
    e->device = device; e->pattern = pattern; e->max_key_len = v->v.max_key_len;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) e->small_ok = prop.sharedMemPerBlock >= (size_t)tkz::kSmallLdsBytesNeeded; }
    int64_t* acc = &e->bytes_allocated;
    const tkz::Vocab& V = v->v;
    hipError_t h = hipSuccess;
    // SHORT and MID in ONE allocation (k_probe addresses both from one uniform base with 32-bit offsets)
    const size_t short_bytes = V.short_slots.size() * sizeof(TkzShortSlot), mid_bytes = V.mid_slots.size() * sizeof(TkzMidSlot);
    if (h == hipSuccess) h = e->t_short.ensure(std::max<size_t>(64, short_bytes + mid_bytes), acc);
    if (h == hipSuccess && short_bytes) h = hipMemcpy(e->t_short.p, V.short_slots.data(), short_bytes, hipMemcpyHostToDevice);
    if (h == hipSuccess && mid_bytes) h = hipMemcpy(e->t_short.as<char>() + short_bytes, V.mid_slots.data(), mid_bytes, hipMemcpyHostToDevice);
    if (h == hipSuccess) h = upload(e->t_long, V.long_slots, acc);
    if (h == hipSuccess) h = upload(e->t_blob, V.long_blob, acc);
    if (h == hipSuccess) h = upload(e->t_pair, V.pair_slots, acc);
    if (h == hipSuccess) h = upload(e->t_byte, V.byte_rank, acc);
    if (h == hipSuccess) h = upload(e->t_bpair, V.bytepair_rank, acc);
    if (h == hipSuccess) h = upload(e->t_bmp, tkz::bmp_class_table(), acc);
    e->bmp_image_bytes = tkz::bmp_class_table().size();
    // {n_docs, n_bytes, n_tokens} of the last batch (tkz_encoder_counts_device): allocated once, here -- never from a call in flight
    if (h == hipSuccess) h = e->t_counts3.ensure(32, acc);
    if (h == hipSuccess) h = hipMemset(e->t_counts3.p, 0, 32);
    if (h != hipSuccess) { tkz_encoder_destroy(e); return fail(TKZ_E_DEVICE, std::string("table upload: ") + hipGetErrorString(h)); }
    e->T.short_slots = e->t_short.as<TkzShortSlot>(); e->T.short_nb = (uint32_t)(V.short_slots.size() / 2); e->T.short_seed = V.short_seed;
    e->T.mid_slots = reinterpret_cast<const TkzMidSlot*>(e->t_short.as<char>() + short_bytes); e->T.mid_ns = (uint32_t)V.mid_slots.size(); e->T.mid_seed = V.mid_seed;
    e->T.long_slots = e->t_long.as<TkzLongSlot>();    e->T.long_mask = (uint32_t)V.long_slots.size() - 1;
    e->T.long_blob = e->t_blob.as<uint8_t>();
    e->T.pair_slots = e->t_pair.as<TkzPairSlot>();    e->T.pair_n = (uint32_t)V.pair_slots.size(); e->T.pair_seed = V.pair_seed; e->T.pair_compact = V.pair_compact ? 1u : 0u;
    e->T.byte_rank = e->t_byte.as<int32_t>();
    e->T.bytepair_rank = e->t_bpair.as<int32_t>();
    e->T.bmp_class = e->t_bmp.as<uint8_t>();
    {   // the piece memo: 2^19 slots of 32 bytes (16 MB), empty
// (measured on the bench workload, k_merge_short per 5 GB: 2^15 slots 8.2 ms, 2^16 7.7, 2^17 7.1, 2^18 6.0, 2^19 5.4, 2^20 5.2, 2^22 8.4 -- the
//  hit rate grows with the table until it stops fitting the Infinity Cache beside everything else)
#ifndef TKZ_MEMO_SLOTS_LOG2
#define TKZ_MEMO_SLOTS_LOG2 19
#endif
        // ($TKZ_MEMO_SLOTS_LOG2 in the environment, 10..22: the tests' handle on it -- a memo that a few thousand pieces fill)
        uint32_t kMemoSlots = 1u << TKZ_MEMO_SLOTS_LOG2;
        { const char* v = getenv("TKZ_MEMO_SLOTS_LOG2"); const int n = v ? atoi(v) : 0; if (n >= 10 && n <= 22) kMemoSlots = 1u << n; }
        h = e->t_memo.ensure(size_t(kMemoSlots) * sizeof(TkzMemoSlot), acc);
        if (h == hipSuccess) h = hipMemset(e->t_memo.p, 0, size_t(kMemoSlots) * sizeof(TkzMemoSlot));
        if (h != hipSuccess) { tkz_encoder_destroy(e); return fail(TKZ_E_OUT_OF_MEMORY, std::string("piece memo: ") + hipGetErrorString(h)); }
        e->memo_slots = kMemoSlots;
        e->T.memo = e->t_memo.as<TkzMemoSlot>(); e->T.memo_n = kMemoSlots;
    }
    e->T.memo_hits = nullptr; e->T.memo_hits_sparse = 0; e->T.promo = nullptr; e->T.promo_n = 0;
    e->short_slots_n = (uint32_t)V.short_slots.size(); e->mid_slots_n = (uint32_t)V.mid_slots.size();
    e->T.max_key_len = V.max_key_len;
    e->T.pattern = pattern;
    e->T.max_rank = 0;
    for (int32_t r : V.ranks) e->T.max_rank = std::max(e->T.max_rank, r);
    e->dec_vocab.reserve(V.keys.size());
    for (size_t i = 0; i < V.keys.size(); ++i) e->dec_vocab.emplace_back(V.ranks[i], V.keys[i]);
    { const tkz_status ds = build_decode_table(e); if (ds != TKZ_OK) { tkz_encoder_destroy(e); return ds; } }
    *out = e;
    return TKZ_OK;
}

namespace {
void destroy_now(tkz_encoder* e);
}
void tkz_encoder_destroy(tkz_encoder* e) {
    if (!e) return;
    {   // handles of tkz_encode_batch_device_begin still outstanding: their _end calls need the encoder and its workspaces -- the last of
        // them frees it (and reports TKZ_E_ARG: the results of a batch whose encoder was destroyed under it are not to be trusted).
        // `destroyed` is set under the lock in either case, so that a _begin racing with this call is refused instead of leasing a
        // workspace of an encoder that is being deleted.
        std::lock_guard<std::mutex> lock(e->mu);
        if (e->destroyed) return;                       // (a second destroy while the first is deferred)
        e->destroyed = true;
        if (e->pending > 0) return;
    }
    destroy_now(e);
}
namespace {
void destroy_now(tkz_encoder* e) {
    join_promotion(e);
    DeviceScope scope;
    (void)scope.enter(e->device);
    DevBuf* bufs[] = {&e->t_short, &e->t_mid, &e->t_long, &e->t_blob, &e->t_pair, &e->t_byte, &e->t_bpair, &e->t_bmp, &e->t_counts3, &e->t_memo, &e->t_stats, &e->t_decoff, &e->t_decblob, &e->t_decids,
                      &e->t_memo_hits, &e->t_promo, &e->t_long_log};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : e->retired) b.release();
    for (Workspace* w : e->pool) { w->release_all(); delete w; }
    delete e;
}
}  // namespace
int32_t tkz_encoder_device(const tkz_encoder* e) { return e ? e->device : -1; }
const int64_t* tkz_encoder_counts_device(const tkz_encoder* e) { return e ? e->t_counts3.as<int64_t>() : nullptr; }

tkz_status tkz_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(TKZ_E_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TKZ_E_NO_DEVICE, "no HIP device available");
    // (portable: page-locked for EVERY device of the process, whichever is current here -- a host with encoders on several GPUs hands the same
    //  buffers to any of them)
    const hipError_t r = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable);
    if (r != hipSuccess) { *out = nullptr; return fail(TKZ_E_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(r)); }
    return TKZ_OK;
}
void tkz_host_free(void* p) { if (p) (void)hipHostFree(p); }

tkz_status tkz_encode_batch_utf8(tkz_encoder* e, const uint8_t* bytes, const int64_t* doc_offsets, int64_t n_docs,
                                 int32_t* out_ids, int64_t out_cap, int64_t* out_offsets, int64_t* needed) {
    if (!out_offsets || (out_cap > 0 && !out_ids)) return fail(TKZ_E_ARG, "null output buffer");
    return encode_host(e, bytes, nullptr, doc_offsets, n_docs, out_ids, out_cap, out_offsets, needed, true, nullptr);
}

tkz_status tkz_encode_batch_device(tkz_encoder* e, const uint8_t* d_bytes, const int64_t* d_doc_offsets, int64_t n_docs,
                                   int64_t total_bytes, int32_t* d_out_ids, int64_t out_cap, int64_t* d_out_offsets,
                                   void* hip_stream, int64_t* total_tokens) {
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (!d_doc_offsets || !d_out_offsets || (total_bytes > 0 && (!d_bytes || !d_out_ids))) return fail(TKZ_E_ARG, "null device buffer");
    if (reinterpret_cast<uintptr_t>(d_bytes) & 15) return fail(TKZ_E_ARG, "d_bytes must be 16-byte aligned");
    Lease lease(e);
    Workspace* ws = lease.ws;
    return encode_device(e, ws, d_bytes, d_doc_offsets, n_docs, total_bytes, d_out_ids, out_cap, d_out_offsets,
                         static_cast<hipStream_t>(hip_stream), true, nullptr, total_tokens);
}

// The same in two halves: _begin enqueues the batch on the stream and returns, _end waits for it and reports as tkz_encode_batch_device
// does (a batch that needs a larger buffer than the first attempt had is run again inside _end).  The call keeps a workspace of the
// encoder from _begin to _end: several batches can be in flight, on one stream or on several.
struct tkz_pending {
    tkz_encoder* e; Lease* lease;
    const uint8_t* d_bytes; const int64_t* d_offs; int64_t n_docs, total; int32_t* d_out; int64_t out_cap; int64_t* d_out_offs; hipStream_t stream;
    int64_t* d_counts3;                    // the caller's block for this batch's {n_docs, n_bytes, n_tokens} (may be null)
};
tkz_status tkz_encode_batch_device_begin_counts(tkz_encoder* e, const uint8_t* d_bytes, const int64_t* d_doc_offsets, int64_t n_docs,
                                                int64_t total_bytes, int32_t* d_out_ids, int64_t out_cap, int64_t* d_out_offsets,
                                                void* hip_stream, int64_t* d_counts3, tkz_pending** pending) {
    if (!pending) return fail(TKZ_E_ARG, "null pending");
    *pending = nullptr;
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (!d_doc_offsets || !d_out_offsets || (total_bytes > 0 && (!d_bytes || !d_out_ids))) return fail(TKZ_E_ARG, "null device buffer");
    if (reinterpret_cast<uintptr_t>(d_bytes) & 15) return fail(TKZ_E_ARG, "d_bytes must be 16-byte aligned");
    { std::lock_guard<std::mutex> lock(e->mu); if (e->destroyed) return fail(TKZ_E_ARG, "encoder destroyed"); ++e->pending; }
    tkz_pending* p = new tkz_pending{e, new Lease(e), d_bytes, d_doc_offsets, n_docs, total_bytes, d_out_ids, out_cap, d_out_offsets, static_cast<hipStream_t>(hip_stream), d_counts3};
    st = encode_device(e, p->lease->ws, d_bytes, d_doc_offsets, n_docs, total_bytes, d_out_ids, out_cap, d_out_offsets, p->stream, true, nullptr, nullptr, nullptr, kCallBegin, d_counts3);
    if (st != TKZ_OK) {
        (void)hipStreamSynchronize(p->stream);
        delete p->lease; delete p;
        const std::string msg = g_err;
        bool last;
        { std::lock_guard<std::mutex> lock(e->mu); last = --e->pending == 0 && e->destroyed; }
        if (last) destroy_now(e);                       // (destroy arrived while this _begin was running and no other handle is left)
        g_err = msg;
        return st;
    }
    *pending = p;
    return TKZ_OK;
}
tkz_status tkz_encode_batch_device_begin(tkz_encoder* e, const uint8_t* d_bytes, const int64_t* d_doc_offsets, int64_t n_docs,
                                         int64_t total_bytes, int32_t* d_out_ids, int64_t out_cap, int64_t* d_out_offsets,
                                         void* hip_stream, tkz_pending** pending) {
    return tkz_encode_batch_device_begin_counts(e, d_bytes, d_doc_offsets, n_docs, total_bytes, d_out_ids, out_cap, d_out_offsets, hip_stream, nullptr, pending);
}
const int64_t* tkz_pending_counts_device(const tkz_pending* p) {
    if (!p) return nullptr;
    return p->d_counts3 ? p->d_counts3 : p->lease->ws->w_counts3.as<int64_t>();
}
tkz_status tkz_encode_batch_device_end(tkz_pending* p, int64_t* total_tokens) {
    if (!p) return fail(TKZ_E_ARG, "null pending");
    tkz_status st;
    tkz_encoder* e = p->e;
    {
        DeviceScope scope;
        st = check_encoder(e, scope);
        bool dead;
        { std::lock_guard<std::mutex> lock(e->mu); dead = e->destroyed; }
        if (st == TKZ_OK && !dead)
            st = encode_device(e, p->lease->ws, p->d_bytes, p->d_offs, p->n_docs, p->total, p->d_out, p->out_cap, p->d_out_offs, p->stream, true, nullptr, total_tokens, nullptr, kCallEnd, p->d_counts3);
        else {
            (void)hipStreamSynchronize(p->stream);
            if (st == TKZ_OK) st = fail(TKZ_E_ARG, "the encoder was destroyed while this batch was in flight");
        }
    }
    delete p->lease;
    delete p;
    bool last;
    { std::lock_guard<std::mutex> lock(e->mu); last = --e->pending == 0 && e->destroyed; }
    if (last) destroy_now(e);
    return st;
}

tkz_status tkz_encode_utf8(tkz_encoder* e, const uint8_t* text, int64_t len, int32_t* out_ids, int64_t out_cap, int64_t* n_out) {
    if (len < 0 || !n_out) return fail(TKZ_E_ARG, "bad argument");
    const int64_t offs[2] = {0, len};
    int64_t oo[2] = {0, 0}, needed = 0;
    tkz_status st = encode_host(e, text, nullptr, offs, 1, out_ids, out_cap, oo, &needed, true, nullptr);
    *n_out = needed;
    return st;
}

tkz_status tkz_encode_utf16(tkz_encoder* e, const uint16_t* text, int64_t len, int32_t* out_ids, int64_t out_cap, int64_t* n_out) {
    if (len < 0 || (!text && len) || !n_out) return fail(TKZ_E_ARG, "bad argument");
    // Encoding.UTF8.GetBytes semantics: a surrogate pair -> 4 bytes, a lone surrogate -> U+FFFD (EF BF BD).
    // Splitting is unaffected: a lone surrogate (Cs) and U+FFFD (So) are both one "other" unit under every pattern.
    std::vector<uint8_t> u8;
    u8.reserve((size_t)len * 3);
    for (int64_t i = 0; i < len; ++i) {
        uint32_t c = text[i];
        if (c >= 0xD800 && c <= 0xDBFF && i + 1 < len && text[i + 1] >= 0xDC00 && text[i + 1] <= 0xDFFF) {
            c = 0x10000 + ((c - 0xD800) << 10) + (text[i + 1] - 0xDC00); ++i;
        } else if (c >= 0xD800 && c <= 0xDFFF) c = 0xFFFD;
        if (c < 0x80) u8.push_back((uint8_t)c);
        else if (c < 0x800) { u8.push_back(0xC0 | (c >> 6)); u8.push_back(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { u8.push_back(0xE0 | (c >> 12)); u8.push_back(0x80 | ((c >> 6) & 0x3F)); u8.push_back(0x80 | (c & 0x3F)); }
        else { u8.push_back(0xF0 | (c >> 18)); u8.push_back(0x80 | ((c >> 12) & 0x3F)); u8.push_back(0x80 | ((c >> 6) & 0x3F)); u8.push_back(0x80 | (c & 0x3F)); }
    }
    return tkz_encode_utf8(e, u8.data(), (int64_t)u8.size(), out_ids, out_cap, n_out);
}

tkz_status tkz_encode_batch_utf16(tkz_encoder* e, const uint16_t* units, const int64_t* unit_offsets, int64_t n_docs,
                                  int32_t* out_ids, int64_t out_cap, int64_t* out_offsets, int64_t* needed) {
    if (!out_offsets || (out_cap > 0 && !out_ids)) return fail(TKZ_E_ARG, "null output buffer");
    if (n_docs < 0 || !unit_offsets || (n_docs > 0 && !units && unit_offsets[n_docs] > 0)) return fail(TKZ_E_ARG, "null buffer");
    static const uint16_t none = 0;
    return encode_host(e, nullptr, units ? units : &none, unit_offsets, n_docs, out_ids, out_cap, out_offsets, needed, true, nullptr);
}

tkz_status tkz_pretokenize_utf8(tkz_encoder* e, const uint8_t* bytes, const int64_t* doc_offsets, int64_t n_docs, uint64_t* out_bitmap_words) {
    if (!out_bitmap_words) return fail(TKZ_E_ARG, "null output buffer");
    return encode_host(e, bytes, nullptr, doc_offsets, n_docs, nullptr, 0, nullptr, nullptr, true, out_bitmap_words);
}

tkz_status tkz_encode_pieces(tkz_encoder* e, const uint8_t* bytes, const int64_t* piece_offsets, int64_t n_pieces,
                             int32_t* out_ids, int64_t out_cap, int64_t* out_offsets, int64_t* needed) {
    if (!out_offsets || (out_cap > 0 && !out_ids)) return fail(TKZ_E_ARG, "null output buffer");
    return encode_host(e, bytes, nullptr, piece_offsets, n_pieces, out_ids, out_cap, out_offsets, needed, false, nullptr);
}

tkz_status tkz_encode_batch_pieces_utf8(tkz_encoder* e, const uint8_t* bytes, const int64_t* doc_offsets, int64_t n_docs,
                                        int32_t* out_ids, int64_t out_cap, int64_t* doc_piece_offsets, int64_t* piece_byte_offsets,
                                        int64_t* piece_token_offsets, int64_t piece_cap, int64_t* n_pieces, int64_t* needed_ids) {
    if (!doc_piece_offsets || !piece_byte_offsets || !piece_token_offsets || !n_pieces || (out_cap > 0 && !out_ids))
        return fail(TKZ_E_ARG, "null output buffer");
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (n_docs < 0 || !doc_offsets || (n_docs > 0 && !bytes && doc_offsets[n_docs] > 0)) return fail(TKZ_E_ARG, "null buffer");
    if (doc_offsets[0] != 0) return fail(TKZ_E_ARG, "doc_offsets[0] must be 0");
    const int64_t total = doc_offsets[n_docs];
    if (total < 0 || piece_cap < 0) return fail(TKZ_E_ARG, "negative size");
    *n_pieces = 0;
    if (needed_ids) *needed_ids = 0;
    if (total == 0) {                                        // no bytes: no pieces (empty documents have none)
        for (int64_t d = 0; d <= n_docs; ++d) { if (doc_offsets[d] != 0) return fail(TKZ_E_ARG, "document offsets must start at 0, be non-decreasing and end at the byte count"); doc_piece_offsets[d] = 0; }
        piece_byte_offsets[0] = 0; piece_token_offsets[0] = 0;
        return TKZ_OK;
    }
    // ONE launch sequence on the device: Regex.Matches -> piece offsets from the bitmap -> encode with a token mark per piece
    Lease lease(e);
    Workspace* ws = lease.ws;
    int64_t* acc = &ws->bytes_allocated;
    const int64_t pcap = std::min<int64_t>(piece_cap, total);           // pieces <= bytes
    const int64_t cap = std::min<int64_t>(out_cap, total);
    HIP_TRY(ws->s_bytes[0].ensure((size_t)total + 64, acc));
    HIP_TRY(ws->s_offs[0].ensure((size_t)(n_docs + 1) * 8, acc));
    HIP_TRY(ws->s_out[0].ensure((size_t)std::max<int64_t>(cap, 1) * 4, acc));
    HIP_TRY(ws->p_boffs.ensure((size_t)(pcap + 1) * 8, acc));
    HIP_TRY(ws->p_toffs.ensure((size_t)(pcap + 1) * 8, acc));
    HIP_TRY(ws->p_docp.ensure((size_t)(n_docs + 1) * 8, acc));
    HIP_TRY(hipMemcpy(ws->s_bytes[0].p, bytes, (size_t)total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ws->s_offs[0].p, doc_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice));
    PiecesOut po{ws->p_boffs.as<int64_t>(), ws->p_toffs.as<int64_t>(), ws->p_docp.as<int64_t>(), pcap, 0};
    int64_t tokens = 0;
    st = encode_device(e, ws, ws->s_bytes[0].as<uint8_t>(), ws->s_offs[0].as<int64_t>(), n_docs, total, ws->s_out[0].as<int32_t>(), cap, nullptr, nullptr, true,
                       nullptr, &tokens, &po);
    *n_pieces = po.n_pieces;
    if (needed_ids) *needed_ids = tokens;
    if (st != TKZ_OK) return st;
    const int64_t np = po.n_pieces;
    if (tokens) HIP_TRY(hipMemcpy(out_ids, ws->s_out[0].p, (size_t)tokens * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(piece_byte_offsets, ws->p_boffs.p, (size_t)(np + 1) * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(piece_token_offsets, ws->p_toffs.p, (size_t)(np + 1) * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(doc_piece_offsets, ws->p_docp.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost));
    return TKZ_OK;
}

// ---- Decode (TikTokenizer.cs:586-604) ----------------------------------------------------------------

tkz_status tkz_encoder_set_special_tokens(tkz_encoder* e, const int32_t* ids, const uint8_t* literals_utf8, const int64_t* literal_offsets, int32_t n) {
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (n < 0 || (n > 0 && (!ids || !literal_offsets || (!literals_utf8 && literal_offsets[n] > 0)))) return fail(TKZ_E_ARG, "bad special-token arguments");
    std::lock_guard<std::mutex> lock(e->mu);
    std::vector<std::pair<int32_t, std::string>> sp;
    for (int32_t i = 0; i < n; ++i) {
        if (literal_offsets[i + 1] < literal_offsets[i] || literal_offsets[i] < 0) return fail(TKZ_E_ARG, "literal offsets must be non-decreasing");
        sp.emplace_back(ids[i], std::string(reinterpret_cast<const char*>(literals_utf8) + literal_offsets[i], (size_t)(literal_offsets[i + 1] - literal_offsets[i])));
    }
    e->dec_special.swap(sp);
    return build_decode_table(e);
}

namespace {
// lengths -> scan -> bytes + document offsets, all on `stream`; d_out may be null when only the size is wanted
tkz_status decode_device(tkz_encoder* e, Workspace* ws, const int32_t* d_ids, const int64_t* d_id_offs, int64_t n_docs, int64_t total_ids, uint8_t* d_out, int64_t out_cap,
                         int64_t* d_out_offs, hipStream_t stream, int64_t* total_bytes) {
    using namespace tkz;
    if (total_bytes) *total_bytes = 0;
    if (n_docs < 0 || total_ids < 0 || out_cap < 0) return fail(TKZ_E_ARG, "negative size");
    if (n_docs == 0 && total_ids != 0) return fail(TKZ_E_ARG, "ids without documents");
    int64_t* acc = &ws->bytes_allocated;
    const int64_t ntiles = std::max<int64_t>(1, dec_tiles(total_ids)), nblk = (ntiles + kScanBlock - 1) / kScanBlock;
    HIP_TRY(ws->d_grp.ensure((size_t)ntiles * 64 * 4, acc));
    HIP_TRY(ws->d_tsum.ensure((size_t)ntiles * 4, acc));
    HIP_TRY(ws->d_tbase.ensure((size_t)ntiles * 8, acc));
    HIP_TRY(ws->d_bsum.ensure((size_t)(nblk + 1) * 8, acc));
    HIP_TRY(ws->d_counters.ensure(64, acc));
    if (!ws->h_counters) HIP_TRY(hipHostMalloc((void**)&ws->h_counters, sizeof(CounterBlock), 0));
    Launch L{stream, nullptr, ws};
    int32_t* counters = ws->d_counters.as<int32_t>();
    int64_t* grand = reinterpret_cast<int64_t*>(ws->d_counters.as<char>() + 8);
    HIP_TRY(hipMemsetAsync(counters, 0, 64, stream));
    launch_dec_len(L, e->D, d_ids, total_ids, ntiles, ws->d_grp.as<int32_t>(), ws->d_tsum.as<int32_t>());
    launch_scan(L, ws->d_tsum.as<int32_t>(), ntiles, ws->d_bsum.as<int64_t>(), ws->d_tbase.as<int64_t>(), grand, -1);
    launch_dec_write(L, e->D, d_ids, total_ids, ntiles, ws->d_tbase.as<int64_t>(), d_out, d_out ? out_cap : 0, d_id_offs, n_docs, ws->d_grp.as<int32_t>(), grand,
                     d_out_offs, counters);
    struct { int32_t err; int32_t pad; int64_t grand; } h{};
    HIP_TRY(hipMemcpyAsync(ws->h_counters, counters, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
    memcpy(&h, ws->h_counters, 16);
    if (h.err & kErrOffsets) return fail(TKZ_E_ARG, "id offsets must start at 0, be non-decreasing and end at the id count");
    if (total_bytes) *total_bytes = h.grand;
    if (h.grand > out_cap) return fail(TKZ_E_CAPACITY, "output capacity too small");
    return TKZ_OK;
}
}  // namespace

tkz_status tkz_decode_batch_device(tkz_encoder* e, const int32_t* d_ids, const int64_t* d_id_offsets, int64_t n_docs, int64_t total_ids,
                                   uint8_t* d_out_bytes, int64_t out_cap, int64_t* d_out_offsets, void* hip_stream, int64_t* total_bytes) {
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (!d_id_offsets || !d_out_offsets || (total_ids > 0 && !d_ids) || (out_cap > 0 && !d_out_bytes)) return fail(TKZ_E_ARG, "null device buffer");
    Lease lease(e);
    Workspace* ws = lease.ws;
    return decode_device(e, ws, d_ids, d_id_offsets, n_docs, total_ids, d_out_bytes, out_cap, d_out_offsets, static_cast<hipStream_t>(hip_stream), total_bytes);
}

tkz_status tkz_decode_batch(tkz_encoder* e, const int32_t* ids, const int64_t* id_offsets, int64_t n_docs, uint8_t* out_bytes, int64_t out_cap,
                            int64_t* out_offsets, int64_t* needed) {
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    if (n_docs < 0 || !id_offsets || !out_offsets || (out_cap > 0 && !out_bytes)) return fail(TKZ_E_ARG, "null buffer");
    if (id_offsets[0] != 0) return fail(TKZ_E_ARG, "id_offsets[0] must be 0");
    const int64_t total = id_offsets[n_docs];
    if (total < 0 || (total > 0 && !ids)) return fail(TKZ_E_ARG, "bad id count");
    if (needed) *needed = 0;
    Lease lease(e);
    Workspace* ws = lease.ws;
    int64_t* acc = &ws->bytes_allocated;
    HIP_TRY(ws->d_ids.ensure((size_t)std::max<int64_t>(total, 1) * 4, acc));
    HIP_TRY(ws->d_idoffs.ensure((size_t)(n_docs + 1) * 8, acc));
    HIP_TRY(ws->d_outoffs.ensure((size_t)(n_docs + 1) * 8, acc));
    // (staging no larger than the result can be: an id yields at most the longest registered byte string, however large a hint out_cap is)
    int64_t longest = 1;
    { std::lock_guard<std::mutex> lock(e->mu); for (const auto& kv : e->dec_special) longest = std::max<int64_t>(longest, (int64_t)kv.second.size()); }
    longest = std::max<int64_t>(longest, e->max_key_len);
    out_cap = std::min<int64_t>(out_cap, total * longest);
    HIP_TRY(ws->d_out.ensure((size_t)std::max<int64_t>(out_cap, 1), acc));
    if (total) HIP_TRY(hipMemcpy(ws->d_ids.p, ids, (size_t)total * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ws->d_idoffs.p, id_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice));
    int64_t nbytes = 0;
    st = decode_device(e, ws, ws->d_ids.as<int32_t>(), ws->d_idoffs.as<int64_t>(), n_docs, total, ws->d_out.as<uint8_t>(), out_cap, ws->d_outoffs.as<int64_t>(), nullptr, &nbytes);
    if (needed) *needed = nbytes;
    if (st != TKZ_OK) return st;
    if (nbytes) HIP_TRY(hipMemcpy(out_bytes, ws->d_out.p, (size_t)nbytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_offsets, ws->d_outoffs.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost));
    return TKZ_OK;
}

tkz_status tkz_encoder_set_option(tkz_encoder* e, int32_t option, int64_t value) {
    if (!e) return fail(TKZ_E_ARG, "null encoder");
    if (option == TKZ_OPT_PRETOK_SEQUENTIAL) { e->pretok_seq = value != 0; return TKZ_OK; }
    if (option == TKZ_OPT_PIECE_STATS) {
        // statistics of the batch path (the single-launch path does not count): the device block is made on first use
        DeviceScope scope;
        tkz_status st = check_encoder(e, scope);
        if (st != TKZ_OK) return st;
        std::lock_guard<std::mutex> lock(e->mu);
        if (value && !e->t_stats.p) {
            HIP_TRY(e->t_stats.ensure(128, &e->bytes_allocated));        // (the block + the copy an attempt is rolled back to)
            HIP_TRY(hipMemset(e->t_stats.p, 0, 128));
        }
        e->piece_stats = value != 0;
        return TKZ_OK;
    }
    if (option == TKZ_OPT_LATENCY_BYTES) {
        if (value < 0) return fail(TKZ_E_ARG, "negative value");
        std::lock_guard<std::mutex> lock(e->mu);
        e->latency_bytes = value;
        return TKZ_OK;
    }
    if (option == TKZ_OPT_CASE_EQUIVALENCE) {
        if (value != 0 && value != 1) return fail(TKZ_E_ARG, "TKZ_OPT_CASE_EQUIVALENCE takes 0 or 1");
        std::lock_guard<std::mutex> lock(e->mu);
        e->case_equiv = value != 0;
        return TKZ_OK;
    }
    if (option == TKZ_OPT_PROMOTE_MIN_BYTES || option == TKZ_OPT_PROMOTE_CAP) {
        if (value < 0) return fail(TKZ_E_ARG, "negative value");
        std::lock_guard<std::mutex> lock(e->mu);
        if (option == TKZ_OPT_PROMOTE_MIN_BYTES) e->promo_min_bytes = value;
        else e->promo_cap = (size_t)std::min<int64_t>(value, (int64_t)kPromoMaxEntries);
        return TKZ_OK;
    }
    if (option == TKZ_OPT_PROMOTE) {
        // 0 / 1: automatic promotion off / on.  2: promote NOW whatever the memo holds (every valid entry counts alike); 3: drop every promotion (the
        // key tables as the vocabulary alone gives them).  2 and 3 replace table images that a call in flight may be probing: refused unless the encoder is idle.
        DeviceScope scope;
        tkz_status st = check_encoder(e, scope);
        if (st != TKZ_OK) return st;
        if (value == 0 || value == 1) {
            { std::lock_guard<std::mutex> lock(e->mu); e->promo_mode = (int)value; }
            if (value == 0) join_promotion(e);     // off: the tables do not change once this has returned
            return TKZ_OK;
        }
        if (value != 2 && value != 3) return fail(TKZ_E_ARG, "TKZ_OPT_PROMOTE takes 0, 1, 2 or 3");
        join_promotion(e);
        {
            std::lock_guard<std::mutex> lock(e->mu);
            for (Workspace* w : e->pool) if (w->busy) return fail(TKZ_E_ARG, "promotions can only be made or dropped by hand while no call of this encoder is in flight");
        }
        if (hipDeviceSynchronize() != hipSuccess) return fail(TKZ_E_DEVICE, "hipDeviceSynchronize");
        // (the images these two replace are RETIRED, not freed here: the check above is not held until the new ones are in place, and a call that started in
        //  between has taken its copy of the table descriptor -- the round-5 advisor; they go when no call is in flight, ~Lease)
        if (value == 2) return promote_from_memo(e, false, true, nullptr);
        {
            std::lock_guard<std::mutex> lock(e->mu);
            e->promo_rounds = 0; e->learn_bytes = 0; e->bytes_at_promo = e->bytes_at_install = e->bytes_seen; e->ew_valid = e->base_valid = e->window_valid = e->last_window_valid = false; e->win_miss = e->win_pieces = 0;
        }
        {
            bool none;
            { std::lock_guard<std::mutex> lock(e->mu); none = e->promo_items.empty(); }
            if (none) return TKZ_OK;
        }
        return drop_promotions(e, true);
    }
    if (option == TKZ_OPT_PIECE_MEMO) {
        // 0: off, 1: on, 2: on and emptied.  Options are set while the encoder is idle: a call in flight on another thread reads
        // T.memo_n when it launches, and emptying the table under running kernels could pair one piece's key with another's tokens
        // -- so value 2 is refused while any workspace is leased, and the device is drained before the table is cleared.
        DeviceScope scope;
        if (value == 2) join_promotion(e);     // (a promotion being built reads the memo back: not under its feet)
        std::lock_guard<std::mutex> lock(e->mu);
        if (value == 2) {
            tkz_status st = check_encoder(e, scope);
            if (st != TKZ_OK) return st;
            for (Workspace* w : e->pool) if (w->busy) return fail(TKZ_E_ARG, "the piece memo can only be emptied while no call of this encoder is in flight");
            if (hipDeviceSynchronize() != hipSuccess) return fail(TKZ_E_DEVICE, "hipDeviceSynchronize");
            if (hipMemset(e->t_memo.p, 0, size_t(e->memo_slots) * sizeof(TkzMemoSlot)) != hipSuccess) return fail(TKZ_E_DEVICE, "hipMemset");
        }
        e->T.memo_n = value ? e->memo_slots : 0u;
        return TKZ_OK;
    }
    if (option == TKZ_OPT_ADAPT) {
        if (value != 0 && value != 1) return fail(TKZ_E_ARG, "TKZ_OPT_ADAPT takes 0 or 1");
        std::lock_guard<std::mutex> lock(e->mu);
        e->adapt = (int)value;
        return TKZ_OK;
    }
    return fail(TKZ_E_ARG, "unknown option");
}

tkz_status tkz_encoder_adapt_stats(tkz_encoder* e, int64_t* out8) {
    if (!e || !out8) return fail(TKZ_E_ARG, "null argument");
    join_promotion(e);                     // (the figures after whatever is being built in the background)
    std::lock_guard<std::mutex> lock(e->mu);
    out8[0] = e->n_promotions; out8[1] = e->n_relearns; out8[2] = (int64_t)e->promo_items.size(); out8[3] = (int64_t)e->retired.size();
    out8[4] = e->base_valid ? (int64_t)(e->base_miss * 1e6) : -1; out8[5] = e->ew_valid ? (int64_t)(e->ew_miss * 1e6) : -1;
    out8[6] = e->bytes_seen - e->bytes_at_install; out8[7] = e->learn_bytes;
    return TKZ_OK;
}

tkz_status tkz_encoder_set_profiling(tkz_encoder* e, int32_t enabled) {
    if (!e) return fail(TKZ_E_ARG, "null encoder");
    e->profiling = enabled != 0;
    return TKZ_OK;
}
tkz_status tkz_encoder_kernel_ms(tkz_encoder* e, double* ms, int64_t* launches, int32_t reset) {
    if (!e) return fail(TKZ_E_ARG, "null encoder");
    std::lock_guard<std::mutex> lock(e->mu);
    for (int k = 0; k < tkz::K_COUNT; ++k) {
        double m = 0; int64_t n = 0;
        for (Workspace* w : e->pool) { m += w->ms[k]; n += w->launches[k]; if (reset) { w->ms[k] = 0; w->launches[k] = 0; } }
        if (ms) ms[k] = m;
        if (launches) launches[k] = n;
    }
    return TKZ_OK;
}
void tkz_encoder_pretok_leftovers(const tkz_encoder* e, int64_t* after_ascii_scanner, int64_t* after_multibyte_scanner) {
    if (after_ascii_scanner) *after_ascii_scanner = e ? e->last_xcount.load() : 0;
    if (after_multibyte_scanner) *after_multibyte_scanner = e ? e->last_xcount2.load() : 0;
}
tkz_status tkz_encoder_piece_stats(tkz_encoder* e, int64_t* out8, int32_t reset) {
    if (!e || !out8) return fail(TKZ_E_ARG, "null argument");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    join_promotion(e);                     // (the count of promoted pieces below is the one after a promotion in the background, if one is running)
    if (!e->t_stats.p) { std::lock_guard<std::mutex> lock(e->mu); out8[7] = (int64_t)e->promo_items.size(); return TKZ_OK; }
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    unsigned long long h[8] = {};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h, e->t_stats.p, sizeof h, hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> lock(e->mu);
    out8[0] = e->stat_batches; out8[1] = (int64_t)h[4]; out8[2] = (int64_t)h[2]; out8[3] = (int64_t)h[3]; out8[4] = e->stat_giants;
    out8[5] = (int64_t)h[0]; out8[6] = (int64_t)h[1]; out8[7] = (int64_t)e->promo_items.size();
    if (reset) { HIP_TRY(hipMemset(e->t_stats.p, 0, 64)); e->stat_batches = e->stat_giants = 0; }
    return TKZ_OK;
}
int64_t tkz_encoder_memo_slots(const tkz_encoder* e) { return e ? (int64_t)e->memo_slots : 0; }
void tkz_encoder_small_path_calls(const tkz_encoder* e, int64_t* calls, int64_t* handed_back) {
    int64_t c = 0, f = 0;
    if (e) { tkz_encoder* m = const_cast<tkz_encoder*>(e); std::lock_guard<std::mutex> lock(m->mu); for (Workspace* w : e->pool) { c += w->small_calls; f += w->small_fallbacks; } }
    if (calls) *calls = c;
    if (handed_back) *handed_back = f;
}
int32_t tkz_encoder_small_path_phases(const tkz_encoder* e, int64_t* clocks16) {       // shader-clock stamps of the last k_small of the first workspace
    if (!e || !clocks16) return 0;
    tkz_encoder* m = const_cast<tkz_encoder*>(e);
    std::lock_guard<std::mutex> lock(m->mu);
    for (Workspace* w : e->pool) if (w->h_small) { memcpy(clocks16, w->small_clocks, 16 * 8); return 16; }      // (a snapshot taken after the call's synchronisation, never the block a kernel may be writing)
    return 0;
}
int32_t tkz_encoder_memo_ways(const tkz_encoder* e) { return e ? (int32_t)kMemoWays : 0; }
int64_t tkz_encoder_memo_bucket(const tkz_encoder* e, const uint8_t* piece, int32_t len) {
    if (!e || !piece || len < 1 || len > 16 || !e->memo_slots) return -1;
    uint32_t kw[4] = {0, 0, 0, 0};
    for (int32_t i = 0; i < len; ++i) { if (!piece[i]) return -1; kw[i >> 2] |= (uint32_t)piece[i] << (8 * (i & 3)); }      // (pieces with a zero byte never use the memo)
    return (int64_t)tkz_mulhi(tkz_hash_memo(kw, (uint32_t)len), e->memo_slots / kMemoWays);
}
// CreateTokenizer is where a drop-in pays construction costs (TokenizerBuilder.cs:210-213), not the first Encode: the workspace of a batch of up to
// max_bytes bytes in max_docs documents -- ~7.8 bytes per input byte -- is allocated here instead of inside the first batch call, where its hipMallocs
// took anything from 5 ms to seconds.  Later batches of up to that size allocate nothing (lists that a miss-heavy text needs longer still grow once).
tkz_status tkz_encoder_reserve(tkz_encoder* e, int64_t max_bytes, int64_t max_docs) {
    using namespace tkz;
    if (!e) return fail(TKZ_E_ARG, "null encoder");
    if (max_bytes < 0 || max_docs < 0) return fail(TKZ_E_ARG, "negative size");
    DeviceScope scope;
    tkz_status st = check_encoder(e, scope);
    if (st != TKZ_OK) return st;
    Lease lease(e);
    Workspace* ws = lease.ws;
    int64_t* acc = &ws->bytes_allocated;
    st = prepare_workspace(ws, max_bytes, max_docs, false, false);
    if (st != TKZ_OK) return st;
    const int64_t nwords = max_bytes / 64 + 1;
    HIP_TRY(ws->w_xq.ensure((size_t)(nwords / kRowsPerWave + 4) * 16, acc));                      // (the o200k scanners' queues)
    if (!ws->h_counters) HIP_TRY(hipHostMalloc((void**)&ws->h_counters, sizeof(CounterBlock), 0));
    HIP_TRY(ensure_streams(ws));
    {   // what the host-buffer entry points stage a chunk in (two input sets, three output sets: encode_host cuts a large batch into chunks of at most 32 MB; the
        // second workspace its pipeline leases for every other chunk is a chunk's size and sizes itself on its first use)
        const int64_t chunk = std::min<int64_t>(max_bytes, int64_t(32) << 20), cdocs = std::min<int64_t>(max_docs, std::max<int64_t>(1, chunk / 16));
        for (int q = 0; q < 3; ++q) {
            if (q < 2) {
                HIP_TRY(ws->s_bytes[q].ensure((size_t)chunk + 64, acc));
                HIP_TRY(ws->s_offs[q].ensure((size_t)(cdocs + 1) * 8, acc));
            }
            HIP_TRY(ws->s_out[q].ensure((size_t)std::max<int64_t>(chunk, 1) * 4, acc));
            HIP_TRY(ws->s_outoffs[q].ensure((size_t)(cdocs + 1) * 8, acc));
        }
    }
    {   // the counters and the log of a learning window (TKZ_OPT_PROMOTE)
        std::lock_guard<std::mutex> lock(e->mu);
        if (e->memo_slots && e->promo_mode == 1) {
            if (e->t_memo_hits.ensure((size_t)e->memo_slots * 4, &e->bytes_allocated) != hipSuccess ||
                e->t_long_log.ensure((size_t)kLongLogCap * kLongLogDwords * 4 + 64, &e->bytes_allocated) != hipSuccess)
                return fail(TKZ_E_OUT_OF_MEMORY, "learning buffers could not be allocated");
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return TKZ_OK;
}
int64_t tkz_encoder_workspace_bytes(const tkz_encoder* e) {
    if (!e) return 0;
    tkz_encoder* m = const_cast<tkz_encoder*>(e);
    std::lock_guard<std::mutex> lock(m->mu);
    int64_t n = e->bytes_allocated;
    for (Workspace* w : e->pool) n += w->bytes_allocated;
    return n;
}
int64_t tkz_encoder_side_by_side_batches(const tkz_encoder* e) {
    if (!e) return 0;
    tkz_encoder* m = const_cast<tkz_encoder*>(e);
    std::lock_guard<std::mutex> lock(m->mu);
    int64_t n = 0;
    for (Workspace* w : e->pool) n += w->forked_batches;
    return n;
}
int64_t tkz_encoder_engine_downloads(const tkz_encoder* e) {
    if (!e) return 0;
    tkz_encoder* m = const_cast<tkz_encoder*>(e);
    std::lock_guard<std::mutex> lock(m->mu);
    int64_t n = 0;
    for (Workspace* w : e->pool) n += w->engine_downloads.load(std::memory_order_relaxed);
    return n;
}
const char* tkz_kernel_name(int32_t k) {
    static const char* const names[] = {"k_docmark", "k_pretok", "k_probe", "k_scan", "k_place", "k_docoffs", "k_merge_long_group", "k_merge_short"};
    return (k >= 0 && k < tkz::K_COUNT) ? names[k] : "?";
}

tkz_status tkz_corpus_generate_device(int32_t device, int32_t kind, uint64_t seed, int64_t first_doc, int64_t n_docs,
                                      int32_t min_len, int32_t max_len, int64_t* d_doc_offsets, uint8_t* d_bytes,
                                      int64_t cap_bytes, void* hip_stream, int64_t* total_bytes) {
    if (kind < 1 || kind > 5 || kind == 4 || n_docs < 0 || min_len < 0 || max_len < min_len || !d_doc_offsets || !total_bytes)
        return fail(TKZ_E_ARG, "bad corpus arguments");
    DeviceScope scope;
    hipError_t r = scope.enter(device);
    if (r != hipSuccess) return fail(TKZ_E_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(r));
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int64_t* d_total = nullptr;
    HIP_TRY(hipMalloc((void**)&d_total, 8));
    tkz::launch_corpus(s, kind, seed, first_doc, n_docs, min_len, max_len, d_doc_offsets, d_bytes, d_bytes ? cap_bytes : 0, d_total);
    hipError_t c = hipMemcpyAsync(total_bytes, d_total, 8, hipMemcpyDeviceToHost, s);
    if (c == hipSuccess) c = hipStreamSynchronize(s);
    if (c == hipSuccess) c = hipGetLastError();
    (void)hipFree(d_total);
    if (c != hipSuccess) return fail(TKZ_E_DEVICE, std::string("corpus generation: ") + hipGetErrorString(c));
    if (d_bytes && *total_bytes > cap_bytes) return fail(TKZ_E_CAPACITY, "corpus buffer too small");
    return TKZ_OK;
}

// ---- shard arithmetic of the multi-GPU partitioning (the communicator itself is tkz_comm.cpp) ----
void tkz_shard_range(int64_t n_docs_total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi) {
    if (world < 1) world = 1;
    // (128-bit product: n_docs_total * world may exceed 2^63 only for absurd inputs, but costs nothing to get right)
    if (lo) *lo = (int64_t)(((__int128)n_docs_total * rank) / world);
    if (hi) *hi = (int64_t)(((__int128)n_docs_total * (rank + 1)) / world);
}

tkz_status tkz_shard_bases(const int64_t* table, int32_t world, int32_t rank, int64_t* bases3, int64_t* totals3) {
    if (!table || world < 1 || rank < 0 || rank >= world) return fail(TKZ_E_ARG, "bad table / rank");
    int64_t b[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    for (int r = 0; r < world; ++r)
        for (int k = 0; k < 3; ++k) {
            const int64_t v = table[3 * r + k];
            if (v < 0) return fail(TKZ_E_ARG, "negative count in the gathered table");
            if (r < rank) b[k] += v;
            t[k] += v;
        }
    if (bases3) memcpy(bases3, b, sizeof b);
    if (totals3) memcpy(totals3, t, sizeof t);
    return TKZ_OK;
}

int64_t tkz_corpus_generate_doc_host(int32_t kind, uint64_t seed, int64_t doc_index, int32_t min_len, int32_t max_len,
                                     uint8_t* buf, int64_t cap) {
    if (kind < 1 || kind > 5 || kind == 4 || min_len < 0 || max_len < min_len) return -1;
    return tkz_corpus_doc(kind, seed, doc_index, min_len, max_len, buf, cap);
}

}  // extern "C"
