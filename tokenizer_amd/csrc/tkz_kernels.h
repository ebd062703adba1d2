// tkz_kernels.h -- host-callable launchers of the HIP kernels (tkz_kernels.hip) and the parameter
// blocks they take.
#pragma once
#include <stdint.h>

#include "tkz_simt.h"
#include "tkz_tables.h"

namespace tkz {

constexpr int kThreads = 256;       // workgroup size of every kernel (4 wavefronts)
constexpr int kSub = 1024;          // bytes of corpus per wavefront of k_probe / k_place (a "sub-tile")
constexpr int kHalo = 64;           // bytes staged past the sub-tile (a short piece may straddle the edge)
constexpr int kShortMax = 16;       // pieces up to this many bytes are merged one per lane, 64 to a wavefront, by k_merge_short
#ifndef TKZ_MERGE_GROUP
#define TKZ_MERGE_GROUP 16
#endif
constexpr int kMergeGroup = TKZ_MERGE_GROUP;   // sub-tiles per wavefront of k_merge_short ("group")
constexpr int kDenseCap = 256 * kMergeGroup;   // tokens of merged short pieces a group keeps packed (what does not fit waits in tmp, like the long pieces' tokens)
#ifndef TKZ_ARENA_DWORDS
#define TKZ_ARENA_DWORDS 2560
#endif
constexpr int kArenaDwords = TKZ_ARENA_DWORDS;  // LDS arena of k_merge_long: a long miss of a batch gets its bytes + 1 dword + 1 bit per byte out of it (2 dwords per byte for
                                    // vocabularies with ranks of 2^21 and more, which keep an ids[] array: tkz_bpe.h)
constexpr int kArenaPiece = 1024;   // ... so pieces up to this many bytes are merged one per lane there (256 + tkz_bpe_var_dwords(1024) = 2336 <= kArenaDwords)
#ifndef TKZ_LANE_PIECE
#define TKZ_LANE_PIECE 128
#endif
constexpr int kLanePiece = TKZ_LANE_PIECE;
constexpr int kLongLogDwords = 12, kLongLogMaxLen = 28;      // (28 = TKZ_MID_KEY_MAX: what the MID key table can hold)
constexpr int kSmallLanePiece = 256;  // ... the single-launch kernel (no k_merge_coop there) merges pieces of up to this many bytes a lane each and hands a batch with a longer missed piece back    // ... but a missed piece longer than this is merged by a whole wavefront (k_merge_coop): one lane takes ~n^2 steps, and the kernel waits for it
constexpr int kMaxPiece = 1 << 30;  // longer single pieces are refused (kErrTooLong)
constexpr int kRowsPerWave = 62;    // k_pretok_rows: output rows per wavefront (64 staged rows, one per lane; the outer two are context)
constexpr int kScanBlock = 1024;    // tiles per workgroup in the tile-count scan
constexpr int kRecordLine = 16;     // piece records per 64-byte line: every sub-tile's records start on a line of their own
constexpr int kMissCapMin = 64, kMissCapMax = 1024;   // entries of a sub-tile's miss list: what a workspace starts with, and the most a sub-tile can need (a miss per byte)

enum { K_DOCMARK = 0, K_PRETOK = 1, K_ENCODE = 2, K_SCAN = 3, K_GATHER = 4, K_DOCOFFS = 5, K_HEAVY = 6, K_MERGE_SHORT = 7, K_COUNT = 8 };

#ifdef TKZ_DEVPROF
#define TKZ_DEV_FLAG(P, bit) (((P).ablate & (bit)) != 0)
#else
#define TKZ_DEV_FLAG(P, bit) false
#endif

struct EncodeParams {
    const uint8_t* bytes; int64_t total;
    const uint64_t* startbits; const uint64_t* docbits; int64_t nwords;   // 1 bit / byte, nwords = total/64 + 1
    const int64_t* offs; int64_t n_docs;                                  // n_docs + 1 document offsets
    int32_t* tmp;                 // [total + pad] tokens of a LONG missed piece wait at the piece's own byte position (tokens <= bytes)
    int32_t* dense;               // [groups x kDenseCap] tokens of the short missed pieces of a group of sub-tiles, packed in piece order
    int32_t* tile_count;          // tokens produced by each sub-tile
    int32_t* prank; int64_t prank_cap;   // one record per piece, in piece order (tkz_kernels.hip, "the encode stage")
    const int32_t* pcount;        // pieces that start in each sub-tile
    const int64_t* pbase;         // ... and the exclusive scan of the counts rounded up to 16: where the sub-tile's records (whole lines) start
    uint32_t* mlist; int32_t mcap;       // per sub-tile mcap entries: the short misses from the front, the long ones from the back; answered in place
    uint4* mquad;                 // 16 bytes beside every list entry: a short miss's bytes on the way in, its <= 4 tokens on the way out
    uint32_t* mcount;             // per sub-tile: short misses | long misses << 16
    const int64_t* docord_base;   // per sub-tile: number of distinct document-start positions before it
    int32_t* doc_tok;             // per document-start position (by ordinal): token index inside its sub-tile
    int32_t* counters;            // [0] error bits, [1] the longest miss list seen (kErrMissCap), [2] the longest list above kMissCapMin that fitted (grown lists only),
                                  // [3] sub-tiles with more than 64 list entries ([2], [3]: k_list_stats, grown lists only)
    int32_t latency;              // a batch run for latency (TKZ_OPT_LATENCY_BYTES): the launchers pick the forms with the smaller units of work
    int32_t lane_piece;           // long misses of up to this many bytes are merged a lane each (k_merge_long), longer ones a wavefront each (k_merge_coop): kLanePiece
    uint8_t* heavy_flag; int64_t nsub;                                    // one byte per sub-tile, set by k_probe: bit 0 = long misses in its list, bit 1 = a giant piece, bit 2 = a long miss of more than kLanePiece bytes (k_merge_coop)
                                                                          // (a flag, not a queue: a queue's one counter serialises a million atomics on mixed text)
    int32_t* pool; unsigned long long* pool_head; int64_t pool_cap;       // scratch for pieces > kArenaPiece (int32 units)
    // pieces > kArenaPiece bytes ("giant"): found by k_giant_find, merged by k_giant_merge (one 1024-thread workgroup each); their
    // tokens wait in tmp at the piece's own byte position, their count in giant_cnt[sub-tile of the piece start]
    int64_t* giant_q; unsigned long long* giant_count; int64_t giant_cap; int32_t* giant_cnt;
    unsigned long long* giant_ticket;       // k_giant_merge: next entry of the (longest first) order, giant_q[2 * giant_cap + t], to be taken
    // long misses of more than kLanePiece bytes (k_merge_coop: a wavefront each): queued by k_list_stats as sub-tile << 10 | index in the sub-tile's long list
    uint64_t* coop_q; unsigned long long* coop_count; unsigned long long* coop_ticket; int64_t coop_cap;
    // the long misses of up to lane_piece bytes, binned by length class across the batch (large batches: k_long_count -> scan -> k_long_scatter -> k_merge_long_q).
    // lq null: the chunk form (k_merge_long).  lq_cnt / lq_base: [16 classes x chunks of 64 sub-tiles]; lq: sub-tile << 30 | list index << 20 | (len - 1) << 10 | byte in the sub-tile
    int32_t* lq_cnt; int64_t* lq_base; int64_t* lq_total; int64_t* lq_bsum; uint64_t* lq; int64_t lq_cap;
    unsigned long long* miss_sums;   // null, or [2]: k_list_stats adds the batch's short and long misses (the sums of mcount) -- TKZ_OPT_ADAPT follows their share of the pieces
    // development builds only (make DEVPROF=1; env TKZ_DEV_ABLATE bit 4): per-phase clock counters of k_probe.  Compiled out of libtkz.so otherwise.
    unsigned long long* devprof;
    int32_t ablate;
    // TKZ_OPT_PIECE_STATS: null, or the encoder's statistics block -- [0] memo lookups, [1] memo hits, [2] short misses, [3] long misses, [4] pieces
    // (what tkz_encoder_piece_stats reports; the timed runs leave it null)
    unsigned long long* stats;
    // a LEARNING batch (tkz_api.cpp): k_merge_long logs the pieces of 17..28 bytes it merged into at most 4 tokens -- {7 dwords of bytes, len | count << 8,
    // 4 tokens} = kLongLogDwords dwords a record, up to long_log_cap records (the counter runs on) -- so that the host can promote the ones that repeat
    // (the indentation runs of source code: "\n" + 19 spaces is a missed piece of 20 bytes that real text holds by the hundred thousand)
    uint32_t* long_log; unsigned long long* long_log_count; int32_t long_log_cap; int32_t long_log_sparse;
    int32_t* pextra;              // null, or (promoted pieces in the tables) per sub-tile: tokens beyond one per piece that its promoted pieces stand for (k_probe -> k_merge_short's counts)
    const uint4* promo;           // token quads of the promoted pieces (TkzTables::promo of the tables this batch was probed with), or null: k_place
    int32_t tc_atomic;            // k_probe zeroes tile_count and k_merge_short ADDS its counts (atomics) instead of storing them: the long-piece kernels, which add theirs, may run beside it (Launch::side)
    int32_t place128;             // launch k_place<128> (two kept list entries per lane) instead of k_place<64>: the previous batch of the workspace was miss-heavy
};

// k_small: one launch for a small batch (tkz_kernels.hip).  Input and output live in page-locked host memory the device reads and writes directly.
// (k_small's static LDS: 83.5 KB of gfx950's 160 KB per workgroup -- more than the 64 KB of earlier CDNA parts: the host takes the single-launch path only
//  on a device whose sharedMemPerBlock covers it, the batch path otherwise)
constexpr int kSmallLdsBytesNeeded = 86 * 1024;
constexpr int kSmallMaxBytes = 131072, kSmallMaxBytesO200k = 65536, kSmallMaxDocs = 8192, kSmallMaxDoc = 1024;
struct SmallArgs {
    const uint8_t* h_bytes; const int64_t* h_offs;          // the batch, in page-locked host memory (h_bytes kSmallMaxBytes + 64 long)
    int32_t* out; int64_t out_cap; int64_t* out_offs;       // ids and document offsets, page-locked host memory
    int64_t* h_result;                                      // [0] status (0 done, 1 take the batch path), [1] error bits, [2] token count
    uint64_t* docbits; uint64_t* startbits;                 // the workspace arrays EncodeParams holds as const, writable
    int32_t* pcount; int64_t* pbase; int64_t* docord_base; int64_t* tile_base;
    int32_t counter_words;                                  // 32-bit words of the counter block to zero
    int64_t* counts3[2];                                    // {n_docs, n_bytes, n_tokens} of the batch, on the device: the encoder's block and the workspace's (either may be null)
};

typedef void (*KernelHook)(void* ctx, int kernel_id, int phase /*0 before, 1 after*/, hipStream_t s);
// side / side2 / ev_fork / ev_join / ev_join2: null, or two more streams and three events of the workspace -- launch_encode runs k_merge_long_q and k_merge_coop there, beside
// k_merge_short on `stream`, with grids of at most side_long_grid / side_coop_grid wavefronts (a large batch only: EncodeParams::tc_atomic says that the token counts of the
// sub-tiles are summed with atomics from zero)
struct Launch { hipStream_t stream; KernelHook hook; void* hook_ctx; hipStream_t side = nullptr, side2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
                int side_long_grid = 2048, side_coop_grid = 1024; };

void launch_docmark(const Launch& L, const int64_t* d_offs, int64_t n_items, int64_t total, uint64_t* bits, int32_t* counters);
// position-parallel Regex.Matches; xq / xcount: queue of row blocks left to the sequential matcher (o200k only)
void launch_pretok_rows(const Launch& L, int pattern, const uint8_t* d_bytes, const int64_t* d_offs, int64_t n_docs, int64_t total,
                        const uint64_t* docbits, uint64_t* startbits, int64_t nrows, const uint8_t* bmp, int32_t* counters,
                        int64_t* xq, unsigned long long* xcount);
void launch_pretok_seq(const Launch& L, int pattern, const uint8_t* d_bytes, const int64_t* d_offs, int64_t n_docs, int64_t total,
                       uint64_t* startbits, const uint8_t* bmp, int32_t* counters);
void launch_ingest(const Launch& L, const uint8_t* h_bytes, int64_t total, uint8_t* d_bytes, const int64_t* h_offs, int64_t n_offs, int64_t* d_offs, void* zero, int64_t zero_bytes);
void launch_probe_sample(const Launch& L, const TkzTables& T, const EncodeParams& P, int64_t nsample);
void launch_encode(const Launch& L, const TkzTables& T, const EncodeParams& P, int64_t nsub);
void launch_small(const Launch& L, const TkzTables& T, const EncodeParams& P, const SmallArgs& A);
void launch_doccount(const Launch& L, const uint64_t* docbits, int64_t nwords, int64_t total, int64_t nsub, int32_t* cnt);
// exclusive scan int32 -> int64 (+ grand total); kid = profiling id of the bracket, or -1
// round_to (a power of two): every count is rounded up to a multiple of it before it is summed
void launch_scan(const Launch& L, const int32_t* tile_count, int64_t ntiles, int64_t* bsum, int64_t* tile_base, int64_t* grand, int kid, int round_to = 1);
void launch_place(const Launch& L, const EncodeParams& P, const int64_t* tile_base, int64_t nsub, int32_t* out, int64_t out_cap);
// (c3a / c3b / c3c: null, or blocks that receive {c3_docs, total, *grand}: launch_counts3's job done by the same launch)
void launch_docoffs(const Launch& L, const int64_t* d_offs, int64_t n_docs, int64_t total, const int64_t* tile_base,
                    const uint64_t* docbits, const int64_t* docord_base, const int32_t* doc_tok, const int64_t* grand, int64_t* out_offs,
                    int64_t c3_docs = 0, int64_t* c3a = nullptr, int64_t* c3b = nullptr, int64_t* c3c = nullptr);
// the counts of two bitmaps per sub-tile in one pass; the scan of one or two count arrays -- a single launch up to 8,192 sub-tiles
void launch_doccount2(const Launch& L, const uint64_t* bits_a, const uint64_t* bits_b, int64_t nwords, int64_t total, int64_t nsub, int32_t* cnt_a, int32_t* cnt_b);
void launch_scan2(const Launch& L, int64_t ntiles, int64_t* bsum, const int32_t* cnt_a, int64_t* base_a, int64_t* grand_a, int round_to_a,
                  const int32_t* cnt_b, int64_t* base_b, int64_t* grand_b, int round_to_b, int kid);
void launch_rebase(const Launch& L, int64_t* offs, int64_t n, int64_t base);
// TKZ_OPT_CASE_EQUIVALENCE: a piece start behind every `'` + U+017F at whose apostrophe a match starts (cl100k on a .NET >= 7 host)
void launch_case_equiv_fix(const Launch& L, const uint8_t* d_bytes, int64_t total, const uint64_t* docbits, uint64_t* startbits);
void launch_miss_stats(const Launch& L, const EncodeParams& P, int64_t nsub);
void launch_counts3(const Launch& L, int64_t n_docs, int64_t total, const int64_t* grand, int64_t* out3, int64_t* out3b = nullptr, int64_t* out3c = nullptr);
// UTF-16 documents -> UTF-8 documents (Encoding.UTF8.GetBytes for a batch): lengths + group prefixes, then (after the scan of
// the tile sums) the bytes and the byte offset of every document
int64_t u16_tiles(int64_t total_units);
void launch_u16_len(const Launch& L, const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t ntiles, int32_t* grp_prefix, int32_t* tile_sum);
void launch_u16_write(const Launch& L, const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t ntiles, const int64_t* tile_base,
                      uint8_t* out, const int64_t* unit_offs, int64_t n_docs, const int32_t* grp_prefix, const int64_t* grand, int64_t* byte_offs);
// piece granularity: byte offset of every piece (n_pieces + 1 entries) and first piece of every document, from the bitmap
void launch_piece_index(const Launch& L, const uint64_t* startbits, int64_t nwords, int64_t total, int64_t nsub, const int64_t* ord_base,
                        int64_t n_pieces, int64_t* piece_offs, const int64_t* d_offs, int64_t n_docs, int64_t* doc_piece);
// batch Decode
int64_t dec_tiles(int64_t total_ids);
void launch_dec_len(const Launch& L, const TkzDecodeTable& D, const int32_t* ids, int64_t total, int64_t ntiles, int32_t* grp_prefix, int32_t* tile_sum);
void launch_dec_write(const Launch& L, const TkzDecodeTable& D, const int32_t* ids, int64_t total, int64_t ntiles, const int64_t* tile_base, uint8_t* out,
                      int64_t out_cap, const int64_t* id_offs, int64_t n_docs, const int32_t* grp_prefix, const int64_t* grand, int64_t* byte_offs, int32_t* counters);
void launch_corpus(hipStream_t s, int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int min_len, int max_len,
                   int64_t* d_offs, uint8_t* d_bytes, int64_t cap_bytes, int64_t* d_total);

}  // namespace tkz
