// tkz_kernels.hip -- the HIP kernels of the batch encode path, written for gfx950 (wave64, 256-thread
// workgroups, LDS-staged tiles) and their launchers.
//
// Pipeline for one batch of documents already resident in HBM (bytes = all documents back to back,
// offs[d] = first byte of document d):
//
//   k_docmark        doc offsets -> 1 bit per byte "a document starts here" (+ a sentinel bit at `total`)
//   k_pretok_rows    Regex.Matches for pattern 1 / cl100k, position-parallel (tkz_pretok.h (2))
//   k_pretok_seq     Regex.Matches for any pattern, one lane per document (tkz_pretok.h (1))
//                    -> 1 bit per byte "a piece starts here"
//   k_doccount/scan  documents and pieces that start in each 1 KiB sub-tile -> ordinal bases
//   k_probe          per sub-tile, one wavefront: enumerate pieces from the bitmap, whole-piece lookup
//                    (TikTokenizer.cs:262) -> one 32-bit record per piece
//   k_merge_short    BytePairEncode (BytePairEncoder.cs:13-76) of the missed pieces of <= 16 bytes, 64 per wavefront
//   k_giant_* / k_merge_coop   the missed pieces of > 1024 bytes (a workgroup each) / of 129..1024 bytes (a wavefront each)
//   k_long_count / k_long_scatter / k_merge_long_q   the missed pieces of 17..128 bytes of a LARGE batch, a lane each: binned by length class across the
//                    whole batch, then merged off that queue 64 of one class at a time (k_merge_long: the same batches formed inside a unit of a few
//                    sub-tiles -- a small batch, and the single-launch kernel)
//   k_scan_*         exclusive scan of the per-sub-tile token counts
//   k_place          ids stored at their final position
//   k_docoffs        out_offsets[d] = tile base + position inside the tile
#include "tkz_kernels.h"

#include <type_traits>

#include "tkz_bpe.h"
#include "tkz_classes.h"
#include "tkz_corpus.h"
#include "tkz_pretok.h"
#include "tkz_simt.h"

namespace tkz {

// -------------------------------------------------------------------------------------------------
// k_docmark
// -------------------------------------------------------------------------------------------------
TKZ_KERNEL(256) void k_docmark(const int64_t* offs, int64_t n_items, int64_t total, uint64_t* bits, int32_t* counters) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d <= n_items; d += stride) {
        const int64_t pos = offs[d];
        bool ok = pos >= 0 && pos <= total;
        if (d == 0) ok = ok && pos == 0;
        if (d == n_items) ok = ok && pos == total;
        else ok = ok && pos <= offs[d + 1];
        if (!ok) { simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrOffsets); continue; }
        // One plain store per bitmap word, by the first item that starts in it (the words were zeroed; 10 M atomics were a read-modify-write of nearly every
        // line of a 640 MB bitmap).  Offsets that are not monotone are reported by the item that sees it and the batch fails: what is stored then is never used.
        const int64_t w = pos >> 6;
        if (d > 0) { const int64_t prev = offs[d - 1]; if (prev >= 0 && prev <= pos && (prev >> 6) == w) continue; }
        uint64_t m = 1ull << (pos & 63);
        int64_t cur = pos;
        for (int64_t e = d + 1; e <= n_items;) {
            const int64_t q = offs[e];
            if (q < cur || (q >> 6) != w) break;
            if (q == cur) {                                // a run of empty documents: to its end by bisection (a million of them are 20 steps, not a million)
                int64_t lo = e + 1, hi = n_items + 1;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (offs[mid] <= cur) lo = mid + 1; else hi = mid; }
                e = lo;
                continue;
            }
            m |= 1ull << (q & 63);
            cur = q; ++e;
        }
        bits[w] = m;
    }
}

// -------------------------------------------------------------------------------------------------
// k_pretok_rows : one wave per kRowsPerWave rows of 64 bytes
// -------------------------------------------------------------------------------------------------
// classification of one row: general (any UTF-8) or, when its 64 bytes are all ASCII, from the flag table
struct RowInfo { TkzRowLane L; TkzRowMasks m; bool simple; };
template <int PATTERN>
TKZ_DEV RowInfo tkz_load_row(const TkzSrc& S, int64_t row, const uint8_t* bmp, const uint16_t* aflags) {
    RowInfo r;
    const int64_t pos = (row << 6) + simt::lane();
    const uint32_t b = tkz_gbyte(S, pos);
    const bool full = row >= 0 && ((row + 1) << 6) <= S.total;
    r.simple = full && simt::ballot(b >= 0x80u) == 0;
    if (r.simple) {
        const uint32_t f = aflags[b];
        r.L.pc = tkz_pc_of_flags(f); r.L.len = 1; r.L.off = 0; r.L.b = b; r.L.bad = 0;
        r.m = tkz_row_masks_ascii(f);
    } else {
        r.L = tkz_classify_byte(S, pos, bmp);
        r.m = tkz_row_masks(r.L, PATTERN == TKZ_PAT_CL100K);
    }
    return r;
}

// The sequential row loop over rows [r0, r1): rows one after the other, scan state carried in scalars.
template <int PATTERN>
TKZ_DEV void tkz_rows_sequential(const TkzSrc& S, const uint64_t* docbits, uint64_t* startbits, int64_t nrows, const uint8_t* bmp,
                                 const uint16_t* aflags, int64_t r0, int64_t r1, int32_t* counters) {
    const int lane = simt::lane();
    auto ds_at = [&](int64_t r) -> uint64_t { return (r >= 0 && r < nrows) ? docbits[r] : 0; };
    // warm-up start: the nearest row boundary at or before r0 across which no scan state flows
    // (the byte before it is neither a digit nor CR/LF, or a document starts exactly there)
    int64_t rw = r0;
    while (rw > 0) {
        if (ds_at(rw) & 1ull) break;
        // (first_lane: every lane looks at the same byte, but the compiler only knows that if told -- a lane-varying
        //  `rw` would drag every mask of the row loop from the scalar unit onto the VALU)
        const int pc = simt::first_lane(tkz_classify_byte(S, (rw << 6) - 1, bmp).pc);
        if (pc != PC_N && pc != PC_CRLF) break;
        --rw;
    }
    TkzScanCarry cy;
    cy.nb63 = 0; cy.carryN = 0; cy.abs63 = 0; cy.sa_from = -1; cy.sa_end = -1; cy.sa_lastcr = -1;
    int64_t row = rw > 0 ? rw - 1 : 0;                     // one extra row so the flags carried from "previous" are real
    RowInfo P = tkz_load_row<PATTERN>(S, row - 1, bmp, aflags);
    RowInfo C = tkz_load_row<PATTERN>(S, row, bmp, aflags);
    uint64_t dsP = ds_at(row - 1), dsC = ds_at(row);
    int clenP = 0, o1msP = 0, bad = 0;
    uint64_t c2P = 0, c3P = 0, o1P = 0;
    for (; row < r1; ++row) {
        const RowInfo N = tkz_load_row<PATTERN>(S, row + 1, bmp, aflags);
        const uint64_t dsN = ds_at(row + 1);
        int clenC, o1msC;
        uint64_t c2C, c3C, o1C, out;
        if (C.simple && N.simple) {
            out = tkz_row_eval_ascii<PATTERN>(P.m, C.m, N.m, dsC, dsN, c2P, c3P, o1P, &c2C, &c3C, &o1C, cy, S, docbits, nrows, bmp, row);
            clenC = (int)((c2C >> lane) & 1ull) * 2 + (int)((c3C >> lane) & 1ull) * 3;
            o1msC = (int)((o1C >> lane) & 1ull);
        } else {
            out = tkz_row_eval<PATTERN>(P.L, C.L, N.L, dsP, dsC, dsN, C.m, N.m, clenP, o1msP, &clenC, &o1msC, &c2C, &c3C, &o1C, cy,
                                        S, docbits, nrows, bmp, row);
        }
        if (row >= r0) {
            if (lane == 0) startbits[row] = out;
            bad |= C.L.bad | ((C.L.off != 0 && ((dsC >> lane) & 1ull)) ? 1 : 0);   // a document may not start inside a char
        }
        P = C; C = N; dsP = dsC; dsC = dsN; clenP = clenC; o1msP = o1msC; c2P = c2C; c3P = c3C; o1P = o1C;
    }
    if (simt::ballot(bad != 0) && lane == 0) simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrUtf8);
}

template <int PATTERN>
TKZ_KERNEL(64) void k_pretok_rows(const uint8_t* bytes, int64_t total, const uint64_t* docbits, uint64_t* startbits,
                                  int64_t nrows, const uint8_t* bmp, int32_t* counters, int64_t* xq, unsigned long long* xcount) {
    // One wavefront per workgroup (the chunk index is then blockIdx.x, which the compiler knows to be wave-uniform).
    // The wavefront owns output rows [r0, r0 + kRowsPerWave); it stages rows r0-1 .. r0+62 (one per lane) in LDS and
    // evaluates them all at once (tkz_block_eval); blocks that scheme refuses go through the sequential row loop.
    TKZ_SHARED uint16_t s_aflags[128];
    TKZ_SHARED uint4 s_blk[(65 * kBlockRowStride) / 16];
    for (int i = simt::tid(); i < 128; i += simt::nthreads()) s_aflags[i] = (uint16_t)tkz_ascii_flags((uint32_t)i, PATTERN == TKZ_PAT_CL100K);
    const int lane = simt::lane();
    const int64_t r0 = simt::bid() * kRowsPerWave;
    const int64_t first = r0 - 1;                          // row of lane 0
    const bool inside = ((first + 64) << 6) <= total;       // every staged row is a full row of the corpus
    if (inside) {
        for (int c = lane; c < 256; c += 64) {             // 16-byte chunk c of the 4 KiB block: row c/4, part c%4
            const int64_t pos = (first << 6) + 16 * (int64_t)c;
            uint4 v; v.x = v.y = v.z = v.w = 0;
            if (pos >= 0) v = tkz_load16(bytes + pos);
            s_blk[((c >> 2) * kBlockRowStride + (c & 3) * 16) / 16] = v;
        }
        if (lane < kBlockRowStride / 16) {                 // the row after the block: its first 16 bytes (a char of the last row may end there), zeros beyond
            uint4 z; z.x = z.y = z.z = z.w = 0;
            const int64_t pos = (first + 64) << 6;
            if (lane == 0 && pos + 16 <= total) z = tkz_load16(bytes + pos);
            s_blk[(64 * kBlockRowStride) / 16 + lane] = z;
        }
    }
    simt::sync();
    if (r0 >= nrows) return;
    if (inside) {
        const int64_t row = first + lane;
        const uint64_t ds = (row >= 0 && row < nrows) ? docbits[row] : 0;
        uint64_t out;
        bool done;
        if constexpr (PATTERN == TKZ_PAT_O200K) {
            TkzBlockCtx X; X.bytes = bytes; X.docbits = docbits; X.total = total; X.nrows = nrows; X.row0 = first;
            done = tkz_block_eval_o200k(reinterpret_cast<const uint8_t*>(s_blk), ds, X, &out);
        } else done = tkz_block_eval<PATTERN>(reinterpret_cast<const uint8_t*>(s_blk), ds, bmp, &out);
        if (done) {
            if (lane >= 1 && lane <= kRowsPerWave && row < nrows) startbits[row] = out;
            if (PATTERN == TKZ_PAT_O200K && lane == 0) reinterpret_cast<uint8_t*>(xq)[simt::bid()] = 0;
            return;
        }
    }
    const int64_t r1 = r0 + kRowsPerWave < nrows ? r0 + kRowsPerWave : nrows;
    if constexpr (PATTERN == TKZ_PAT_O200K) {
        // no row-sequential scanner for o200k: the block's rows start as the document-start bits and the block goes to
        // k_pretok_seq_blocks (sequential matcher from the nearest guaranteed match start)
        // (a flag per block, not a queue: one queue counter serialises a quarter of a million atomics per GB of non-ASCII text)
        for (int64_t r = r0 + lane; r < r1; r += 64) startbits[r] = docbits[r];
        if (lane == 0) reinterpret_cast<uint8_t*>(xq)[simt::bid()] = 1;
    } else {
        TkzSrc S;
        S.bytes = bytes; S.total = total; S.stage = nullptr; S.lo = 0; S.hi = 0;
        tkz_rows_sequential<PATTERN>(S, docbits, startbits, nrows, bmp, s_aflags, r0, r1, counters);
    }
}

// The same for ONE block by ONE wavefront of a larger workgroup (k_small): pattern 1 / cl100k only, wavefront-level synchronisation only.
constexpr int kPretokBlkQuads = (65 * kBlockRowStride) / 16;
template <int PATTERN>
TKZ_DEV void tkz_pretok_block(int64_t blk, uint4* s_blk, const uint16_t* s_aflags, const uint8_t* bytes, int64_t total, const uint64_t* docbits,
                              uint64_t* startbits, int64_t nrows, const uint8_t* bmp, int32_t* counters) {
    static_assert(PATTERN != TKZ_PAT_O200K, "o200k blocks need the two follow-up kernels");
    const int lane = simt::lane();
    const int64_t r0 = blk * kRowsPerWave;
    if (r0 >= nrows) return;
    const int64_t first = r0 - 1;                          // row of lane 0
    const bool inside = ((first + 64) << 6) <= total;       // every staged row is a full row of the corpus
    if (inside) {
        (void)simt::ballot(true);                          // (the block before this one is no longer read)
        for (int c = lane; c < 256; c += 64) {
            const int64_t pos = (first << 6) + 16 * (int64_t)c;
            uint4 v; v.x = v.y = v.z = v.w = 0;
            if (pos >= 0) v = tkz_load16(bytes + pos);
            s_blk[((c >> 2) * kBlockRowStride + (c & 3) * 16) / 16] = v;
        }
        if (lane < kBlockRowStride / 16) {
            uint4 z; z.x = z.y = z.z = z.w = 0;
            const int64_t pos = (first + 64) << 6;
            if (lane == 0 && pos + 16 <= total) z = tkz_load16(bytes + pos);
            s_blk[(64 * kBlockRowStride) / 16 + lane] = z;
        }
        (void)simt::ballot(true);
        const int64_t row = first + lane;
        const uint64_t ds = (row >= 0 && row < nrows) ? docbits[row] : 0;
        uint64_t out;
        if (tkz_block_eval<PATTERN>(reinterpret_cast<const uint8_t*>(s_blk), ds, bmp, &out)) {
            if (lane >= 1 && lane <= kRowsPerWave && row < nrows) startbits[row] = out;
            return;
        }
    }
    const int64_t r1 = r0 + kRowsPerWave < nrows ? r0 + kRowsPerWave : nrows;
    TkzSrc S;
    S.bytes = bytes; S.total = total; S.stage = nullptr; S.lo = 0; S.hi = 0;
    tkz_rows_sequential<PATTERN>(S, docbits, startbits, nrows, bmp, s_aflags, r0, r1, counters);
}

// o200k, second pass: the blocks k_pretok_rows<O200K> left over because they hold multi-byte chars, through the char-level block
// evaluator (tkz_block_eval_o200k_mb).  A kernel of its own so that its registers stay out of the ASCII scanner's.  What it
// refuses as well goes into the second queue, for k_pretok_seq_blocks.
TKZ_KERNEL_OCC(64, 3) void k_pretok_mb_blocks(const uint8_t* bytes, int64_t total, const uint64_t* docbits, uint64_t* startbits, int64_t nrows,
                                       const uint8_t* bmp, const uint8_t* xflag, int64_t nblk, unsigned long long* xcount, int64_t* xq2, unsigned long long* xcount2,
                                       int by_code_point) {
    TKZ_SHARED uint4 s_blk[(65 * kBlockRowStride) / 16];
    const int lane = simt::lane();
    unsigned long long mine = 0;
    for (int64_t blk = simt::bid(); blk < nblk; blk += simt::nblocks()) {
        if (xflag[blk] != 1) continue;                         // (one wavefront per workgroup: the branch is uniform)
        ++mine;
        const int64_t r0 = blk * kRowsPerWave, first = r0 - 1;
        bool done = false;
        uint64_t out = 0;
        simt::sync();                                          // (the previous block's rows are no longer read)
        // rows first .. first+63 and the first 16 bytes of the row behind them; what lies outside the corpus reads as 0 -- the
        // sentinel bit at `total` makes whatever follows the last document a document of its own, so the padding changes nothing
        for (int c = lane; c < 260; c += 64) {
            const int64_t pos = (first << 6) + 16 * (int64_t)c;
            uint4 v; v.x = v.y = v.z = v.w = 0;
            if (pos >= 0 && pos + 16 <= total) v = tkz_load16(bytes + pos);
            else if (pos >= 0 && pos < total) {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int j = 0; j < 16; ++j) if (pos + j < total) w[j >> 2] |= (uint32_t)bytes[pos + j] << (8 * (j & 3));
                v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
            }
            if (c < 256) s_blk[((c >> 2) * kBlockRowStride + (c & 3) * 16) / 16] = v;
            else {
                uint4 z; z.x = z.y = z.z = z.w = 0;
                s_blk[(64 * kBlockRowStride) / 16 + (c - 256)] = c == 256 ? v : z;
            }
        }
        if (lane == 0) { uint4 z; z.x = z.y = z.z = z.w = 0; s_blk[(64 * kBlockRowStride) / 16 + 4] = z; }
        simt::sync();
        {
            const int64_t row = first + lane;
            const uint64_t ds = (row >= 0 && row < nrows) ? docbits[row] : 0;
            TkzBlockCtx X; X.bytes = bytes; X.docbits = docbits; X.total = total; X.nrows = nrows; X.row0 = first;
            done = tkz_block_eval_o200k_mb(reinterpret_cast<const uint8_t*>(s_blk), ds, X, bmp, by_code_point != 0, &out);
            if (done && lane >= 1 && lane <= kRowsPerWave && row < nrows) startbits[row] = out;
        }
        if (!done && lane == 0) xq2[simt::atomic_add64(xcount2, 1ull)] = blk;
    }
    if (mine && lane == 0) simt::atomic_add64(xcount, mine);    // (statistics: tkz_encoder_pretok_leftovers)
}

// one document through the sequential matcher from match start p0; piece starts inside [b0, b1) are OR-ed into startbits
TKZ_DEV void tkz_seq_emit(int pattern, const TkzDoc& doc, int64_t a, int64_t p0, int64_t b0, int64_t b1, uint64_t* startbits) {
    int64_t curw = -1; uint64_t acc = 0;
    for (int64_t p = p0; p < doc.n && a + p < b1;) {
        const int64_t g = a + p;
        if (g >= b0) {
            if ((g >> 6) != curw) {
                if (acc) simt::atomic_or64((unsigned long long*)&startbits[curw], acc);
                curw = g >> 6; acc = 0;
            }
            acc |= 1ull << (g & 63);
        }
        p = tkz_match_at(pattern, doc, p);
    }
    if (acc) simt::atomic_or64((unsigned long long*)&startbits[curw], acc);
}

constexpr int kSeqBack = 768, kSeqFwd = 384, kSeqSeg = 62, kSeqWin = kSeqBack + kRowsPerWave * 64 + kSeqFwd;
static_assert(kSeqSeg * 64 == kRowsPerWave * 64 && kSeqWin % 16 == 0, "64 segments tile a block");
// byte i is certainly a match start when (byte i-1, byte i) are: CR/LF then an ASCII letter or digit; a non-white-space ASCII char then an
// ASCII white-space char that is not CR/LF; a non-digit ASCII char then an ASCII digit (no alternative of the patterns spans any of these)
TKZ_HD bool tkz_sync_rule(uint32_t pc, uint32_t c) {
    if (pc >= 0x80u || c >= 0x80u) return false;
    const bool c_ws = c == ' ' || c - 9u < 5u, pc_ws = pc == ' ' || pc - 9u < 5u;
    if ((pc == '\n' || pc == '\r') && ((c | 0x20u) - 'a' < 26u || c - '0' < 10u)) return true;
    if (c_ws && c != '\n' && c != '\r' && !pc_ws) return true;
    return c - '0' < 10u && !(pc - '0' < 10u);
}
// one block through the sequential matcher straight from HBM, by ONE lane: every document that overlaps [b0, b1) from the nearest sync
// point at or before b0
TKZ_DEV void tkz_seq_block_global(const uint8_t* bytes, const int64_t* offs, int64_t n_docs, int64_t total, uint64_t* startbits, int pattern,
                                  const uint8_t* bmp, int64_t b0, int64_t b1, int32_t* counters) {
    int64_t lo = 0, hi = n_docs;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (offs[mid + 1] <= b0) lo = mid + 1; else hi = mid; }
    for (int64_t d = lo; d < n_docs && offs[d] < b1; ++d) {
        const int64_t a = offs[d], e = offs[d + 1];
        if (e <= a || a < 0 || e > total) continue;
        TkzDoc doc; doc.b = bytes + a; doc.n = e - a; doc.bmp = bmp; doc.by_code_point = pattern == TKZ_PAT_O200K;
        int64_t p = 0;
        if (a < b0) {
            int64_t i = b0 - a;
            for (; i > 0; --i) if (tkz_sync_rule(doc.b[i - 1], doc.b[i])) break;
            p = i;
        }
        int bad = 0;
        int64_t v = a < b0 ? b0 - a : 0;                    // validate from the start of the char that contains the block's first byte
        for (int back = 0; v > 0 && back < 3 && (doc.b[v] & 0xC0) == 0x80; ++back) --v;
        for (; v < doc.n && a + v < b1;) { const TkzChar ch = tkz_doc_char(doc, v); bad |= ch.bad; v += ch.len; }
        if (bad) { simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrUtf8); continue; }
        tkz_seq_emit(pattern, doc, a, p, b0, b1, startbits);
    }
}

// The blocks k_pretok_rows<O200K> left over.  For every document that overlaps the block's
// bytes the sequential matcher runs from the nearest position at or before the block that is certainly a match start --
// the document start; a letter/digit right after a CR/LF; an ASCII non-CR/LF white-space char after a non-white-space
// ASCII char; an ASCII digit after a non-digit ASCII char (no alternative of the pattern can span any of these) -- and
// the piece starts that fall inside the block are OR-ed in (one wavefront per block).  When more than a quarter of all blocks were left over
// (text that is mostly non-ASCII) the kernel instead runs one lane per DOCUMENT over the whole batch: OR-ing the bits
// of blocks that were already done is harmless.
TKZ_KERNEL(256) void k_pretok_seq_blocks(const uint8_t* bytes, const int64_t* offs, int64_t n_docs, int64_t total, uint64_t* startbits,
                                         int64_t nrows, int pattern, const uint8_t* bmp, const int64_t* xq, const unsigned long long* xcount,
                                         int32_t* counters) {
    const int64_t nx = (int64_t)*xcount;
    const int64_t nblk = (nrows + kRowsPerWave - 1) / kRowsPerWave;
    const int64_t stride = simt::nblocks() * simt::nthreads();
    if (nx * 4 > nblk) {
        for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d < n_docs; d += stride) {
            const int64_t a = offs[d], e = offs[d + 1];
            if (e <= a || a < 0 || e > total) continue;
            TkzDoc doc; doc.b = bytes + a; doc.n = e - a; doc.bmp = bmp; doc.by_code_point = pattern == TKZ_PAT_O200K;
            int bad = 0;
            for (int64_t v = 0; v < doc.n;) { const TkzChar ch = tkz_doc_char(doc, v); bad |= ch.bad; v += ch.len; }
            if (bad) { simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrUtf8); continue; }
            tkz_seq_emit(pattern, doc, a, 0, a, e, startbits);
        }
        return;
    }
    // One WAVEFRONT per block.  The block (+ 768 bytes before it, 384 after) is staged in LDS -- the matcher reads a char at a time,
    // every read a dependent round trip when it goes to HBM: a lane took ~1 us per byte that way -- and cut into 64 segments of 62
    // bytes at sync points: lane l matches from the first sync point at or after the start of its segment up to the one lane l+1
    // starts from (lane 0: from the nearest sync point before the block).  When the window holds no sync point where one is needed
    // (text without blanks, digits or line breaks for hundreds of bytes) lane 0 does the block from HBM, as before.
    TKZ_SHARED uint4 s_win_all[kThreads / 64][kSeqWin / 16];
    uint8_t* const win = reinterpret_cast<uint8_t*>(s_win_all[simt::wave()]);
    const int lane = simt::lane();
    for (int64_t q = (simt::bid() * simt::nthreads() + simt::tid()) >> 6; q < nx; q += stride >> 6) {
        const int64_t r0 = xq[q] * kRowsPerWave;
        const int64_t r1 = r0 + kRowsPerWave < nrows ? r0 + kRowsPerWave : nrows;
        const int64_t b0 = r0 << 6, b1 = (r1 << 6) < total ? (r1 << 6) : total;
        if (b0 >= b1) continue;
        const int64_t wlo = b0 > kSeqBack ? b0 - kSeqBack : 0;             // (multiples of 64)
        const int64_t whi = b1 + kSeqFwd < total ? b1 + kSeqFwd : total;
        (void)simt::ballot(true);                                           // (the previous block's window is no longer read)
        for (int64_t p = wlo + 16 * lane; p < whi; p += 16 * 64) {
            uint4 v; v.x = v.y = v.z = v.w = 0;
            if (p + 16 <= total) v = tkz_load16(bytes + p);
            else {
                uint32_t w[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16; ++k) if (p + k < total) w[k >> 2] |= (uint32_t)bytes[p + k] << (8 * (k & 3));
                v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
            }
            *reinterpret_cast<uint4*>(win + (p - wlo)) = v;
        }
        (void)simt::ballot(true);
        auto wb = [&](int64_t abs) -> uint32_t { return win[abs - wlo]; };
        auto doc_of = [&](int64_t pos) -> int64_t {                         // the document that holds byte `pos` (documents tile [0, total))
            int64_t lo = 0, hi = n_docs;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (offs[mid + 1] <= pos) lo = mid + 1; else hi = mid; }
            return lo;
        };
        // where the matcher may stop reading: the end of the text, or 8 bytes behind the first sync point at or after the block's end
        int64_t cut = -1;
        if (whi == total) cut = total;
        else {
            int64_t f = -1;
            for (int64_t base = b1; base + 8 < whi && f < 0; base += 64) {
                const int64_t i = base + lane;
                const bool ok = i + 8 < whi && tkz_sync_rule(wb(i - 1), wb(i));
                const uint64_t m = simt::ballot(ok);
                if (m) f = base + tkz_ctz64(m);
            }
            const int64_t dn = doc_of(b1);                                  // a document start at or after b1 is a sync point too
            const int64_t ds1 = offs[dn] >= b1 ? offs[dn] : offs[dn + 1];
            if (ds1 + 8 < whi && (f < 0 || ds1 < f)) f = ds1;
            if (f >= 0) cut = f + 8;
        }
        // lane 0 starts from the nearest sync point at or before the block's first byte
        const int64_t d0 = doc_of(b0), a0 = offs[d0];
        int64_t start0 = -1;
        if (a0 == b0) start0 = b0;
        else {
            const int64_t lower = a0 > wlo ? a0 : wlo;                       // candidates i need byte i-1 of the same document inside the window
            for (int64_t base = b0; base > lower && start0 < 0; base -= 64) {
                const int64_t i = base - lane;
                const bool ok = i > lower && tkz_sync_rule(wb(i - 1), wb(i));
                const uint64_t m = simt::ballot(ok);
                if (m) start0 = base - tkz_ctz64(m);
            }
            if (start0 < 0 && a0 >= wlo) start0 = a0;
        }
        if (cut < 0 || start0 < 0) {                                        // no sync point in reach: the block from HBM, one lane
            if (lane == 0) tkz_seq_block_global(bytes, offs, n_docs, total, startbits, pattern, bmp, b0, b1, counters);
            continue;
        }
        // first sync point of every segment, then the first one at or after every segment's start (suffix minimum over the lanes)
        const int64_t t0 = b0 + (int64_t)kSeqSeg * lane, t1 = t0 + kSeqSeg < b1 ? t0 + kSeqSeg : b1;
        int64_t own = b1;
        if (lane > 0 && t0 < b1) {
            for (int64_t i = t0; i < t1; ++i) if (tkz_sync_rule(wb(i - 1), wb(i))) { own = i; break; }
            const int64_t dn = doc_of(t0);
            const int64_t ds1 = offs[dn] >= t0 ? offs[dn] : offs[dn + 1];
            if (ds1 < t1 && ds1 < own) own = ds1;
        }
        int64_t st = own;
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t o = ((int64_t)simt::shfl((int)(st >> 32), (lane + d) & 63) << 32) | (uint32_t)simt::shfl((int)st, (lane + d) & 63);
            if (lane + d < 64 && o < st) st = o;
        }
        int64_t lim = ((int64_t)simt::shfl((int)(st >> 32), (lane + 1) & 63) << 32) | (uint32_t)simt::shfl((int)st, (lane + 1) & 63);
        if (lane == 63) lim = b1;
        if (lane == 0) st = start0;
        // UTF-8 validation of the chars whose lead byte lies in my segment (lane 0: from the start of the char that holds the block's first byte)
        int bad = 0;
        if (t0 < b1) {
            int64_t v = t0;
            int64_t d = doc_of(v);
            if (lane == 0) { for (int back = 0; v > offs[d] && v > wlo && back < 3 && (wb(v) & 0xC0) == 0x80; ++back) --v; }
            else { for (int k = 0; k < 3 && v < t1 && v > offs[d] && v < offs[d + 1] && (wb(v) & 0xC0) == 0x80; ++k) ++v; }
            while (v < t1) {
                while (offs[d + 1] <= v) ++d;
                const int64_t a = offs[d], e = offs[d + 1];
                TkzDoc doc; doc.b = win + (a - wlo); doc.n = (e < cut ? e : cut) - a; doc.bmp = bmp; doc.by_code_point = pattern == TKZ_PAT_O200K;
                const TkzChar ch = tkz_doc_char(doc, v - a);
                bad |= ch.bad;
                v += ch.len;
            }
        }
        if (simt::ballot(bad != 0)) { if (lane == 0) simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrUtf8); continue; }
        // the matches that start in [st, lim)
        if (st < lim) {
            int64_t pos = st, d = doc_of(st), curw = -1;
            uint64_t acc = 0;
            while (pos < lim) {
                while (offs[d + 1] <= pos) ++d;
                const int64_t a = offs[d], e = offs[d + 1];
                TkzDoc doc; doc.b = win + (a - wlo); doc.n = (e < cut ? e : cut) - a; doc.bmp = bmp; doc.by_code_point = pattern == TKZ_PAT_O200K;
                if (pos >= b0) {
                    if ((pos >> 6) != curw) {
                        if (acc) simt::atomic_or64((unsigned long long*)&startbits[curw], acc);
                        curw = pos >> 6; acc = 0;
                    }
                    acc |= 1ull << (pos & 63);
                }
                pos = a + tkz_match_at(pattern, doc, pos - a);
            }
            if (acc) simt::atomic_or64((unsigned long long*)&startbits[curw], acc);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_pretok_seq : one lane per document (any pattern).  startbits must be pre-loaded with docbits.
// -------------------------------------------------------------------------------------------------
TKZ_KERNEL(256) void k_pretok_seq(const uint8_t* bytes, const int64_t* offs, int64_t n_docs, int64_t total, uint64_t* startbits,
                                  int pattern, const uint8_t* bmp, int32_t* counters) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d < n_docs; d += stride) {
        const int64_t a = offs[d], b = offs[d + 1];
        if (b <= a || a < 0 || b > total) continue;       // (bad offsets are reported by k_docmark)
        TkzDoc doc; doc.b = bytes + a; doc.n = b - a; doc.bmp = bmp; doc.by_code_point = pattern == TKZ_PAT_O200K;
        int bad = 0;
        for (int64_t p = 0; p < doc.n;) { const TkzChar c = tkz_doc_char(doc, p); bad |= c.bad; p += c.len; }
        if (bad) { simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrUtf8); continue; }
        int64_t curw = -1; uint64_t acc = 0;
        for (int64_t p = 0; p < doc.n;) {
            const int64_t g = a + p;
            if ((g >> 6) != curw) {
                if (acc) simt::atomic_or64((unsigned long long*)&startbits[curw], acc);
                curw = g >> 6; acc = 0;
            }
            acc |= 1ull << (g & 63);
            p = tkz_match_at(pattern, doc, p);
        }
        if (acc) simt::atomic_or64((unsigned long long*)&startbits[curw], acc);
    }
}

// -------------------------------------------------------------------------------------------------
// The encode stage: PROBE -> MERGE -> PLACE, three kernels joined by one record per piece and a short list of the pieces that missed.
//
// The per-piece driver of the reference (TikTokenizer.cs:250-274) does, for every regex match: whole-piece lookup (:262), else
// BytePairEncode (:268), then appends the ids (:256,264,269).  On the device those three steps have very different shapes --
// 94 % of the pieces of English/code text end at the lookup, a merge is a chain of ~10 dependent table gathers, and the append
// needs a prefix sum over everything before it -- so they are three kernels, each shaped for its own bottleneck, joined by
//
//   prank[p]   one 32-bit record per piece, in piece order.  Sub-tile s owns [pbase[s], pbase[s] + pcount[s]); pbase is the scan of the
//              counts ROUNDED UP TO 16 records, so that every sub-tile's records are whole 64-byte lines of its own: k_probe writes them
//              once, in full lines, k_place reads them once.
//              hit : rank | MARK                          (MARK: a document -- or, at piece granularity, a piece -- starts here)
//              miss: MISS | MARK | idx                    (idx: its entry in the sub-tile's miss list; LONG: the list of the 17+-byte misses)
//                    MISS | GIANT | MARK | relpos         (a piece of more than 1024 bytes: k_giant_*)
//   mlist[s * mcap + i]   the sub-tile's MISS LIST: the misses of <= 16 bytes from the front (i = 0, 1, ...), those of 17..1024 bytes from the
//              back (i = mcap - 1, mcap - 2, ...); mcount[s] = n_short | n_long << 16.  An entry is  relpos | (len - 1) << 10  when k_probe
//              writes it and  DONE | DENSE? | (count - 1) << 12 | offset  once its piece is merged (the merge kernels answer in place).
//              The merge kernels read ONLY these lists (~6 % of the pieces), never the records; mcap starts at 64 entries per sub-tile
//              and the batch is redone once with a larger one when a sub-tile needed more (text where nearly every piece misses).
//   mquad[s * mcap + i]   16 bytes beside every SHORT entry: k_probe leaves the piece's bytes there (it has them in LDS; k_merge_short then needs no
//              gather into the text: ONE coalesced load per entry instead of two scattered requests, twice), and k_merge_short leaves the
//              piece's tokens there when there are at most four (INLINE: k_place gets them in the round trip that fetches the answers).
//   dense[]    the tokens of the merged SHORT pieces of a group of 16 sub-tiles, packed in list order (4096 per group; a miss costs
//              ~4 tokens = one 16-byte store next to its neighbour's, not a 64-byte line of its own in k_merge_short and again in k_place)
//   tmp[b]     4 B per input byte, touched only under LONG missed pieces (and what overflows dense[]): a piece's tokens fit inside
//              its own byte span (tokens <= bytes)
//
//   k_probe        one wavefront per 1 KiB sub-tile, 32 wavefronts per CU: piece enumeration from the bitmap, the 13..28-byte pieces
//                  first (MID table, results parked in LDS), then SHORT probes (the first candidate bucket = two 16-byte gathers per
//                  piece, the second bucket only for the lanes that did not find their key; two batches of 64 pieces in flight; key
//                  dwords computed once and kept), every batch's 64 records stored as four full lines, misses appended to the list
//   k_merge_short  one wavefront per group of 16 sub-tiles: their short-miss lists are packed 64 to a wavefront -- looked up in
//                  the piece memo (the reference's LRUCache on the device), the survivors merged, every lane busy (tkz_bpe_lane) --
//                  instead of the ~14 of 64 a sub-tile has on its own.  Also the per-sub-tile token counts.
//   k_giant_order / k_giant_merge, k_long_* + k_merge_long_q (k_merge_long), k_merge_coop   the rare long ones: > 1024 bytes by a whole workgroup
//                  (tkz_bpe_long_tail: batches of merges that cannot disturb one another, rounds for chains of equal pairs); 17..128 bytes one LANE each in a
//                  span of an LDS arena, 64 pieces of one length class a batch; 129..1024 bytes one WAVEFRONT each, with the workgroup's merger
//   (scan of the token counts)
//   k_place        one wavefront per sub-tile: count per record -> prefix -> ids staged in LDS and stored as whole 16-byte quads at
//                  their final position, token index of every marked piece for the document offsets
// Workgroup b of k_probe / k_merge_short / k_place works on sub-tile range (b % 8) * (n / 8) + b / 8: consecutive sub-tiles -- which share the
// lines at the edges of their id and record ranges -- run on ONE XCD (workgroup b is dispatched to XCD b % 8) and meet in its L2.
// -------------------------------------------------------------------------------------------------
constexpr uint32_t kPrMiss = 1u << 31, kPrMark = 1u << 30, kPrLong = 1u << 29, kPrGiant = 1u << 28;
constexpr uint32_t kPrRankMask = (1u << 28) - 1u;       // ranks are < 2^27 (TKZ_MAX_RANK)
constexpr uint32_t kMrDone = 1u << 31, kMrDense = 1u << 30, kMrInline = 1u << 29;
constexpr int kMrLenShift = 10, kMrCntShift = 12;
constexpr uint32_t kMrOffMask = (1u << kMrCntShift) - 1u;
static_assert(kDenseCap <= (1 << kMrCntShift) && kSub <= (1 << kMrLenShift), "the offset field of a result entry");
#ifndef TKZ_PROBE_U
#define TKZ_PROBE_U 2
#endif
#ifndef TKZ_MS_THREADS
#define TKZ_MS_THREADS 256
#endif
#ifndef TKZ_PROBE_OCC
#define TKZ_PROBE_OCC 8
#endif
constexpr int kGroup = kMergeGroup;                     // sub-tiles per wavefront of k_merge_short
constexpr int kMsThreads = TKZ_MS_THREADS;              // ... and the workgroup size of that kernel
#ifndef TKZ_PROBE_PER
#define TKZ_PROBE_PER 1
#endif
constexpr int kProbePer = TKZ_PROBE_PER;                // consecutive sub-tiles per wavefront of k_probe (the next one's text is requested ahead)
constexpr int kMidMax = kSub / (TKZ_SHORT_KEY_MAX + 1) + 2;   // pieces of 13+ bytes that can start in one sub-tile

// The workgroup a block stands for: blocks b, b + 8, b + 16 ... of a grid run on one XCD (observed dispatch: XCD = b % 8; for speed only,
// nothing depends on it), so XCD x takes the x-th eighth of the work, in order.  nblocks is a multiple of 8.
#ifdef TKZ_NO_XCD_REMAP      // (development: A/B of the mapping)
TKZ_DEV int64_t tkz_xcd_block(int64_t b, int64_t nblocks) { return b; }
#else
TKZ_DEV int64_t tkz_xcd_block(int64_t b, int64_t nblocks) { return (b & 7) * (nblocks >> 3) + (b >> 3); }
#endif

// exclusive prefix over the wave of a small non-negative value (< 2^BITS), and the wave total;
// bit-sliced: one ballot + mbcnt per bit, no LDS traffic
template <int BITS>
TKZ_DEV int tkz_wave_scan(int v, int* total) {
    const uint64_t below = tkz_lowmask(simt::lane());
    int pre = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < BITS; ++j) {
        const uint64_t m = simt::ballot((v >> j) & 1);
        pre += tkz_popc64(m & below) << j;
        tot += tkz_popc64(m) << j;
    }
    *total = tot;
    return pre;
}

// exclusive prefix sum over the wave of any non-negative value (idle lanes contribute 0: every lane of the wave must be active)
TKZ_DEV int tkz_wave_scan_sum(int v, int* total) {
    const int x = simt::scan_inclusive(v);
    *total = simt::last_lane(x);
    return x - v;
}

// dword i of the key that starts at byte s of the LDS-staged sub-tile, zero-padded past len
TKZ_DEV uint32_t tkz_key_dword(const uint32_t* s_bytes, int s, int len, int i) {
    const int w = (s >> 2) + i;
    const uint32_t x = simt::alignbit(s_bytes[w + 1], s_bytes[w], (uint32_t)(s & 3) * 8u);
    const int nb = len - 4 * i;
    return nb >= 4 ? x : (nb <= 0 ? 0u : (x & ((1u << (8 * nb)) - 1u)));
}

// LDS of one wavefront of the probe stage (+ the mask table, shared by the workgroup)
constexpr int kProbeLdsBytes = (kSub + kHalo) + 2 * (kSub + 2) + 2 * kMidMax + 4 * kMidMax + 4 * (kSub / 32);     // per wavefront; a multiple of 4
constexpr int kProbeLdsQuads = (kProbeLdsBytes + 15) / 16;
struct ProbeLds { uint32_t* bytes; uint16_t* pstart; uint16_t* mid; uint32_t* midres; uint32_t* mark; const uint4* kmask; };
TKZ_DEV ProbeLds tkz_probe_lds(uint4* wave_quads, const uint4* kmask) {
    ProbeLds L;
    uint8_t* b = reinterpret_cast<uint8_t*>(wave_quads);
    L.bytes = reinterpret_cast<uint32_t*>(b); b += kSub + kHalo;
    L.midres = reinterpret_cast<uint32_t*>(b); b += 4 * kMidMax;
    L.mark = reinterpret_cast<uint32_t*>(b); b += 4 * (kSub / 32);
    L.pstart = reinterpret_cast<uint16_t*>(b); b += 2 * (kSub + 2);
    L.mid = reinterpret_cast<uint16_t*>(b);
    L.kmask = kmask;
    return L;
}
// byte mask of a zero-padded key of 0..12 bytes (three dwords): written by threads 0..12 of the workgroup, followed by a barrier
TKZ_DEV void tkz_probe_kmask_init(uint4* s_kmask) {
    if (simt::tid() <= TKZ_SHORT_KEY_MAX) {
        const int len = simt::tid();
        uint32_t m[3];
        for (int i = 0; i < 3; ++i) { const int nb = len - 4 * i; m[i] = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u)); }
        uint4 v; v.x = m[0]; v.y = m[1]; v.z = m[2]; v.w = 0;
        s_kmask[len] = v;
    }
}
// the text of a sub-tile (+ halo) in flight: 16 bytes per lane, the halo by lanes 0..3; bytes beyond the corpus read as 0
struct ProbeText { uint4 v0, v1; };
static_assert((kSub + kHalo) / 16 <= 128, "two 16-byte loads per lane cover the sub-tile and its halo");
TKZ_DEV void tkz_probe_request_text(const EncodeParams& P, int64_t sub, ProbeText* tx) {
    const int lane = simt::lane();
    const int64_t base = sub * kSub;
    uint4 v0, v1;
    v0.x = v0.y = v0.z = v0.w = 0; v1 = v0;
    const int64_t p0 = base + 16 * (int64_t)lane, p1 = base + 16 * (int64_t)(lane + 64);
    if (p0 + 16 <= P.total) v0 = tkz_load16_nt(P.bytes + p0);
    else {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; ++j) if (p0 + j < P.total) w[j >> 2] |= (uint32_t)P.bytes[p0 + j] << (8 * (j & 3));
        v0.x = w[0]; v0.y = w[1]; v0.z = w[2]; v0.w = w[3];
    }
    if (lane + 64 < (kSub + kHalo) / 16) {
        if (p1 + 16 <= P.total) v1 = tkz_load16_nt(P.bytes + p1);
        else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; ++j) if (p1 + j < P.total) w[j >> 2] |= (uint32_t)P.bytes[p1 + j] << (8 * (j & 3));
            v1.x = w[0]; v1.y = w[1]; v1.z = w[2]; v1.w = w[3];
        }
    }
    tx->v0 = v0; tx->v1 = v1;
}
// one sub-tile, by one wavefront (no workgroup barrier inside: every wavefront is on its own).  tx: the sub-tile's text, requested by the
// caller; on return it holds the request for sub-tile `next` (< 0: none), issued as soon as this one's text had moved into LDS -- a
// wavefront that probes consecutive sub-tiles never waits for text again.
// REPORT: a list that does not fit is reported here, by two atomics on the counter block (the single-launch kernel: one workgroup); the batch path
// leaves that to k_list_stats, which reads the lengths from mcount -- when nearly every sub-tile overflows (the first batch of text that is
// full of missed pieces) ten million atomics on one line cost 100 ms.
template <bool REPORT>
TKZ_DEV void tkz_probe_subtile(const TkzTables& T, const EncodeParams& P, int64_t sub, const ProbeLds& LD, ProbeText& tx, int64_t next) {
    const int lane = simt::lane();
    uint32_t* s_bytes = LD.bytes;
    uint16_t* s_pstart = LD.pstart;
    uint16_t* s_mid = LD.mid;
    uint32_t* s_midres = LD.midres;
    uint32_t* s_mark = LD.mark;
    const uint4* s_kmask = LD.kmask;
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(s_bytes);
    const int64_t base = sub * kSub;
    const int nb = (int)(P.total - base < kSub ? P.total - base : kSub);
    const uint8_t* gbase = P.bytes + base;
    const bool prof = TKZ_DEV_FLAG(P, 16);
    long long t_0 = prof ? simt::clock() : 0, t_1 = 0, t_2 = 0, t_3 = 0;
    // ---- everything the wavefront needs from memory is requested first, in ONE round trip: the sub-tile (+ halo), 16 B per lane
    // (the halo by lanes 0..3), the bitmap words one per lane, and the 64 bitmap words after the sub-tile in which the end of
    // its last piece is looked for -- and only then consumed ----
    // (the text itself was requested even earlier: by the caller, or while the sub-tile before this one was probed -- tx)
    uint4 v0 = tx.v0, v1 = tx.v1;
    uint64_t myword = 0, mymark = 0;                      // lanes 0..15: piece-start word / mark word of the sub-tile
    if (lane < kSub / 64) {
        const int64_t w = sub * (kSub / 64) + lane;
        if (w < P.nwords) { myword = P.startbits[w]; mymark = P.docbits[w]; }
    }
    const int64_t from = base + nb, w0 = from >> 6;        // the search for the end of the last piece starts here
    uint64_t ahead = (w0 + lane < P.nwords) ? P.startbits[w0 + lane] : 0ull;
    const int64_t pb = P.pbase[sub];
    // ---- consume ----
    { uint4* s4 = reinterpret_cast<uint4*>(s_bytes); s4[lane] = v0; if (lane + 64 < (kSub + kHalo) / 16) s4[lane + 64] = v1; }
    if (next >= 0) tkz_probe_request_text(P, next, &tx);
    if (lane < kSub / 64) {
        const int lim = nb - lane * 64;                    // bits at or beyond the end of the corpus are not pieces
        if (lim <= 0) myword = 0; else if (lim < 64) myword &= tkz_lowmask(lim);
        s_mark[2 * lane] = (uint32_t)mymark; s_mark[2 * lane + 1] = (uint32_t)(mymark >> 32);
    }
    // end of the last piece that starts here = first piece start at or after base+nb (the sentinel at `total` bounds it)
    int64_t last_end;
    {
        int64_t found = -1;
        for (int64_t c = 0; found < 0; ++c) {
            const int64_t w = w0 + c * 64 + lane;
            uint64_t v = c == 0 ? ahead : (w < P.nwords ? P.startbits[w] : 0);
            if (w == w0) v &= ~tkz_lowmask((int)(from & 63));
            const uint64_t any = simt::ballot(v != 0);
            if (any) {
                const int src = tkz_ctz64(any);
                const uint32_t lo = simt::shflu((uint32_t)v, src), hi = simt::shflu((uint32_t)(v >> 32), src);
                found = ((w0 + c * 64 + src) << 6) + tkz_ctz64(((uint64_t)hi << 32) | lo);
            } else if (w0 + (c + 1) * 64 >= P.nwords) found = P.total;
        }
        last_end = found;
    }
    // (anything beyond kArenaPiece bytes is a giant piece, whose real length k_giant_find takes from the bitmap)
    const int last_end_rel = (int)(last_end - base < 2 * kSub + 2 ? last_end - base : 2 * kSub + 2);
    // ---- ordered compaction of the piece starts (bit-sliced ballot prefix, no LDS scan) ----
    const uint32_t wlo = simt::shflu((uint32_t)myword, lane >> 2), whi = simt::shflu((uint32_t)(myword >> 32), lane >> 2);
    const uint32_t bits16 = (uint32_t)(((((uint64_t)whi << 32) | wlo) >> (16 * (lane & 3))) & 0xFFFFull);
    int np;
    {
        int off = tkz_wave_scan<5>(tkz_popc32(bits16), &np);
        for (uint32_t b = bits16; b; b &= b - 1) s_pstart[off++] = (uint16_t)(16 * lane + tkz_ctz32(b));
        if (lane == 0) s_pstart[np] = (uint16_t)last_end_rel;         // (sentinel: the length of piece k is s_pstart[k + 1] - s_pstart[k])
    }
    (void)simt::ballot(true);                              // (LDS written by other lanes of this wavefront is read below)
    if (prof) t_1 = simt::clock();
    const char* tb0 = reinterpret_cast<const char*>(T.short_slots);
    const uint32_t mid_off = (uint32_t)(reinterpret_cast<const char*>(T.mid_slots) - tb0);   // (SHORT and MID share one allocation)
    // ---- the pieces of 13+ bytes first: a few per cent of all pieces, twice the instructions of a short one (13..28 bytes: MID table;
    // 29..1024: the LONG table, rare).  Looked up together, 64 per batch, and their answers parked in LDS by ordinal: the main loop below
    // then handles nothing but keys of up to 12 bytes and stores every batch's 64 records as four FULL lines (patching these few records
    // in afterwards punched holes into every line: 1.7x the write traffic) ----
    int nmid = 0;
#pragma unroll 1
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int k = k0 + lane;
        int len = 0;
        if (k < np) len = (int)s_pstart[k + 1] - (int)s_pstart[k];
        const uint64_t midm = simt::ballot(len > TKZ_SHORT_KEY_MAX && len <= kArenaPiece);
        if ((midm >> lane) & 1ull) s_mid[nmid + tkz_popc64(midm & tkz_lowmask(lane))] = (uint16_t)k;
        nmid += tkz_popc64(midm);
    }
    (void)simt::ballot(true);
#pragma unroll 1
    for (int m0 = 0; m0 < nmid; m0 += 64) {
        const bool valid = m0 + lane < nmid;
        int s = 0, len = 13;
        uint32_t kk[7] = {0, 0, 0, 0, 0, 0, 0}, oa = 0, ob = 0;
        if (valid) {
            const int k = s_mid[m0 + lane];
            s = s_pstart[k];
            len = (int)s_pstart[k + 1] - s;
        }
        const bool is_mid = valid && len <= TKZ_MID_KEY_MAX;
        if (is_mid) {
#pragma unroll
            for (int i = 0; i < 7; ++i) kk[i] = tkz_key_dword(s_bytes, s, len, i);
            uint32_t s1, s2;
            tkz_mid_slots(T, kk, (uint32_t)len, &s1, &s2);
            oa = mid_off + 32u * s1; ob = mid_off + 32u * s2;
        }
        const uint4 a0 = tkz_load16(tb0 + oa), a1 = tkz_load16(tb0 + oa + 16u), b0 = tkz_load16(tb0 + ob), b1 = tkz_load16(tb0 + ob + 16u);
        int32_t rank = TKZ_RANK_NONE;
        if (is_mid) rank = tkz_match_mid(kk, (uint32_t)len, a0, a1, b0, b1);
        else if (valid) {
            if (s + len <= kSub + kHalo) rank = tkz_lookup_long(T, [&](int i) -> uint32_t { return sb[s + i]; }, (uint32_t)len);
            else rank = tkz_lookup_long(T, [&](int i) -> uint32_t { return gbase[s + i]; }, (uint32_t)len);
        }
        if (valid) s_midres[m0 + lane] = (uint32_t)rank;
    }
    (void)simt::ballot(true);
    if (prof) t_2 = simt::clock();
    // ---- the main loop: two batches of 64 pieces per iteration, their gathers in flight together (what a sub-tile costs is its count of
    // dependent round trips to the tables) ----
    uint32_t* const ml = P.mlist + sub * (int64_t)P.mcap;
    uint4* const mq = P.mquad + sub * (int64_t)P.mcap;
    int ns = 0, nl = 0, midseen = 0;
    int promo_extra = 0;                                   // tokens beyond one per piece that the promoted pieces of this sub-tile stand for (P.pextra): this lane's share
    bool giant = false, coopl = false;                     // coopl: a long miss of more than kLanePiece bytes (k_merge_coop's: bit 2 of the sub-tile's flag)
    constexpr int U = TKZ_PROBE_U;                         // batches of 64 pieces whose gathers are in flight together
#ifdef TKZ_DEVPROF
    // (development counters: where the main loop's ticks go and how many lanes each part has switched on -- the lane-cycle table of DESIGN.md 3)
    long long pf_first = 0, pf_second = 0, pf_emit = 0, pf_t = 0;
    int pf_batches = 0, pf_valid = 0, pf_want2 = 0, pf_iters2 = 0;
#define TKZ_PROBE_MARK(acc) do { if (prof) { const long long pf_now = simt::clock(); acc += pf_now - pf_t; pf_t = pf_now; } } while (0)
    if (prof) pf_t = simt::clock();
#else
#define TKZ_PROBE_MARK(acc) do { } while (0)
#endif
#pragma unroll 1
    for (int k0 = 0; k0 < np; k0 += 64 * U) {
        int ps[U], plen[U];
        uint32_t hs[U], kw0[U], kw1[U], kw2[U];
        uint4 a0[U], a1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 64 * u + lane;
            ps[u] = 0; plen[u] = 0;
            if (k < np) { ps[u] = s_pstart[k]; plen[u] = (int)s_pstart[k + 1] - ps[u]; }
            // the key's three dwords, zero-padded: four aligned dwords, three v_alignbit, one mask row -- computed ONCE and kept
            const int s = ps[u], w = s >> 2;
            const uint32_t sh = (uint32_t)(s & 3) * 8u;
            const uint32_t d0 = s_bytes[w], d1 = s_bytes[w + 1], d2 = s_bytes[w + 2], d3 = s_bytes[w + 3];
            const bool is_short = plen[u] >= 1 && plen[u] <= TKZ_SHORT_KEY_MAX;
            const uint4 km = s_kmask[is_short ? plen[u] : 0];
            kw0[u] = simt::alignbit(d1, d0, sh) & km.x; kw1[u] = simt::alignbit(d2, d1, sh) & km.y; kw2[u] = simt::alignbit(d3, d2, sh) & km.z;
            const uint32_t slot = tkz_short_slot_first(T, kw0[u], kw1[u], kw2[u], (uint32_t)plen[u], &hs[u]);
            // the FIRST candidate bucket of every piece (idle lanes and longer pieces gather slot 0: a load inside a divergent branch
            // is waited for inside it)
            const uint32_t oa = is_short ? 16u * slot : 0u;
            a0[u] = tkz_load16(tb0 + oa); a1[u] = tkz_load16(tb0 + oa + 16u);
        }
        int32_t rk[U];
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool is_short = plen[u] >= 1 && plen[u] <= TKZ_SHORT_KEY_MAX;
            rk[u] = tkz_match_short2x(kw0[u], kw1[u], kw2[u], (uint32_t)plen[u], a0[u], a1[u]);       // (a piece that is not short matches no slot: len 0 or > 12)
            more = more || (is_short && rk[u] == TKZ_RANK_NONE);
        }
        // the second bucket, for the lanes the first one did not settle only (the keys the builder could not keep in their first
        // bucket, and the pieces that are not keys at all): ~15 % of the requests of the first step instead of another 100 %
        bool two_pass = simt::ballot(more) != 0;
        TKZ_PROBE_MARK(pf_first);
#ifdef TKZ_DEVPROF
        if (prof) {
            for (int u = 0; u < U; ++u) { const int nv = np - (k0 + 64 * u); if (nv > 0) { ++pf_batches; pf_valid += nv < 64 ? nv : 64; } }
            for (int u = 0; u < U; ++u) pf_want2 += tkz_popc64(simt::ballot(plen[u] >= 1 && plen[u] <= TKZ_SHORT_KEY_MAX && rk[u] == TKZ_RANK_NONE));
            pf_iters2 += two_pass ? 1 : 0;
            pf_t = simt::clock();
        }
#endif
        if (two_pass) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool want = plen[u] >= 1 && plen[u] <= TKZ_SHORT_KEY_MAX && rk[u] == TKZ_RANK_NONE;
                if (want) {
                    const uint32_t ob = 16u * tkz_short_slot_second(T, hs[u]);
                    a0[u] = tkz_load16(tb0 + ob); a1[u] = tkz_load16(tb0 + ob + 16u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool want = plen[u] >= 1 && plen[u] <= TKZ_SHORT_KEY_MAX && rk[u] == TKZ_RANK_NONE;
                if (want) rk[u] = tkz_match_short2x(kw0[u], kw1[u], kw2[u], (uint32_t)plen[u], a0[u], a1[u]);
            }
        }
        TKZ_PROBE_MARK(pf_second);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 64 * u + lane;
            const bool valid = k < np;
            const int s = ps[u], len = plen[u];
            const bool is_mid = len > TKZ_SHORT_KEY_MAX && len <= kArenaPiece;
            const uint64_t midm = simt::ballot(is_mid);
            int32_t rank = rk[u];                                                // Encoder.TryGetValue(piece) (TikTokenizer.cs:262)
            if (is_mid) rank = (int32_t)s_midres[midseen + tkz_popc64(midm & tkz_lowmask(lane))];
            midseen += tkz_popc64(midm);
            const bool is_giant = len > kArenaPiece;                             // (k_giant_merge looks a giant piece up itself)
            const bool miss = valid && !is_giant && rank == TKZ_RANK_NONE;
            if (P.pextra) {                                                      // (wave-uniform: only when the tables hold promoted pieces)
                const uint32_t ex = (valid && !is_giant && !miss && ((uint32_t)rank & kPromoFlag)) ? (((uint32_t)rank >> kPromoCntShift) & 3u) : 0u;
                promo_extra += (int)ex;                                          // (per lane: one add a batch; summed over the wavefront once, at the end)
            }
            const bool miss_s = miss && len <= kShortMax, miss_l = miss && len > kShortMax;
            const uint64_t sm = simt::ballot(miss_s), lm = simt::ballot(miss_l);
            const int is_ = ns + tkz_popc64(sm & tkz_lowmask(lane)), il = nl + tkz_popc64(lm & tkz_lowmask(lane));
            ns += tkz_popc64(sm); nl += tkz_popc64(lm);
            // (the two lists grow towards each other; an entry that would run into the other list is not written -- ns + nl > mcap is then
            //  reported below and the batch redone with longer lists)
            const uint32_t ent = (uint32_t)s | ((uint32_t)(len - 1) << kMrLenShift);
            if (miss_s && is_ + nl < P.mcap) {
                ml[is_] = ent;
                uint4 kq; kq.x = kw0[u]; kq.y = kw1[u]; kq.z = kw2[u]; kq.w = 0;          // the piece's bytes, zero-padded, for k_merge_short
                if (len > TKZ_SHORT_KEY_MAX) { kq.x = tkz_key_dword(s_bytes, s, len, 0); kq.y = tkz_key_dword(s_bytes, s, len, 1); kq.z = tkz_key_dword(s_bytes, s, len, 2); kq.w = tkz_key_dword(s_bytes, s, len, 3); }
                mq[is_] = kq;
            }
            if (miss_l && il + ns < P.mcap) ml[P.mcap - 1 - il] = ent;
            uint32_t rec = ((s_mark[s >> 5] >> (s & 31)) & 1u) ? kPrMark : 0u;
            if (REPORT) giant = giant || (miss_l && len > kSmallLanePiece);      // (k_small, the only REPORT user, hands a batch with such a piece back like one with a giant piece: k_merge_coop is a kernel of the batch path)
            else coopl = coopl || (miss_l && len > P.lane_piece);
            if (is_giant) { rec |= kPrMiss | kPrGiant | (uint32_t)s; giant = giant || valid; }
            else if (miss) rec |= kPrMiss | (miss_l ? (kPrLong | (uint32_t)il) : (uint32_t)is_);
            else rec |= (uint32_t)rank;
            if (valid && pb + k < P.prank_cap) tkz_store_nt(&P.prank[pb + k], (int32_t)rec);
        }
        TKZ_PROBE_MARK(pf_emit);
    }
#undef TKZ_PROBE_MARK
#ifdef TKZ_DEVPROF
    if (prof && lane == 0) {
        simt::atomic_add64(&P.devprof[48], (unsigned long long)pf_first); simt::atomic_add64(&P.devprof[49], (unsigned long long)pf_second);
        simt::atomic_add64(&P.devprof[50], (unsigned long long)pf_emit); simt::atomic_add64(&P.devprof[51], (unsigned long long)pf_batches);
        simt::atomic_add64(&P.devprof[52], (unsigned long long)pf_valid); simt::atomic_add64(&P.devprof[53], (unsigned long long)pf_want2);
        simt::atomic_add64(&P.devprof[54], (unsigned long long)pf_iters2);
    }
#endif
    if (prof && lane == 0) {
        t_3 = simt::clock();
        simt::atomic_add64(&P.devprof[0], 1); simt::atomic_add64(&P.devprof[1], (unsigned long long)(t_3 - t_0));
        simt::atomic_add64(&P.devprof[2], (unsigned long long)(t_1 - t_0)); simt::atomic_add64(&P.devprof[3], (unsigned long long)(t_3 - t_2));
        simt::atomic_add64(&P.devprof[4], (unsigned long long)(t_2 - t_1)); simt::atomic_add64(&P.devprof[5], (unsigned long long)nmid);
        simt::atomic_add64(&P.devprof[6], (unsigned long long)np);
    }
    // sub-tiles with long misses (bit 0: statistics only, the lists say it), a giant piece (bit 1: k_giant_find, k_merge_long, k_place), a long miss of more
    // than kLanePiece bytes (bit 2: k_merge_coop -- k_merge_long used to flag those chunks itself: a pointer, an index and a flag alive across its merge
    // loops, 14 more spilled scalars and 4 % of that kernel on mixed text)
    const uint32_t f = (nl ? 1u : 0u) | (simt::ballot(giant) ? 2u : 0u) | (simt::ballot(coopl) ? 4u : 0u);
    if (P.pextra) { int tot; (void)tkz_wave_scan_sum(promo_extra, &tot); promo_extra = tot; }      // (wave-uniform branch)
    if (lane == 0) {
        P.heavy_flag[sub] = (uint8_t)f;
        P.mcount[sub] = (uint32_t)ns | ((uint32_t)nl << 16);
        if (P.pextra) P.pextra[sub] = promo_extra;
        if (!REPORT && P.tc_atomic) P.tile_count[sub] = 0;                   // (the merge kernels of both streams add to it: launch_encode)
        if (REPORT && ns + nl > P.mcap) { simt::atomic_or((unsigned*)&P.counters[0], (unsigned)kErrMissCap); simt::atomic_max((unsigned*)&P.counters[1], (unsigned)(ns + nl)); }
        if (pb + np > P.prank_cap) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)kErrCapacity);
    }
}
TKZ_KERNEL_OCC(256, TKZ_PROBE_OCC) void k_probe(TkzTables T, EncodeParams P) {
    TKZ_SHARED uint4 s_wave[kThreads / 64][kProbeLdsQuads];
    TKZ_SHARED uint4 s_kmask[TKZ_SHORT_KEY_MAX + 1];
    tkz_probe_kmask_init(s_kmask);
    simt::sync();                                         // (the only workgroup barrier: from here on every wavefront is on its own)
    const int64_t sub0 = (tkz_xcd_block(simt::bid(), simt::nblocks()) * (kThreads / 64) + simt::wave()) * kProbePer;
    if (sub0 >= P.nsub) return;
    const ProbeLds LD = tkz_probe_lds(s_wave[simt::wave()], s_kmask);
    ProbeText tx;
    tkz_probe_request_text(P, sub0, &tx);
#pragma unroll 1
    for (int it = 0; it < kProbePer && sub0 + it < P.nsub; ++it) {
        const int64_t nxt = (it + 1 < kProbePer && sub0 + it + 1 < P.nsub) ? sub0 + it + 1 : -1;
        tkz_probe_subtile<false>(T, P, sub0 + it, LD, tx, nxt);
        (void)simt::ballot(true);                             // (the next sub-tile reuses the wavefront's LDS)
    }
}

// the 16 bytes of the corpus that start at byte position `abs`, as four little-endian dwords: the five dwords around them are fetched
// with TWO requests per lane (16 + 4 bytes from the dword-aligned address; a scattered gather costs a request per lane and instruction)
struct alignas(4) TkzDwords4 { uint32_t x, y, z, w; };
TKZ_DEV void tkz_load_piece16(const uint8_t* bytes, int64_t total, int64_t abs, uint32_t* pw) {
    const int64_t a0 = abs & ~(int64_t)3;
    const uint32_t sh = (uint32_t)(abs & 3) * 8u;
    uint32_t w[5];
    if (a0 + 20 <= total) {
        const TkzDwords4 v = *reinterpret_cast<const TkzDwords4*>(bytes + a0);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        w[4] = *reinterpret_cast<const uint32_t*>(bytes + a0 + 16);
    } else {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int64_t p = a0 + 4 * i;
            if (p + 4 <= total) w[i] = *reinterpret_cast<const uint32_t*>(bytes + p);
            else { w[i] = 0; for (int j = 0; j < 4; ++j) if (p + j < total) w[i] |= (uint32_t)bytes[p + j] << (8 * j); }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[i] = simt::alignbit(w[i + 1], w[i], sh);
}

// the result a merge kernel leaves in a miss-list entry: the piece's tokens wait in dense[group * kDenseCap + off ..) (DENSE) or in tmp at
// the piece's own byte position (off = relpos)
TKZ_HD uint32_t tkz_result_entry(bool dense, int cnt, int off) { return kMrDone | (dense ? kMrDense : 0u) | ((uint32_t)(cnt - 1) << kMrCntShift) | (uint32_t)off; }
TKZ_HD uint32_t tkz_result_inline(int cnt) { return kMrDone | kMrInline | ((uint32_t)(cnt - 1) << kMrCntShift); }       // <= 4 tokens, in the entry's quad
TKZ_HD int tkz_result_cnt(uint32_t r) { return (int)((r >> kMrCntShift) & 1023u) + 1; }
TKZ_HD int tkz_result_off(uint32_t r) { return (int)(r & kMrOffMask); }

// largest q in [0, N) with pre[q] <= g  (pre: non-decreasing exclusive prefix sums in LDS, g below the total): the list a position belongs to
template <int N, class Pre>
TKZ_DEV int tkz_find_list(const Pre* pre, int g) {
    int q = 0;
#pragma unroll
    for (int step = N / 2; step >= 1; step >>= 1) if (pre[q + step] <= g) q += step;
    return q;
}

// k_merge_short: one wavefront per group of kGroup = 16 sub-tiles.  It walks the group's short-miss lists (k_probe's mlist: ~14 entries per
// sub-tile) 64 entries at a time; the entries that have not been looked up yet go through the PIECE MEMO (the device form of the
// reference's LRUCache, tkz_tables.h): a hit has its <= 4 tokens at once, the survivors stay in the wave's list, and only when that is
// full of survivors are 64 of them merged, one per lane (BytePairEncode, TikTokenizer.cs:268).  Merged pieces of <= 4 tokens claim
// their memo slot if it is empty.  On the bench corpus 3 of 4 missed pieces are memo hits: a merge costs ~45 scattered gathers and ~8
// dependent round trips, a memo lookup 4 and one.  Every entry is answered in place (tkz_result_entry); the records are never read.
// (LDS: 4 x 9.25 KB of merge state + 2.3 KB = 40 KB per workgroup, and <= 128 VGPRs: four workgroups = 16 wavefronts per CU)
constexpr int kMsStride = TkzBpeGeom<16>::kStride, kMsIdStride = TkzBpeGeom<16>::kIdStride;
#ifndef TKZ_MS_SHORT_LEN
#define TKZ_MS_SHORT_LEN 8
#endif
#ifndef TKZ_MS_TWO_MIN
#define TKZ_MS_TWO_MIN 320
#endif
constexpr int kMsTwoListsMin = TKZ_MS_TWO_MIN;         // ... when the group's lists hold at least this many entries
constexpr int kMsShortLen = TKZ_MS_SHORT_LEN;          // k_merge_short's two lists: pieces of up to this many bytes | longer ones
// LDS of one wavefront of k_merge_short (+ the byte-id table shared by the workgroup)
struct MsLds { uint4* pr; uint32_t* ids; uint16_t* rec; uint16_t* idx; int* extra; uint16_t* pre; const uint16_t* brank16; };
// (four workgroups of four wavefronts a CU: 40,960 bytes a workgroup with the 512 of the byte-id table -- 10,112 a wavefront at most)
constexpr int kMsLdsBytes = kMsStride * 64 * 4 + kMsIdStride * 64 * 4 + 2 * (2 * 64 + 2 * 64) + 4 * kGroup + 2 * (kGroup + 1);   // per wavefront, rounded to 16 below
constexpr int kMsLdsQuads = (kMsLdsBytes + 15) / 16;
static_assert((kMsThreads / 64) * kMsLdsQuads * 16 + 512 <= 40960, "k_merge_short: four workgroups a CU (160 KB of LDS, allocated in units of 1,280 bytes)");
TKZ_DEV MsLds tkz_ms_lds(uint4* wave_quads, const uint16_t* brank16) {
    MsLds L;
    uint8_t* b = reinterpret_cast<uint8_t*>(wave_quads);
    L.pr = reinterpret_cast<uint4*>(b); b += kMsStride * 64 * 4;              // per lane pr[16] at a conflict-free stride for 16-byte reads
    L.ids = reinterpret_cast<uint32_t*>(b); b += kMsIdStride * 64 * 4;        // per lane ids[16]
    L.extra = reinterpret_cast<int*>(b); b += 4 * kGroup;                     // tokens the merges added to every sub-tile of the group
    // the wave's TWO lists of pieces that missed the memo (up to 8 bytes | 9..16 bytes), 64 entries each: relpos | (len-1) << 10, and
    // (sub-tile of the group) << 10 | index in its miss list
    L.rec = reinterpret_cast<uint16_t*>(b); b += 2 * 2 * 64;
    L.idx = reinterpret_cast<uint16_t*>(b); b += 2 * 2 * 64;
    L.pre = reinterpret_cast<uint16_t*>(b);                                   // exclusive prefix of the lists' lengths (16 lists of at most 1,024 entries)
    L.brank16 = brank16;
    return L;
}
// the id of every single byte in LDS, not 16 gathers per piece (0xFFFF: not a key, 0xFFFE: too large for 16 bits, look it up); by the whole
// workgroup, followed by a barrier
TKZ_DEV void tkz_ms_brank_init(const TkzTables& T, uint16_t* s_brank16) {
    for (int i = simt::tid(); i < 256; i += simt::nthreads()) {
        const uint32_t v = (uint32_t)T.byte_rank[i];
        s_brank16[i] = (uint16_t)(v >= (uint32_t)TKZ_PSEUDO_BASE ? 0xFFFFu : (v < 0xFFFEu ? v : 0xFFFEu));
    }
}
// one group of kGroup sub-tiles, by one wavefront
TKZ_DEV void tkz_merge_short_group(const TkzTables& T, const EncodeParams& P, int64_t grp, const MsLds& LD) {
    constexpr int NMAX = 16, STRIDE = kMsStride, IDSTRIDE = kMsIdStride;
    static_assert(kGroup == 16 && kShortMax == 16, "list entries: 4 bits of sub-tile, 10 of index; 10 of relpos, 4 of length, 1 flag");
    const int lane = simt::lane();
    const uint16_t* s_brank16 = LD.brank16;
    auto byte_id = [&](uint32_t b) -> uint32_t {
        const uint32_t v = s_brank16[b];
        return v < 0xFFFEu ? v : (v == 0xFFFFu ? (uint32_t)TKZ_PSEUDO_BASE + b : (uint32_t)T.byte_rank[b]);
    };
    auto pair_rank = [&](uint32_t b0, uint32_t b1) -> int32_t { return T.bytepair_rank[(b0 << 8) | b1]; };
    const int64_t sub0 = grp * kGroup;
    if (sub0 >= P.nsub) return;
    uint16_t* s_rec = LD.rec;
    uint16_t* s_idx = LD.idx;
    int* s_extra = LD.extra;
    uint16_t* s_pre = LD.pre;
    uint32_t* ids = LD.ids + lane * IDSTRIDE;
    uint32_t* pr = reinterpret_cast<uint32_t*>(LD.pr) + lane * STRIDE;
    // lane q < kGroup looks after sub-tile q of the group: how many pieces start there, how long its short-miss list is
    int my_np = 0, my_ns = 0;
    if (lane < kGroup && sub0 + lane < P.nsub) { my_np = P.pcount[sub0 + lane]; my_ns = (int)(P.mcount[sub0 + lane] & 0xFFFFu); }
    // (a list longer than mcap was cut by k_probe, which has reported it: the batch is redone; keep to what was written)
    { const int nl_ = lane < kGroup && sub0 + lane < P.nsub ? (int)(P.mcount[sub0 + lane] >> 16) : 0; if (my_ns + nl_ > P.mcap) my_ns = 0; }
    int ntotal;
    { const int pre = tkz_wave_scan_sum(my_ns, &ntotal); if (lane <= kGroup) s_pre[lane] = (uint16_t)pre; }
    if (lane < kGroup) s_extra[lane] = 0;
    uint32_t* const ml0 = P.mlist + sub0 * (int64_t)P.mcap;
    int err = 0, dused = 0, done = 0;
    int nl0 = 0, nl1 = 0;                                       // entries in the wave's two lists (pieces of up to kMsShortLen bytes | longer ones): all have missed the memo
    int st_look = 0, st_hit = 0;                                // TKZ_OPT_PIECE_STATS: memo lookups and hits of this group (one pair of atomics at its end)
    int32_t* const dense = P.dense + grp * kDenseCap;
    const bool memo = T.memo_n != 0;
    // a LEARNING batch (tkz_api.cpp, promote_from_memo): hits of every eighth group are sampled per memo slot (memo_phase below) -- the host promotes
    // the hottest entries into the SHORT / MID tables afterwards.  Null in every other batch.
    const bool sparse = T.memo_hits_sparse != 0;                            // (a large batch: every eighth group, one lane a time; a small one: every hit)
    const bool count_hits = T.memo_hits != nullptr && (!sparse || (grp & 7) == 0);       // (wave-uniform)
    // where the `cnt` tokens of a piece go -- up to four: into the entry's own quad (the caller stores them); more: packed behind those of
    // the pieces before it in the group's dense region (in tmp, at the piece's own byte position, once that is full) -- and the answer in
    // its list entry; every lane of the wavefront calls it (a scan inside)
    auto assign = [&](bool have, int cnt, int si, int j, int rel) -> int32_t* {
        int btot;
        const bool big = have && cnt > 4;
        const int doff = dused + tkz_wave_scan_sum(big ? cnt : 0, &btot);
        int32_t* dst = nullptr;
        if (have) {
            if (!big) ml0[si * (int64_t)P.mcap + j] = tkz_result_inline(cnt);
            else {
                const bool packed = doff + cnt <= kDenseCap;
                dst = packed ? dense + doff : P.tmp + ((sub0 + si) * kSub + rel);
                ml0[si * (int64_t)P.mcap + j] = tkz_result_entry(packed, cnt, packed ? doff : rel);
            }
            if (cnt > 1) simt::atomic_add(&s_extra[si], cnt - 1);
        }
        dused += btot;
        return dst;
    };
    // the piece of a list entry: its 16 bytes, zero beyond its length, as k_probe left them beside the entry; *nul = it holds a zero byte
    uint4* const mq0 = P.mquad + sub0 * (int64_t)P.mcap;
    auto key_of = [&](uint4 kq, int len, uint32_t* kw, bool* nul) {
        kw[0] = kq.x; kw[1] = kq.y; kw[2] = kq.z; kw[3] = kq.w;
        uint32_t z = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nb = len - 4 * i;
            const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
            z |= (kw[i] - 0x01010101u) & ~kw[i] & 0x80808080u & m;
        }
        *nul = z != 0;
    };
    // the entries that have not been through the memo: hits are done (tokens, answer), the others stay in the list, in order.
    // A piece that holds a zero byte never uses the memo -- neither looked up nor inserted: a slot's words are either still zero or final
    // (an entry never changes once valid), so whatever mixture of old and new words a reader may be handed -- the key and the value are
    // two loads, and a slot can be claimed between them -- a key without zero bytes inside its length can only ever equal a COMPLETE key of the
    // same length, and then the value it is paired with is either that key's (valid) or not valid at all.
    // (r16 / ix / fkq: the entry and the quad of the `take` lanes that hold a NEW entry, fetched together by those very lanes: the memo slot is the only
    //  round trip between taking an entry and answering it.)  Returns, per lane, whether its piece has to be merged; the caller lists it.
    auto memo_phase = [&](int take, uint32_t r16, uint32_t ix, uint4 fkq) -> bool {
        (void)simt::ballot(true);
        const bool mine = lane < take;
        const int si = (int)(ix >> 10), j = (int)(ix & 1023u), rel = (int)(r16 & 1023u), len = (int)((r16 >> 10) & 15u) + 1;
        bool hit = false;
        uint32_t hit_slot = 0;
        uint4 vv; vv.x = vv.y = vv.z = vv.w = 0;
        if (mine && memo) {
            uint32_t kw[4];
            bool nul;
            key_of(fkq, len, kw, &nul);
            const uint32_t b = tkz_mulhi(tkz_hash_memo(kw, (uint32_t)len), T.memo_n / kMemoWays) * kMemoWays;
#pragma unroll
            for (int wy = 0; wy < (int)kMemoWays; ++wy) {
                const TkzMemoSlot* slot = &T.memo[b + wy];
                const uint4 kk = tkz_load16(&slot->k[0]);
                const uint4 v = tkz_load16(&slot->v[0]);
                // (the valid tag is in ALL FOUR value words: a value whose 16-byte store became visible word by word -- nothing in HIP's memory
                //  model says a dwordx4 access is single-copy atomic across CUs / XCDs -- is not valid until its last word is there)
                const bool h = !nul && ((v.x & v.y & v.z & v.w) & kMemoValid) && v.x != kMemoBusy && kk.x == kw[0] && kk.y == kw[1] && kk.z == kw[2] && kk.w == kw[3] &&
                               ((v.y >> 27) & 15u) == (uint32_t)(len - 1);
                if (h) { hit = true; vv = v; hit_slot = b + wy; }
            }
        }
        // A learning batch: in a large batch ONE lane of the wavefront (another one every time) reports its hit -- a sample of one hit in 64 x 8.  Not every hit: the
        // hottest pieces of Zipf-distributed text are hit millions of times a batch, and device-scope atomics on ONE address serialise at about a
        // microsecond each (the first batch under synth100k took 2.4 s instead of 25 ms when every hit of every eighth group was counted).  A counter
        // also stops at kMemoHitsSat (read with an agent-scope atomic load: the line in this XCD's L2 may be stale): such a slot is promoted anyway.
        if (count_hits) {
            if (hit && (!sparse || lane == ((done * 5 + nl0 + nl1) & 63)) && tkz_atomic_load_agent(&T.memo_hits[hit_slot]) < kMemoHitsSat)
                simt::atomic_add(reinterpret_cast<int*>(&T.memo_hits[hit_slot]), 1);
        }
        if (P.stats) { st_look += tkz_popc64(simt::ballot(mine && memo)); st_hit += tkz_popc64(simt::ballot(hit)); }   // (statistics run only: wave-uniform, null in the timed runs)
        const int cnt = hit ? (int)((vv.x >> 29) & 3u) + 1 : 0;
        (void)assign(hit, cnt, si, j, rel);
        if (hit) {                                            // (a memo entry holds <= 4 tokens: they go into the entry's quad)
            uint4 tq; tq.x = vv.x & 0x07FFFFFFu; tq.y = vv.y & 0x07FFFFFFu; tq.z = vv.z & 0x07FFFFFFu; tq.w = vv.w & 0x07FFFFFFu;
            mq0[si * (int64_t)P.mcap + j] = tq;
        }
        (void)simt::ballot(true);
        return mine && !hit;
    };
    // one batch: lane i < n merges the piece of entry i of list c
    auto run_batch = [&](int c, int n) {
        (void)simt::ballot(true);
        int cnt = 0, si = 0, j = 0, rel = 0, e1 = 0, len = 1;
        bool nul = false;
        uint32_t alive = 1;
        uint32_t kw[4] = {0, 0, 0, 0};
        if (lane < n) {
            const uint32_t r16 = s_rec[64 * c + lane], ix = s_idx[64 * c + lane];
            si = (int)(ix >> 10); j = (int)(ix & 1023u); rel = (int)(r16 & 1023u); len = (int)((r16 >> 10) & 15u) + 1;
            key_of(tkz_load16(&mq0[si * (int64_t)P.mcap + j]), len, kw, &nul);
            cnt = tkz_bpe_lane_f<NMAX>(T, kw, len, ids, pr, byte_id, pair_rank, &alive, &e1);
            err |= e1;
        }
        int32_t* dst = assign(lane < n, cnt, si, j, rel);
        if (lane < n) {
            uint32_t t4[4] = {0, 0, 0, 0};
            int i = 0;
            for (uint32_t a = alive; a; a &= a - 1) { const uint32_t t = ids[tkz_ctz32(a)]; if (dst) dst[i] = (int32_t)t; if (i < 4) { if (i == 0) t4[0] = t; else if (i == 1) t4[1] = t; else if (i == 2) t4[2] = t; else t4[3] = t; } ++i; }
            if (cnt <= 4) { uint4 tq; tq.x = t4[0]; tq.y = t4[1]; tq.z = t4[2]; tq.w = t4[3]; mq0[si * (int64_t)P.mcap + j] = tq; }
            // a piece of <= 4 tokens takes a memo slot of its bucket if one is free (an entry is never replaced: a hit stays valid for good)
            if (memo && cnt <= 4 && !e1 && !nul) {
                const uint32_t b = tkz_mulhi(tkz_hash_memo(kw, (uint32_t)len), T.memo_n / kMemoWays) * kMemoWays;
                bool placed = false;
#pragma unroll
                for (int wy = 0; wy < (int)kMemoWays; ++wy) {
                    TkzMemoSlot* slot = &T.memo[b + wy];
                    // (a plain load first: a piece that lost its slot to another one comes back millions of times on repetitive text, and a
                    //  failing compare-and-swap is still an atomic on one hot address)
                    if (!placed && *reinterpret_cast<volatile uint32_t*>(&slot->v[0]) == 0u && simt::atomic_cas(&slot->v[0], 0u, kMemoBusy) == 0u) {
                        // The key and the value are one 16-byte store each, and there is NO fence between or behind them.  The protocol does
                        // not rely on those stores (or a reader's 16-byte loads) being single-copy atomic: EVERY value word carries the valid
                        // tag (bit 31), so a value is accepted only when all four words are the new ones -- any mixture with the old words
                        // (zero, or BUSY in v[0]) is a miss --, and then its length field is the key's; a key seen word by word is a mixture of
                        // the new words and zero words, and a looked-up piece (no zero byte inside its length) can equal such a mixture only
                        // if the missing words lie beyond its length -- where the complete key is zero too: every mixture is a miss or the
                        // right hit, in whatever order the eight words become visible.  (A device-scope release fence here wrote the XCD's dirty L2 lines back for EVERY insertion: the
                        // first batch on an empty memo with many misses -- 369 M short misses of the held-out vocabulary -- spent 566 ms
                        // in this kernel, ten times what it takes with the memo switched off.)
                        uint4 kk; kk.x = kw[0]; kk.y = kw[1]; kk.z = kw[2]; kk.w = kw[3];
                        *reinterpret_cast<uint4*>(&slot->k[0]) = kk;
                        uint4 nv; nv.x = kMemoValid | ((uint32_t)(cnt - 1) << 29) | t4[0]; nv.y = kMemoValid | ((uint32_t)(len - 1) << 27) | t4[1]; nv.z = kMemoValid | t4[2]; nv.w = kMemoValid | t4[3];
                        *reinterpret_cast<uint4*>(&slot->v[0]) = nv;
                        placed = true;
                    }
                }
            }
        }
        (void)simt::ballot(true);
    };
    (void)simt::ballot(true);
    // Entry g of the group's concatenated lists is entry g - pre[q] of sub-tile q.  Lane i always has entry done + i of the lists in flight
    // (entry, quad: two independent loads), requested while the entries before them are looked up in the memo.
    auto request = [&](int g, uint32_t* r16, uint32_t* ix, uint4* kq) {
        *r16 = 0; *ix = 0; kq->x = kq->y = kq->z = kq->w = 0;
        if (g < ntotal) {
            const int q = tkz_find_list<kGroup>(s_pre, g), j = g - (int)s_pre[q];
            const uint32_t ent = tkz_load_nt(&ml0[q * (int64_t)P.mcap + j]);
            *kq = tkz_load16(&mq0[q * (int64_t)P.mcap + j]);
            *r16 = (ent & 1023u) | (((ent >> kMrLenShift) & 15u) << 10);
            *ix = (uint32_t)((q << 10) | j);
        }
    };
    // The pieces that have to be merged wait in TWO lists, by length (round 6): a batch lasts as long as its longest piece -- a merge a step, a round trip a
    // merge --, and with lengths 1..16 in one batch the lanes of the short pieces idled for half of it (lane use 0.4).  A list that reaches 64 entries is
    // merged; what is left of both at the end goes as ONE batch if it fits one (the bench text: ~40 survivors a group), else as one batch a list.
    // (Two lists only for a group with many misses -- kMsTwoListsMin entries, 20 a sub-tile: source text has ~860 a group and gains (k_merge_short 1.66 -> 1.10 ms on
    //  436 MB), the bench text has ~160, ends with ONE batch of ~40 pieces either way and only paid for the second list: 2.39 -> 2.55 ms.  profiles/r06/variants_merge_short2.txt)
    const bool two = ntotal >= kMsTwoListsMin;
    uint32_t a_r16, a_ix;
    uint4 a_kq;
    request(lane, &a_r16, &a_ix, &a_kq);
    while (done < ntotal) {
        const int take = ntotal - done < 64 ? ntotal - done : 64;
        const uint32_t f_r16 = a_r16, f_ix = a_ix;
        const uint4 f_kq = a_kq;
        done += take;
        request(done + lane, &a_r16, &a_ix, &a_kq);            // (the entries behind these: in flight during the memo lookups)
        const bool merge_it = memo_phase(take, f_r16, f_ix, f_kq);
        const int cls = two && ((f_r16 >> 10) & 15u) >= (uint32_t)kMsShortLen ? 1 : 0;       // (the field holds len - 1)
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            const uint64_t m = simt::ballot(merge_it && cls == c);
            const int had = c ? nl1 : nl0;
            const int pos = had + tkz_popc64(m & tkz_lowmask(lane)), tot = had + tkz_popc64(m);
            const bool here = merge_it && cls == c;
            if (here && pos < 64) { s_rec[64 * c + pos] = (uint16_t)f_r16; s_idx[64 * c + pos] = (uint16_t)f_ix; }
            if (tot >= 64) {                                   // the list is full: its batch, then the entries that did not fit
                run_batch(c, 64);
                if (here && pos >= 64) { s_rec[64 * c + pos - 64] = (uint16_t)f_r16; s_idx[64 * c + pos - 64] = (uint16_t)f_ix; }
                if (c) nl1 = tot - 64; else nl0 = tot - 64;
            } else if (c) nl1 = tot; else nl0 = tot;
        }
    }
    if (nl0 + nl1 <= 64 && nl1 > 0) {                           // (one batch for both lists' leftovers)
        (void)simt::ballot(true);
        if (lane < nl1) { s_rec[nl0 + lane] = s_rec[64 + lane]; s_idx[nl0 + lane] = s_idx[64 + lane]; }
        nl0 += nl1; nl1 = 0;
    }
#pragma unroll 1
    for (int c = 0; c < 2; ++c) { const int n = c ? nl1 : nl0; if (n > 0) run_batch(c, n); }
    (void)simt::ballot(true);
    // tokens of every sub-tile of the group: one per piece, plus what the merges added (k_merge_long adds its own later)
    if (lane < kGroup && sub0 + lane < P.nsub) {
        const int tc = my_np + s_extra[lane] + (P.pextra ? P.pextra[sub0 + lane] : 0);
        if (P.tc_atomic) simt::atomic_add(&P.tile_count[sub0 + lane], tc);            // (the long pieces' kernels run beside this one and add theirs: k_probe has zeroed the counts)
        else P.tile_count[sub0 + lane] = tc;
    }
    if (err) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)err);
    if (P.stats && lane == 0 && st_look) { simt::atomic_add64(&P.stats[0], (unsigned long long)st_look); simt::atomic_add64(&P.stats[1], (unsigned long long)st_hit); }
}
// (LDS: 4 x 9.5 KB of merge state + 0.5 KB = 38.5 KB per workgroup, and <= 128 VGPRs: four workgroups = 16 wavefronts per CU)
// A batch whose miss lists or record buffer turned out too small (k_probe / k_list_stats flag it) is run again by the host with larger ones: the
// kernels behind those two have nothing to add to an attempt that is thrown away, and on a fresh encoder's first miss-heavy batch that attempt was
// costing as much as the one that counts.
TKZ_DEV bool tkz_attempt_failed(const EncodeParams& P) { return (P.counters[0] & (kErrMissCap | kErrCapacity)) != 0; }

TKZ_KERNEL_OCC(kMsThreads, 4) void k_merge_short(TkzTables T, EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED uint4 s_wave[kMsThreads / 64][kMsLdsQuads];
    TKZ_SHARED uint16_t s_brank16[256];
    tkz_ms_brank_init(T, s_brank16);
    simt::sync();
    const int64_t grp = tkz_xcd_block(simt::bid(), simt::nblocks()) * (kMsThreads / 64) + simt::wave();
    tkz_merge_short_group(T, P, grp, tkz_ms_lds(s_wave[simt::wave()], s_brank16));
}

// The pieces of 17..256 bytes that have to be merged (longer ones: k_merge_coop below), and the token counts of the giant ones.  One lane per piece with its state in a
// span of an LDS arena sized for it (tkz_bpe_lane_u up to 64 bytes; tkz_bpe_lane_varc / _var beyond: pair ranks [| ids] [| alive bits], preceded by the piece's bytes).
// The long-miss lists of the 64 sub-tiles of a chunk are walked as one list, kLongSeg entries at a time, and every such segment is
// SORTED BY LENGTH (a counting sort over 16 length classes, in LDS) before it is cut into batches of up to 64 lanes: a merge costs a scan
// over the whole piece and a piece of n bytes takes ~n/2 of them, so a 60-byte piece is ten times the work of a 17-byte one -- in list
// order a batch took as long as its longest piece while the other lanes idled (33 % of the lanes active on CJK text).  Spans are padded to
// an odd number of 16-byte quads: pieces of one class have equal spans, and with an even quad stride their 16-byte reads collide.
// A chunk = 64 consecutive sub-tiles (one per lane: their long-miss lists walked as one list).  Until round 5 a chunk was the unit of work a wavefront
// takes; REAL text showed what that costs: its long pieces are not spread evenly -- one file of CJK prose is 64 KiB after 64 KiB of nothing but 100..250-byte
// pieces, ~10 ms of merging for the ONE wavefront that owns such a chunk while the other 3,000 have long finished (k_merge_long 13 ms of a 17 ms step on
// 436 MB of source text, for 0.3 ms of work).  Smaller chunks cut that tail and cost sparse text its lane fill (16 sub-tiles: the headline workload's one
// long miss per KiB makes batches of 16 lanes instead of 64, k_merge_long 1.4 -> 3.7 ms).  So: a chunk whose lists hold more than kLongDense entries is
// split into kLongParts units of work, a sparse one stays whole.
#ifndef TKZ_LONG_PARTS
#define TKZ_LONG_PARTS 4
#endif
#ifndef TKZ_LONG_DENSE
#define TKZ_LONG_DENSE 192
#endif
constexpr int kLongParts = TKZ_LONG_PARTS, kLongDense = TKZ_LONG_DENSE;
static_assert(kLongParts >= 1 && kLongParts <= 64 && 64 % kLongParts == 0, "a part is a whole number of the chunk's 64 sub-tiles");
constexpr int kLongSeg = 384;
constexpr int kFastPiece = 64;           // long misses of up to this many bytes: the fast batches of k_merge_long (tkz_bpe_lane_u)
static_assert(kFastPiece <= kLanePiece && kFastPiece <= kSmallLanePiece, "every caller's lane limit covers the fast batches");
constexpr int kLenClasses = 16;
TKZ_HD int tkz_len_class(int len) {          // 17..1024, monotone
    if (len <= 32) return (len - 17) >> 2;               // 17-20, 21-24, 25-28, 29-32
    if (len <= 64) return 4 + ((len - 33) >> 3);         // 33-40, 41-48, 49-56, 57-64
    if (len <= 128) return len <= 96 ? 8 : 9;
    if (len <= 256) return len <= 192 ? 10 : 11;
    if (len <= 512) return len <= 384 ? 12 : 13;
    return len <= 768 ? 14 : 15;
}
// LDS of the wavefront of k_merge_long
struct LongLds { uint32_t* arena; int arena_dwords; int* pre; int* cls; uint16_t* ord; int32_t* brank; };
constexpr int kLongLdsBytes = kArenaDwords * 4 + 4 * 68 + 4 * kLenClasses + 2 * kLongSeg + 4 * 256;
constexpr int kLongLdsQuads = (kLongLdsBytes + 15) / 16;
TKZ_DEV LongLds tkz_long_lds(uint4* quads) {
    LongLds L;
    uint8_t* b = reinterpret_cast<uint8_t*>(quads);
    L.arena = reinterpret_cast<uint32_t*>(b); b += kArenaDwords * 4; L.arena_dwords = kArenaDwords;
    L.brank = reinterpret_cast<int32_t*>(b); b += 4 * 256;                    // id of every single byte: in LDS, not a gather per byte
    L.pre = reinterpret_cast<int*>(b); b += 4 * 68;
    L.cls = reinterpret_cast<int*>(b); b += 4 * kLenClasses;
    L.ord = reinterpret_cast<uint16_t*>(b);                                   // the segment's list positions, by length class
    return L;
}
TKZ_DEV void tkz_long_brank_init(const TkzTables& T, int32_t* s_brank) {     // by one wavefront
    for (int i = simt::lane(); i < 256; i += 64) s_brank[i] = T.byte_rank[i];
    (void)simt::ballot(true);
}
// The development counters of the long-miss kernels (make DEVPROF=1, TKZ_DEV_ABLATE bit 4): clock ticks of a wavefront by phase, batches, lanes, merges
struct LongProf {
    bool on = false;
    long long t[6] = {0, 0, 0, 0, 0, 0};            // sort / queue | batch formation | bytes | first level | merges | emission and answers
    long long batches = 0, lanes = 0, steps = 0, merges = 0, units = 0, t0 = 0, begin = 0;
};
#ifdef TKZ_DEVPROF
#define TKZ_PF_MARK(i) do { if (PF.on) { const long long pf_now = simt::clock(); PF.t[i] += pf_now - PF.t0; PF.t0 = pf_now; } } while (0)
TKZ_DEV void tkz_long_prof_begin(const EncodeParams& P, LongProf& PF) { PF.on = TKZ_DEV_FLAG(P, 16) && P.devprof; if (PF.on) PF.begin = PF.t0 = simt::clock(); }
TKZ_DEV void tkz_long_prof_end(const EncodeParams& P, LongProf& PF) {
    if (!PF.on) return;
    int mt;
    (void)tkz_wave_scan_sum((int)PF.merges, &mt);
    if (simt::lane() == 0) {
        simt::atomic_add64(&P.devprof[32], 1); simt::atomic_add64(&P.devprof[33], (unsigned long long)(simt::clock() - PF.begin));
        for (int i = 0; i < 6; ++i) simt::atomic_add64(&P.devprof[34 + i], (unsigned long long)PF.t[i]);
        simt::atomic_add64(&P.devprof[40], (unsigned long long)PF.batches); simt::atomic_add64(&P.devprof[41], (unsigned long long)PF.lanes);
        simt::atomic_add64(&P.devprof[42], (unsigned long long)PF.steps); simt::atomic_add64(&P.devprof[43], (unsigned long long)mt);
        simt::atomic_add64(&P.devprof[44], (unsigned long long)PF.units);
    }
}
#else
#define TKZ_PF_MARK(i) do { } while (0)
TKZ_DEV void tkz_long_prof_begin(const EncodeParams&, LongProf&) {}
TKZ_DEV void tkz_long_prof_end(const EncodeParams&, LongProf&) {}
#endif
// ONE BATCH of long misses: the lanes for which `valid` holds (a prefix of the wavefront, lane 0 among them) each bring a piece -- entry j of sub-tile
// sub's long list, len bytes at byte rel of the sub-tile -- in ascending order of tkz_len_class.  Merges a prefix of them, one piece a lane, answers their
// list entries and returns how many it took (>= 1); the caller comes back with the rest.  (COMPACT: ranks below 2^21, i.e. every published vocabulary.)
// (on_limit(n): called as soon as the number of pieces the batch takes is known, before anything is merged -- the queue form requests its next entries there)
template <bool COMPACT, class OnLimit>
TKZ_DEV int tkz_long_batch(const TkzTables& T, const EncodeParams& P, const LongLds& LD, bool valid, int64_t sub, int j, int rel, int len, int lane_piece, int& err, LongProf& PF, OnLimit on_limit) {
    uint32_t* s_arena = LD.arena;
    const int32_t* s_brank = LD.brank;
    const int lane = simt::lane();
    constexpr bool compact = COMPACT;                           // no ids[] array (tkz_bpe_lane_varc): 40 % more pieces per batch
    const int64_t abs = sub * kSub + rel;
    const int64_t lidx = sub * (int64_t)P.mcap + (P.mcap - 1 - j);     // the piece's list entry (and quad)
    bool log_it = false;                             // (a learning batch: this lane's piece goes into the log below)
    uint4 log_tok; log_tok.x = log_tok.y = log_tok.z = log_tok.w = 0;
    int log_cnt = 0;
    const uint32_t* log_bw = s_arena;                // the piece's bytes in the arena
    int limit;
    // ---- a FAST batch (round 6): the next pieces of up to 32 (else: up to 64) bytes, every one in a span of the size the longest of them
    //      needs, merged by tkz_bpe_lane_u (tkz_bpe.h).  The positions are sorted by length class and 32 | 33, 64 | 65 are class edges,
    //      so the pieces of a batch are a prefix of the lanes.
    const int len0 = simt::first_lane(len);          // (lane 0 always holds a piece)
    if (COMPACT && len0 <= kFastPiece) {           // (kFastPiece <= lane_piece: static_assert below)
        const int cut = len0 <= 32 ? 32 : kFastPiece;
        const int nok = tkz_ctz64z(~simt::ballot(valid && len <= cut));
        const int maxlen = (int)~simt::wave_min_u32(~(uint32_t)(lane < nok ? len : 0));
        const int nq = (maxlen + 3) >> 2;            // dwords of bytes = quads of pair keys, the same for every piece of the batch
        const int nbd = (nq + 3) & ~3;
        const int spanq = ((nbd >> 2) + nq) | 1;     // an odd number of quads: the spans' 16-byte reads then sit on distinct banks
        const int fit = (LD.arena_dwords / 4) / spanq; // (>= 1: the arena covers a piece of lane_piece bytes)
        limit = nok < fit ? nok : fit;
        on_limit(limit);
        TKZ_PF_MARK(1);
#ifdef TKZ_DEVPROF
        int pf_mg = 0;
#endif
        if (lane < limit) {
            uint32_t* bw = s_arena + lane * (spanq * 4);
            log_bw = bw;
            {   // nbd dwords of text from the piece's first byte on (past its end: whatever follows; past the end of the text: 0): every load is requested
                // before the first is used (a loop over the dwords waited for one round trip to the text after the other)
                const int64_t a0 = abs & ~(int64_t)3;
                const uint32_t sh = (uint32_t)(abs & 3) * 8u;
                auto fetch = [&](auto NDW) {
                    constexpr int ND = decltype(NDW)::value;
                    uint32_t t[ND + 1];
                    if (a0 + 4 * (int64_t)(ND + 1) <= P.total) {
#pragma unroll
                        for (int w = 0; w <= ND; ++w) t[w] = *reinterpret_cast<const uint32_t*>(P.bytes + a0 + 4 * w);
                    } else {
                        for (int w = 0; w <= ND; ++w) {
                            uint32_t v = 0;
                            for (int b = 0; b < 4; ++b) if (a0 + 4 * w + b < P.total) v |= (uint32_t)P.bytes[a0 + 4 * w + b] << (8 * b);
                            t[w] = v;
                        }
                    }
                    uint4* bw4 = reinterpret_cast<uint4*>(bw);
#pragma unroll
                    for (int g = 0; g < ND / 4; ++g)
                        if (4 * g < nbd) {                  // (wave-uniform)
                            uint4 v;
                            v.x = simt::alignbit(t[4 * g + 1], t[4 * g], sh); v.y = simt::alignbit(t[4 * g + 2], t[4 * g + 1], sh);
                            v.z = simt::alignbit(t[4 * g + 3], t[4 * g + 2], sh); v.w = simt::alignbit(t[4 * g + 4], t[4 * g + 3], sh);
                            bw4[g] = v;
                        }
                };
                if (cut == 32) fetch(std::integral_constant<int, 8>{}); else fetch(std::integral_constant<int, 16>{});
            }
            uint32_t* pr = bw + nbd;
            uint32_t q4[4];
            int e1 = 0, cnt;
            TKZ_PF_MARK(2);
            tkz_bpe_lane_u_init(T, bw, len, nq, pr);
            TKZ_PF_MARK(3);
            if (cut == 32) {
                uint32_t alive;
                cnt = tkz_bpe_lane_u_merge<uint32_t>(T, bw, len, nq, pr, s_brank, &alive);
                TKZ_PF_MARK(4);
                tkz_bpe_lane_u_emit<uint32_t>(bw, len, nq, pr, s_brank, alive, cnt > 4, P.tmp + abs, q4, &e1);
            } else {
                uint64_t alive;
                cnt = tkz_bpe_lane_u_merge<uint64_t>(T, bw, len, nq, pr, s_brank, &alive);
                TKZ_PF_MARK(4);
                tkz_bpe_lane_u_emit<uint64_t>(bw, len, nq, pr, s_brank, alive, cnt > 4, P.tmp + abs, q4, &e1);
            }
            err |= e1;
#ifdef TKZ_DEVPROF
            pf_mg = len - cnt;
#endif
            // (five tokens and more wait in tmp at the piece's position; up to four go into the entry's quad, where k_place finds them in the
            //  round trip that fetches the answers)
            if (cnt <= 4) {
                uint4 tq; tq.x = q4[0]; tq.y = q4[1]; tq.z = q4[2]; tq.w = q4[3];
                P.mquad[lidx] = tq;
                P.mlist[lidx] = tkz_result_inline(cnt);
                if (P.long_log && len <= kLongLogMaxLen && !e1 && (!P.long_log_sparse || ((sub >> 6) & 7) == 0)) { log_it = true; log_tok = tq; log_cnt = cnt; }
            } else P.mlist[lidx] = tkz_result_entry(false, cnt, rel);
            if (cnt > 1) simt::atomic_add(&P.tile_count[sub], cnt - 1);
        }
        TKZ_PF_MARK(5);
#ifdef TKZ_DEVPROF
        if (PF.on) {
            PF.merges += pf_mg;                                        // (per lane; summed over the wave at the end)
            const int mx = (int)~simt::wave_min_u32(~(uint32_t)pf_mg);
            if (lane == 0) { ++PF.batches; PF.lanes += limit; PF.steps += mx; }
        }
#endif
    } else {
    // ---- the general batch: pieces of more than kFastPiece bytes (and every piece of a vocabulary with ranks of 2^21 and more), each in a span of
    //      its own size, up to the first whose state no longer fits the arena
    const int nbw = (len + 3) >> 2;
    const bool mine = valid && len <= lane_piece;    // (a longer piece keeps its entry as it is: k_merge_coop answers it -- k_probe has flagged its sub-tile)
    int need = 0;
    if (mine) {
        need = ((nbw + 3) & ~3) + (compact ? tkz_bpe_varc_dwords(len) : tkz_bpe_var_dwords(len));
        if (!((need >> 2) & 1)) need += 4;           // an odd number of quads: equal spans then sit on distinct banks
    }
    int btot;
    const int aoff = tkz_wave_scan_sum(need, &btot);
    const uint64_t bad = simt::ballot(mine && aoff + need > LD.arena_dwords);
    limit = bad ? tkz_ctz64(bad) : 64;               // lanes at or beyond `limit` wait for the next batch (limit >= 1: one piece always fits)
    on_limit(limit);
    if (mine && lane < limit) {
        uint32_t* bw = &s_arena[aoff];                   // (len + 3) / 4 dwords of bytes, then the merge state
        log_bw = bw;
        {
            const int64_t a0 = abs & ~(int64_t)3;
            const uint32_t sh = (uint32_t)(abs & 3) * 8u;
            uint32_t prev = 0;
            if (a0 + 4 <= P.total) prev = *reinterpret_cast<const uint32_t*>(P.bytes + a0);
            else for (int b = 0; b < 4; ++b) if (a0 + b < P.total) prev |= (uint32_t)P.bytes[a0 + b] << (8 * b);
            for (int w = 0; w < nbw; ++w) {
                const int64_t p = a0 + 4 * (w + 1);
                uint32_t nx = 0;
                if (p + 4 <= P.total) nx = *reinterpret_cast<const uint32_t*>(P.bytes + p);
                else for (int b = 0; b < 4; ++b) if (p + b < P.total) nx |= (uint32_t)P.bytes[p + b] << (8 * b);
                bw[w] = simt::alignbit(nx, prev, sh);
                prev = nx;
            }
        }
        const uint8_t* pbytes = reinterpret_cast<const uint8_t*>(bw);
        uint32_t* st = bw + ((nbw + 3) & ~3);             // (the state arrays are read 16 bytes at a time)
        auto at = [&](int i) -> uint32_t { return pbytes[i]; };
        int e1 = 0, cnt;
        if constexpr (COMPACT) {
            cnt = tkz_bpe_lane_varc(T, at, len, st, &e1, s_brank); tkz_bpe_varc_emit(at, st, len, s_brank, P.tmp + abs);
        } else {
            cnt = T.max_rank <= kVarPackedMaxRank ? tkz_bpe_lane_var<true>(T, at, len, st, &e1, s_brank) : tkz_bpe_lane_var<false>(T, at, len, st, &e1, s_brank);
            tkz_bpe_var_emit(st, len, P.tmp + abs);
        }
        err |= e1;
        // (the tokens are in tmp at the piece's position; up to four also go into the entry's quad, where k_place finds them
        //  in the round trip that fetches the answers)
        if (cnt <= 4) {
            const int32_t* tk = P.tmp + abs;
            uint4 tq; tq.x = (uint32_t)tk[0]; tq.y = cnt > 1 ? (uint32_t)tk[1] : 0u; tq.z = cnt > 2 ? (uint32_t)tk[2] : 0u; tq.w = cnt > 3 ? (uint32_t)tk[3] : 0u;
            P.mquad[lidx] = tq;
            P.mlist[lidx] = tkz_result_inline(cnt);
            if (P.long_log && len <= kLongLogMaxLen && !e1 && (!P.long_log_sparse || ((sub >> 6) & 7) == 0)) { log_it = true; log_tok = tq; log_cnt = cnt; }
        } else P.mlist[lidx] = tkz_result_entry(false, cnt, rel);
        if (cnt > 1) simt::atomic_add(&P.tile_count[sub], cnt - 1);
    }
    }
    if (P.long_log) {                                // (wave-uniform; a learning batch only) one atomic for the wavefront's records
        int ltot;
        const int lpre = tkz_wave_scan_sum(log_it ? 1 : 0, &ltot);
        unsigned long long lbase = 0;
        if (lane == 0 && ltot) lbase = simt::atomic_add64(P.long_log_count, (unsigned long long)ltot);
        lbase = ((unsigned long long)simt::shflu((uint32_t)(lbase >> 32), 0) << 32) | simt::shflu((uint32_t)lbase, 0);
        if (log_it && lbase + (unsigned long long)lpre < (unsigned long long)P.long_log_cap) {
            uint32_t* rec = P.long_log + (lbase + (unsigned long long)lpre) * kLongLogDwords;
            for (int w = 0; w < 7; ++w) {                    // (the piece's bytes, still in the arena)
                const int nbz = len - 4 * w;
                rec[w] = nbz <= 0 ? 0u : (nbz >= 4 ? log_bw[w] : (log_bw[w] & ((1u << (8 * nbz)) - 1u)));
            }
            rec[7] = (uint32_t)len | ((uint32_t)log_cnt << 8);
            rec[8] = log_tok.x; rec[9] = log_tok.y; rec[10] = log_tok.z; rec[11] = log_tok.w;
        }
    }
    (void)simt::ballot(true);
    return limit;
}
// chunks c0, c0 + cstep, ... of 64 sub-tiles each, by one wavefront
// (COMPACT: ranks below 2^21, i.e. every published vocabulary -- no ids[] array)
template <bool COMPACT, int PARTS = kLongParts, int DENSE = kLongDense>
// (lane_piece: pieces of up to this many bytes are merged here -- kLanePiece on the batch path, which has k_merge_coop for the longer ones; the
//  single-launch kernel has no such kernel and takes kSmallLanePiece)
TKZ_DEV void tkz_merge_long_chunks(const TkzTables& T, const EncodeParams& P, int64_t c0, int64_t cstep, const LongLds& LD, const int lane_piece = kLanePiece) {
    int* s_pre = LD.pre;
    int* s_cls = LD.cls;
    uint16_t* s_ord = LD.ord;
    const int lane = simt::lane();
    int err = 0;
    LongProf PF;
    tkz_long_prof_begin(P, PF);
    // A unit of work = a chunk of 64 sub-tiles, or -- when the chunk's lists hold more than kLongDense entries -- one of its kLongParts parts
    // (the parts of a sparse chunk other than part 0 have nothing to do).
    // (part-major: units 0 .. nchunks-1 are part 0 of every chunk.  Chunk-major -- unit u = part u % 4 of chunk u / 4 -- put the only units that have
    //  work in sparse text, the parts 0, on workgroups 0, 4, 8 ...: workgroups go round the 8 XCDs, so two of them did everything, 1.4 -> 5 ms.)
    const int64_t nchunks = (P.nsub + 63) / 64;
    for (int64_t u = c0; u < nchunks * PARTS; u += cstep) {
        const int64_t c = u % nchunks;
        const int part = (int)(u / nchunks);
        const int64_t t = c * 64 + lane;
        int my_nl = 0;
        if (t < P.nsub) {
            const uint32_t mc = P.mcount[t];
            my_nl = (int)(mc >> 16);
            if ((int)(mc & 0xFFFFu) + my_nl > P.mcap) my_nl = 0;               // (cut list: reported by k_probe, the batch is redone)
            // the giant piece of the sub-tile (its last piece): merged by k_giant_merge, its token count is added here (by the chunk's part 0)
            if (part == 0 && (P.heavy_flag[t] & 2u)) { const int g = P.giant_cnt[t]; if (g > 1) simt::atomic_add(&P.tile_count[t], g - 1); }   // (< 0: no pool, the call is retried)
        }
        int ntotal;
        (void)tkz_wave_scan_sum(my_nl, &ntotal);
        if (ntotal > DENSE) { if (lane / (64 / PARTS) != part) my_nl = 0; }      // dense: this unit's share of the sub-tiles
        else if (part != 0) continue;                                                    // sparse: the whole chunk is part 0's
        const int pre = tkz_wave_scan_sum(my_nl, &ntotal);
        if (!ntotal) continue;
        (void)simt::ballot(true);
        s_pre[lane] = pre;
        if (lane == 0) s_pre[64] = ntotal;
        (void)simt::ballot(true);
        // position g of the 64 lists taken as one: entry g - pre[q] (from the back) of sub-tile q
        auto entry_at = [&](int g, int* q, int* j) -> uint32_t {
            *q = tkz_find_list<64>(s_pre, g); *j = g - s_pre[*q];
            return tkz_load_nt(&P.mlist[(c * 64 + *q) * (int64_t)P.mcap + (P.mcap - 1 - *j)]);
        };
#ifdef TKZ_DEVPROF
        if (PF.on) { ++PF.units; PF.t0 = simt::clock(); }
#endif
        for (int seg0 = 0; seg0 < ntotal; seg0 += kLongSeg) {
            const int nseg = ntotal - seg0 < kLongSeg ? ntotal - seg0 : kLongSeg;
            // ---- counting sort of the segment's positions by length class ----
            if (lane < kLenClasses) s_cls[lane] = 0;
            (void)simt::ballot(true);
            uint32_t mycls = 0;                                  // the classes of this lane's (up to 8) entries, 4 bits each
#pragma unroll
            for (int i = 0; i < kLongSeg / 64; ++i) {
                const int e = lane + 64 * i;
                if (e < nseg) {
                    int q, j;
                    const uint32_t ent = entry_at(seg0 + e, &q, &j);
                    const int cl = tkz_len_class((int)((ent >> kMrLenShift) & 1023u) + 1);
                    mycls |= (uint32_t)cl << (4 * i);
                    simt::atomic_add(&s_cls[cl], 1);
                }
            }
            (void)simt::ballot(true);
            {
                int tot;
                const int v = lane < kLenClasses ? s_cls[lane] : 0;
                const int x = tkz_wave_scan_sum(v, &tot);
                (void)simt::ballot(true);
                if (lane < kLenClasses) s_cls[lane] = x;
            }
            (void)simt::ballot(true);
#pragma unroll
            for (int i = 0; i < kLongSeg / 64; ++i) {
                const int e = lane + 64 * i;
                if (e < nseg) s_ord[simt::atomic_add(&s_cls[(mycls >> (4 * i)) & 15u], 1)] = (uint16_t)e;
            }
            (void)simt::ballot(true);
            TKZ_PF_MARK(0);
            // ---- batches: the next sorted positions, one per lane, up to the first whose state no longer fits the arena ----
            for (int done = 0; done < nseg;) {
                const bool valid = done + lane < nseg;
                int q = 0, j = 0, len = 1, rel = 0;
                if (valid) {
                    const uint32_t ent = entry_at(seg0 + (int)s_ord[done + lane], &q, &j);
                    rel = (int)(ent & 1023u); len = (int)((ent >> kMrLenShift) & 1023u) + 1;
                }
                const int64_t sub = c * 64 + q;
                const int limit = tkz_long_batch<COMPACT>(T, P, LD, valid, sub, j, rel, len, lane_piece, err, PF, [](int) {});
                (void)simt::ballot(true);
                done += limit < nseg - done ? limit : nseg - done;
            }
        }
    }
    if (err) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)err);
    tkz_long_prof_end(P, PF);
}
// (a kernel of its own per form so that the general forms' registers stay out of the compact one)
// (LATENCY: a small batch -- TKZ_OPT_LATENCY_BYTES --, where the call waits for the slowest wavefront: a unit of work is 4 sub-tiles whatever their lists hold
//  -- ~20 entries of ordinary text, one batch of lanes -- instead of up to 64 sub-tiles and five batches one after the other: 93 us of a 1 MB call)
constexpr int kLongPartsLatency = 16;
template <bool COMPACT, bool LATENCY>
TKZ_KERNEL_OCC(64, 4) void k_merge_long(TkzTables T, EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED uint4 s_lds[kLongLdsQuads];
    const LongLds LD = tkz_long_lds(s_lds);
    tkz_long_brank_init(T, LD.brank);
    if (LATENCY) tkz_merge_long_chunks<COMPACT, kLongPartsLatency, -1>(T, P, simt::bid(), simt::nblocks(), LD, P.lane_piece);
    else tkz_merge_long_chunks<COMPACT>(T, P, simt::bid(), simt::nblocks(), LD, P.lane_piece);
}

// ---- The long misses of a LARGE batch, binned by length class ACROSS the batch (round 6) ----------------------------------------------------------
// The chunk form above forms its batches inside one unit of work (a chunk of 64 sub-tiles or a quarter of it): on mixed text that is ~290 entries cut
// by length class into seven batches of 41 lanes on average (the development counters: lanes/batch 40.7 of 64), and a file of CJK prose in real text is
// a few chunks that keep their wavefronts for milliseconds.  Here the entries of the WHOLE batch are binned first:
//   k_long_count    per chunk of 64 sub-tiles, the number of its long misses in each of the 16 length classes -> lq_cnt[class * nchunks + chunk]
//   (scan)          exclusive scan of lq_cnt in that order = class-major: all entries of class 0, chunk after chunk, then class 1, ...
//   k_long_scatter  every entry to its place in the queue:  lq[] = sub-tile << 30 | index in its long list << 20 | (len - 1) << 10 | byte in the sub-tile
//   k_merge_long_q  a wavefront takes RANGES of kLqRange queue entries (kLqRangeLong beyond kFastPiece bytes) -- the longest classes first -- and merges them (tkz_long_batch):
//                   full batches of ONE length class, whichever chunks the pieces come from; the work is dealt out 256 pieces at a time, not by chunk.
// No atomics on shared counters (the two passes over the lists cost ~0.1 ms a GB of mixed text); a small batch (TKZ_OPT_LATENCY_BYTES) and the
// single-launch kernel keep the chunk form: three more launches are 15 us.
constexpr int kLqRange = 256, kLqRangeLong = 16;
#ifndef TKZ_QARENA_DWORDS
#define TKZ_QARENA_DWORDS 2304
#endif
constexpr int kQArenaDwords = TKZ_QARENA_DWORDS;   // the arena of k_merge_long_q
static_assert(kQArenaDwords % 4 == 0 && kQArenaDwords >= (kLanePiece + 3) / 4 + 8 + 2 * kLanePiece + 16, "a piece of kLanePiece bytes fits the arena in every form");
#ifdef TKZ_HOSTEMU
constexpr int kLongQGrid = 8;           // (the CPU emulator pays for every idle workgroup)
#else
constexpr int kLongQGrid = 8192;        // wavefronts of k_merge_long_q (13 fit a CU: 3,328 run at once; the ranges beyond are taken round-robin)
#endif
// the entries of chunk c's long lists that a lane merges (<= lane_piece bytes), lane = sub-tile: f(list index j, entry)
template <class F>
TKZ_DEV void tkz_long_walk(const EncodeParams& P, int64_t c, int lane_piece, F f) {
    const int64_t t = c * 64 + simt::lane();
    int nl = 0;
    if (t < P.nsub) {
        const uint32_t mc = P.mcount[t];
        nl = (int)(mc >> 16);
        if ((int)(mc & 0xFFFFu) + nl > P.mcap) nl = 0;                         // (cut list: reported by k_list_stats, the batch is redone)
    }
    // (four entries a step, their loads requested together: a list is read from its end down, one entry a round trip otherwise)
    const uint32_t* top = &P.mlist[t * (int64_t)P.mcap + (P.mcap - 1)];
    for (int j = 0; j < nl; j += 4) {
        uint32_t ent[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ent[k] = j + k < nl ? tkz_load_nt(top - (j + k)) : 0xFFFFFFFFu;        // (0xFFFFFFFF: 1024 bytes, no lane's piece)
#pragma unroll
        // (an entry that is answered already is k_merge_coop's, which may run BEFORE or beside this walk -- launch_encode --: its length was above lane_piece)
        for (int k = 0; k < 4; ++k) if (j + k < nl && !(ent[k] & kMrDone) && (int)((ent[k] >> kMrLenShift) & 1023u) + 1 <= lane_piece) f(t, j + k, ent[k]);
    }
}
TKZ_KERNEL(256) void k_long_count(EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED int s_hist[4][kLenClasses];
    const int lane = simt::lane(), wave = simt::wave();
    const int64_t nchunks = (P.nsub + 63) / 64;
    for (int64_t c = simt::bid() * 4 + wave; c < nchunks; c += simt::nblocks() * 4) {
        if (lane < kLenClasses) s_hist[wave][lane] = 0;
        (void)simt::ballot(true);
        tkz_long_walk(P, c, P.lane_piece, [&](int64_t, int, uint32_t ent) { simt::atomic_add(&s_hist[wave][tkz_len_class((int)((ent >> kMrLenShift) & 1023u) + 1)], 1); });
        (void)simt::ballot(true);
        if (lane < kLenClasses) P.lq_cnt[(int64_t)lane * nchunks + c] = s_hist[wave][lane];
        (void)simt::ballot(true);
    }
}
TKZ_KERNEL(256) void k_long_scatter(EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED int s_hist[4][kLenClasses];
    const int lane = simt::lane(), wave = simt::wave();
    const int64_t nchunks = (P.nsub + 63) / 64;
    for (int64_t c = simt::bid() * 4 + wave; c < nchunks; c += simt::nblocks() * 4) {
        if (lane < kLenClasses) s_hist[wave][lane] = 0;
        (void)simt::ballot(true);
        // the giant piece of a sub-tile (its last piece): merged by k_giant_merge before this kernel, its token count joins the sub-tile's here
        { const int64_t t = c * 64 + lane; if (t < P.nsub && (P.heavy_flag[t] & 2u)) { const int g = P.giant_cnt[t]; if (g > 1) simt::atomic_add(&P.tile_count[t], g - 1); } }   // (< 0: no pool, the call is retried)
        tkz_long_walk(P, c, P.lane_piece, [&](int64_t t, int j, uint32_t ent) {
            const int cl = tkz_len_class((int)((ent >> kMrLenShift) & 1023u) + 1);
            const int64_t at = P.lq_base[(int64_t)cl * nchunks + c] + simt::atomic_add(&s_hist[wave][cl], 1);
            if (at < P.lq_cap) P.lq[at] = ((uint64_t)t << 30) | ((uint64_t)j << 20) | (uint64_t)(ent & 0xFFFFFu);      // (ent: relpos | (len - 1) << 10)
        });
        (void)simt::ballot(true);
    }
}
template <bool COMPACT>
TKZ_KERNEL_OCC(64, 4) void k_merge_long_q(TkzTables T, EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    // LDS: the arena and the byte ids, 10 KB together -- sixteen wavefronts a CU (the chunk form's 12.4 KB: thirteen) --; 2,304 dwords are 64 spans of the
    // nine quads a piece of up to 28 bytes takes (52 of the eleven of 29..32 bytes)
    TKZ_SHARED uint4 s_lds[(kQArenaDwords + 256) / 4];
    LongLds LD;
    LD.arena = reinterpret_cast<uint32_t*>(s_lds); LD.arena_dwords = kQArenaDwords; LD.brank = reinterpret_cast<int32_t*>(s_lds) + kQArenaDwords;
    LD.pre = nullptr; LD.cls = nullptr; LD.ord = nullptr;
    tkz_long_brank_init(T, LD.brank);
    const int lane = simt::lane();
    int err = 0;
    LongProf PF;
    tkz_long_prof_begin(P, PF);
    const int64_t total = *P.lq_total < P.lq_cap ? *P.lq_total : P.lq_cap;
    // Ranges: kLqRange entries of the classes the fast batches take (up to kFastPiece bytes: four or five batches of one class), kLqRangeLong of the longer
    // ones (a lane takes ~n^2 steps for those and a batch holds ~15 of them: in ranges of 256 real text's 100..128-byte pieces kept a few wavefronts for
    // 3.5 ms).  From the END of the queue -- the longest classes first --, so that what runs last is short.
    const int64_t nchunks = (P.nsub + 63) / 64;
    int64_t nfast = P.lq_base[(int64_t)tkz_len_class(kFastPiece + 1) * nchunks];
    if (nfast > total) nfast = total;
    // (a SHORT queue -- real text: half a million long misses in 268 MB -- is dealt out in smaller ranges, down to one batch of 64: with 256 a range there
    //  are fewer ranges than wavefronts and the kernel lasts as long as one wavefront's four or five batches one after the other)
    int64_t rfast = kLqRange;
    while (rfast > 64 && (nfast + rfast - 1) / rfast < simt::nblocks()) rfast >>= 1;
    const int64_t nrf = (nfast + rfast - 1) / rfast, nrl = (total - nfast + kLqRangeLong - 1) / kLqRangeLong;
    auto range_of = [&](int64_t r, int64_t* lo, int64_t* hi) {
        if (r >= nrf) { *lo = nfast + (r - nrf) * kLqRangeLong; *hi = *lo + kLqRangeLong < total ? *lo + kLqRangeLong : total; }
        else { *lo = r * rfast; *hi = *lo + rfast < nfast ? *lo + rfast : nfast; }
    };
    auto entry_of = [&](int64_t pos, int64_t hi) -> uint64_t { return pos < hi ? P.lq[pos] : 0ull; };
    int64_t rg = nrf + nrl - 1 - simt::bid();
    int64_t lo = 0, hi = 0;
    if (rg >= 0) range_of(rg, &lo, &hi);
    uint64_t ent = rg >= 0 ? entry_of(lo + lane, hi) : 0ull;        // this batch's entry; the next batch's is requested as soon as this one's size is known
    while (rg >= 0) {
#ifdef TKZ_DEVPROF
        if (PF.on) { ++PF.units; PF.t0 = simt::clock(); }
#endif
        int64_t nlo = 0, nhi = 0;
        const int64_t nrg = rg - simt::nblocks();
        if (nrg >= 0) range_of(nrg, &nlo, &nhi);
        for (int64_t at = lo; at < hi;) {
            const bool valid = at + lane < hi;
            const int64_t sub = valid ? (int64_t)(ent >> 30) : 0;
            const int j = valid ? (int)((ent >> 20) & 1023u) : 0, len = valid ? (int)((ent >> 10) & 1023u) + 1 : 1, rel = valid ? (int)(ent & 1023u) : 0;
            uint64_t ent_next = 0;
            TKZ_PF_MARK(0);
            const int limit = tkz_long_batch<COMPACT>(T, P, LD, valid, sub, j, rel, len, P.lane_piece, err, PF, [&](int n) {
                const int64_t nx = at + n;                                  // the next batch: the rest of this range, else the wavefront's next range
                ent_next = nx < hi ? entry_of(nx + lane, hi) : (nrg >= 0 ? entry_of(nlo + lane, nhi) : 0ull);
            });
            at += limit;
            ent = ent_next;
        }
        rg = nrg; lo = nlo; hi = nhi;
    }
    if (err) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)err);
    tkz_long_prof_end(P, PF);
}

// The missed pieces of kLanePiece + 1 .. kArenaPiece bytes: ONE WAVEFRONT per piece, with the giant pieces' merger (tkz_bpe_long_tail: batches of proposals with
// local bounds + rounds for chains of equal pairs -- its workgroup is this kernel's single wavefront, 32 slots a lane).  A lane of k_merge_long takes ~n^2 steps
// for a piece of n bytes and the kernel lasts as long as its slowest lane: one 1000-byte run of one letter held it for 3.4 ms, 40 times what the rest of a
// 1 GB batch needed; the wavefront takes ~30 batches whatever the length.  (Below kLanePiece the lanes win: 64 pieces a wavefront instead of one.)
// k_merge_long leaves such an entry as k_probe wrote it; k_probe has flagged the sub-tile (bit 2 of heavy_flag), k_list_stats queues the flagged sub-tiles' entries.
constexpr int kCoopScratch = (4 * kTailSubs + 8) * 64 + 16 + 16;
constexpr int kCoopLdsBytes = 2 * 4 * kArenaPiece + 4 * (kArenaPiece / 32) + kCoopScratch;
#ifdef TKZ_HOSTEMU
constexpr int kCoopGrid = 4;          // (the CPU emulator pays for every idle workgroup)
#else
constexpr int kCoopGrid = 2048;
#endif
constexpr int kCoopLdsQuads = (kCoopLdsBytes + 15) / 16;
// one wavefront, number `bid` of `nblocks`, over the queue (the body of k_merge_coop; k_merge_latency runs it beside the other two merge stages)
TKZ_DEV void tkz_merge_coop_queue(const TkzTables& T, const EncodeParams& P, uint4* s_lds, unsigned long long bid, unsigned long long nblocks) {
    int32_t* ids = reinterpret_cast<int32_t*>(s_lds);
    int32_t* pr = ids + kArenaPiece;
    uint32_t* alive = reinterpret_cast<uint32_t*>(pr + kArenaPiece);
    uint8_t* scratch = reinterpret_cast<uint8_t*>(alive + kArenaPiece / 32);
    const int lane = simt::lane();
    int err = 0;
    // The pieces wait in a queue (k_list_stats: sub-tile << 10 | index in its long list); a wavefront takes the next one off a ticket counter until
    // there is none.  (Until round 5 a wavefront owned chunks of 64 sub-tiles and merged the pieces of a chunk one after the other: a file of CJK prose --
    // 64 KiB of nothing but such pieces -- kept ONE wavefront busy for milliseconds.)
    // (a workgroup's FIRST piece is the one with its own number -- no atomic at all when the queue is shorter than the grid, in particular when it is
    //  empty: two thousand device-scope atomics on one address are not free --, the later ones come off the ticket counter)
    const unsigned long long count = *P.coop_count < (unsigned long long)P.coop_cap ? *P.coop_count : (unsigned long long)P.coop_cap;
    bool first = true;
    for (;;) {
        unsigned long long tk = bid;
        if (!first) {
            if (lane == 0) tk = nblocks + simt::atomic_add64(P.coop_ticket, 1ull);
            tk = ((unsigned long long)simt::shflu((uint32_t)(tk >> 32), 0) << 32) | simt::shflu((uint32_t)tk, 0);
        }
        first = false;
        if (tk >= count) break;
        const uint64_t qe = P.coop_q[tk];
        const int64_t sub = (int64_t)(qe >> 10);
        const int jj = (int)(qe & 1023u);
        const uint32_t ee = P.mlist[sub * (int64_t)P.mcap + (P.mcap - 1 - jj)];
        if (ee & kMrDone) continue;                             // (cannot happen: an entry is queued once)
        const int rel = (int)(ee & 1023u), n = (int)((ee >> kMrLenShift) & 1023u) + 1;
        const int64_t abs = sub * kSub + rel;
        const uint8_t* gb = P.bytes + abs;
        for (int k = lane; k < n; k += 64) {
            const uint32_t b = gb[k];
            ids[k] = T.byte_rank[b];
            pr[k] = (k + 1 < n) ? T.bytepair_rank[(b << 8) | gb[k + 1]] : TKZ_RANK_NONE;
        }
        simt::sync();
        tkz_bpe_long_tail<true>(T, n, ids, pr, alive, scratch, nullptr);
        const int cnt = tkz_bpe_long_tail_emit<true>(n, ids, pr, alive, P.tmp + abs, &err);
        if (lane == 0) {
            P.mlist[sub * (int64_t)P.mcap + (P.mcap - 1 - jj)] = tkz_result_entry(false, cnt, rel);      // (the tokens are in tmp at the piece's position)
            if (cnt > 1) simt::atomic_add(&P.tile_count[sub], cnt - 1);
        }
        simt::sync();
    }
    if (err) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)err);
}
TKZ_KERNEL(64) void k_merge_coop(TkzTables T, EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED uint4 s_lds[kCoopLdsQuads];
    tkz_merge_coop_queue(T, P, s_lds, (unsigned long long)simt::bid(), (unsigned long long)simt::nblocks());
}

// A SMALL batch (TKZ_OPT_LATENCY_BYTES) waits for the slowest wavefront of every kernel and for every dispatch: its two large merge stages -- the short misses, the
// long ones in the chunk form -- touch different list entries and add their tokens to the sub-tiles' counts with atomics (EncodeParams::tc_atomic: k_probe has
// zeroed them), so they run as ONE launch: workgroups [0, nb_short) are k_merge_short's, block for block (four groups of 16 sub-tiles each, the same XCD order:
// the piece memo is shared through an XCD's L2 -- a wavefront a workgroup, spread over the eight XCDs, cut a 160 KB batch's memo hits from most lookups to 6 %),
// workgroups [nb_short, nb_short + nb_long) take the units of the long lists, a wavefront each.  A 1 MB call: k_merge_short 33 us + k_merge_long 45 us one after
// the other, each behind its own dispatch; 240 -> 210 us a call.
constexpr int kMergeLatWaveQuads = kMsLdsQuads > kLongLdsQuads ? kMsLdsQuads : kLongLdsQuads;
template <bool COMPACT>
TKZ_KERNEL_OCC(kMsThreads, 3) void k_merge_latency(TkzTables T, EncodeParams P, int nb_short, int nb_long) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED uint4 s_wave[kMsThreads / 64][kMergeLatWaveQuads];
    TKZ_SHARED uint16_t s_brank16[256];
    const int b = (int)simt::bid();
    if (b < nb_short) {                                     // (wave-uniform, block-uniform: the barrier below is reached by the whole workgroup or not at all)
        tkz_ms_brank_init(T, s_brank16);
        simt::sync();
        const int64_t grp = tkz_xcd_block(b, nb_short) * (kMsThreads / 64) + simt::wave();
        tkz_merge_short_group(T, P, grp, tkz_ms_lds(s_wave[simt::wave()], s_brank16));
    } else {
        const LongLds LD = tkz_long_lds(s_wave[simt::wave()]);
        tkz_long_brank_init(T, LD.brank);
        tkz_merge_long_chunks<COMPACT, kLongPartsLatency, -1>(T, P, (int64_t)(b - nb_short) * (kMsThreads / 64) + simt::wave(), (int64_t)nb_long * (kMsThreads / 64), LD, P.lane_piece);
    }
}

// ids at their final position: count per record -> prefix -> the sub-tile's ids staged in LDS -> stored as whole 16-byte quads (one
// store instruction = 16 full lines; per-lane 4-byte stores at variable offsets wrote 1.8x the bytes); the token index (inside the
// sub-tile) of every marked piece
#ifndef TKZ_PLACE_OCC
#define TKZ_PLACE_OCC 7
#endif
// Answers (and quads) of a sub-tile's miss lists that k_place keeps in LDS: SLOTS = 64 (one per lane: the short list from slot 0 up, the long one from
// slot 63 down; a sub-tile of the bench corpus averages 14 short misses and one long one) or 128 (two per lane, the second half only for
// sub-tiles that keep more than 64).  The 128-slot form costs every sub-tile ~5 % (its instructions, its LDS) and saves a miss-heavy batch
// 40 % of the kernel (a vocabulary that has not seen the text: 75 misses per sub-tile, most of them above 64, which the 64-slot form hands
// to its general path): the host launches it when more than a fifth of the sub-tiles of the workspace's previous batch held more than 64
// entries (k_probe counts them on every 64th sub-tile; EncodeParams::place128).
constexpr int kPlacePer = 4;                               // consecutive sub-tiles per wavefront of k_place
constexpr int kPlaceBig = 8;                               // the general path: token runs longer than this are copied by the whole wavefront, not staged
#ifndef TKZ_PLACE_FAST_BIG
#define TKZ_PLACE_FAST_BIG 64
#endif
constexpr int kPlaceFastBig = TKZ_PLACE_FAST_BIG;                          // the fast path takes sub-tiles whose longest token run is at most this (a lane copies its piece's tokens into the stage)
constexpr int kStageMin = 64 * kPlaceBig + 16;             // staged ids: after a flush, the <= 64 x kPlaceBig ids of one batch of records + the alignment shift always fit
// ... and what k_place's wavefronts keep: the fast path stages the ids of 256 records at once and hands a chunk that does not fit to the general path (64 records at a
// time, four times the scans) -- text of two tokens a piece (source text under the gpt2 table: 1.75) did that for every fifth chunk with 528 ids of room
#ifndef TKZ_PLACE_STAGE
#define TKZ_PLACE_STAGE 784
#endif
constexpr int kStageBatch = TKZ_PLACE_STAGE;
static_assert(kStageBatch >= kStageMin && kStageBatch % 16 == 0, "whole quads, and room for one batch of the general path");
// LDS of one wavefront of k_place: the staged ids, the first answers of the sub-tile's short-miss list and of the long one, and the quads
// of those entries (the tokens of pieces of <= 4)
// ... the token position of every kept entry's piece (the fast path: the lane that owns the record tells the lane that owns the entry)
template <int SLOTS, int STAGE> constexpr int kPlaceLdsQuadsT = STAGE / 4 + SLOTS + SLOTS / 4 + 1 + SLOTS / 4;
constexpr int kPlaceLdsQuads = kPlaceLdsQuadsT<64, kStageMin>;       // (the single-launch kernel always uses the 64-slot form, with the smallest stage)
struct PlaceLds { int32_t* stage; uint4* quad; uint32_t* res; int32_t* pos; };
template <int SLOTS, int STAGE>
TKZ_DEV PlaceLds tkz_place_lds(uint4* wave_quads) {
    constexpr int kStage = STAGE;
    PlaceLds L;
    L.stage = reinterpret_cast<int32_t*>(wave_quads);
    L.quad = wave_quads + kStage / 4;
    L.res = reinterpret_cast<uint32_t*>(wave_quads + kStage / 4 + SLOTS);                       // SLOTS answers + slot SLOTS: "one token"
    L.pos = reinterpret_cast<int32_t*>(wave_quads + kStage / 4 + SLOTS + SLOTS / 4 + 1);
    return L;
}
// sub-tiles sub0 .. sub0 + kPlacePer - 1, by one wavefront
// PROMO: the tables hold promoted pieces (tkz_tables.h): a hit record may carry a promo code instead of a rank -- its count is in the record, its
// tokens are one 16-byte gather from P.promo.  The form without them is the kernel of every batch before the first promotion, instruction for instruction.
template <int SLOTS, bool PROMO, int STAGE>
TKZ_DEV void tkz_place_subtiles(const EncodeParams& P, const int64_t* tile_base, int32_t* out, int64_t out_cap, int64_t sub0, const PlaceLds& LD) {
    constexpr int kPlaceSlots = SLOTS, kPlaceRes = SLOTS / 2, kStage = STAGE;
    const int lane = simt::lane();
    if (sub0 >= P.nsub) return;
    int32_t* stage = LD.stage;
    uint32_t* s_res = LD.res;
    uint4* s_quad = LD.quad;
    int32_t* s_pos = LD.pos;
    // A wavefront places kPlacePer consecutive sub-tiles, one after the other; what it needs to know about a sub-tile before it can ask for
    // its records (seven wave-uniform words) is requested while the sub-tile before it is placed: one of the sub-tile's three dependent
    // round trips leaves the chain.
    struct Sc { int64_t pb, tb, ord0; int np; uint32_t mc, hf; int gc; };
    auto load_sc = [&](int64_t sub) -> Sc {
        Sc c;
        c.pb = P.pbase[sub]; c.tb = tile_base[sub]; c.ord0 = P.docord_base[sub]; c.np = P.pcount[sub]; c.mc = P.mcount[sub]; c.hf = P.heavy_flag[sub]; c.gc = P.giant_cnt[sub];
        return c;
    };
    Sc nxt = load_sc(sub0);
    // the lane's four records of the FIRST chunk of a sub-tile (records 4 * lane .. + 3: one 16-byte load)
    auto load_recs = [&](int64_t pb_, int np_, int kk, uint32_t* r) {
        const int k0 = kk + 4 * lane;
        r[0] = r[1] = r[2] = r[3] = 0u;
        if (k0 < np_) {
            if (pb_ + k0 + 4 <= P.prank_cap) { const uint4 v = tkz_load16_nt(&P.prank[pb_ + k0]); r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
            else for (int j = 0; j < 4; ++j) if (pb_ + k0 + j < P.prank_cap) r[j] = (uint32_t)P.prank[pb_ + k0 + j];
        }
    };
#pragma unroll 1
    for (int it = 0; it < kPlacePer && sub0 + it < P.nsub; ++it) {
    const int64_t sub = sub0 + it;
    const Sc cur = nxt;
    if (it + 1 < kPlacePer && sub + 1 < P.nsub) nxt = load_sc(sub + 1);
    const int64_t pb = cur.pb, tb = cur.tb, base = sub * kSub, ord0 = cur.ord0;
    const int np = cur.np;
    const uint32_t mc = cur.mc;
    const int ns = (int)(mc & 0xFFFFu), nl = (int)(mc >> 16);
    const uint32_t* const ml = P.mlist + sub * (int64_t)P.mcap;
    const uint4* const mqd = P.mquad + sub * (int64_t)P.mcap;
    const int32_t* const dense = P.dense + (sub / kGroup) * kDenseCap;
    const bool has_giant = (cur.hf & 2u) != 0;
    int gcnt = has_giant ? cur.gc : 0;
    if (gcnt < 0) gcnt = 0;
    const bool lists_ok = ns + nl <= P.mcap;               // (cut lists: the batch is redone, nothing of this pass is used)
    // the answers of the merge kernels, in the same round trip as the first records
    // slot = lane: slots 0 .. ks-1 hold the first ks entries of the short list, slots 63 .. 64-kl the first kl of the long one
    static_assert(SLOTS == 64 || SLOTS == 128, "one or two kept entries per lane: slots lane and lane + 64");
    const int kl = ns + nl <= kPlaceSlots ? nl : (nl < kPlaceRes ? nl : kPlaceRes), ks = ns < kPlaceSlots - kl ? ns : kPlaceSlots - kl;
    const bool two = SLOTS == 128 && ks + kl > 64;         // (wave-uniform: the second slot of every lane is in use; never with 64 slots)
    const int top = two ? kPlaceSlots - 1 : 63;           // entry e of the long list is kept in slot top - e
    bool fast_ok;
    {
        // slot s holds entry s of the short list (s < ks) or entry top - s of the long one (s > top - kl); lane l loads slot l and -- only when
        // more than 64 entries are kept (top = 127 then, else 63) -- slot l + 64
        uint32_t a[2] = {0u, 0u};
        uint4 qd[2];
        bool big = false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            qd[h].x = qd[h].y = qd[h].z = qd[h].w = 0;
            if (h == 1 && !two) break;
            const int slot = lane + 64 * h;
            const bool shortside = slot < ks;
            const int e = shortside ? slot : top - slot;
            const bool want = lists_ok && (shortside || e < kl);
            const int64_t at = shortside ? e : P.mcap - 1 - e;
            if (want) { a[h] = tkz_load_nt(&ml[at]); qd[h] = tkz_load16_nt(&mqd[at]); }
            big = big || (want && tkz_result_cnt(a[h]) > kPlaceFastBig);
        }
        (void)simt::ballot(true);                                // (the sub-tile before this one is done with them)
        s_res[lane] = a[0]; s_quad[lane] = qd[0]; s_pos[lane] = -1;
        if constexpr (SLOTS == 128) { s_res[lane + 64] = a[1]; s_quad[lane + 64] = qd[1]; s_pos[lane + 64] = -1; }
        if (lane == 0) s_res[kPlaceSlots] = tkz_result_inline(1);
        // The fast path (below) takes a sub-tile whose lists are wholly in LDS (kPlaceSlots entries together) and hold no token run longer
        // than kPlaceFastBig, and that has no giant piece: nearly all of them.
        fast_ok = lists_ok && !has_giant && ns + nl <= kPlaceSlots && !simt::ballot(big);
    }
    (void)simt::ballot(true);
    // the answer a merge kernel left for a missed piece (its list entry): how many tokens, and where they wait
    auto answer = [&](uint32_t rec) -> uint32_t {
        if (rec & kPrGiant) return tkz_result_entry(false, 1, (int)(rec & 1023u));     // (count: gcnt, see below; the tokens wait in tmp at the piece's position)
        const int idx = (int)(rec & 1023u);
        const bool lg = (rec & kPrLong) != 0;
        if (!lists_ok) return tkz_result_inline(1);
        return (lg ? idx < kl : idx < ks) ? s_res[lg ? top - idx : idx] : ml[lg ? P.mcap - 1 - idx : idx];
    };
    // ... its tokens: INLINE (<= 4 tokens, nearly every missed piece: in the entry's quad, which for the first entries of both lists is
    // already in LDS), else in the group's dense region or in tmp
    auto inline_quad = [&](uint32_t rec) -> uint4 {
        const int idx = (int)(rec & 1023u);
        const bool lg = (rec & kPrLong) != 0;
        if (!lists_ok) { uint4 z; z.x = z.y = z.z = z.w = 0; return z; }
        return (lg ? idx < kl : idx < ks) ? s_quad[lg ? top - idx : idx] : tkz_load16(&mqd[lg ? P.mcap - 1 - idx : idx]);
    };
    auto token_src = [&](uint32_t res) -> const int32_t* { return (res & kMrDense) ? dense + tkz_result_off(res) : P.tmp + base + tkz_result_off(res); };
    // stage[i] holds the id of token sbase + i of the sub-tile; sbase is chosen so that stage[0] sits on a 16-byte boundary of `out`
    const uintptr_t out_addr = reinterpret_cast<uintptr_t>(out);
    auto quad_base = [&](int tok) -> int { return tok - (int)(((out_addr >> 2) + (uintptr_t)(tb + tok)) & 3u); };
    int running = 0, marks = 0, flushed = 0, sbase = quad_base(0);
    auto flush = [&](int upto) {                             // stores tokens [flushed, upto) of the sub-tile
        (void)simt::ballot(true);
        const int lo = flushed - sbase, hi = upto - sbase;
        for (int q4 = lane; 4 * q4 < hi; q4 += 64) {
            const int i0 = 4 * q4;
            if (i0 + 4 <= lo) continue;
            const uint4 v = reinterpret_cast<const uint4*>(stage)[q4];
            int32_t* dst = out + tb + sbase + i0;
            if (i0 >= lo && i0 + 4 <= hi && tb + sbase + i0 + 4 <= out_cap) tkz_store16_nt(dst, v);
            else {
                const int32_t w[4] = {(int32_t)v.x, (int32_t)v.y, (int32_t)v.z, (int32_t)v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i0 + i >= lo && i0 + i < hi && tb + sbase + i0 + i < out_cap) dst[i] = w[i];
            }
        }
        (void)simt::ballot(true);
        flushed = upto; sbase = quad_base(upto);
    };
    // ---- the fast path: FOUR CONSECUTIVE records per lane (one 16-byte load), so that 256 records cost one wave scan, one pass over the
    // marks and one pass over the missed pieces instead of four of each -- k_place runs at the VALU issue limit, and what it issues is
    // mostly per-batch overhead.  A record's token count comes from LDS without a branch (a hit reads the "one token" slot); the
    // tokens of a missed piece are written by the lane that OWNS its list entry's slot (slot e: entry e of the short list, slot top - e: entry e
    // of the long one; lane l owns slots l and l + 64): the lane that holds the record only tells it the position. ----
    auto place_fast = [&](int kk) -> bool {
        const int k0 = kk + 4 * lane;
        uint32_t r[4];
        load_recs(pb, np, kk, r);
        int c[4], idx[4];
        bool ok[4], ms[4], pm[4];
        uint4 pq0;                                               // the token quad of the lane's FIRST promoted record
        int pj = -1, pdst = 0, pcnt = 0;
        int t = 0, mk = 0;
        if constexpr (PROMO) {
#pragma unroll
            for (int j = 3; j >= 0; --j) { pm[j] = k0 + j < np && pb + k0 + j < P.prank_cap && !(r[j] & kPrMiss) && (r[j] & kPromoFlag); if (pm[j]) pj = j; }
            // One promoted piece in a hundred: a lane's four records hold one at most, nearly always.  ITS quad is requested here, ahead of the scan, and its
            // tokens are staged after the missed pieces' below -- the round trip to the promo array overlaps both (four registers across them; all four
            // records' quads, requested after the scan and waited for at once, were a dependent round trip per 256 records: 1.0 of k_place's 4.75 ms).  A
            // second promoted record of the same lane fetches its quad on the spot.
            pq0 = tkz_load16(&P.promo[pj >= 0 ? (r[pj >= 0 ? pj : 0] & kPromoIdxMask) : 0u]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ok[j] = k0 + j < np && pb + k0 + j < P.prank_cap;
            ms[j] = ok[j] && (r[j] & kPrMiss);
            idx[j] = ((r[j] & kPrLong) ? top - (int)(r[j] & 1023u) : (int)(r[j] & 1023u)) & (kPlaceSlots - 1);     // the slot of the piece's list entry
            const uint32_t a = s_res[ms[j] ? idx[j] : kPlaceSlots];
            c[j] = ok[j] ? tkz_result_cnt(a) : 0;
            if constexpr (PROMO) { if (pm[j]) c[j] = (int)((r[j] >> kPromoCntShift) & 3u) + 1; }
            t += c[j];
            mk += (ok[j] && (r[j] & kPrMark)) ? 1 : 0;
        }
        int both;
        const int pre = tkz_wave_scan_sum(t | (mk << 20), &both);
        const int tot = both & 0xFFFFF, mtot = both >> 20;
        if (running + tot - sbase > kStage) {
            if (flushed < running) flush(running);
            if (running + tot - sbase > kStage) return false;        // (more than two tokens a record: the general path, 64 records at a time)
        }
        int pos = running + (pre & 0xFFFFF), mi = marks + (pre >> 20);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ok[j] && (r[j] & kPrMark)) P.doc_tok[ord0 + mi++] = pos;
            if (ms[j]) s_pos[idx[j]] = pos;
            else if (PROMO && pm[j]) {
                if (j == pj) { pdst = pos - sbase; pcnt = c[j]; }
                else {                                           // (rare: a second promoted record in this lane's four)
                    const uint4 q = tkz_load16(&P.promo[r[j] & kPromoIdxMask]);
                    int32_t* dst = stage + (pos - sbase);
                    dst[0] = (int32_t)q.x;
                    if (c[j] > 1) dst[1] = (int32_t)q.y;
                    if (c[j] > 2) dst[2] = (int32_t)q.z;
                    if (c[j] > 3) dst[3] = (int32_t)q.w;
                }
            }
            else if (ok[j]) stage[pos - sbase] = (int32_t)(r[j] & kPrRankMask);
            pos += c[j];
        }
        (void)simt::ballot(true);
        // the missed pieces of this chunk, by the lanes that own their entries (slot lane, and slot lane + 64 when that half is in use)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            const int slot = lane + 64 * h;
            const int p = s_pos[slot];
            if (p >= 0) {                                        // (set once per entry, by the lane that holds its record)
                const uint32_t ans = s_res[slot];
                const int cnt = tkz_result_cnt(ans);
                int32_t* dst = stage + (p - sbase);
                if (ans & kMrInline) {
                    const uint4 q = s_quad[slot];
                    dst[0] = (int32_t)q.x;
                    if (cnt > 1) dst[1] = (int32_t)q.y;
                    if (cnt > 2) dst[2] = (int32_t)q.z;
                    if (cnt > 3) dst[3] = (int32_t)q.w;
                } else {
                    const int32_t* src = token_src(ans);
                    for (int i = 0; i < cnt; ++i) dst[i] = src[i];
                }
                s_pos[slot] = -1;
            }
        }
        if constexpr (PROMO) {
            if (pcnt) {
                int32_t* dst = stage + pdst;
                dst[0] = (int32_t)pq0.x;
                if (pcnt > 1) dst[1] = (int32_t)pq0.y;
                if (pcnt > 2) dst[2] = (int32_t)pq0.z;
                if (pcnt > 3) dst[3] = (int32_t)pq0.w;
            }
        }
        running += tot; marks += mtot;
        return true;
    };
#pragma unroll 1
    for (int kk = 0; kk < np; kk += 256) {                     // four record loads in flight per lane
      if (fast_ok && place_fast(kk)) continue;
      // ---- the general path (giant pieces, long token runs, lists longer than LDS keeps): 64 records at a time ----
#pragma unroll 1
      for (int j = 0; j < 4; ++j) {
        const int k0 = kk + 64 * j;
        if (k0 >= np) break;
        const int k = k0 + lane;
        const bool valid = k < np && pb + k < P.prank_cap;
        const uint32_t rec = valid ? (uint32_t)tkz_load_nt(&P.prank[pb + k]) : 0u;
        const uint32_t res = (rec & kPrMiss) ? answer(rec) : 0u;           // the merge kernels' answer for a missed piece
        const bool miss = (rec & kPrMiss) != 0;
        const bool promoted = PROMO && valid && !miss && (rec & kPromoFlag);
        uint4 pquad; pquad.x = pquad.y = pquad.z = pquad.w = 0;
        if constexpr (PROMO) { if (promoted) pquad = tkz_load16(&P.promo[rec & kPromoIdxMask]); }
        const int cnt = !valid ? 0 : !miss ? (promoted ? (int)((rec >> kPromoCntShift) & 3u) + 1 : 1) : (rec & kPrGiant) ? gcnt : tkz_result_cnt(res);
        int tot;
        const int pos = running + tkz_wave_scan_sum(cnt, &tot);
        const uint64_t mm = simt::ballot(valid && (rec & kPrMark));
        if (valid && (rec & kPrMark)) P.doc_tok[ord0 + marks + tkz_popc64(mm & tkz_lowmask(lane))] = pos;
        const uint64_t big = simt::ballot(valid && miss && cnt > kPlaceBig);
        if (!big) {
            if (running + tot - sbase > kStage) flush(running);
            if (valid) {
                int32_t* dst = stage + (pos - sbase);
                if (promoted) {
                    dst[0] = (int32_t)pquad.x;
                    if (cnt > 1) dst[1] = (int32_t)pquad.y;
                    if (cnt > 2) dst[2] = (int32_t)pquad.z;
                    if (cnt > 3) dst[3] = (int32_t)pquad.w;
                }
                else if (!miss) dst[0] = (int32_t)(rec & kPrRankMask);
                else if (res & kMrInline) {
                    const uint4 q = inline_quad(rec);
                    dst[0] = (int32_t)q.x;
                    if (cnt > 1) dst[1] = (int32_t)q.y;
                    if (cnt > 2) dst[2] = (int32_t)q.z;
                    if (cnt > 3) dst[3] = (int32_t)q.w;
                } else {
                    const int32_t* src = token_src(res);
                    for (int i = 0; i < cnt; ++i) dst[i] = src[i];
                }
            }
        } else {
            // a long token run (a merged piece of many bytes, a giant piece) in this batch of 64 records: what is staged goes out, then
            // this batch's ids go straight to their positions, the long runs copied by the whole wavefront, one after the other
            flush(running);
            const int32_t* src = (miss && !(res & kMrInline)) ? token_src(res) : nullptr;
            if (valid) {
                int32_t* dst = out + tb + pos;
                if (promoted) {
                    const int32_t w[4] = {(int32_t)pquad.x, (int32_t)pquad.y, (int32_t)pquad.z, (int32_t)pquad.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i < cnt && tb + pos + i < out_cap) dst[i] = w[i];
                }
                else if (!miss) { if (tb + pos < out_cap) tkz_store_nt(&dst[0], (int32_t)(rec & kPrRankMask)); }
                else if (res & kMrInline) {
                    const uint4 q = inline_quad(rec);
                    const int32_t w[4] = {(int32_t)q.x, (int32_t)q.y, (int32_t)q.z, (int32_t)q.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i < cnt && tb + pos + i < out_cap) dst[i] = w[i];
                } else if (cnt <= kPlaceBig) {
                    for (int i = 0; i < cnt; ++i) if (tb + pos + i < out_cap) dst[i] = src[i];
                }
            }
            for (uint64_t bg = big; bg; bg &= bg - 1) {
                const int src_lane = tkz_ctz64(bg);
                const int c = simt::shfl(cnt, src_lane), p0 = simt::shfl(pos, src_lane);
                const uint32_t lo32 = simt::shflu((uint32_t)(reinterpret_cast<uintptr_t>(src)), src_lane), hi32 = simt::shflu((uint32_t)(reinterpret_cast<uintptr_t>(src) >> 32), src_lane);
                const int32_t* bsrc = reinterpret_cast<const int32_t*>(((uintptr_t)hi32 << 32) | lo32);
                int32_t* dst = out + tb + p0;
                for (int i = lane; i < c; i += 64) if (tb + p0 + i < out_cap) dst[i] = bsrc[i];
            }
            flushed = running + tot; sbase = quad_base(flushed);
        }
        running += tot;
        marks += tkz_popc64(mm);
      }
    }
    if (flushed < running) flush(running);
    }
}
template <int SLOTS, bool PROMO>
TKZ_KERNEL_OCC(256, TKZ_PLACE_OCC) void k_place(EncodeParams P, const int64_t* tile_base, int32_t* out, int64_t out_cap) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED uint4 s_wave[kThreads / 64][kPlaceLdsQuadsT<SLOTS, kStageBatch>];
    const int64_t sub0 = (tkz_xcd_block(simt::bid(), simt::nblocks()) * (kThreads / 64) + simt::wave()) * kPlacePer;
    tkz_place_subtiles<SLOTS, PROMO, kStageBatch>(P, tile_base, out, out_cap, sub0, tkz_place_lds<SLOTS, kStageBatch>(s_wave[simt::wave()]));
}

// -------------------------------------------------------------------------------------------------
// giant pieces (> kArenaPiece bytes: a run of thousands of letters, of '=' ...): found on the bitmap, merged by a whole
// 1024-thread workgroup each (tkz_bpe_long, rounds) before the encode kernels run
// -------------------------------------------------------------------------------------------------
// only a sub-tile k_probe flagged can hold the start of a giant piece, and only its LAST piece start can be one (a giant piece reaches past the
// end of its sub-tile): sub-tile t's, queued with its length (k_list_stats calls this for the flagged sub-tiles)
TKZ_DEV void tkz_giant_find_one(int64_t t, const uint64_t* startbits, int64_t nwords, int64_t total, int64_t* gq, unsigned long long* gcount, int64_t gcap) {
    int64_t p = -1;
    for (int k = kSub / 64 - 1; k >= 0 && p < 0; --k) {
        const int64_t w = t * (kSub / 64) + k;
        const uint64_t m = w < nwords ? startbits[w] : 0;
        if (m) p = (w << 6) + tkz_msb64(m);
    }
    if (p < 0 || p >= total) return;                         // (no piece starts here / the sentinel)
    int64_t next = total;
    for (int64_t v = (t + 1) * (kSub / 64); v < nwords; ++v) {
        const uint64_t x = startbits[v];
        if (x) { next = (v << 6) + tkz_ctz64(x); break; }
    }
    if (next - p > kArenaPiece) {
        const unsigned long long q = simt::atomic_add64(gcount, 1ull);
        if ((int64_t)q < gcap) { gq[2 * q] = p; gq[2 * q + 1] = next - p; }
    }
}
// The order the giant pieces are taken in: longest first (a diverse piece of tens of KiB is merged in thousands of rounds and sets the
// kernel's duration, so those must start first), by rank counting in one workgroup; identity beyond 8192 pieces.
constexpr int kGiantSort = 8192;
TKZ_KERNEL(1024) void k_giant_order(EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED int32_t s_len[kGiantSort];
    const int64_t n = (int64_t)*P.giant_count < P.giant_cap ? (int64_t)*P.giant_count : P.giant_cap;
    int64_t* order = P.giant_q + 2 * P.giant_cap;
    if (n > kGiantSort) { for (int64_t i = simt::tid(); i < n; i += simt::nthreads()) order[i] = i; return; }
    for (int i = simt::tid(); i < (int)n; i += simt::nthreads()) { const int64_t l = P.giant_q[2 * i + 1]; s_len[i] = l > 0x7FFFFFFF ? 0x7FFFFFFF : (int32_t)l; }
    simt::sync();
    for (int i = simt::tid(); i < (int)n; i += simt::nthreads()) {
        const int32_t li = s_len[i];
        int rank = 0;
        for (int j = 0; j < (int)n; ++j) { const int32_t lj = s_len[j]; rank += (lj > li || (lj == li && j < i)) ? 1 : 0; }
        order[rank] = i;
    }
}

TKZ_KERNEL(1024) void k_giant_merge(TkzTables T, EncodeParams P) {
    if (tkz_attempt_failed(P)) return;
    TKZ_SHARED alignas(16) int32_t s_state[kBpeLongLdsBytes / 4];             // ids | pair ranks | flag bytes of up to 16 Ki parts + the tail's bounds: 152 KB of the CU's 160
    TKZ_SHARED int64_t s_off;
    TKZ_SHARED int32_t s_whole;
    TKZ_SHARED unsigned long long s_ticket;
    const int64_t n = (int64_t)*P.giant_count < P.giant_cap ? (int64_t)*P.giant_count : P.giant_cap;
    const int64_t* order = P.giant_q + 2 * P.giant_cap;
    int err = 0;
    // every workgroup keeps taking the next piece of the longest-first order until none is left (a fixed share per workgroup left the
    // ones with two long pieces working twice as long as the others)
    for (;;) {
        if (simt::tid() == 0) s_ticket = simt::atomic_add64(P.giant_ticket, 1ull);
        simt::sync();
        const int64_t t = (int64_t)s_ticket;
        simt::sync();
        if (t >= n) break;
        const int64_t q = order[t];
        const int64_t p = P.giant_q[2 * q], len64 = P.giant_q[2 * q + 1];
        int cnt = 0;
        if (len64 > kMaxPiece) err |= kErrTooLong;
        else {
            const int len = (int)len64;
            const uint8_t* gb = P.bytes + p;
            auto at = [&](int i) -> uint32_t { return gb[i]; };
            int32_t* dst = P.tmp + p;
            if (simt::tid() == 0) {
                s_whole = tkz_lookup_long(T, at, (uint32_t)len);                 // TikTokenizer.cs:262 (a key this long is unusual, not impossible)
                const unsigned long long need = 6ull * (unsigned long long)len;
                const unsigned long long o = simt::atomic_add64(P.pool_head, need);
                s_off = (o + need <= (unsigned long long)P.pool_cap) ? (int64_t)o : -1;
            }
            simt::sync();
            const int32_t whole = s_whole;
            const int64_t off = s_off;
            simt::sync();
            if (whole != TKZ_RANK_NONE) { if (simt::tid() == 0) dst[0] = whole; cnt = 1; }
            else if (off < 0) { err |= kErrPool; cnt = -1; }
            else {
                int32_t* arr = P.pool + off;
                cnt = tkz_bpe_long(T, at, len, arr, arr + len, arr + 2 * (int64_t)len, arr + 3 * (int64_t)len, arr + 4 * (int64_t)len, arr + 5 * (int64_t)len, dst, &err, s_state, TKZ_DEV_FLAG(P, 16) ? P.devprof : nullptr);
            }
        }
        if (simt::tid() == 0) P.giant_cnt[p / kSub] = cnt;
        simt::sync();
    }
    if (err) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)err);
}

// documents that start in each sub-tile (for the document ordinals)
TKZ_KERNEL(256) void k_doccount(const uint64_t* docbits, int64_t nwords, int64_t total, int64_t nsub, int32_t* cnt) {
    // one lane per bitmap word (coalesced), the kSub/64 = 16 words of a sub-tile summed across 16 lanes
    static_assert(kSub / 64 == 16, "k_doccount sums 16 lanes per sub-tile");
    const int lane = simt::lane();
    const int64_t stride = simt::nblocks() * simt::nthreads(), nw = nsub * (kSub / 64);
    for (int64_t w0 = simt::bid() * simt::nthreads() + (simt::tid() & ~63); w0 < nw; w0 += stride) {
        const int64_t w = w0 + lane;
        uint64_t m = (w < nw && w < nwords) ? docbits[w] : 0ull;
        const int64_t lim = total - (w << 6);              // (the sentinel bit at `total` is not a start)
        if (lim <= 0) m = 0; else if (lim < 64) m &= tkz_lowmask((int)lim);
        int c = tkz_popc64(m);
        c += simt::shfl_xor(c, 1); c += simt::shfl_xor(c, 2); c += simt::shfl_xor(c, 4); c += simt::shfl_xor(c, 8);
        if ((lane & 15) == 0 && w < nw) cnt[w >> 4] = c;
    }
}

// the same for TWO bitmaps in one pass (document starts and piece starts: one launch instead of two, the words of both read by the same lane)
TKZ_KERNEL(256) void k_doccount2(const uint64_t* bits_a, const uint64_t* bits_b, int64_t nwords, int64_t total, int64_t nsub, int32_t* cnt_a, int32_t* cnt_b) {
    static_assert(kSub / 64 == 16, "k_doccount2 sums 16 lanes per sub-tile");
    const int lane = simt::lane();
    const int64_t stride = simt::nblocks() * simt::nthreads(), nw = nsub * (kSub / 64);
    for (int64_t w0 = simt::bid() * simt::nthreads() + (simt::tid() & ~63); w0 < nw; w0 += stride) {
        const int64_t w = w0 + lane;
        uint64_t ma = (w < nw && w < nwords) ? bits_a[w] : 0ull, mb = (w < nw && w < nwords) ? bits_b[w] : 0ull;
        const int64_t lim = total - (w << 6);              // (the sentinel bit at `total` is not a start)
        if (lim <= 0) { ma = 0; mb = 0; } else if (lim < 64) { ma &= tkz_lowmask((int)lim); mb &= tkz_lowmask((int)lim); }
        int c = tkz_popc64(ma) | (tkz_popc64(mb) << 16);  // (at most 1024 of either in a sub-tile: the two sums share the shuffles)
        c += simt::shfl_xor(c, 1); c += simt::shfl_xor(c, 2); c += simt::shfl_xor(c, 4); c += simt::shfl_xor(c, 8);
        if ((lane & 15) == 0 && w < nw) { cnt_a[w >> 4] = c & 0xFFFF; cnt_b[w >> 4] = c >> 16; }
    }
}

// -------------------------------------------------------------------------------------------------
// exclusive scan of tile_count (int32) -> tile_base (int64); counters[2..3] (int64) = grand total
// -------------------------------------------------------------------------------------------------
// Up to kScanSmallMax sub-tiles: ONE workgroup, one launch -- a thread owns n / 1024 consecutive counts, one workgroup scan of the
// threads' sums -- for one or two arrays at once; the three-kernel form below (partials, top, final) costs three dependent launches per array, and a
// batch of a few megabytes is made of little else than launches.
constexpr int kScanSmallMax = 8192;          // (8 MB of text.  At 65,536 sub-tiles the one workgroup took 35 us a scan against 17 us for the three kernels: measured on the 64 MB single document of --kind 5)
TKZ_KERNEL(1024) void k_scan_small(const int32_t* cnt_a, int64_t* base_a, int64_t* grand_a, int round_a,
                                   const int32_t* cnt_b, int64_t* base_b, int64_t* grand_b, int round_b, int n) {
    const int tid = simt::tid(), nth = simt::nthreads();
    const int per = (n + nth - 1) / nth, lo = tid * per, hi = lo + per < n ? lo + per : n;
    for (int arr = 0; arr < 2; ++arr) {
        const int32_t* cnt = arr ? cnt_b : cnt_a;
        int64_t* base = arr ? base_b : base_a;
        int64_t* grand = arr ? grand_b : grand_a;
        const int round = arr ? round_b : round_a;
        if (!cnt) continue;                                   // (workgroup-uniform)
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += (cnt[i] + round) & ~round;
        int tot;
        int run = tkz_block_scan(sum, &tot);                  // (<= 1024 per sub-tile, <= 65536 sub-tiles: 2^26 at most)
        for (int i = lo; i < hi; ++i) { base[i] = run; run += (cnt[i] + round) & ~round; }
        if (tid == 0 && grand) *grand = tot;
        simt::sync();
    }
}
// (`round`: every count is rounded up to a multiple of round + 1 first -- the piece records of a sub-tile are whole 64-byte lines)
TKZ_KERNEL(256) void k_scan_partials(const int32_t* cnt, int64_t n, int64_t* bsum, int round) {
    TKZ_SHARED int64_t s_w[4];
    const int64_t i0 = simt::bid() * kScanBlock;
    int64_t v = 0;
    for (int j = simt::tid(); j < kScanBlock; j += kThreads) if (i0 + j < n) v += (cnt[i0 + j] + round) & ~round;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = simt::shflu((uint32_t)v, simt::lane() ^ d), hi = simt::shflu((uint32_t)((uint64_t)v >> 32), simt::lane() ^ d);
        v += (int64_t)(((uint64_t)hi << 32) | lo);
    }
    if (simt::lane() == 0) s_w[simt::wave()] = v;
    simt::sync();
    if (simt::tid() == 0) bsum[simt::bid()] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
TKZ_KERNEL(256) void k_scan_top(int64_t* bsum, int64_t nblk, int64_t* grand) {   // one workgroup, kThreads partial sums per step
    TKZ_SHARED int64_t s_w[kThreads / 64];
    const int lane = simt::lane(), wave = simt::wave();
    int64_t carry = 0;
    for (int64_t i0 = 0; i0 < nblk; i0 += kThreads) {
        const int64_t i = i0 + simt::tid();
        const int64_t v = i < nblk ? bsum[i] : 0;
        int64_t x = v;                                   // inclusive scan inside the wavefront
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = simt::shflu((uint32_t)x, lane - d < 0 ? lane : lane - d);
            const uint32_t hi = simt::shflu((uint32_t)((uint64_t)x >> 32), lane - d < 0 ? lane : lane - d);
            if (lane >= d) x += (int64_t)(((uint64_t)hi << 32) | lo);
        }
        if (lane == 63) s_w[wave] = x;
        simt::sync();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < kThreads / 64; ++w) { const int64_t t = s_w[w]; if (w < wave) woff += t; tot += t; }
        simt::sync();
        if (i < nblk) bsum[i] = carry + woff + x - v;
        carry += tot;
    }
    if (simt::tid() == 0) *grand = carry;
}
TKZ_KERNEL(256) void k_scan_final(const int32_t* cnt, int64_t n, const int64_t* boff, int64_t* base, int round) {
    const int64_t i0 = simt::bid() * kScanBlock;
    // each thread owns kScanBlock/kThreads consecutive tiles
    constexpr int per = kScanBlock / kThreads;
    int local[per]; int sum = 0;
    for (int j = 0; j < per; ++j) { const int64_t i = i0 + (int64_t)simt::tid() * per + j; local[j] = i < n ? (cnt[i] + round) & ~round : 0; sum += local[j]; }
    // block scan in 64-bit via two 32-bit scans would overflow only past 2^31 tokens per 1024 tiles (4 MiB of text): impossible
    int tot;
    int pre = tkz_block_scan(sum, &tot);
    int64_t run = boff[simt::bid()] + pre;
    for (int j = 0; j < per; ++j) { const int64_t i = i0 + (int64_t)simt::tid() * per + j; if (i < n) base[i] = run; run += local[j]; }
}

// -------------------------------------------------------------------------------------------------
// k_docoffs
// -------------------------------------------------------------------------------------------------
// (c3a / c3b / c3c: null, or blocks that receive {c3_docs, total, *grand} -- the batch's counts, k_counts3's job, by the first thread: one launch less)
TKZ_KERNEL(256) void k_docoffs(const int64_t* offs, int64_t n_docs, int64_t total, const int64_t* tile_base, const uint64_t* docbits,
                               const int64_t* docord_base, const int32_t* doc_tok, const int64_t* grand, int64_t* out_offs,
                               int64_t c3_docs, int64_t* c3a, int64_t* c3b, int64_t* c3c) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    if (simt::bid() == 0 && simt::tid() == 0) {
        const int64_t g = *grand;
        if (c3a) { c3a[0] = c3_docs; c3a[1] = total; c3a[2] = g; }
        if (c3b) { c3b[0] = c3_docs; c3b[1] = total; c3b[2] = g; }
        if (c3c) { c3c[0] = c3_docs; c3c[1] = total; c3c[2] = g; }
    }
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d <= n_docs; d += stride) {
        const int64_t pos = offs[d];
        if (pos >= total) { out_offs[d] = *grand; continue; }
        if (pos < 0) { out_offs[d] = 0; continue; }
        const int64_t sub = pos / kSub, wpos = pos >> 6;
        int64_t ord = docord_base[sub];
        for (int64_t w = sub * (kSub / 64); w < wpos; ++w) ord += tkz_popc64(docbits[w]);
        ord += tkz_popc64(docbits[wpos] & tkz_lowmask((int)(pos & 63)));
        out_offs[d] = tile_base[sub] + doc_tok[ord];
    }
}

// TKZ_OPT_CASE_EQUIVALENCE (cl100k on a .NET >= 7 host): `(?i:'s|'t|...)` there is matched with the runtime's case-equivalence tables, under which
// U+017F (LATIN SMALL LETTER LONG S, bytes C5 BF) is an `s`: an apostrophe at which a match starts, followed by U+017F, IS the contraction -- the piece
// ends behind the long s.  Every scanner treats U+017F as the letter it is (net6.0: ASCII case pairs only; the apostrophe is then the one-char prefix
// of the word that follows), so the only difference is one more piece start: right behind the long s.  This pass over the text adds it, whichever scanner
// wrote the bitmap.  (None of the other letters of the seven literals has a non-ASCII equivalent: U+212A KELVIN SIGN is a `k`.)
TKZ_KERNEL(256) void k_case_equiv_fix(const uint8_t* bytes, int64_t total, const uint64_t* docbits, uint64_t* startbits) {
    const int64_t stride = simt::nblocks() * simt::nthreads() * 8;
    for (int64_t i0 = (simt::bid() * (int64_t)simt::nthreads() + simt::tid()) * 8; i0 < total; i0 += stride) {
        uint64_t w = 0;                                      // (bytes is 16-byte aligned; nothing is read beyond `total`)
        if (i0 + 8 <= total) w = *reinterpret_cast<const uint64_t*>(bytes + i0);
        else for (int k = 0; i0 + k < total; ++k) w |= (uint64_t)bytes[i0 + k] << (8 * k);
        const uint64_t x = w ^ 0x2727272727272727ull;
        if (!((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull)) continue;      // no apostrophe among these eight bytes
        for (int k = 0; k < 8; ++k) {
            const int64_t i = i0 + k;
            if (i + 3 > total || ((w >> (8 * k)) & 0xFFu) != 0x27u) continue;
            if (bytes[i + 1] != 0xC5u || bytes[i + 2] != 0xBFu) continue;
            if (!((startbits[i >> 6] >> (i & 63)) & 1ull)) continue;                     // no match starts at the apostrophe
            if (((docbits[(i + 1) >> 6] >> ((i + 1) & 63)) & 1ull) || ((docbits[(i + 2) >> 6] >> ((i + 2) & 63)) & 1ull)) continue;   // (the document ends behind the apostrophe)
            simt::atomic_or64((unsigned long long*)&startbits[(i + 3) >> 6], 1ull << ((i + 3) & 63));
        }
    }
}

// document offsets of a chunk cut out of a larger batch: made relative to the chunk's first byte
// A small host batch on page-locked buffers: the text and the offsets come over PCIe by this kernel's own loads (they are device-visible), into the
// staging buffers the other kernels read, and the workspace's zero region is cleared by the same launch -- in place of two copy commands, two fills and
// the stream hand-over between them (68 us before the first kernel of a 1 MB call; this is ~25).  h_bytes 16-byte aligned.
TKZ_KERNEL(256) void k_ingest(const uint8_t* h_bytes, int64_t total, uint8_t* d_bytes, const int64_t* h_offs, int64_t n_offs, int64_t* d_offs, uint4* zero, int64_t zero_quads) {
    const int64_t i0 = (int64_t)simt::bid() * simt::nthreads() + simt::tid(), stride = (int64_t)simt::nblocks() * simt::nthreads();
    const int64_t full = total >> 4;
    for (int64_t i = i0; i < full; i += stride) *reinterpret_cast<uint4*>(d_bytes + 16 * i) = tkz_load16(h_bytes + 16 * i);
    for (int64_t i = (full << 4) + i0; i < total; i += stride) d_bytes[i] = h_bytes[i];          // (nothing is read beyond the caller's buffer)
    for (int64_t i = i0; i < n_offs; i += stride) d_offs[i] = h_offs[i];
    const uint4 z = {0u, 0u, 0u, 0u};
    for (int64_t i = i0; i < zero_quads; i += stride) zero[i] = z;
}
TKZ_KERNEL(256) void k_rebase(int64_t* offs, int64_t n, int64_t base) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t i = simt::bid() * simt::nthreads() + simt::tid(); i < n; i += stride) offs[i] -= base;
}

// What the miss lists of the batch needed, from a pass over mcount (a few atomics per workgroup; k_probe used to keep these itself, every wavefront
// reading -- and many updating -- ONE line of the counter block: 8 ms of a miss-heavy batch, 100 ms of a batch whose lists all overflow):
//   counters[0] |= kErrMissCap, counters[1] = the longest list, when some list did not fit mcap (the host grows the lists and runs the batch again);
//   counters[2] = the longest list above kMissCapMin that did fit (the host lets grown lists shrink again by it);
//   counters[3] = how many sub-tiles hold more than 64 entries (the host picks k_place's form for the workspace's next batch by it).
// (... and, in the same pass over the sub-tiles, k_giant_find's: a sub-tile k_probe flagged as holding a giant piece queues it -- one launch less)
TKZ_KERNEL(256) void k_list_stats(const uint32_t* mcount, int64_t nsub, int32_t mcap, int32_t* counters,
                                  const uint8_t* heavy_flag, const uint64_t* startbits, int64_t nwords, int64_t total, int64_t* gq, unsigned long long* gcount, int64_t gcap,
                                  const uint32_t* mlist, uint64_t* coop_q, unsigned long long* coop_count, int64_t coop_cap, int lane_piece, unsigned long long* miss_sums) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    const int lane = simt::lane();
    int mx = 0, big = 0, over = 0;
    long long n_short = 0, n_long = 0;       // (miss_sums: what the batch's pieces missed -- the host follows that share from batch to batch, TKZ_OPT_ADAPT)
    for (int64_t i0 = simt::bid() * simt::nthreads() + (simt::tid() & ~63); i0 < nsub; i0 += stride) {      // (wave-uniform: the queueing below is a wave scan)
        const int64_t i = i0 + lane;
        const bool in = i < nsub;
        const uint32_t m = in ? mcount[i] : 0u;
        const int n = (int)(m & 0xFFFFu) + (int)(m >> 16);
        n_short += (int)(m & 0xFFFFu); n_long += (int)(m >> 16);
        if (n <= mcap) { if (n > mx) mx = n; } else if (n > big) big = n;
        over += n > 64 ? 1 : 0;
        const uint32_t hf = in ? heavy_flag[i] : 0u;
        if (hf & 2u) tkz_giant_find_one(i, startbits, nwords, total, gq, gcount, gcap);
        // the long misses of more than kLanePiece bytes (k_probe flagged their sub-tiles: bit 2) are queued for k_merge_coop: sub-tile << 10 | index in
        // its long list.  One atomic per wavefront of 64 sub-tiles, not per piece.
        if (simt::ballot((hf & 4u) != 0)) {
            const int nl = (int)(m >> 16);
            int mine = 0;
            // (four entries a step, their loads requested together: a lane walks its sub-tile's long list alone -- real text has lists of forty entries, one
            //  round trip each was 139 us of this kernel on 268 MB)
            const uint32_t* const top = mlist + i * (int64_t)mcap + (mcap - 1);
            const bool walk = (hf & 4u) && n <= mcap;
            if (walk)
                for (int j = 0; j < nl; j += 4) {
                    uint32_t ent[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ent[k] = j + k < nl ? top[-(j + k)] : 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) mine += ((int)((ent[k] >> kMrLenShift) & 1023u) + 1 > lane_piece) ? 1 : 0;
                }
            int tot;
            const int pre = tkz_wave_scan_sum(mine, &tot);
            unsigned long long base = 0;
            if (lane == 0 && tot) base = simt::atomic_add64(coop_count, (unsigned long long)tot);
            base = ((unsigned long long)simt::shflu((uint32_t)(base >> 32), 0) << 32) | simt::shflu((uint32_t)base, 0);
            if (mine) {
                unsigned long long at = base + (unsigned long long)pre;
                for (int j = 0; j < nl; j += 4) {
                    uint32_t ent[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ent[k] = j + k < nl ? top[-(j + k)] : 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((int)((ent[k] >> kMrLenShift) & 1023u) + 1 > lane_piece) { if ((int64_t)at < coop_cap) coop_q[at] = ((uint64_t)i << 10) | (uint64_t)(j + k); ++at; }
                }
            }
        }
    }
    int tot;
    (void)tkz_wave_scan_sum(over, &tot);
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = simt::shfl(mx, simt::lane() ^ d), b = simt::shfl(big, simt::lane() ^ d);
        mx = o > mx ? o : mx; big = b > big ? b : big;
    }
    // (64-bit sums over the wave by two 32-bit scans: a wavefront's share is far below 2^31)
    int ts = 0, tl = 0;
    if (miss_sums) { (void)tkz_wave_scan_sum((int)n_short, &ts); (void)tkz_wave_scan_sum((int)n_long, &tl); }
    // One set of atomics a WORKGROUP: its four wavefronts meet in LDS first.  (A set per wavefront was 4,096 x 6 atomics on the same few words: ~110 us of this
    // kernel whatever the batch -- 268 MB of real text or the 5 GB headline.)
    TKZ_SHARED int s_red[kThreads / 64][5];
    if (simt::lane() == 0) { int* r = s_red[simt::wave()]; r[0] = big; r[1] = mx; r[2] = tot; r[3] = ts; r[4] = tl; }
    simt::sync();
    if (simt::tid() == 0) {
        long long sum_s = 0, sum_l = 0;
        big = 0; mx = 0; tot = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            const int* r = s_red[w];
            big = r[0] > big ? r[0] : big; mx = r[1] > mx ? r[1] : mx; tot += r[2]; sum_s += r[3]; sum_l += r[4];
        }
        if (big) { simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrMissCap); simt::atomic_max((unsigned*)&counters[1], (unsigned)big); }
        if (mx > kMissCapMin) simt::atomic_max((unsigned*)&counters[2], (unsigned)mx);
        if (tot) simt::atomic_add(&counters[3], tot);
        if (miss_sums && (sum_s | sum_l)) { simt::atomic_add64(&miss_sums[0], (unsigned long long)sum_s); simt::atomic_add64(&miss_sums[1], (unsigned long long)sum_l); }
    }
}

// TKZ_OPT_PIECE_STATS: the lengths of the miss lists of all sub-tiles, summed (short | long << 16 per sub-tile)
TKZ_KERNEL(256) void k_miss_stats(const uint32_t* mcount, const int32_t* pcount, int64_t nsub, unsigned long long* stats) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    int ns = 0, nl = 0, np = 0;
    for (int64_t i = simt::bid() * simt::nthreads() + simt::tid(); i < nsub; i += stride) { const uint32_t m = mcount[i]; ns += (int)(m & 0xFFFFu); nl += (int)(m >> 16); np += pcount[i]; }
    int ts, tl, tp;
    (void)tkz_wave_scan_sum(ns, &ts);
    (void)tkz_wave_scan_sum(nl, &tl);
    (void)tkz_wave_scan_sum(np, &tp);
    if (simt::lane() == 0 && (ts | tl | tp)) {
        simt::atomic_add64(&stats[2], (unsigned long long)ts); simt::atomic_add64(&stats[3], (unsigned long long)tl); simt::atomic_add64(&stats[4], (unsigned long long)tp);
    }
}

// {n_docs, n_bytes, n_tokens} of the batch, on the device: what tkz_comm_allgather_counts_device sends (no host round trip)
// (up to three copies: the encoder's "last batch" block, the workspace's block of THIS batch, a block of the caller's)
TKZ_KERNEL(64) void k_counts3(int64_t n_docs, int64_t total, const int64_t* grand, int64_t* out3, int64_t* out3b, int64_t* out3c) {
    if (simt::tid() == 0) {
        const int64_t g = grand ? *grand : 0;
        out3[0] = n_docs; out3[1] = total; out3[2] = g;
        if (out3b) { out3b[0] = n_docs; out3b[1] = total; out3b[2] = g; }
        if (out3c) { out3c[0] = n_docs; out3c[1] = total; out3c[2] = g; }
    }
}

// -------------------------------------------------------------------------------------------------
// UTF-16 documents -> UTF-8 documents on the device (the batch form of Encoding.UTF8.GetBytes, TikTokenizer.cs:261):
// a host whose strings are UTF-16 (.NET, Java, JS) uploads the code units as they are.  A surrogate pair becomes one
// 4-byte char, a lone surrogate (also: a pair cut by a document boundary) becomes EF BF BD, as GetBytes does.
//   k_u16_len     per 1024-unit tile (one wavefront, 16 units per lane): UTF-8 length of every unit -> tile sum and the
//                 exclusive prefix of every 16-unit group inside the tile
//   (scan of the tile sums: k_scan_*)
//   k_u16_write   the bytes, staged per tile in LDS and copied out coalesced
//   k_u16_docoffs byte offset of every document = tile base + group prefix + the units of the group before it
// -------------------------------------------------------------------------------------------------
constexpr int kU16Tile = 1024;      // code units per wavefront
constexpr int kU16Lane = 16;        // ... per lane
// UTF-8 length of unit u at a position whose neighbours are prev/next (0 when there is none inside the document):
// 0 for the low half of a pair (its bytes belong to the high half)
TKZ_HD int tkz_u16_len(uint32_t u, uint32_t prev, bool has_prev, uint32_t next, bool has_next) {
    if (u < 0x80u) return 1;
    if (u < 0x800u) return 2;
    if (u - 0xD800u < 0x400u) return (has_next && next - 0xDC00u < 0x400u) ? 4 : 3;      // high surrogate: pair, or alone -> U+FFFD
    if (u - 0xDC00u < 0x400u) return (has_prev && prev - 0xD800u < 0x400u) ? 0 : 3;      // low surrogate: second half, or alone -> U+FFFD
    return 3;
}
TKZ_HD int tkz_u16_put(uint32_t u, uint32_t next, int len, uint8_t* o) {
    if (len == 1) { o[0] = (uint8_t)u; }
    else if (len == 2) { o[0] = (uint8_t)(0xC0u | (u >> 6)); o[1] = (uint8_t)(0x80u | (u & 0x3Fu)); }
    else if (len == 4) {
        const uint32_t c = 0x10000u + ((u - 0xD800u) << 10) + (next - 0xDC00u);
        o[0] = (uint8_t)(0xF0u | (c >> 18)); o[1] = (uint8_t)(0x80u | ((c >> 12) & 0x3Fu)); o[2] = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu)); o[3] = (uint8_t)(0x80u | (c & 0x3Fu));
    } else if (len == 3) {
        const uint32_t c = (u - 0xD800u < 0x800u) ? 0xFFFDu : u;
        o[0] = (uint8_t)(0xE0u | (c >> 12)); o[1] = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu)); o[2] = (uint8_t)(0x80u | (c & 0x3Fu));
    }
    return len;
}
// the units of this lane's group (and the one after it), its document-start bits, the lengths; returns the lane sum
struct TkzU16Lane { uint32_t u[kU16Lane + 1]; uint8_t len[kU16Lane]; };
TKZ_DEV int tkz_u16_lane(const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t tile, TkzU16Lane* L) {
    const int lane = simt::lane();
    const int64_t p0 = tile * kU16Tile + (int64_t)lane * kU16Lane;
    uint32_t w[8];
    if (p0 + kU16Lane <= total) {
        const uint4 a = tkz_load16(units + p0), b = tkz_load16(units + p0 + 8);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
        for (int k = 0; k < 8; ++k) {
            const int64_t q = p0 + 2 * k;
            w[k] = (q < total ? (uint32_t)units[q] : 0u) | ((q + 1 < total ? (uint32_t)units[q + 1] : 0u) << 16);
        }
    }
#pragma unroll
    for (int k = 0; k < kU16Lane; ++k) L->u[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
    // neighbours across the lane edge: the last unit of the lane below, the first unit of the lane above
    uint32_t prev = simt::shflu(L->u[kU16Lane - 1], (lane + 63) & 63), next = simt::shflu(L->u[0], (lane + 1) & 63);
    if (lane == 0) prev = p0 > 0 ? (uint32_t)units[p0 - 1] : 0u;
    if (lane == 63) next = p0 + kU16Lane < total ? (uint32_t)units[p0 + kU16Lane] : 0u;
    L->u[kU16Lane] = next;
    // document-start bits of positions p0 .. p0 + 16 (17 bits)
    const int64_t wd = p0 >> 6; const int sh = (int)(p0 & 63);
    uint64_t ds = p0 < total ? docbits[wd] >> sh : 0ull;                       // (docbits has total/64 + 1 words)
    if (sh == 48 && p0 + kU16Lane <= total) ds |= (docbits[wd + 1] & 1ull) << 16;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kU16Lane; ++k) {
        const int64_t q = p0 + k;
        int len = 0;
        if (q < total) {
            const bool has_prev = q > 0 && !((ds >> k) & 1ull);                 // not the first unit of its document
            const bool has_next = q + 1 < total && !((ds >> (k + 1)) & 1ull);   // not the last one
            len = tkz_u16_len(L->u[k], k ? L->u[k - 1] : prev, has_prev, L->u[k + 1], has_next);
        }
        L->len[k] = (uint8_t)len; sum += len;
    }
    return sum;
}
TKZ_KERNEL(256) void k_u16_len(const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t ntiles, int32_t* grp_prefix, int32_t* tile_sum) {
    const int64_t tile = simt::bid() * (kThreads / 64) + simt::wave();
    if (tile >= ntiles) return;
    TkzU16Lane L;
    const int sum = tkz_u16_lane(units, total, docbits, tile, &L);
    int tot;
    const int pre = tkz_wave_scan_sum(sum, &tot);
    grp_prefix[tile * 64 + simt::lane()] = pre;
    if (simt::lane() == 0) tile_sum[tile] = tot;
}
TKZ_KERNEL(256) void k_u16_write(const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t ntiles, const int64_t* tile_base,
                                 uint8_t* out) {
    TKZ_SHARED uint8_t s_stage[kThreads / 64][3 * kU16Tile + 16];            // (a unit yields at most 3 bytes: a pair is 4 for 2 units)
    const int64_t tile = simt::bid() * (kThreads / 64) + simt::wave();
    if (tile >= ntiles) return;
    TkzU16Lane L;
    const int sum = tkz_u16_lane(units, total, docbits, tile, &L);
    int tot;
    int pos = tkz_wave_scan_sum(sum, &tot);
    uint8_t* st = s_stage[simt::wave()];
#pragma unroll
    for (int k = 0; k < kU16Lane; ++k) pos += tkz_u16_put(L.u[k], L.u[k + 1], L.len[k], st + pos);
    (void)simt::ballot(true);          // (the staging area is private to the wavefront: its LDS accesses are ordered, no barrier)
    uint8_t* dst = out + tile_base[tile];
    for (int i = simt::lane(); i < tot; i += 64) dst[i] = st[i];
}
TKZ_KERNEL(256) void k_u16_docoffs(const uint16_t* units, int64_t total, const uint64_t* docbits, const int64_t* unit_offs, int64_t n_docs,
                                   const int64_t* tile_base, const int32_t* grp_prefix, const int64_t* grand, int64_t* byte_offs) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d <= n_docs; d += stride) {
        const int64_t p = unit_offs[d];
        if (p >= total) { byte_offs[d] = *grand; continue; }
        if (p < 0) { byte_offs[d] = 0; continue; }
        const int64_t g0 = p & ~(int64_t)(kU16Lane - 1);
        int64_t v = tile_base[p / kU16Tile] + grp_prefix[p / kU16Lane];
        for (int64_t q = g0; q < p; ++q) {                                 // the units of the group that belong to the document before
            const bool has_prev = q > 0 && !((docbits[q >> 6] >> (q & 63)) & 1ull);
            const bool has_next = q + 1 < total && !((docbits[(q + 1) >> 6] >> ((q + 1) & 63)) & 1ull);
            v += tkz_u16_len(units[q], q > 0 ? units[q - 1] : 0u, has_prev, q + 1 < total ? units[q + 1] : 0u, has_next);
        }
        byte_offs[d] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// piece granularity (EncodeTrimSuffix / EncodeTrimPrefix walk the regex matches, TikTokenizer.cs:288-341, :483-519): the byte
// offset of every piece and the first piece of every document, straight from the piece-start bitmap -- no host round trip
//   (k_doccount + scan over STARTBITS give every sub-tile its first piece ordinal)
//   k_piece_index   one wavefront per sub-tile: positions of the set bits below `total`, in order
//   k_doc_piece     first piece of document d = number of piece starts before its first byte
// The encode kernels then run with the piece-start bitmap AS the document bitmap: they record the token position of every
// piece start, and k_docoffs over the piece offsets yields the token range of every piece in the same pass as the ids.
// -------------------------------------------------------------------------------------------------
TKZ_KERNEL(256) void k_piece_index(const uint64_t* startbits, int64_t nwords, int64_t total, int64_t nsub, const int64_t* ord_base,
                                   int64_t n_pieces, int64_t* piece_offs) {
    static_assert(kSub / 64 == 16, "one bitmap word per lane 0..15");
    const int lane = simt::lane();
    const int64_t sub = simt::bid() * (kThreads / 64) + simt::wave();
    if (sub == 0 && lane == 0) piece_offs[n_pieces] = total;
    if (sub >= nsub) return;
    const int64_t w = sub * (kSub / 64) + lane;
    uint64_t m = (lane < kSub / 64 && w < nwords) ? startbits[w] : 0ull;
    const int64_t lim = total - (w << 6);                 // bits at or beyond `total` (the sentinel) are not pieces
    if (lim <= 0) m = 0; else if (lim < 64) m &= tkz_lowmask((int)lim);
    int tot;
    const int pre = tkz_wave_scan<7>(tkz_popc64(m), &tot);
    int64_t o = ord_base[sub] + pre;
    for (; m; m &= m - 1) piece_offs[o++] = (w << 6) + tkz_ctz64(m);
}
TKZ_KERNEL(256) void k_doc_piece(const int64_t* offs, int64_t n_docs, int64_t total, const uint64_t* startbits, const int64_t* ord_base,
                                 int64_t n_pieces, int64_t* doc_piece) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d <= n_docs; d += stride) {
        const int64_t pos = offs[d];
        if (pos >= total) { doc_piece[d] = n_pieces; continue; }
        if (pos < 0) { doc_piece[d] = 0; continue; }
        const int64_t sub = pos / kSub, wpos = pos >> 6;
        int64_t ord = ord_base[sub];
        for (int64_t w = sub * (kSub / 64); w < wpos; ++w) ord += tkz_popc64(startbits[w]);
        ord += tkz_popc64(startbits[wpos] & tkz_lowmask((int)(pos & 63)));
        doc_piece[d] = ord;
    }
}

// -------------------------------------------------------------------------------------------------
// Decode for a batch (TikTokenizer.Decode, TikTokenizer.cs:586-604): id -> bytes through the decoder table, ids that are in
// neither the vocabulary nor the registered special tokens contribute nothing (:591-599), documents concatenated.
//   k_dec_len      per 1024-id tile (one wavefront, 16 ids per lane): byte length of every id -> tile sum and the exclusive
//                  prefix of every 16-id group inside the tile
//   (scan of the tile sums: k_scan_*)
//   k_dec_write    the bytes, staged per tile in LDS and copied out coalesced (direct copies when a tile exceeds the stage)
//   k_dec_docoffs  byte offset of every document = tile base + group prefix + the ids of the group before it
// -------------------------------------------------------------------------------------------------
constexpr int kDecTile = 1024, kDecLane = 16, kDecStage = 12288;
TKZ_DEV bool tkz_dec_find(const TkzDecodeTable& D, int32_t id, uint32_t* off, uint32_t* len) {
    int64_t k;
    if (D.dense) { if (id < 0 || (int64_t)id >= D.n) return false; k = id; }
    else {
        int64_t lo = 0, hi = D.n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (D.ids[mid] < id) lo = mid + 1; else hi = mid; }
        if (lo >= D.n || D.ids[lo] != id) return false;
        k = lo;
    }
    *off = D.off[k]; *len = D.off[k + 1] - D.off[k];
    return *len != 0;
}
struct TkzDecLane { uint32_t off[kDecLane]; uint32_t len[kDecLane]; };
TKZ_DEV int tkz_dec_lane(const TkzDecodeTable& D, const int32_t* ids, int64_t total, int64_t tile, TkzDecLane* L) {
    const int64_t p0 = tile * kDecTile + (int64_t)simt::lane() * kDecLane;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kDecLane; ++k) {
        L->off[k] = 0; L->len[k] = 0;
        if (p0 + k < total) { uint32_t o, n; if (tkz_dec_find(D, ids[p0 + k], &o, &n)) { L->off[k] = o; L->len[k] = n; } }
        sum += (int)L->len[k];
    }
    return sum;
}
TKZ_KERNEL(256) void k_dec_len(TkzDecodeTable D, const int32_t* ids, int64_t total, int64_t ntiles, int32_t* grp_prefix, int32_t* tile_sum) {
    const int64_t tile = simt::bid() * (kThreads / 64) + simt::wave();
    if (tile >= ntiles) return;
    TkzDecLane L;
    const int sum = tkz_dec_lane(D, ids, total, tile, &L);
    int tot;
    const int pre = tkz_wave_scan_sum(sum, &tot);
    grp_prefix[tile * 64 + simt::lane()] = pre;
    if (simt::lane() == 0) tile_sum[tile] = tot;
}
TKZ_KERNEL(256) void k_dec_write(TkzDecodeTable D, const int32_t* ids, int64_t total, int64_t ntiles, const int64_t* tile_base, uint8_t* out, int64_t out_cap) {
    TKZ_SHARED uint8_t s_stage[kThreads / 64][kDecStage];
    const int64_t tile = simt::bid() * (kThreads / 64) + simt::wave();
    if (tile >= ntiles) return;
    TkzDecLane L;
    const int sum = tkz_dec_lane(D, ids, total, tile, &L);
    int tot;
    int pos = tkz_wave_scan_sum(sum, &tot);
    const int64_t base = tile_base[tile];
    if (base + tot > out_cap) return;                       // (the host reports TKZ_E_CAPACITY from the grand total)
    uint8_t* dst = out + base;
    if (tot <= kDecStage) {
        uint8_t* st = s_stage[simt::wave()];
#pragma unroll 1
        for (int k = 0; k < kDecLane; ++k) { const uint8_t* src = D.blob + L.off[k]; for (uint32_t j = 0; j < L.len[k]; ++j) st[pos + (int)j] = src[j]; pos += (int)L.len[k]; }
        (void)simt::ballot(true);      // (the staging area is private to the wavefront: its LDS accesses are ordered, no barrier)
        for (int i = simt::lane(); i < tot; i += 64) dst[i] = st[i];
    } else {
#pragma unroll 1
        for (int k = 0; k < kDecLane; ++k) { const uint8_t* src = D.blob + L.off[k]; for (uint32_t j = 0; j < L.len[k]; ++j) dst[pos + (int)j] = src[j]; pos += (int)L.len[k]; }
    }
}
TKZ_KERNEL(256) void k_dec_docoffs(TkzDecodeTable D, const int32_t* ids, int64_t total, const int64_t* id_offs, int64_t n_docs, const int64_t* tile_base,
                                   const int32_t* grp_prefix, const int64_t* grand, int64_t* byte_offs, int32_t* counters) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d <= n_docs; d += stride) {
        const int64_t p = id_offs[d];
        bool ok = p >= 0 && p <= total;
        if (d == 0) ok = ok && p == 0;
        if (d == n_docs) ok = ok && p == total; else ok = ok && p <= id_offs[d + 1];
        if (!ok) { simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrOffsets); byte_offs[d] = 0; continue; }
        if (p >= total) { byte_offs[d] = *grand; continue; }
        int64_t v = tile_base[p / kDecTile] + grp_prefix[p / kDecLane];
        for (int64_t q = p & ~(int64_t)(kDecLane - 1); q < p; ++q) { uint32_t o, n; if (tkz_dec_find(D, ids[q], &o, &n)) v += n; }
        byte_offs[d] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// synthetic corpus (tkz_corpus.h): lengths, then bytes
// -------------------------------------------------------------------------------------------------
TKZ_KERNEL(256) void k_corpus_lengths(int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int min_len, int max_len, int64_t* offs) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d < n_docs; d += stride)
        offs[d + 1] = tkz_corpus_doc(kind, seed, first_doc + d, min_len, max_len, (uint8_t*)0, 0);
    if (simt::bid() == 0 && simt::tid() == 0) offs[0] = 0;
}
TKZ_KERNEL(256) void k_corpus_fill(int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int min_len, int max_len,
                                   const int64_t* offs, uint8_t* bytes, int64_t cap) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d < n_docs; d += stride) {
        const int64_t a = offs[d], b = offs[d + 1];
        if (b <= cap) tkz_corpus_doc(kind, seed, first_doc + d, min_len, max_len, bytes + a, b - a);
    }
}
// in-place inclusive scan of offs[1..n] by one workgroup (lengths -> offsets); n up to 10^8 is ~100 ms, one-off setup
TKZ_KERNEL(256) void k_offsets_scan(int64_t* offs, int64_t n, int64_t* total) {
    TKZ_SHARED int64_t s_part[kThreads];
    const int tid = simt::tid();
    const int64_t per = (n + kThreads - 1) / kThreads;
    const int64_t lo = 1 + tid * per, hi = lo + per < n + 1 ? lo + per : n + 1;
    int64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += offs[i];
    s_part[tid] = s;
    simt::sync();
    int64_t run = 0;
    for (int t = 0; t < tid; ++t) run += s_part[t];
    for (int64_t i = lo; i < hi; ++i) { run += offs[i]; offs[i] = run; }
    simt::sync();
    if (tid == 0) *total = n ? offs[n] : 0;
}

// -------------------------------------------------------------------------------------------------
// k_small: the WHOLE launch sequence for a small batch in ONE kernel, one workgroup (ITokenizer.Encode of a prompt, TikTokenizer.cs:178-207;
// the shape of BASELINE.json configs[0]).  The batch path above is ~25 launches and as many dependent kernel boundaries: ~160 us for a
// 64-byte prompt, whatever the kernels do.  Here the phases are the same device functions (tkz_probe_subtile, tkz_merge_short_group,
// tkz_merge_long_chunks, tkz_place_subtiles), run one after the other by the four wavefronts of one workgroup with a workgroup
// barrier between them; the text and the offsets are read straight from page-locked host memory and the ids, the document offsets and
// the status go straight back into it: one launch, one stream synchronisation, no copy commands.
//   limits (checked by the host): total <= kSmallMaxBytes (128 sub-tiles), n_docs <= kSmallMaxDocs; o200k, which has no row evaluator and is
//   split by the sequential matcher -- the definition -- one lane per document over text staged in LDS: total <= kSmallMaxBytesO200k and
//   every document <= kSmallMaxDoc bytes
//   a giant piece, a missed piece of more than kLanePiece bytes (k_merge_coop's), a miss list or record buffer that is too small: status != 0, and the
//   host takes the batch path (which has the retries)
// -------------------------------------------------------------------------------------------------
constexpr int kSmallWaves = 16;            // the workgroup is 256 threads for up to 4 sub-tiles, 1024 beyond
constexpr int kSmallLdsQuads = kSmallWaves * kPretokBlkQuads + 16;       // 83.5 KB: the largest phase (a 4 KiB block of the pre-tokenizer per wavefront)
static_assert(kSmallLdsQuads >= kSmallWaves * kProbeLdsQuads + TKZ_SHORT_KEY_MAX + 1 && kSmallLdsQuads >= 8 * kMsLdsQuads + 32 && kSmallLdsQuads >= kLongLdsQuads &&
              kSmallLdsQuads >= kSmallWaves * kPlaceLdsQuads && kSmallLdsQuads * 16 >= kSmallMaxBytesO200k + 64 && kSmallMaxBytes <= 8 * kGroup * kSub,
              "every phase fits the one LDS block; the o200k text too; at most 8 groups of k_merge_short");
TKZ_KERNEL(1024) void k_small(TkzTables T, EncodeParams P, SmallArgs A) {
    static_assert(kSmallLdsQuads * 16 <= kSmallLdsBytesNeeded, "tkz_kernels.h: what the host checks against the device's LDS per workgroup");
    TKZ_SHARED uint4 s_raw[kSmallLdsQuads];
    TKZ_SHARED int s_flag;
    const int tid = simt::tid(), lane = simt::lane(), wave = simt::wave();
    const int kThreads = simt::nthreads(), nwaves = kThreads >> 6;          // (shadows the constant: every stride below is the actual workgroup size)
    const int64_t total = P.total, n_docs = P.n_docs, nwords = P.nwords;
    const int nsub = (int)P.nsub;
    uint8_t* const d_bytes = const_cast<uint8_t*>(P.bytes);
    int64_t* const d_offs = const_cast<int64_t*>(P.offs);
    uint8_t* const s_text = reinterpret_cast<uint8_t*>(s_raw);
    int stamp_i = 4;                                              // h_result[4..]: the shader clock at the end of every phase (thread 0)
    auto stamp = [&]() { if (tid == 0 && stamp_i < 20) A.h_result[stamp_i] = (int64_t)simt::clock(); ++stamp_i; };
    stamp();
    // ---- 0. the input from page-locked host memory into HBM (the later phases gather from it) and into LDS (the matcher reads it char by
    // char); everything the phases accumulate into, zeroed ----
    for (int64_t i = tid; 16 * i < total; i += kThreads) {
        uint4 v = tkz_load16(A.h_bytes + 16 * i);                 // (the host buffer is kSmallMaxBytes + 64 long: whole quads can be read)
        if (16 * i + 16 > total) {                               // bytes beyond the text read as 0
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 16; ++j) if (16 * i + j >= total) w[j >> 2] &= ~(0xFFu << (8 * (j & 3)));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        *reinterpret_cast<uint4*>(d_bytes + 16 * i) = v;
        if (tkz_pat_is_o200k(T.pattern)) s_raw[i] = v;             // (only the sequential matcher reads the text from LDS)
    }
    for (int64_t d = tid; d <= n_docs; d += kThreads) d_offs[d] = A.h_offs[d];
    for (int64_t w = tid; w < nwords + 1; w += kThreads) A.docbits[w] = 0;
    for (int i = tid; i < A.counter_words; i += kThreads) P.counters[i] = 0;
    for (int t = tid; t < nsub; t += kThreads) P.heavy_flag[t] = 0;
    if (tid == 0) s_flag = 0;
    simt::sync();
    stamp();
    // ---- 1. document marks (k_docmark) ----
    for (int64_t d = tid; d <= n_docs; d += kThreads) {
        const int64_t pos = d_offs[d];
        bool ok = pos >= 0 && pos <= total;
        if (d == 0) ok = ok && pos == 0;
        if (d == n_docs) ok = ok && pos == total;
        else ok = ok && pos <= d_offs[d + 1];
        if (!ok) { simt::atomic_or((unsigned*)&P.counters[0], (unsigned)kErrOffsets); continue; }
        simt::atomic_or64((unsigned long long*)&A.docbits[pos >> 6], 1ull << (pos & 63));
    }
    simt::sync();
    for (int64_t w = tid; w < nwords; w += kThreads) A.startbits[w] = A.docbits[w];
    simt::sync();
    stamp();
    // ---- 2. Regex.Matches.  Pattern 1 / cl100k: the row evaluator (one lane per BYTE of a 64-byte row, rows one after the other, scan state
    // carried: tkz_rows_sequential), the rows dealt out to the four wavefronts -- the sequential matcher took 55 us for a 64-byte prompt
    // (thousands of dependent reads), this takes ~2.  o200k has no row evaluator: the matcher, one lane per document, on the text in LDS ----
    if (tkz_pat_is_o200k(T.pattern)) {
        for (int64_t d = tid; d < n_docs; d += kThreads) {
            const int64_t a = d_offs[d], b = d_offs[d + 1];
            if (b <= a || a < 0 || b > total) continue;
            TkzDoc doc; doc.b = s_text + a; doc.n = b - a; doc.bmp = T.bmp_class; doc.by_code_point = T.pattern == TKZ_PAT_O200K;
            int bad = 0;
            for (int64_t p = 0; p < doc.n;) { const TkzChar c = tkz_doc_char(doc, p); bad |= c.bad; p += c.len; }
            if (bad) { simt::atomic_or((unsigned*)&P.counters[0], (unsigned)kErrUtf8); continue; }
            tkz_seq_emit(T.pattern, doc, a, 0, a, b, A.startbits);
        }
    } else {
        simt::sync();                                             // (the text in LDS is not needed on this path: its place take the flag table and the blocks)
        uint16_t* s_aflags = reinterpret_cast<uint16_t*>(s_raw + kSmallWaves * kPretokBlkQuads);
        for (int i = tid; i < 128; i += kThreads) s_aflags[i] = (uint16_t)tkz_ascii_flags((uint32_t)i, T.pattern == TKZ_PAT_CL100K);
        simt::sync();
        for (int64_t blk = wave; blk * kRowsPerWave < nwords; blk += nwaves) {
            if (T.pattern == TKZ_PAT_P1) tkz_pretok_block<TKZ_PAT_P1>(blk, s_raw + wave * kPretokBlkQuads, s_aflags, d_bytes, total, A.docbits, A.startbits, nwords, T.bmp_class, P.counters);
            else tkz_pretok_block<TKZ_PAT_CL100K>(blk, s_raw + wave * kPretokBlkQuads, s_aflags, d_bytes, total, A.docbits, A.startbits, nwords, T.bmp_class, P.counters);
        }
    }
    simt::sync();
    stamp();
    // ---- 3. documents and pieces that start in each sub-tile, their scans (k_doccount + k_scan_*): one lane per sub-tile ----
    if (wave == 0) {
        int dcarry = 0, pcarry = 0;
        for (int q0 = 0; q0 < nsub; q0 += 64) {                   // 64 sub-tiles per round, one per lane
            const int q = q0 + lane;
            int dc = 0, pc = 0;
            if (q < nsub) {
                for (int k = 0; k < kSub / 64; ++k) {
                    const int64_t w = (int64_t)q * (kSub / 64) + k;
                    uint64_t md = w < nwords ? A.docbits[w] : 0ull, mp = w < nwords ? A.startbits[w] : 0ull;
                    const int64_t lim = total - (w << 6);                  // (the sentinel bit at `total` is not a start)
                    if (lim <= 0) { md = 0; mp = 0; } else if (lim < 64) { md &= tkz_lowmask((int)lim); mp &= tkz_lowmask((int)lim); }
                    dc += tkz_popc64(md); pc += tkz_popc64(mp);
                }
            }
            int dtot, ptot;
            const int dpre = dcarry + tkz_wave_scan_sum(dc, &dtot);
            const int ppre = pcarry + tkz_wave_scan_sum((pc + kRecordLine - 1) & ~(kRecordLine - 1), &ptot);
            if (q < nsub) { A.docord_base[q] = dpre; A.pcount[q] = pc; A.pbase[q] = ppre; }
            dcarry += dtot; pcarry += ptot;
        }
        if (lane == 0 && pcarry > P.prank_cap) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)kErrCapacity);
    }
    simt::sync();
    stamp();
    // ---- 4. whole-piece lookups (k_probe) ----
    {
        uint4* s_kmask = s_raw + nwaves * kProbeLdsQuads;
        tkz_probe_kmask_init(s_kmask);
        simt::sync();
        for (int sub = wave; sub < nsub; sub += nwaves) {
            ProbeText tx;
            tkz_probe_request_text(P, sub, &tx);
            tkz_probe_subtile<true>(T, P, sub, tkz_probe_lds(s_raw + wave * kProbeLdsQuads, s_kmask), tx, -1);
            (void)simt::ballot(true);
        }
    }
    simt::sync();
    if (tid == 0) { int f = P.counters[0] != 0; for (int t = 0; t < nsub; ++t) f |= (P.heavy_flag[t] & 2u) != 0; s_flag = f; }
    simt::sync();
    if (s_flag) {                                                 // a giant piece, an error, a buffer that is too small: the batch path's business
        if (tid == 0) { A.h_result[0] = 1; A.h_result[1] = P.counters[0]; }
        return;
    }
    stamp();
    // ---- 5. BytePairEncode of the short misses (k_merge_short: one group of 16 sub-tiles), then of the long ones (k_merge_long: one chunk) ----
    {
        uint16_t* s_brank16 = reinterpret_cast<uint16_t*>(s_raw + 8 * kMsLdsQuads);
        tkz_ms_brank_init(T, s_brank16);
        simt::sync();
        for (int g = wave; g < 8 && g * kGroup < nsub; g += nwaves) tkz_merge_short_group(T, P, g, tkz_ms_lds(s_raw + wave * kMsLdsQuads, s_brank16));
    }
    simt::sync();
    stamp();
    if (wave == 0) {
        const LongLds LD = tkz_long_lds(s_raw);
        tkz_long_brank_init(T, LD.brank);
        if (T.max_rank <= kVarCompactMaxRank) tkz_merge_long_chunks<true>(T, P, 0, 1, LD, kSmallLanePiece);
        else tkz_merge_long_chunks<false>(T, P, 0, 1, LD, kSmallLanePiece);
    }
    simt::sync();
    stamp();
    // ---- 6. scan of the token counts (k_scan_*), ids to their places (k_place), document offsets (k_docoffs) ----
    if (wave == 0) {
        int carry = 0;
        for (int q0 = 0; q0 < nsub; q0 += 64) {
            const int q = q0 + lane;
            int tot;
            const int pre = carry + tkz_wave_scan_sum(q < nsub ? P.tile_count[q] : 0, &tot);
            if (q < nsub) A.tile_base[q] = pre;
            carry += tot;
        }
        if (lane == 0) { A.h_result[2] = carry; A.tile_base[nsub] = carry; }
    }
    simt::sync();
    stamp();
    for (int s0 = wave * kPlacePer; s0 < nsub; s0 += nwaves * kPlacePer)
    {
        if (P.promo) tkz_place_subtiles<64, true, kStageMin>(P, A.tile_base, A.out, A.out_cap, s0, tkz_place_lds<64, kStageMin>(s_raw + wave * kPlaceLdsQuads));
        else tkz_place_subtiles<64, false, kStageMin>(P, A.tile_base, A.out, A.out_cap, s0, tkz_place_lds<64, kStageMin>(s_raw + wave * kPlaceLdsQuads));
    }
    simt::sync();
    stamp();
    {
        const int64_t grand = A.tile_base[nsub];
        for (int64_t d = tid; d <= n_docs; d += kThreads) {
            const int64_t pos = d_offs[d];
            if (pos >= total) { A.out_offs[d] = grand; continue; }
            const int64_t sub = pos / kSub, wpos = pos >> 6;
            int64_t ord = A.docord_base[sub];
            for (int64_t w = sub * (kSub / 64); w < wpos; ++w) ord += tkz_popc64(A.docbits[w]);
            ord += tkz_popc64(A.docbits[wpos] & tkz_lowmask((int)(pos & 63)));
            A.out_offs[d] = A.tile_base[sub] + P.doc_tok[ord];
        }
    }
    simt::sync();
    stamp();
    if (tid == 0) {
        const int32_t err = P.counters[0]; A.h_result[1] = err; A.h_result[0] = err ? 1 : 0;
        // {n_docs, n_bytes, n_tokens} of this batch on the device, as k_counts3 leaves them on the batch path (a batch that is handed back
        // gets them from there)
        if (!err) for (int k = 0; k < 2; ++k) if (A.counts3[k]) { A.counts3[k][0] = n_docs; A.counts3[k][1] = total; A.counts3[k][2] = A.tile_base[nsub]; }
    }
}

// =================================================================================================
// launchers
// =================================================================================================
static inline void hook(const Launch& L, int id, int phase) { if (L.hook) L.hook(L.hook_ctx, id, phase, L.stream); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// a launch needs one workgroup at least (a grid of 0 is an invalid configuration on HIP; the kernels return at once for items beyond their count):
// an empty chunk of a host batch -- a UTF-16 document that spans a chunk cut -- has no tiles at all
static inline int64_t grid1(int64_t g) { return g < 1 ? 1 : g; }
static inline int64_t xcd_grid(int64_t blocks) { return (blocks + 7) & ~(int64_t)7; }     // tkz_xcd_block: XCD x takes the x-th eighth of the blocks
static inline int64_t grid_for(int64_t items) { const int64_t g = cdiv(items, kThreads); return g < 1 ? 1 : (g > 16384 ? 16384 : g); }

void launch_docmark(const Launch& L, const int64_t* d_offs, int64_t n_items, int64_t total, uint64_t* bits, int32_t* counters) {
    hook(L, K_DOCMARK, 0);
    TKZ_LAUNCH(k_docmark, grid_for(n_items + 1), kThreads, L.stream, d_offs, n_items, total, bits, counters);
    hook(L, K_DOCMARK, 1);
}
void launch_pretok_rows(const Launch& L, int pattern, const uint8_t* d_bytes, const int64_t* d_offs, int64_t n_docs, int64_t total,
                        const uint64_t* docbits, uint64_t* startbits, int64_t nrows, const uint8_t* bmp, int32_t* counters,
                        int64_t* xq, unsigned long long* xcount) {
    const int64_t grid = cdiv(nrows, kRowsPerWave);           // one 64-lane workgroup per chunk of rows
    hook(L, K_PRETOK, 0);
    if (pattern == TKZ_PAT_P1)
        TKZ_LAUNCH(k_pretok_rows<TKZ_PAT_P1>, grid, 64, L.stream, d_bytes, total, docbits, startbits, nrows, bmp, counters, xq, xcount);
    else if (pattern == TKZ_PAT_CL100K)
        TKZ_LAUNCH(k_pretok_rows<TKZ_PAT_CL100K>, grid, 64, L.stream, d_bytes, total, docbits, startbits, nrows, bmp, counters, xq, xcount);
    else {
        TKZ_LAUNCH(k_pretok_rows<TKZ_PAT_O200K>, grid, 64, L.stream, d_bytes, total, docbits, startbits, nrows, bmp, counters, xq, xcount);
        // blocks with multi-byte chars: the char-level block evaluator; what that refuses as well: the sequential matcher
        int64_t* xq2 = xq + grid + 1;
        unsigned long long* xcount2 = xcount + 1;
        TKZ_LAUNCH(k_pretok_mb_blocks, grid < 8192 ? grid : 8192, 64, L.stream, d_bytes, total, docbits, startbits, nrows, bmp, (const uint8_t*)xq, grid,
                   xcount, xq2, xcount2, (int)(pattern == TKZ_PAT_O200K));
        TKZ_LAUNCH(k_pretok_seq_blocks, grid_for(n_docs > grid ? n_docs : grid), kThreads, L.stream, d_bytes, d_offs, n_docs, total, startbits, nrows,
                   pattern, bmp, (const int64_t*)xq2, (const unsigned long long*)xcount2, counters);
    }
    hook(L, K_PRETOK, 1);
}
void launch_pretok_seq(const Launch& L, int pattern, const uint8_t* d_bytes, const int64_t* d_offs, int64_t n_docs, int64_t total,
                       uint64_t* startbits, const uint8_t* bmp, int32_t* counters) {
    hook(L, K_PRETOK, 0);
    TKZ_LAUNCH(k_pretok_seq, grid_for(n_docs), kThreads, L.stream, d_bytes, d_offs, n_docs, total, startbits, pattern, bmp, counters);
    hook(L, K_PRETOK, 1);
}
void launch_encode(const Launch& L, const TkzTables& T, const EncodeParams& P, int64_t nsub) {
    hook(L, K_ENCODE, 0);
    TKZ_LAUNCH(k_probe, xcd_grid(cdiv(nsub, (kThreads / 64) * kProbePer)), kThreads, L.stream, T, P);
    hook(L, K_ENCODE, 1);
    // (the list statistics and, in the same pass, the giant pieces of the sub-tiles k_probe flagged: queued for k_giant_order / k_giant_merge)
    { const int64_t g = grid_for(nsub); TKZ_LAUNCH(k_list_stats, g < 1024 ? g : 1024, kThreads, L.stream, (const uint32_t*)P.mcount, nsub, P.mcap, P.counters,
                                                     (const uint8_t*)P.heavy_flag, P.startbits, P.nwords, P.total, P.giant_q, P.giant_count, P.giant_cap,
                                                     (const uint32_t*)P.mlist, P.coop_q, P.coop_count, P.coop_cap, (int)P.lane_piece, P.miss_sums); }
    // The short misses and the long ones touch different list entries and the sub-tiles' token counts are summed with atomics from zero (P.tc_atomic), so the
    // kernels of the two kinds may run side by side.  What makes that worth having: k_merge_long_q and k_merge_coop last as long as their slowest wavefronts, not as
    // long as their work (0.43 + 0.31 of the 3.7 ms of a 268 MB batch of real text, for 0.6 M pieces), while k_merge_short keeps the whole chip busy.
    // How: this chip starts no workgroup of a second kernel while a first one still has workgroups waiting (tools/stream_overlap_probe.hip: 8,192 + 64 workgroups on two
    // streams take 20 + 5 ms whatever the streams' priorities; 1,024 + 64 take 5) -- so the two tail kernels go FIRST, each on a stream of its own with a grid that fits
    // the chip beside the other, and k_merge_short behind them on L.stream takes what is left and, as their wavefronts retire, everything.
    // The counting, the scan and the scatter of the class queue stay in front on L.stream (they are short and the queue kernel needs them).
    // (k_merge_coop ALONE beside k_merge_short, for workspaces whose class queue is long, was measured too: mixed text 15.97 -> 15.87 ms, headline 20.59 -> 20.84 -- not kept.)
    const bool fork = L.side && L.side2 && L.ev_fork && L.ev_join && L.ev_join2 && P.tc_atomic && P.latency == 0 && P.lq != nullptr;
        // giant pieces (queued by k_list_stats): ordered, merged; then the pieces of 17..1024 bytes and the giants' token counts
#ifdef TKZ_HOSTEMU
    constexpr int kGiantGrid = 2;       // (the CPU emulator pays for every thread of an idle workgroup)
#else
    constexpr int kGiantGrid = 256;
#endif
    const bool latency = P.latency != 0 || !P.lq;
    auto coop = [&](hipStream_t st, int64_t cap) {
        // the pieces k_merge_long leaves to a whole wavefront, off the queue k_list_stats filled (every wavefront exits at once when it is empty).  A small batch
        // gets as many wavefronts as a large one: 16 of them took 107 us over the queue of a 1 MB call
        const int64_t cgrid = latency ? 512 : cdiv(nsub, 64);
        TKZ_LAUNCH(k_merge_coop, cgrid < cap ? cgrid : cap, 64, st, T, P);
    };
    // (a small batch whose token counts are summed with atomics: the three merge stages as one launch, behind the giant pieces -- k_merge_latency)
    const bool fused = !fork && P.latency != 0 && P.tc_atomic != 0 && P.lq != nullptr;
    if (fused) {
        hook(L, K_HEAVY, 0);
        TKZ_LAUNCH(k_giant_order, 1, 1024, L.stream, P);
        TKZ_LAUNCH(k_giant_merge, kGiantGrid, 1024, L.stream, T, P);
        hook(L, K_HEAVY, 1);
        const int64_t nb_short = xcd_grid(cdiv(nsub, (kMsThreads / 64) * kGroup)), chunks = cdiv(nsub, 64) * kLongPartsLatency, nb_long = cdiv(chunks < 65536 ? chunks : 65536, kMsThreads / 64);
        hook(L, K_MERGE_SHORT, 0);
        if (T.max_rank <= kVarCompactMaxRank) TKZ_LAUNCH((k_merge_latency<true>), nb_short + nb_long, kMsThreads, L.stream, T, P, (int)nb_short, (int)nb_long);
        else TKZ_LAUNCH((k_merge_latency<false>), nb_short + nb_long, kMsThreads, L.stream, T, P, (int)nb_short, (int)nb_long);
        // (the queue of the 129+-byte pieces stays a launch of its own BEHIND the chunk form of the long lists, as it always was: inside the same launch its pieces came
        //  out wrong on the GPU -- test_pieces_vs_oracle_bpe, test_mid_pieces_share_the_arena -- while the emulator, whose workgroups take turns, saw nothing)
        coop(L.stream, kCoopGrid);
        hook(L, K_MERGE_SHORT, 1);
        return;
    }
    if (!fork) {                   // (the serial form keeps k_merge_short in front, as it always was)
        hook(L, K_MERGE_SHORT, 0);
        TKZ_LAUNCH(k_merge_short, xcd_grid(cdiv(nsub, (kMsThreads / 64) * kGroup)), kMsThreads, L.stream, T, P);
        hook(L, K_MERGE_SHORT, 1);
    }
    hook(L, K_HEAVY, 0);
    TKZ_LAUNCH(k_giant_order, 1, 1024, L.stream, P);
    TKZ_LAUNCH(k_giant_merge, kGiantGrid, 1024, L.stream, T, P);   // takes pieces off the ordered queue; exits at once when it is empty
    if (!latency) {         // the queue form: the batch's long misses binned by length class ...
        const int64_t nchunks = cdiv(nsub, 64), g4 = cdiv(nchunks, 4);
        TKZ_LAUNCH(k_long_count, g4 < 4096 ? g4 : 4096, kThreads, L.stream, P);
        launch_scan2(L, nchunks * kLenClasses, P.lq_bsum, P.lq_cnt, P.lq_base, P.lq_total, 1, nullptr, nullptr, nullptr, 1, -1);
        TKZ_LAUNCH(k_long_scatter, g4 < 4096 ? g4 : 4096, kThreads, L.stream, P);
    }
    hipStream_t sq = L.stream;                           // the stream of the queue kernel
    int64_t qcap = kLongQGrid;
    if (fork) {
        hook(L, K_HEAVY, 1);            // (the bracket of this form: what runs in front of the three; K_MERGE_SHORT's is the three side by side)
        hook(L, K_MERGE_SHORT, 0);
        (void)hipEventRecord(L.ev_fork, L.stream); (void)hipStreamWaitEvent(L.side, L.ev_fork, 0); (void)hipStreamWaitEvent(L.side2, L.ev_fork, 0);
        sq = L.side; qcap = L.side_long_grid;
    }
    if (latency) {          // the chunk form: strides over units of 4 sub-tiles
        const int64_t chunks = cdiv(nsub, 64) * kLongPartsLatency, grid = chunks < 65536 ? chunks : 65536;
        if (T.max_rank <= kVarCompactMaxRank) TKZ_LAUNCH((k_merge_long<true, true>), grid, 64, sq, T, P);
        else TKZ_LAUNCH((k_merge_long<false, true>), grid, 64, sq, T, P);
    } else {                // ... and merged off the queue 256 at a time
        const int64_t ranges = cdiv(P.lq_cap, kLqRangeLong), grid = ranges < qcap ? (ranges < 1 ? 1 : ranges) : qcap;
        if (T.max_rank <= kVarCompactMaxRank) TKZ_LAUNCH((k_merge_long_q<true>), grid, 64, sq, T, P);
        else TKZ_LAUNCH((k_merge_long_q<false>), grid, 64, sq, T, P);
    }
    if (fork) {
        coop(L.side2, L.side_coop_grid);
        (void)hipEventRecord(L.ev_join, L.side); (void)hipEventRecord(L.ev_join2, L.side2);
        TKZ_LAUNCH(k_merge_short, xcd_grid(cdiv(nsub, (kMsThreads / 64) * kGroup)), kMsThreads, L.stream, T, P);
        (void)hipStreamWaitEvent(L.stream, L.ev_join, 0); (void)hipStreamWaitEvent(L.stream, L.ev_join2, 0);
        hook(L, K_MERGE_SHORT, 1);
    } else {
        coop(L.stream, kCoopGrid);
        hook(L, K_HEAVY, 1);
    }
}
// k_probe and the list statistics over the first `nsample` sub-tiles only (tkz_api.cpp: the sizing attempt of a fresh workspace)
void launch_probe_sample(const Launch& L, const TkzTables& T, const EncodeParams& P, int64_t nsample) {
    if (nsample > P.nsub) nsample = P.nsub;
    hook(L, K_ENCODE, 0);
    TKZ_LAUNCH(k_probe, xcd_grid(cdiv(nsample, (kThreads / 64) * kProbePer)), kThreads, L.stream, T, P);
    hook(L, K_ENCODE, 1);
    { const int64_t g = grid_for(nsample); TKZ_LAUNCH(k_list_stats, g < 1024 ? g : 1024, kThreads, L.stream, (const uint32_t*)P.mcount, nsample, P.mcap, P.counters,
                                                     (const uint8_t*)P.heavy_flag, P.startbits, P.nwords, P.total, P.giant_q, P.giant_count, P.giant_cap,
                                                     (const uint32_t*)P.mlist, P.coop_q, P.coop_count, P.coop_cap, (int)P.lane_piece, P.miss_sums); }
}
void launch_ingest(const Launch& L, const uint8_t* h_bytes, int64_t total, uint8_t* d_bytes, const int64_t* h_offs, int64_t n_offs, int64_t* d_offs, void* zero, int64_t zero_bytes) {
    const int64_t quads = std::max<int64_t>(total / 16 + 1, (zero_bytes + 15) / 16);
    const int64_t g = cdiv(quads, kThreads * 4);               // (four quads a thread: ~64 loads over PCIe in flight per wavefront is what fills the link)
    TKZ_LAUNCH(k_ingest, g < 1 ? 1 : (g > 2048 ? 2048 : g), kThreads, L.stream, h_bytes, total, d_bytes, h_offs, n_offs, d_offs, reinterpret_cast<uint4*>(zero), (zero_bytes + 15) / 16);
}
void launch_small(const Launch& L, const TkzTables& T, const EncodeParams& P, const SmallArgs& A) {
    TKZ_LAUNCH(k_small, 1, P.nsub <= 4 ? 256 : 1024, L.stream, T, P, A);
}
void launch_place(const Launch& L, const EncodeParams& P, const int64_t* tile_base, int64_t nsub, int32_t* out, int64_t out_cap) {
    hook(L, K_GATHER, 0);
    const int64_t grid = xcd_grid(cdiv(nsub, (kThreads / 64) * kPlacePer));
    if (P.promo) {      // (promoted pieces in the tables: the records may carry promo codes)
        if (P.place128) TKZ_LAUNCH((k_place<128, true>), grid, kThreads, L.stream, P, tile_base, out, out_cap);
        else TKZ_LAUNCH((k_place<64, true>), grid, kThreads, L.stream, P, tile_base, out, out_cap);
    } else {
        if (P.place128) TKZ_LAUNCH((k_place<128, false>), grid, kThreads, L.stream, P, tile_base, out, out_cap);
        else TKZ_LAUNCH((k_place<64, false>), grid, kThreads, L.stream, P, tile_base, out, out_cap);
    }
    hook(L, K_GATHER, 1);
}
void launch_doccount(const Launch& L, const uint64_t* docbits, int64_t nwords, int64_t total, int64_t nsub, int32_t* cnt) {
    TKZ_LAUNCH(k_doccount, grid_for(nsub * (kSub / 64)), kThreads, L.stream, docbits, nwords, total, nsub, cnt);
}
void launch_doccount2(const Launch& L, const uint64_t* bits_a, const uint64_t* bits_b, int64_t nwords, int64_t total, int64_t nsub, int32_t* cnt_a, int32_t* cnt_b) {
    TKZ_LAUNCH(k_doccount2, grid_for(nsub * (kSub / 64)), kThreads, L.stream, bits_a, bits_b, nwords, total, nsub, cnt_a, cnt_b);
}
void launch_scan2(const Launch& L, int64_t ntiles, int64_t* bsum, const int32_t* cnt_a, int64_t* base_a, int64_t* grand_a, int round_to_a,
                  const int32_t* cnt_b, int64_t* base_b, int64_t* grand_b, int round_to_b, int kid) {
    if (ntiles <= kScanSmallMax) {
        if (kid >= 0) hook(L, kid, 0);
        TKZ_LAUNCH(k_scan_small, 1, 1024, L.stream, cnt_a, base_a, grand_a, round_to_a > 1 ? round_to_a - 1 : 0, cnt_b, base_b, grand_b, round_to_b > 1 ? round_to_b - 1 : 0, (int)ntiles);
        if (kid >= 0) hook(L, kid, 1);
        return;
    }
    launch_scan(L, cnt_a, ntiles, bsum, base_a, grand_a, kid, round_to_a);
    if (cnt_b) launch_scan(L, cnt_b, ntiles, bsum, base_b, grand_b, kid, round_to_b);
}
void launch_scan(const Launch& L, const int32_t* tile_count, int64_t ntiles, int64_t* bsum, int64_t* tile_base, int64_t* grand, int kid, int round_to) {
    const int64_t nblk = grid1(cdiv(ntiles, kScanBlock));     // (no tiles: one workgroup that writes a zero sum)
    const int round = round_to > 1 ? round_to - 1 : 0;       // (round_to: a power of two)
    if (kid >= 0) hook(L, kid, 0);
    TKZ_LAUNCH(k_scan_partials, nblk, kThreads, L.stream, tile_count, ntiles, bsum, round);
    TKZ_LAUNCH(k_scan_top, 1, kThreads, L.stream, bsum, nblk, grand);
    TKZ_LAUNCH(k_scan_final, nblk, kThreads, L.stream, tile_count, ntiles, (const int64_t*)bsum, tile_base, round);
    if (kid >= 0) hook(L, kid, 1);
}
void launch_docoffs(const Launch& L, const int64_t* d_offs, int64_t n_docs, int64_t total, const int64_t* tile_base,
                    const uint64_t* docbits, const int64_t* docord_base, const int32_t* doc_tok, const int64_t* grand, int64_t* out_offs,
                    int64_t c3_docs, int64_t* c3a, int64_t* c3b, int64_t* c3c) {
    hook(L, K_DOCOFFS, 0);
    TKZ_LAUNCH(k_docoffs, grid_for(n_docs + 1), kThreads, L.stream, d_offs, n_docs, total, tile_base, docbits, docord_base, doc_tok, grand, out_offs, c3_docs, c3a, c3b, c3c);
    hook(L, K_DOCOFFS, 1);
}
void launch_case_equiv_fix(const Launch& L, const uint8_t* d_bytes, int64_t total, const uint64_t* docbits, uint64_t* startbits) {
    TKZ_LAUNCH(k_case_equiv_fix, grid_for((total + 7) / 8), kThreads, L.stream, d_bytes, total, docbits, startbits);
}
void launch_rebase(const Launch& L, int64_t* offs, int64_t n, int64_t base) {
    TKZ_LAUNCH(k_rebase, grid_for(n), kThreads, L.stream, offs, n, base);
}
void launch_miss_stats(const Launch& L, const EncodeParams& P, int64_t nsub) {
    const int64_t g = grid_for(nsub);
    TKZ_LAUNCH(k_miss_stats, g < 1024 ? g : 1024, kThreads, L.stream, (const uint32_t*)P.mcount, P.pcount, nsub, P.stats);
}
void launch_counts3(const Launch& L, int64_t n_docs, int64_t total, const int64_t* grand, int64_t* out3, int64_t* out3b, int64_t* out3c) {
    TKZ_LAUNCH(k_counts3, 1, 64, L.stream, n_docs, total, grand, out3, out3b, out3c);
}
void launch_u16_len(const Launch& L, const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t ntiles, int32_t* grp_prefix, int32_t* tile_sum) {
    TKZ_LAUNCH(k_u16_len, grid1(cdiv(ntiles, kThreads / 64)), kThreads, L.stream, units, total, docbits, ntiles, grp_prefix, tile_sum);
}
void launch_u16_write(const Launch& L, const uint16_t* units, int64_t total, const uint64_t* docbits, int64_t ntiles, const int64_t* tile_base,
                      uint8_t* out, const int64_t* unit_offs, int64_t n_docs, const int32_t* grp_prefix, const int64_t* grand, int64_t* byte_offs) {
    TKZ_LAUNCH(k_u16_write, grid1(cdiv(ntiles, kThreads / 64)), kThreads, L.stream, units, total, docbits, ntiles, tile_base, out);
    TKZ_LAUNCH(k_u16_docoffs, grid_for(n_docs + 1), kThreads, L.stream, units, total, docbits, unit_offs, n_docs, tile_base, grp_prefix, grand, byte_offs);
}
int64_t u16_tiles(int64_t total_units) { return cdiv(total_units, kU16Tile); }
void launch_piece_index(const Launch& L, const uint64_t* startbits, int64_t nwords, int64_t total, int64_t nsub, const int64_t* ord_base,
                        int64_t n_pieces, int64_t* piece_offs, const int64_t* d_offs, int64_t n_docs, int64_t* doc_piece) {
    TKZ_LAUNCH(k_piece_index, grid1(cdiv(nsub, kThreads / 64)), kThreads, L.stream, startbits, nwords, total, nsub, ord_base, n_pieces, piece_offs);
    TKZ_LAUNCH(k_doc_piece, grid_for(n_docs + 1), kThreads, L.stream, d_offs, n_docs, total, startbits, ord_base, n_pieces, doc_piece);
}
int64_t dec_tiles(int64_t total_ids) { return cdiv(total_ids, kDecTile); }
void launch_dec_len(const Launch& L, const TkzDecodeTable& D, const int32_t* ids, int64_t total, int64_t ntiles, int32_t* grp_prefix, int32_t* tile_sum) {
    TKZ_LAUNCH(k_dec_len, grid1(cdiv(ntiles, kThreads / 64)), kThreads, L.stream, D, ids, total, ntiles, grp_prefix, tile_sum);
}
void launch_dec_write(const Launch& L, const TkzDecodeTable& D, const int32_t* ids, int64_t total, int64_t ntiles, const int64_t* tile_base, uint8_t* out,
                      int64_t out_cap, const int64_t* id_offs, int64_t n_docs, const int32_t* grp_prefix, const int64_t* grand, int64_t* byte_offs, int32_t* counters) {
    TKZ_LAUNCH(k_dec_write, grid1(cdiv(ntiles, kThreads / 64)), kThreads, L.stream, D, ids, total, ntiles, tile_base, out, out_cap);
    TKZ_LAUNCH(k_dec_docoffs, grid_for(n_docs + 1), kThreads, L.stream, D, ids, total, id_offs, n_docs, tile_base, grp_prefix, grand, byte_offs, counters);
}
void launch_corpus(hipStream_t s, int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int min_len, int max_len,
                   int64_t* d_offs, uint8_t* d_bytes, int64_t cap_bytes, int64_t* d_total) {
    TKZ_LAUNCH(k_corpus_lengths, grid_for(n_docs), kThreads, s, kind, seed, first_doc, n_docs, min_len, max_len, d_offs);
    TKZ_LAUNCH(k_offsets_scan, 1, kThreads, s, d_offs, n_docs, d_total);
    TKZ_LAUNCH(k_corpus_fill, grid_for(n_docs), kThreads, s, kind, seed, first_doc, n_docs, min_len, max_len,
               (const int64_t*)d_offs, d_bytes, cap_bytes);
}

}  // namespace tkz
