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
//   k_encode_tiles   per 4 KiB tile: enumerate pieces from the bitmap, whole-piece lookup
//                    (TikTokenizer.cs:262), BytePairEncode on a miss (BytePairEncoder.cs:13-76), tokens
//                    written densely into the tile's own span of `tmp`; per-document token positions
//   k_scan_*         exclusive scan of the per-tile token counts
//   k_gather         tmp -> out_ids at the tile's final offset (coalesced copy)
//   k_docoffs        out_offsets[d] = tile base + position inside the tile
#include "tkz_kernels.h"

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
        simt::atomic_or64((unsigned long long*)&bits[pos >> 6], 1ull << (pos & 63));
    }
}

// -------------------------------------------------------------------------------------------------
// k_pretok_rows : one wave per kRowsPerWave rows of 64 bytes
// -------------------------------------------------------------------------------------------------
template <int PATTERN>
TKZ_KERNEL(256) void k_pretok_rows(const uint8_t* bytes, int64_t total, const uint64_t* docbits, uint64_t* startbits,
                                   int64_t nrows, const uint8_t* bmp, int32_t* counters) {
    const int64_t wave_id = simt::bid() * (simt::nthreads() >> 6) + simt::wave();
    const int64_t r0 = wave_id * kRowsPerWave;
    if (r0 >= nrows) return;                               // whole wave leaves together
    const int64_t r1 = r0 + kRowsPerWave < nrows ? r0 + kRowsPerWave : nrows;
    const int lane = simt::lane();

    // warm-up start: the nearest row boundary at or before r0 across which no scan state flows
    // (the byte before it is neither a digit nor CR/LF, or a document starts exactly there)
    int64_t rw = r0;
    while (rw > 0) {
        if (docbits[rw] & 1ull) break;
        const int pc = tkz_classify_byte(bytes, total, (rw << 6) - 1, bmp).pc;
        if (pc != PC_N && pc != PC_CRLF) break;
        --rw;
    }
    TkzScanCarry cy;
    cy.nb63 = 0; cy.carryN = 0; cy.abs63 = 0; cy.sa_from = -1; cy.sa_end = -1; cy.sa_lastcr = -1;

    int64_t row = rw > 0 ? rw - 1 : 0;                     // one extra row so the per-lane flags of "previous" are real
    TkzRowLane P = tkz_classify_byte(bytes, total, ((row - 1) << 6) + lane, bmp);
    TkzRowLane C = tkz_classify_byte(bytes, total, (row << 6) + lane, bmp);
    TkzRowMasks mC = tkz_row_masks(C);
    uint64_t dsP = row >= 1 ? docbits[row - 1] : 0, dsC = docbits[row];
    int clenP = 0, o1msP = 0, bad = 0;
    for (; row < r1; ++row) {
        const TkzRowLane N = tkz_classify_byte(bytes, total, ((row + 1) << 6) + lane, bmp);
        const TkzRowMasks mN = tkz_row_masks(N);
        const uint64_t dsN = row + 1 < nrows ? docbits[row + 1] : 0;
        int clenC, o1msC;
        const uint64_t out = tkz_row_eval<PATTERN>(P, C, N, dsP, dsC, dsN, mC, mN, clenP, o1msP, &clenC, &o1msC, cy,
                                                   bytes, total, docbits, nrows, bmp, row);
        if (row >= r0) {
            if (lane == 0) startbits[row] = out;
            bad |= C.bad | ((C.off != 0 && ((dsC >> lane) & 1ull)) ? 1 : 0);   // a document may not start inside a char
        }
        P = C; C = N; mC = mN; dsP = dsC; dsC = dsN; clenP = clenC; o1msP = o1msC;
    }
    if (simt::ballot(bad != 0) && lane == 0) simt::atomic_or((unsigned*)&counters[0], (unsigned)kErrUtf8);
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
        TkzDoc doc; doc.b = bytes + a; doc.n = b - a; doc.bmp = bmp;
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
// k_encode_tiles
// -------------------------------------------------------------------------------------------------
TKZ_DEV int64_t tkz_lower_bound(const int64_t* a, int64_t lo, int64_t hi, int64_t v) {   // first i in [lo,hi) with a[i] >= v
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
// every document that starts at byte `pos` begins at token `val` of its tile
TKZ_DEV void tkz_mark_docs(const int64_t* offs, int64_t d0, int64_t d1, int64_t pos, int32_t* doc_local, int32_t val) {
    for (int64_t d = tkz_lower_bound(offs, d0, d1, pos); d < d1 && offs[d] == pos; ++d) doc_local[d] = val;
}

TKZ_KERNEL(256) void k_encode_tiles(TkzTables T, EncodeParams P) {
    TKZ_SHARED uint32_t s_bytes[(kTile + kHalo) / 4];
    TKZ_SHARED uint64_t s_bits[kTile / 64];
    TKZ_SHARED uint64_t s_docb[kTile / 64];
    TKZ_SHARED uint16_t s_pstart[kTile + 2];
    TKZ_SHARED uint32_t s_longmask[kTile / 32];
    TKZ_SHARED uint32_t s_scr[32 * kThreads];            // short: ids[16][256] pr[16][256]; long: 4 arrays of kLdsLong
    TKZ_SHARED int s_np, s_nlong, s_i0, s_i1;
    TKZ_SHARED int64_t s_last_end, s_d0, s_d1, s_l0;

    const int tid = simt::tid();
    const int64_t tile = simt::bid();
    const int64_t base = tile * kTile;
    const int nb = (int)(P.total - base < kTile ? P.total - base : kTile);
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(s_bytes);

    // ---- stage the tile (+ halo) in LDS: 16 B per lane, coalesced ----
    for (int i = tid; i < (kTile + kHalo) / 16; i += kThreads) {
        const int64_t pos = base + 16 * (int64_t)i;
        uint4 v;
        if (pos + 16 <= P.total) v = tkz_load16(P.bytes + pos);
        else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; ++j) if (pos + j < P.total) w[j >> 2] |= (uint32_t)P.bytes[pos + j] << (8 * (j & 3));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        s_bytes[4 * i + 0] = v.x; s_bytes[4 * i + 1] = v.y; s_bytes[4 * i + 2] = v.z; s_bytes[4 * i + 3] = v.w;
    }
    if (tid < kTile / 64) {
        const int64_t w = tile * (kTile / 64) + tid;
        uint64_t sbit = w < P.nwords ? P.startbits[w] : 0, dbit = w < P.nwords ? P.docbits[w] : 0;
        const int lim = nb - tid * 64;                       // bits at or beyond the end of the corpus are not pieces
        if (lim <= 0) sbit = 0; else if (lim < 64) sbit &= tkz_lowmask(lim);
        s_bits[tid] = sbit; s_docb[tid] = dbit;
    }
    if (tid < kTile / 32) s_longmask[tid] = 0;
    if (tid == 0) { s_nlong = 0; s_d0 = tkz_lower_bound(P.offs, 0, P.n_docs + 1, base); }
    if (tid == 64) s_d1 = tkz_lower_bound(P.offs, 0, P.n_docs + 1, base + nb);

    // ---- end of the last piece that starts in this tile: first piece start at or after base+nb ----
    {
        const int64_t from = base + nb, w0 = from >> 6;
        int64_t found = -1;
        for (int64_t c = 0; found < 0; ++c) {
            const int64_t w = w0 + c * kThreads + tid;
            uint64_t v = w < P.nwords ? P.startbits[w] : 0;
            if (w == w0) v &= ~tkz_lowmask((int)(from & 63));
            uint64_t cand = v ? (uint64_t)((w << 6) + tkz_ctz64(v)) : ~0ull;
            cand = tkz_block_min64(cand);
            if (cand != ~0ull) found = (int64_t)cand;
            else if (w0 + (c + 1) * kThreads >= P.nwords) found = P.total;   // cannot happen: the sentinel bit sits at `total`
        }
        if (tid == 0) s_last_end = found;
    }
    simt::sync();

    // ---- enumerate the piece starts of the tile (order-preserving compaction) ----
    {
        const uint32_t bits16 = (uint32_t)(s_bits[tid >> 2] >> (16 * (tid & 3))) & 0xFFFFu;
        int np;
        int off = tkz_block_scan(tkz_popc32(bits16), &np);
        for (uint32_t b = bits16; b; b &= b - 1) s_pstart[off++] = (uint16_t)(16 * tid + tkz_ctz32(b));
        if (tid == 0) s_np = np;
    }
    simt::sync();
    const int np = s_np;
    const int64_t last_end_rel = s_last_end - base;
    const int64_t first_abs = np ? base + s_pstart[0] : base;
    for (int k = tid; k < np; k += kThreads) {
        const int64_t e = k + 1 < np ? (int64_t)s_pstart[k + 1] : last_end_rel;
        if (e - s_pstart[k] > kShortMax) { simt::atomic_or((unsigned*)&s_longmask[k >> 5], 1u << (k & 31)); simt::atomic_add(&s_nlong, 1); }
    }
    simt::sync();
    int nlong = s_nlong;
    const int64_t d0 = s_d0, d1 = s_d1;

    int running = 0;                                      // tokens of this tile so far (uniform)
    int err = 0;
    int k0 = 0;
    while (k0 < np) {
        // next long piece at or after k0
        int next_long = np;
        if (nlong > 0) {
            int w = k0 >> 5;
            uint32_t m = s_longmask[w] & (0xFFFFFFFFu << (k0 & 31));
            while (!m && ++w < (np + 31) / 32) m = s_longmask[w];
            if (m) next_long = w * 32 + tkz_ctz32(m);
        }
        if (next_long > k0) {
            // ---------------- a batch of up to 256 short pieces, one per lane ----------------
            const int kend = k0 + kThreads < next_long ? k0 + kThreads : next_long;
            const int k = k0 + tid;
            const bool active = k < kend;
            int cnt = 0; int32_t tok0 = 0; uint32_t alive = 0; int s = 0;
            uint32_t* ids = &s_scr[tid];
            uint32_t* pr = &s_scr[16 * kThreads + tid];
            if (active) {
                s = s_pstart[k];
                const int len = (int)((k + 1 < np ? (int64_t)s_pstart[k + 1] : last_end_rel) - s);
                auto at = [&](int i) -> uint32_t { return sb[s + i]; };
                int32_t rank;
                if (len <= TKZ_SHORT_KEY_MAX) {
                    const int w = s >> 2, sh = (s & 3) * 8;
                    const uint64_t a01 = ((uint64_t)s_bytes[w + 1] << 32) | s_bytes[w];
                    const uint64_t a12 = ((uint64_t)s_bytes[w + 2] << 32) | s_bytes[w + 1];
                    const uint64_t a23 = ((uint64_t)s_bytes[w + 3] << 32) | s_bytes[w + 2];
                    uint32_t q0 = (uint32_t)(a01 >> sh), q1 = (uint32_t)(a12 >> sh), q2 = (uint32_t)(a23 >> sh);
                    if (len < 4) { q0 &= (1u << (8 * len)) - 1u; q1 = 0; q2 = 0; }
                    else if (len < 8) { q1 &= (len == 4) ? 0u : ((1u << (8 * (len - 4))) - 1u); q2 = 0; }
                    else if (len < 12) { q2 &= (len == 8) ? 0u : ((1u << (8 * (len - 8))) - 1u); }
                    rank = tkz_lookup_short(T, q0, q1, q2, (uint32_t)len);
                } else {
                    rank = tkz_lookup_long(T, at, (uint32_t)len);
                }
                if (rank != TKZ_RANK_NONE) { cnt = 1; tok0 = rank; }                       // TikTokenizer.cs:262-265
                else cnt = tkz_bpe_short(T, at, len, ids, pr, kThreads, &alive, &err);     // TikTokenizer.cs:268
            }
            int tot;
            const int pre = tkz_block_scan(cnt, &tot);
            if (active) {
                int32_t* dst = P.tmp + first_abs + running + pre;
                if (alive == 0) dst[0] = tok0;
                else { int i = 0; for (uint32_t a = alive; a; a &= a - 1) dst[i++] = (int32_t)ids[tkz_ctz32(a) * kThreads]; }
                if ((s_docb[s >> 6] >> (s & 63)) & 1ull) tkz_mark_docs(P.offs, d0, d1, base + s, P.doc_local, running + pre);
            }
            running += tot;
            k0 = kend;
        } else {
            // ---------------- one long piece, the whole workgroup ----------------
            const int s = s_pstart[k0];
            const int64_t len64 = (k0 + 1 < np ? (int64_t)s_pstart[k0 + 1] : last_end_rel) - s;
            const int64_t abs0 = base + s;
            const uint8_t* gb = P.bytes + abs0;
            auto at = [&](int i) -> uint32_t { return gb[i]; };
            int32_t* dst = P.tmp + first_abs + running;
            int cnt = 0;
            if (len64 > kMaxPiece) { err |= kErrTooLong; }
            else {
                const int len = (int)len64;
                if (tid == 0) {
                    s_i0 = tkz_lookup_long(T, at, (uint32_t)len);
                    if ((s_docb[s >> 6] >> (s & 63)) & 1ull) tkz_mark_docs(P.offs, d0, d1, abs0, P.doc_local, running);
                }
                simt::sync();
                const int32_t whole = s_i0;
                if (whole != TKZ_RANK_NONE) { if (tid == 0) dst[0] = whole; cnt = 1; }
                else {
                    int32_t* arr = reinterpret_cast<int32_t*>(s_scr);
                    int stride = kLdsLong;
                    bool ok = true;
                    if (len > kLdsLong) {                 // giant piece: arrays in the global pool
                        if (tid == 0) {
                            const unsigned long long need = 4ull * (unsigned long long)len;
                            const unsigned long long o = simt::atomic_add64(P.pool_head, need);
                            s_l0 = (o + need <= (unsigned long long)P.pool_cap) ? (int64_t)o : -1;
                        }
                        simt::sync();
                        if (s_l0 < 0) { ok = false; err |= kErrPool; }
                        else { arr = P.pool + s_l0; stride = len; }
                    }
                    if (ok) cnt = tkz_bpe_long(T, at, len, arr, arr + stride, arr + 2 * stride, arr + 3 * stride, dst, &err);
                }
                simt::sync();
            }
            running += cnt;
            k0 += 1;
            nlong -= 1;
        }
    }
    if (tid == 0) { P.tile_count[tile] = running; P.tile_first[tile] = first_abs; }
    if (err) simt::atomic_or((unsigned*)&P.counters[0], (unsigned)err);
}

// -------------------------------------------------------------------------------------------------
// exclusive scan of tile_count (int32) -> tile_base (int64); counters[2..3] (int64) = grand total
// -------------------------------------------------------------------------------------------------
TKZ_KERNEL(256) void k_scan_partials(const int32_t* cnt, int64_t n, int64_t* bsum) {
    TKZ_SHARED int64_t s_w[4];
    const int64_t i0 = simt::bid() * kScanBlock;
    int64_t v = 0;
    for (int j = simt::tid(); j < kScanBlock; j += kThreads) if (i0 + j < n) v += cnt[i0 + j];
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = simt::shflu((uint32_t)v, simt::lane() ^ d), hi = simt::shflu((uint32_t)((uint64_t)v >> 32), simt::lane() ^ d);
        v += (int64_t)(((uint64_t)hi << 32) | lo);
    }
    if (simt::lane() == 0) s_w[simt::wave()] = v;
    simt::sync();
    if (simt::tid() == 0) bsum[simt::bid()] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
TKZ_KERNEL(256) void k_scan_top(int64_t* bsum, int64_t nblk, int64_t* grand) {   // one workgroup, serial over chunks
    TKZ_SHARED int64_t s_carry;
    if (simt::tid() == 0) {
        int64_t run = 0;
        for (int64_t i = 0; i < nblk; ++i) { const int64_t v = bsum[i]; bsum[i] = run; run += v; }
        *grand = run; s_carry = run;
    }
    simt::sync();
}
TKZ_KERNEL(256) void k_scan_final(const int32_t* cnt, int64_t n, const int64_t* boff, int64_t* base) {
    const int64_t i0 = simt::bid() * kScanBlock;
    // each thread owns kScanBlock/kThreads consecutive tiles
    constexpr int per = kScanBlock / kThreads;
    int local[per]; int sum = 0;
    for (int j = 0; j < per; ++j) { const int64_t i = i0 + (int64_t)simt::tid() * per + j; local[j] = i < n ? cnt[i] : 0; sum += local[j]; }
    // block scan in 64-bit via two 32-bit scans would overflow only past 2^31 tokens per 1024 tiles (4 MiB of text): impossible
    int tot;
    int pre = tkz_block_scan(sum, &tot);
    int64_t run = boff[simt::bid()] + pre;
    for (int j = 0; j < per; ++j) { const int64_t i = i0 + (int64_t)simt::tid() * per + j; if (i < n) base[i] = run; run += local[j]; }
}

// -------------------------------------------------------------------------------------------------
// k_gather / k_docoffs
// -------------------------------------------------------------------------------------------------
TKZ_KERNEL(256) void k_gather(const int32_t* tmp, const int32_t* tile_count, const int64_t* tile_first, const int64_t* tile_base,
                              int32_t* out, int64_t out_cap) {
    const int64_t t = simt::bid();
    const int cnt = tile_count[t];
    const int32_t* src = tmp + tile_first[t];
    const int64_t b = tile_base[t];
    for (int i = simt::tid(); i < cnt; i += kThreads) if (b + i < out_cap) out[b + i] = src[i];
}
TKZ_KERNEL(256) void k_docoffs(const int64_t* offs, int64_t n_docs, int64_t total, const int64_t* tile_base, const int32_t* doc_local,
                               const int64_t* grand, int64_t* out_offs) {
    const int64_t stride = simt::nblocks() * simt::nthreads();
    for (int64_t d = simt::bid() * simt::nthreads() + simt::tid(); d <= n_docs; d += stride) {
        const int64_t pos = offs[d];
        out_offs[d] = pos >= total ? *grand : (pos < 0 ? 0 : tile_base[pos / kTile] + doc_local[d]);
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

// =================================================================================================
// launchers
// =================================================================================================
static inline void hook(const Launch& L, int id, int phase) { if (L.hook) L.hook(L.hook_ctx, id, phase, L.stream); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t grid_for(int64_t items) { const int64_t g = cdiv(items, kThreads); return g < 1 ? 1 : (g > 16384 ? 16384 : g); }

void launch_docmark(const Launch& L, const int64_t* d_offs, int64_t n_items, int64_t total, uint64_t* bits, int32_t* counters) {
    hook(L, K_DOCMARK, 0);
    TKZ_LAUNCH(k_docmark, grid_for(n_items + 1), kThreads, L.stream, d_offs, n_items, total, bits, counters);
    hook(L, K_DOCMARK, 1);
}
void launch_pretok_rows(const Launch& L, int pattern, const uint8_t* d_bytes, int64_t total, const uint64_t* docbits,
                        uint64_t* startbits, int64_t nrows, const uint8_t* bmp, int32_t* counters) {
    const int64_t waves = cdiv(nrows, kRowsPerWave), grid = cdiv(waves, kThreads / 64);
    hook(L, K_PRETOK, 0);
    if (pattern == TKZ_PAT_P1)
        TKZ_LAUNCH(k_pretok_rows<TKZ_PAT_P1>, grid, kThreads, L.stream, d_bytes, total, docbits, startbits, nrows, bmp, counters);
    else
        TKZ_LAUNCH(k_pretok_rows<TKZ_PAT_CL100K>, grid, kThreads, L.stream, d_bytes, total, docbits, startbits, nrows, bmp, counters);
    hook(L, K_PRETOK, 1);
}
void launch_pretok_seq(const Launch& L, int pattern, const uint8_t* d_bytes, const int64_t* d_offs, int64_t n_docs, int64_t total,
                       uint64_t* startbits, const uint8_t* bmp, int32_t* counters) {
    hook(L, K_PRETOK, 0);
    TKZ_LAUNCH(k_pretok_seq, grid_for(n_docs), kThreads, L.stream, d_bytes, d_offs, n_docs, total, startbits, pattern, bmp, counters);
    hook(L, K_PRETOK, 1);
}
void launch_encode(const Launch& L, const TkzTables& T, const EncodeParams& P, int64_t ntiles) {
    hook(L, K_ENCODE, 0);
    TKZ_LAUNCH(k_encode_tiles, ntiles, kThreads, L.stream, T, P);
    hook(L, K_ENCODE, 1);
}
void launch_scan(const Launch& L, const int32_t* tile_count, int64_t ntiles, int64_t* bsum, int64_t* tile_base, int64_t* grand) {
    const int64_t nblk = cdiv(ntiles, kScanBlock);
    hook(L, K_SCAN, 0);
    TKZ_LAUNCH(k_scan_partials, nblk, kThreads, L.stream, tile_count, ntiles, bsum);
    TKZ_LAUNCH(k_scan_top, 1, kThreads, L.stream, bsum, nblk, grand);
    TKZ_LAUNCH(k_scan_final, nblk, kThreads, L.stream, tile_count, ntiles, (const int64_t*)bsum, tile_base);
    hook(L, K_SCAN, 1);
}
void launch_gather(const Launch& L, const int32_t* tmp, const int32_t* tile_count, const int64_t* tile_first,
                   const int64_t* tile_base, int64_t ntiles, int32_t* out, int64_t out_cap) {
    hook(L, K_GATHER, 0);
    TKZ_LAUNCH(k_gather, ntiles, kThreads, L.stream, tmp, tile_count, tile_first, tile_base, out, out_cap);
    hook(L, K_GATHER, 1);
}
void launch_docoffs(const Launch& L, const int64_t* d_offs, int64_t n_docs, int64_t total, const int64_t* tile_base,
                    const int32_t* doc_local, const int64_t* grand, int64_t* out_offs) {
    hook(L, K_DOCOFFS, 0);
    TKZ_LAUNCH(k_docoffs, grid_for(n_docs + 1), kThreads, L.stream, d_offs, n_docs, total, tile_base, doc_local, grand, out_offs);
    hook(L, K_DOCOFFS, 1);
}
void launch_corpus(hipStream_t s, int kind, uint64_t seed, int64_t first_doc, int64_t n_docs, int min_len, int max_len,
                   int64_t* d_offs, uint8_t* d_bytes, int64_t cap_bytes, int64_t* d_total) {
    TKZ_LAUNCH(k_corpus_lengths, grid_for(n_docs), kThreads, s, kind, seed, first_doc, n_docs, min_len, max_len, d_offs);
    TKZ_LAUNCH(k_offsets_scan, 1, kThreads, s, d_offs, n_docs, d_total);
    TKZ_LAUNCH(k_corpus_fill, grid_for(n_docs), kThreads, s, kind, seed, first_doc, n_docs, min_len, max_len,
               (const int64_t*)d_offs, d_bytes, cap_bytes);
}

}  // namespace tkz
