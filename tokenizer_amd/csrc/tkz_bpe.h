// tkz_bpe.h -- BytePairEncoder.BytePairEncode (Tokenizer_C#/TokenizerLib/Utils/BytePairEncoder.cs:13-76)
// on the device, in two shapes that perform exactly the reference's merge sequence:
//
//   tkz_bpe_short   one LANE per piece of <= 16 bytes.  The (Index, Rank) list of the reference
//                   becomes: a 16-bit alive mask of part starts, ids[k] = token id of the part that
//                   starts at byte k, pr[k] = packed (rank << 4 | k) of the pair (part at k, next
//                   part) or NOKEY.  One u32 min over the pr slots is the reference's leftmost
//                   strict-min scan (:47-54): equal ranks tie-break on the lower position.
//   tkz_bpe_long    one WORKGROUP per piece of any length: the list is a doubly linked list in
//                   LDS or global scratch, each round is a workgroup-wide min of (rank, position)
//                   followed by the reference's three updates (:58-63).
//
// In both, `ranks.TryGetValue(slice)` of GetRank (:25-36) is a PAIR-table probe on the ids of the two
// adjacent parts (tkz_tables.h), and the first-level ranks come from the directly indexed two-byte
// table.  rank == token id throughout (the value the reference emits at :70-75 for a merged part is
// the rank under which the merge was found).
#pragma once
#include <stdint.h>

#include "tkz_simt.h"
#include "tkz_tables.h"

#define TKZ_NOKEY 0xFFFFFFFFu

enum : int32_t { kErrUtf8 = 1, kErrKeyNotFound = 2, kErrOffsets = 4, kErrPool = 8, kErrTooLong = 16, kErrCapacity = 32 };

TKZ_HD uint32_t tkz_mkkey(int32_t rank, int k) { return rank == TKZ_RANK_NONE ? TKZ_NOKEY : (((uint32_t)rank << 4) | (uint32_t)k); }

// Piece of n bytes, 1 <= n <= 16.  ids/pr: scratch of 16 entries each, entry k at [k * stride].
// Returns the number of tokens; *alive_out has one bit per surviving part (token k is ids[k * stride]).
template <class ByteAt>
TKZ_HD int tkz_bpe_short(const TkzTables& T, ByteAt at, int n, uint32_t* ids, uint32_t* pr, int stride,
                         uint32_t* alive_out, int* err) {
    uint32_t prevb = at(0);
    ids[0] = (uint32_t)T.byte_rank[prevb];
    for (int k = 1; k < n; ++k) {                       // parts = single bytes; initial pair ranks (:37-44)
        const uint32_t b = at(k);
        ids[k * stride] = (uint32_t)T.byte_rank[b];
        pr[(k - 1) * stride] = tkz_mkkey(T.bytepair_rank[(prevb << 8) | b], k - 1);
        prevb = b;
    }
    pr[(n - 1) * stride] = TKZ_NOKEY;
    uint32_t alive = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u);
    for (;;) {                                          // while (byteIndicesAndRanks.Count > 1) (:45)
        uint32_t key = TKZ_NOKEY;
        for (int k = 0; k + 1 < n; ++k) { const uint32_t v = pr[k * stride]; key = v < key ? v : key; }
        if (key == TKZ_NOKEY) break;                    // minRank == int.MaxValue (:65-68)
        const int j = (int)(key & 15u);
        const uint32_t m = key >> 4;
        const int r = tkz_ctz32(alive >> (j + 1)) + j + 1;      // the part being swallowed
        alive &= ~(1u << r);                            // RemoveAt(j + 1) (:63)
        ids[j * stride] = m;
        pr[r * stride] = TKZ_NOKEY;
        const uint32_t hi = alive & ~((2u << r) - 1u);
        pr[j * stride] = hi ? tkz_mkkey(tkz_lookup_pair(T, m, ids[tkz_ctz32(hi) * stride]), j) : TKZ_NOKEY;       // (:58)
        const uint32_t lo = alive & ((1u << j) - 1u);
        if (lo) { const int l = tkz_msb32(lo); pr[l * stride] = tkz_mkkey(tkz_lookup_pair(T, ids[l * stride], m), l); }   // (:59-62)
    }
    int cnt = 0;
    for (uint32_t a = alive; a; a &= a - 1) {
        if (ids[tkz_ctz32(a) * stride] >= (uint32_t)TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;     // ranks[...] throws (:17,:73)
        ++cnt;
    }
    *alive_out = alive;
    return cnt;
}

#ifndef TKZ_NO_SIMT
// ---- workgroup collectives (blockDim.x a multiple of 64, <= 1024) ---------------------------------
// exclusive prefix sum of v over the workgroup; *total = sum over all threads
TKZ_DEV int tkz_block_scan(int v, int* total) {
    TKZ_SHARED int s_w[16];
    const int lane = simt::lane(), wave = simt::wave(), nw = simt::nthreads() >> 6;
    int x = v;
    for (int d = 1; d < 64; d <<= 1) { const int y = simt::shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) s_w[wave] = x;
    simt::sync();
    int woff = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const int s = s_w[w]; if (w < wave) woff += s; tot += s; }
    simt::sync();
    *total = tot;
    return x - v + woff;
}
// minimum of a 64-bit key over the workgroup (every thread gets it)
TKZ_DEV uint64_t tkz_block_min64(uint64_t key) {
    TKZ_SHARED uint64_t s_m[16];
    const int lane = simt::lane(), wave = simt::wave(), nw = simt::nthreads() >> 6;
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = simt::shflu((uint32_t)key, lane ^ d), hi = simt::shflu((uint32_t)(key >> 32), lane ^ d);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        key = o < key ? o : key;
    }
    if (lane == 0) s_m[wave] = key;
    simt::sync();
    uint64_t m = ~0ull;
    for (int w = 0; w < nw; ++w) { const uint64_t s = s_m[w]; m = s < m ? s : m; }
    simt::sync();
    return m;
}

// Piece of n >= 2 bytes processed by the whole workgroup.  ids/pr/nxt/prv: n entries each (LDS or
// global).  Tokens are written to dst in order; returns their number.
template <class ByteAt>
TKZ_DEV int tkz_bpe_long(const TkzTables& T, ByteAt at, int n, int32_t* ids, int32_t* pr, int32_t* nxt, int32_t* prv,
                         int32_t* dst, int* err) {
    const int tid = simt::tid(), G = simt::nthreads();
    for (int k = tid; k < n; k += G) {
        const uint32_t b = at(k);
        ids[k] = T.byte_rank[b];
        pr[k] = (k + 1 < n) ? T.bytepair_rank[(b << 8) | at(k + 1)] : TKZ_RANK_NONE;
        nxt[k] = k + 1; prv[k] = k - 1;
    }
    simt::sync();
    for (;;) {
        uint64_t key = ~0ull;
        for (int k = tid; k < n; k += G) {
            const int32_t r = pr[k];
            if (r != TKZ_RANK_NONE) { const uint64_t c = ((uint64_t)(uint32_t)r << 32) | (uint32_t)k; key = c < key ? c : key; }
        }
        key = tkz_block_min64(key);                     // leftmost minimum (:47-54)
        if (key == ~0ull) break;                        // (:65-68)
        if (tid == 0) {
            const int j = (int)(uint32_t)key;
            const int32_t m = (int32_t)(key >> 32);
            const int r = nxt[j], rr = nxt[r];
            ids[j] = m; ids[r] = -1; pr[r] = TKZ_RANK_NONE;   // RemoveAt(j + 1) (:63)
            nxt[j] = rr;
            if (rr < n) prv[rr] = j;
            pr[j] = rr < n ? tkz_lookup_pair(T, (uint32_t)m, (uint32_t)ids[rr]) : TKZ_RANK_NONE;    // (:58)
            const int l = prv[j];
            if (l >= 0) pr[l] = tkz_lookup_pair(T, (uint32_t)ids[l], (uint32_t)m);                  // (:59-62)
        }
        simt::sync();
    }
    // emit surviving parts in order (:70-75)
    const int c = (n + G - 1) / G;
    const int lo = tid * c < n ? tid * c : n, hi = lo + c < n ? lo + c : n;
    int cnt = 0;
    for (int k = lo; k < hi; ++k) cnt += ids[k] != -1;
    int tot;
    int i = tkz_block_scan(cnt, &tot);
    for (int k = lo; k < hi; ++k) {
        const int32_t id = ids[k];
        if (id == -1) continue;
        if (id >= TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;
        dst[i++] = id;
    }
    simt::sync();
    return tot;
}
#endif  // TKZ_NO_SIMT
