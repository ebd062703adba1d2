// tkz_bpe.h -- BytePairEncoder.BytePairEncode (Tokenizer_C#/TokenizerLib/Utils/BytePairEncoder.cs:13-76)
// on the device, in two shapes that perform exactly the reference's merge sequence:
//
//   tkz_bpe_short   one LANE per piece of <= 16 bytes.  The (Index, Rank) list of the reference
//                   becomes: a 16-bit alive mask of part starts, ids[k] = token id of the part that
//                   starts at byte k, pr[k] = packed (rank << 4 | k) of the pair (part at k, next
//                   part) or NOKEY.  One u32 min over the 16 pr slots is the reference's leftmost
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

// Per-lane scratch of tkz_bpe_short: 16 ids + 16 pair keys, each array 16-byte aligned so that the
// min scan is four 16-byte LDS reads.  kBpeLaneStride (in dwords) spaces the lanes of a wave so that
// those reads are bank-conflict free (20 dwords: the 16 lanes of a read group land on 16 distinct
// 4-bank slots of the 64-bank LDS).
constexpr int kBpeLaneStride = 20;

TKZ_HD uint32_t tkz_min3u(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }

// Piece of n bytes, 1 <= n <= 16, given as four little-endian dwords w0..w3 (bytes past n ignored).
// ids/pr: this lane's 16-entry arrays (16-byte aligned).  brank: the 256-entry single-byte id table
// (in LDS on the device).  Returns the number of tokens; *alive_out has one bit per surviving part
// (token k is ids[k]).  Written for memory-level parallelism: all first-level gathers are issued
// together, and each merge costs ONE round trip to the pair table (both re-ranked pairs, both cuckoo slots of each, in flight together).
TKZ_HD int tkz_bpe_short(const TkzTables& T, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int n,
                         uint32_t* ids, uint32_t* pr, const int32_t* brank, uint32_t* alive_out, int* err) {
    uint32_t bk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t w = k < 4 ? w0 : (k < 8 ? w1 : (k < 12 ? w2 : w3));
        bk[k] = (w >> (8 * (k & 3))) & 0xFFu;
    }
    // Unconditional, clamped gathers: a load inside a lane-divergent branch is waited for inside that
    // branch, which would serialise these 31 independent loads (bytes past n index valid table entries).
    uint32_t idv[16], prv[16];
    int32_t r2[15];
#pragma unroll
    for (int k = 0; k < 16; ++k) idv[k] = (uint32_t)brank[bk[k]];                             // parts = single bytes
#pragma unroll
    for (int k = 0; k < 15; ++k) r2[k] = T.bytepair_rank[(bk[k] << 8) | bk[k + 1]];           // initial pair ranks (:37-44)
#pragma unroll
    for (int k = 0; k < 15; ++k) prv[k] = (k + 1 < n) ? tkz_mkkey(r2[k], k) : TKZ_NOKEY;
    prv[15] = TKZ_NOKEY;
    uint4* ids4 = reinterpret_cast<uint4*>(ids);
    uint4* pr4 = reinterpret_cast<uint4*>(pr);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint4 a, b;
        a.x = idv[4 * q]; a.y = idv[4 * q + 1]; a.z = idv[4 * q + 2]; a.w = idv[4 * q + 3];
        b.x = prv[4 * q]; b.y = prv[4 * q + 1]; b.z = prv[4 * q + 2]; b.w = prv[4 * q + 3];
        ids4[q] = a; pr4[q] = b;
    }
    uint32_t alive = (1u << n) - 1u;                    // n <= 16
    for (;;) {                                          // while (byteIndicesAndRanks.Count > 1) (:45)
        const uint4 p0 = pr4[0], p1 = pr4[1], p2 = pr4[2], p3 = pr4[3];
        const uint32_t m0 = tkz_min3u(p0.x, p0.y, p0.z), m1 = tkz_min3u(p0.w, p1.x, p1.y), m2 = tkz_min3u(p1.z, p1.w, p2.x);
        const uint32_t m3 = tkz_min3u(p2.y, p2.z, p2.w), m4 = tkz_min3u(p3.x, p3.y, p3.z);
        const uint32_t key = tkz_min3u(tkz_min3u(m0, m1, m2), m3, m4 < p3.w ? m4 : p3.w);   // leftmost strict min (:47-54)
        if (key == TKZ_NOKEY) break;                    // minRank == int.MaxValue (:65-68)
        const int j = (int)(key & 15u);
        const uint32_t m = key >> 4;
        const int r = tkz_ctz32(alive >> (j + 1)) + j + 1;      // the part being swallowed
        alive &= ~(1u << r);                            // RemoveAt(j + 1) (:63)
        // the two re-ranked pairs (:58, :59-62): both first probes are issued before either is examined
        const uint32_t hi = alive & ~((2u << r) - 1u);
        const uint32_t lo = alive & ((1u << j) - 1u);
        const int l = lo ? tkz_msb32(lo) : 0;
        const int rr = hi ? tkz_ctz32(hi) : 0;
        const uint32_t idr = ids[rr], idl = ids[l];     // (unconditional: see above)
        uint32_t r1, r2, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load16(&T.pair_slots[r1]), vr2 = tkz_load16(&T.pair_slots[r2]);
        const uint4 vl1 = tkz_load16(&T.pair_slots[l1]), vl2 = tkz_load16(&T.pair_slots[l2]);
        ids[j] = m;
        pr[r] = TKZ_NOKEY;
        pr[j] = hi ? tkz_mkkey(tkz_match_pair(m, idr, vr1, vr2), j) : TKZ_NOKEY;
        if (lo) pr[l] = tkz_mkkey(tkz_match_pair(idl, m, vl1, vl2), l);
    }
    int cnt = 0;
    for (uint32_t a = alive; a; a &= a - 1) {
        if (ids[tkz_ctz32(a)] >= (uint32_t)TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;     // ranks[...] throws (:17,:73)
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
