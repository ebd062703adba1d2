// tkz_bpe.h -- BytePairEncoder.BytePairEncode (Tokenizer_C#/TokenizerLib/Utils/BytePairEncoder.cs:13-76)
// on the device, in two shapes that perform exactly the reference's merge sequence:
//
//   tkz_bpe_lane    one LANE per piece of <= 16 (or <= 32) bytes.  The (Index, Rank) list of the reference
//                   becomes: a 16-bit alive mask of part starts, ids[k] = token id of the part that
//                   starts at byte k, pr[k] = packed (rank << 4|5 | k) of the pair (part at k, next
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

enum : int32_t { kErrUtf8 = 1, kErrKeyNotFound = 2, kErrOffsets = 4, kErrPool = 8, kErrTooLong = 16, kErrCapacity = 32, kErrMissCap = 64 };


// Per-lane scratch of tkz_bpe_lane<NMAX>: NMAX pair keys, 16-byte aligned so that the min scan is NMAX/4 16-byte LDS reads -- the
// lane stride kStride (in dwords) is chosen so that those reads are bank-conflict free: stride/4 odd puts the 16 lanes of a b128
// read group on 16 distinct 4-bank slots -- and NMAX ids, read and written one dword at a time, at the odd lane stride kIdStride
// (every dword of LDS the merge state does not take is occupancy: k_merge_short is a chain of dependent gathers).
template <int NMAX> struct TkzBpeGeom;
template <> struct TkzBpeGeom<16> { static constexpr int kStride = 20, kIdStride = 17, kShift = 4; };
template <> struct TkzBpeGeom<32> { static constexpr int kStride = 36, kIdStride = 33, kShift = 5; };
constexpr int kBpeLaneStride = TkzBpeGeom<16>::kStride;

TKZ_HD uint32_t tkz_min3u(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }
TKZ_HD uint32_t tkz_lowmask32(int n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

// One LANE merges one piece of n bytes, 1 <= n <= NMAX (16 or 32), given as little-endian dwords w[0 .. NMAX/4)
// (bytes past n are ignored).  ids/pr: this lane's NMAX-entry arrays (pr 16-byte aligned).  brank: the 256-entry
// single-byte id table.  Returns the number of tokens; *alive_out has one bit per surviving part (token k is ids[k]).
// Written for memory-level parallelism: all first-level gathers are issued together (unconditionally: a load inside
// a lane-divergent branch is waited for inside that branch), and each merge costs ONE round trip to the pair table
// (both re-ranked pairs, both cuckoo slots of each, in flight together).
// byte_id(b): id of the single byte b;  pair_rank(b0, b1): rank of the key b0 b1 or TKZ_RANK_NONE -- the caller decides where those come
// from (k_merge_short keeps the byte ids and the ranks of the lower-case letter pairs in LDS: a gather is a request per lane)
template <int NMAX, class ByteId, class PairRank>
TKZ_HD int tkz_bpe_lane_f(const TkzTables& T, const uint32_t* w, int n, uint32_t* ids, uint32_t* pr, ByteId byte_id, PairRank pair_rank,
                          uint32_t* alive_out, int* err);
template <int NMAX>
TKZ_HD int tkz_bpe_lane(const TkzTables& T, const uint32_t* w, int n, uint32_t* ids, uint32_t* pr, const int32_t* brank,
                        uint32_t* alive_out, int* err) {
    return tkz_bpe_lane_f<NMAX>(T, w, n, ids, pr, [&](uint32_t b) -> uint32_t { return (uint32_t)brank[b]; },
                                [&](uint32_t b0, uint32_t b1) -> int32_t { return T.bytepair_rank[(b0 << 8) | b1]; }, alive_out, err);
}
template <int NMAX, class ByteId, class PairRank>
TKZ_HD int tkz_bpe_lane_f(const TkzTables& T, const uint32_t* w, int n, uint32_t* ids, uint32_t* pr, ByteId byte_id, PairRank pair_rank,
                          uint32_t* alive_out, int* err) {
    constexpr int SH = TkzBpeGeom<NMAX>::kShift;
    uint4* pr4 = reinterpret_cast<uint4*>(pr);
    // first-level state, 16 bytes at a time (keeps the register footprint of the 32-byte variant that of the 16-byte one)
#pragma unroll 1
    for (int c = 0; c < NMAX / 16; ++c) {
        uint32_t bk[17];
#pragma unroll
        for (int k = 0; k < 16; ++k) bk[k] = (w[4 * c + (k >> 2)] >> (8 * (k & 3))) & 0xFFu;
        bk[16] = (c + 1 < NMAX / 16) ? (w[4 * c + 4] & 0xFFu) : 0u;
        uint32_t idv[16], prv[16];
        int32_t r2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) idv[k] = byte_id(bk[k]);                                 // parts = single bytes
        // initial pair ranks (:37-44): only the pairs the piece has (every gather is a request to the memory pipeline, and those
        // requests -- not bytes, not flops -- are what the merge kernels are made of)
#pragma unroll
        for (int k = 0; k < 16; ++k) r2[k] = (16 * c + k + 1 < n) ? pair_rank(bk[k], bk[k + 1]) : TKZ_RANK_NONE;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int g = 16 * c + k;
            prv[k] = (g + 1 < n && r2[k] != TKZ_RANK_NONE) ? (((uint32_t)r2[k] << SH) | (uint32_t)g) : TKZ_NOKEY;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) ids[16 * c + k] = idv[k];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 b;
            b.x = prv[4 * q]; b.y = prv[4 * q + 1]; b.z = prv[4 * q + 2]; b.w = prv[4 * q + 3];
            pr4[4 * c + q] = b;
        }
    }
    uint32_t alive = tkz_lowmask32(n);
    for (;;) {                                          // while (byteIndicesAndRanks.Count > 1) (:45)
        uint32_t key = TKZ_NOKEY;
#pragma unroll
        for (int q = 0; q < NMAX / 4; ++q) {            // leftmost strict min (:47-54): packed (rank, position), ties -> lower position
            const uint4 p = pr4[q];
            key = tkz_min3u(key, tkz_min3u(p.x, p.y, p.z), p.w);
        }
        if (key == TKZ_NOKEY) break;                    // minRank == int.MaxValue (:65-68)
        const int j = (int)(key & (uint32_t)(NMAX - 1));
        const uint32_t m = key >> SH;
        const int r = tkz_ctz32(alive & ~tkz_lowmask32(j + 1));   // the part being swallowed
        alive &= ~(1u << r);                            // RemoveAt(j + 1) (:63)
        // the two re-ranked pairs (:58, :59-62)
        const uint32_t hi = alive & ~tkz_lowmask32(r + 1);
        const uint32_t lo = alive & tkz_lowmask32(j);
        const int l = lo ? tkz_msb32(lo) : 0;
        const int rr = hi ? tkz_ctz32(hi) : 0;
        const uint32_t idr = ids[rr], idl = ids[l];     // (unconditional: see above)
        uint32_t r1, r2s, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2s);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load_pair_slot(T, r1), vr2 = tkz_load_pair_slot(T, r2s);
        const uint4 vl1 = tkz_load_pair_slot(T, l1), vl2 = tkz_load_pair_slot(T, l2);
        ids[j] = m;
        pr[r] = TKZ_NOKEY;
        const int32_t rkr = tkz_match_pair(T, m, idr, vr1, vr2), rkl = tkz_match_pair(T, idl, m, vl1, vl2);
        pr[j] = (hi && rkr != TKZ_RANK_NONE) ? (((uint32_t)rkr << SH) | (uint32_t)j) : TKZ_NOKEY;
        if (lo) pr[l] = rkl != TKZ_RANK_NONE ? (((uint32_t)rkl << SH) | (uint32_t)l) : TKZ_NOKEY;
    }
    int cnt = 0;
    for (uint32_t a = alive; a; a &= a - 1) {
        if (ids[tkz_ctz32(a)] >= (uint32_t)TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;     // ranks[...] throws (:17,:73)
        ++cnt;
    }
    *alive_out = alive;
    return cnt;
}

// One LANE merges one piece of n >= 1 bytes with its state in a caller-provided span of tkz_bpe_var_dwords(n) dwords:
//   ids[n4] | pr[n4] | alive[a4]      n4 = n rounded up to 4, a4 = ceil(n / 32) rounded up to 4 (one bit per part start)
// The same loop as tkz_bpe_lane with the piece length a run-time value (k_merge_long gives every missed piece of a
// pass a span of its own size out of one LDS arena: CJK runs, emoji sequences and long identifiers are merged side by
// side instead of one after the other).  PACKED (vocabularies whose ranks stay below 2^22, i.e. every published one):
// pr holds rank << 10 | position and the leftmost strict minimum (:47-54) is a v_min3 tree over 16-byte LDS reads, as
// in tkz_bpe_lane; otherwise pr holds plain ranks and the minimum is a first-wins scan (6x the VALU work per entry).
// Returns the number of tokens: ids[k] for every set bit k of alive[], in order (tkz_bpe_var_emit).
constexpr int kVarPosBits = 10;                         // positions < 1024 (kArenaPiece)
constexpr int32_t kVarPackedMaxRank = (1 << (32 - kVarPosBits)) - 2;
TKZ_HD int tkz_bpe_var_n4(int n) { return (n + 3) & ~3; }
TKZ_HD int tkz_bpe_var_a4(int n) { return (((n + 31) >> 5) + 3) & ~3; }
TKZ_HD int tkz_bpe_var_dwords(int n) { return 2 * tkz_bpe_var_n4(n) + tkz_bpe_var_a4(n); }

template <bool PACKED, class ByteAt>
TKZ_HD int tkz_bpe_lane_var(const TkzTables& T, ByteAt at, int n, uint32_t* st, int* err, const int32_t* brank = nullptr) {
    if (!brank) brank = T.byte_rank;
    constexpr uint32_t NONE = PACKED ? TKZ_NOKEY : (uint32_t)TKZ_RANK_NONE;
    auto entry = [](int32_t rank, int pos) -> uint32_t {
        if (rank == TKZ_RANK_NONE) return NONE;
        return PACKED ? (((uint32_t)rank << kVarPosBits) | (uint32_t)pos) : (uint32_t)rank;
    };
    const int n4 = tkz_bpe_var_n4(n), nw = (n + 31) >> 5;
    uint32_t* ids = st; uint32_t* pr = st + n4; uint32_t* am = st + 2 * n4;
    uint4* ids4 = reinterpret_cast<uint4*>(ids); uint4* pr4 = reinterpret_cast<uint4*>(pr);
    // first-level state, 16 bytes per step: all 32 gathers of a step are in flight together
#pragma unroll 1
    for (int c = 0; c < n4; c += 16) {
        uint32_t b[17];
#pragma unroll
        for (int k = 0; k < 17; ++k) b[k] = c + k < n ? at(c + k) : 0u;
        uint32_t idv[16]; int32_t r2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) idv[k] = c + k < n ? (uint32_t)brank[b[k]] : 0u;           // parts = single bytes
#pragma unroll
        for (int k = 0; k < 16; ++k) r2[k] = c + k + 1 < n ? T.bytepair_rank[(b[k] << 8) | b[k + 1]] : TKZ_RANK_NONE;   // initial pair ranks (:37-44), only the pairs there are
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c + 4 * q < n4) {
                uint4 a, p;
                a.x = idv[4 * q]; a.y = idv[4 * q + 1]; a.z = idv[4 * q + 2]; a.w = idv[4 * q + 3];
                p.x = c + 4 * q + 1 < n ? entry(r2[4 * q], c + 4 * q) : NONE;
                p.y = c + 4 * q + 2 < n ? entry(r2[4 * q + 1], c + 4 * q + 1) : NONE;
                p.z = c + 4 * q + 3 < n ? entry(r2[4 * q + 2], c + 4 * q + 2) : NONE;
                p.w = c + 4 * q + 4 < n ? entry(r2[4 * q + 3], c + 4 * q + 3) : NONE;
                ids4[(c >> 2) + q] = a; pr4[(c >> 2) + q] = p;
            }
        }
    }
    for (int w = 0; w < nw; ++w) am[w] = tkz_lowmask32(n - 32 * w);
    int cnt = n;
    for (;;) {                                          // while (byteIndicesAndRanks.Count > 1) (:45)
        uint32_t m = NONE; int j = 0;
        if (PACKED) {
#pragma unroll 4
            for (int q = 0; q < (n4 >> 2); ++q) {       // leftmost strict min (:47-54): packed (rank, position), ties -> lower position
                const uint4 p = pr4[q];
                m = tkz_min3u(m, tkz_min3u(p.x, p.y, p.z), p.w);
            }
            if (m == NONE) break;                       // minRank == int.MaxValue (:65-68)
            j = (int)(m & ((1u << kVarPosBits) - 1u));
            m >>= kVarPosBits;
        } else {
#pragma unroll 2
            for (int q = 0; q < (n4 >> 2); ++q) {       // leftmost strict min (:47-54), first wins
                const uint4 p = pr4[q];
                if (p.x < m) { m = p.x; j = 4 * q; }
                if (p.y < m) { m = p.y; j = 4 * q + 1; }
                if (p.z < m) { m = p.z; j = 4 * q + 2; }
                if (p.w < m) { m = p.w; j = 4 * q + 3; }
            }
            if (m == NONE) break;                       // minRank == int.MaxValue (:65-68)
        }
        // r: the part being swallowed (next part after j), rr: the one after it, l: the part before j
        int w = (j + 1) >> 5;
        uint32_t bits = w < nw ? am[w] & (0xFFFFFFFFu << ((j + 1) & 31)) : 0u;
        while (!bits && ++w < nw) bits = am[w];
        const int r = 32 * w + tkz_ctz32(bits);         // exists: pr[j] was a rank
        bits &= bits - 1;
        am[w] &= ~(1u << (r & 31));                     // RemoveAt(j + 1) (:63)
        while (!bits && ++w < nw) bits = am[w];
        const bool hasr = bits != 0;
        const int rr = hasr ? 32 * w + tkz_ctz32(bits) : 0;
        w = j >> 5;
        bits = am[w] & tkz_lowmask32(j & 31);
        while (!bits && --w >= 0) bits = am[w];
        const bool hasl = bits != 0;
        const int l = hasl ? 32 * w + tkz_msb32(bits) : 0;
        const uint32_t idr = ids[rr], idl = ids[l];     // (unconditional: see tkz_bpe_lane)
        uint32_t r1, r2s, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2s);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load_pair_slot(T, r1), vr2 = tkz_load_pair_slot(T, r2s);
        const uint4 vl1 = tkz_load_pair_slot(T, l1), vl2 = tkz_load_pair_slot(T, l2);
        ids[j] = m;                                     // the merged part carries the rank it was found under
        pr[r] = NONE;
        const int32_t rkr = tkz_match_pair(T, m, idr, vr1, vr2), rkl = tkz_match_pair(T, idl, m, vl1, vl2);
        pr[j] = hasr ? entry(rkr, j) : NONE;                                // (:58)
        if (hasl) pr[l] = entry(rkl, l);                                    // (:59-62)
        --cnt;
    }
    for (int w = 0; w < nw; ++w)
        for (uint32_t a = am[w]; a; a &= a - 1)
            if (ids[32 * w + tkz_ctz32(a)] >= (uint32_t)TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;   // ranks[...] throws (:17,:73)
    return cnt;
}
// ---- the same loop WITHOUT an ids[] array (vocabularies whose ranks stay below 2^21: every published one) -----------------
// What limits k_merge_long is how many pieces fit into a CU's LDS at once (a merge is a dependent round trip to the pair table,
// so throughput = pieces in flight / latency), and ids[] is 40 % of a piece's state.  It is redundant:
//   * a part that has never merged is a single byte and its id is brank[byte] (the bytes are staged in LDS anyway);
//   * a part j that HAS merged swallowed the part that began at j+1, so slot j+1 of pr[] is dead for good: it holds the id.
// Dead slots have the top bit set (DEAD | id, or NONE): a live key (rank << 10 | position, rank < 2^21) is always smaller, so the
// v_min3 scan needs no mask, and "no pair left" reads as min >= DEAD.   Layout: pr[n4] | alive[a4].
constexpr int32_t kVarCompactMaxRank = (1 << 21) - 2;
constexpr uint32_t kVarDead = 0x80000000u;
TKZ_HD int tkz_bpe_varc_dwords(int n) { return tkz_bpe_var_n4(n) + tkz_bpe_var_a4(n); }

template <class ByteAt>
TKZ_HD uint32_t tkz_bpe_varc_id(ByteAt at, int n, const uint32_t* pr, const uint32_t* am, const int32_t* brank, int x) {
    const int y = x + 1 < n ? x + 1 : x;                 // (x = n-1 never merges to its right: am bit of x itself is set)
    const bool merged = x + 1 < n && !((am[y >> 5] >> (y & 31)) & 1u);
    const uint32_t a = pr[y] & ~kVarDead, b = (uint32_t)brank[at(x)];     // (both loads unconditional)
    return merged ? a : b;
}

template <class ByteAt>
TKZ_HD int tkz_bpe_lane_varc(const TkzTables& T, ByteAt at, int n, uint32_t* st, int* err, const int32_t* brank) {
    auto entry = [](int32_t rank, int pos) -> uint32_t {
        return rank == TKZ_RANK_NONE ? TKZ_NOKEY : (((uint32_t)rank << kVarPosBits) | (uint32_t)pos);
    };
    const int n4 = tkz_bpe_var_n4(n), nw = (n + 31) >> 5;
    uint32_t* pr = st; uint32_t* am = st + n4;
    uint4* pr4 = reinterpret_cast<uint4*>(pr);
#pragma unroll 1
    for (int c = 0; c < n4; c += 16) {                   // first-level pair ranks (:37-44), 16 bytes per step, their gathers in flight together
        uint32_t b[17];
#pragma unroll
        for (int k = 0; k < 17; ++k) b[k] = c + k < n ? at(c + k) : 0u;
        int32_t r2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) r2[k] = c + k + 1 < n ? T.bytepair_rank[(b[k] << 8) | b[k + 1]] : TKZ_RANK_NONE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c + 4 * q < n4) {
                uint4 p;
                p.x = c + 4 * q + 1 < n ? entry(r2[4 * q], c + 4 * q) : TKZ_NOKEY;
                p.y = c + 4 * q + 2 < n ? entry(r2[4 * q + 1], c + 4 * q + 1) : TKZ_NOKEY;
                p.z = c + 4 * q + 3 < n ? entry(r2[4 * q + 2], c + 4 * q + 2) : TKZ_NOKEY;
                p.w = c + 4 * q + 4 < n ? entry(r2[4 * q + 3], c + 4 * q + 3) : TKZ_NOKEY;
                pr4[(c >> 2) + q] = p;
            }
        }
    }
    for (int w = 0; w < nw; ++w) am[w] = tkz_lowmask32(n - 32 * w);
    int cnt = n;
    for (;;) {                                          // while (byteIndicesAndRanks.Count > 1) (:45)
        uint32_t m = TKZ_NOKEY;
#pragma unroll 4
        for (int q = 0; q < (n4 >> 2); ++q) {           // leftmost strict min (:47-54): packed (rank, position), ties -> lower position
            const uint4 p = pr4[q];
            m = tkz_min3u(m, tkz_min3u(p.x, p.y, p.z), p.w);
        }
        if (m >= kVarDead) break;                       // minRank == int.MaxValue (:65-68)
        const int j = (int)(m & ((1u << kVarPosBits) - 1u));
        m >>= kVarPosBits;
        // r: the part being swallowed (next part after j), rr: the one after it, l: the part before j
        int w = (j + 1) >> 5;
        uint32_t bits = w < nw ? am[w] & (0xFFFFFFFFu << ((j + 1) & 31)) : 0u;
        while (!bits && ++w < nw) bits = am[w];
        const int r = 32 * w + tkz_ctz32(bits);         // exists: pr[j] was a rank
        bits &= bits - 1;
        am[w] &= ~(1u << (r & 31));                     // RemoveAt(j + 1) (:63)
        while (!bits && ++w < nw) bits = am[w];
        const bool hasr = bits != 0;
        const int rr = hasr ? 32 * w + tkz_ctz32(bits) : 0;
        w = j >> 5;
        bits = am[w] & tkz_lowmask32(j & 31);
        while (!bits && --w >= 0) bits = am[w];
        const bool hasl = bits != 0;
        const int l = hasl ? 32 * w + tkz_msb32(bits) : 0;
        const uint32_t idr = tkz_bpe_varc_id(at, n, pr, am, brank, rr), idl = tkz_bpe_varc_id(at, n, pr, am, brank, l);
        uint32_t r1, r2s, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2s);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load_pair_slot(T, r1), vr2 = tkz_load_pair_slot(T, r2s);
        const uint4 vl1 = tkz_load_pair_slot(T, l1), vl2 = tkz_load_pair_slot(T, l2);
        pr[r] = TKZ_NOKEY;                              // dead for good
        pr[j + 1] = kVarDead | m;                       // ... and the slot behind j carries the id of the merged part (= the rank it was found under)
        const int32_t rkr = tkz_match_pair(T, m, idr, vr1, vr2), rkl = tkz_match_pair(T, idl, m, vl1, vl2);
        pr[j] = hasr ? entry(rkr, j) : TKZ_NOKEY;                           // (:58)
        if (hasl) pr[l] = entry(rkl, l);                                    // (:59-62)
        --cnt;
    }
    for (int w = 0; w < nw; ++w)
        for (uint32_t a = am[w]; a; a &= a - 1)
            if (tkz_bpe_varc_id(at, n, pr, am, brank, 32 * w + tkz_ctz32(a)) >= (uint32_t)TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;   // ranks[...] throws (:17,:73)
    return cnt;
}
template <class ByteAt>
TKZ_HD void tkz_bpe_varc_emit(ByteAt at, const uint32_t* st, int n, const int32_t* brank, int32_t* dst) {
    const int n4 = tkz_bpe_var_n4(n), nw = (n + 31) >> 5;
    const uint32_t* pr = st; const uint32_t* am = st + n4;
    int i = 0;
    for (int w = 0; w < nw; ++w)
        for (uint32_t a = am[w]; a; a &= a - 1) dst[i++] = (int32_t)tkz_bpe_varc_id(at, n, pr, am, brank, 32 * w + tkz_ctz32(a));
}

// ---- ... and for pieces of up to 64 bytes with the alive bits in REGISTERS (one 64-bit mask) --------------------------------------
// A merge of the LDS-mask form above is a chain of ~8 dependent LDS round trips (the scan, two walks over the alive words, the ids of the two
// neighbours through their alive bits) before its one round trip to the pair table; k_merge_long runs at 3 wavefronts per SIMD and is bound
// by exactly that chain.  With the mask in registers the neighbours are two bit scans, and what is left in LDS is the scan and one read per
// neighbour id.  State: pr[n4] only.
TKZ_HD uint64_t tkz_lowmask64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
template <class ByteAt>
TKZ_HD uint32_t tkz_bpe_varc64_id(ByteAt at, int n, const uint32_t* pr, uint64_t alive, const int32_t* brank, int x) {
    const int y = x + 1 < n ? x + 1 : x;
    const bool merged = x + 1 < n && !((alive >> y) & 1ull);
    const uint32_t a = pr[y] & ~kVarDead, b = (uint32_t)brank[at(x)];     // (both loads unconditional)
    return merged ? a : b;
}
template <class ByteAt>
TKZ_HD int tkz_bpe_lane_varc64(const TkzTables& T, ByteAt at, int n, uint32_t* pr, int* err, const int32_t* brank, uint64_t* alive_out) {
    auto entry = [](int32_t rank, int pos) -> uint32_t {
        return rank == TKZ_RANK_NONE ? TKZ_NOKEY : (((uint32_t)rank << kVarPosBits) | (uint32_t)pos);
    };
    const int n4 = tkz_bpe_var_n4(n);
    uint4* pr4 = reinterpret_cast<uint4*>(pr);
#pragma unroll 1
    for (int c = 0; c < n4; c += 16) {                   // first-level pair ranks (:37-44), 16 bytes per step, their gathers in flight together
        uint32_t b[17];
#pragma unroll
        for (int k = 0; k < 17; ++k) b[k] = c + k < n ? at(c + k) : 0u;
        int32_t r2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) r2[k] = c + k + 1 < n ? T.bytepair_rank[(b[k] << 8) | b[k + 1]] : TKZ_RANK_NONE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (c + 4 * q < n4) {
                uint4 p;
                p.x = c + 4 * q + 1 < n ? entry(r2[4 * q], c + 4 * q) : TKZ_NOKEY;
                p.y = c + 4 * q + 2 < n ? entry(r2[4 * q + 1], c + 4 * q + 1) : TKZ_NOKEY;
                p.z = c + 4 * q + 3 < n ? entry(r2[4 * q + 2], c + 4 * q + 2) : TKZ_NOKEY;
                p.w = c + 4 * q + 4 < n ? entry(r2[4 * q + 3], c + 4 * q + 3) : TKZ_NOKEY;
                pr4[(c >> 2) + q] = p;
            }
        }
    }
    uint64_t alive = tkz_lowmask64(n);
    int cnt = n;
    for (;;) {                                          // while (byteIndicesAndRanks.Count > 1) (:45)
        uint32_t m = TKZ_NOKEY;
#pragma unroll 4
        for (int q = 0; q < (n4 >> 2); ++q) {           // leftmost strict min (:47-54): packed (rank, position), ties -> lower position
            const uint4 p = pr4[q];
            m = tkz_min3u(m, tkz_min3u(p.x, p.y, p.z), p.w);
        }
        if (m >= kVarDead) break;                       // minRank == int.MaxValue (:65-68)
        const int j = (int)(m & ((1u << kVarPosBits) - 1u));
        m >>= kVarPosBits;
        // r: the part being swallowed (next part after j), rr: the one after it, l: the part before j
        const uint64_t hi = alive & ~tkz_lowmask64(j + 1);          // (not empty: pr[j] was a rank)
        const int r = tkz_ctz64(hi);
        const uint64_t hi2 = hi & (hi - 1);
        const bool hasr = hi2 != 0;
        const int rr = hasr ? tkz_ctz64(hi2) : 0;
        alive &= ~(1ull << r);                          // RemoveAt(j + 1) (:63)
        const uint64_t lo = alive & tkz_lowmask64(j);
        const bool hasl = lo != 0;
        const int l = hasl ? tkz_msb64(lo) : 0;
        const uint32_t idr = tkz_bpe_varc64_id(at, n, pr, alive, brank, rr), idl = tkz_bpe_varc64_id(at, n, pr, alive, brank, l);
        uint32_t r1, r2s, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2s);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load_pair_slot(T, r1), vr2 = tkz_load_pair_slot(T, r2s);
        const uint4 vl1 = tkz_load_pair_slot(T, l1), vl2 = tkz_load_pair_slot(T, l2);
        pr[r] = TKZ_NOKEY;                              // dead for good
        pr[j + 1] = kVarDead | m;                       // ... and the slot behind j carries the id of the merged part (= the rank it was found under)
        const int32_t rkr = tkz_match_pair(T, m, idr, vr1, vr2), rkl = tkz_match_pair(T, idl, m, vl1, vl2);
        pr[j] = hasr ? entry(rkr, j) : TKZ_NOKEY;                           // (:58)
        if (hasl) pr[l] = entry(rkl, l);                                    // (:59-62)
        --cnt;
    }
    for (uint64_t a = alive; a; a &= a - 1)
        if (tkz_bpe_varc64_id(at, n, pr, alive, brank, tkz_ctz64(a)) >= (uint32_t)TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;   // ranks[...] throws (:17,:73)
    *alive_out = alive;
    return cnt;
}
template <class ByteAt>
TKZ_HD void tkz_bpe_varc64_emit(ByteAt at, const uint32_t* pr, uint64_t alive, int n, const int32_t* brank, int32_t* dst) {
    int i = 0;
    for (uint64_t a = alive; a; a &= a - 1) dst[i++] = (int32_t)tkz_bpe_varc64_id(at, n, pr, alive, brank, tkz_ctz64(a));
}

// the tokens of a piece merged by tkz_bpe_lane_var, in order
TKZ_HD void tkz_bpe_var_emit(const uint32_t* st, int n, int32_t* dst) {
    const int n4 = tkz_bpe_var_n4(n), nw = (n + 31) >> 5;
    const uint32_t* am = st + 2 * n4;
    int i = 0;
    for (int w = 0; w < nw; ++w)
        for (uint32_t a = am[w]; a; a &= a - 1) dst[i++] = (int32_t)st[32 * w + tkz_ctz32(a)];
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

// minimum of a 32-bit value over the workgroup
TKZ_DEV uint32_t tkz_block_min32(uint32_t v) {
    TKZ_SHARED uint32_t s_m[16];
    const int lane = simt::lane(), wave = simt::wave(), nw = simt::nthreads() >> 6;
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = simt::shflu(v, lane ^ d); v = o < v ? o : v; }
    if (lane == 0) s_m[wave] = v;
    simt::sync();
    uint32_t m = 0xFFFFFFFFu;
    for (int w = 0; w < nw; ++w) { const uint32_t x = s_m[w]; m = x < m ? x : m; }
    simt::sync();
    return m;
}
// exclusive max-scan over the threads of the workgroup (identity -1)
TKZ_DEV int tkz_block_exclusive_max(int v) {
    TKZ_SHARED int s_w[16];
    const int lane = simt::lane(), wave = simt::wave(), nw = simt::nthreads() >> 6;
    int x = v;
    for (int d = 1; d < 64; d <<= 1) { const int y = simt::shfl_up(x, d); if (lane >= d && y > x) x = y; }
    if (lane == 63) s_w[wave] = x;
    simt::sync();
    int carry = -1;
    for (int w = 0; w < wave && w < nw; ++w) carry = s_w[w] > carry ? s_w[w] : carry;
    simt::sync();
    const int prevx = simt::shfl_up(x, 1);
    int ex = lane == 0 ? -1 : prevx;
    return ex > carry ? ex : carry;
}

// A piece of n >= 2 bytes merged by the whole workgroup, in ROUNDS.  The parts are a dense array; one round merges
// every pair whose rank equals the current minimum m, exactly as the reference would get to them one after the other:
//   * leftmost first, so inside a chain of adjacent candidates (a a a a ...) every other one merges (:47-54, :63);
//   * a merge re-ranks the pair to its left and the pair to its right (:58-62).  If one of those new ranks is BELOW m
//     the reference would take that pair next, before the remaining rank-m pairs: the round therefore applies only
//     the merges up to and including the leftmost one that creates such a pair; the rest wait for a later round.
//     (The pair to the right is the "transient" one -- merged token + the still unmerged next part -- because the
//     reference has not yet reached the candidates further right.)  With well-formed vocabularies the cut never
//     triggers and a run of n equal bytes collapses in about log2(n) rounds; for arbitrary rank tables it keeps the
//     result identical to the one-merge-at-a-time loop.
// Arrays (n entries each, LDS or global): idsA/prA (state), s1/s2 (scratch), idsB/prB (next state).  pr[i] is the rank
// of (part i, part i+1).  Tokens are written to dst in order; returns their number.
// The rounds, on whatever arrays the state is in.  Runs until no pair has a rank (returns true) or, when stop_at > 0, until the
// state has shrunk to stop_at parts or fewer (returns false: the caller moves the state to faster memory and calls again).
// ids / pr are left pointing at the current state, cnt at its length.
// Is the state dominated by ONE pair rank?  (A run of one letter is: after a few odd merges at its end -- poor rounds -- thousands of equal
// pairs merge in the next round; a chain of words is not.)  Three probes: the ranks of the pairs at 1/4, 1/2 and 3/4 of the state; a
// probe that a sixteenth of all pairs share says "homogeneous": such a piece stays with the rounds.  Called by the whole workgroup.
TKZ_DEV bool tkz_bpe_long_homogeneous(const int32_t* pr, int cnt) {
    const int tid = simt::tid(), G = simt::nthreads();
#ifdef TKZ_TAIL_ALWAYS       // (development / test builds: every piece the rounds are slow on -- with TKZ_TAIL_FEW set high: every piece -- goes to the tail)
    return false;
#endif
    if (cnt < 64) return false;
    const int32_t v1 = pr[cnt / 4], v2 = pr[cnt / 2], v3 = pr[(3 * cnt) / 4];
    const int c = (cnt + G - 1) / G;
    const int lo = tid * c < cnt ? tid * c : cnt, hi = lo + c < cnt ? lo + c : cnt;
    int n1 = 0, n2 = 0, n3 = 0;
    for (int i = lo; i < hi; ++i) { const int32_t r = pr[i]; n1 += r == v1; n2 += r == v2; n3 += r == v3; }
    int t1, t2, t3;
    (void)tkz_block_scan(n1, &t1); (void)tkz_block_scan(n2, &t2); (void)tkz_block_scan(n3, &t3);
    const int thr = cnt / 16;
    return (v1 != TKZ_RANK_NONE && t1 >= thr) || (v2 != TKZ_RANK_NONE && t2 >= thr) || (v3 != TKZ_RANK_NONE && t3 >= thr);
}
// few / few_cap (> 0): also returns false -- *slow = true -- after kBpeSlowRounds consecutive rounds that merged fewer than `few` pairs each,
// once the state is down to few_cap parts (such a piece is handed to tkz_bpe_long_tail).
constexpr int kBpeSlowRounds = 3;
TKZ_DEV bool tkz_bpe_long_rounds(const TkzTables& T, int& cnt, int32_t*& ids, int32_t*& pr, int32_t*& s1, int32_t* s2, int32_t*& idsN, int32_t* prN, int stop_at, int* rounds = nullptr,
                                 int few = 0, int few_cap = 0, bool* slow = nullptr) {
    const int tid = simt::tid(), G = simt::nthreads();
    constexpr int32_t kNotMerge = 0x7FFFFFFE;
    int nslow = 0;
    for (;;) {
        if (rounds) ++*rounds;
        if (stop_at > 0 && cnt <= stop_at) return false;
        if (few > 0 && nslow >= kBpeSlowRounds && cnt <= few_cap) {
            if (!tkz_bpe_long_homogeneous(pr, cnt)) { *slow = true; return false; }
            nslow = 0;
        }
        const int c = (cnt + G - 1) / G;                  // contiguous block of parts per thread
        const int lo = tid * c < cnt ? tid * c : cnt, hi = lo + c < cnt ? lo + c : cnt;
        // 1. the minimum rank
        uint32_t mymin = (uint32_t)TKZ_RANK_NONE;
        for (int i = lo; i < hi; ++i) { const uint32_t r = (uint32_t)pr[i]; mymin = r < mymin ? r : mymin; }
        const int32_t m = (int32_t)tkz_block_min32(mymin);
        if (m == TKZ_RANK_NONE) return true;              // (:65-68)
        // 2. candidates, position in their chain -> merge flags (s1)
        int lastNon = -1;                                 // last non-candidate index inside my block
        for (int i = lo; i < hi; ++i) if (pr[i] != m) lastNon = i;
        const int carry = tkz_block_exclusive_max(lastNon);   // last non-candidate before my block
        {
            int ln = carry;
            for (int i = lo; i < hi; ++i) {
                if (pr[i] != m) { ln = i; s1[i] = 0; }
                else s1[i] = ((i - (ln + 1)) & 1) == 0 ? 1 : 0;   // even offset from the chain start
            }
        }
        simt::sync();
        // 3. the two re-ranked pairs of every merge: s2 = rank of (left part, merged), prN = rank of (merged, next part) (transient)
        uint32_t firstViol = 0xFFFFFFFFu;
        for (int i = lo; i < hi; ++i) {
            int32_t L = kNotMerge, R = TKZ_RANK_NONE;
            if (s1[i]) {
                L = TKZ_RANK_NONE;
                if (i >= 1) {
                    const uint32_t left = (i >= 2 && s1[i - 2]) ? (uint32_t)m : (uint32_t)ids[i - 1];
                    L = tkz_lookup_pair(T, left, (uint32_t)m);
                }
                if (i + 2 < cnt) R = tkz_lookup_pair(T, (uint32_t)m, (uint32_t)ids[i + 2]);
                if ((L < m || R < m) && (uint32_t)i < firstViol) firstViol = (uint32_t)i;
            }
            s2[i] = L; prN[i] = R;
        }
        const uint32_t istar = tkz_block_min32(firstViol);   // merges beyond istar wait (includes a sync)
        // 4. new state, compacted into idsN / prN' (prN holds R: read my own entries before overwriting -> two passes)
        int alive = 0;
        for (int i = lo; i < hi; ++i) alive += !(i >= 1 && s2[i - 1] != kNotMerge && (uint32_t)(i - 1) <= istar);
        int tot;
        int o = tkz_block_scan(alive, &tot);
        // values are computed into registers per element and written to idsN / s1 (s1 is free again after the scan's sync)
        for (int i = lo; i < hi; ++i) {
            const bool swallowed = i >= 1 && s2[i - 1] != kNotMerge && (uint32_t)(i - 1) <= istar;
            if (swallowed) continue;
            const bool mg = s2[i] != kNotMerge && (uint32_t)i <= istar;
            int32_t nid, npr;
            if (mg) {
                nid = m;
                if (i + 2 < cnt) { const bool mg2 = s2[i + 2] != kNotMerge && (uint32_t)(i + 2) <= istar; npr = mg2 ? s2[i + 2] : prN[i]; }
                else npr = TKZ_RANK_NONE;
            } else {
                nid = ids[i];
                const bool mg1 = i + 1 < cnt && s2[i + 1] != kNotMerge && (uint32_t)(i + 1) <= istar;
                npr = mg1 ? s2[i + 1] : pr[i];
            }
            idsN[o] = nid; s1[o] = npr; ++o;              // s1[o]: o <= i, and s1[<= i] is no longer read by anyone
        }
        simt::sync();
        // s1 now holds the new pr; make it the pr array of the next round
        { int32_t* t = pr; pr = s1; s1 = t; }
        { int32_t* t = ids; ids = idsN; idsN = t; }
        nslow = cnt - tot < few ? nslow + 1 : 0;
        cnt = tot;
        simt::sync();
    }
}

// The same rounds with the state in LDS in its COMPACT form -- ids[cap] | pr[cap] | one flag byte per part (9 bytes per part, so
// 16 Ki parts fit a CU's LDS) -- for a workgroup of exactly 1024 threads, each owning C = cap / 1024 consecutive parts in a fully
// unrolled loop (what it computes for its parts stays in registers between the phases).  What a round needs from a merge's
// neighbours is recomputed by the thread that needs it instead of being passed through scratch arrays:
//   merge at i (flag[i], i <= istar):  id' = m;  pr' = rank(m, id[i+2])   or rank(m, m) when i+2 merges too      (:58)
//   part j left of a merge (flag[j+1]): id' = id[j];  pr' = rank(id[j], m)                                      (:59-62)
//   part right of a merge (flag[j-1]): swallowed                                                               (:63)
// and the round is cut after the leftmost merge that creates a pair ranked below m, exactly as in tkz_bpe_long_rounds.
// Returns true when no pair is left, false after kBpeSlowRounds consecutive rounds that merged fewer than `few` pairs each (what is left is
// then better served by tkz_bpe_long_tail: a diverse piece goes on for thousands of rounds of one or two merges each, ~13 us a round).
template <int C>
TKZ_DEV bool tkz_bpe_long_rounds_lds(const TkzTables& T, int& cnt, int32_t* ids, int32_t* pr, uint8_t* flag, int few, int* rounds = nullptr) {
    const int tid = simt::tid();
    int nslow = 0;
    for (;;) {
        if (rounds) ++*rounds;
        const int c = (cnt + 1023) >> 10;                 // parts per thread this round (<= C)
        const int lo = tid * c < cnt ? tid * c : cnt;
        // 1. the minimum rank
        int32_t mypr[C], myid[C];
        uint32_t mymin = (uint32_t)TKZ_RANK_NONE;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int i = lo + q;
            const bool in = q < c && i < cnt;
            mypr[q] = in ? pr[i] : TKZ_RANK_NONE; myid[q] = in ? ids[i] : 0;
            mymin = (uint32_t)mypr[q] < mymin ? (uint32_t)mypr[q] : mymin;
        }
        const int32_t m = (int32_t)tkz_block_min32(mymin);
        if (m == TKZ_RANK_NONE) return true;              // (:65-68)
        // 2. candidates; leftmost first inside a chain of adjacent candidates: every other one, counted from the chain's start
        int lastNon = -1;
#pragma unroll
        for (int q = 0; q < C; ++q) if (q < c && lo + q < cnt && mypr[q] != m) lastNon = lo + q;
        int ln = tkz_block_exclusive_max(lastNon);
        uint32_t myflags = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int i = lo + q;
            if (q < c && i < cnt) {
                if (mypr[q] != m) { ln = i; flag[i] = 0; }
                else { const bool mg = ((i - (ln + 1)) & 1) == 0; flag[i] = mg ? 1 : 0; myflags |= mg ? (1u << q) : 0u; }
            }
        }
        simt::sync();
        // 3. the leftmost merge that creates a pair ranked below m: merges beyond it wait for a later round
        uint32_t firstViol = 0xFFFFFFFFu;
        int32_t Rv[C];                                    // rank(m, next part) of my merges (the transient pair)
#pragma unroll
        for (int q = 0; q < C; ++q) {
            Rv[q] = TKZ_RANK_NONE;
            if ((myflags >> q) & 1u) {
                const int i = lo + q;
                int32_t L = TKZ_RANK_NONE;
                if (i >= 1) L = tkz_lookup_pair(T, (i >= 2 && flag[i - 2]) ? (uint32_t)m : (uint32_t)ids[i - 1], (uint32_t)m);
                if (i + 2 < cnt) Rv[q] = tkz_lookup_pair(T, (uint32_t)m, (uint32_t)ids[i + 2]);
                if ((L < m || Rv[q] < m) && (uint32_t)i < firstViol) firstViol = (uint32_t)i;
                // a merging part's own pair rank is m (known): its slot carries rank(left part, m) to whoever needs it in step 4 -- the
                // part on its left, or a merge two to the left -- instead of a second table round trip there
                pr[i] = L;
            }
        }
        const uint32_t istar = tkz_block_min32(firstViol);   // (its barriers publish the slots written above)
        // 4. the new state of my parts, in registers; then compaction in place (everything is read before anything is written)
        int32_t nid[C], npr[C];
        int alive = 0;
        uint32_t keep = 0;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const int i = lo + q;
            nid[q] = 0; npr[q] = TKZ_RANK_NONE;
            if (!(q < c && i < cnt)) continue;
            if (i >= 1 && flag[i - 1] && (uint32_t)(i - 1) <= istar) continue;                       // swallowed (:63)
            keep |= 1u << q; ++alive;
            if (((myflags >> q) & 1u) && (uint32_t)i <= istar) {
                nid[q] = m;
                if (i + 2 < cnt) npr[q] = (flag[i + 2] && (uint32_t)(i + 2) <= istar) ? pr[i + 2] : Rv[q];      // rank(m, m) left there by the merge at i + 2
            } else {
                nid[q] = myid[q];
                npr[q] = (i + 1 < cnt && flag[i + 1] && (uint32_t)(i + 1) <= istar) ? pr[i + 1] : mypr[q];          // rank(id, m) left there by the merge at i + 1
            }
        }
        int tot;
        int o = tkz_block_scan(alive, &tot);              // (its barriers also separate the reads above from the writes below)
#pragma unroll
        for (int q = 0; q < C; ++q) if ((keep >> q) & 1u) { ids[o] = nid[q]; pr[o] = npr[q]; ++o; }
        nslow = cnt - tot < few ? nslow + 1 : 0;         // (ONE poor round says nothing: a run of one letter has a single odd merge between rounds of thousands)
        cnt = tot;
        simt::sync();
        if (nslow >= kBpeSlowRounds) {
            if (!tkz_bpe_long_homogeneous(pr, cnt)) return false;
            nslow = 0;
        }
    }
}

// The TAIL of a long diverse piece: the reference's loop itself (BytePairEncoder.cs:45-64), one merge at a time, by ONE wavefront on the
// state in LDS -- ids[cap] | pr[cap] as the rounds above leave them, plus one alive bit per part and the minimum rank of every block
// of 32 parts (both in what was the flag area).  A merge is: the minimum over the block minima (a few 16-byte reads per lane and a DPP
// reduction), the leftmost part of that rank inside the first such block (:47-54), its neighbours through the alive bits, the two pair
// lookups (:58-62; the same address in every lane: one request), three writes, and the minima of the <= 3 blocks touched: ~1 us, where a round
// costs ~13 us and a 32 KiB run of words needs ~3000 of them for a handful of merges each.  Parts are never moved: what is left is
// compacted by the caller.  Workgroups of 1024 threads; wavefront 0 works, the others wait at the barrier behind it.
constexpr int kTailBlock = 32;
constexpr uint32_t kTailDead = 0x80000000u;
// IDS_LDS: the ids of the parts are in LDS beside pr[] (the state the LDS rounds leave, <= kBpeLongLds parts).  Otherwise (the state the
// GLOBAL rounds leave, <= kBpeTailCap parts: only pr[] fits LDS): ids[] -- global memory -- holds the ids the parts had when the tail began,
// and a part that has merged since keeps its id in the slot behind it, which died with its first merge and stays dead (kTailDead | id).
template <bool IDS_LDS>
TKZ_DEV uint32_t tkz_tail_id(const int32_t* ids, const int32_t* pr, const uint32_t* alive, int cnt, int x) {
    if (IDS_LDS) return (uint32_t)ids[x];
    const bool merged = x + 1 < cnt && !((alive[(x + 1) >> 5] >> ((x + 1) & 31)) & 1u);
    return merged ? ((uint32_t)pr[x + 1] & ~kTailDead) : (uint32_t)ids[x];
}
// Many merges per round trip, exactly.  Thread b of the workgroup owns block b (32 consecutive slots) and proposes the block's smallest
// pair (rank, then position): its neighbours, their ids and the two pair ranks the merge would create are looked up by all threads at
// once -- one trip to the pair table for up to 1024 merges instead of one each.  Which of them may be applied NOW is decided from
// `bound`_b = the smallest key that could come before anything else once b's proposal has merged: the second smallest pair of the block,
// the two pairs the merge creates.  With tau = the minimum of all bounds, every proposal below tau is -- at its turn in the reference's
// order (BytePairEncoder.cs:47-54) -- the leftmost minimum of the whole piece with the neighbourhood it was looked up in: nothing that is
// not itself such a proposal can precede it.  Two proposals interfere exactly when the part of one is among the two parts behind the other
// (the one it swallows, the one it is paired with next): tau is capped at the later of the two.  The global minimum is always applied.
// Keys: rank << 10 | block, 64 bits (<= 1024 blocks; equal keys count as "not below": the next batch takes them).
// scratch: 8 * nthreads + 2 * nthreads + 16 * (nthreads / 64) bytes of LDS behind the alive bits.
template <bool IDS_LDS>
TKZ_DEV void tkz_bpe_long_tail(const TkzTables& T, int cnt, int32_t* ids, int32_t* pr, uint32_t* alive, void* scratch, unsigned long long* prof = nullptr) {
    const int tid = simt::tid(), lane = simt::lane(), wave = simt::wave(), G = simt::nthreads();
    const int nblk = (cnt + kTailBlock - 1) / kTailBlock, nw = (cnt + 31) >> 5;       // (nblk <= G: cnt <= kBpeTailCap, 1024 threads)
    constexpr uint32_t NONE = (uint32_t)TKZ_RANK_NONE;
    constexpr uint64_t NOKEY = ~0ull;                           // (keys: 64 bits -- ranks go up to TKZ_MAX_RANK = 2^27)
    uint64_t* s_key = reinterpret_cast<uint64_t*>(scratch);
    uint16_t* s_j = reinterpret_cast<uint16_t*>(s_key + G);
    uint64_t* s_red = reinterpret_cast<uint64_t*>(s_j + G);
    auto wave_min_key = [](uint64_t v) -> uint64_t {
        const uint32_t hi = (uint32_t)(v >> 32), mh = simt::wave_min_u32(hi);
        const uint32_t ml = simt::wave_min_u32(hi == mh ? (uint32_t)v : 0xFFFFFFFFu);
        return ((uint64_t)mh << 32) | ml;
    };
    auto make_key = [](uint32_t rank, int blk) -> uint64_t { return rank >= (uint32_t)TKZ_RANK_NONE ? ~0ull : (((uint64_t)rank << 10) | (uint64_t)(uint32_t)blk); };
    // (the last part never has a pair: its pr is TKZ_RANK_NONE already; slots beyond cnt are padded so that whole blocks can be read)
    for (int i = cnt + tid; i < nblk * kTailBlock; i += G) pr[i] = TKZ_RANK_NONE;
    for (int w = tid; w < nw; w += G) alive[w] = tkz_lowmask32(cnt - 32 * w);
    simt::sync();
    static_assert(kTailBlock == 32, "eight quads a block");
    const int blk = tid;
    const bool owner = blk < nblk;
    long long n_batch = 0, n_merge = 0, n_cand = 0, n_cap = 0;
    const long long tq0 = prof ? simt::clock() : 0;
    for (;;) {
        // ---- the block's smallest pair (leftmost of its rank) and its second smallest ----
        // (a dead slot holds TKZ_RANK_NONE or kTailDead | id: as unsigned values both lie at or above TKZ_RANK_NONE)
        uint32_t m = NONE, s2 = NONE;
        int jj = 0;
        if (owner) {
            const uint4* q = reinterpret_cast<const uint4*>(pr + blk * kTailBlock);
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = q[k];             // (eight 16-byte reads, requested together)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t x = w4[i];
                    if (x < m) { s2 = m; m = x; jj = 4 * k + i; } else if (x < s2) s2 = x;
                }
            }
        }
        const uint64_t k1 = make_key(m, blk), key2 = make_key(s2, blk);
        const bool cand = k1 != NOKEY;
        const int j = blk * kTailBlock + jj;
        s_key[tid] = k1;
        s_j[tid] = (uint16_t)j;
        // ---- the neighbours of the proposal, and what its merge would create ----
        uint64_t bound = NOKEY;
        bool hasl = false, hasr = false;
        int r = 0, l = 0, rr = -1;
        int32_t rkr = TKZ_RANK_NONE, rkl = TKZ_RANK_NONE;
        if (cand) {
            // r: the part being swallowed (the next one alive after j), rr: the one after it, l: the one before j
            int w = (j + 1) >> 5;
            uint32_t bits = w < nw ? alive[w] & (0xFFFFFFFFu << ((j + 1) & 31)) : 0u;
            while (!bits && ++w < nw) bits = alive[w];
            r = 32 * w + tkz_ctz32(bits);                        // exists: pr[j] was a rank
            bits &= bits - 1;
            while (!bits && ++w < nw) bits = alive[w];
            hasr = bits != 0;
            rr = hasr ? 32 * w + tkz_ctz32(bits) : -1;
            w = j >> 5;
            bits = alive[w] & tkz_lowmask32(j & 31);
            while (!bits && --w >= 0) bits = alive[w];
            hasl = bits != 0;
            l = hasl ? 32 * w + tkz_msb32(bits) : 0;
            const uint32_t idr = tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, hasr ? rr : 0), idl = tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, l);
            rkr = hasr ? tkz_lookup_pair(T, m, idr) : TKZ_RANK_NONE;        // (:58)
            rkl = hasl ? tkz_lookup_pair(T, idl, m) : TKZ_RANK_NONE;        // (:59-62)
            const uint64_t nkr = make_key((uint32_t)rkr, j >> 5), nkl = make_key((uint32_t)rkl, l >> 5);
            bound = key2 < nkr ? key2 : nkr;
            if (nkl < bound) bound = nkl;
        }
        simt::sync();                                            // (every proposal is posted)
        if (cand) {
            auto meets = [&](int p) {
                const int f = p >> 5;
                if (p < 0 || f == blk) return;
                const uint64_t kf = s_key[f];
                if (kf != NOKEY && (int)s_j[f] == p) { const uint64_t later = kf > k1 ? kf : k1; if (later < bound) bound = later; }
            };
            meets(r); meets(rr);
        }
        {
            const uint64_t wg = wave_min_key(k1), wt = wave_min_key(bound);
            if (lane == 0) { s_red[2 * wave] = wg; s_red[2 * wave + 1] = wt; }
        }
        simt::sync();
        uint64_t g, tau;
        {
            const int nwv = G >> 6;
            const uint64_t a = lane < nwv ? s_red[2 * lane] : NOKEY, c = lane < nwv ? s_red[2 * lane + 1] : NOKEY;
            g = wave_min_key(a); tau = wave_min_key(c);
        }
        if (g == NOKEY) break;                                   // minRank == int.MaxValue (:65-68)
        const bool go = cand && (k1 == g || k1 < tau);
        if (prof) {                                              // (development builds: how many merges a trip to the pair table buys)
            const int ng = tkz_popc64(simt::ballot(go)), nc = tkz_popc64(simt::ballot(cand));
            if (tid == 0) ++n_batch;
            if (lane == 0) { n_merge += ng; n_cand += nc; }
        }
        if (go) {
            simt::atomic_and(&alive[r >> 5], ~(1u << (r & 31)));   // RemoveAt(j + 1) (:63)  (two threads may clear bits of one word)
            pr[r] = TKZ_RANK_NONE;
            if (IDS_LDS) ids[j] = (int32_t)m;                    // the merged part carries the rank it was found under ...
            else pr[j + 1] = (int32_t)(kTailDead | m);           // ... in the slot behind it (dead since this part's first merge: r == j + 1 then)
            pr[j] = rkr;
            if (hasl) pr[l] = rkl;
        }
        simt::sync();
    }
    if (prof && lane == 0) {
        simt::atomic_add64(&prof[17], (unsigned long long)n_merge); simt::atomic_add64(&prof[18], (unsigned long long)n_cand);
        if (tid == 0) {
            simt::atomic_add64(&prof[16], (unsigned long long)n_batch); simt::atomic_add64(&prof[19], (unsigned long long)n_cap);
            simt::atomic_max64(&prof[20], (unsigned long long)n_batch);
            simt::atomic_add64(&prof[22], (unsigned long long)(simt::clock() - tq0)); simt::atomic_max64(&prof[23], (unsigned long long)(simt::clock() - tq0));
        }
    }
    simt::sync();
}
// the survivors of a tail, in order (:70-75); returns their number
template <bool IDS_LDS>
TKZ_DEV int tkz_bpe_long_tail_emit(int cnt, const int32_t* ids, const int32_t* pr, const uint32_t* alive, int32_t* dst, int* err) {
    const int tid = simt::tid(), G = simt::nthreads();
    const int c0 = (cnt + G - 1) / G;
    const int lo0 = tid * c0 < cnt ? tid * c0 : cnt, hi0 = lo0 + c0 < cnt ? lo0 + c0 : cnt;
    int mine = 0;
    for (int i = lo0; i < hi0; ++i) mine += (int)((alive[i >> 5] >> (i & 31)) & 1u);
    int tot;
    int o = tkz_block_scan(mine, &tot);
    for (int i = lo0; i < hi0; ++i) {
        if (!((alive[i >> 5] >> (i & 31)) & 1u)) continue;
        const int32_t id = (int32_t)tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, i);
        if (id >= TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;
        dst[o++] = id;
    }
    simt::sync();
    return tot;
}

// lds: 9 * kBpeLongLds bytes of LDS (workgroups of 1024 threads) or null.  The state starts in the global arrays when the piece is
// longer than kBpeLongLds parts and MOVES INTO LDS as soon as it has shrunk to that: a round is ten passes over the state and as
// many workgroup barriers, and most rounds of a long diverse piece (one per distinct rank) happen when a few thousand parts are
// left -- in L2 / Infinity Cache a round costs ~25 us, in LDS ~5.
constexpr int kBpeLongLds = 16384;
#ifndef TKZ_TAIL_FEW
#define TKZ_TAIL_FEW 16
#endif
#ifndef TKZ_TAIL_FEW_GLOBAL
#define TKZ_TAIL_FEW_GLOBAL 256
#endif
constexpr int kBpeTailFew = TKZ_TAIL_FEW;      // rounds in LDS that merge fewer pairs than this (three in a row) hand the piece to the one-merge-at-a-time tail
constexpr int kBpeTailFewGlobal = TKZ_TAIL_FEW_GLOBAL;   // ... rounds in global memory (~150 us each, ten times an LDS round)
constexpr int kBpeTailCap = 32768;             // parts whose pair ranks and alive bits, with the tail's scratch, fit the 9 * kBpeLongLds bytes of LDS (one thread a block of 32)
static_assert(kBpeLongLds / 8 + (kBpeLongLds / 32 + 64) * 4 <= kBpeLongLds, "alive bits + block minima fit the flag area");
static_assert(kBpeTailCap * 4 + kBpeTailCap / 8 + 1024 * 10 + 16 * 16 <= 9 * kBpeLongLds && kBpeTailCap / 32 <= 1024, "the tail's state for kBpeTailCap parts (pair ranks, alive bits, scratch of 1024 threads) fits the workgroup's LDS");
template <class ByteAt>
TKZ_DEV int tkz_bpe_long(const TkzTables& T, ByteAt at, int n, int32_t* idsA, int32_t* prA, int32_t* s1g, int32_t* s2g,
                         int32_t* idsB, int32_t* prB, int32_t* dst, int* err, int32_t* lds = nullptr, unsigned long long* prof = nullptr) {
    const int tid = simt::tid(), G = simt::nthreads();
    const bool use_lds = lds != nullptr && G == 1024;
    const bool start_in_lds = use_lds && n <= kBpeLongLds;
    int32_t* ids = start_in_lds ? lds : idsA;
    int32_t* pr = start_in_lds ? lds + kBpeLongLds : prA;
    for (int k = tid; k < n; k += G) {
        const uint32_t b = at(k);
        ids[k] = T.byte_rank[b];
        pr[k] = (k + 1 < n) ? T.bytepair_rank[(b << 8) | at(k + 1)] : TKZ_RANK_NONE;
    }
    simt::sync();
    int cnt = n;
    bool done = false;
    long long t0 = prof ? simt::clock() : 0, t1 = t0;
    int rg = 0, rl = 0;
    auto finish_prof = [&](int tokens) {
        if (prof && tid == 0) {
            const long long t2 = simt::clock();
            simt::atomic_add64(&prof[8], 1ull); simt::atomic_add64(&prof[9], (unsigned long long)(t1 - t0)); simt::atomic_add64(&prof[10], (unsigned long long)(t2 - t1));
            simt::atomic_add64(&prof[11], (unsigned long long)n); simt::atomic_add64(&prof[12], (unsigned long long)tokens);
            simt::atomic_max64(&prof[13], (unsigned long long)(t2 - t0));
            simt::atomic_add64(&prof[14], (unsigned long long)rg); simt::atomic_add64(&prof[15], (unsigned long long)rl);
        }
    };
    if (!start_in_lds) {
        int32_t* s1 = s1g; int32_t* idsN = idsB;
        bool slow = false;
        done = tkz_bpe_long_rounds(T, cnt, ids, pr, s1, s2g, idsN, prB, use_lds ? kBpeLongLds : 0, prof ? &rg : nullptr,
                                   use_lds ? kBpeTailFewGlobal : 0, kBpeTailCap, &slow);
        if (prof) t1 = simt::clock();
        if (!done && slow) {
            // A diverse piece: the rounds in global memory merge a handful of pairs each at ~150 us a round, and the piece is still too long
            // for the LDS rounds.  Its pair ranks alone fit LDS (4 bytes a part, up to kBpeTailCap parts): the tail takes it from here, the ids
            // the parts have now staying where they are, in global memory.
            int32_t* lpr = lds;
            uint32_t* alive = reinterpret_cast<uint32_t*>(lds + kBpeTailCap);
            void* bmin = alive + kBpeTailCap / 32;                // (the tail's scratch)
            for (int k = tid; k < cnt; k += G) lpr[k] = pr[k];
            simt::sync();
            if (prof && tid == 0) simt::atomic_add64(&prof[7], (unsigned long long)cnt);
            tkz_bpe_long_tail<false>(T, cnt, ids, lpr, alive, bmin, prof);
            const int tot = tkz_bpe_long_tail_emit<false>(cnt, ids, lpr, alive, dst, err);
            finish_prof(tot);
            return tot;
        }
        if (!done) {                                     // the state fits LDS now: move it
            for (int k = tid; k < cnt; k += G) { lds[k] = ids[k]; lds[kBpeLongLds + k] = pr[k]; }
            simt::sync();
            ids = lds; pr = lds + kBpeLongLds;
        }
    }
    bool tail = false;
    if (!done) tail = !tkz_bpe_long_rounds_lds<kBpeLongLds / 1024>(T, cnt, ids, pr, reinterpret_cast<uint8_t*>(lds + 2 * kBpeLongLds), kBpeTailFew, prof ? &rl : nullptr);
    if (tail) {
        // what the rounds left: one merge at a time, by one wavefront (tkz_bpe_long_tail); then the survivors, in order
        uint32_t* alive = reinterpret_cast<uint32_t*>(lds + 2 * kBpeLongLds);              // (the flag area: cap bytes = alive bits + block minima)
        void* bmin = alive + kBpeLongLds / 32;                    // (the tail's scratch)
        if (prof && tid == 0) simt::atomic_add64(&prof[7], (unsigned long long)cnt);
        tkz_bpe_long_tail<true>(T, cnt, ids, pr, alive, bmin, prof);
        const int tot = tkz_bpe_long_tail_emit<true>(cnt, ids, pr, alive, dst, err);
        finish_prof(tot);
        return tot;
    }
    finish_prof(cnt);
    // emit surviving parts in order (:70-75)
    const int c = (cnt + G - 1) / G;
    const int lo = tid * c < cnt ? tid * c : cnt, hi = lo + c < cnt ? lo + c : cnt;
    for (int i = lo; i < hi; ++i) {
        const int32_t id = ids[i];
        if (id >= TKZ_PSEUDO_BASE) *err |= kErrKeyNotFound;
        dst[i] = id;
    }
    simt::sync();
    return cnt;
}
#endif  // TKZ_NO_SIMT
