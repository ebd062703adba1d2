// tkz_bpe.h -- BytePairEncoder.BytePairEncode (Tokenizer_C#/TokenizerLib/Utils/BytePairEncoder.cs:13-76)
// on the device, in two shapes that perform exactly the reference's merge sequence:
//
//   tkz_bpe_lane    one LANE per piece of <= 16 (or <= 32) bytes.  The (Index, Rank) list of the reference
//                   becomes: a 16-bit alive mask of part starts, ids[k] = token id of the part that
//                   starts at byte k, pr[k] = packed (rank << 4|5 | k) of the pair (part at k, next
//                   part) or NOKEY.  One u32 min over the 16 pr slots is the reference's leftmost
//                   strict-min scan (:47-54): equal ranks tie-break on the lower position.
//   tkz_bpe_long    one WORKGROUP per piece of any length (the same code with a workgroup of ONE wavefront for the pieces of
//                   257..1024 bytes): the parts stay in their slots in LDS (pair ranks, ids, alive bits) and every step
//                   applies MANY of the reference's merges at once, each exactly as the reference would get to it --
//                   batches of proposals that provably cannot be disturbed before their turn, rounds for chains of equal
//                   pairs (tkz_bpe_long_tail); pieces beyond 32 Ki parts are first brought down to that by rounds on
//                   dense arrays in global memory (tkz_bpe_long_rounds).
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
    // leftmost strict min (:47-54): packed (rank, position), ties -> lower position
    auto scan = [&]() -> uint32_t {
        uint32_t key = TKZ_NOKEY;
#pragma unroll
        for (int q = 0; q < NMAX / 4; ++q) {
            const uint4 p = pr4[q];
            key = tkz_min3u(key, tkz_min3u(p.x, p.y, p.z), p.w);
        }
        return key;
    };
    // (round 6: the scan is taken out of the merge's chain of dependent round trips, as in tkz_bpe_lane_u below: while the four gathers of a merge are in
    //  flight the two slots it re-ranks are blanked and the minimum over the others is taken; the next pair is the smaller of that and the two new keys)
    uint32_t key = scan();
    while (key != TKZ_NOKEY) {                          // while (byteIndicesAndRanks.Count > 1) (:45); minRank == int.MaxValue (:65-68)
        const int j = (int)(key & (uint32_t)(NMAX - 1));
        const uint32_t m = key >> SH;
        const int r = tkz_ctz32(alive & ~tkz_lowmask32(j + 1));   // the part being swallowed
        alive &= ~(1u << r);                            // RemoveAt(j + 1) (:63)
        // the two re-ranked pairs (:58, :59-62)
        const uint32_t hi = alive & ~tkz_lowmask32(r + 1);
        const uint32_t lo = alive & tkz_lowmask32(j);
        const int l = lo ? tkz_msb32(lo) : j;
        const int rr = hi ? tkz_ctz32(hi) : 0;
        const uint32_t idr = ids[rr], idl = ids[l];     // (unconditional: see above)
        uint32_t r1, r2s, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2s);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load_pair_slot(T, r1), vr2 = tkz_load_pair_slot(T, r2s);
        const uint4 vl1 = tkz_load_pair_slot(T, l1), vl2 = tkz_load_pair_slot(T, l2);
        ids[j] = m;
        pr[r] = TKZ_NOKEY;
        pr[j] = TKZ_NOKEY;                              // (blanked for the scan below)
        pr[l] = TKZ_NOKEY;                              // (without a left part: slot j once more)
        const uint32_t rest = scan();
        const int32_t rkr = tkz_match_pair(T, m, idr, vr1, vr2), rkl = tkz_match_pair(T, idl, m, vl1, vl2);
        const uint32_t ej = (hi && rkr != TKZ_RANK_NONE) ? (((uint32_t)rkr << SH) | (uint32_t)j) : TKZ_NOKEY;
        const uint32_t el = lo ? (rkl != TKZ_RANK_NONE ? (((uint32_t)rkl << SH) | (uint32_t)l) : TKZ_NOKEY) : ej;
        pr[j] = ej;
        pr[l] = el;
        key = tkz_min3u(rest, ej, el);
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

// ---- round 6: the lane form the 17..64-byte pieces run on (k_merge_long's fast batches) -------------------------------------------
// The same merge sequence as above (BytePairEncoder.cs:45-64), rebuilt around what the ISA of round 5's form (tkz_bpe_lane_varc with a 64-bit alive mask in registers) showed: of ~230 VALU
// wave-instructions a merge, ~140 were 64-bit mask arithmetic, lane-divergent branches around the neighbour searches and a scan loop whose trip
// count differed from lane to lane.  Here
//   * every piece of a batch has a span of ONE size: nq quads of pair keys behind its bytes (nq = the batch's longest piece; a shorter piece's
//     spare quads hold NOKEY), so the min scan is a loop on the SCALAR unit and its reads are the same instruction in every lane;
//   * the alive bits are ONE register of the width the batch needs (M = uint32_t up to 32 bytes: 70 % of mixed text's long misses);
//   * nothing in a merge is a branch: a missing neighbour is an index that reads something harmless and a value that is replaced by a select;
//   * a pair key is rank << 10 | position built with one shift-or: "no such pair" (int.MaxValue, :23,35) shifts to 0xFFFFFC00 | position, which like
//     every dead slot lies at or above kVarDead -- "no pair left" (:65-68) is min >= kVarDead as before;
//   * the first-level ranks (:37-44) are read off the piece's dwords (one byte permute an index) and the tokens leave in one walk that also finds
//     the bytes that are no key (:73).
// State: bw[] = the piece's bytes as dwords in LDS, 16-byte aligned, readable up to dword nq (what lies past byte n is never used as a value);
// pr[4 * nq] pair keys / dead slots (a merged part keeps its id in the dead slot behind its first byte, as in tkz_bpe_lane_varc).
template <class M> TKZ_HD M tkz_um_bit(int i) { return (M)((M)1 << i); }
template <class M> TKZ_HD M tkz_um_low(int n) { return n >= (int)(8 * sizeof(M)) ? (M)~(M)0 : (M)(((M)1 << n) - (M)1); }
TKZ_HD int tkz_um_ctz(uint32_t x) { return tkz_ctz32(x); }
TKZ_HD int tkz_um_ctz(uint64_t x) { return tkz_ctz64(x); }
TKZ_HD int tkz_um_msb(uint32_t x) { return tkz_msb32(x); }
TKZ_HD int tkz_um_msb(uint64_t x) { return tkz_msb64(x); }
TKZ_HD int tkz_um_popc(uint32_t x) { return tkz_popc32(x); }
TKZ_HD int tkz_um_popc(uint64_t x) { return tkz_popc64(x); }
// index into bytepair_rank of the pair that starts at byte k (0..3) of the little-endian dword w (wn: the dword behind it)
TKZ_HD uint32_t tkz_pair_index(uint32_t w, uint32_t wn, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_perm_b32: byte 0 <- the pair's second byte, byte 1 <- its first, bytes 2, 3 <- 0
    return k == 0 ? __builtin_amdgcn_perm(wn, w, 0x0c0c0001u) : k == 1 ? __builtin_amdgcn_perm(wn, w, 0x0c0c0102u)
         : k == 2 ? __builtin_amdgcn_perm(wn, w, 0x0c0c0203u) : __builtin_amdgcn_perm(wn, w, 0x0c0c0304u);
#else
    const uint32_t b0 = (w >> (8 * k)) & 0xFFu, b1 = k < 3 ? (w >> (8 * k + 8)) & 0xFFu : wn & 0xFFu;
    return (b0 << 8) | b1;
#endif
}
// the id of the part that starts at byte x (alive): a single byte's from brank, a merged part's from the dead slot behind x
template <class M>
TKZ_HD uint32_t tkz_bpe_u_id(const uint8_t* pb, const uint32_t* pr, const int32_t* brank, M dead, int x, int last_slot) {
    const int y = x < last_slot ? x + 1 : last_slot;
    const uint32_t a = pr[y] & ~kVarDead, b = (uint32_t)brank[pb[x]];          // (both loads unconditional)
    return ((dead >> x) & (M)2) ? a : b;                                       // bit x + 1 of dead: the part has swallowed its right neighbour
}
// first-level pair ranks (:37-44): four positions a dword, sixteen positions a step -- all sixteen gathers of a step are requested before the first is
// used (the loop over single quads waited for its four gathers quad after quad: nq round trips to the table instead of nq / 4)
TKZ_HD void tkz_bpe_lane_u_init(const TkzTables& T, const uint32_t* bw, int n, int nq, uint32_t* pr) {
    uint4* pr4 = reinterpret_cast<uint4*>(pr);
    const uint4* bw4 = reinterpret_cast<const uint4*>(bw);
    for (int q0 = 0; q0 < nq; q0 += 4) {
        const uint4 wv = bw4[q0 >> 2];                  // (the bytes are padded to whole quads; the dword behind them is the first pair key: never used as a value)
        const uint32_t w[5] = {wv.x, wv.y, wv.z, wv.w, bw[q0 + 4]};
        int32_t r[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = T.bytepair_rank[tkz_pair_index(w[k >> 2], w[(k >> 2) + 1], k & 3)];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int q = q0 + d;
            if (q < nq) {                               // (wave-uniform)
                uint4 p;
                p.x = 4 * q + 1 < n ? (((uint32_t)r[4 * d] << kVarPosBits) | (uint32_t)(4 * q)) : TKZ_NOKEY;
                p.y = 4 * q + 2 < n ? (((uint32_t)r[4 * d + 1] << kVarPosBits) | (uint32_t)(4 * q + 1)) : TKZ_NOKEY;
                p.z = 4 * q + 3 < n ? (((uint32_t)r[4 * d + 2] << kVarPosBits) | (uint32_t)(4 * q + 2)) : TKZ_NOKEY;
                p.w = 4 * q + 4 < n ? (((uint32_t)r[4 * d + 3] << kVarPosBits) | (uint32_t)(4 * q + 3)) : TKZ_NOKEY;
                pr4[q] = p;
            }
        }
    }
}
// the merges; returns the number of tokens, *alive_out: one bit per part that is left (its first byte)
template <class M>
TKZ_HD int tkz_bpe_lane_u_merge(const TkzTables& T, const uint32_t* bw, int n, int nq, uint32_t* pr, const int32_t* brank, M* alive_out) {
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(bw);
    uint4* pr4 = reinterpret_cast<uint4*>(pr);
    const int last_slot = 4 * nq - 1;
    const M nmask = tkz_um_low<M>(n);
    M alive = nmask;
    // leftmost strict min (:47-54): packed (rank, position), ties -> lower position
    auto scan = [&]() -> uint32_t {
        uint32_t k0 = TKZ_NOKEY, k1 = TKZ_NOKEY;
        for (int q = 0; q < nq; q += 2) {
            const uint4 a = pr4[q];
            k0 = tkz_min3u(k0, tkz_min3u(a.x, a.y, a.z), a.w);
            if (q + 1 < nq) { const uint4 b = pr4[q + 1]; k1 = tkz_min3u(k1, tkz_min3u(b.x, b.y, b.z), b.w); }
        }
        return k0 < k1 ? k0 : k1;
    };
    // The scan is kept OUT of the merge's chain of dependent round trips (alive bits -> neighbour ids in LDS -> pair table -> match): while the four
    // gathers of a merge are in flight, the slots the merge re-ranks are blanked and the minimum over everything else is taken; the next pair is then
    // the smaller of that and the two new keys (keys differ in their position bits, so the minimum is the reference's (rank, position) order exactly).
    uint32_t key = scan();
    while (key < kVarDead) {                            // while (byteIndicesAndRanks.Count > 1) (:45); minRank == int.MaxValue (:65-68)
        const int j = (int)(key & ((1u << kVarPosBits) - 1u));
        const uint32_t m = key >> kVarPosBits;
        // r: the part being swallowed (next part after j), rr: the one after it, l: the part before j
        const M hi = alive & (M)(((M)~(M)1) << j);      // (not empty: pr[j] was a rank)
        const int r = tkz_um_ctz(hi);
        const M hi2 = hi & (hi - 1);
        const bool hasr = hi2 != 0;
        const int rr = tkz_um_ctz((M)(hi2 | tkz_um_bit<M>(8 * (int)sizeof(M) - 1)));
        alive &= ~tkz_um_bit<M>(r);                     // RemoveAt(j + 1) (:63)
        const M lo = alive & (M)(tkz_um_bit<M>(j) - 1);
        const bool hasl = lo != 0;
        const int l = hasl ? tkz_um_msb(lo) : j;
        const M dead = nmask & ~alive;
        const uint32_t idr = tkz_bpe_u_id<M>(pb, pr, brank, dead, rr < last_slot ? rr : last_slot, last_slot);
        const uint32_t idl = tkz_bpe_u_id<M>(pb, pr, brank, dead, l, last_slot);
        uint32_t r1, r2s, l1, l2;
        tkz_pair_slots(T, m, idr, &r1, &r2s);
        tkz_pair_slots(T, idl, m, &l1, &l2);
        const uint4 vr1 = tkz_load_pair_slot(T, r1), vr2 = tkz_load_pair_slot(T, r2s);
        const uint4 vl1 = tkz_load_pair_slot(T, l1), vl2 = tkz_load_pair_slot(T, l2);
        pr[r] = TKZ_NOKEY;                              // dead for good
        pr[j + 1] = kVarDead | m;                       // ... and the slot behind j carries the id of the merged part (= the rank it was found under)
        pr[j] = TKZ_NOKEY;                              // (blanked for the scan below)
        pr[l] = TKZ_NOKEY;
        const uint32_t rest = scan();
        const int32_t rkr = tkz_match_pair(T, m, idr, vr1, vr2), rkl = tkz_match_pair(T, idl, m, vl1, vl2);
        const uint32_t ej = hasr ? (((uint32_t)rkr << kVarPosBits) | (uint32_t)j) : TKZ_NOKEY;      // (:58)
        const uint32_t el = hasl ? (((uint32_t)rkl << kVarPosBits) | (uint32_t)l) : ej;             // (:59-62; without a left part: slot j once more)
        pr[j] = ej;
        pr[l] = el;
        key = tkz_min3u(rest, ej, el);
    }
    *alive_out = alive;
    return tkz_um_popc(alive);
}
template <class M>
TKZ_HD int tkz_bpe_lane_u(const TkzTables& T, const uint32_t* bw, int n, int nq, uint32_t* pr, const int32_t* brank, M* alive_out) {
    tkz_bpe_lane_u_init(T, bw, n, nq, pr);
    return tkz_bpe_lane_u_merge<M>(T, bw, n, nq, pr, brank, alive_out);
}
// the tokens of such a piece, in order: dst[0 .. count) when store, the first four also in q4; *err: a surviving byte that is no key (:17,:73)
template <class M>
TKZ_HD void tkz_bpe_lane_u_emit(const uint32_t* bw, int n, int nq, const uint32_t* pr, const int32_t* brank, M alive, bool store, int32_t* dst, uint32_t* q4, int* err) {
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(bw);
    const int last_slot = 4 * nq - 1;
    M a = alive;
    uint32_t bad = 0;
    auto next = [&]() -> uint32_t {                     // (a != 0)
        const int p = tkz_um_ctz(a);
        a &= a - 1;
        const int pn = a ? tkz_um_ctz(a) : n;           // the next part's start: the part is bytes [p, pn)
        const int y = p < last_slot ? p + 1 : last_slot;
        const uint32_t m = pr[y] & ~kVarDead, b = (uint32_t)brank[pb[p]];      // (both loads unconditional)
        const uint32_t id = pn - p > 1 ? m : b;
        bad |= id >= (uint32_t)TKZ_PSEUDO_BASE ? 1u : 0u;
        return id;
    };
    // (two parts a step -- both parts' LDS reads requested together -- was measured: emission 11 -> 13 % of the kernel on mixed text, 15 -> 19 % on real text)
#pragma unroll
    for (int i = 0; i < 4; ++i) {                       // (the first four by name: the quad lives in registers)
        q4[i] = 0u;
        if (a) { q4[i] = next(); if (store) dst[i] = (int32_t)q4[i]; }
    }
    int32_t* d = dst + 4;
    while (a) *d++ = (int32_t)next();                   // (a fifth token: the caller stores -- count > 4)
    if (bad) *err |= kErrKeyNotFound;
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
TKZ_DEV bool tkz_bpe_long_rounds(const TkzTables& T, int& cnt, int32_t*& ids, int32_t*& pr, int32_t*& s1, int32_t* s2, int32_t*& idsN, int32_t* prN, int stop_at, int* rounds = nullptr) {
    const int tid = simt::tid(), G = simt::nthreads();
    constexpr int32_t kNotMerge = 0x7FFFFFFE;
    for (;;) {
        if (stop_at > 0 && cnt <= stop_at) return false;
        if (rounds) ++*rounds;
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
        cnt = tot;
        simt::sync();
    }
}

// The merger of everything up to kBpeTailCap parts (tkz_bpe_long_tail), by one workgroup on the state in LDS: the parts NEVER MOVE -- a slot per part as the
// piece began (a byte each, or what the rounds in global memory left), pr[slot] = rank of (part, next alive part), one alive bit a part, ids beside them or
// in global memory -- and every step applies many of the reference's merges (BytePairEncoder.cs:45-64) at once, each exactly as the reference would get to it:
//   * a BATCH OF PROPOSALS (below): every thread proposes the smallest pair of each of its sub-blocks, looks up what its merge would create, and the proposals
//     that provably cannot be disturbed before their turn are applied -- hundreds a batch on diverse text (a 32 KiB chain of words: 37 batches; it took
//     2,283 when a proposal had to lie below a bound over the WHOLE piece, and ~3,000 rounds of one rank each before that);
//   * a ROUND FOR THE LOWEST RANK (tkz_tail_chain_round) when a batch merged few: a chain of equal pairs (a run of one letter) serialises the proposals.
// What is left is compacted by tkz_bpe_long_tail_emit.
constexpr int kTailBlock = 32;
constexpr uint32_t kTailDead = 0x80000000u;
// IDS_LDS: the ids of the parts are in LDS beside pr[] (<= kBpeLongLds parts).  Otherwise (<= kBpeTailCap parts: only pr[] fits LDS): ids[] -- global
// memory -- holds the ids the parts had when the tail began, and a part that has merged since keeps its id in the slot behind it, which died with its
// first merge and stays dead (kTailDead | id).
template <bool IDS_LDS>
TKZ_DEV uint32_t tkz_tail_id(const int32_t* ids, const int32_t* pr, const uint32_t* alive, int cnt, int x) {
    if (IDS_LDS) return (uint32_t)ids[x];
    const bool merged = x + 1 < cnt && !((alive[(x + 1) >> 5] >> ((x + 1) & 31)) & 1u);
    return merged ? ((uint32_t)pr[x + 1] & ~kTailDead) : (uint32_t)ids[x];
}
// ONE round of tkz_bpe_long_rounds -- every pair of the LOWEST rank gm at once, leftmost first inside a chain of adjacent candidates (every other one,
// counted from the chain's start), cut after the leftmost merge that creates a pair ranked below gm -- on the tail's state: the parts stay in their slots
// (alive bits, neighbours by bit scans) instead of being compacted.  The tail's batches below serialise on such chains (a run of one letter inside a long
// piece: every proposal waits for the pair on its left, which waits for the one on ITS left ...); the tail calls this when a batch of proposals merged few.
// The position inside a chain is the part's index among the alive parts (a workgroup scan of the alive counts) minus that of the last part before it
// whose pair is not a candidate (an exclusive maximum over the workgroup), exactly as in the rounds.  What a merge needs from its neighbours:
//   merge at j (r: the part it swallows, rr: the one behind, l / ll: the ones before):   id' = gm;  pr'[j] = rank(gm, id(rr)), or rank(gm, gm) when rr merges too  (:58)
//   the part before it: swallowed itself when ll merges; otherwise pr'[l] = rank(id(l), gm)                                                             (:59-62)
// pass 1 looks up L = rank(left part as it will be, gm) and the transient R = rank(gm, id(rr) as it is) of every merge (violations: L or R below gm) and
// leaves L in the merging part's own slot; pass 2a, for the merges at or before the cut, computes pr'[j] (the L of the merge at rr, or R) into r's slot --
// r dies in this round, nobody reads its slot -- and decides, while the alive bits are still whole, whether l survives; pass 2b writes.
// s_mm: one word a thread (the merge flags of every block).  Returns the number of merges applied (the same in every thread); ends with a barrier.
template <bool IDS_LDS>
TKZ_DEV int tkz_tail_chain_round(const TkzTables& T, int cnt, int32_t* ids, int32_t* pr, uint32_t* alive, uint32_t* s_mm, uint32_t gm) {
    const int tid = simt::tid(), blk = tid;
    const int nw = (cnt + 31) >> 5;
    const bool owner = blk < nw;
    const uint32_t aw = owner ? alive[blk] : 0u;
    uint32_t cm = 0;                                             // my alive parts whose pair has rank gm
    if (owner) {
        const uint4* qp = reinterpret_cast<const uint4*>(pr + blk * kTailBlock);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 v = qp[k];
            cm |= ((v.x == gm ? 1u : 0u) | (v.y == gm ? 2u : 0u) | (v.z == gm ? 4u : 0u) | (v.w == gm ? 8u : 0u)) << (4 * k);
        }
        cm &= aw;
    }
    int tot;
    const int base = tkz_block_scan(tkz_popc32(aw), &tot);      // the index of my first alive part among all alive parts
    const uint32_t non = aw & ~cm;
    const int lastNon = non ? base + tkz_popc32(aw & tkz_lowmask32(tkz_msb32(non))) : -1;
    int ln = tkz_block_exclusive_max(lastNon);                   // ... of the last part before my block whose pair is not a candidate
    uint32_t mm = 0;
    {
        int idx = base;
        for (uint32_t b = aw; b; b &= b - 1, ++idx) {
            const int i = tkz_ctz32(b);
            if ((cm >> i) & 1u) { if (((idx - (ln + 1)) & 1) == 0) mm |= 1u << i; }
            else ln = idx;
        }
    }
    s_mm[tid] = mm;
    simt::sync();
    auto next_alive = [&](int x) -> int {                        // the first alive part behind slot x, -1: none
        int w = (x + 1) >> 5;
        uint32_t bits = w < nw ? alive[w] & (0xFFFFFFFFu << ((x + 1) & 31)) : 0u;
        while (!bits && ++w < nw) bits = alive[w];
        return bits ? 32 * w + tkz_ctz32(bits) : -1;
    };
    auto prev_alive = [&](int x) -> int {                        // the last alive part before slot x, -1: none
        int w = x >> 5;
        uint32_t bits = alive[w] & tkz_lowmask32(x & 31);
        while (!bits && --w >= 0) bits = alive[w];
        return bits ? 32 * w + tkz_msb32(bits) : -1;
    };
    auto merges = [&](int x) -> bool { return x >= 0 && ((s_mm[x >> 5] >> (x & 31)) & 1u) != 0; };
    // pass 1: the two re-ranked pairs of every merge; the leftmost merge that creates a pair ranked below gm
    uint32_t firstViol = 0xFFFFFFFFu;
    for (uint32_t b = mm; b; b &= b - 1) {
        const int j = blk * kTailBlock + tkz_ctz32(b);
        const int r = next_alive(j);                             // exists: pr[j] was a rank
        const int rr = next_alive(r);
        const int l = prev_alive(j);
        int32_t L = TKZ_RANK_NONE, R = TKZ_RANK_NONE;
        if (l >= 0) L = tkz_lookup_pair(T, merges(prev_alive(l)) ? gm : tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, l), gm);
        if (rr >= 0) R = tkz_lookup_pair(T, gm, tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, rr));
        if ((L < (int32_t)gm || R < (int32_t)gm) && (uint32_t)j < firstViol) firstViol = (uint32_t)j;
        pr[j] = L;                                               // (a merging part's own pair rank is gm, known: its slot carries L to pass 2)
    }
    const uint32_t istar = tkz_block_min32(firstViol);           // merges beyond istar wait (its barriers publish the slots written above)
    // pass 2a: pr'[j] into r's slot; does l survive?  (alive bits still whole)
    uint32_t wl = 0;
    for (uint32_t b = mm; b; b &= b - 1) {
        const int i = tkz_ctz32(b), j = blk * kTailBlock + i;
        if ((uint32_t)j > istar) continue;
        const int r = next_alive(j), rr = next_alive(r);
        int32_t npr = TKZ_RANK_NONE;
        if (rr >= 0) npr = (merges(rr) && (uint32_t)rr <= istar) ? pr[rr] : tkz_lookup_pair(T, gm, tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, rr));
        pr[r] = npr;
        const int l = prev_alive(j);
        if (l >= 0 && !merges(prev_alive(l))) wl |= 1u << i;
    }
    simt::sync();
    // pass 2b: the writes
    int nm = 0;
    for (uint32_t b = mm; b; b &= b - 1) {
        const int i = tkz_ctz32(b), j = blk * kTailBlock + i;
        if ((uint32_t)j > istar) { pr[j] = (int32_t)gm; continue; }       // cut: the pair waits, with its rank
        const int r = next_alive(j);                             // (only this merge clears r's bit)
        const int32_t L = pr[j], npr = pr[r];
        const int l = ((wl >> i) & 1u) ? prev_alive(j) : -1;     // (a part that survives keeps its bit: the scan finds it whatever the others clear)
        simt::atomic_and(&alive[r >> 5], ~(1u << (r & 31)));     // RemoveAt(j + 1) (:63)
        pr[r] = TKZ_RANK_NONE;
        if (IDS_LDS) ids[j] = (int32_t)gm;
        else pr[j + 1] = (int32_t)(kTailDead | gm);              // (the slot behind j: dead since this part's first merge, r == j + 1 then)
        pr[j] = npr;
        if (l >= 0) pr[l] = L;
        ++nm;
    }
    int merged;
    (void)tkz_block_scan(nm, &merged);                           // (ends with a barrier)
    return merged;
}

// Many merges per round trip, exactly.  Thread b of the workgroup owns block b (32 consecutive slots) and proposes the block's smallest
// pair (rank, then position): its neighbours, their ids and the two pair ranks the merge would create are looked up by all threads at
// once -- one trip to the pair table for up to 1024 merges instead of one each.  Which of them may be applied NOW is decided from
// `bound`_b = the smallest key that could come before anything else once b's proposal has merged: the second smallest pair of the block,
// the two pairs the merge creates.  With tau = the minimum of all bounds, every proposal below tau is -- at its turn in the reference's
// order (BytePairEncoder.cs:47-54) -- the leftmost minimum of the whole piece with the neighbourhood it was looked up in: nothing that is
// not itself such a proposal can precede it.  Two proposals interfere exactly when the part of one is among the two parts behind the other
// (the one it swallows, the one it is paired with next): tau is capped at the later of the two.  The global minimum is always applied.
// Keys: rank << 15 | slot, 64 bits (slots < kBpeTailCap = 2^15: the reference's order -- rank, then position -- exactly; equal keys count
// as "not below").
//
// Several proposals a thread (kTailSubs: one per sub-block of 32 / kTailSubs slots, the thread's bound the minimum over them): a trip to
// the pair table costs the same for four lookups as for one, and a sub-block of 8 slots seldom holds two pairs of the lowest ranks.
//
// tau is LOCAL when the vocabulary's keys are short (max_key_len <= kTailLocalKeyMax, every published vocabulary): the minimum of the bounds of the
// blocks within max_key_len slots of the proposal's neighbourhood only -- one block in a thousand with a low bound no longer holds back the
// other 999.  Why that is exact, for ANY rank table:
// let s = (j, r) be a proposal with key k, l / rr the parts before and behind it, and suppose every block that owns a slot in
// [l - max_key_len, rrr + max_key_len] (rrr: the part after rr) has bound > k.  Everything the reference merges while s is waiting has a key below k
// (it takes the minimum each time, :47-54).  Take the FIRST such merge that touches l, j, r or rr.  Either both its parts are as they are now
// -- then it is (ll, l), (l, j), (r, rr) or (rr, rrr), a pair that exists now: the proposal of its sub-block (then the two proposals meet and
// the later one is capped, as before) or a pair at or above that block's bound > k -- or one of its parts is l or rr, unchanged, and the other
// a part P formed since, ending right before l or starting at rrr.  P is a token: at most max_key_len bytes, so at most that many slots (a slot
// was a part of >= 1 byte when the tail began, and slots never move), and every merge that went into P lies inside the window, below k.  The
// first of THOSE that is not a pair existing now pairs the result of a now-existing pair o (key < k, hence a proposal, since its
// block's bound > k) with (a) an unchanged neighbour -- but those two ranks were looked up and are part of o's bound > k --, or (b) the result
// of another proposal below k next to it -- but two proposals that meet cap the bound of the left one's block at the later of the two, < k.
// Contradiction: nothing touches s's neighbourhood before s's turn, and s merges then exactly as looked up now; and the pairs its merge
// creates lie above k (they are part of its own bound), so nothing that was to come before s is displaced by merging s early.  (The global
// minimum g is applied whatever the bounds say.)  Slots, not bytes, measure the window: it is wider than needed, never narrower.
// scratch: (4 * kTailSubs + 8) * nthreads + 16 * (nthreads / 64) + 16 bytes of LDS behind the alive bits.
#ifndef TKZ_TAIL_SUBS
#define TKZ_TAIL_SUBS 4
#endif
constexpr int kTailSubs = TKZ_TAIL_SUBS, kTailSB = kTailBlock / kTailSubs;
constexpr int kTailChainFew = 8;          // a batch of proposals that merges fewer pairs than this is followed by a round for the lowest rank (tkz_tail_chain_round)
constexpr int kTailLocalKeyMax = 1024;    // (beyond: the window would be most of the piece anyway)
constexpr int kTailPosBits = 15;
static_assert(kTailBlock == 32 && kTailSubs * kTailSB == kTailBlock && (kTailSB & (kTailSB - 1)) == 0, "sub-blocks tile a block of 32 slots");
// WINDOW mode (open_lo / open_hi: round 5): the cnt parts are a WINDOW of a longer piece -- parts exist before slot 0 and / or behind slot cnt - 1 that this
// call knows nothing about.  The local bound then also says which proposals may be applied without knowing them: exactly those whose window [l - reach,
// rrr + reach] lies inside the parts that are here (the unknown blocks count as bound 0: "something smaller may be there").  No global minimum (the window's
// is not the piece's), no rounds for the lowest rank; the call ends when a batch applies nothing.  pr[cnt - 1] must be TKZ_RANK_NONE on entry (the caller
// keeps the last part's pair with the first part behind the window): tkz_bpe_window_sweep.
template <bool IDS_LDS>
TKZ_DEV void tkz_bpe_long_tail(const TkzTables& T, int cnt, int32_t* ids, int32_t* pr, uint32_t* alive, void* scratch, unsigned long long* prof = nullptr,
                               const bool open_lo = false, const bool open_hi = false) {
    const int tid = simt::tid(), lane = simt::lane(), wave = simt::wave(), G = simt::nthreads();
    const int nblk = (cnt + kTailBlock - 1) / kTailBlock, nw = (cnt + 31) >> 5;       // (nblk <= G: cnt <= kBpeTailCap, 1024 threads)
    constexpr uint32_t NONE = (uint32_t)TKZ_RANK_NONE;
    constexpr uint32_t NOPROP = 0xFFFFFFFFu;
    constexpr uint64_t NOKEY = ~0ull;                           // (keys: 64 bits -- ranks go up to TKZ_MAX_RANK = 2^27)
    uint32_t* s_prop = reinterpret_cast<uint32_t*>(scratch);    // the proposal of every sub-block: rank << 5 | slot inside its block of 32, or NOPROP
    uint64_t* s_bound = reinterpret_cast<uint64_t*>(s_prop + G * kTailSubs);
    uint64_t* s_red = s_bound + G;
    auto wave_min_key = [](uint64_t v) -> uint64_t {
        const uint32_t hi = (uint32_t)(v >> 32), mh = simt::wave_min_u32(hi);
        const uint32_t ml = simt::wave_min_u32(hi == mh ? (uint32_t)v : 0xFFFFFFFFu);
        return ((uint64_t)mh << 32) | ml;
    };
    auto make_key = [](uint32_t rank, int slot) -> uint64_t { return rank >= (uint32_t)TKZ_RANK_NONE ? ~0ull : (((uint64_t)rank << kTailPosBits) | (uint64_t)(uint32_t)slot); };
    // (the last part never has a pair: its pr is TKZ_RANK_NONE already; slots beyond cnt are padded so that whole blocks can be read)
    for (int i = cnt + tid; i < nblk * kTailBlock; i += G) pr[i] = TKZ_RANK_NONE;
    for (int w = tid; w < nw; w += G) alive[w] = tkz_lowmask32(cnt - 32 * w);
    if (tid == 0) { reinterpret_cast<int*>(s_red + 2 * (G >> 6))[0] = 0; reinterpret_cast<int*>(s_red + 2 * (G >> 6))[1] = 0; }
    simt::sync();
#ifdef TKZ_TAIL_GLOBAL_TAU       // (development builds: the global tau whatever the vocabulary)
    const bool local = false;
#else
    const bool local = T.max_key_len <= kTailLocalKeyMax;
#endif
    const bool windowed = open_lo || open_hi;                    // (only with the local bound: the caller checks)
    const int reach = T.max_key_len;
    const int blk = tid;
    const bool owner = blk < nblk;
    long long n_batch = 0, n_merge = 0, n_cand = 0, n_chain = 0;
    const long long tq0 = prof ? simt::clock() : 0;
    int* s_cnt = reinterpret_cast<int*>(s_red + 2 * (G >> 6));  // merges of the batch: two counters, taken in turn (the one not in use is reset between two barriers)
    int it = 0;
    bool chain = false;                                          // the next step is a round for the lowest rank (the same in every thread)
    for (;;) {
        if (chain) {
            uint32_t mymin = NONE;
            if (owner) {
                const uint4* qp = reinterpret_cast<const uint4*>(pr + blk * kTailBlock);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const uint4 v = qp[k]; mymin = tkz_min3u(mymin, v.x < v.y ? v.x : v.y, v.z < v.w ? v.z : v.w); }
            }
            const uint32_t gm = tkz_block_min32(mymin);
            if (gm >= NONE) break;                               // minRank == int.MaxValue (:65-68)
            const int merged = tkz_tail_chain_round<IDS_LDS>(T, cnt, ids, pr, alive, s_prop, gm);
            if (prof && tid == 0) { ++n_chain; n_merge += merged; }
            chain = merged >= kTailChainFew;                     // (a run collapsing: stay with the rounds; else back to the proposals)
            continue;
        }
        // ---- every sub-block's smallest pair (leftmost of its rank) and its second smallest ----
        // (a dead slot holds TKZ_RANK_NONE or kTailDead | id: as unsigned values both lie at or above TKZ_RANK_NONE)
        uint32_t m[kTailSubs], s2[kTailSubs];
        int jj[kTailSubs], j2[kTailSubs];
#pragma unroll
        for (int q = 0; q < kTailSubs; ++q) { m[q] = NONE; s2[q] = NONE; jj[q] = 0; j2[q] = 0; }
        if (owner) {
            const uint4* qp = reinterpret_cast<const uint4*>(pr + blk * kTailBlock);
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = qp[k];            // (eight 16-byte reads, requested together)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * k + i, q = e / kTailSB;
                    const uint32_t x = w4[i];
                    if (x < m[q]) { s2[q] = m[q]; j2[q] = jj[q]; m[q] = x; jj[q] = e; } else if (x < s2[q]) { s2[q] = x; j2[q] = e; }
                }
            }
        }
        // (what a proposal needs after the barriers is kept small: its key -- rank and slot --, its bound, r / rr / l (-1: none), the two looked-up ranks)
        uint64_t k1[kTailSubs], bound[kTailSubs];
        int r[kTailSubs], l[kTailSubs], rr[kTailSubs];
        int32_t rkr[kTailSubs], rkl[kTailSubs];
        int wlo = blk, whi = blk;
        bool beyond = false;                                     // (window mode) a proposal of this thread reaches parts that are not here
        uint64_t mykey = NOKEY;
#pragma unroll
        for (int q = 0; q < kTailSubs; ++q) {
            const int jq = blk * kTailBlock + jj[q];
            k1[q] = make_key(m[q], jq);
            const bool cand = k1[q] != NOKEY;
            s_prop[tid * kTailSubs + q] = cand ? ((m[q] << 5) | (uint32_t)jj[q]) : NOPROP;
            mykey = k1[q] < mykey ? k1[q] : mykey;
            // ---- the neighbours of the proposal, and what its merge would create ----
            bound[q] = NOKEY; r[q] = -1; l[q] = -1; rr[q] = -1; rkr[q] = TKZ_RANK_NONE; rkl[q] = TKZ_RANK_NONE;
            if (cand) {
                // r: the part being swallowed (the next one alive after j), rr: the one after it, l: the one before j
                int w = (jq + 1) >> 5;
                uint32_t bits = w < nw ? alive[w] & (0xFFFFFFFFu << ((jq + 1) & 31)) : 0u;
                while (!bits && ++w < nw) bits = alive[w];
                r[q] = 32 * w + tkz_ctz32(bits);                 // exists: pr[j] was a rank
                bits &= bits - 1;
                while (!bits && ++w < nw) bits = alive[w];
                const bool hasr = bits != 0;
                rr[q] = hasr ? 32 * w + tkz_ctz32(bits) : -1;
                if (local) {                                     // the window's right end: max_key_len slots beyond the part after rr (rr, or r, when there is none)
                    int last = hasr ? rr[q] : r[q];
                    bool has_rrr = false;
                    if (hasr) {
                        bits &= bits - 1;
                        while (!bits && ++w < nw) bits = alive[w];
                        if (bits) { last = 32 * w + tkz_ctz32(bits) + reach; has_rrr = true; }
                    }
                    if (open_hi && (!has_rrr || last > cnt - 1)) beyond = true;      // (the part after rr, and max_key_len slots behind it, must be HERE)
                    const int hi = (last < cnt - 1 ? last : cnt - 1) >> 5;
                    whi = hi > whi ? hi : whi;
                }
                w = jq >> 5;
                bits = alive[w] & tkz_lowmask32(jq & 31);
                while (!bits && --w >= 0) bits = alive[w];
                const bool hasl = bits != 0;
                l[q] = hasl ? 32 * w + tkz_msb32(bits) : -1;
                if (local && hasl) { const int lo = (l[q] - reach > 0 ? l[q] - reach : 0) >> 5; wlo = lo < wlo ? lo : wlo; }
                if (open_lo && (!hasl || l[q] - reach < 0)) beyond = true;
                const uint32_t idr = tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, hasr ? rr[q] : 0), idl = tkz_tail_id<IDS_LDS>(ids, pr, alive, cnt, hasl ? l[q] : 0);
                rkr[q] = hasr ? tkz_lookup_pair(T, m[q], idr) : TKZ_RANK_NONE;        // (:58)
                rkl[q] = hasl ? tkz_lookup_pair(T, idl, m[q]) : TKZ_RANK_NONE;        // (:59-62)
                const uint64_t key2 = make_key(s2[q], blk * kTailBlock + j2[q]);
                const uint64_t nkr = make_key((uint32_t)rkr[q], jq), nkl = make_key((uint32_t)rkl[q], hasl ? l[q] : 0);
                bound[q] = key2 < nkr ? key2 : nkr;
                if (nkl < bound[q]) bound[q] = nkl;
            }
        }
        simt::sync();                                            // (every proposal is posted)
        if (tid == 0) s_cnt[(it + 1) & 1] = 0;
        uint64_t mybound = NOKEY;
#pragma unroll
        for (int q = 0; q < kTailSubs; ++q) {
            if (k1[q] != NOKEY) {
                auto meets = [&](int p) {
                    if (p < 0) return;
                    const int f = p / kTailSB;
                    if (f == tid * kTailSubs + q) return;
                    const uint32_t pf = s_prop[f];
                    if (pf != NOPROP && (f / kTailSubs) * kTailBlock + (int)(pf & 31u) == p) {
                        const uint64_t kf = make_key(pf >> 5, p), later = kf > k1[q] ? kf : k1[q];
                        if (later < bound[q]) bound[q] = later;
                    }
                };
                meets(r[q]); meets(rr[q]);
            }
            mybound = bound[q] < mybound ? bound[q] : mybound;
        }
        s_bound[tid] = mybound;
        {
            const uint64_t wg = wave_min_key(mykey), wt = wave_min_key(mybound);
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
        if (local && mykey != NOKEY) {                           // the bounds of the window's blocks only (s_bound: published by the barrier above)
            tau = NOKEY;
            for (int f = wlo; f <= whi; ++f) { const uint64_t bf = s_bound[f]; tau = bf < tau ? bf : tau; }
        }
        if (windowed) { if (beyond) tau = 0; g = NOKEY - 1; }    // (g: a key no proposal has -- the window's minimum is not the piece's)
        {
            int tg = 0, tc = 0;
#pragma unroll
            for (int q = 0; q < kTailSubs; ++q) { tg += tkz_popc64(simt::ballot(k1[q] != NOKEY && (k1[q] == g || k1[q] < tau))); if (prof) tc += tkz_popc64(simt::ballot(k1[q] != NOKEY)); }
            if (lane == 0 && tg) simt::atomic_add(&s_cnt[it & 1], tg);
            if (prof) {                                          // (development builds: how many merges a trip to the pair table buys)
                if (tid == 0) ++n_batch;
                if (lane == 0) { n_merge += tg; n_cand += tc; }
            }
        }
#pragma unroll
        for (int q = 0; q < kTailSubs; ++q) {
            if (k1[q] != NOKEY && (k1[q] == g || k1[q] < tau)) {
                const int jq = (int)(k1[q] & ((1u << kTailPosBits) - 1u));
                const uint32_t mq = (uint32_t)(k1[q] >> kTailPosBits);
                simt::atomic_and(&alive[r[q] >> 5], ~(1u << (r[q] & 31)));   // RemoveAt(j + 1) (:63)  (several threads may clear bits of one word)
                pr[r[q]] = TKZ_RANK_NONE;
                if (IDS_LDS) ids[jq] = (int32_t)mq;              // the merged part carries the rank it was found under ...
                else pr[jq + 1] = (int32_t)(kTailDead | mq);     // ... in the slot behind it (dead since this part's first merge: r == j + 1 then)
                pr[jq] = rkr[q];
                if (l[q] >= 0) pr[l[q]] = rkl[q];
            }
        }
        simt::sync();
        if (windowed) { if (s_cnt[it & 1] == 0) break; }         // nothing applied: what is left waits for the parts beyond the window
        else chain = s_cnt[it & 1] < kTailChainFew;              // few merges: a chain of equal pairs may be holding the proposals back
        ++it;
    }
    if (prof && lane == 0) {
        simt::atomic_add64(&prof[17], (unsigned long long)n_merge); simt::atomic_add64(&prof[18], (unsigned long long)n_cand);
        if (tid == 0) {
            simt::atomic_add64(&prof[16], (unsigned long long)n_batch); simt::atomic_add64(&prof[19], (unsigned long long)n_chain);
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

// lds: kBpeLongLdsBytes bytes of LDS (workgroups of 1024 threads) or null.  A piece of up to kBpeLongLds bytes has its whole state in LDS from the start;
// up to kBpeTailCap parts, its pair ranks (the ids stay in the global arrays); a longer one is first brought down to kBpeTailCap parts by the rounds on the
// global arrays (~150 us a round, ONE rank a round: a run of one letter halves in a round, a diverse piece of that size crawls -- the known cliff of this
// path).  Without LDS (any other workgroup shape): the rounds to the end.
constexpr int kBpeLongLds = 16384;
constexpr int kBpeTailCap = 32768;             // parts whose pair ranks and alive bits, with the tail's scratch, fit the workgroup's LDS (one thread a block of 32)
constexpr int kBpeTailScratch = (4 * kTailSubs + 8) * 1024 + 16 * 16 + 16;  // (tkz_bpe_long_tail: proposals, bounds, the two reductions, the batch's merge count; 1024 threads)
// the workgroup's LDS: the tail's state (pair ranks of kBpeTailCap parts, or ids and pair ranks of kBpeLongLds; alive bits; scratch)
constexpr int kBpeLongLdsBytes = (kBpeTailCap * 4 + kBpeTailCap / 8 + kBpeTailScratch + 15) & ~15;
static_assert(kBpeLongLdsBytes >= 8 * kBpeLongLds + kBpeLongLds / 8 + kBpeTailScratch, "ids and pair ranks of kBpeLongLds parts fit too");
static_assert(kBpeTailCap / 32 <= 1024 && kBpeTailCap <= (1 << kTailPosBits), "one thread a block of 32 slots; a slot index fits the key");
static_assert(kBpeLongLdsBytes + 64 <= 160 * 1024, "gfx950: 160 KB of LDS a workgroup");
// One SWEEP over a piece of more than kBpeTailCap parts (dense arrays in global memory: ids / pr -> idsN / prN): the parts are taken kBpeLongLds at a time
// (windows starting at k * kBpeLongLds - shift), every window is merged by tkz_bpe_long_tail in window mode -- everything the local bound allows without
// knowing what lies beyond the window's ends -- and what is alive afterwards is appended to the new arrays.  Each applied merge is one the reference makes
// (the bound's proof, with the unknown blocks at 0); the parts within ~max_key_len slots of a window's ends wait for the next sweep, whose windows are
// shifted by half a window.  A diverse piece shrinks ~3.5x a sweep: 100 KB of letters is one sweep of 7 windows (~40 batches each) away from the
// ordinary tail -- it was ~10^4 rounds in global memory, one rank a round, seconds.  Returns the new part count.
TKZ_DEV int tkz_bpe_window_sweep(const TkzTables& T, int cnt, const int32_t* ids, const int32_t* pr, int32_t* idsN, int32_t* prN, int32_t* lds, int shift) {
    const int tid = simt::tid(), G = simt::nthreads();
    int32_t* lids = lds;
    int32_t* lpr = lds + kBpeLongLds;
    uint32_t* alive = reinterpret_cast<uint32_t*>(lds + 2 * kBpeLongLds);
    void* scratch = alive + kBpeLongLds / 32;
    int out = 0;
    for (int a = 0; a < cnt;) {
        int e = a == 0 && shift > 0 ? kBpeLongLds - shift : a + kBpeLongLds;
        if (e > cnt) e = cnt;
        const int wn = e - a;
        const bool open_lo = a > 0, open_hi = e < cnt;
        for (int k = tid; k < wn; k += G) { lids[k] = ids[a + k]; lpr[k] = (k == wn - 1 && open_hi) ? TKZ_RANK_NONE : pr[a + k]; }
        const int32_t pr_last = pr[e - 1];                       // the last part's pair with the first part behind the window: neither of them changes
        simt::sync();
        if (wn >= 2) tkz_bpe_long_tail<true>(T, wn, lids, lpr, alive, scratch, nullptr, open_lo, open_hi);
        else { if (tid == 0) alive[0] = 1u; simt::sync(); }
        // what is alive, in order, behind what the windows before left
        const int c0 = (wn + G - 1) / G;
        const int lo0 = tid * c0 < wn ? tid * c0 : wn, hi0 = lo0 + c0 < wn ? lo0 + c0 : wn;
        int mine = 0;
        for (int i = lo0; i < hi0; ++i) mine += (int)((alive[i >> 5] >> (i & 31)) & 1u);
        int tot;
        int o = out + tkz_block_scan(mine, &tot);
        for (int i = lo0; i < hi0; ++i) {
            if (!((alive[i >> 5] >> (i & 31)) & 1u)) continue;
            idsN[o] = lids[i];
            prN[o] = (o == out + tot - 1 && open_hi) ? pr_last : lpr[i];
            ++o;
        }
        simt::sync();
        out += tot;
        a = e;
    }
    return out;
}

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
    int rg = 0;
    auto finish_prof = [&](int tokens) {
        if (prof && tid == 0) {
            const long long t2 = simt::clock();
            simt::atomic_add64(&prof[8], 1ull); simt::atomic_add64(&prof[9], (unsigned long long)(t1 - t0)); simt::atomic_add64(&prof[10], (unsigned long long)(t2 - t1));
            simt::atomic_add64(&prof[11], (unsigned long long)n); simt::atomic_add64(&prof[12], (unsigned long long)tokens);
            simt::atomic_max64(&prof[13], (unsigned long long)(t2 - t0));
            if ((unsigned long long)(t2 - t0) >= prof[13]) { prof[24] = (unsigned long long)n; prof[25] = (unsigned long long)rg; prof[27] = (unsigned long long)tokens; prof[28] = (unsigned long long)(t1 - t0); prof[29] = at(0) | (at(n / 2) << 8) | ((unsigned long long)at(n - 1) << 16); }   // (the slowest piece: racy, development only)
            simt::atomic_add64(&prof[14], (unsigned long long)rg);
        }
    };
    // more parts than the tail's LDS holds: sweeps of windows first (each ~3.5x fewer parts on diverse text), as long as they pay
    if (use_lds && cnt > kBpeTailCap && T.max_key_len <= kTailLocalKeyMax && 4 * T.max_key_len < kBpeLongLds) {
        int32_t* cur_ids = ids; int32_t* cur_pr = pr; int32_t* nxt_ids = idsB; int32_t* nxt_pr = prB;
        for (int sweep = 0; cnt > kBpeTailCap && sweep < 12; ++sweep) {
            const int before = cnt;
            cnt = tkz_bpe_window_sweep(T, cnt, cur_ids, cur_pr, nxt_ids, nxt_pr, lds, (sweep & 1) ? kBpeLongLds / 2 : 0);
            { int32_t* t = cur_ids; cur_ids = nxt_ids; nxt_ids = t; } { int32_t* t = cur_pr; cur_pr = nxt_pr; nxt_pr = t; }
            if (prof) ++rg;
            if (cnt > before - before / 8) break;                // (a run of one letter, a chain of equal pairs: the rounds below halve those)
        }
        // the state is in (cur_ids, cur_pr); the rounds below want it in (ids, pr) with (idsB, prB) free
        if (cur_ids != ids) {
            for (int k = tid; k < cnt; k += G) { ids[k] = cur_ids[k]; pr[k] = cur_pr[k]; }
            simt::sync();
        }
    }
    if (!start_in_lds && (!use_lds || cnt > kBpeTailCap)) {
        int32_t* s1 = s1g; int32_t* idsN = idsB;
        done = tkz_bpe_long_rounds(T, cnt, ids, pr, s1, s2g, idsN, prB, use_lds ? kBpeTailCap : 0, prof ? &rg : nullptr);
    }
    if (prof) t1 = simt::clock();
    if (!done) {                                          // (use_lds, cnt <= kBpeTailCap)
        int32_t* lpr = pr;
        uint32_t* alive = reinterpret_cast<uint32_t*>(lds + 2 * kBpeLongLds);
        if (!start_in_lds) {                              // the pair ranks move into LDS (4 bytes a part), the ids the parts have now stay in global memory
            lpr = lds; alive = reinterpret_cast<uint32_t*>(lds + kBpeTailCap);
            for (int k = tid; k < cnt; k += G) lpr[k] = pr[k];
            simt::sync();
        }
        void* scratch = alive + (start_in_lds ? kBpeLongLds : kBpeTailCap) / 32;
        if (prof && tid == 0) simt::atomic_add64(&prof[7], (unsigned long long)cnt);
        int tot;
        if (start_in_lds) { tkz_bpe_long_tail<true>(T, cnt, ids, lpr, alive, scratch, prof); tot = tkz_bpe_long_tail_emit<true>(cnt, ids, lpr, alive, dst, err); }
        else { tkz_bpe_long_tail<false>(T, cnt, ids, lpr, alive, scratch, prof); tot = tkz_bpe_long_tail_emit<false>(cnt, ids, lpr, alive, dst, err); }
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
