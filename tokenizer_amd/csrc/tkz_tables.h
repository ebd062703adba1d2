// tkz_tables.h -- layouts of the device-resident vocabulary tables, the hash functions shared by the
// host builder (tkz_vocab.cpp) and the HIP kernels (tkz_kernels.hip), and the probe functions.
//
// The reference keeps ONE structure, Dictionary<byte[],int> with an O(len) hash/compare
// (Tokenizer_C#/TokenizerLib/TikTokenizer.cs:101, Utils/BytePairComparer.cs:8-43) and probes it with
// freshly allocated byte slices (Utils/BytePairEncoder.cs:25-36).  Only its exact-match semantics are
// observable.  On the device the same map is held as four hash tables, a few MB in total.  They are sized for the 4 MiB L2
// of ONE XCD (every XCD keeps its own copy of the hot lines): with cl100k-sized vocabularies the tables of round 1
// (power-of-two sizes at a load of 0.3-0.4: 9 MB) missed L2 on most gathers and the encode kernel fetched 4x its input
// from the fabric.  Every table is a CUCKOO table with an arbitrary (not power-of-two) size -- slot = mulhi(hash, size) --
// so a probe is a fixed number of independent 16-byte gathers issued together, never a chain: on a SIMT machine a probe
// sequence costs the MAXIMUM over the 64 lanes, so a bounded probe count matters more than the mean.
//
//   SHORT table  keys of 1..12 bytes, the key INLINE in a 16-byte slot {k0,k1,k2, rank|len<<27}; buckets of two slots,
//                two candidate buckets per key ((2,2) cuckoo, load 0.8): four 16-byte gathers resolve the whole-piece
//                lookup (TikTokenizer.cs:262) for ~90 % of pieces.
//   MID table    keys of 13..28 bytes inline in a 32-byte slot {k0..k6, rank|(len-12)<<27}; two candidate slots ((2,1)
//                cuckoo, load 0.45): the same four 16-byte gathers.  (Round 1 kept these keys behind a linear-probing
//                table with a byte-by-byte compare against a blob: with a 100 k-key vocabulary a third of all pieces
//                are longer than 12 bytes and that lookup was 55 % of the encode kernel.)
//   LONG table   keys of 29..max_key_len bytes (runs of white space / punctuation: a few hundred keys):
//                {hash, rank, blob offset, len} + key bytes in a blob, linear probing.
//   PAIR table   (id_left, id_right) -> rank(left ++ right) for EVERY split of every key into two
//                keys.  Parts in the merge loop are always vocabulary keys (single bytes first, then
//                merged tokens whose id is by construction the rank of the concatenation), so
//                `ranks.TryGetValue(bytes[a..c])` == PAIR[(id(bytes[a..b]), id(bytes[b..c]))]
//                -- a fixed-width probe instead of hashing a variable-length slice.
//                A single byte that is NOT a key gets the pseudo id TKZ_PSEUDO_BASE + byte; it can still
//                take part in merges exactly as in the reference and raises KeyNotFound only if it
//                survives to emission (BytePairEncoder.cs:73).
#pragma once
#include <stdint.h>

#include "tkz_simt.h"

#define TKZ_RANK_NONE 0x7FFFFFFF            /* int.MaxValue sentinel, BytePairEncoder.cs:23,35 */
#define TKZ_PSEUDO_BASE 0x7FFFFC00          /* ids >= this are "single byte not in vocab" */
#define TKZ_MAX_RANK ((1 << 27) - 2)        /* ranks must be in [0, TKZ_MAX_RANK]: (rank << 5 | 31) stays below the NOKEY sentinel */

#define TKZ_SHORT_KEY_MAX 12
#define TKZ_MID_KEY_MAX 28

// 16 B; rank_len == 0 marks an empty slot (len is 1..12, so a used slot is never 0).
// rank_len = rank | (len << 27): ranks must be < 2^27 (TKZ_MAX_RANK); the largest vocabulary the
// reference names (o200k_base) tops out below 2^18.  Larger ranks are rejected at load (TKZ_E_UNSUPPORTED).
struct alignas(16) TkzShortSlot {
    uint32_t k0, k1, k2;      // key bytes, little-endian packed, zero padded
    uint32_t rank_len;
};
#define TKZ_SHORT_RANK_BITS 27
#define TKZ_SHORT_RANK_MASK ((1u << TKZ_SHORT_RANK_BITS) - 1u)

// 32 B: keys of 13..28 bytes; rank_len = rank | (len - 12) << 27, never 0 for a used slot
struct alignas(16) TkzMidSlot {
    uint32_t k[7];
    uint32_t rank_len;
};

struct alignas(16) TkzLongSlot {    // len == 0 marks an empty slot
    uint32_t hash;      // full 32-bit hash (cheap reject before touching the blob)
    int32_t rank;
    uint32_t off;       // byte offset of the key in the blob
    uint32_t len;
};

// Wide form: one entry per 16-byte slot; valid == 0 marks an empty slot.
// Compact form (every id below TKZ_PAIR_CID_LIMIT, i.e. every published vocabulary): a 16-byte slot is a BUCKET of two 8-byte
// entries  valid:1 | rank:21 | b:21 | a:21  (a, b as compact ids: a rank, or TKZ_PAIR_CID_LIMIT + byte for a single byte that is
// not a key); the four words overlay a/b/rank/valid.  Two entries per bucket let the cuckoo table run at a load of 0.8 instead
// of 0.4: a quarter of the bytes for the same two 16-byte gathers per probe, so that the table stays in one XCD's L2.
struct alignas(16) TkzPairSlot {
    uint32_t a, b;
    int32_t rank;
    uint32_t valid;
};
#define TKZ_PAIR_CID_LIMIT 0x1FFF00u
TKZ_HD uint32_t tkz_pair_cid(uint32_t id) { return id >= (uint32_t)TKZ_PSEUDO_BASE ? id - (uint32_t)TKZ_PSEUDO_BASE + TKZ_PAIR_CID_LIMIT : id; }
TKZ_HD uint64_t tkz_pair_key42(uint32_t a, uint32_t b) { return (uint64_t)tkz_pair_cid(a) | ((uint64_t)tkz_pair_cid(b) << 21); }

// The piece memo: the device form of the reference's LRUCache (LRUCache.cs; used at TikTokenizer.cs:254,270 to skip BytePairEncode for a
// piece seen before).  A direct-mapped table of 32-byte slots keyed by the piece's bytes (<= 16, zero padded) holding its <= 4 tokens:
//   key   k0..k3
//   val.x 0 = empty | 0xFFFFFFFF = being written | VALID (bit 31) | (count - 1) << 29 | token 0        (tokens are ranks < 2^27)
//   val.y VALID | (len - 1) << 27 | token 1,   val.z VALID | token 2,   val.w VALID | token 3
//   (VALID in every value word: a reader accepts a value only when all four words are the new ones, so the protocol does not lean on a
//    16-byte store or load being atomic across CUs / XCDs)
// Only k_merge_short reads and writes it: pieces that missed the vocabulary tables look themselves up before they are merged, and merged
// pieces of <= 4 tokens claim their slot if it is EMPTY (compare-and-swap on val.x; an entry never changes once it is valid, so a hit in
// one kernel cannot be invalidated by another).  A pure memo: results are identical with and without it.
struct alignas(16) TkzMemoSlot { uint32_t k[4]; uint32_t v[4]; };
constexpr uint32_t kMemoValid = 0x80000000u, kMemoBusy = 0xFFFFFFFFu;
// slots per bucket (one 64-byte line for 2): a piece may sit in any slot of its bucket and takes the first free one
#ifndef TKZ_MEMO_WAYS
#define TKZ_MEMO_WAYS 1
#endif
constexpr uint32_t kMemoWays = TKZ_MEMO_WAYS;

// PROMOTED pieces (round 5).  The memo answers a missed piece in k_merge_short: list entry + quad + a 32-byte slot + the answer, per piece and
// batch, for text whose hot pieces are not keys (a vocabulary that has not seen the text: 7 of 8 short misses are memo hits).  Those answers
// never change, so the host PROMOTES the hottest memo entries into the SHORT / MID tables themselves (tkz_api.cpp: promote_from_memo): the slot
// of a promoted piece carries, in place of a rank,
//     kPromoFlag | (count - 1) << kPromoCntShift | index          (count <= 4; index: a quad of tokens in TkzTables::promo)
// k_probe needs no change -- it writes the slot's value into the piece's record like any rank --, the merge kernels never see the piece, and
// k_place, which already reads every record, takes the count from the record and the tokens with ONE 16-byte gather from the (cache-resident)
// promo array.  Ranks are < 2^26 whenever anything is promoted (every published vocabulary is below 2^18; checked by the host).
constexpr uint32_t kPromoFlag = 1u << 26, kPromoIdxMask = (1u << 22) - 1u;
constexpr int kPromoCntShift = 22;
constexpr uint32_t kPromoMaxEntries = 1u << 22;
constexpr uint32_t kMemoHitsSat = 64;          // a learning batch counts a slot's (sampled) hits up to here

struct TkzTables {      // device pointers + sizes, passed to kernels by value
    TkzMemoSlot* memo; uint32_t memo_n;                                          // piece memo (null / 0: none)
    uint32_t* memo_hits; uint32_t memo_hits_sparse;                              // null, or (a LEARNING batch) memo_n counters: hits per slot (k_merge_short); sparse: sampled, one hit in 512
    const uint4* promo; uint32_t promo_n;                                        // token quads of the promoted pieces (null / 0: none)
    const TkzShortSlot* short_slots; uint32_t short_nb; uint32_t short_seed;    // short_nb buckets of two slots
    const TkzMidSlot* mid_slots;     uint32_t mid_ns;   uint32_t mid_seed;      // mid_ns slots
    const TkzLongSlot* long_slots;   uint32_t long_mask;
    const uint8_t* long_blob;
    const TkzPairSlot* pair_slots;   uint32_t pair_n; uint32_t pair_seed; uint32_t pair_compact;   // pair_n slots; compact: a slot is a two-entry bucket
    const int32_t* byte_rank;        // [256] id of the single byte: its rank, or TKZ_PSEUDO_BASE + b
    const int32_t* bytepair_rank;    // [65536] rank of the two-byte key (b0<<8|b1), TKZ_RANK_NONE if absent
    const uint8_t* bmp_class;        // [65536] Unicode class of each BMP code unit (tkz_classes.h)
    int32_t max_key_len;
    int32_t pattern;
    int32_t max_rank;                // largest rank in the vocabulary (kernels pack rank and position into one word when it is small)
};

// Decoder (TikTokenizer.cs:81 `Decoder = Encoder.ToDictionary(kvp => kvp.Value, kvp => kvp.Key)` + SpecialTokensDecoder): id -> bytes.
// dense: entry k IS id k (every published vocabulary: ids are dense from 0); otherwise `ids` is sorted and searched.
struct TkzDecodeTable {
    const uint32_t* off;      // [n + 1] byte offsets into blob; an id without bytes has off[k] == off[k + 1]
    const uint8_t* blob;
    const int32_t* ids;       // [n] sorted ids (sparse form only)
    int64_t n;
    int32_t dense;
};

TKZ_HD uint32_t tkz_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}
// The table hashes are deliberately CHEAP (one multiply per key dword + one finishing multiply): k_probe issues them for every
// piece of the corpus and is bound by VALU issue, not by memory; the builder verifies every insertion and retries with another
// seed, so a weak hash can cost build time but never correctness.  Slots are taken from the HIGH bits (tkz_mulhi).
TKZ_HD uint32_t tkz_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
TKZ_HD uint32_t tkz_fmix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; return h; }
// slot of a hash in a table of n slots (n arbitrary, h uniform over 32 bits): one v_mul_hi_u32
TKZ_HD uint32_t tkz_mulhi(uint32_t h, uint32_t n) { return (uint32_t)(((uint64_t)h * (uint64_t)n) >> 32); }
// the second hash of every cuckoo table, derived from the first
TKZ_HD uint32_t tkz_hash_second(uint32_t h1) { const uint32_t h = h1 * 0x846ca68bU + 0x165667B1u; return h ^ (h >> 15); }
// a short key given as zero-padded little-endian dwords
TKZ_HD uint32_t tkz_hash_short(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint32_t seed) {
    uint32_t h = (k0 ^ tkz_rotl(k1, 13) ^ tkz_rotl(k2, 23) ^ (len << 27) ^ seed) * 0x9E3779B1u;     // (32-bit multiplies run at a quarter of the VALU rate)
    return tkz_fmix(h ^ (h >> 15) ^ k1);
}
TKZ_HD uint32_t tkz_hash_short2(uint32_t h1) { return tkz_hash_second(h1); }
TKZ_HD uint32_t tkz_hash_memo(const uint32_t* k, uint32_t len) {
    const uint32_t h = (k[0] ^ tkz_rotl(k[1], 13) ^ tkz_rotl(k[2], 23) ^ tkz_rotl(k[3], 7) ^ (len << 27)) * 0x9E3779B1u;
    return tkz_fmix(h ^ (h >> 15) ^ k[1] ^ k[3]);
}
// a key of 13..28 bytes given as seven zero-padded little-endian dwords
TKZ_HD uint32_t tkz_hash_mid(const uint32_t* k, uint32_t len, uint32_t seed) {
    uint32_t h = (k[0] + seed) * 0x9E3779B1u;
    h ^= tkz_rotl(k[1] * 0x85EBCA77u, 15);
    h ^= tkz_rotl(k[2] * 0xC2B2AE3Du, 7);
    h += tkz_rotl(k[3] * 0x27D4EB2Fu, 11);
    h ^= tkz_rotl(k[4] * 0x165667B1u, 19);
    h += tkz_rotl(k[5] * 0x9E3779B9u, 3);
    h ^= tkz_rotl(k[6] * 0x85EBCA6Bu, 23);
    return tkz_fmix(h + len * 0x27D4EB2Fu);
}
// streaming form for long keys: feed ceil(len/4) zero-padded dwords in order
TKZ_HD uint32_t tkz_hash_long_init(uint32_t len) { return 0x2545F491u ^ (len * 0x9E3779B9u); }
TKZ_HD uint32_t tkz_hash_long_step(uint32_t h, uint32_t w) { return tkz_mix32(h ^ w) + 0x632BE5ABu; }
TKZ_HD uint32_t tkz_hash_pair(uint32_t a, uint32_t b, uint32_t seed) {
    return tkz_fmix(((a + seed) * 0x9E3779B1u) ^ tkz_rotl(b * 0x85EBCA77u, 13));
}
TKZ_HD uint32_t tkz_hash_pair2(uint32_t h1) { return tkz_hash_second(h1); }

// ---- probes ---------------------------------------------------------------------------------------
TKZ_HD uint4 tkz_load16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

// the two candidate BUCKETS of a short key (slot index of the first of the bucket's two slots)
TKZ_HD void tkz_short_slots(const TkzTables& T, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint32_t* s1, uint32_t* s2) {
    const uint32_t h = tkz_hash_short(k0, k1, k2, len, T.short_seed);
    *s1 = 2u * tkz_mulhi(h, T.short_nb); *s2 = 2u * tkz_mulhi(tkz_hash_short2(h), T.short_nb);
}
// ... in two steps: the FIRST candidate bucket (the builder puts the keys of the lowest ranks -- the frequent ones -- there whenever it
// can), and from its hash the second one, which k_probe only fetches for the lanes the first bucket did not settle: a scattered
// gather costs the memory pipeline one request per lane, and those requests are what k_probe runs out of
TKZ_HD uint32_t tkz_short_slot_first(const TkzTables& T, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint32_t* h) {
    *h = tkz_hash_short(k0, k1, k2, len, T.short_seed);
    return 2u * tkz_mulhi(*h, T.short_nb);
}
TKZ_HD uint32_t tkz_short_slot_second(const TkzTables& T, uint32_t h) { return 2u * tkz_mulhi(tkz_hash_short2(h), T.short_nb); }
TKZ_HD bool tkz_short_slot_is(uint4 v, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len) {
    return v.x == k0 && v.y == k1 && v.z == k2 && (v.w >> TKZ_SHORT_RANK_BITS) == len;
}
// Encoder.TryGetValue(piece) for a piece of 1..12 bytes, given the contents of its two candidate buckets (TikTokenizer.cs:262)
TKZ_HD int32_t tkz_match_short(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint4 a0, uint4 a1, uint4 b0, uint4 b1) {
    if (tkz_short_slot_is(a0, k0, k1, k2, len)) return (int32_t)(a0.w & TKZ_SHORT_RANK_MASK);
    if (tkz_short_slot_is(a1, k0, k1, k2, len)) return (int32_t)(a1.w & TKZ_SHORT_RANK_MASK);
    if (tkz_short_slot_is(b0, k0, k1, k2, len)) return (int32_t)(b0.w & TKZ_SHORT_RANK_MASK);
    if (tkz_short_slot_is(b1, k0, k1, k2, len)) return (int32_t)(b1.w & TKZ_SHORT_RANK_MASK);
    return TKZ_RANK_NONE;
}
TKZ_HD int32_t tkz_match_short2(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint4 a0, uint4 a1) {     // one bucket
    if (tkz_short_slot_is(a0, k0, k1, k2, len)) return (int32_t)(a0.w & TKZ_SHORT_RANK_MASK);
    if (tkz_short_slot_is(a1, k0, k1, k2, len)) return (int32_t)(a1.w & TKZ_SHORT_RANK_MASK);
    return TKZ_RANK_NONE;
}
// the same, without branches (k_probe runs it for every piece of the corpus): a slot matches iff the OR of the differences is zero.  A
// length outside 1..12 matches nothing (an empty slot has length 0, but its key dwords are zero too: callers pass a zero key only with len 0...
// so len == 0 is excluded explicitly).
TKZ_HD int32_t tkz_match_short2x(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint4 a0, uint4 a1) {
    const uint32_t d0 = (a0.x ^ k0) | (a0.y ^ k1) | (a0.z ^ k2) | ((a0.w >> TKZ_SHORT_RANK_BITS) ^ len);
    const uint32_t d1 = (a1.x ^ k0) | (a1.y ^ k1) | (a1.z ^ k2) | ((a1.w >> TKZ_SHORT_RANK_BITS) ^ len);
    const uint32_t r = d0 == 0 ? a0.w : a1.w;
    return ((d0 == 0 || d1 == 0) && len != 0) ? (int32_t)(r & TKZ_SHORT_RANK_MASK) : TKZ_RANK_NONE;
}
TKZ_HD int32_t tkz_lookup_short(const TkzTables& T, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len) {
    uint32_t s1, s2;
    tkz_short_slots(T, k0, k1, k2, len, &s1, &s2);
    return tkz_match_short(k0, k1, k2, len, tkz_load16(&T.short_slots[s1]), tkz_load16(&T.short_slots[s1 + 1]),
                           tkz_load16(&T.short_slots[s2]), tkz_load16(&T.short_slots[s2 + 1]));
}
// the two candidate slots of a 13..28-byte key, and the match given both halves of both
TKZ_HD void tkz_mid_slots(const TkzTables& T, const uint32_t* k, uint32_t len, uint32_t* s1, uint32_t* s2) {
    const uint32_t h = tkz_hash_mid(k, len, T.mid_seed);
    *s1 = tkz_mulhi(h, T.mid_ns); *s2 = tkz_mulhi(tkz_hash_short2(h), T.mid_ns);
}
TKZ_HD bool tkz_mid_slot_is(uint4 lo, uint4 hi, const uint32_t* k, uint32_t len) {
    return lo.x == k[0] && lo.y == k[1] && lo.z == k[2] && lo.w == k[3] && hi.x == k[4] && hi.y == k[5] && hi.z == k[6] &&
           (hi.w >> TKZ_SHORT_RANK_BITS) == len - 12u;
}
TKZ_HD int32_t tkz_match_mid(const uint32_t* k, uint32_t len, uint4 a0, uint4 a1, uint4 b0, uint4 b1) {
    if (tkz_mid_slot_is(a0, a1, k, len)) return (int32_t)(a1.w & TKZ_SHORT_RANK_MASK);
    if (tkz_mid_slot_is(b0, b1, k, len)) return (int32_t)(b1.w & TKZ_SHORT_RANK_MASK);
    return TKZ_RANK_NONE;
}
TKZ_HD int32_t tkz_lookup_mid(const TkzTables& T, const uint32_t* k, uint32_t len) {
    uint32_t s1, s2;
    tkz_mid_slots(T, k, len, &s1, &s2);
    const uint4* m = reinterpret_cast<const uint4*>(T.mid_slots);
    return tkz_match_mid(k, len, tkz_load16(&m[2 * s1]), tkz_load16(&m[2 * s1 + 1]), tkz_load16(&m[2 * s2]), tkz_load16(&m[2 * s2 + 1]));
}

// ranks.TryGetValue(left ++ right) through the ids of the two parts (BytePairEncoder.cs:25-36)
TKZ_HD void tkz_pair_slots(const TkzTables& T, uint32_t a, uint32_t b, uint32_t* s1, uint32_t* s2) {
    const uint32_t h = tkz_hash_pair(a, b, T.pair_seed);
    *s1 = tkz_mulhi(h, T.pair_n); *s2 = tkz_mulhi(tkz_hash_pair2(h), T.pair_n);
}
// a slot of the pair table by a 32-BIT byte offset from the (wave-uniform) table base: the load takes the base from scalar registers and the
// offset from one VGPR, instead of a 64-bit address made of a 64-bit shift, two masks and a 64-bit add per slot (the table is far below 4 GB:
// tkz_vocab.cpp refuses more than 2^27 slots)
TKZ_HD uint4 tkz_load_pair_slot(const TkzTables& T, uint32_t slot) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(T.pair_slots) + (uint32_t)(slot << 4));
}
TKZ_HD int32_t tkz_match_pair(const TkzTables& T, uint32_t a, uint32_t b, uint4 v1, uint4 v2) {
    if (T.pair_compact) {
        // entry = valid:1 | rank:21 | b:21 | a:21 as two dwords: lo = a | b << 21, hi = b >> 11 | rank << 10 | valid << 31.  Branch-free and in
        // 32-bit operations (the early-return form compiled into four nested EXEC-mask branches with 64-bit compares: the merge kernels are
        // bound by instruction issue, and a divergent branch is instructions on the scalar unit as well): an entry matches when
        // (lo ^ want_lo) | ((hi & 0x800003FF) ^ want_hi) is zero; at most one of the four does (the builder never stores a key twice).
        const uint32_t ca = tkz_pair_cid(a), cb = tkz_pair_cid(b);
        const uint32_t wlo = ca | (cb << 21), whi = (cb >> 11) | 0x80000000u, mhi = 0x800003FFu;
        const uint32_t d0 = (v1.x ^ wlo) | ((v1.y & mhi) ^ whi), d1 = (v1.z ^ wlo) | ((v1.w & mhi) ^ whi);
        const uint32_t d2 = (v2.x ^ wlo) | ((v2.y & mhi) ^ whi), d3 = (v2.z ^ wlo) | ((v2.w & mhi) ^ whi);
        const uint32_t hit = (d0 == 0 ? v1.y : 0u) | (d1 == 0 ? v1.w : 0u) | (d2 == 0 ? v2.y : 0u) | (d3 == 0 ? v2.w : 0u);
        return hit ? (int32_t)((hit >> 10) & 0x1FFFFFu) : TKZ_RANK_NONE;      // (a matching entry has its valid bit set: hit != 0)
    }
    if (v1.w != 0 && v1.x == a && v1.y == b) return (int32_t)v1.z;
    if (v2.w != 0 && v2.x == a && v2.y == b) return (int32_t)v2.z;
    return TKZ_RANK_NONE;
}
TKZ_HD int32_t tkz_lookup_pair(const TkzTables& T, uint32_t a, uint32_t b) {
    uint32_t s1, s2;
    tkz_pair_slots(T, a, b, &s1, &s2);
    return tkz_match_pair(T, a, b, tkz_load_pair_slot(T, s1), tkz_load_pair_slot(T, s2));
}

// Encoder.TryGetValue(piece) for a piece of 29..max_key_len bytes; `at(i)` yields byte i of the piece.
template <class ByteAt>
TKZ_HD int32_t tkz_lookup_long(const TkzTables& T, ByteAt at, uint32_t len) {
    if ((int32_t)len > T.max_key_len || len <= TKZ_MID_KEY_MAX) return TKZ_RANK_NONE;
    uint32_t h = tkz_hash_long_init(len);
    for (uint32_t off = 0; off < len; off += 4) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4 && off + j < len; ++j) w |= (uint32_t)at(off + j) << (8 * j);
        h = tkz_hash_long_step(h, w);
    }
    uint32_t s = h & T.long_mask;
    for (;;) {
        const uint4 v = tkz_load16(&T.long_slots[s]);   // {hash, rank, off, len}
        if (v.w == 0) return TKZ_RANK_NONE;
        if (v.x == h && v.w == len) {
            const uint8_t* k = T.long_blob + v.z;
            uint32_t i = 0;
            while (i < len && k[i] == at(i)) ++i;
            if (i == len) return (int32_t)v.y;
        }
        s = (s + 1) & T.long_mask;
    }
}

// Encoder.TryGetValue(piece) for a piece of any length (TikTokenizer.cs:262), one lane, no batching: the slow general form
template <class ByteAt>
TKZ_HD int32_t tkz_lookup_any(const TkzTables& T, ByteAt at, uint32_t len) {
    if (len == 0) return TKZ_RANK_NONE;
    if (len <= TKZ_MID_KEY_MAX) {
        uint32_t k[7] = {0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = 0; i < len; ++i) k[i >> 2] |= (uint32_t)at(i) << (8 * (i & 3));
        return len <= TKZ_SHORT_KEY_MAX ? tkz_lookup_short(T, k[0], k[1], k[2], len) : tkz_lookup_mid(T, k, len);
    }
    return tkz_lookup_long(T, at, len);
}
