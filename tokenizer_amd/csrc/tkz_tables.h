// tkz_tables.h -- layouts of the device-resident vocabulary tables and the hash functions shared
// by the host builder (tkz_vocab.cpp) and the HIP kernels (tkz_kernels.hip).
//
// The reference keeps ONE structure, Dictionary<byte[],int> with an O(len) hash/compare
// (Tokenizer_C#/TokenizerLib/TikTokenizer.cs:101, Utils/BytePairComparer.cs:8-43) and probes it with
// freshly allocated byte slices (Utils/BytePairEncoder.cs:25-36).  Only its exact-match semantics are
// observable.  On the device the same map is held as three open-addressed tables (all a few MB:
// L2 / Infinity-Cache resident, never an HBM stream):
//
//   SHORT table  keys of 1..12 bytes, the key INLINE in a 16-byte slot: one 16 B gather resolves a
//                whole-piece lookup (TikTokenizer.cs:262) for ~95 % of pieces.
//   LONG table   keys of 13..max_key_len bytes: {hash, rank, blob offset, len} + key bytes in a blob.
//   PAIR table   (rank_left, rank_right) -> rank(left ++ right) for EVERY split of every key into two
//                keys.  Parts in the merge loop are always vocabulary keys (single bytes first, then
//                merged tokens whose rank is by construction the rank of the concatenation), so
//                `ranks.TryGetValue(bytes[a..c])` == PAIR[(rank(bytes[a..b]), rank(bytes[b..c]))]
//                -- a fixed-width probe instead of hashing a variable-length slice.
//                A single byte that is NOT a key gets the pseudo id TKZ_PSEUDO_BASE + byte; it can still
//                take part in merges exactly as in the reference and raises KeyNotFound only if it
//                survives to emission (BytePairEncoder.cs:73).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define TKZ_HD __host__ __device__ __forceinline__
#else
#define TKZ_HD inline
#endif

#define TKZ_RANK_NONE 0x7FFFFFFF            /* int.MaxValue sentinel, BytePairEncoder.cs:23,35 */
#define TKZ_PSEUDO_BASE 0x7FFFFC00          /* ids >= this are "single byte not in vocab" */
#define TKZ_MAX_RANK ((1 << 27) - 1)        /* ranks must be in [0, TKZ_MAX_RANK] */

#define TKZ_SHORT_KEY_MAX 12

// 16 B; rank_len == 0 marks an empty slot (len is 1..12, so a used slot is never 0).
// rank_len = rank | (len << 27): ranks must be < 2^27 (TKZ_MAX_RANK); the largest shipped vocabulary
// (o200k_base) tops out below 2^18.  Larger ranks are rejected at load with TKZ_E_UNSUPPORTED.
struct TkzShortSlot {
    uint32_t k0, k1, k2;      // key bytes, little-endian packed, zero padded
    uint32_t rank_len;
};
#define TKZ_SHORT_RANK_BITS 27
#define TKZ_SHORT_RANK_MASK ((1u << TKZ_SHORT_RANK_BITS) - 1u)

struct TkzLongSlot {    // 16 B; len == 0 marks an empty slot
    uint32_t hash;      // full 32-bit hash (cheap reject before touching the blob)
    int32_t rank;
    uint32_t off;       // byte offset of the key in the blob
    uint32_t len;
};

struct TkzPairSlot {    // 16 B; valid == 0 marks an empty slot
    uint32_t a, b;
    int32_t rank;
    uint32_t valid;
};

struct TkzTables {      // device pointers + masks, passed to kernels by value
    const TkzShortSlot* short_slots; uint32_t short_mask;
    const TkzLongSlot* long_slots;   uint32_t long_mask;
    const uint8_t* long_blob;
    const TkzPairSlot* pair_slots;   uint32_t pair_mask;
    const int32_t* byte_rank;        // [256] rank of the single byte, or TKZ_PSEUDO_BASE + b
    const int32_t* bytepair_rank;    // [65536] PAIR restricted to two single bytes (direct index), TKZ_RANK_NONE if absent
    const uint8_t* bmp_class;        // [65536] class of each BMP code unit (see tkz_classes.h)
    int32_t max_key_len;
    int32_t pattern;
};

TKZ_HD uint32_t tkz_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}
// hash of a key given as zero-padded little-endian dwords
TKZ_HD uint32_t tkz_hash_short(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len) {
    uint32_t h = tkz_mix32(k0 + 0x9E3779B9u * len);
    h = tkz_mix32(h ^ (k1 * 0x85EBCA6Bu));
    h = tkz_mix32(h + (k2 * 0xC2B2AE35u));
    return h;
}
// streaming form for long keys: feed ceil(len/4) zero-padded dwords in order
TKZ_HD uint32_t tkz_hash_long_init(uint32_t len) { return 0x2545F491u ^ (len * 0x9E3779B9u); }
TKZ_HD uint32_t tkz_hash_long_step(uint32_t h, uint32_t w) { return tkz_mix32(h ^ w) + 0x632BE5ABu; }
TKZ_HD uint32_t tkz_hash_pair(uint32_t a, uint32_t b) {
    return tkz_mix32(a * 0x9E3779B9u ^ tkz_mix32(b + 0x7F4A7C15u));
}
