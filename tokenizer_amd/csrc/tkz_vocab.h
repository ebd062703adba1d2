// tkz_vocab.h -- host side of the vocabulary: .tiktoken parsing and the builder of the device
// table images (layouts in tkz_tables.h).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "tkz_tables.h"

namespace tkz {

// Unicode class codes stored in TkzTables::bmp_class (generated table, unicode13_classes.inc)
enum : uint8_t { UC_OTHER = 0, UC_LU = 1, UC_LL = 2, UC_LT = 3, UC_LM = 4, UC_LO = 5, UC_M = 6, UC_N = 7, UC_WS = 8 };

struct Vocab {
    // insertion-ordered keys + ranks (a repeated key overwrites its rank, TikTokenizer.cs:125)
    std::vector<std::string> keys;
    std::vector<int32_t> ranks;
    std::unordered_map<std::string, int32_t> index;   // key -> position in keys/ranks
    int32_t max_key_len = 0;

    // host images of the device tables
    std::vector<TkzShortSlot> short_slots;    // two per bucket
    std::vector<TkzMidSlot> mid_slots;
    std::vector<TkzLongSlot> long_slots;
    std::vector<uint8_t> long_blob;
    std::vector<TkzPairSlot> pair_slots;
    std::vector<int32_t> byte_rank;       // 256
    std::vector<int32_t> bytepair_rank;   // 65536
    int64_t pair_entries = 0;
    uint32_t short_seed = 0, mid_seed = 0, pair_seed = 0;   // seeds under which the cuckoo insertion succeeded
    bool pair_compact = false;                // pair_slots are two-entry buckets of 8-byte entries (tkz_tables.h)

    bool lookup(const std::string& k, int32_t* rank) const {
        auto it = index.find(k);
        if (it == index.end()) return false;
        *rank = ranks[it->second];
        return true;
    }
};

// The SHORT (keys of 1..12 bytes) and MID (13..28) whole-key tables from (key, value) items: value = the key's rank, or -- for a promoted piece
// (tkz_tables.h) -- its promo code.  Items in order of frequency, most frequent first (the builder keeps the first sixteenth of the SHORT keys
// in their first bucket, which is all k_probe fetches for them); longer and empty keys are skipped.
struct KeyItem { std::string key; uint32_t value; };
void build_key_tables(const std::vector<KeyItem>& items, std::vector<TkzShortSlot>* short_slots, uint32_t* short_seed,
                      std::vector<TkzMidSlot>* mid_slots, uint32_t* mid_seed);

// Parses a .tiktoken image.  Returns 0 (TKZ_OK) or a negative tkz_status; msg receives a description.
int parse_tiktoken(const uint8_t* file, size_t n, Vocab* out, std::string* msg);
// Builds every table image (requires parse_tiktoken to have succeeded).
int build_tables(Vocab* v, std::string* msg);
// 65536-entry BMP class table (shared by all vocabularies)
const std::vector<uint8_t>& bmp_class_table();

}  // namespace tkz
