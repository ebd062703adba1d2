// tkz_vocab.cpp -- .tiktoken loader and device-table builder (host C++).
//
// Replaces TikTokenizer.LoadTikTokenBpe + the rank-collision check of Init
// (Tokenizer_C#/TokenizerLib/TikTokenizer.cs:99-139, :74-91): same file format, same failure modes,
// reported as status codes instead of exceptions.
#include "tkz_vocab.h"
#include "tkz_classes.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/tkz.h"

namespace tkz {

namespace {

int b64val(int c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}

// Convert.FromBase64String semantics for a field without embedded white space.
bool b64decode(const uint8_t* s, size_t n, std::string* out) {
    out->clear();
    if (n % 4) return false;
    for (size_t i = 0; i < n; i += 4) {
        int v[4], pad = 0;
        for (int k = 0; k < 4; ++k) {
            int c = s[i + k];
            if (c == '=') {
                if (i + 4 != n || k < 2) return false;
                v[k] = 0; ++pad;
            } else {
                if (pad) return false;
                v[k] = b64val(c);
                if (v[k] < 0) return false;
            }
        }
        uint32_t w = (uint32_t(v[0]) << 18) | (uint32_t(v[1]) << 12) | (uint32_t(v[2]) << 6) | uint32_t(v[3]);
        out->push_back(char(w >> 16));
        if (pad < 2) out->push_back(char(w >> 8));
        if (pad < 1) out->push_back(char(w));
    }
    return true;
}

// The reference reads the file through a StreamReader (UTF-8), so white space is a property of decoded chars: a lone byte 0x85 or
// 0xA0 is U+FFFD (not white space), the sequences C2 85 / C2 A0 / ... are.  int.TryParse trims only the ASCII set.
inline bool is_ws(int c) { return c == ' ' || (c >= 9 && c <= 13); }
inline size_t ws_char_at(const uint8_t* p, size_t n) {      // bytes of the char.IsWhiteSpace char that starts at p, 0 if none
    if (n >= 1 && is_ws(p[0])) return 1;
    if (n >= 2 && p[0] == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
    if (n >= 3 && p[0] == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;
    if (n >= 3 && p[0] == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3;
    if (n >= 3 && p[0] == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;
    if (n >= 3 && p[0] == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;
    return 0;
}

const struct { uint16_t a, b; uint8_t c; } kRanges[] = {
#include "unicode13_classes.inc"
};

inline uint32_t next_pow2(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return uint32_t(p); }

inline uint32_t load_dword(const std::string& k, size_t off) {
    uint32_t w = 0;
    for (size_t j = 0; j < 4 && off + j < k.size(); ++j) w |= uint32_t(uint8_t(k[off + j])) << (8 * j);
    return w;
}

}  // namespace

const struct { uint32_t a, b; uint8_t c; } kSuppRanges[] = {
#include "unicode13_supp.inc"
};

// Class of every code point below 0x40000 (the BMP and planes 1-3: everything Unicode 13.0 assigns apart from the two ranges of plane 14),
// one byte each, then {uint32 n, n x {first, last | class << 24}} for the ranges above (tkz_supp_class searches those few linearly).
// The scanners of pattern 1 / cl100k only index the first 65,536 entries (UTF-16 code units); o200k matches by code point.
const std::vector<uint8_t>& bmp_class_table() {
    static const std::vector<uint8_t> table = [] {
        std::vector<uint8_t> t(TKZ_UCD_DIRECT, UC_OTHER);
        for (const auto& r : kRanges)
            for (unsigned u = r.a; u <= r.b; ++u) t[u] = r.c;
        std::vector<uint32_t> hi;
        for (const auto& r : kSuppRanges) {
            if (r.b < TKZ_UCD_DIRECT) { for (uint32_t u = r.a; u <= r.b; ++u) t[u] = r.c; }
            else { hi.push_back(r.a); hi.push_back(r.b | (uint32_t(r.c) << 24)); }     // (no range straddles the limit: planes 4-13 are unassigned)
        }
        const uint32_t n = uint32_t(hi.size() / 2);
        hi.insert(hi.begin(), n);
        t.resize(TKZ_UCD_DIRECT + hi.size() * 4);
        memcpy(t.data() + TKZ_UCD_DIRECT, hi.data(), hi.size() * 4);
        return t;
    }();
    return table;
}

int parse_tiktoken(const uint8_t* file, size_t n, Vocab* out, std::string* msg) {
    size_t pos = 0;
    if (n >= 3 && file[0] == 0xEF && file[1] == 0xBB && file[2] == 0xBF) pos = 3;  // StreamReader eats a BOM
    std::string key;
    int64_t lineno = 0;
    while (pos < n) {
        ++lineno;
        size_t ls = pos;
        while (pos < n && file[pos] != '\n' && file[pos] != '\r') ++pos;     // ReadLine (:107)
        size_t le = pos;
        if (pos < n) pos += (file[pos] == '\r' && pos + 1 < n && file[pos + 1] == '\n') ? 2 : 1;
        bool blank = true;                                                    // IsNullOrWhiteSpace (:109)
        for (size_t i = ls; i < le;) { const size_t w = ws_char_at(file + i, le - i); if (!w) { blank = false; break; } i += w; }
        if (blank) continue;
        int nsp = 0; size_t sp = 0;                                           // Split(' ') -> 2 fields (:114-118)
        for (size_t i = ls; i < le; ++i) if (file[i] == ' ') { if (!nsp) sp = i; ++nsp; }
        if (nsp != 1) { *msg = "Failed to load from BPE encoder file stream: Invalid format in the BPE encoder file stream (line " + std::to_string(lineno) + ")"; return TKZ_E_FORMAT; }
        if (!b64decode(file + ls, sp - ls, &key)) { *msg = "Failed to load from BPE encoder file stream: invalid base64 (line " + std::to_string(lineno) + ")"; return TKZ_E_FORMAT; }
        size_t a = sp + 1, b = le;                                            // int.TryParse (:122)
        while (a < b && is_ws(file[a])) ++a;
        while (b > a && is_ws(file[b - 1])) --b;
        bool neg = false;
        if (a < b && (file[a] == '+' || file[a] == '-')) { neg = file[a] == '-'; ++a; }
        bool ok = a < b;
        int64_t val = 0;
        for (size_t i = a; ok && i < b; ++i) {
            if (file[i] < '0' || file[i] > '9') ok = false;
            else { val = val * 10 + (file[i] - '0'); if (val > int64_t(INT_MAX) + 1) ok = false; }
        }
        if (neg) val = -val;
        if (!ok || val > INT_MAX || val < INT_MIN) {
            *msg = "Failed to load from BPE encoder file stream: Can't parse rank to integer (line " + std::to_string(lineno) + ")";
            return TKZ_E_FORMAT;
        }
        if (val < 0 || val > TKZ_MAX_RANK) {
            *msg = "rank " + std::to_string(val) + " outside the supported range [0, 2^27) (line " + std::to_string(lineno) + ")";
            return TKZ_E_UNSUPPORTED;
        }
        auto it = out->index.find(key);
        if (it != out->index.end()) out->ranks[it->second] = int32_t(val);     // bpeDict[key] = rank overwrites (:125)
        else {
            out->index.emplace(key, int32_t(out->keys.size()));
            out->keys.push_back(key);
            out->ranks.push_back(int32_t(val));
            out->max_key_len = std::max<int32_t>(out->max_key_len, int32_t(key.size()));
        }
    }
    // Decoder = Encoder.ToDictionary(rank -> key): a repeated rank throws ArgumentException (:82-87)
    std::vector<int32_t> sorted(out->ranks);
    std::sort(sorted.begin(), sorted.end());
    for (size_t i = 1; i < sorted.size(); ++i)
        if (sorted[i] == sorted[i - 1]) { *msg = "Encoder and decoder sizes don't match (rank " + std::to_string(sorted[i]) + " appears twice)"; return TKZ_E_DUP_RANK; }
    return TKZ_OK;
}

namespace {
// Cuckoo insertion (two hashes, one slot each): every key ends up in one of its two candidate slots, so a lookup
// is two independent gathers.  Random-walk eviction; on failure the caller retries with another seed or a larger table.
template <class Slot, class Occupied, class Slots2>
bool cuckoo_insert(std::vector<Slot>& table, Slot item, Occupied occupied, Slots2 slots2) {
    uint32_t s1, s2;
    slots2(item, &s1, &s2);
    uint32_t pos = s1;
    for (int kick = 0; kick < 4000; ++kick) {
        if (!occupied(table[pos])) { table[pos] = item; return true; }
        std::swap(item, table[pos]);
        slots2(item, &s1, &s2);
        pos = (pos == s1) ? s2 : s1;                  // the evicted key moves to its other slot
    }
    return false;
}
}  // namespace

void build_key_tables(const std::vector<KeyItem>& items, std::vector<TkzShortSlot>* short_slots, uint32_t* short_seed,
                      std::vector<TkzMidSlot>* mid_slots, uint32_t* mid_seed) {
    const size_t nk = items.size();
    size_t n_short = 0, n_mid = 0;
    for (const auto& it : items) {
        if (it.key.empty()) continue;
        if (it.key.size() <= TKZ_SHORT_KEY_MAX) ++n_short; else if (it.key.size() <= TKZ_MID_KEY_MAX) ++n_mid;
    }
    // SHORT: (2,2) cuckoo -- buckets of two 16-byte slots, two candidate buckets per key -- at a load of 0.8 (the threshold
    // of that scheme is ~0.89); sizes are arbitrary (slot = mulhi(hash, size)), grown 4 % at a time when 16 seeds fail
    {
        uint32_t nb = uint32_t(std::max<uint64_t>(8, (uint64_t(n_short) * 10 + 15) / 16));          // n / (2 * 0.8)
        for (bool done = false; !done; nb += nb / 25 + 1) {
            for (uint32_t seed = 1; seed <= 16 && !done; ++seed) {
                (*short_slots).assign(size_t(nb) * 2, TkzShortSlot{0, 0, 0, 0});
                TkzTables T{}; T.short_nb = nb; T.short_seed = seed;
                bool ok = true;
                uint32_t rng = 0x9E3779B9u * seed;
                // k_probe fetches a key's FIRST bucket and the second one only when that did not settle the lookup, so the keys that
                // occur most -- by construction of a BPE vocabulary the ones of the lowest ranks -- are put into their first bucket
                // and pinned there (a quarter of the keys, in rank order, as long as a slot is free); the walk never moves them
                std::vector<size_t> by_rank;                  // (the items arrive most frequent first)
                for (size_t i = 0; i < nk; ++i) if (!items[i].key.empty() && items[i].key.size() <= TKZ_SHORT_KEY_MAX) by_rank.push_back(i);
                std::vector<uint8_t> pinned(size_t(nb) * 2, 0), done_key(nk, 0);
                auto item_of = [&](size_t i) {
                    const std::string& k = items[i].key;
                    return TkzShortSlot{load_dword(k, 0), load_dword(k, 4), load_dword(k, 8), items[i].value | (uint32_t(k.size()) << TKZ_SHORT_RANK_BITS)};
                };
                for (size_t q = 0; q < by_rank.size() / 16; ++q) {
                    const TkzShortSlot item = item_of(by_rank[q]);
                    uint32_t s1, s2;
                    tkz_short_slots(T, item.k0, item.k1, item.k2, item.rank_len >> TKZ_SHORT_RANK_BITS, &s1, &s2);
                    for (uint32_t c = s1; c < s1 + 2; ++c)
                        if ((*short_slots)[c].rank_len == 0) { (*short_slots)[c] = item; pinned[c] = 1; done_key[by_rank[q]] = 1; break; }
                }
                for (size_t q = 0; q < by_rank.size() && ok; ++q) {
                    if (done_key[by_rank[q]]) continue;
                    TkzShortSlot item = item_of(by_rank[q]);
                    ok = false;
                    for (int kick = 0; kick < 4000; ++kick) {
                        uint32_t s1, s2;
                        tkz_short_slots(T, item.k0, item.k1, item.k2, item.rank_len >> TKZ_SHORT_RANK_BITS, &s1, &s2);
                        const uint32_t cand[4] = {s1, s1 + 1, s2, s2 + 1};
                        bool placed = false;
                        for (uint32_t c : cand) if ((*short_slots)[c].rank_len == 0) { (*short_slots)[c] = item; placed = true; break; }
                        if (placed) { ok = true; break; }
                        uint32_t movable[4]; int nm = 0;
                        for (uint32_t c : cand) if (!pinned[c]) movable[nm++] = c;
                        if (!nm) { for (uint32_t c : cand) { pinned[c] = 0; movable[nm++] = c; } }      // (both buckets pinned: a pin is given up)
                        rng = rng * 1664525u + 1013904223u;
                        std::swap(item, (*short_slots)[movable[(rng >> 16) % (uint32_t)nm]]);          // random-walk eviction
                    }
                }
                if (ok) {
                    // repair pass, most frequent keys first: a key that ended up in its second bucket moves to the first one when a slot
                    // is free there, or when one of the two keys in it is a rarer one that has a free slot in its own other bucket
                    auto slots_of = [&](const TkzShortSlot& it, uint32_t* a, uint32_t* b) { tkz_short_slots(T, it.k0, it.k1, it.k2, it.rank_len >> TKZ_SHORT_RANK_BITS, a, b); };
                    auto same = [](const TkzShortSlot& x, const TkzShortSlot& y) { return x.k0 == y.k0 && x.k1 == y.k1 && x.k2 == y.k2 && x.rank_len == y.rank_len; };
                    for (size_t q = 0; q < by_rank.size(); ++q) {
                        const TkzShortSlot item = item_of(by_rank[q]);
                        uint32_t s1, s2;
                        slots_of(item, &s1, &s2);
                        if (s1 == s2 || same((*short_slots)[s1], item) || same((*short_slots)[s1 + 1], item)) continue;
                        const uint32_t at = same((*short_slots)[s2], item) ? s2 : s2 + 1;
                        uint32_t dest = UINT32_MAX;
                        for (uint32_t c = s1; c < s1 + 2 && dest == UINT32_MAX; ++c) if ((*short_slots)[c].rank_len == 0) dest = c;
                        for (uint32_t c = s1; c < s1 + 2 && dest == UINT32_MAX; ++c) {
                            const TkzShortSlot occ = (*short_slots)[c];
                            if ((occ.rank_len & TKZ_SHORT_RANK_MASK) <= (item.rank_len & TKZ_SHORT_RANK_MASK)) continue;     // (a more frequent key stays)
                            uint32_t o1, o2;
                            slots_of(occ, &o1, &o2);
                            const uint32_t other = (c & ~1u) == o1 ? o2 : o1;
                            if (other == (c & ~1u)) continue;
                            for (uint32_t d = other; d < other + 2; ++d)
                                if ((*short_slots)[d].rank_len == 0) { (*short_slots)[d] = occ; dest = c; break; }
                        }
                        if (dest != UINT32_MAX) { (*short_slots)[dest] = item; (*short_slots)[at] = TkzShortSlot{0, 0, 0, 0}; }
                    }
                    *short_seed = seed; done = true;
                }
            }
            if (done) break;
        }
    }
    // MID: (2,1) cuckoo of 32-byte slots at a load of 0.45
    {
        uint32_t ns = uint32_t(std::max<uint64_t>(8, (uint64_t(n_mid) * 20 + 8) / 9));
        for (bool done = false; !done; ns += ns / 25 + 1) {
            for (uint32_t seed = 1; seed <= 16 && !done; ++seed) {
                (*mid_slots).assign(ns, TkzMidSlot{{0, 0, 0, 0, 0, 0, 0}, 0});
                TkzTables T{}; T.mid_ns = ns; T.mid_seed = seed;
                bool ok = true;
                for (size_t i = 0; i < nk && ok; ++i) {
                    const std::string& k = items[i].key;
                    if (k.size() <= TKZ_SHORT_KEY_MAX || k.size() > TKZ_MID_KEY_MAX) continue;
                    TkzMidSlot item;
                    for (int d = 0; d < 7; ++d) item.k[d] = load_dword(k, 4 * size_t(d));
                    item.rank_len = items[i].value | (uint32_t(k.size() - 12) << TKZ_SHORT_RANK_BITS);
                    ok = cuckoo_insert((*mid_slots), item, [](const TkzMidSlot& s) { return s.rank_len != 0; },
                                       [&](const TkzMidSlot& s, uint32_t* a, uint32_t* b) { tkz_mid_slots(T, s.k, (s.rank_len >> TKZ_SHORT_RANK_BITS) + 12u, a, b); });
                }
                if (ok) { *mid_seed = seed; done = true; }
            }
            if (done) break;
        }
    }
}

int build_tables(Vocab* v, std::string* msg) {
    (void)msg;
    const size_t nk = v->keys.size();
    // ---- single bytes ----
    v->byte_rank.assign(256, 0);
    for (int b = 0; b < 256; ++b) {
        int32_t r;
        std::string k(1, char(b));
        v->byte_rank[b] = v->lookup(k, &r) ? r : int32_t(TKZ_PSEUDO_BASE + b);
    }
    // ---- SHORT / MID / LONG whole-key tables ----
    size_t n_long = 0, blob = 0;
    for (const auto& k : v->keys) {
        if (k.empty()) continue;                       // (an empty key can never equal a regex match or a merge slice)
        if (k.size() > TKZ_MID_KEY_MAX) { ++n_long; blob += k.size(); }
    }
    const uint32_t long_cap = next_pow2(std::max<uint64_t>(16, uint64_t(n_long) * 2));
    v->long_slots.assign(long_cap, TkzLongSlot{0, 0, 0, 0});
    v->long_blob.clear();
    v->long_blob.reserve(blob + 16);
    for (size_t i = 0; i < nk; ++i) {
        const std::string& k = v->keys[i];
        if (k.size() <= TKZ_MID_KEY_MAX) continue;
        const uint32_t len = uint32_t(k.size());
        uint32_t h = tkz_hash_long_init(len);
        for (size_t off = 0; off < k.size(); off += 4) h = tkz_hash_long_step(h, load_dword(k, off));
        uint32_t s = h & (long_cap - 1);
        while (v->long_slots[s].len) s = (s + 1) & (long_cap - 1);
        v->long_slots[s] = TkzLongSlot{h, v->ranks[i], uint32_t(v->long_blob.size()), len};
        v->long_blob.insert(v->long_blob.end(), k.begin(), k.end());
    }
    v->long_blob.resize(v->long_blob.size() + 16, 0);   // kernels may read a few bytes past a key
    {   // SHORT + MID: the keys in rank order (by construction of a BPE vocabulary the lowest ranks occur most)
        std::vector<KeyItem> items;
        items.reserve(nk);
        std::vector<size_t> order(nk);
        for (size_t i = 0; i < nk; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return v->ranks[a] < v->ranks[b]; });
        for (size_t i : order) if (!v->keys[i].empty() && v->keys[i].size() <= TKZ_MID_KEY_MAX) items.push_back(KeyItem{v->keys[i], uint32_t(v->ranks[i])});
        build_key_tables(items, &v->short_slots, &v->short_seed, &v->mid_slots, &v->mid_seed);
    }
    // ---- PAIR table: every split of every key into two keys (or not-in-vocab single bytes) ----
    struct P { uint32_t a, b; int32_t r; };
    std::vector<P> pairs;
    pairs.reserve(nk * 3);
    auto id_of = [&](const std::string& s, int32_t* id) -> bool {
        if (v->lookup(s, id)) return true;
        if (s.size() == 1) { *id = int32_t(TKZ_PSEUDO_BASE + uint8_t(s[0])); return true; }
        return false;
    };
    for (size_t i = 0; i < nk; ++i) {
        const std::string& k = v->keys[i];
        for (size_t p = 1; p < k.size(); ++p) {
            int32_t a, b;
            if (id_of(k.substr(0, p), &a) && id_of(k.substr(p), &b)) pairs.push_back(P{uint32_t(a), uint32_t(b), v->ranks[i]});
        }
    }
    v->pair_entries = int64_t(pairs.size());
    v->bytepair_rank.assign(65536, TKZ_RANK_NONE);
    // compact form: every id fits 21 bits -> 8-byte entries, two per 16-byte bucket, (2,2)-cuckoo at a load of <= 0.85
    bool compact = true;
    for (const P& q : pairs) if (tkz_pair_cid(q.a) >= (1u << 21) || tkz_pair_cid(q.b) >= (1u << 21) || uint32_t(q.r) >= TKZ_PAIR_CID_LIMIT) { compact = false; break; }
    for (int32_t r : v->ranks) if (uint32_t(r) >= TKZ_PAIR_CID_LIMIT) { compact = false; break; }
    v->pair_compact = false;
    if (compact) {
        uint32_t buckets = uint32_t(std::max<uint64_t>(16, (uint64_t(pairs.size()) * 10 + 16) / 17));      // entries / 1.7: a load of 0.85
        for (int grow = 0; grow < 40 && !v->pair_compact; ++grow, buckets += buckets / 25 + 1) {
            for (uint32_t seed = 1; seed <= 16 && !v->pair_compact; ++seed) {
                std::vector<uint64_t> ent(size_t(buckets) * 2, 0);
                TkzTables T{}; T.pair_n = buckets; T.pair_seed = seed;
                bool ok = true;
                uint32_t rng = 0x9E3779B9u * seed;
                for (size_t i = 0; i < pairs.size() && ok; ++i) {
                    uint32_t a = pairs[i].a, b = pairs[i].b; int32_t r = pairs[i].r;
                    ok = false;
                    for (int kick = 0; kick < 2000; ++kick) {
                        uint32_t s1, s2;
                        tkz_pair_slots(T, a, b, &s1, &s2);
                        const uint64_t e = tkz_pair_key42(a, b) | (uint64_t(uint32_t(r)) << 42) | (1ull << 63);
                        uint64_t* cand[4] = {&ent[2 * size_t(s1)], &ent[2 * size_t(s1) + 1], &ent[2 * size_t(s2)], &ent[2 * size_t(s2) + 1]};
                        bool placed = false;
                        for (uint64_t* c : cand) if (*c == 0) { *c = e; placed = true; break; }
                        if (placed) { ok = true; break; }
                        rng = rng * 1664525u + 1013904223u;
                        uint64_t* victim = cand[(rng >> 16) & 3];
                        const uint64_t old = *victim;
                        *victim = e;
                        // the evicted entry goes on: recover its ids (compact ids map back to ids)
                        auto uncid = [](uint32_t c) { return c >= TKZ_PAIR_CID_LIMIT ? c - TKZ_PAIR_CID_LIMIT + uint32_t(TKZ_PSEUDO_BASE) : c; };
                        a = uncid(uint32_t(old & 0x1FFFFFu)); b = uncid(uint32_t((old >> 21) & 0x1FFFFFu)); r = int32_t((old >> 42) & 0x1FFFFFu);
                    }
                }
                if (ok) {
                    v->pair_slots.assign(buckets, TkzPairSlot{0, 0, 0, 0});
                    for (uint32_t k = 0; k < buckets; ++k) {
                        const uint64_t e0 = ent[2 * size_t(k)], e1 = ent[2 * size_t(k) + 1];
                        v->pair_slots[k] = TkzPairSlot{uint32_t(e0), uint32_t(e0 >> 32), int32_t(uint32_t(e1)), uint32_t(e1 >> 32)};
                    }
                    v->pair_seed = seed; v->pair_compact = true;
                }
            }
        }
    }
    uint32_t pair_cap = uint32_t(std::max<uint64_t>(16, (uint64_t(pairs.size()) * 20 + 8) / 9));          // a load of 0.45
    for (bool done = v->pair_compact; !done; pair_cap += pair_cap / 25 + 1) {
        for (uint32_t seed = 1; seed <= 16 && !done; ++seed) {
            v->pair_slots.assign(pair_cap, TkzPairSlot{0, 0, 0, 0});
            TkzTables T{}; T.pair_n = pair_cap; T.pair_seed = seed;
            bool ok = true;
            for (size_t i = 0; i < pairs.size() && ok; ++i)
                ok = cuckoo_insert(v->pair_slots, TkzPairSlot{pairs[i].a, pairs[i].b, pairs[i].r, 1},
                                   [](const TkzPairSlot& s) { return s.valid != 0; },
                                   [&](const TkzPairSlot& s, uint32_t* a, uint32_t* b) { tkz_pair_slots(T, s.a, s.b, a, b); });
            if (ok) { v->pair_seed = seed; done = true; }
        }
        if (done) break;
    }
    if (v->pair_slots.size() >= (size_t(1) << 27)) { if (msg) *msg = "pair table too large (the kernels address it with 32-bit byte offsets)"; return TKZ_E_UNSUPPORTED; }
    // two-single-byte pairs, directly indexed by the BYTES (not ranks): first-level lookups
    for (int a = 0; a < 256; ++a)
        for (int b = 0; b < 256; ++b) {
            std::string k; k.push_back(char(a)); k.push_back(char(b));
            int32_t r;
            if (v->lookup(k, &r)) v->bytepair_rank[(a << 8) | b] = r;
        }
    return TKZ_OK;
}

}  // namespace tkz
