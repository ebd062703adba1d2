// tkz_tokenizer.hpp -- C++ host mirror of the reference's tokenizer interface for the Encode path, over the
// C ABI of include/tkz.h.  Header-only; link with libtkz.so.
//
//   tkz::TikTokenizer          ITokenizer.Encode x2 + EncodeBatch     Tokenizer_C#/TokenizerLib/ITokenizer.cs:12,28
//                              EncodeBatchFlat: (ids, offsets) in page-locked buffers that are kept from call to call (tkz::FlatBatch)
//                              EncodeTrimSuffix / EncodeTrimPrefix x2   ITokenizer.cs:30-44, TikTokenizer.cs:288-579
//   tkz::TokenizerBuilder      CreateTokenizer(stream, specials, pattern)   TokenizerBuilder.cs:210-213
//
// Text is UTF-8 (std::string); EncodeUtf16 takes the code units of a .NET string.  Special-token
// segmentation (EncodeInternal / FindNextSpecialToken, TikTokenizer.cs:141-170,230-241) runs on the host and
// every plain segment of a batch goes to the GPU in ONE tkz_encode_batch_utf8 call.  Errors are exceptions
// named after the reference's: FormatException -> tkz::FormatError, ArgumentException -> tkz::DuplicateRankError,
// KeyNotFoundException -> tkz::KeyNotFoundError, NotImplementedException -> tkz::NotImplementedError.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "tkz.h"

namespace tkz {

struct Error : std::runtime_error { int status; Error(int s, const std::string& m) : std::runtime_error(m), status(s) {} };
struct FormatError : Error { using Error::Error; };
struct DuplicateRankError : Error { using Error::Error; };
struct KeyNotFoundError : Error { using Error::Error; };
struct NotImplementedError : Error { using Error::Error; };

inline void check(tkz_status s) {
    if (s == TKZ_OK) return;
    const std::string m = tkz_last_error();
    switch (s) {
        case TKZ_E_FORMAT: throw FormatError(s, m);
        case TKZ_E_DUP_RANK: throw DuplicateRankError(s, m);
        case TKZ_E_KEY_NOT_FOUND: throw KeyNotFoundError(s, m);
        case TKZ_E_UNSUPPORTED: throw NotImplementedError(s, m);
        default: throw Error(s, m);
    }
}

using SpecialTokens = std::vector<std::pair<std::string, int32_t>>;   // registration order matters (alternation order)

// A grow-only page-locked host buffer (tkz_host_alloc): copies from and to it run at the PCIe rate.
class PinnedBuffer {
public:
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { tkz_host_free(p_); }
    void* ensure(size_t bytes) {
        if (bytes > cap_) {
            tkz_host_free(p_); p_ = nullptr; cap_ = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            check(tkz_host_alloc(want, &p_));
            cap_ = want;
        }
        return p_;
    }
    template <class T> T* as() const { return static_cast<T*>(p_); }
private:
    void* p_ = nullptr; size_t cap_ = 0;
};

// The result of TikTokenizer::EncodeBatchFlat: text t is ids()[offsets()[t] .. offsets()[t + 1]).  It owns the page-locked buffers the
// call works in (the gathered text, the ids, the offsets) and keeps them from call to call: hand the same FlatBatch to every call of a
// loop and nothing is allocated after the first.  The views are valid until the next call that is given this object.
class FlatBatch {
public:
    const int32_t* ids() const { return ids_; }
    const int64_t* offsets() const { return offsets_; }
    int64_t n_texts() const { return n_texts_; }
    int64_t n_ids() const { return n_texts_ >= 0 && offsets_ ? offsets_[n_texts_] : 0; }
    std::vector<int32_t> text(int64_t t) const { return std::vector<int32_t>(ids_ + offsets_[t], ids_ + offsets_[t + 1]); }
    // where the last call's time went, in milliseconds: the offsets pass, waiting for gathered sub-batches, inside tkz_encode_batch_utf8
    double last_offsets_ms = 0, last_wait_ms = 0, last_encode_ms = 0;
private:
    friend class TikTokenizer;
    PinnedBuffer in_bytes_, in_offs_, sub_offs_, out_ids_, out_offs_;
    std::vector<int32_t> spliced_ids_; std::vector<int64_t> spliced_offs_;     // (only when special tokens were spliced in)
    const int32_t* ids_ = nullptr; const int64_t* offsets_ = nullptr; int64_t n_texts_ = 0;
    double tokens_per_byte_ = 0;                                               // the densest batch seen: sizes the id buffer of the next one
};

class TikTokenizer {
public:
    TikTokenizer(const std::string& tikTokenBpeFile, SpecialTokens specialTokensEncoder, const std::string& pattern, int device = 0)
        : specials_(std::move(specialTokensEncoder)) {
        int32_t pat = 0;
        check(tkz_pattern_from_regex(pattern.c_str(), &pat));
        tkz_vocab* v = nullptr;
        check(tkz_vocab_from_tiktoken(reinterpret_cast<const uint8_t*>(tikTokenBpeFile.data()), tikTokenBpeFile.size(), &v));
        const tkz_status s = tkz_encoder_create(v, pat, device, &enc_);
        tkz_vocab_destroy(v);
        check(s);
    }
    ~TikTokenizer() { tkz_encoder_destroy(enc_); }
    TikTokenizer(const TikTokenizer&) = delete;
    TikTokenizer& operator=(const TikTokenizer&) = delete;

    // Encode(string text, IReadOnlyCollection<string> allowedSpecial)        TikTokenizer.cs:178-185
    std::vector<int32_t> Encode(const std::string& text, const std::vector<std::string>& allowedSpecial) const {
        return EncodeBatch({text}, allowedSpecial)[0];
    }
    // Encode(string text, bool applySpecialTokens = true)                     TikTokenizer.cs:193-207
    std::vector<int32_t> Encode(const std::string& text, bool applySpecialTokens = true) const {
        return EncodeBatch({text}, applySpecialTokens)[0];
    }
    std::vector<std::vector<int32_t>> EncodeBatch(const std::vector<std::string>& texts, bool applySpecialTokens = true) const {
        std::vector<std::string> all;
        if (applySpecialTokens) for (const auto& s : specials_) all.push_back(s.first);
        return EncodeBatch(texts, all);
    }
    std::vector<std::vector<int32_t>> EncodeBatch(const std::vector<std::string>& texts, const std::vector<std::string>& allowedSpecial) const {
        struct Item { size_t text; int32_t special; int64_t segment; };
        std::vector<Item> plan;
        std::vector<uint8_t> bytes;
        std::vector<int64_t> offs{0};
        for (size_t t = 0; t < texts.size(); ++t) {
            for (const Segment& g : segments(texts[t], allowedSpecial)) {
                if (g.special) { plan.push_back({t, g.id, -1}); continue; }
                plan.push_back({t, 0, static_cast<int64_t>(offs.size()) - 1});
                bytes.insert(bytes.end(), texts[t].begin() + g.begin, texts[t].begin() + g.end);
                offs.push_back(static_cast<int64_t>(bytes.size()));
            }
        }
        const int64_t nseg = static_cast<int64_t>(offs.size()) - 1;
        std::vector<int32_t> ids(bytes.size() ? bytes.size() : 1);
        std::vector<int64_t> ooff(static_cast<size_t>(nseg) + 1, 0);
        int64_t needed = 0;
        if (bytes.empty()) bytes.push_back(0);
        check(tkz_encode_batch_utf8(enc_, bytes.data(), offs.data(), nseg, ids.data(), static_cast<int64_t>(ids.size()), ooff.data(), &needed));
        std::vector<std::vector<int32_t>> out(texts.size());
        for (const Item& it : plan) {
            if (it.segment < 0) { out[it.text].push_back(it.special); continue; }
            out[it.text].insert(out[it.text].end(), ids.begin() + ooff[it.segment], ids.begin() + ooff[it.segment + 1]);
        }
        return out;
    }

    // EncodeBatch without a vector per text (FlatBatch above).  The texts are gathered into page-locked memory by `threads` host threads
    // (0: as many as the batch is worth, at most 8), encoded by ONE tkz_encode_batch_utf8 call that reads and writes page-locked buffers,
    // and -- when no special token applies, the reference's plain path (TikTokenizer.cs:180-183,196-199) -- the views of `out` are the
    // call's own output, untouched.
    void EncodeBatchFlat(const std::vector<std::string>& texts, FlatBatch& out, bool applySpecialTokens = true, int threads = 0) const {
        EncodeBatchFlat(texts, applySpecialTokens ? all_specials() : std::vector<std::string>{}, out, threads);
    }
    void EncodeBatchFlat(const std::vector<std::string>& texts, const std::vector<std::string>& allowedSpecial, FlatBatch& out, int threads = 0) const {
        const bool plain = allowedSpecial.empty() || specials_.empty();
        out.n_texts_ = static_cast<int64_t>(texts.size());
        if (plain) {                                           // one segment per text, the texts themselves
            encode_segments(static_cast<int64_t>(texts.size()), [&](int64_t i) { return std::pair<const char*, size_t>(texts[static_cast<size_t>(i)].data(), texts[static_cast<size_t>(i)].size()); }, out, threads);
            out.ids_ = out.out_ids_.as<int32_t>(); out.offsets_ = out.out_offs_.as<int64_t>();
            return;
        }
        // what goes to the device: (source pointer, length) of every plain segment, and where the special ids go
        struct Item { size_t text; int32_t special; int64_t segment; };
        std::vector<Item> plan;
        std::vector<std::pair<const char*, size_t>> segs;
        for (size_t t = 0; t < texts.size(); ++t)
            for (const Segment& g : segments(texts[t], allowedSpecial)) {
                if (g.special) { plan.push_back({t, g.id, -1}); continue; }
                plan.push_back({t, 0, static_cast<int64_t>(segs.size())});
                segs.emplace_back(texts[t].data() + g.begin, g.end - g.begin);
            }
        const int64_t nseg = static_cast<int64_t>(segs.size());
        encode_segments(nseg, [&](int64_t i) { return segs[static_cast<size_t>(i)]; }, out, threads);
        const int32_t* ids = out.out_ids_.as<int32_t>();
        const int64_t* ooff = out.out_offs_.as<int64_t>();
        // splice the special ids in between the segments' id ranges
        size_t nspecial = 0;
        for (const Item& it : plan) if (it.segment < 0) ++nspecial;
        out.spliced_ids_.resize(static_cast<size_t>(ooff[nseg]) + nspecial);
        out.spliced_offs_.assign(texts.size() + 1, 0);
        int64_t w = 0; size_t cur = 0;
        for (const Item& it : plan) {
            while (cur < it.text) out.spliced_offs_[++cur] = w;
            if (it.segment < 0) { out.spliced_ids_[static_cast<size_t>(w++)] = it.special; continue; }
            const int64_t n = ooff[it.segment + 1] - ooff[it.segment];
            std::memcpy(out.spliced_ids_.data() + w, ids + ooff[it.segment], static_cast<size_t>(n) * 4);
            w += n;
        }
        while (cur < texts.size()) out.spliced_offs_[++cur] = w;
        out.ids_ = out.spliced_ids_.data(); out.offsets_ = out.spliced_offs_.data();
    }

    // The same for hosts whose strings are UTF-16 (a .NET `string`, Java, JavaScript): the plain path (no special tokens applied) on code units --
    // what bindings/csharp/GpuTikTokenizer.EncodeBatchFlat does with `string.CopyTo` + tkz_encode_batch_utf16.  The units are gathered into
    // page-locked memory by `threads` host threads, uploaded as they are (the library cuts the batch into chunks and runs Encoding.UTF8.GetBytes
    // -- TikTokenizer.cs:261 -- on the device while the next chunk is on its way) and the ids come back into page-locked memory.
    void EncodeBatchFlatUtf16(const std::vector<std::u16string>& texts, FlatBatch& out, int threads = 0) const {
        const int64_t n = static_cast<int64_t>(texts.size());
        out.n_texts_ = n;
        int nth = threads > 0 ? threads : static_cast<int>(std::min<int64_t>(16, n >> 14));
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && nth > static_cast<int>(hw)) nth = static_cast<int>(hw);
        if (nth < 1) nth = 1;
        int64_t* offs = static_cast<int64_t*>(out.in_offs_.ensure((static_cast<size_t>(n) + 1) * 8));
        offs[0] = 0;
        for (int64_t i = 0; i < n; ++i) offs[i + 1] = offs[i] + static_cast<int64_t>(texts[static_cast<size_t>(i)].size());
        const int64_t total = offs[n];
        uint16_t* units = static_cast<uint16_t*>(out.in_bytes_.ensure(static_cast<size_t>(total) * 2 + 64));
        {
            Joiner j;
            for (int t = 0; t < nth; ++t)
                j.pool.emplace_back([&, t] {
                    const int64_t lo = t == 0 ? 0 : std::lower_bound(offs, offs + n, total / nth * t) - offs;
                    const int64_t hi = t == nth - 1 ? n : std::lower_bound(offs, offs + n, total / nth * (t + 1)) - offs;
                    for (int64_t i = lo; i < hi; ++i) std::memcpy(units + offs[i], texts[static_cast<size_t>(i)].data(), texts[static_cast<size_t>(i)].size() * 2);
                });
        }
        int64_t* ooff = static_cast<int64_t*>(out.out_offs_.ensure((static_cast<size_t>(n) + 1) * 8));
        // a code unit is at most three UTF-8 bytes and a token at least one byte: 3 * total ids always suffice; text has a token per ~4 units, so the
        // first call gets room for one per two and the call is repeated with the exact count when that was not enough
        int64_t cap = std::max<int64_t>(1, std::min<int64_t>(3 * total, std::max<int64_t>(total / 2 + 4096, static_cast<int64_t>(static_cast<double>(total) * out.tokens_per_byte_ * 1.1))));
        for (;;) {
            int32_t* ids = static_cast<int32_t*>(out.out_ids_.ensure(static_cast<size_t>(cap) * 4));
            int64_t needed = 0;
            const tkz_status st = tkz_encode_batch_utf16(enc_, units, offs, n, ids, cap, ooff, &needed);
            if (st == TKZ_E_CAPACITY && needed > cap) { cap = needed; continue; }
            check(st);
            if (total > 0) out.tokens_per_byte_ = std::max(out.tokens_per_byte_, static_cast<double>(needed) / static_cast<double>(total));
            break;
        }
        out.ids_ = out.out_ids_.as<int32_t>(); out.offsets_ = out.out_offs_.as<int64_t>();
    }

    using Trimmed = std::pair<std::vector<int32_t>, std::string>;   // (List<int> TokenIds, string Text)
    // EncodeTrimSuffix(string, IReadOnlyCollection<string> allowedSpecial, int maxTokenCount)     TikTokenizer.cs:394-403
    Trimmed EncodeTrimSuffix(const std::string& text, const std::vector<std::string>& allowedSpecial, int maxTokenCount) const {
        std::vector<int32_t> ids;
        int64_t tokenCount = 0; size_t encodeLength = 0;
        for (const PieceItem& it : piece_items(text, allowedSpecial)) {            // the walk of :288-341 / :343-392
            tokenCount += static_cast<int64_t>(it.ids.size());
            if (tokenCount > maxTokenCount) break;                                // the piece that overflows is dropped, with all after it
            ids.insert(ids.end(), it.ids.begin(), it.ids.end());
            encodeLength = it.end;
            if (tokenCount >= maxTokenCount) break;
        }
        return {ids, text.substr(0, encodeLength)};
    }
    // EncodeTrimSuffix(string, int maxTokenCount, bool applySpecialTokens = true)                  TikTokenizer.cs:412-429
    Trimmed EncodeTrimSuffix(const std::string& text, int maxTokenCount, bool applySpecialTokens = true) const {
        return EncodeTrimSuffix(text, applySpecialTokens ? all_specials() : std::vector<std::string>{}, maxTokenCount);
    }
    // EncodeTrimPrefix(string, IReadOnlyCollection<string> allowedSpecial, int maxTokenCount)     TikTokenizer.cs:529-536
    Trimmed EncodeTrimPrefix(const std::string& text, const std::vector<std::string>& allowedSpecial, int maxTokenCount) const {
        std::vector<int32_t> ids;
        std::vector<std::pair<int64_t, size_t>> boundaries{{0, 0}};               // tokenCountMap (:438-441)
        int64_t tokenCount = 0;
        for (const PieceItem& it : piece_items(text, allowedSpecial)) {
            tokenCount += static_cast<int64_t>(it.ids.size());
            ids.insert(ids.end(), it.ids.begin(), it.ids.end());
            boundaries.push_back({tokenCount, it.end});
        }
        if (tokenCount <= maxTokenCount) return {ids, text};                      // TrimPrefix (:470-483)
        const int64_t prefix = tokenCount - maxTokenCount;
        int64_t cutTokens = 0; size_t cutLen = 0;
        for (const auto& b : boundaries) if (b.first >= prefix) { cutTokens = b.first; cutLen = b.second; break; }
        return {std::vector<int32_t>(ids.begin() + cutTokens, ids.end()), text.substr(cutLen)};
    }
    // EncodeTrimPrefix(string, int maxTokenCount, bool applySpecialTokens = true)                  TikTokenizer.cs:545-564
    Trimmed EncodeTrimPrefix(const std::string& text, int maxTokenCount, bool applySpecialTokens = true) const {
        return EncodeTrimPrefix(text, applySpecialTokens ? all_specials() : std::vector<std::string>{}, maxTokenCount);
    }
    // a batch of UTF-16 strings (plain path): the code units go to the device as they are (tkz_encode_batch_utf16)
    std::vector<std::vector<int32_t>> EncodeBatchUtf16(const std::vector<std::u16string>& texts) const {
        std::vector<uint16_t> units;
        std::vector<int64_t> offs{0};
        for (const auto& t : texts) { units.insert(units.end(), t.begin(), t.end()); offs.push_back(static_cast<int64_t>(units.size())); }
        const size_t cap = units.size() * 3 + 1;
        std::vector<int32_t> ids(cap);
        std::vector<int64_t> ooff(texts.size() + 1, 0);
        int64_t needed = 0;
        if (units.empty()) units.push_back(0);
        check(tkz_encode_batch_utf16(enc_, units.data(), offs.data(), static_cast<int64_t>(texts.size()), ids.data(), static_cast<int64_t>(cap), ooff.data(), &needed));
        std::vector<std::vector<int32_t>> out(texts.size());
        for (size_t t = 0; t < texts.size(); ++t) out[t].assign(ids.begin() + ooff[t], ids.begin() + ooff[t + 1]);
        return out;
    }
    // the code units of a .NET string; plain path only (Encode(text, false))
    std::vector<int32_t> EncodeUtf16(const std::u16string& text) const {
        std::vector<int32_t> ids(text.size() * 3 + 1);
        int64_t n = 0;
        check(tkz_encode_utf16(enc_, reinterpret_cast<const uint16_t*>(text.data()), static_cast<int64_t>(text.size()), ids.data(),
                               static_cast<int64_t>(ids.size()), &n));
        ids.resize(static_cast<size_t>(n));
        return ids;
    }
    tkz_encoder* native() const { return enc_; }
    // the device workspace of batches of up to max_bytes / max_docs, allocated now instead of inside the first batch call (tkz_encoder_reserve): what
    // TokenizerBuilder.CreateTokenizer (TokenizerBuilder.cs:210-213) is for a drop-in -- construction pays, not the first Encode
    void Reserve(int64_t max_bytes, int64_t max_docs) { check(tkz_encoder_reserve(enc_, max_bytes, max_docs)); }
    // The split is whatever the HOST's regex engine makes of the pattern (TikTokenizer.cs:77 compiles it in the running process).  A host on another
    // runtime than net6.0 hands its Unicode classification over (classes[cp] in 0..8 for cp < n: 65536 code units or 1114112 code points; nullptr:
    // the built-in Unicode 13.0 table) and says how it reads cl100k's (?i:...): .NET >= 7 folds U+017F onto `s`.
    void SetUnicodeClasses(const uint8_t* classes, int64_t n_code_points) { check(tkz_encoder_set_unicode_classes(enc_, classes, n_code_points)); }
    void SetCaseEquivalence(bool dotnet7_or_later) { check(tkz_encoder_set_option(enc_, TKZ_OPT_CASE_EQUIVALENCE, dotnet7_or_later ? 1 : 0)); }

private:
    struct Segment { bool special; int32_t id; size_t begin, end; };          // bytes [begin, end) of the text
    // EncodeInternal + FindNextSpecialToken (TikTokenizer.cs:141-170,230-241): plain segments and special literals in order
    std::vector<Segment> segments(const std::string& text, const std::vector<std::string>& allowedSpecial) const {
        std::vector<Segment> out;
        size_t start = 0;
        for (;;) {
            size_t hit_pos = std::string::npos; int hit = -1;
            if (!allowedSpecial.empty()) {
                size_t find = start;
                for (;;) {
                    hit = -1;
                    size_t p = find;
                    for (; p < text.size(); ++p) { hit = match_at(text, p); if (hit >= 0) break; }
                    if (hit < 0) break;
                    bool ok = false;
                    for (const auto& a : allowedSpecial) if (a == specials_[hit].first) { ok = true; break; }
                    if (ok) { hit_pos = p; break; }
                    find = p + 1;                                       // startFind = nextSpecial.Index + 1 (one UTF-16 unit; literals are ASCII)
                    while (find < text.size() && (static_cast<uint8_t>(text[find]) & 0xC0) == 0x80) ++find;
                }
            }
            const size_t end = hit >= 0 ? hit_pos : text.size();
            if (end > start) out.push_back({false, 0, start, end});
            if (hit < 0) break;
            out.push_back({true, specials_[hit].second, hit_pos, hit_pos + specials_[hit].first.size()});   // EncodeSpecialToken (:215-220)
            start = hit_pos + specials_[hit].first.size();
            if (start >= text.size()) break;
        }
        return out;
    }
    std::vector<std::string> all_specials() const {
        std::vector<std::string> all;
        for (const auto& s : specials_) all.push_back(s.first);
        return all;
    }
    // one item per regex piece of every plain segment and one per special token: its ids and the byte position where it ends
    struct PieceItem { std::vector<int32_t> ids; size_t end; };
    std::vector<PieceItem> piece_items(const std::string& text, const std::vector<std::string>& allowedSpecial) const {
        const std::vector<Segment> segs = segments(text, allowedSpecial);
        std::vector<uint8_t> bytes;
        std::vector<int64_t> offs{0};
        for (const Segment& g : segs)
            if (!g.special) { bytes.insert(bytes.end(), text.begin() + g.begin, text.begin() + g.end); offs.push_back(static_cast<int64_t>(bytes.size())); }
        const int64_t nseg = static_cast<int64_t>(offs.size()) - 1;
        const size_t cap = bytes.size() ? bytes.size() : 1;
        std::vector<int32_t> ids(cap);
        std::vector<int64_t> dpo(static_cast<size_t>(nseg) + 1, 0), pbo(cap + 1, 0), pto(cap + 1, 0);
        int64_t npieces = 0, needed = 0;
        if (bytes.empty()) bytes.push_back(0);
        check(tkz_encode_batch_pieces_utf8(enc_, bytes.data(), offs.data(), nseg, ids.data(), static_cast<int64_t>(cap), dpo.data(), pbo.data(),
                                           pto.data(), static_cast<int64_t>(cap), &npieces, &needed));
        std::vector<PieceItem> out;
        int64_t k = 0;
        for (const Segment& g : segs) {
            if (g.special) { out.push_back({{g.id}, g.end}); continue; }
            for (int64_t p = dpo[k]; p < dpo[k + 1]; ++p)
                out.push_back({std::vector<int32_t>(ids.begin() + pto[p], ids.begin() + pto[p + 1]),
                               g.begin + static_cast<size_t>(pbo[p + 1] - offs[k])});
            ++k;
        }
        return out;
    }
    // The work of EncodeBatchFlat on nseg segments, seg_at(i) = (pointer, length) of segment i: the ids land in out.out_ids_, the offsets
    // (nseg + 1) in out.out_offs_.  Everything that touches every segment runs on `threads` host threads (0: one per ~16 k segments, at most
    // 16): the segment offsets (a two-pass prefix sum) and the gather into page-locked memory.  The gather is cut into sub-batches of ~128 MB
    // and runs AHEAD of the device: the workers go through the sub-batches in order without waiting for anything, the calling thread hands
    // sub-batch k to tkz_encode_batch_utf8 (which overlaps upload, kernels and download among its own chunks) as soon as it is gathered.
    // A token is at least one byte, English-like text has one per ~4: room for a token per two bytes first; when that was not enough, room
    // for a token per byte (always enough) and the calls again.
    struct Joiner { std::vector<std::thread> pool; ~Joiner() { for (std::thread& t : pool) if (t.joinable()) t.join(); } };
    template <class SegAt>
    void encode_segments(int64_t nseg, SegAt seg_at, FlatBatch& out, int threads) const {
        int nth = threads > 0 ? threads : static_cast<int>(std::min<int64_t>(16, nseg >> 14));
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && nth > static_cast<int>(hw)) nth = static_cast<int>(hw);
        if (nth < 1) nth = 1;
        const auto t_begin = std::chrono::steady_clock::now();
        auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
        int64_t* offs = static_cast<int64_t*>(out.in_offs_.ensure((static_cast<size_t>(nseg) + 1) * 8));
        auto slice = [&](int t) { return nseg * t / nth; };      // thread t owns segments [slice(t), slice(t + 1)) of every pass over the segments
        // 1. offsets: every thread sums its slice, then writes its offsets from the sum of the slices before it
        offs[0] = 0;
        if (nth == 1) { for (int64_t i = 0; i < nseg; ++i) offs[i + 1] = offs[i] + static_cast<int64_t>(seg_at(i).second); }
        else {
            std::vector<int64_t> part(static_cast<size_t>(nth) + 1, 0);
            { Joiner j; for (int t = 0; t < nth; ++t) j.pool.emplace_back([&, t] { int64_t sum = 0; for (int64_t i = slice(t); i < slice(t + 1); ++i) sum += static_cast<int64_t>(seg_at(i).second); part[static_cast<size_t>(t) + 1] = sum; }); }
            for (int t = 0; t < nth; ++t) part[static_cast<size_t>(t) + 1] += part[static_cast<size_t>(t)];
            { Joiner j; for (int t = 0; t < nth; ++t) j.pool.emplace_back([&, t] { int64_t at = part[static_cast<size_t>(t)]; for (int64_t i = slice(t); i < slice(t + 1); ++i) { at += static_cast<int64_t>(seg_at(i).second); offs[i + 1] = at; } }); }
        }
        const int64_t total = offs[nseg];
        out.last_offsets_ms = ms_since(t_begin); out.last_wait_ms = out.last_encode_ms = 0;
        uint8_t* bytes = static_cast<uint8_t*>(out.in_bytes_.ensure(static_cast<size_t>(total) + 64));
        // 2. sub-batches
        static const int64_t kSubBytes = [] { const char* v = std::getenv("TKZ_FLAT_SUBBATCH_BYTES"); const long long n = v ? std::atoll(v) : 0; return n > 0 ? static_cast<int64_t>(n) : (int64_t(128) << 20); }();
        const int nsb = total >= 2 * kSubBytes ? static_cast<int>(std::min<int64_t>(64, total / kSubBytes)) : 1;
        std::vector<int64_t> cut(static_cast<size_t>(nsb) + 1, 0);
        cut[static_cast<size_t>(nsb)] = nseg;
        for (int k = 1; k < nsb; ++k) cut[static_cast<size_t>(k)] = std::lower_bound(offs, offs + nseg, total / nsb * k) - offs;
        // the offsets of sub-batch k, starting at 0 as the entry point wants them: sub[cut[k] + k .. cut[k + 1] + k]
        int64_t* sub = nsb > 1 ? static_cast<int64_t*>(out.sub_offs_.ensure((static_cast<size_t>(nseg) + static_cast<size_t>(nsb) + 1) * 8)) : offs;
        std::vector<std::atomic<int>> done(static_cast<size_t>(nsb));
        for (auto& d : done) d.store(0, std::memory_order_relaxed);
        auto gather_slice = [&](int k, int t) {                  // thread t's share of sub-batch k: about the same number of bytes for every thread
            const int64_t lo0 = cut[static_cast<size_t>(k)], hi0 = cut[static_cast<size_t>(k) + 1], b0 = offs[lo0], nb = offs[hi0] - b0;
            const int64_t lo = t == 0 ? lo0 : std::lower_bound(offs + lo0, offs + hi0, b0 + nb / nth * t) - offs;
            const int64_t hi = t == nth - 1 ? hi0 : std::lower_bound(offs + lo0, offs + hi0, b0 + nb / nth * (t + 1)) - offs;
            for (int64_t i = lo; i < hi; ++i) {
                const std::pair<const char*, size_t> g = seg_at(i);
                std::memcpy(bytes + offs[i], g.first, g.second);
                if (nsb > 1) sub[i + k] = offs[i] - b0;
            }
            if (nsb > 1 && t == nth - 1) sub[hi0 + k] = nb;
        };
        Joiner workers;
        if (nth == 1 && nsb == 1) { gather_slice(0, 0); done[0].store(1, std::memory_order_release); }
        else for (int t = 0; t < nth; ++t)
            workers.pool.emplace_back([&, t] { for (int k = 0; k < nsb; ++k) { gather_slice(k, t); done[static_cast<size_t>(k)].fetch_add(1, std::memory_order_release); } });
        const int need = (nth == 1 && nsb == 1) ? 1 : nth;
        // 3. encode
        int64_t* ooff = static_cast<int64_t*>(out.out_offs_.ensure((static_cast<size_t>(nseg) + 1) * 8));
        bool full = false;
        for (int attempt = 0;; ++attempt) {
            const int64_t cap = full ? std::max<int64_t>(1, total)
                                     : std::max<int64_t>(1, std::min<int64_t>(total, std::max<int64_t>(total / 2 + 4096, static_cast<int64_t>(static_cast<double>(total) * out.tokens_per_byte_ * 1.1))));
            int32_t* ids = static_cast<int32_t*>(out.out_ids_.ensure(static_cast<size_t>(cap) * 4));
            int64_t tok_base = 0;
            bool over = false;
            for (int k = 0; k < nsb && !over; ++k) {
                const auto t_w = std::chrono::steady_clock::now();
                while (done[static_cast<size_t>(k)].load(std::memory_order_acquire) < need) std::this_thread::yield();
                out.last_wait_ms += ms_since(t_w);
                const auto t_e = std::chrono::steady_clock::now();
                const int64_t d0 = cut[static_cast<size_t>(k)], nd = cut[static_cast<size_t>(k) + 1] - d0;
                int64_t needed = 0;
                const tkz_status st = tkz_encode_batch_utf8(enc_, bytes + offs[d0], nsb > 1 ? sub + d0 + k : offs, nd, ids + tok_base, cap - tok_base, ooff + d0, &needed);
                out.last_encode_ms += ms_since(t_e);
                if (st == TKZ_E_CAPACITY) { over = true; break; }
                check(st);
                if (tok_base) for (int64_t i = d0; i <= d0 + nd; ++i) ooff[i] += tok_base;
                tok_base += needed;
            }
            if (!over) break;
            if (attempt || cap >= total) check(TKZ_E_CAPACITY);                      // (cannot happen: a token per byte is always enough)
            full = true;
        }
        if (total > 0) out.tokens_per_byte_ = std::max(out.tokens_per_byte_, static_cast<double>(ooff[nseg]) / static_cast<double>(total));
    }
    int match_at(const std::string& text, size_t p) const {       // first registered literal that matches at p
        for (size_t i = 0; i < specials_.size(); ++i) {
            const std::string& lit = specials_[i].first;
            if (!lit.empty() && text.compare(p, lit.size(), lit) == 0) return static_cast<int>(i);
        }
        return -1;
    }
    SpecialTokens specials_;
    tkz_encoder* enc_ = nullptr;
};

struct TokenizerBuilder {
    // TokenizerBuilder.CreateTokenizer(Stream, IReadOnlyDictionary<string,int>, string pattern, int cacheSize)
    static TikTokenizer* CreateTokenizer(const std::string& tikTokenBpeFile, SpecialTokens specialTokensEncoder, const std::string& pattern,
                                         int cacheSize = 8192, int device = 0) {
        TikTokenizer* t = new TikTokenizer(tikTokenBpeFile, std::move(specialTokensEncoder), pattern, device);
        // the reference's LRU piece memo (LRUCache.cs; no effect on results) lives on the device with a fixed size: cacheSize says whether it is used
        if (cacheSize <= 0) (void)tkz_encoder_set_option(t->native(), TKZ_OPT_PIECE_MEMO, 0);
        return t;
    }
};

}  // namespace tkz
