// tools/train_bpe.cpp -- plain greedy byte-level BPE trainer (development tool, not product code).
//
// Produces stand-in rank files of the SIZE of the vocabularies the reference downloads at run time and that do not exist
// offline (cl100k_base: 100,256 keys, TokenizerBuilder.cs:113-120; o200k_base: ~200 k keys, tokenizer_ts/src/tokenizerBuilder.ts:133-137),
// so that the device tables, the parity tests and the bench run at the real table sizes.  The files are valid .tiktoken images
// ("base64 SP rank" per line, TikTokenizer.cs:106-129) with the structure of a trained BPE vocabulary: all 256 single bytes
// first (in tiktoken's byte order, so that "!" has rank 0), then one key per merge, rank = merge order, every key the
// concatenation of two earlier keys, no key longer than 128 bytes (the longest key of the published vocabularies).
//
// Input  (stdin, binary):  u64 n_words, then per word: u32 len, i64 count, len bytes   (the pieces of a pre-tokenised corpus)
// Output (stdout): the .tiktoken text.        usage: train_bpe <n_keys> [max_key_len=128]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

struct Word { std::vector<int32_t> sym; int64_t cnt; };

static std::string b64(const std::string& s) {
    static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string o;
    size_t i = 0;
    for (; i + 2 < s.size(); i += 3) {
        const uint32_t w = (uint8_t(s[i]) << 16) | (uint8_t(s[i + 1]) << 8) | uint8_t(s[i + 2]);
        o += T[w >> 18]; o += T[(w >> 12) & 63]; o += T[(w >> 6) & 63]; o += T[w & 63];
    }
    if (i + 1 == s.size()) { const uint32_t w = uint8_t(s[i]) << 16; o += T[w >> 18]; o += T[(w >> 12) & 63]; o += "=="; }
    else if (i + 2 == s.size()) { const uint32_t w = (uint8_t(s[i]) << 16) | (uint8_t(s[i + 1]) << 8); o += T[w >> 18]; o += T[(w >> 12) & 63]; o += T[(w >> 6) & 63]; o += '='; }
    return o;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: train_bpe <n_keys> [max_key_len]\n"); return 2; }
    const size_t n_keys = (size_t)atoll(argv[1]);
    const size_t max_len = argc > 2 ? (size_t)atoll(argv[2]) : 128;
    uint64_t nw = 0;
    if (fread(&nw, 8, 1, stdin) != 1) return 2;
    // byte -> id in tiktoken's byte order (printable bytes first)
    std::vector<int> order;
    for (int b = 33; b <= 126; ++b) order.push_back(b);
    for (int b = 161; b <= 172; ++b) order.push_back(b);
    for (int b = 174; b <= 255; ++b) order.push_back(b);
    for (int b = 0; b < 256; ++b) if (std::find(order.begin(), order.end(), b) == order.end()) order.push_back(b);
    int byte_id[256];
    std::vector<std::string> tok(256);
    for (int i = 0; i < 256; ++i) { byte_id[order[i]] = i; tok[i] = std::string(1, char(order[i])); }
    std::vector<Word> words(nw);
    std::string buf;
    for (uint64_t i = 0; i < nw; ++i) {
        uint32_t len; int64_t cnt;
        if (fread(&len, 4, 1, stdin) != 1 || fread(&cnt, 8, 1, stdin) != 1) return 2;
        buf.resize(len);
        if (len && fread(&buf[0], 1, len, stdin) != len) return 2;
        words[i].cnt = cnt;
        words[i].sym.resize(len);
        for (uint32_t k = 0; k < len; ++k) words[i].sym[k] = byte_id[uint8_t(buf[k])];
    }
    std::unordered_map<uint64_t, int64_t> pc;
    std::unordered_map<uint64_t, std::vector<int32_t>> where;
    pc.reserve(1 << 22); where.reserve(1 << 22);
    auto key = [](int32_t a, int32_t b) { return (uint64_t(uint32_t(a)) << 32) | uint32_t(b); };
    for (size_t w = 0; w < words.size(); ++w) {
        const auto& s = words[w].sym;
        for (size_t k = 0; k + 1 < s.size(); ++k) {
            const uint64_t p = key(s[k], s[k + 1]);
            pc[p] += words[w].cnt;
            auto& v = where[p];
            if (v.empty() || v.back() != (int32_t)w) v.push_back((int32_t)w);
        }
    }
    typedef std::pair<int64_t, uint64_t> HE;                       // (count, ~key): max count first, ties -> smaller key
    std::priority_queue<HE> heap;
    for (auto& kv : pc) heap.push(HE(kv.second, ~kv.first));
    std::unordered_set<std::string> have;
    for (auto& t : tok) have.insert(t);
    while (tok.size() < n_keys && !heap.empty()) {
        const HE top = heap.top(); heap.pop();
        const uint64_t p = ~top.second;
        auto it = pc.find(p);
        if (it == pc.end() || it->second != top.first || top.first <= 0) continue;     // stale entry
        const int32_t a = int32_t(p >> 32), b = int32_t(uint32_t(p));
        const std::string merged = tok[a] + tok[b];
        if (merged.size() > max_len || have.count(merged)) { pc.erase(it); continue; }   // refused for good
        const int32_t id = (int32_t)tok.size();
        tok.push_back(merged); have.insert(merged);
        std::vector<int32_t> ws;
        ws.swap(where[p]);
        where.erase(p);
        pc.erase(p);
        std::unordered_map<uint64_t, int64_t> delta;
        for (int32_t w : ws) {
            auto& s = words[w].sym;
            const int64_t c = words[w].cnt;
            std::vector<int32_t> out;
            out.reserve(s.size());
            bool any = false;
            for (size_t k = 0; k < s.size();) {
                if (k + 1 < s.size() && s[k] == a && s[k + 1] == b) { out.push_back(id); k += 2; any = true; }
                else out.push_back(s[k++]);
            }
            if (!any) continue;
            for (size_t k = 0; k + 1 < s.size(); ++k) delta[key(s[k], s[k + 1])] -= c;
            for (size_t k = 0; k + 1 < out.size(); ++k) {
                const uint64_t q = key(out[k], out[k + 1]);
                delta[q] += c;
                if (out[k] == id || out[k + 1] == id) { auto& v = where[q]; if (v.empty() || v.back() != w) v.push_back(w); }
            }
            s.swap(out);
        }
        for (auto& kv : delta) {
            if (kv.first == p || kv.second == 0) continue;
            auto f = pc.find(kv.first);
            if (f == pc.end()) { if (kv.second > 0) { pc[kv.first] = kv.second; heap.push(HE(kv.second, ~kv.first)); } continue; }
            f->second += kv.second;
            if (f->second > 0) heap.push(HE(f->second, ~kv.first));
        }
    }
    if (tok.size() < n_keys) { fprintf(stderr, "train_bpe: corpus exhausted at %zu keys (wanted %zu)\n", tok.size(), n_keys); return 1; }
    for (size_t i = 0; i < tok.size(); ++i) printf("%s %zu\n", b64(tok[i]).c_str(), i);
    size_t mx = 0, tot = 0;
    for (auto& t : tok) { mx = std::max(mx, t.size()); tot += t.size(); }
    fprintf(stderr, "train_bpe: %zu keys, longest %zu bytes, mean %.2f\n", tok.size(), mx, double(tot) / tok.size());
    return 0;
}
