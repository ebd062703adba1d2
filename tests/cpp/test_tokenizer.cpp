// Drives the C++ host mirror (include/tkz_tokenizer.hpp) through the C ABI.  Built by tests/test_cpp_host.py against the
// emulated library on CPU and against libtkz.so on the GPU box.  argv: gpt2.tiktoken lib.rs.txt tokens_gpt2.json
#include <cstdio>
#include <fstream>
#include <sstream>

#include "tkz_tokenizer.hpp"

static std::string slurp(const char* p) { std::ifstream f(p, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string vocab = slurp(argv[1]), text = slurp(argv[2]), golden_json = slurp(argv[3]);
    std::vector<int32_t> golden;
    for (size_t i = 0; i < golden_json.size();) {
        if (golden_json[i] >= '0' && golden_json[i] <= '9') { size_t j = i; int v = 0; while (j < golden_json.size() && golden_json[j] >= '0' && golden_json[j] <= '9') v = v * 10 + (golden_json[j++] - '0'); golden.push_back(v); i = j; }
        else ++i;
    }
    const std::string p1 = "'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
    tkz::SpecialTokens specials = {{"<|endoftext|>", 50256}, {"<|im_start|>", 50300}, {"<|im_end|>", 50301}};
    tkz::TikTokenizer tok(vocab, specials, p1);
    REQUIRE(tok.Encode(text, false) == golden);                                   // TikTokenizerUnitTest.cs:227-245
    REQUIRE(tok.Encode(text, std::vector<std::string>{"<|im_start|>", "<|im_end|>"}) == golden);
    REQUIRE(tok.Encode("", true).empty());
    const auto hw = tok.Encode("Hello World", false);
    const auto sp = tok.Encode("<|im_start|>Hello World<|im_end|>");
    REQUIRE(sp.size() == hw.size() + 2 && sp.front() == 50300 && sp.back() == 50301);
    REQUIRE(std::vector<int32_t>(sp.begin() + 1, sp.end() - 1) == hw);
    const auto plain = tok.Encode("<|im_start|>Hello World<|im_end|>", false);
    REQUIRE(plain.size() > sp.size() && plain.front() != 50300);
    const auto only_end = tok.Encode("<|im_start|>x<|im_end|>", std::vector<std::string>{"<|im_end|>"});
    REQUIRE(only_end.back() == 50301 && only_end.front() != 50300);
    const auto batch = tok.EncodeBatch({"", "Hello World", "<|im_end|>", text.substr(0, 2000)});
    REQUIRE(batch.size() == 4 && batch[0].empty() && batch[1] == hw && batch[2] == std::vector<int32_t>{50301});
    REQUIRE(batch[3] == tok.Encode(text.substr(0, 2000)));
    REQUIRE(tok.EncodeUtf16(u"Hello World") == hw);
    {   // EncodeBatchFlat: the same ids as EncodeBatch, plain (the call's own buffers) and with special tokens spliced in; the FlatBatch is reused
        tkz::FlatBatch fb;
        std::vector<std::string> texts = {"", "Hello World", "<|im_end|>", text.substr(0, 2000), "", "<|im_start|>Hello World<|im_end|>", "<|im_start|>"};
        for (int round = 0; round < 2; ++round) {
            for (int apply = 0; apply < 2; ++apply) {
                const auto ref = tok.EncodeBatch(texts, apply != 0);
                tok.EncodeBatchFlat(texts, fb, apply != 0, round ? 3 : 0);
                REQUIRE(fb.n_texts() == (int64_t)texts.size() && fb.offsets()[0] == 0);
                for (size_t t = 0; t < texts.size(); ++t) REQUIRE(fb.text((int64_t)t) == ref[t]);
                REQUIRE(fb.n_ids() == fb.offsets()[texts.size()]);
            }
            texts.push_back(text);                    // (the second round: a larger batch in the same buffers, gathered by three threads)
            for (int k = 0; k < 40; ++k) texts.push_back(text.substr((size_t)k * 100, 3000));
        }
        tok.EncodeBatchFlat({}, fb);
        REQUIRE(fb.n_texts() == 0 && fb.n_ids() == 0);
        // text that is one token per byte (no merges in the table for these): the first guess of the id buffer is too small, the call is repeated
        const std::string dense(20000, '\x01');
        tok.EncodeBatchFlat({dense, dense, dense, dense}, fb, false);       // (with TKZ_FLAT_SUBBATCH_BYTES = 30000: two sub-batches, the second attempt in both)
        REQUIRE(fb.text(0) == tok.Encode(dense, false) && fb.text(1) == fb.text(0) && fb.text(3) == fb.text(0) && fb.n_ids() == 80000);
        tok.EncodeBatchFlat({"Hello World"}, fb, false);                    // (a sparse batch after a dense one: the buffers are simply large enough)
        REQUIRE(fb.text(0) == hw);
    }
    {
        const std::u16string lone = {u'a', char16_t(0xD83D)}, pair = {char16_t(0xD83D), char16_t(0xDE00), u'b'}, half = {char16_t(0xDE00), u'c'};
        const auto b16 = tok.EncodeBatchUtf16({u"Hello World", u"", lone, half, pair});
        REQUIRE(b16.size() == 5 && b16[0] == hw && b16[1].empty());
        {   // the flat form on UTF-16 strings: the same ids, the FlatBatch's page-locked buffers reused across calls of either kind
            tkz::FlatBatch fb16;
            std::vector<std::u16string> t16 = {u"Hello World", u"", lone, half, pair};
            for (int round = 0; round < 2; ++round) {
                tok.EncodeBatchFlatUtf16(t16, fb16, round ? 2 : 0);
                REQUIRE(fb16.n_texts() == (int64_t)t16.size() && fb16.offsets()[0] == 0 && fb16.n_ids() == fb16.offsets()[t16.size()]);
                const auto ref16 = tok.EncodeBatchUtf16(t16);
                for (size_t t = 0; t < t16.size(); ++t) REQUIRE(fb16.text((int64_t)t) == ref16[t]);
                for (int k = 0; k < 60; ++k) { std::u16string w; for (char ch : text.substr((size_t)k * 97, 2500)) w.push_back((char16_t)(unsigned char)ch); t16.push_back(w); }
            }
            tok.EncodeBatchFlatUtf16({}, fb16);
            REQUIRE(fb16.n_texts() == 0 && fb16.n_ids() == 0);
        }
        REQUIRE(b16[2] == tok.Encode("a\xEF\xBF\xBD", false));                    // a lone high half at the end of a document -> U+FFFD
        REQUIRE(b16[3] == tok.Encode("\xEF\xBF\xBD" "c", false));                 // ... and the low half that starts the next one
        REQUIRE(b16[4] == tok.Encode("\xF0\x9F\x98\x80" "b", false));            // a pair inside one document is one 4-byte char
    }
    // EncodeTrimSuffix / EncodeTrimPrefix (TikTokenizerUnitTest.cs:128-225 restated; expected values from the oracle's TrimOracle)
    {
        const std::string t = "<|im_start|>Hello TempWorld \xF0\x9F\x98\x80 \xE6\xBC\xA2\xE5\xAD\x97<|im_end|>";
        const std::vector<std::string> allow = {"<|endoftext|>", "<|im_start|>", "<|im_end|>"};
        const std::vector<int32_t> full = {50300, 15496, 24189, 10603, 30325, 222, 10545, 120, 95, 27764, 245, 50301};
        REQUIRE(tok.Encode(t) == full);
        auto r = tok.EncodeTrimSuffix(t, allow, 0);
        REQUIRE(r.first.empty() && r.second.empty());
        r = tok.EncodeTrimSuffix(t, allow, 2);
        REQUIRE((r.first == std::vector<int32_t>{50300, 15496}) && r.second == "<|im_start|>Hello");
        r = tok.EncodeTrimSuffix(t, allow, 5);                                     // " 😀" (2 tokens) would make 6: dropped
        REQUIRE((r.first == std::vector<int32_t>{50300, 15496, 24189, 10603}) && r.second == "<|im_start|>Hello TempWorld");
        r = tok.EncodeTrimSuffix(t, 7);                                            // applySpecialTokens = true
        REQUIRE(r.first == std::vector<int32_t>(full.begin(), full.begin() + 6) && r.second == "<|im_start|>Hello TempWorld \xF0\x9F\x98\x80");
        r = tok.EncodeTrimSuffix(t, allow, 12);
        REQUIRE(r.first == full && r.second == t);
        r = tok.EncodeTrimSuffix(t, 5, false);                                     // specials as plain text
        REQUIRE(r.second == "<|im_start" && r.first.size() <= 5);
        r = tok.EncodeTrimPrefix(t, allow, 5);
        REQUIRE((r.first == std::vector<int32_t>{50301}) && r.second == "<|im_end|>");
        r = tok.EncodeTrimPrefix(t, allow, 7);
        REQUIRE(r.first == std::vector<int32_t>(full.begin() + 6, full.end()) && r.second == " \xE6\xBC\xA2\xE5\xAD\x97<|im_end|>");
        r = tok.EncodeTrimPrefix(t, 12);
        REQUIRE(r.first == full && r.second == t);
        r = tok.EncodeTrimPrefix(t, 5, false);
        REQUIRE(r.second == "im_end|>" && r.first.size() <= 5);
        r = tok.EncodeTrimPrefix(t, allow, 0);
        REQUIRE(r.first.empty() && r.second.empty());
    }
    bool threw = false;
    try { tkz::TikTokenizer bad("YQ== 0\nYg== 0\n", {}, p1); } catch (const tkz::DuplicateRankError&) { threw = true; }
    REQUIRE(threw);
    threw = false;
    try { tkz::TikTokenizer bad(vocab, {}, "\\w+"); } catch (const tkz::NotImplementedError&) { threw = true; }
    REQUIRE(threw);
    std::printf("cpp host mirror ok: %zu golden ids\n", golden.size());
    return 0;
}
