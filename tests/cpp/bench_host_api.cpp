// Times the ITokenizer-shaped host surface -- tkz::TikTokenizer::EncodeBatchFlat(std::vector<std::string>) of include/tkz_tokenizer.hpp --
// on a batch of documents bench.py hands over in a file: what a C++ host that holds its texts as strings gets, gather and PCIe included
// (`value_host_api` of the bench line; never `value`).
//   argv: vocab.tiktoken  regex-file  sample.bin  [threads]      sample.bin = int64 n_docs, int64 offsets[n_docs + 1], bytes
// Prints one JSON line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <cstring>

#include "tkz_tokenizer.hpp"

static std::string slurp(const char* p) { std::ifstream f(p, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: bench_host_api vocab regex-file sample.bin [threads]\n"); return 2; }
    const int threads = argc > 4 ? std::atoi(argv[4]) : 0;
    try {
        const std::string vocab = slurp(argv[1]), regex = slurp(argv[2]);
        std::vector<std::string> texts;
        int64_t total = 0;
        {
            std::ifstream f(argv[3], std::ios::binary);
            int64_t n = 0;
            f.read(reinterpret_cast<char*>(&n), 8);
            std::vector<int64_t> offs(static_cast<size_t>(n) + 1);
            f.read(reinterpret_cast<char*>(offs.data()), static_cast<std::streamsize>((n + 1) * 8));
            texts.resize(static_cast<size_t>(n));
            for (int64_t d = 0; d < n; ++d) {
                texts[static_cast<size_t>(d)].resize(static_cast<size_t>(offs[d + 1] - offs[d]));
                f.read(&texts[static_cast<size_t>(d)][0], static_cast<std::streamsize>(offs[d + 1] - offs[d]));
            }
            total = offs[static_cast<size_t>(n)];
            if (!f) { std::fprintf(stderr, "short sample file\n"); return 2; }
        }
        tkz::TikTokenizer tok(vocab, {}, regex);
        tkz::FlatBatch fb;
        tok.EncodeBatchFlat(texts, fb, false, threads);                 // untimed: the page-locked buffers and the encoder's staging take their size
        double best = 0, b_off = 0, b_wait = 0, b_enc = 0;
        const int reps = 3;
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            tok.EncodeBatchFlat(texts, fb, false, threads);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (static_cast<double>(total) / s / 1e6 > best) { best = static_cast<double>(total) / s / 1e6; b_off = fb.last_offsets_ms; b_wait = fb.last_wait_ms; b_enc = fb.last_encode_ms; }
        }
        // a position-weighted sum of the ids mod 2^64 (numpy computes the same in one expression): bench.py compares it with the device path's
        uint64_t sum = 0;
        for (int64_t i = 0; i < fb.n_ids(); ++i) sum += (static_cast<uint64_t>(static_cast<uint32_t>(fb.ids()[i])) + 1) * (static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull + 1);
        // ---- the same documents as UTF-16 strings (what a .NET host holds): tkz::TikTokenizer::EncodeBatchFlatUtf16 (page-locked buffers, threaded
        // gather, ONE tkz_encode_batch_utf16 call), and the calls bindings/csharp/GpuTikTokenizer.EncodeBatchFlat makes, as it makes them: the strings
        // copied one after the other by the calling thread into an ordinary (pageable) array, a fresh int array for the ids every call.  The rate is
        // counted in UTF-8 bytes, so that it compares with the figures above.
        double best16 = 0, best16_cs = 0;
        uint64_t sum16 = 0;
        int64_t ntok16 = 0;
        {
            std::vector<std::u16string> t16(texts.size());
            for (size_t d = 0; d < texts.size(); ++d) {
                const std::string& u8 = texts[d];
                std::u16string& o = t16[d];
                o.reserve(u8.size());
                for (size_t i = 0; i < u8.size();) {
                    uint32_t c = static_cast<unsigned char>(u8[i]);
                    int len = c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
                    if (len == 2) c = ((c & 0x1Fu) << 6) | (static_cast<unsigned char>(u8[i + 1]) & 0x3Fu);
                    else if (len == 3) c = ((c & 0x0Fu) << 12) | ((static_cast<unsigned char>(u8[i + 1]) & 0x3Fu) << 6) | (static_cast<unsigned char>(u8[i + 2]) & 0x3Fu);
                    else if (len == 4) c = ((c & 0x07u) << 18) | ((static_cast<unsigned char>(u8[i + 1]) & 0x3Fu) << 12) | ((static_cast<unsigned char>(u8[i + 2]) & 0x3Fu) << 6) | (static_cast<unsigned char>(u8[i + 3]) & 0x3Fu);
                    if (c < 0x10000) o.push_back(static_cast<char16_t>(c));
                    else { c -= 0x10000; o.push_back(static_cast<char16_t>(0xD800 + (c >> 10))); o.push_back(static_cast<char16_t>(0xDC00 + (c & 0x3FF))); }
                    i += static_cast<size_t>(len);
                }
            }
            tkz::FlatBatch fb16;
            tok.EncodeBatchFlatUtf16(t16, fb16, threads);
            for (int r = 0; r < reps; ++r) {
                const auto t0 = std::chrono::steady_clock::now();
                tok.EncodeBatchFlatUtf16(t16, fb16, threads);
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                best16 = std::max(best16, static_cast<double>(total) / s / 1e6);
            }
            ntok16 = fb16.n_ids();
            for (int64_t i = 0; i < fb16.n_ids(); ++i) sum16 += (static_cast<uint64_t>(static_cast<uint32_t>(fb16.ids()[i])) + 1) * (static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull + 1);
            // as the C# class does it
            for (int r = 0; r < reps; ++r) {
                const auto t0 = std::chrono::steady_clock::now();
                std::vector<int64_t> uo(t16.size() + 1, 0);
                int64_t tu = 0;
                for (size_t i = 0; i < t16.size(); ++i) { uo[i] = tu; tu += static_cast<int64_t>(t16[i].size()); }
                uo[t16.size()] = tu;
                std::vector<uint16_t> un(static_cast<size_t>(std::max<int64_t>(1, tu)));
                for (size_t i = 0; i < t16.size(); ++i) std::memcpy(un.data() + uo[i], t16[i].data(), t16[i].size() * 2);
                int64_t cap = std::max<int64_t>(1, std::min<int64_t>(3 * tu, tu / 2 + 4096));
                std::vector<int64_t> so(t16.size() + 1);
                for (;;) {
                    std::vector<int32_t> ids(static_cast<size_t>(cap));
                    int64_t needed = 0;
                    const tkz_status st = tkz_encode_batch_utf16(tok.native(), un.data(), uo.data(), static_cast<int64_t>(t16.size()), ids.data(), cap, so.data(), &needed);
                    if (st == TKZ_E_CAPACITY && needed > cap) { cap = needed; continue; }
                    tkz::check(st);
                    break;
                }
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                best16_cs = std::max(best16_cs, static_cast<double>(total) / s / 1e6);
            }
        }
        std::printf("{\"utf16\": {\"value\": %.1f, \"value_as_the_csharp_class_calls\": %.1f, \"tokens\": %lld, \"ids_checksum\": \"%016llx\"}, ",
                    best16, best16_cs, static_cast<long long>(ntok16), static_cast<unsigned long long>(sum16));
        std::printf("\"value\": %.1f, \"unit\": \"MB/s\", \"docs\": %lld, \"bytes\": %lld, \"tokens\": %lld, \"ids_checksum\": \"%016llx\", \"gather_threads\": %d, \"reps\": %d, \"ms\": {\"call\": %.2f, \"offsets_pass\": %.2f, \"waiting_for_gather\": %.2f, \"in_tkz_encode_batch_utf8\": %.2f}}\n",
                    best, static_cast<long long>(texts.size()), static_cast<long long>(total), static_cast<long long>(fb.n_ids()),
                    static_cast<unsigned long long>(sum), threads, reps, static_cast<double>(total) / best / 1e3, b_off, b_wait, b_enc);
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "bench_host_api: %s\n", ex.what());
        return 1;
    }
    return 0;
}
