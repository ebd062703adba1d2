// Times the ITokenizer-shaped host surface -- tkz::TikTokenizer::EncodeBatchFlat(std::vector<std::string>) of include/tkz_tokenizer.hpp --
// on a batch of documents bench.py hands over in a file: what a C++ host that holds its texts as strings gets, gather and PCIe included
// (`value_host_api` of the bench line; never `value`).
//   argv: vocab.tiktoken  regex-file  sample.bin  [threads]      sample.bin = int64 n_docs, int64 offsets[n_docs + 1], bytes
// Prints one JSON line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <future>
#include <thread>
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
        // gather, ONE tkz_encode_batch_utf16 call), and the calls bindings/csharp/GpuTikTokenizer.EncodeBatchFlat makes, as it makes them (below).  The rate
        // is counted in UTF-8 bytes, so that it compares with the figures above.
        double best16 = 0, best16_cs = 0, best16_cs_managed = 0;
        uint64_t sum16 = 0, sum16_cs = 0;
        int64_t ntok16 = 0, ntok16_cs = 0;
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
            // as the C# class does it (bindings/csharp/GpuTikTokenizer, round 6), call for call -- EncodeSegments: sub-batches of 64 M code units, two page-locked
            // unit buffers, the strings of sub-batch k + 1 copied by `gather` threads (Parallel.For over slices) while a task runs tkz_encode_batch_utf16 on
            // sub-batch k, every sub-batch's ids written by the device call straight behind those of the one before into ONE page-locked id buffer sized by the
            // densest batch seen so far.  `pinned` = EncodeBatchFlatPinned (the ids stay there); `managed` = EncodeBatchFlat: a fresh zero-filled array of exactly
            // the ids' number, filled by all cores (what `new int[n]` + Buffer.MemoryCopy under Parallel.For cost).  Warm: a tokenizer's second batch.
            {
                const int64_t kSubUnits = int64_t(64) << 20;
                const size_t nseg = t16.size();
                const unsigned hw = std::thread::hardware_concurrency();
                tkz::PinnedBuffer set_units[2], all_ids;
                double tokens_per_unit = 0;
                for (int managed = 0; managed < 2; ++managed)
                for (int r = 0; r < reps + 1; ++r) {
                    const auto t0 = std::chrono::steady_clock::now();
                    std::vector<int64_t> uo(nseg + 1, 0), so(nseg + 1, 0);
                    int64_t tu = 0;
                    for (size_t i = 0; i < nseg; ++i) { uo[i] = tu; tu += static_cast<int64_t>(t16[i].size()); }
                    uo[nseg] = tu;
                    std::vector<size_t> cuts{0};
                    for (size_t i = 1; i < nseg; ++i) if (uo[i + 1] - uo[cuts.back()] > kSubUnits) cuts.push_back(i);
                    cuts.push_back(nseg);
                    const size_t nsub = cuts.size() - 1;
                    int64_t cap = std::max<int64_t>(1, std::min<int64_t>(3 * tu, std::max<int64_t>(tu / 2 + 4096, static_cast<int64_t>(static_cast<double>(tu) * tokens_per_unit * 1.1))));
                    all_ids.ensure(static_cast<size_t>(cap) * 4);
                    int64_t done = 0;
                    std::future<void> pending;
                    for (size_t k = 0; k < nsub; ++k) {
                        const size_t lo = cuts[k], hi = cuts[k + 1], n = hi - lo;
                        const int64_t u0 = uo[lo], nu = uo[hi] - u0;
                        uint16_t* units = static_cast<uint16_t*>(set_units[k & 1].ensure(static_cast<size_t>(nu + 32) * 2));
                        const size_t slices = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, n / 4096));
                        {
                            std::vector<std::thread> pool;
                            for (size_t sl = 0; sl < slices; ++sl)
                                pool.emplace_back([&, sl] {
                                    const size_t a = lo + n * sl / slices, b = lo + n * (sl + 1) / slices;
                                    for (size_t i = a; i < b; ++i) std::memcpy(units + (uo[i] - u0), t16[i].data(), t16[i].size() * 2);
                                });
                            for (auto& th : pool) th.join();
                        }
                        if (pending.valid()) pending.get();
                        pending = std::async(std::launch::async, [&, lo, n, u0, nu, units] {
                            std::vector<int64_t> rel(n + 1), oo(n + 1, 0);
                            for (size_t i = 0; i <= n; ++i) rel[i] = uo[lo + i] - u0;
                            int64_t needed = 0;
                            for (;;) {
                                const tkz_status st = tkz_encode_batch_utf16(tok.native(), units, rel.data(), static_cast<int64_t>(n), all_ids.as<int32_t>() + done, cap - done, oo.data(), &needed);
                                if (st == TKZ_E_CAPACITY && needed > cap - done) {          // (GrowIdsKeeping: a larger buffer that keeps what is there)
                                    const int64_t want = done + needed + static_cast<int64_t>(static_cast<double>(needed) / static_cast<double>(std::max<int64_t>(1, nu)) * static_cast<double>(tu - u0 - nu) * 1.1) + 4096;
                                    std::vector<int32_t> keep(all_ids.as<int32_t>(), all_ids.as<int32_t>() + done);
                                    all_ids.ensure(static_cast<size_t>(want) * 4);
                                    std::memcpy(all_ids.as<int32_t>(), keep.data(), keep.size() * 4);
                                    cap = want;
                                    continue;
                                }
                                tkz::check(st);
                                break;
                            }
                            for (size_t i = 1; i <= n; ++i) so[lo + i] = done + oo[i];
                            done += needed;
                        });
                    }
                    if (pending.valid()) pending.get();
                    if (tu > 0) tokens_per_unit = std::max(tokens_per_unit, static_cast<double>(done) / static_cast<double>(tu));
                    std::vector<int32_t> arr;
                    if (managed) {                                                       // ids = new int[n] (zero-filled) + Parallel.For of Buffer.MemoryCopy
                        arr.assign(static_cast<size_t>(std::max<int64_t>(1, done)), 0);
                        const size_t parts = static_cast<size_t>(std::max<int64_t>(1, std::min<int64_t>(hw ? hw : 1, done >> 20)));
                        std::vector<std::thread> pool;
                        for (size_t pt = 0; pt < parts; ++pt)
                            pool.emplace_back([&, pt] {
                                const int64_t a = done * static_cast<int64_t>(pt) / static_cast<int64_t>(parts), b = done * static_cast<int64_t>(pt + 1) / static_cast<int64_t>(parts);
                                std::memcpy(arr.data() + a, all_ids.as<int32_t>() + a, static_cast<size_t>(b - a) * 4);
                            });
                        for (auto& th : pool) th.join();
                    }
                    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (r > 0) (managed ? best16_cs_managed : best16_cs) = std::max(managed ? best16_cs_managed : best16_cs, static_cast<double>(total) / s / 1e6);
                    ntok16_cs = so[nseg];
                }
                for (int64_t i = 0; i < ntok16_cs; ++i) sum16_cs += (static_cast<uint64_t>(static_cast<uint32_t>(all_ids.as<int32_t>()[i])) + 1) * (static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull + 1);
            }
        }
        std::printf("{\"utf16\": {\"value\": %.1f, \"value_as_csharp\": %.1f, \"value_as_csharp_managed_array\": %.1f, \"tokens\": %lld, \"ids_checksum\": \"%016llx\", \"as_csharp_same_ids\": %s}, ",
                    best16, best16_cs, best16_cs_managed, static_cast<long long>(ntok16), static_cast<unsigned long long>(sum16), ntok16_cs == ntok16 && sum16_cs == sum16 ? "true" : "false");
        std::printf("\"value\": %.1f, \"unit\": \"MB/s\", \"docs\": %lld, \"bytes\": %lld, \"tokens\": %lld, \"ids_checksum\": \"%016llx\", \"gather_threads\": %d, \"reps\": %d, \"ms\": {\"call\": %.2f, \"offsets_pass\": %.2f, \"waiting_for_gather\": %.2f, \"in_tkz_encode_batch_utf8\": %.2f}}\n",
                    best, static_cast<long long>(texts.size()), static_cast<long long>(total), static_cast<long long>(fb.n_ids()),
                    static_cast<unsigned long long>(sum), threads, reps, static_cast<double>(total) / best / 1e3, b_off, b_wait, b_enc);
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "bench_host_api: %s\n", ex.what());
        return 1;
    }
    return 0;
}
