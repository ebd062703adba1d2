// tools/dotnet_baseline -- times the REAL reference (Microsoft.DeepDev.TokenizerLib, Tokenizer_C#/TokenizerLib) on a sample of the
// bench corpus, for bench.py's cpu_baseline.reference_dotnet (SURVEY.md 8d).  SOURCE ONLY in this image (no .NET SDK): bench.py
// runs it only when `dotnet` is on PATH and $TKZ_REFERENCE_DIR points at a checkout of microsoft/Tokenizer.
//
//   dotnet run -c Release --project tools/dotnet_baseline -- <sample.bin>
//   sample.bin = int64 n_docs | int64 offsets[n_docs + 1] | bytes;   $TKZ_BENCH_VOCAB = the .tiktoken file;  $TKZ_BENCH_PATTERN = the regex
//
// One TikTokenizer per thread (its LRU cache is locked: a shared instance would serialise), documents statically partitioned,
// Encode(text, applySpecialTokens: false) per document -- the path TikTokenizer.cs:201-205 -> :250-274.
using System;
using System.Collections.Generic;
using System.Diagnostics;
using System.IO;
using System.Text;
using System.Threading.Tasks;
using Microsoft.DeepDev;

internal static class Program
{
    private static int Main(string[] args)
    {
        byte[] raw = File.ReadAllBytes(args[0]);
        long n = BitConverter.ToInt64(raw, 0);
        var offs = new long[n + 1];
        Buffer.BlockCopy(raw, 8, offs, 0, (int)(8 * (n + 1)));
        int data0 = (int)(8 * (n + 2));
        var texts = new string[n];
        for (long d = 0; d < n; ++d) texts[d] = Encoding.UTF8.GetString(raw, data0 + (int)offs[d], (int)(offs[d + 1] - offs[d]));
        byte[] vocab = File.ReadAllBytes(Environment.GetEnvironmentVariable("TKZ_BENCH_VOCAB")!);
        string pattern = Environment.GetEnvironmentVariable("TKZ_BENCH_PATTERN")!;
        int threads = Environment.ProcessorCount;
        var tokenizers = new ITokenizer[threads];
        for (int t = 0; t < threads; ++t)
            tokenizers[t] = TokenizerBuilder.CreateTokenizer(new MemoryStream(vocab), new Dictionary<string, int>(), pattern);
        long tokens = 0;
        var sw = Stopwatch.StartNew();
        Parallel.For(0, threads, new ParallelOptions { MaxDegreeOfParallelism = threads }, t =>
        {
            long mine = 0;
            for (long d = n * t / threads; d < n * (t + 1) / threads; ++d) mine += tokenizers[t].Encode(texts[d], false).Count;
            System.Threading.Interlocked.Add(ref tokens, mine);
        });
        sw.Stop();
        double mbps = offs[n] / sw.Elapsed.TotalSeconds / 1e6;
        Console.WriteLine("{\"value\": " + mbps.ToString("F2", System.Globalization.CultureInfo.InvariantCulture) + ", \"unit\": \"MB/s\", \"cores\": " + threads +
                          ", \"kind\": \"reference\", \"tokens\": " + tokens + ", \"runtime\": \"" + Environment.Version + "\"}");
        return 0;
    }
}
