// GpuTikTokenizer.cs -- ITokenizer over libtkz (MI355X).  SOURCE ONLY: the build image has no .NET toolchain
// (no dotnet / mono / csc), so this file is not compiled or tested here; the same boundary is exercised through
// ctypes (tokenizer_amd/_native.py) and C++ (include/tkz_tokenizer.hpp).  See INTEGRATION.md.
//
// Written against what the reference's project gives it: netstandard2.0, LangVersion 8.0, Nullable enable, no package
// references beyond the BCL (TokenizerLib.csproj:4-5,10) plus <AllowUnsafeBlocks>.  So: no System.Memory (no Span, no
// AsSpan, no Encoding overloads over spans -- the char* / byte* overloads instead), no System.Range / System.Index (Substring,
// not text[a..b]), no Marshal.PtrToStringUTF8 and no UnmanagedType.LPUTF8Str (UTF-8 strings cross the boundary as byte[]).
// INTEGRATION.md lists every BCL member used with where it comes from.
//
// Drop it into Tokenizer_C#/TokenizerLib next to TikTokenizer.cs.  It keeps the two Encode overloads of
// ITokenizer (ITokenizer.cs:12,28) and adds EncodeBatch; special-token segmentation is the reference's own
// EncodeInternal / FindNextSpecialToken (TikTokenizer.cs:141-170,230-241) with the plain segments sent to the
// GPU in one batch.  The trim variants (TikTokenizer.cs:288-579) get the token count and length of every regex piece
// from tkz_encode_batch_pieces_utf8 and decide the cut on the host; Decode / DecodeBatch run on the device too
// (tkz_decode_batch).  ShardedEncoder at the end is the multi-GPU form: one process per GPU, contiguous document
// ranges, ONE RCCL all-gather of the per-rank counts through libtkz's own communicator (tkz_comm_*), token shard files.
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using System.Runtime.InteropServices;
using System.Text;
using System.Text.RegularExpressions;

namespace Microsoft.DeepDev
{
    internal static class Tkz
    {
        private const string Lib = "tkz";   // libtkz.so

        [DllImport(Lib)] internal static extern IntPtr tkz_last_error();
        [DllImport(Lib)] internal static extern int tkz_vocab_from_tiktoken(byte[] file, UIntPtr n, out IntPtr vocab);
        [DllImport(Lib)] internal static extern void tkz_vocab_destroy(IntPtr vocab);
        [DllImport(Lib)] internal static extern int tkz_pattern_from_regex(byte[] regexUtf8Z, out int pattern);   // .NET semantics (TikTokenizer.cs:77): the o200k string gives TKZ_PATTERN_O200K_DOTNET
        [DllImport(Lib)] internal static extern int tkz_encoder_create(IntPtr vocab, int pattern, int device, out IntPtr encoder);
        [DllImport(Lib)] internal static extern int tkz_encoder_set_option(IntPtr encoder, int option, long value);
        [DllImport(Lib)] internal static extern int tkz_encoder_reserve(IntPtr encoder, long maxBytes, long maxDocs);   // the workspace of batches up to that size, allocated now (TokenizerBuilder.cs:210-213: construction pays, not the first Encode)
        [DllImport(Lib)] internal static extern int tkz_encoder_set_unicode_classes(IntPtr encoder, byte[] classes, long nCodePoints);   // the HOST's Unicode classification (TikTokenizer.cs:77: the running process's regex engine defines the split)
        [DllImport(Lib)] internal static extern int tkz_host_alloc(UIntPtr bytes, out IntPtr p);        // page-locked memory: copies to and from it run at the PCIe rate
        [DllImport(Lib)] internal static extern void tkz_host_free(IntPtr p);
        [DllImport(Lib)] internal static extern void tkz_encoder_destroy(IntPtr encoder);
        [DllImport(Lib)] internal static extern unsafe int tkz_encode_batch_utf8(IntPtr encoder, byte* bytes, long* docOffsets, long nDocs,
                                                                                  int* outIds, long outCap, long* outOffsets, out long needed);
        [DllImport(Lib)] internal static extern unsafe int tkz_encode_utf16(IntPtr encoder, char* text, long len, int* outIds, long outCap, out long nOut);
        [DllImport(Lib)] internal static extern unsafe int tkz_encode_batch_utf16(IntPtr encoder, char* units, long* unitOffsets, long nDocs,
                                                                                   int* outIds, long outCap, long* outOffsets, out long needed);
        [DllImport(Lib)] internal static extern unsafe int tkz_encoder_set_special_tokens(IntPtr encoder, int* ids, byte* literalsUtf8, long* literalOffsets, int n);
        [DllImport(Lib)] internal static extern unsafe int tkz_decode_batch(IntPtr encoder, int* ids, long* idOffsets, long nDocs, byte* outBytes, long outCap, long* outOffsets, out long needed);
        // multi-GPU: the count exchange and the shard arithmetic (include/tkz.h, "multi-GPU"), token shard files
        [DllImport(Lib)] internal static extern int tkz_comm_unique_id(byte[] id128);
        [DllImport(Lib)] internal static extern int tkz_comm_create(byte[] id128, int rank, int world, int device, out IntPtr comm);
        [DllImport(Lib)] internal static extern void tkz_comm_destroy(IntPtr comm);
        [DllImport(Lib)] internal static extern int tkz_comm_world(IntPtr comm);
        [DllImport(Lib)] internal static extern int tkz_comm_allgather_counts(IntPtr comm, long nDocs, long nBytes, long nTokens, long[] table);
        [DllImport(Lib)] internal static extern void tkz_shard_range(long nDocsTotal, int rank, int world, out long lo, out long hi);
        [DllImport(Lib)] internal static extern int tkz_shard_bases(long[] table, int world, int rank, long[] bases3, long[] totals3);
        [DllImport(Lib)] internal static extern int tkz_shard_write(byte[] pathUtf8Z, int[] ids, long nTokens, long[] offsets, long nDocs, long docBase, long tokenBase);
        [DllImport(Lib)] internal static extern unsafe int tkz_encode_batch_pieces_utf8(IntPtr encoder, byte* bytes, long* docOffsets, long nDocs, int* outIds, long outCap,
                                                                                         long* docPieceOffsets, long* pieceByteOffsets, long* pieceTokenOffsets,
                                                                                         long pieceCap, out long nPieces, out long neededIds);

        /// <summary>A C string argument: the UTF-8 bytes and a terminating zero.</summary>
        internal static byte[] Utf8Z(string s)
        {
            var b = new byte[Encoding.UTF8.GetByteCount(s) + 1];
            Encoding.UTF8.GetBytes(s, 0, s.Length, b, 0);
            return b;
        }

        /// <summary>A zero-terminated UTF-8 string owned by the library (tkz_last_error).</summary>
        internal static string Utf8ToString(IntPtr p)
        {
            if (p == IntPtr.Zero) return "";
            int n = 0;
            while (Marshal.ReadByte(p, n) != 0) ++n;
            var b = new byte[n];
            Marshal.Copy(p, b, 0, n);
            return Encoding.UTF8.GetString(b, 0, n);
        }

        internal static void Check(int status)
        {
            if (status == 0) return;
            string msg = Utf8ToString(tkz_last_error());
            switch (status)
            {
                case -1: throw new InvalidOperationException("Failed to load from BPE encoder file stream: " + msg, new FormatException(msg));  // TikTokenizer.cs:133-136
                case -2: throw new ArgumentException(msg);                    // TikTokenizer.cs:84-87
                case -3: throw new KeyNotFoundException(msg);                 // BytePairEncoder.cs:17,73
                case -7: throw new NotImplementedException(msg);              // TokenizerBuilder.cs:179
                case -9: throw new PlatformNotSupportedException(msg);        // no HIP device: there is no CPU fallback in libtkz
                case -10: throw new OutOfMemoryException(msg);                // device memory for the workspace
                default: throw new InvalidOperationException("libtkz status " + status + ": " + msg);
            }
        }
    }

    /// <summary>The native encoder (device tables + workspaces): released exactly once, also when Dispose is never called.</summary>
    internal sealed class EncoderHandle : SafeHandle
    {
        public EncoderHandle(IntPtr h) : base(IntPtr.Zero, true) { SetHandle(h); }
        public override bool IsInvalid => handle == IntPtr.Zero;
        protected override bool ReleaseHandle() { Tkz.tkz_encoder_destroy(handle); return true; }
    }

    /// <summary>Two grow-only page-locked buffers (tkz_host_alloc): the code units on their way to the device, the ids on their way back.  A tokenizer
    /// keeps a pool of these sets; every EncodeBatchFlat in flight rents its own (two for a batch that goes in sub-batches) and returns them.</summary>
    internal sealed class PinnedBuffers : IDisposable
    {
        private IntPtr units, ids; private long unitsCap, idsCap;
        private static IntPtr Ensure(ref IntPtr p, ref long cap, long bytes)
        {
            if (bytes > cap)
            {
                if (p != IntPtr.Zero) Tkz.tkz_host_free(p);
                p = IntPtr.Zero; cap = 0;
                long want = bytes + bytes / 4 + 4096;
                Tkz.Check(Tkz.tkz_host_alloc((UIntPtr)(ulong)want, out p));
                cap = want;
            }
            return p;
        }
        public IntPtr Units(long bytes) => Ensure(ref units, ref unitsCap, bytes);
        public IntPtr Ids(long bytes) => Ensure(ref ids, ref idsCap, bytes);
        /// <summary>A larger id buffer that still holds the first keepBytes of the old one.</summary>
        public unsafe void GrowIdsKeeping(long keepBytes, long newBytes)
        {
            if (newBytes <= idsCap) return;
            long want = newBytes + newBytes / 4 + 4096;
            Tkz.Check(Tkz.tkz_host_alloc((UIntPtr)(ulong)want, out IntPtr p));
            if (ids != IntPtr.Zero)
            {
                if (keepBytes > 0) Buffer.MemoryCopy((void*)ids, (void*)p, want, keepBytes);
                Tkz.tkz_host_free(ids);
            }
            ids = p; idsCap = want;
        }
        public void Dispose()
        {
            if (units != IntPtr.Zero) Tkz.tkz_host_free(units);
            if (ids != IntPtr.Zero) Tkz.tkz_host_free(ids);
            units = ids = IntPtr.Zero; unitsCap = idsCap = 0;
        }
        ~PinnedBuffers() { Dispose(); }
    }

    /// <summary>The result of GpuTikTokenizer.EncodeBatchFlatPinned: the ids of all texts in page-locked memory, text t at [Offsets[t], Offsets[t + 1]).
    /// Owns a buffer set of the tokenizer's pool until it is disposed.</summary>
    public sealed class FlatBatchResult : IDisposable
    {
        private readonly GpuTikTokenizer owner; private PinnedBuffers holder;
        public long[] Offsets { get; }
        public long Count => Offsets[Offsets.Length - 1];
        public IntPtr Ids => holder != null ? holder.Ids(0) : throw new ObjectDisposedException(nameof(FlatBatchResult));
        internal FlatBatchResult(GpuTikTokenizer owner, PinnedBuffers holder, long[] offsets) { this.owner = owner; this.holder = holder; Offsets = offsets; }
        public unsafe int this[long i] => ((int*)Ids)[i];
        /// <summary>The ids of text t as a managed array.</summary>
        public unsafe int[] Text(int t)
        {
            long n = Offsets[t + 1] - Offsets[t];
            var a = new int[n];
            if (n > 0) fixed (int* dst = a) Buffer.MemoryCopy((int*)Ids + Offsets[t], dst, n * 4, n * 4);
            return a;
        }
        public void Dispose() { PinnedBuffers h = System.Threading.Interlocked.Exchange(ref holder, null); if (h != null) owner.ReturnBuffers(h); }
    }

    public sealed class GpuTikTokenizer : ITokenizer, IDisposable
    {
        private readonly EncoderHandle handle;
        private IntPtr encoder => handle.DangerousGetHandle();
        private readonly IReadOnlyDictionary<string, int> specialTokensEncoder;
        private readonly HashSet<string> specialTokens;         // Contains() only: the alternation below keeps the dictionary's order
        private readonly Regex specialTokensRegex;

        /// <summary>Same arguments as TokenizerBuilder.CreateTokenizer (TokenizerBuilder.cs:210-213) plus the HIP device index.</summary>
        /// <param name="reserveBytes">when &gt; 0: the device workspace of batches of up to this many UTF-8 bytes (in up to reserveDocs texts) is allocated here,
        /// by the constructor, instead of inside the first EncodeBatch (tkz_encoder_reserve: ~7.8 device bytes per byte of text)</param>
        public GpuTikTokenizer(Stream tikTokenBpeFileStream, IReadOnlyDictionary<string, int> specialTokensEncoder, string pattern, int cacheSize = 8192, int device = 0,
                               long reserveBytes = 0, long reserveDocs = 0)
        {
            byte[] file;
            using (var ms = new MemoryStream()) { tikTokenBpeFileStream.CopyTo(ms); file = ms.ToArray(); }
            Tkz.Check(Tkz.tkz_pattern_from_regex(Tkz.Utf8Z(pattern), out int pat));   // only the reference's own three patterns are implemented
            Tkz.Check(Tkz.tkz_vocab_from_tiktoken(file, (UIntPtr)file.Length, out IntPtr vocab));
            IntPtr enc;
            try { Tkz.Check(Tkz.tkz_encoder_create(vocab, pat, device, out enc)); }
            finally { Tkz.tkz_vocab_destroy(vocab); }
            handle = new EncoderHandle(enc);
            this.specialTokensEncoder = specialTokensEncoder;
            specialTokens = new HashSet<string>(specialTokensEncoder.Keys);
            // the alternation in the order of specialTokensEncoder.Keys, as the reference builds it (TikTokenizer.cs:78): when one
            // special token is a prefix of another, the order of the alternatives decides the match
            specialTokensRegex = new Regex(string.Join("|", specialTokensEncoder.Keys.Select(s => Regex.Escape(s))), RegexOptions.Compiled);
            RegisterSpecialTokensForDecode();
            // the LRU piece memo (LRUCache.cs; no effect on results) lives on the device with a fixed size: cacheSize says whether it is used
            if (cacheSize <= 0) Tkz.Check(Tkz.tkz_encoder_set_option(handle.DangerousGetHandle(), 2 /* TKZ_OPT_PIECE_MEMO */, 0));
            UseThisRuntimesRegexSemantics();
            if (reserveBytes > 0) Tkz.Check(Tkz.tkz_encoder_reserve(encoder, reserveBytes, Math.Max(1, reserveDocs)));
        }

        /// <summary>The reference's split is `new Regex(pattern, RegexOptions.Compiled)` of the RUNNING process (TikTokenizer.cs:77): \p{L}, \p{N} ... are the
        /// categories of this runtime's Unicode data (13.0 under net6.0, 15.0 under .NET 8) and (?i:...) is this runtime's case folding (ASCII pairs up to
        /// .NET 6; the case-equivalence tables from .NET 7 on, under which U+017F is an `s`).  libtkz is built for net6.0; here it is told what THIS
        /// runtime says: the class of every UTF-16 code unit (all that pattern 1, cl100k and the o200k string through .NET's engine ever look at), and
        /// the case mode.</summary>
        private void UseThisRuntimesRegexSemantics()
        {
            var classes = new byte[65536];
            for (int c = 0; c < 65536; ++c)
            {
                char ch = (char)c;
                byte k;
                switch (char.GetUnicodeCategory(ch))
                {
                    case System.Globalization.UnicodeCategory.UppercaseLetter: k = 1; break;
                    case System.Globalization.UnicodeCategory.LowercaseLetter: k = 2; break;
                    case System.Globalization.UnicodeCategory.TitlecaseLetter: k = 3; break;
                    case System.Globalization.UnicodeCategory.ModifierLetter: k = 4; break;
                    case System.Globalization.UnicodeCategory.OtherLetter: k = 5; break;
                    case System.Globalization.UnicodeCategory.NonSpacingMark:
                    case System.Globalization.UnicodeCategory.SpacingCombiningMark:
                    case System.Globalization.UnicodeCategory.EnclosingMark: k = 6; break;
                    case System.Globalization.UnicodeCategory.DecimalDigitNumber:
                    case System.Globalization.UnicodeCategory.LetterNumber:
                    case System.Globalization.UnicodeCategory.OtherNumber: k = 7; break;
                    // \s of System.Text.RegularExpressions: [\f\n\r\t\v\x85\p{Z}]
                    case System.Globalization.UnicodeCategory.SpaceSeparator:
                    case System.Globalization.UnicodeCategory.LineSeparator:
                    case System.Globalization.UnicodeCategory.ParagraphSeparator: k = 8; break;
                    default: k = (byte)((c >= 9 && c <= 13) || c == 0x85 ? 8 : 0); break;
                }
                classes[c] = k;
            }
            Tkz.Check(Tkz.tkz_encoder_set_unicode_classes(encoder, classes, classes.Length));
            // Environment.Version: 4.x on .NET Framework, 3.1 on .NET Core 3.1, 5 / 6 / 7 / 8 ... on .NET 5+
            Tkz.Check(Tkz.tkz_encoder_set_option(encoder, 7 /* TKZ_OPT_CASE_EQUIVALENCE */, Environment.Version.Major >= 7 ? 1 : 0));
        }

        // SpecialTokensDecoder (TikTokenizer.cs:79) for the device Decode
        private unsafe void RegisterSpecialTokensForDecode()
        {
            if (specialTokensEncoder.Count == 0) return;
            var ids = specialTokensEncoder.Values.ToArray();
            var lits = specialTokensEncoder.Keys.Select(k => Encoding.UTF8.GetBytes(k)).ToArray();
            var offs = new long[lits.Length + 1];
            for (int i = 0; i < lits.Length; ++i) offs[i + 1] = offs[i] + lits[i].Length;
            var blob = new byte[Math.Max(1, offs[lits.Length])];
            for (int i = 0; i < lits.Length; ++i) lits[i].CopyTo(blob, offs[i]);
            fixed (int* pi = ids) fixed (byte* pb = blob) fixed (long* po = offs)
                Tkz.Check(Tkz.tkz_encoder_set_special_tokens(encoder, pi, pb, po, ids.Length));
        }

        public List<int> Encode(string text, IReadOnlyCollection<string> allowedSpecial)
            => EncodeBatch(new[] { text }, allowedSpecial)[0];

        public List<int> Encode(string text, bool applySpecialTokens = true)
            => EncodeBatch(new[] { text }, applySpecialTokens && specialTokens.Count > 0 ? specialTokens : null)[0];

        /// <summary>Encodes every text as Encode(text, allowedSpecial) would; all plain segments go to the GPU as one batch.
        /// One List per text, each filled with ONE AddRange over its segment of the flat result.</summary>
        public List<List<int>> EncodeBatch(IReadOnlyList<string> texts, IReadOnlyCollection<string>? allowedSpecial = null)
        {
            var (ids, offsets) = EncodeBatchFlat(texts, allowedSpecial);
            var result = new List<List<int>>(texts.Count);
            for (int t = 0; t < texts.Count; ++t)
            {
                int n = (int)(offsets[t + 1] - offsets[t]);
                var one = new List<int>(n);
                if (n > 0) one.AddRange(new ArraySegment<int>(ids, (int)offsets[t], n));      // (ICollection<int>.CopyTo: one block copy)
                result.Add(one);
            }
            return result;
        }

        /// <summary>EncodeBatch without a List per text: text t is Ids[Offsets[t] .. Offsets[t + 1]).  When no special token applies
        /// (the reference's plain path, TikTokenizer.cs:180-183,196-199) Offsets is the device call's own output and Ids holds exactly
        /// Offsets[texts.Count] ids; otherwise the special ids are spliced in between the plain segments' ids with block copies.</summary>
        public unsafe (int[] Ids, long[] Offsets) EncodeBatchFlat(IReadOnlyList<string> texts, IReadOnlyCollection<string>? allowedSpecial = null)
        {
            bool plain = allowedSpecial is null || allowedSpecial.Count == 0 || specialTokensEncoder.Count == 0;
            // 1. segmentation on the host: (text index, plain segment) and literal special ids, in order
            var plan = new List<(int text, int special, int segment)>();
            var segments = new List<(string text, int start, int end)>(texts.Count);
            for (int t = 0; t < texts.Count; ++t)
            {
                string text = texts[t];
                if (plain)
                {
                    segments.Add((text, 0, text.Length));                     // (one segment per text, empty ones included: the offsets line up)
                    continue;
                }
                int start = 0;
                while (text.Length > 0)
                {
                    Match next; int startFind = start;
                    while (true)                                              // FindNextSpecialToken (TikTokenizer.cs:230-241)
                    {
                        next = specialTokensRegex.Match(text, startFind);
                        if (!next.Success || allowedSpecial!.Contains(next.Value)) break;
                        startFind = next.Index + 1;
                    }
                    int end = next.Success ? next.Index : text.Length;
                    if (end > start) { plan.Add((t, -1, segments.Count)); segments.Add((text, start, end)); }
                    if (!next.Success) break;
                    plan.Add((t, specialTokensEncoder[next.Value], -1));        // EncodeSpecialToken (:215-220)
                    start = next.Index + next.Length;
                    if (start >= text.Length) break;
                }
            }
            // 2. the plain segments through the device (EncodeSegments below: page-locked buffers, sub-batches, the gather ahead of the device call); the ids
            //    arrive in ONE page-locked buffer and leave it as a managed array of exactly their number -- `new int[]` is zero-filled by the runtime and
            //    then written: on a million texts that copy is the larger part of this method's time, which is why EncodeBatchFlatPinned exists
            var segOffsets = new long[segments.Count + 1];
            int[] ids;
            PinnedBuffers holder = EncodeSegments(segments, segOffsets);
            try
            {
                long n = segOffsets[segments.Count];
                ids = new int[Math.Max(1, n)];
                IntPtr src = holder.Ids(0);
                int parts = (int)Math.Max(1, Math.Min(Environment.ProcessorCount, n >> 20));     // (first touch of a fresh array: page faults, spread over the cores)
                System.Threading.Tasks.Parallel.For(0, parts, part =>
                {
                    long lo = n * part / parts, hi = n * (part + 1) / parts;
                    fixed (int* dst = ids) Buffer.MemoryCopy((int*)src + lo, dst + lo, (hi - lo) * 4, (hi - lo) * 4);
                });
            }
            finally { ReturnBuffers(holder); }
            if (plain) return (ids, segOffsets);
            // 3. splice the special ids in: block copies of the segments' id ranges
            long nSpecial = 0;
            foreach (var item in plan) if (item.segment < 0) ++nSpecial;
            var flat = new int[Math.Max(1, segOffsets[segments.Count] + nSpecial)];
            var offsets = new long[texts.Count + 1];
            long w = 0; int cur = 0;
            foreach (var (t, special, segment) in plan)
            {
                while (cur < t) offsets[++cur] = w;
                if (segment < 0) { flat[w++] = special; continue; }
                long n = segOffsets[segment + 1] - segOffsets[segment];
                Array.Copy(ids, segOffsets[segment], flat, w, n);
                w += n;
            }
            while (cur < texts.Count) offsets[++cur] = w;
            return (flat, offsets);
        }

        private const long SubBatchUnits = 64L << 20;                          // code units per device call of a large batch (128 MB)
        private readonly System.Collections.Concurrent.ConcurrentBag<PinnedBuffers> bufferPool = new System.Collections.Concurrent.ConcurrentBag<PinnedBuffers>();
        private volatile bool disposed;
        private long tokensPerUnitQ20;                                           // the densest batch seen, tokens per code unit << 20: sizes the id buffer of the next one
        private PinnedBuffers RentBuffers()
        {
            if (disposed) throw new ObjectDisposedException(nameof(GpuTikTokenizer));
            return bufferPool.TryTake(out PinnedBuffers b) ? b : new PinnedBuffers();
        }
        internal void ReturnBuffers(PinnedBuffers b)
        {
            if (disposed) { b.Dispose(); GC.SuppressFinalize(b); } else bufferPool.Add(b);
        }

        /// <summary>The plain segments as batches of UTF-16 code units in PAGE-LOCKED memory (tkz_host_alloc; buffer sets are pooled, one per call in flight:
        /// concurrent EncodeBatch callers do not wait for one another -- the library leases a workspace per call as well).  The strings are copied by all
        /// cores (Parallel.For over slices); Encoding.UTF8.GetBytes (TikTokenizer.cs:261) is done for the whole batch on the device by
        /// tkz_encode_batch_utf16.  A large batch goes in SUB-BATCHES of ~128 MB of code units: the gather of sub-batch k + 1 runs while the device encodes
        /// sub-batch k (two unit buffers), as include/tkz_tokenizer.hpp's EncodeBatchFlat does.  Every sub-batch's ids are written by the device call straight
        /// behind those of the one before, into ONE page-locked buffer: the returned set's Ids(0), segOffsets[segments.Count] of them; segOffsets are global.
        /// The caller gives the returned set back with ReturnBuffers.</summary>
        private unsafe PinnedBuffers EncodeSegments(List<(string text, int start, int end)> segments, long[] segOffsets)
        {
            int nseg = segments.Count;
            var unitOffsets = new long[nseg + 1];
            long total = 0;
            for (int i = 0; i < nseg; ++i) { unitOffsets[i] = total; total += segments[i].end - segments[i].start; }
            unitOffsets[nseg] = total;
            var cuts = new List<int> { 0 };                                   // sub-batch k = segments [cuts[k], cuts[k + 1])
            for (int i = 1; i < nseg; ++i)
                if (unitOffsets[i + 1] - unitOffsets[cuts[cuts.Count - 1]] > SubBatchUnits) cuts.Add(i);
            cuts.Add(nseg);
            int nsub = cuts.Count - 1;
            // A code unit is at most three UTF-8 bytes and a token at least one byte, so 3 * total ids always suffice; English text has a token per ~4 units
            // and CJK text about one per unit: the buffer gets room for what the densest batch so far needed (+ 10 %), a token per two units at least, and
            // grows -- keeping what is in it -- when a sub-batch reports TKZ_E_CAPACITY (-4): once in a tokenizer's life per kind of text, not once per batch.
            long learnt = (long)((double)total * (System.Threading.Interlocked.Read(ref tokensPerUnitQ20) / 1048576.0) * 1.1);
            long cap = Math.Max(1, Math.Min(3 * total, Math.Max(total / 2 + 4096, learnt)));
            PinnedBuffers holder = RentBuffers();                             // its Units: sub-batches 0, 2, 4 ...; its Ids: the whole batch's
            PinnedBuffers second = nsub > 1 ? RentBuffers() : null;           // its Units: sub-batches 1, 3, 5 ...
            bool ok = false;
            try
            {
                holder.Ids(cap * 4);
                long[] state = { 0, cap };                                    // ids so far; capacity of the id buffer (both only touched by the one device call in flight)
                System.Threading.Tasks.Task pending = null;
                for (int k = 0; k < nsub; ++k)
                {
                    PinnedBuffers set = (k & 1) == 0 ? holder : second;       // (its Units were last read by sub-batch k - 2, whose call has been waited for)
                    int lo = cuts[k], hi = cuts[k + 1], n = hi - lo;
                    long u0 = unitOffsets[lo], nu = unitOffsets[hi] - u0;
                    IntPtr unitsPtr = set.Units((nu + 32) * 2);               // (an IntPtr: a lambda cannot capture a pointer-typed local)
                    int slices = Math.Max(1, Math.Min(Environment.ProcessorCount, n / 4096));
                    System.Threading.Tasks.Parallel.For(0, slices, sl =>
                    {
                        char* dstUnits = (char*)unitsPtr;
                        int a = lo + (int)((long)n * sl / slices), b = lo + (int)((long)n * (sl + 1) / slices);
                        for (int i = a; i < b; ++i)
                        {
                            int len = segments[i].end - segments[i].start;
                            if (len == 0) continue;
                            fixed (char* src = segments[i].text)
                                Buffer.MemoryCopy(src + segments[i].start, dstUnits + (unitOffsets[i] - u0), (long)len * 2, (long)len * 2);
                        }
                    });
                    if (pending != null) pending.Wait();                      // the device call of sub-batch k - 1 (its exception surfaces here)
                    pending = System.Threading.Tasks.Task.Run(() => EncodeSubBatch(holder, unitsPtr, unitOffsets, lo, n, nu, total - u0, segOffsets, state));
                }
                if (pending != null) pending.Wait();
                if (total > 0)
                {
                    long q = (long)((double)state[0] / total * 1048576.0), seen;
                    while ((seen = System.Threading.Interlocked.Read(ref tokensPerUnitQ20)) < q &&
                           System.Threading.Interlocked.CompareExchange(ref tokensPerUnitQ20, q, seen) != seen) { }
                }
                ok = true;
                return holder;
            }
            finally
            {
                if (second != null) ReturnBuffers(second);
                if (!ok) ReturnBuffers(holder);
            }
        }
        // One device call: the n segments from `lo` on, nu code units at unitsPtr, their ids behind the state[0] ids already in holder's id buffer
        // (capacity state[1]); fills segOffsets[lo + 1 .. lo + n] (global) and advances state[0].  unitsLeft: code units from this sub-batch to the batch's end.
        private unsafe void EncodeSubBatch(PinnedBuffers holder, IntPtr unitsPtr, long[] unitOffsets, int lo, int n, long nu, long unitsLeft, long[] segOffsets, long[] state)
        {
            var rel = new long[n + 1];
            long u0 = unitOffsets[lo];
            for (int i = 0; i <= n; ++i) rel[i] = unitOffsets[lo + i] - u0;
            var outOffs = new long[n + 1];
            long needed;
            bool held = false;
            try
            {
                handle.DangerousAddRef(ref held);                                // (Dispose on another thread: the native encoder outlives this call)
                while (true)
                {
                    long done = state[0], room = state[1] - done;
                    int* pi = (int*)holder.Ids(0) + done;
                    int st;
                    fixed (long* po = rel) fixed (long* poo = outOffs)
                        st = Tkz.tkz_encode_batch_utf16(encoder, (char*)unitsPtr, po, n, pi, room, poo, out needed);
                    if (st == -4 && needed > room)
                    {   // this sub-batch needs `needed`; what follows it, at the same density at least
                        long want = done + needed + (long)((double)needed / Math.Max(1, nu) * (unitsLeft - nu) * 1.1) + 4096;
                        holder.GrowIdsKeeping(done * 4, want * 4);
                        state[1] = want;
                        continue;
                    }
                    Tkz.Check(st);
                    break;
                }
            }
            finally { if (held) handle.DangerousRelease(); }
            for (int i = 1; i <= n; ++i) segOffsets[lo + i] = state[0] + outOffs[i];
            state[0] += needed;
            GC.KeepAlive(holder);                                                // (the buffers' finalizer must not run while the native call reads them)
            GC.KeepAlive(this);
        }

        /// <summary>EncodeBatchFlat with the ids LEFT in page-locked memory (no special tokens: the reference's plain path, TikTokenizer.cs:180-183): text t is
        /// ids [Offsets[t], Offsets[t + 1]) of the result.  Nothing is copied after the device has written the ids -- on a million texts the managed
        /// `int[]` of EncodeBatchFlat (zero-filled by the runtime, then written) costs more than the encoding.  Dispose the result to give the buffers back.</summary>
        public FlatBatchResult EncodeBatchFlatPinned(IReadOnlyList<string> texts)
        {
            var segments = new List<(string text, int start, int end)>(texts.Count);
            for (int t = 0; t < texts.Count; ++t) segments.Add((texts[t], 0, texts[t].Length));
            var offsets = new long[texts.Count + 1];
            PinnedBuffers holder = EncodeSegments(segments, offsets);
            return new FlatBatchResult(this, holder, offsets);
        }

        // One item per regex piece of every plain segment and one per special token, in order: its ids and its length in
        // UTF-16 units (piece.Length / nextSpecial.Value.Length of TikTokenizer.cs:295,367).
        private unsafe List<(int[] Ids, int Length)> PieceItems(string text, IReadOnlyCollection<string>? allowedSpecial)
        {
            var plan = new List<(int special, int start, int end)>();          // special < 0: plain segment text[start..end]
            int start = 0;
            while (text.Length > 0)
            {
                Match? next = null; int end = text.Length;
                if (allowedSpecial != null && allowedSpecial.Count > 0)
                {
                    int startFind = start;
                    while (true)                                              // FindNextSpecialToken (TikTokenizer.cs:230-241)
                    {
                        next = specialTokensRegex.Match(text, startFind);
                        if (!next.Success || allowedSpecial.Contains(next.Value)) break;
                        startFind = next.Index + 1;
                    }
                    if (next.Success) end = next.Index;
                }
                if (end > start) plan.Add((-1, start, end));
                if (next is null || !next.Success) break;
                plan.Add((specialTokensEncoder[next.Value], next.Index, next.Index + next.Length));
                start = next.Index + next.Length;
                if (start >= text.Length) break;
            }
            var segs = plan.Where(p => p.special < 0).ToList();
            var offsets = new long[segs.Count + 1];
            long total = 0;
            fixed (char* pc = text)                                          // (the char* overloads: netstandard2.0 has no Span)
                for (int i = 0; i < segs.Count; ++i) { offsets[i] = total; total += Encoding.UTF8.GetByteCount(pc + segs[i].start, segs[i].end - segs[i].start); }
            offsets[segs.Count] = total;
            var bytes = new byte[Math.Max(1, total)];
            fixed (char* pc = text) fixed (byte* pbytes = bytes)
                for (int i = 0; i < segs.Count; ++i)
                    Encoding.UTF8.GetBytes(pc + segs[i].start, segs[i].end - segs[i].start, pbytes + offsets[i], (int)(offsets[i + 1] - offsets[i]));
            int cap = (int)Math.Max(1, total);
            var ids = new int[cap]; var dpo = new long[segs.Count + 1]; var pbo = new long[cap + 1]; var pto = new long[cap + 1];
            fixed (byte* pb = bytes) fixed (long* po = offsets) fixed (int* pi = ids) fixed (long* pd = dpo) fixed (long* pp = pbo) fixed (long* pt = pto)
                Tkz.Check(Tkz.tkz_encode_batch_pieces_utf8(encoder, pb, po, segs.Count, pi, cap, pd, pp, pt, cap, out _, out _));
            var items = new List<(int[] Ids, int Length)>();
            int k = 0;
            foreach (var (special, s0, e0) in plan)
            {
                if (special >= 0) { items.Add((new[] { special }, e0 - s0)); continue; }
                for (long p = dpo[k]; p < dpo[k + 1]; ++p)
                {
                    var tok = new int[pto[p + 1] - pto[p]];
                    Array.Copy(ids, pto[p], tok, 0, tok.Length);
                    // UTF-16 length of the piece: chars + supplementary-plane chars (lead bytes, 0xF0.. counted twice)
                    int len = 0;
                    for (long b = pbo[p]; b < pbo[p + 1]; ++b) { if ((bytes[b] & 0xC0) != 0x80) ++len; if (bytes[b] >= 0xF0) ++len; }
                    items.Add((tok, len));
                }
                ++k;
            }
            return items;
        }

        public (List<int> TokenIds, string Text) EncodeTrimSuffix(string text, IReadOnlyCollection<string> allowedSpecial, int maxTokenCount)
        {
            var tokenIds = new List<int>();
            int tokenCount = 0, encodeLength = 0;
            foreach (var (ids, length) in PieceItems(text, allowedSpecial))           // the walk of TikTokenizer.cs:288-341 / :343-392
            {
                tokenCount += ids.Length;
                if (tokenCount > maxTokenCount) break;                            // the piece that overflows is dropped, with everything after it
                tokenIds.AddRange(ids);
                encodeLength += length;
                if (tokenCount >= maxTokenCount) break;
            }
            return (tokenIds, encodeLength == text.Length ? text : text.Substring(0, encodeLength));
        }
        public (List<int> TokenIds, string Text) EncodeTrimSuffix(string text, int maxTokenCount, bool applySpecialTokens = true)
            => EncodeTrimSuffix(text, applySpecialTokens && specialTokens.Count > 0 ? specialTokens : null!, maxTokenCount);

        public (List<int> TokenIds, string Text) EncodeTrimPrefix(string text, IReadOnlyCollection<string> allowedSpecial, int maxTokenCount)
        {
            var tokenIds = new List<int>();
            int tokenCount = 0, encodeLength = 0;
            var tokenCountMap = new SortedDictionary<int, int> { { 0, 0 } };          // TikTokenizer.cs:438-441
            foreach (var (ids, length) in PieceItems(text, allowedSpecial))
            {
                tokenCount += ids.Length; encodeLength += length;
                tokenIds.AddRange(ids);
                tokenCountMap[tokenCount] = encodeLength;
            }
            if (tokenCount <= maxTokenCount) return (tokenIds, text);                  // TrimPrefix (:470-483)
            int prefixTokenCount = tokenCount - maxTokenCount, cutTokens = 0, cutLength = 0;
            foreach (var pair in tokenCountMap) if (pair.Key >= prefixTokenCount) { cutTokens = pair.Key; cutLength = pair.Value; break; }
            return (tokenIds.GetRange(cutTokens, tokenIds.Count - cutTokens), text.Substring(cutLength));
        }
        public (List<int> TokenIds, string Text) EncodeTrimPrefix(string text, int maxTokenCount, bool applySpecialTokens = true)
            => EncodeTrimPrefix(text, applySpecialTokens && specialTokens.Count > 0 ? specialTokens : null!, maxTokenCount);

        /// <summary>TikTokenizer.Decode (TikTokenizer.cs:586-604): ids in neither table are dropped; on the device.</summary>
        public string Decode(int[] tokens) => DecodeBatch(new[] { tokens })[0];

        public unsafe List<string> DecodeBatch(IReadOnlyList<int[]> batches)
        {
            var offs = new long[batches.Count + 1];
            for (int i = 0; i < batches.Count; ++i) offs[i + 1] = offs[i] + batches[i].Length;
            var flat = new int[Math.Max(1, offs[batches.Count])];
            for (int i = 0; i < batches.Count; ++i) batches[i].CopyTo(flat, offs[i]);
            var outOffs = new long[batches.Count + 1];
            var bytes = new byte[Math.Max(16, 8 * flat.Length)];
            while (true)
            {
                int st; long needed;
                fixed (int* pi = flat) fixed (long* po = offs) fixed (byte* pb = bytes) fixed (long* poo = outOffs)
                    st = Tkz.tkz_decode_batch(encoder, pi, po, batches.Count, pb, bytes.Length, poo, out needed);
                if (st == -4) { bytes = new byte[needed]; continue; }              // TKZ_E_CAPACITY: `needed` is the exact size
                Tkz.Check(st);
                break;
            }
            var result = new List<string>(batches.Count);
            for (int i = 0; i < batches.Count; ++i) result.Add(Encoding.UTF8.GetString(bytes, (int)outOffs[i], (int)(outOffs[i + 1] - outOffs[i])));
            return result;
        }

        /// <summary>Idempotent.  The pooled page-locked buffers go at once; a buffer set a call in flight holds is freed when that call returns it
        /// (ReturnBuffers), never under it; the native encoder goes when the last call that holds a reference on the SafeHandle has returned.</summary>
        public void Dispose()
        {
            disposed = true;
            while (bufferPool.TryTake(out PinnedBuffers b)) { b.Dispose(); GC.SuppressFinalize(b); }
            handle.Dispose();
        }
    }

    /// <summary>One rank of a sharded job (BASELINE configs[3]): rank r of `world` encodes documents tkz_shard_range(r) on its own GPU;
    /// the only exchange is one RCCL all-gather of {docs, bytes, tokens} per batch (tkz_comm_allgather_counts), from which every rank
    /// gets the global index of its first document and token; results go to one token shard file per rank (tkz_shard_write).
    /// The 128-byte communicator id travels from rank 0 to the others by whatever the host has (a file, a socket, MPI).</summary>
    public sealed class ShardedEncoder : IDisposable
    {
        private readonly IntPtr comm;
        public int Rank { get; }
        public int World { get; }
        public static byte[] NewCommunicatorId() { var id = new byte[128]; Tkz.Check(Tkz.tkz_comm_unique_id(id)); return id; }
        public ShardedEncoder(byte[] communicatorId, int rank, int world, int device)
        {
            Tkz.Check(Tkz.tkz_comm_create(communicatorId, rank, world, device, out comm));
            Rank = rank; World = world;
        }
        public (long Lo, long Hi) MyDocuments(long nDocsTotal) { Tkz.tkz_shard_range(nDocsTotal, Rank, World, out long lo, out long hi); return (lo, hi); }
        /// <summary>After encoding this rank's documents: exchange the counts, write the shard.  Returns the job totals {docs, bytes, tokens}.</summary>
        public long[] Publish(string shardPath, int[] ids, long nTokens, long[] offsets, long nDocs, long nBytes)
        {
            var table = new long[3 * World];
            Tkz.Check(Tkz.tkz_comm_allgather_counts(comm, nDocs, nBytes, nTokens, table));
            var bases = new long[3]; var totals = new long[3];
            Tkz.Check(Tkz.tkz_shard_bases(table, World, Rank, bases, totals));
            Tkz.Check(Tkz.tkz_shard_write(Tkz.Utf8Z(shardPath), ids, nTokens, offsets, nDocs, bases[0], bases[2]));
            return totals;
        }
        public void Dispose() { Tkz.tkz_comm_destroy(comm); }
    }
}
