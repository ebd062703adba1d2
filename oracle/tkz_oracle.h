/*
 * tkz_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's encode hot path, used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / reported CPU
 * baseline.  Nothing under tokenizer_amd/ links, imports or calls this.
 *
 * What it restates (reference = microsoft/Tokenizer, C# TokenizerLib):
 *   - .tiktoken loading              Tokenizer_C#/TokenizerLib/TikTokenizer.cs:99-139, :84-87
 *   - BytePairEncode                 Tokenizer_C#/TokenizerLib/Utils/BytePairEncoder.cs:13-76
 *   - per-piece driver (plain path)  Tokenizer_C#/TokenizerLib/TikTokenizer.cs:250-274
 *   - LRU piece memo                 Tokenizer_C#/TokenizerLib/Utils/LRUCache.cs:7-136
 *   - special-token segmentation     Tokenizer_C#/TokenizerLib/TikTokenizer.cs:141-170,215-241
 *   - the split regexes              Tokenizer_C#/TokenizerLib/TokenizerBuilder.cs:112 (cl100k),
 *                                    :128,140,155,167 (pattern 1);
 *                                    tokenizer_ts/src/tokenizerBuilder.ts:79-89 (o200k)
 *
 * The regex arithmetic lives in a third-party dependency that is not in the reference tree:
 * System.Text.RegularExpressions of the .NET BCL (no lock file; net6.0 per
 * Tokenizer_C#/TokenizerTest/TokenizerTest.csproj:4 => Unicode 13.0 category data).  The
 * matcher below restates that engine's documented semantics for exactly these three patterns:
 * leftmost-first alternation, greedy quantifiers with backtracking, one UTF-16 code unit per
 * class test (surrogate halves are category Cs), \s == char.IsWhiteSpace, (?i:) with ASCII case
 * pairs only (net6.0 culture lower-casing; U+017F does not fold to 's').
 *
 * PINNING: the oracle is pinned against the reference's own golden vectors that close offline
 * (gpt2 vocab: lib.rs.txt -> tokens_gpt2.json / tokens_r50k_base.json, 11,378 ids; see
 * tests/test_oracle_golden.py).  cl100k / p50k / o200k id-level vectors need vocab files the
 * reference downloads at run time and that are absent here: for those vocabularies parity is
 * "unpinned at id level" (the algorithm is vocabulary-independent; the split for cl100k/o200k is
 * additionally cross-checked against an independent regex engine in tests/test_oracle_regex.py).
 */
#ifndef TKZ_ORACLE_H
#define TKZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pattern ids (same numbering as include/tkz.h on purpose, so tests can pass one through) */
/* o200k exists twice.  The regex string is only in the TypeScript reference (tokenizer_ts/src/tokenizerBuilder.ts:79-89), whose engine
 * (`new RegExp(pattern, "gu")`, tikTokenizer.ts:100) matches by CODE POINT with ECMAScript's \s: TKZO_PATTERN_O200K.  The C# reference
 * can only run that string through TokenizerBuilder.CreateTokenizer(stream, specials, pattern) (TokenizerBuilder.cs:210-213 ->
 * `new Regex(pattern, RegexOptions.Compiled)`, TikTokenizer.cs:77), i.e. by UTF-16 code UNIT (a supplementary-plane char is two OTHER
 * units and never the one-unit prefix) with .NET's \s (U+0085 is white space, U+FEFF is not): TKZO_PATTERN_O200K_DOTNET.  The two
 * differ only on supplementary-plane letters / digits / marks / the prefix rule, U+0085 and U+FEFF. */
enum { TKZO_PATTERN_P1 = 1, TKZO_PATTERN_CL100K = 2, TKZO_PATTERN_O200K = 3, TKZO_PATTERN_O200K_DOTNET = 4 };

/* error codes */
enum {
    TKZO_OK = 0,
    TKZO_E_FORMAT = -1,       /* InvalidOperationException(FormatException)  TikTokenizer.cs:114-136 */
    TKZO_E_DUP_RANK = -2,     /* ArgumentException "sizes don't match"       TikTokenizer.cs:84-87   */
    TKZO_E_KEY_NOT_FOUND = -3,/* KeyNotFoundException                        BytePairEncoder.cs:17,73 */
    TKZO_E_CAPACITY = -4,
    TKZO_E_UTF8 = -5,
    TKZO_E_ARG = -6
};

typedef struct tkzo_vocab tkzo_vocab;
typedef struct tkzo_encoder tkzo_encoder;

/* Parse a .tiktoken image.  Returns NULL and sets *err on failure. */
tkzo_vocab* tkzo_vocab_load(const uint8_t* file, size_t n, int* err);
void tkzo_vocab_free(tkzo_vocab* v);
int64_t tkzo_vocab_size(const tkzo_vocab* v);
int tkzo_vocab_max_key_len(const tkzo_vocab* v);
/* rank of an exact byte string or -1 */
int32_t tkzo_vocab_rank(const tkzo_vocab* v, const uint8_t* key, int len);
/* enumerate: copies key i (insertion order) into buf (cap bytes), returns its length, rank in *rank */
int tkzo_vocab_entry(const tkzo_vocab* v, int64_t i, uint8_t* buf, int cap, int32_t* rank);

/* An encoder = vocab + pattern + memo (cache_size entries; 0 disables the memo). */
tkzo_encoder* tkzo_encoder_create(const tkzo_vocab* v, int pattern, int cache_size);
void tkzo_encoder_free(tkzo_encoder* e);
/* register a special token (UTF-8 literal) -> id; TikTokenizer.cs:78-79 */
int tkzo_encoder_add_special(tkzo_encoder* e, const uint8_t* lit, int len, int32_t id);

/* BytePairEncoder.BytePairEncode(bytes, ranks).  Returns token count or <0. */
int64_t tkzo_bpe(const tkzo_vocab* v, const uint8_t* bytes, int64_t n, int32_t* out, int64_t cap);

/* Regex.Matches over a valid UTF-8 document: writes the BYTE offset of every piece start into
 * starts (cap entries) and returns the piece count (or <0).  The concatenation of pieces may skip
 * units no alternative matches (cannot happen for the three shipped patterns). */
/* Overrides for the tests of tkz_encoder_set_unicode_classes / TKZ_OPT_CASE_EQUIVALENCE (process-wide): classes[cp] in 0..8 for cp < n (n = 65536
 * or 1114112; NULL: the built-in Unicode 13.0 data; the array must stay alive); U+017F as an `s` in cl100k's (?i:...). */
void tkzo_set_unicode_classes(const uint8_t* classes, int64_t n);
void tkzo_set_case_equivalence(int on);
int64_t tkzo_split_utf8(int pattern, const uint8_t* text, int64_t n, int64_t* starts,
                        int64_t* lens, int64_t cap);
/* Same over UTF-16 code units (lone surrogates allowed); offsets are in code units. */
int64_t tkzo_split_utf16(int pattern, const uint16_t* text, int64_t n, int64_t* starts,
                         int64_t* lens, int64_t cap);

/* TikTokenizer.Encode(text, applySpecialTokens:false) / Encode(text, emptySet): plain path. */
int64_t tkzo_encode_utf8(tkzo_encoder* e, const uint8_t* text, int64_t n, int32_t* out, int64_t cap);
int64_t tkzo_encode_utf16(tkzo_encoder* e, const uint16_t* text, int64_t n, int32_t* out, int64_t cap);
/* TikTokenizer.EncodeInternal(text, allowedSpecial): allowed = indices into the registered
 * specials (n_allowed may be 0 => plain path, exactly like TikTokenizer.cs:180-183). */
int64_t tkzo_encode_special_utf8(tkzo_encoder* e, const uint8_t* text, int64_t n,
                                 const int32_t* allowed, int n_allowed, int32_t* out, int64_t cap);

/* Batch helper for the CPU baseline: docs [0,n_docs) given as concatenated UTF-8 + offsets.
 * threads >= 1; docs are statically partitioned, one encoder (own memo) per thread.
 * out_counts[d] = tokens of doc d; ids are written per doc at out + doc_offsets[d]
 * (tokens <= bytes, so the byte offset is always a valid slot).  Returns total tokens or <0. */
/* wall time (ns) of `threads` threads each running the same register-only loop: what parallel throughput the host grants */
int64_t tkzo_parallel_probe(int threads);
/* Checker for batches too large to hold a second copy of: encodes every document and compares it in place with
 * want_ids[want_offsets[d] .. want_offsets[d+1]).  Returns the number of documents that differ (negative: error). */
int64_t tkzo_check_batch(const tkzo_vocab* v, int pattern, int cache_size, const uint8_t* bytes, const int64_t* doc_offsets, int64_t n_docs,
                         const int32_t* want_ids, const int64_t* want_offsets, int threads, int64_t* first_bad, int64_t* tokens);
int64_t tkzo_encode_batch(const tkzo_vocab* v, int pattern, int cache_size, const uint8_t* bytes,
                          const int64_t* doc_offsets, int64_t n_docs, int32_t* out,
                          int32_t* out_counts, int threads);

#ifdef __cplusplus
}
#endif
#endif
