/*
 * tkz.h -- C ABI of libtkz, the MI355X-native batch BPE encoder.
 *
 * Drop-in boundary for ONE path of microsoft/Tokenizer's TokenizerLib: the plain Encode path
 *   TikTokenizer.Encode(string, List<int>, int, int)      Tokenizer_C#/TokenizerLib/TikTokenizer.cs:250-274
 *   BytePairEncoder.BytePairEncode(byte[], ranks)         Tokenizer_C#/TokenizerLib/Utils/BytePairEncoder.cs:13-76
 * reached through ITokenizer.Encode(text, allowedSpecial) / Encode(text, applySpecialTokens)
 * (Tokenizer_C#/TokenizerLib/ITokenizer.cs:12,28; TikTokenizer.cs:178-207).
 *
 * The reference has no FFI of its own; these entry points are what a
 * P/Invoke layer under `class GpuTikTokenizer : ITokenizer` binds (INTEGRATION.md shows the
 * DllImport stubs; bindings/csharp/ holds the class).  Plain pointers and sizes only; no exceptions
 * cross the boundary: every call returns a tkz_status and tkz_last_error() holds the message for
 * the calling thread.  There is NO CPU fallback: without a usable HIP device every encoder entry
 * point fails with TKZ_E_NO_DEVICE.
 *
 * Error mapping to the reference's exceptions:
 *   TKZ_E_FORMAT        InvalidOperationException(FormatException)   TikTokenizer.cs:114-136
 *   TKZ_E_DUP_RANK      ArgumentException                            TikTokenizer.cs:82-87
 *   TKZ_E_KEY_NOT_FOUND KeyNotFoundException                         BytePairEncoder.cs:17,73
 *   TKZ_E_UNSUPPORTED   NotImplementedException (unknown encoder)    TokenizerBuilder.cs:179
 */
#ifndef TKZ_H
#define TKZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tkz_status {
    TKZ_OK = 0,
    TKZ_E_FORMAT = -1,
    TKZ_E_DUP_RANK = -2,
    TKZ_E_KEY_NOT_FOUND = -3,
    TKZ_E_CAPACITY = -4,      /* out_cap too small; *needed holds the required id count */
    TKZ_E_INVALID_UTF8 = -5,  /* the UTF-8 entry points require well-formed UTF-8 (a C# string always converts to it) */
    TKZ_E_ARG = -6,
    TKZ_E_UNSUPPORTED = -7,   /* pattern string that is not one of the three the reference defines; rank outside [0, 2^27) */
    TKZ_E_DEVICE = -8,        /* HIP runtime error */
    TKZ_E_NO_DEVICE = -9,     /* no HIP device / HIP runtime unusable: there is NO CPU fallback */
    TKZ_E_OUT_OF_MEMORY = -10 /* device memory for the workspace could not be allocated (OutOfMemoryException) */
} tkz_status;

/* The three split regexes the reference defines.  A pattern is an enum, not a regex string:
 * libtkz ships a hand-written scanner per pattern and refuses anything else.
 *   TKZ_PATTERN_P1            gpt2 / r50k_base / p50k_base / p50k_edit   TokenizerBuilder.cs:128,140,155,167
 *   TKZ_PATTERN_CL100K        cl100k_base                                TokenizerBuilder.cs:112
 *   TKZ_PATTERN_O200K_DOTNET  o200k_base as the C# reference runs it: the regex string of tokenizer_ts/src/tokenizerBuilder.ts:79-89 handed to
 *                             TokenizerBuilder.CreateTokenizer(stream, specials, pattern) (TokenizerBuilder.cs:210-213) and compiled by
 *                             `new Regex(pattern, RegexOptions.Compiled)` (TikTokenizer.cs:77): every class test looks at ONE UTF-16 code unit (a
 *                             supplementary-plane char is two "other" units whatever its Unicode class, and never the one-unit prefix of a
 *                             word), \s is .NET's (U+0085 is white space, U+FEFF is not) -- the semantics of the other two patterns.
 *   TKZ_PATTERN_O200K         the same string as the TypeScript reference runs it: `new RegExp(pattern, "gu")` (tokenizer_ts/src/tikTokenizer.ts:100),
 *                             one class test per code POINT, ECMAScript \s (U+FEFF is white space, U+0085 is not).
 * The two o200k variants agree on every text without supplementary-plane chars, U+0085 and U+FEFF. */
typedef enum tkz_pattern { TKZ_PATTERN_P1 = 1, TKZ_PATTERN_CL100K = 2, TKZ_PATTERN_O200K = 3, TKZ_PATTERN_O200K_DOTNET = 4 } tkz_pattern;
/* Which engine's reading of a regex string is wanted (tkz_pattern_from_regex_engine). */
typedef enum tkz_regex_engine { TKZ_ENGINE_DOTNET = 0, TKZ_ENGINE_ECMASCRIPT = 1 } tkz_regex_engine;

typedef struct tkz_vocab tkz_vocab;
typedef struct tkz_encoder tkz_encoder;

/* Message for the last failing call on this thread ("" if none). */
const char* tkz_last_error(void);

/* Replaces TikTokenizer.LoadTikTokenBpe(Stream) + Init's rank-collision check
 * (TikTokenizer.cs:99-139, :74-91).  `file` is the whole .tiktoken image (lines "base64 SP rank"). */
tkz_status tkz_vocab_from_tiktoken(const uint8_t* file, size_t n, tkz_vocab** out);
void tkz_vocab_destroy(tkz_vocab* v);
int64_t tkz_vocab_size(const tkz_vocab* v);
int32_t tkz_vocab_max_key_len(const tkz_vocab* v);
/* Entries of the (id_left, id_right) -> rank table built for the merge loop (informational). */
int64_t tkz_vocab_pair_table_entries(const tkz_vocab* v);
/* Bytes of one device table image (informational): 0 SHORT, 1 MID, 2 LONG (slots + blob), 3 PAIR,
 * 4 the direct-index tables (byte ids + byte-pair ranks); -1 = all of them.  0 for an unknown table. */
int64_t tkz_vocab_table_bytes(const tkz_vocab* v, int32_t which);
/* Encoder[key] on the host copy: rank of an exact byte string, or -1. */
int32_t tkz_vocab_rank(const tkz_vocab* v, const uint8_t* key, int32_t len);

/* The Unicode class the split scanners assign to code points first .. first + n - 1 (0 other, 1 Lu, 2 Ll, 3 Lt, 4 Lm, 5 Lo, 6 M, 7 N,
 * 8 white space: .NET's \s), from the table the device holds (Unicode 13.0, the data of the reference's net6.0 target).  Pattern 1 and
 * cl100k and TKZ_PATTERN_O200K_DOTNET look only code UNITS up (the entries below 0x10000); TKZ_PATTERN_O200K looks code points up
 * (and reads U+FEFF as white space, U+0085 as not).  Informational: lets a host verify the table. */
void tkz_unicode_classes(uint32_t first, int32_t n, uint8_t* out);

/* Map one of the reference's regex strings (exact text) to a tkz_pattern; anything else is
 * TKZ_E_UNSUPPORTED.  Replaces `new Regex(pattern, RegexOptions.Compiled)` (TikTokenizer.cs:77): the result has .NET's semantics
 * (the o200k string gives TKZ_PATTERN_O200K_DOTNET).  tkz_pattern_from_regex_engine names the engine: TKZ_ENGINE_ECMASCRIPT maps the
 * o200k string to TKZ_PATTERN_O200K (the TypeScript reference's reading) and refuses the other two strings, which libtkz implements
 * with .NET's semantics only. */
tkz_status tkz_pattern_from_regex(const char* regex_utf8, int32_t* pattern_out);
tkz_status tkz_pattern_from_regex_engine(const char* regex_utf8, int32_t engine, int32_t* pattern_out);

/* Replaces TokenizerBuilder.CreateTokenizer(stream, specials, pattern, cacheSize)
 * (TokenizerBuilder.cs:210-213) for the plain path: builds the device tables on HIP device
 * `device` (>= 0).  The vocab may be destroyed afterwards. */
tkz_status tkz_encoder_create(const tkz_vocab* v, int32_t pattern, int32_t device, tkz_encoder** out);
void tkz_encoder_destroy(tkz_encoder* e);
int32_t tkz_encoder_device(const tkz_encoder* e);
/* tkz_unicode_classes, read from the table image the encoder's DEVICE holds (downloaded for the call): verifies the upload. */
tkz_status tkz_encoder_unicode_classes(tkz_encoder* e, uint32_t first, int32_t n, uint8_t* out);

/* The HOST's Unicode classification.  The reference's split is whatever the running process's regex engine makes of \p{L}, \p{N}, \p{Lu} ... and \s
 * (`new Regex(pattern, RegexOptions.Compiled)`, TikTokenizer.cs:77): the categories of the runtime's Unicode data -- 13.0 under net6.0, 15.0 under .NET 8,
 * whatever the engine ships in Node.  libtkz is built with the 13.0 table (net6.0, the reference's target); a host on another runtime hands ITS table
 * over: classes[cp] in 0..8 (the codes of tkz_unicode_classes; 8 = what the host's \s matches) for cp < n_code_points, n_code_points = 65536 (a .NET host
 * classifies code UNITS: `char.GetUnicodeCategory` / `char.IsWhiteSpace` over the BMP is all pattern 1, cl100k and TKZ_PATTERN_O200K_DOTNET ever look
 * at) or 1114112 (every code point: TKZ_PATTERN_O200K).  The ASCII range keeps its classes (the same in every Unicode version) and a surrogate code
 * unit is class 0.  classes == NULL: the built-in table again.  Refused with TKZ_E_ARG while a call of the encoder is in flight. */
tkz_status tkz_encoder_set_unicode_classes(tkz_encoder* e, const uint8_t* classes, int64_t n_code_points);

/* Page-locked host memory for the buffers a host hands to the host-buffer entry points: copies from and to it run asynchronously at
 * the PCIe rate (tkz_encode_batch_utf8 cuts a batch of 12 MB or more into document ranges of 16 MB, keeps two launch sequences enqueued ahead and
 * overlaps the upload of a range with the kernels of the two before it and the download of the one before those -- results in memory from this
 * function leave on a copy engine of their own, tkz_encoder_engine_downloads; from pageable memory every copy is staged by the runtime first:
 * about half the rate).  A host without a HIP
 * binding of its own (C#, a plain C++ program) gets such memory here.  The memory is page-locked for every HIP device of the process
 * (hipHostMallocPortable), whichever device is current at the call.  tkz_host_free(NULL) is a no-op. */
tkz_status tkz_host_alloc(size_t bytes, void** out);
void tkz_host_free(void* p);

/* ---- the hot path ------------------------------------------------------------------------ */

/* EncodeBatch over HOST buffers: n_docs documents, document d = bytes[doc_offsets[d] .. doc_offsets[d+1]).
 * Each document is encoded exactly as ITokenizer.Encode(text, applySpecialTokens:false) would encode
 * the string it is the UTF-8 form of.  out_ids receives all ids, document after document;
 * out_offsets (n_docs+1 entries) the id range of each document.  Caller-allocated; out_cap >= total
 * bytes is always sufficient (a token is at least one byte).  On TKZ_E_CAPACITY nothing useful is
 * in out_ids and *needed (if non-NULL) holds the required capacity. */
tkz_status tkz_encode_batch_utf8(tkz_encoder* e, const uint8_t* bytes, const int64_t* doc_offsets,
                                 int64_t n_docs, int32_t* out_ids, int64_t out_cap,
                                 int64_t* out_offsets, int64_t* needed);

/* Same with every buffer already resident in HBM on the encoder's device (d_bytes 16-byte aligned).
 * Work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream); the call returns after
 * the stream has drained and *total_tokens is final (it is also the required capacity when the call
 * returns TKZ_E_CAPACITY). */
tkz_status tkz_encode_batch_device(tkz_encoder* e, const uint8_t* d_bytes, const int64_t* d_doc_offsets,
                                   int64_t n_docs, int64_t total_bytes, int32_t* d_out_ids,
                                   int64_t out_cap, int64_t* d_out_offsets, void* hip_stream,
                                   int64_t* total_tokens);

/* The same in two halves, for callers that keep several batches in flight (or do their own work while one runs):
 * _begin enqueues the batch on `hip_stream` and returns without waiting; _end waits for it, reports errors and
 * *total_tokens exactly as tkz_encode_batch_device, and frees the handle whatever it returns.  A batch that turns
 * out to need a larger internal buffer than its first attempt had (more pieces, longer miss lists, scratch for
 * pieces over 1024 bytes) is run again inside _end.  The buffers must stay valid, and the outputs unread, until
 * _end has returned.  Every pending call holds one workspace of the encoder: about 7 bytes per input byte on English / code text; text where
 * most pieces miss the vocabulary (CJK under an English table) makes the per-sub-tile miss lists grow from 64 entries towards 1024, i.e. from
 * 1.25 to at most 20 more bytes per input byte for that workspace -- they shrink again when later batches do not need them.
 * tkz_encoder_destroy while handles are outstanding is deferred: every such handle's _end returns TKZ_E_ARG and the last one frees the encoder. */
typedef struct tkz_pending tkz_pending;
tkz_status tkz_encode_batch_device_begin(tkz_encoder* e, const uint8_t* d_bytes, const int64_t* d_doc_offsets,
                                         int64_t n_docs, int64_t total_bytes, int32_t* d_out_ids,
                                         int64_t out_cap, int64_t* d_out_offsets, void* hip_stream,
                                         tkz_pending** pending);
tkz_status tkz_encode_batch_device_end(tkz_pending* pending, int64_t* total_tokens);
/* _begin with a place for THIS batch's {n_docs, n_bytes, n_tokens}: d_counts3 (3 int64 on the device, the caller's, may be NULL) is written by
 * the batch's own stream work -- what tkz_comm_allgather_counts_device sends.  Final once _end has returned TKZ_OK (a batch that is run again
 * inside _end writes it again, on the same stream): enqueue the all-gather behind _end and any number of batches may be in flight, each with
 * its own block.  tkz_pending_counts_device returns d_counts3, or -- when it was NULL -- a block of the handle's workspace that is valid
 * until _end (for a gather enqueued between _begin and _end, which sees the FIRST attempt's counts: fine for a caller that treats a batch
 * whose _end reports anything but TKZ_OK, or that had to be run again, as failed). */
tkz_status tkz_encode_batch_device_begin_counts(tkz_encoder* e, const uint8_t* d_bytes, const int64_t* d_doc_offsets,
                                                int64_t n_docs, int64_t total_bytes, int32_t* d_out_ids,
                                                int64_t out_cap, int64_t* d_out_offsets, void* hip_stream,
                                                int64_t* d_counts3, tkz_pending** pending);
const int64_t* tkz_pending_counts_device(const tkz_pending* pending);

/* EncodeBatch for hosts whose strings are UTF-16 (.NET `string`, Java, JavaScript): document d is the code units
 * units[unit_offsets[d] .. unit_offsets[d+1]).  The units are uploaded as they are and converted to UTF-8 ON THE
 * DEVICE exactly as Encoding.UTF8.GetBytes does (TikTokenizer.cs:261: a surrogate pair becomes one 4-byte char, a lone
 * surrogate -- also a pair cut by a document boundary -- becomes EF BF BD); the host is spared the transcode, which
 * is the slow half of feeding `string`s to the UTF-8 entry point.  Results as tkz_encode_batch_utf8;
 * out_cap >= 3 * total units is always sufficient.  Host buffers. */
tkz_status tkz_encode_batch_utf16(tkz_encoder* e, const uint16_t* units, const int64_t* unit_offsets,
                                  int64_t n_docs, int32_t* out_ids, int64_t out_cap,
                                  int64_t* out_offsets, int64_t* needed);

/* Single-string entries for `string` callers.  UTF-16: the split sees the code units as .NET's Regex
 * does (a supplementary-plane char is two "other" units, a lone surrogate one); each piece is
 * converted as Encoding.UTF8.GetBytes does (lone surrogate -> EF BF BD), TikTokenizer.cs:261. */
tkz_status tkz_encode_utf8(tkz_encoder* e, const uint8_t* text, int64_t len, int32_t* out_ids,
                           int64_t out_cap, int64_t* n_out);
tkz_status tkz_encode_utf16(tkz_encoder* e, const uint16_t* text, int64_t len, int32_t* out_ids,
                            int64_t out_cap, int64_t* n_out);

/* EncodeBatch with PIECE granularity -- what EncodeTrimSuffix / EncodeTrimPrefix consume
 * (TikTokenizer.cs:288-341 and :483-519 walk the regex matches of a text and need the token count and
 * the length of each).  Every document is split into its pieces and each piece is encoded:
 * piece k = bytes[piece_byte_offsets[k] .. piece_byte_offsets[k+1]), its ids are
 * out_ids[piece_token_offsets[k] .. piece_token_offsets[k+1]); the pieces of document d are
 * [doc_piece_offsets[d], doc_piece_offsets[d+1]).  piece_cap is the capacity of the two piece arrays
 * minus one (piece_cap >= total bytes is always enough); TKZ_E_CAPACITY with *n_pieces / *needed_ids
 * holding the required sizes otherwise.  Host buffers.  One launch sequence: the piece offsets are built on the device from
 * the piece-start bitmap and the encode kernels mark the token position of every piece start. */
tkz_status tkz_encode_batch_pieces_utf8(tkz_encoder* e, const uint8_t* bytes, const int64_t* doc_offsets,
                                        int64_t n_docs, int32_t* out_ids, int64_t out_cap,
                                        int64_t* doc_piece_offsets, int64_t* piece_byte_offsets,
                                        int64_t* piece_token_offsets, int64_t piece_cap,
                                        int64_t* n_pieces, int64_t* needed_ids);

/* ---- Decode (TikTokenizer.cs:586-604) for a batch ---------------------------------------------
 * Document d of the result is the concatenation of the byte strings of ids[id_offsets[d] .. id_offsets[d+1]): a vocabulary id
 * yields its key, a registered special token its UTF-8 literal, any other id nothing (the reference drops unknown ids silently,
 * :591-599).  The bytes are what the reference hands to Encoding.UTF8.GetString; the string conversion is the host's.
 * tkz_encoder_set_special_tokens registers SpecialTokensDecoder (TikTokenizer.cs:79): literal i = literals_utf8[literal_offsets[i]
 * .. literal_offsets[i+1]) for id ids[i] (a vocabulary id is never shadowed, :591-598).  out_cap too small: TKZ_E_CAPACITY and
 * the required byte count in *total_bytes / *needed. */
tkz_status tkz_encoder_set_special_tokens(tkz_encoder* e, const int32_t* ids, const uint8_t* literals_utf8, const int64_t* literal_offsets, int32_t n);
tkz_status tkz_decode_batch_device(tkz_encoder* e, const int32_t* d_ids, const int64_t* d_id_offsets, int64_t n_docs, int64_t total_ids,
                                   uint8_t* d_out_bytes, int64_t out_cap, int64_t* d_out_offsets, void* hip_stream, int64_t* total_bytes);
tkz_status tkz_decode_batch(tkz_encoder* e, const int32_t* ids, const int64_t* id_offsets, int64_t n_docs, uint8_t* out_bytes, int64_t out_cap,
                            int64_t* out_offsets, int64_t* needed);

/* ---- token shard files -------------------------------------------------------
 * The on-disk form of one rank's EncodeBatch result: a 64-byte header ("TKZSHRD1", version, n_docs, n_tokens, doc_base,
 * token_base), then offsets int64[n_docs + 1] (relative to the shard), then ids int32[n_tokens]; little-endian.  The
 * reference has no batch or file format (it returns List<int>), so there is nothing to be compatible with.  doc_base /
 * token_base are this rank's bases from tkz_shard_bases: the files of all ranks concatenate into the global result.
 * The device form streams straight from HBM through page-locked chunks (copy of chunk k+1 overlaps the write of chunk k). */
tkz_status tkz_shard_write(const char* path, const int32_t* ids, int64_t n_tokens, const int64_t* offsets, int64_t n_docs,
                           int64_t doc_base, int64_t token_base);
tkz_status tkz_shard_write_device(const char* path, int32_t device, const int32_t* d_ids, int64_t n_tokens, const int64_t* d_offsets,
                                  int64_t n_docs, int64_t doc_base, int64_t token_base);
tkz_status tkz_shard_read_header(const char* path, int64_t* n_docs, int64_t* n_tokens, int64_t* doc_base, int64_t* token_base);

/* ---- multi-GPU ------------------------------------------------------------------------------
 * The reference is single-process; the batch path shards by CONTIGUOUS DOCUMENT RANGES, one process per GPU,
 * vocabulary tables replicated, token ids never leave the GPU that produced them.  The only exchange is ONE all-gather of
 * {n_docs, n_bytes, n_tokens} (3 x int64 per rank) per batch, issued directly on RCCL (ncclAllGather over xGMI) -- no torch,
 * no Python needed in the host.  RCCL is bound at run time (dlopen of librccl.so.1): without it tkz_comm_unique_id /
 * tkz_comm_create fail with TKZ_E_UNSUPPORTED, everything else works.
 *
 *   rank 0:      tkz_comm_unique_id(id)        -> distribute the 128 bytes to the other ranks by any means (file, socket, env)
 *   every rank:  tkz_comm_create(id, rank, world, device, &comm)            (ncclCommInitRank; collective)
 *   per batch:   tkz_shard_range(n_docs_total, rank, world, &lo, &hi); encode documents [lo, hi) on this rank's encoder;
 *                tkz_comm_allgather_counts_device(comm, tkz_encoder_counts_device(enc), d_table, stream)   (asynchronous)
 *             or tkz_comm_allgather_counts(comm, n_docs, n_bytes, n_tokens, table)                          (host, blocking)
 *                tkz_shard_bases(table, world, rank, bases, totals): global index of this shard's first document / byte / token */
typedef struct tkz_comm tkz_comm;
enum { TKZ_COMM_ID_BYTES = 128 };
tkz_status tkz_comm_unique_id(uint8_t* id128);
tkz_status tkz_comm_create(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, tkz_comm** out);
void tkz_comm_destroy(tkz_comm* c);
int32_t tkz_comm_world(const tkz_comm* c);     /* as reported by the communicator (ncclCommCount) */
int32_t tkz_comm_rank(const tkz_comm* c);
const char* tkz_comm_backend(const tkz_comm* c);   /* "rccl <major>.<minor>.<patch>" */
/* d_mine: 3 int64 on the device; d_table: world * 3 int64 on the device (row r = rank r's counts); enqueued on hip_stream */
tkz_status tkz_comm_allgather_counts_device(tkz_comm* c, const int64_t* d_mine, int64_t* d_table, void* hip_stream);
tkz_status tkz_comm_allgather_counts(tkz_comm* c, int64_t n_docs, int64_t n_bytes, int64_t n_tokens, int64_t* table);
/* {n_docs, n_bytes, n_tokens} of the encoder's LAST batch (whichever entry point ran it, the single-launch path included), resident on its
 * device and valid once that batch's stream work is done.  One block per encoder: for callers with one batch at a time.  With several in
 * flight (tkz_encode_batch_device_begin, host threads sharing an encoder) give every batch its own block: tkz_encode_batch_device_begin_counts. */
const int64_t* tkz_encoder_counts_device(const tkz_encoder* e);
/* documents [*lo, *hi) of a job of n_docs_total belong to `rank` */
void tkz_shard_range(int64_t n_docs_total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);
/* from a gathered table: bases3 = {doc, byte, token} index of this rank's first unit, totals3 = the job totals */
tkz_status tkz_shard_bases(const int64_t* table, int32_t world, int32_t rank, int64_t* bases3, int64_t* totals3);

/* ---- stage-level entry points (used by the parity tests; same kernels as the hot path) --- */

/* Regex.Matches only: writes the piece-start bitmap (bit i of word i/64 set <=> a piece starts at
 * byte i; total/64 + 1 words, the bit at `total` is a sentinel) for the batch.  Host buffers. */
tkz_status tkz_pretokenize_utf8(tkz_encoder* e, const uint8_t* bytes, const int64_t* doc_offsets,
                                int64_t n_docs, uint64_t* out_bitmap_words);
/* BytePairEncode + whole-piece lookup only: piece p = bytes[piece_offsets[p] .. piece_offsets[p+1])
 * is encoded as ONE regex match (arbitrary bytes allowed).  Host buffers. */
tkz_status tkz_encode_pieces(tkz_encoder* e, const uint8_t* bytes, const int64_t* piece_offsets,
                             int64_t n_pieces, int32_t* out_ids, int64_t out_cap,
                             int64_t* out_offsets, int64_t* needed);

/* Options.  TKZ_OPT_PRETOK_SEQUENTIAL: 1 = split with the one-lane-per-document scanner instead of
 * the position-parallel one; both must give identical bitmaps. */
enum { TKZ_OPT_PRETOK_SEQUENTIAL = 1,
       /* The piece memo: the device form of the reference's LRUCache (LRUCache.cs, used at TikTokenizer.cs:254,270): pieces of up to 16
        * bytes that had to be merged leave their (up to 4) tokens in a 524,288-slot (16 MB) table on the device and later batches take them from
        * there instead of running BytePairEncode again.  A pure memo: ids are identical with and without it.  Value 0 = off, 1 = on
        * (default), 2 = on and emptied.  Set options while no call of the encoder is in flight (2 is refused with TKZ_E_ARG otherwise). */
       TKZ_OPT_PIECE_MEMO = 2,
       /* 1: the batch path counts what it meets (tkz_encoder_piece_stats); 0 (default): it does not -- the counting adds a few atomics per
        * wavefront and one small kernel per batch, so measure with it off. */
       TKZ_OPT_PIECE_STATS = 3,
       /* Promoted pieces.  The memo's answers never change, so the encoder moves its hottest entries into the whole-piece key tables themselves: such
        * a piece is then found by the lookup every piece goes through anyway (TikTokenizer.cs:262 -- with its <= 4 tokens in place of a rank) instead of
        * being listed, looked up in the memo and answered again in every batch.  Same ids by construction.  Value 1 (default): automatic -- the first
        * batch of at least 8 MB on the batch path counts the memo's hits per slot and the hottest entries (at most 65,536) are promoted when it ends (the
        * memo is copied back and the key tables rebuilt on the host: tens of milliseconds, once), once more a gigabyte of text later, and again whenever
        * the text has drifted (TKZ_OPT_ADAPT); 0: never automatically; 2: promote now whatever the memo holds; 3: drop every promotion.  2 and 3 are refused
        * with TKZ_E_ARG while a call is in flight. */
       TKZ_OPT_PROMOTE = 4,
       /* tuning knobs of the automatic promotion: the smallest batch (bytes) that may be a learning batch (default 8 MB), and the most promoted
        * pieces the key tables hold (default 65,536; at most 2^22) */
       TKZ_OPT_PROMOTE_MIN_BYTES = 5, TKZ_OPT_PROMOTE_CAP = 6,
       /* cl100k's `(?i:'s|'t|'re|'ve|'m|'ll|'d)` as the host's engine reads it: 0 (default) = ASCII case pairs only -- net6.0, the reference's target; 1 =
        * with .NET >= 7's case-equivalence tables, under which U+017F (long s) is an `s`: an apostrophe at which a match starts, followed by U+017F,
        * is the contraction (none of the other letters of the seven literals has a non-ASCII equivalent).  No effect on the other patterns (pattern 1 is
        * case-sensitive, o200k lists its case variants). */
       TKZ_OPT_CASE_EQUIVALENCE = 7,
       /* Batches of at most this many bytes are run for latency rather than throughput (default 16 MB; 0: never): at these sizes the call waits for the
        * slowest wavefront of each kernel, so the merge of the long missed pieces is dealt out in units of 4 sub-tiles instead of 64 (93 -> 44 us of a 1 MB
        * call) and its queue of very long pieces gets a full grid.  Same ids either way.  (TKZ_LATENCY_BYTES in the environment sets the value an encoder
        * is created with.) */
       TKZ_OPT_LATENCY_BYTES = 8,
       /* The cache adapts (default 1).  The reference's LRUCache evicts and refills for ever (LRUCache.cs:79-121); the device memo takes no entry once it is
        * full and a promoted piece stays promoted.  With this option on the encoder follows, from batch to batch, the share of pieces that miss its key
        * tables as a whole (counted by a kernel that runs anyway); when that share has left the level at which it settled after the last promotion -- by
        * more than a quarter and a percentage point, either way, 256 MB of text or more after it -- the encoder learns again: every promotion is dropped
        * (so a piece that stopped hitting is not chosen again), the memo is emptied (while no other call is in flight), the next learning window counts
        * hits and the hottest pieces of the text as it is now are promoted.  Also with it on: the learning rounds go on (1, 2, 4, 8 ... GB apart: a change
        * of text that does not move the miss share is still learnt; a round only adds pieces, a list that reaches its cap starts over), and batches smaller
        * than TKZ_OPT_PROMOTE_MIN_BYTES add up to a learning window instead of never learning.  0: learn in the first batch of at least that size and once more a gigabyte later, never again
        * (round 5's behaviour).  Same ids either way. */
       TKZ_OPT_ADAPT = 9 };
tkz_status tkz_encoder_set_option(tkz_encoder* e, int32_t option, int64_t value);

/* ---- measurement ------------------------------------------------------------------------- */

enum { TKZ_K_DOCMARK = 0, TKZ_K_PRETOK = 1, TKZ_K_PROBE = 2 /* k_probe alone */, TKZ_K_SCAN = 3, TKZ_K_PLACE = 4,
       TKZ_K_DOCOFFS = 5, TKZ_K_MERGE_LONG = 6 /* k_giant_order + k_giant_merge + k_long_count / scan / k_long_scatter + k_merge_long_q (k_merge_long) + k_merge_coop */, TKZ_K_MERGE_SHORT = 7 /* k_merge_short alone */,
       TKZ_K_COUNT = 8 };
/* When enabled, every kernel launch of tkz_encode_batch_device is bracketed by HIP events on the
 * launch stream; tkz_encoder_kernel_ms returns the accumulated milliseconds and launch counts per
 * kernel since the last reset (arrays of TKZ_K_COUNT).  Not inside any bracket (microseconds each): the fill of the workspace's zero region,
 * k_doccount2 and the scan of its counts, k_list_stats (which also finds the giant pieces and fills k_merge_coop's queue).
 * A batch that runs the long pieces' kernels beside k_merge_short (tkz_encoder_side_by_side_batches) has k_merge_long_q and k_merge_coop INSIDE the
 * TKZ_K_MERGE_SHORT bracket -- the three side by side, from the fork to the join -- and only what runs in front of them (the giant pieces, the class
 * queue's counting, scan and scatter) in TKZ_K_MERGE_LONG.  A batch of at most TKZ_OPT_LATENCY_BYTES likewise: its TKZ_K_MERGE_SHORT bracket is k_merge_latency
 * (k_merge_short's workgroups and the chunk form of k_merge_long in one launch) + k_merge_coop, its TKZ_K_MERGE_LONG bracket the giant pieces' two kernels. */
tkz_status tkz_encoder_set_profiling(tkz_encoder* e, int32_t enabled);
tkz_status tkz_encoder_kernel_ms(tkz_encoder* e, double* ms, int64_t* launches, int32_t reset);
/* o200k only, informational: of the 4 KiB blocks of the last batch, how many the ASCII block scanner handed on (blocks with multi-byte
 * chars or a state it cannot carry), and how many of those the multi-byte block scanner handed on to the sequential matcher. */
void tkz_encoder_pretok_leftovers(const tkz_encoder* e, int64_t* after_ascii_scanner, int64_t* after_multibyte_scanner);
/* The single-launch path: a host-buffer call (tkz_encode_utf8 / _utf16, tkz_encode_batch_utf8) whose batch is at most 128 KiB in at most
 * 8192 documents (o200k: at most 64 KiB, documents of at most 1 KiB) runs as ONE kernel launch that reads the text from, and writes the ids into, page-locked host memory
 * (no copy commands; 16 launches otherwise).  Informational: how many calls took it, and how many of those the kernel handed back to the
 * batch path (a piece of more than 1024 bytes or one of more than 256 that is not a key, an error to diagnose, workspace to grow). */
void tkz_encoder_small_path_calls(const tkz_encoder* e, int64_t* calls, int64_t* handed_back);
/* development: the shader-clock stamps the last single-launch kernel left at the end of each of its phases (16 values; returns how many) */
int32_t tkz_encoder_small_path_phases(const tkz_encoder* e, int64_t* clocks16);
/* What the batches since the last reset met, with TKZ_OPT_PIECE_STATS on (8 values): [0] batches, [1] pieces (regex matches), [2] pieces of
 * at most 16 bytes that missed the vocabulary as a whole (TikTokenizer.cs:262 -> :268), [3] of 17..1024 bytes (merged a lane each up to 128 bytes, a wavefront each beyond), [4] of more than 1024 bytes (a workgroup each),
 * [5] piece-memo lookups and [6] hits among them (the misses went through BytePairEncode), [7] promoted pieces the key tables hold now
 * (TKZ_OPT_PROMOTE: they count as whole-piece hits).  Whole-piece hit rate =
 * 1 - ([2] + [3] + [4]) / [1]. */
tkz_status tkz_encoder_piece_stats(tkz_encoder* e, int64_t* out8, int32_t reset);
/* TKZ_OPT_ADAPT, informational (8 values; waits for a promotion being built in the background): [0] promotions installed so far, [1] times the encoder
 * decided to learn again, [2] promoted pieces the key tables hold now, [3] replaced table images not freed yet (they go when no call is in flight),
 * [4] the share of pieces that missed the key tables as it settled after the last promotion, in millionths (-1: not settled yet), [5] the same share
 * over the recent batches (-1: none yet), [6] bytes encoded since the key tables last changed, [7] bytes of the learning window in progress. */
tkz_status tkz_encoder_adapt_stats(tkz_encoder* e, int64_t* out8);
/* Slots of the piece memo (TKZ_OPT_PIECE_MEMO) and slots per bucket, informational; the bucket a piece of 1..16 bytes would use
 * (-1: none -- a piece that holds a zero byte never uses the memo).  The tests use the last one to build pieces that contend for one bucket. */
int64_t tkz_encoder_memo_slots(const tkz_encoder* e);
int32_t tkz_encoder_memo_ways(const tkz_encoder* e);
int64_t tkz_encoder_memo_bucket(const tkz_encoder* e, const uint8_t* piece, int32_t len);
/* Allocates, now, the workspace of a batch of up to max_bytes bytes in up to max_docs documents (about 7.8 device bytes per input byte: records, miss
 * lists, bitmaps, scratch, the staging of the host-buffer entry points, streams): replaces the allocations the first batch call of a fresh encoder
 * otherwise makes inside the call.  The reference pays its construction costs in TokenizerBuilder.CreateTokenizer (TokenizerBuilder.cs:210-213) --
 * call this right after tkz_encoder_create.  Batches up to that size then allocate nothing; larger ones grow the workspace as before.  An encoder whose
 * host threads run batches concurrently reserves one workspace per call of this function. */
tkz_status tkz_encoder_reserve(tkz_encoder* e, int64_t max_bytes, int64_t max_docs);
/* Device bytes currently held by the encoder (tables + workspace). */
int64_t tkz_encoder_workspace_bytes(const tkz_encoder* e);
/* Informational: batches whose kernels of the long missed pieces (k_merge_long_q, k_merge_coop) ran BESIDE k_merge_short on streams of their own instead of behind it --
 * a batch above TKZ_OPT_LATENCY_BYTES on a workspace whose previous such batch left at most 2^20 long misses (the launch sequence of a large batch). */
int64_t tkz_encoder_side_by_side_batches(const tkz_encoder* e);
/* Informational: copies of results (ids, offsets) of host-buffer calls that left the device on a copy engine of their own, named through the HSA runtime
 * (csrc/tkz_sdma.h) -- the chunks of a batch of 12 MB or more whose result buffers are page-locked (tkz_host_alloc, hipHostMalloc).  0 when the runtime
 * library or its entry points are missing, or TKZ_D2H_ENGINE=-1: such downloads go through hipMemcpyAsync. */
int64_t tkz_encoder_engine_downloads(const tkz_encoder* e);
const char* tkz_kernel_name(int32_t k);

/* Synthetic corpus of BASELINE.json's configs, generated ON DEVICE by a counter-based generator
 * (csrc/tkz_corpus.h); the same function compiled for the host regenerates any document for spot
 * checks.  kind: 1 = ASCII English/code-like (configs 1,2,4), 2 = mixed UTF-8 CJK+emoji (config 3),
 * 3 = long-context with long single-class runs (config 5), 5 = words of the 4096-word table drawn uniformly, each behind a single space
 * (back to back and taken as ONE document this is the shape of the reference's own benchmark, PerfBenchmark/Program.cs:14-32).
 * d_doc_offsets (n_docs+1 entries, device) is always written; d_bytes (device, capacity cap_bytes) is
 * filled when non-NULL and large enough; *total_bytes is returned either way, so a first call with
 * d_bytes == NULL sizes the buffer.  Document d depends only on (kind, seed, first_doc + d, min_len, max_len). */
tkz_status tkz_corpus_generate_device(int32_t device, int32_t kind, uint64_t seed, int64_t first_doc,
                                      int64_t n_docs, int32_t min_len, int32_t max_len,
                                      int64_t* d_doc_offsets, uint8_t* d_bytes, int64_t cap_bytes,
                                      void* hip_stream, int64_t* total_bytes);
/* Host regeneration of ONE document (returns its byte length); buf may be NULL to query the length. */
int64_t tkz_corpus_generate_doc_host(int32_t kind, uint64_t seed, int64_t doc_index, int32_t min_len,
                                     int32_t max_len, uint8_t* buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TKZ_H */
