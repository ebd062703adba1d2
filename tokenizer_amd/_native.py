"""ctypes binding of libtkz (include/tkz.h).  The library is the in-tree HIP build
(tokenizer_amd/lib/libtkz.so, produced by tokenizer_amd/csrc/Makefile or __graft_entry__.build()).
There is no fallback: if the library or a HIP device is missing, construction raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ($TKZ_LIBTKZ: development only -- e.g. the `make DEVPROF=1 OUT=../lib_dev` build with per-stage cycle counters)
DEFAULT_LIB = os.environ.get("TKZ_LIBTKZ") or os.path.join(_HERE, "lib", "libtkz.so")

OK = 0
E_FORMAT, E_DUP_RANK, E_KEY_NOT_FOUND, E_CAPACITY, E_INVALID_UTF8, E_ARG, E_UNSUPPORTED, E_DEVICE, E_NO_DEVICE, E_OUT_OF_MEMORY = range(-1, -11, -1)
ENGINE_DOTNET, ENGINE_ECMASCRIPT = 0, 1
P1, CL100K, O200K, O200K_DOTNET = 1, 2, 3, 4   # O200K: ECMAScript engine (TS reference); O200K_DOTNET: the same string through .NET Regex
OPT_PRETOK_SEQUENTIAL = 1
OPT_PIECE_MEMO = 2
OPT_PIECE_STATS = 3
OPT_PROMOTE_MIN_BYTES, OPT_PROMOTE_CAP = 5, 6
OPT_CASE_EQUIVALENCE = 7    # cl100k's (?i:...) with .NET >= 7's case-equivalence tables ('ſ is 's)
OPT_LATENCY_BYTES = 8       # batches of at most this many bytes merge long missed pieces a wavefront each (tkz.h)
OPT_ADAPT = 9               # 1 (default): the encoder learns again when the text has drifted; small batches add up to a learning window (tkz.h)
OPT_PROMOTE = 4        # 0 / 1: automatic promotion of hot memo entries into the key tables off / on; 2: promote now; 3: drop the promotions
K_NAMES = ["k_docmark", "k_pretok", "k_probe", "k_scan", "k_place", "k_docoffs", "k_merge_long_group", "k_merge_short"]


class TkzError(Exception):
    """A non-OK tkz_status.  `.code` is the status; subclasses mirror the reference's exception types."""

    def __init__(self, code, msg):
        super().__init__("tkz status %d: %s" % (code, msg))
        self.code = code


class FormatError(TkzError):          # InvalidOperationException(FormatException)  TikTokenizer.cs:114-136
    pass


class DuplicateRankError(TkzError):   # ArgumentException                            TikTokenizer.cs:82-87
    pass


class KeyNotFoundError(TkzError, KeyError):   # KeyNotFoundException                 BytePairEncoder.cs:17,73
    pass


class UnsupportedError(TkzError, NotImplementedError):   # NotImplementedException   TokenizerBuilder.cs:179
    pass


_EXC = {E_FORMAT: FormatError, E_DUP_RANK: DuplicateRankError, E_KEY_NOT_FOUND: KeyNotFoundError, E_UNSUPPORTED: UnsupportedError}


class Library:
    def __init__(self, path=None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise RuntimeError("libtkz not built: %s is missing (run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C tokenizer_amd/csrc`); there is no CPU fallback" % path)
        self.path = path
        # PyTorch wheels bundle their own HIP runtime.  If torch is going to be used in this process (device
        # buffers, streams, torch.distributed) it has to initialise first, so that libtkz binds to the same
        # runtime instance; two runtimes in one process leave the second without a GPU.
        if os.path.basename(path) == "libtkz.so":
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.init()
            except ImportError:
                pass
        L = self.L = C.CDLL(path)
        vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
        pv = C.POINTER(C.c_void_p)
        pi64 = C.POINTER(C.c_int64)
        L.tkz_last_error.restype = C.c_char_p
        L.tkz_vocab_from_tiktoken.argtypes = [vp, sz, pv]
        L.tkz_vocab_destroy.argtypes = [vp]
        L.tkz_vocab_destroy.restype = None
        L.tkz_vocab_size.argtypes = [vp]
        L.tkz_vocab_size.restype = i64
        L.tkz_vocab_max_key_len.argtypes = [vp]
        L.tkz_vocab_pair_table_entries.argtypes = [vp]
        L.tkz_vocab_pair_table_entries.restype = i64
        L.tkz_encoder_memo_slots.argtypes = [vp]
        L.tkz_encoder_memo_slots.restype = i64
        L.tkz_encoder_memo_ways.argtypes = [vp]
        L.tkz_encoder_small_path_calls.argtypes = [vp, pi64, pi64]
        L.tkz_encoder_small_path_calls.restype = None
        L.tkz_encoder_small_path_phases.argtypes = [vp, vp]
        L.tkz_encoder_memo_bucket.argtypes = [vp, vp, i32]
        L.tkz_encoder_memo_bucket.restype = i64
        L.tkz_unicode_classes.argtypes = [C.c_uint32, C.c_int32, vp]
        L.tkz_unicode_classes.restype = None
        L.tkz_encoder_unicode_classes.argtypes = [vp, C.c_uint32, C.c_int32, vp]
        L.tkz_encoder_set_unicode_classes.argtypes = [vp, vp, i64]
        L.tkz_encoder_pretok_leftovers.argtypes = [vp, pi64, pi64]
        L.tkz_encoder_pretok_leftovers.restype = None
        L.tkz_vocab_table_bytes.argtypes = [vp, C.c_int32]
        L.tkz_vocab_table_bytes.restype = i64
        L.tkz_vocab_rank.argtypes = [vp, vp, i32]
        L.tkz_pattern_from_regex.argtypes = [C.c_char_p, C.POINTER(i32)]
        L.tkz_pattern_from_regex_engine.argtypes = [C.c_char_p, i32, C.POINTER(i32)]
        L.tkz_encoder_create.argtypes = [vp, i32, i32, pv]
        L.tkz_encoder_destroy.argtypes = [vp]
        L.tkz_encoder_destroy.restype = None
        L.tkz_encoder_device.argtypes = [vp]
        L.tkz_encode_batch_utf8.argtypes = [vp, vp, vp, i64, vp, i64, vp, pi64]
        L.tkz_host_alloc.argtypes = [C.c_size_t, pv]
        L.tkz_host_free.argtypes = [vp]
        L.tkz_host_free.restype = None
        L.tkz_encode_batch_device.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, pi64]
        L.tkz_encode_batch_device_begin.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, C.POINTER(vp)]
        L.tkz_encode_batch_device_end.argtypes = [vp, pi64]
        L.tkz_encode_batch_device_begin_counts.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, vp, C.POINTER(vp)]
        L.tkz_pending_counts_device.argtypes = [vp]
        L.tkz_pending_counts_device.restype = vp
        L.tkz_encode_batch_utf16.argtypes = [vp, vp, vp, i64, vp, i64, vp, pi64]
        L.tkz_encode_utf8.argtypes = [vp, vp, i64, vp, i64, pi64]
        L.tkz_encode_utf16.argtypes = [vp, vp, i64, vp, i64, pi64]
        L.tkz_pretokenize_utf8.argtypes = [vp, vp, vp, i64, vp]
        L.tkz_encode_pieces.argtypes = [vp, vp, vp, i64, vp, i64, vp, pi64]
        L.tkz_encode_batch_pieces_utf8.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, pi64, pi64]
        L.tkz_encoder_set_option.argtypes = [vp, i32, i64]
        L.tkz_encoder_piece_stats.argtypes = [vp, vp, i32]
        L.tkz_encoder_adapt_stats.argtypes = [vp, vp]
        L.tkz_encoder_reserve.argtypes = [vp, i64, i64]
        L.tkz_encoder_set_profiling.argtypes = [vp, i32]
        L.tkz_encoder_kernel_ms.argtypes = [vp, vp, vp, i32]
        L.tkz_encoder_workspace_bytes.argtypes = [vp]
        L.tkz_encoder_workspace_bytes.restype = i64
        L.tkz_encoder_side_by_side_batches.argtypes = [vp]
        L.tkz_encoder_side_by_side_batches.restype = i64
        L.tkz_encoder_engine_downloads.argtypes = [vp]
        L.tkz_encoder_engine_downloads.restype = i64
        L.tkz_kernel_name.argtypes = [i32]
        L.tkz_kernel_name.restype = C.c_char_p
        L.tkz_corpus_generate_device.argtypes = [i32, i32, C.c_uint64, i64, i64, i32, i32, vp, vp, i64, vp, pi64]
        L.tkz_corpus_generate_doc_host.argtypes = [i32, C.c_uint64, i64, i32, i32, vp, i64]
        L.tkz_corpus_generate_doc_host.restype = i64
        L.tkz_shard_range.argtypes = [i64, i32, i32, pi64, pi64]
        L.tkz_shard_range.restype = None
        L.tkz_shard_bases.argtypes = [vp, i32, i32, vp, vp]
        L.tkz_encoder_counts_device.argtypes = [vp]
        L.tkz_encoder_counts_device.restype = vp
        L.tkz_encoder_set_special_tokens.argtypes = [vp, vp, vp, vp, i32]
        L.tkz_decode_batch_device.argtypes = [vp, vp, vp, i64, i64, vp, i64, vp, vp, pi64]
        L.tkz_decode_batch.argtypes = [vp, vp, vp, i64, vp, i64, vp, pi64]
        L.tkz_shard_write.argtypes = [C.c_char_p, vp, i64, vp, i64, i64, i64]
        L.tkz_shard_write_device.argtypes = [C.c_char_p, i32, vp, i64, vp, i64, i64, i64]
        L.tkz_shard_read_header.argtypes = [C.c_char_p, pi64, pi64, pi64, pi64]
        self.has_comm = hasattr(L, "tkz_comm_create")      # (the CPU emulator build of the tests has no communicator)
        if self.has_comm:
            L.tkz_comm_unique_id.argtypes = [vp]
            L.tkz_comm_create.argtypes = [vp, i32, i32, i32, pv]
            L.tkz_comm_destroy.argtypes = [vp]
            L.tkz_comm_destroy.restype = None
            L.tkz_comm_world.argtypes = [vp]
            L.tkz_comm_rank.argtypes = [vp]
            L.tkz_comm_backend.argtypes = [vp]
            L.tkz_comm_backend.restype = C.c_char_p
            L.tkz_comm_allgather_counts_device.argtypes = [vp, vp, vp, vp]
            L.tkz_comm_allgather_counts.argtypes = [vp, i64, i64, i64, vp]

    def check(self, status):
        if status != OK:
            msg = (self.L.tkz_last_error() or b"").decode("utf-8", "replace")
            raise _EXC.get(status, TkzError)(status, msg)


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library()
    return _default


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Vocab:
    """LoadTikTokenBpe + Init's duplicate-rank check (TikTokenizer.cs:99-139, :74-91)."""

    def __init__(self, tiktoken_bytes: bytes, lib: Library = None):
        self.lib = lib or default_library()
        h = C.c_void_p()
        buf = (C.c_uint8 * max(1, len(tiktoken_bytes))).from_buffer_copy(tiktoken_bytes or b"\0")
        self.lib.check(self.lib.L.tkz_vocab_from_tiktoken(buf, len(tiktoken_bytes), C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.L.tkz_vocab_destroy(self._h)
            self._h = None

    def __len__(self):
        return self.lib.L.tkz_vocab_size(self._h)

    @property
    def max_key_len(self):
        return self.lib.L.tkz_vocab_max_key_len(self._h)

    @property
    def pair_table_entries(self):
        return self.lib.L.tkz_vocab_pair_table_entries(self._h)

    def table_bytes(self):
        """Bytes of every device table image: {"short", "mid", "long", "pair", "direct", "total"}."""
        f = self.lib.L.tkz_vocab_table_bytes
        d = {k: int(f(self._h, i)) for i, k in enumerate(("short", "mid", "long", "pair", "direct"))}
        d["total"] = int(f(self._h, -1))
        return d

    def rank(self, key: bytes):
        buf = (C.c_uint8 * max(1, len(key))).from_buffer_copy(key or b"\0")
        return self.lib.L.tkz_vocab_rank(self._h, buf, len(key))


class Encoder:
    """The device encoder (tables in HBM + workspace)."""

    def __init__(self, vocab: Vocab, pattern: int, device: int = 0):
        self.lib = vocab.lib
        h = C.c_void_p()
        self.lib.check(self.lib.L.tkz_encoder_create(vocab._h, pattern, device, C.byref(h)))
        self._h = h
        self.pattern = pattern

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.tkz_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_option(self, opt, value):
        self.lib.check(self.lib.L.tkz_encoder_set_option(self._h, opt, value))

    def set_unicode_classes(self, classes):
        """The host's Unicode classification (tkz_encoder_set_unicode_classes): uint8[65536] (code units) or uint8[1114112] (code points) of class
        codes 0..8, or None for the built-in Unicode 13.0 table."""
        if classes is None:
            self.lib.check(self.lib.L.tkz_encoder_set_unicode_classes(self._h, None, 0))
            return
        a = np.ascontiguousarray(classes, dtype=np.uint8)
        self.lib.check(self.lib.L.tkz_encoder_set_unicode_classes(self._h, a.ctypes.data, len(a)))

    def unicode_classes(self, first, n):
        """Classes of code points first .. first + n - 1 as the DEVICE's table has them (tkz_encoder_unicode_classes)."""
        out = np.zeros(n, np.uint8)
        self.lib.check(self.lib.L.tkz_encoder_unicode_classes(self._h, first, n, out.ctypes.data))
        return out

    def piece_stats(self, reset=False):
        """What the batch path met since the last reset, with OPT_PIECE_STATS on (tkz_encoder_piece_stats)."""
        out = np.zeros(8, np.int64)
        self.lib.check(self.lib.L.tkz_encoder_piece_stats(self._h, out.ctypes.data, 1 if reset else 0))
        pieces, sm, lm, gm, look, hit = (int(out[i]) for i in (1, 2, 3, 4, 5, 6))
        return {"batches": int(out[0]), "pieces": pieces, "short_misses": sm, "long_misses": lm, "giant_pieces": gm,
                "whole_piece_hit_rate": round(1.0 - (sm + lm + gm) / pieces, 5) if pieces else None,
                "memo_lookups": look, "memo_hits": hit, "memo_hit_rate": round(hit / look, 5) if look else None,
                "merged_short": look - hit if look else sm, "promoted_pieces_in_tables": int(out[7])}

    def reserve(self, max_bytes, max_docs):
        """The workspace of batches of up to max_bytes / max_docs, allocated now instead of inside the first batch call (tkz_encoder_reserve)."""
        self.lib.check(self.lib.L.tkz_encoder_reserve(self._h, int(max_bytes), int(max_docs)))

    def adapt_stats(self):
        """TKZ_OPT_ADAPT's bookkeeping (tkz_encoder_adapt_stats): promotions, re-learns, promoted pieces, the settled and the recent miss share."""
        out = np.zeros(8, np.int64)
        self.lib.check(self.lib.L.tkz_encoder_adapt_stats(self._h, out.ctypes.data))
        return {"promotions": int(out[0]), "relearns": int(out[1]), "promoted_pieces": int(out[2]), "retired_images": int(out[3]),
                "settled_miss_share": None if out[4] < 0 else out[4] / 1e6, "recent_miss_share": None if out[5] < 0 else out[5] / 1e6,
                "bytes_since_tables_changed": int(out[6]), "learning_window_bytes": int(out[7])}

    def set_profiling(self, on):
        self.lib.check(self.lib.L.tkz_encoder_set_profiling(self._h, 1 if on else 0))

    def kernel_ms(self, reset=False):
        ms = np.zeros(len(K_NAMES), np.float64)
        n = np.zeros(len(K_NAMES), np.int64)
        self.lib.check(self.lib.L.tkz_encoder_kernel_ms(self._h, ms.ctypes.data, n.ctypes.data, 1 if reset else 0))
        return {K_NAMES[i]: (float(ms[i]), int(n[i])) for i in range(len(K_NAMES))}

    @property
    def memo_slots(self):
        return int(self.lib.L.tkz_encoder_memo_slots(self._h))

    @property
    def memo_ways(self):
        return int(self.lib.L.tkz_encoder_memo_ways(self._h))

    def memo_bucket(self, piece: bytes):
        """Bucket of the piece memo a piece of 1..16 bytes uses (-1: none; pieces holding a zero byte never use the memo)."""
        return int(self.lib.L.tkz_encoder_memo_bucket(self._h, piece, len(piece)))

    def small_path_calls(self):
        """(calls that took the single-launch path for small batches, how many of those it handed back to the batch path)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self.lib.L.tkz_encoder_small_path_calls(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def small_path_phases(self):
        """Shader-clock stamps at the end of the phases of the last single-launch kernel (development)."""
        a = np.zeros(16, np.int64)
        n = self.lib.L.tkz_encoder_small_path_phases(self._h, a.ctypes.data)
        return a[:n].tolist()

    def pretok_leftovers(self):
        """(blocks the o200k ASCII block scanner handed on, blocks the multi-byte block scanner handed on to the sequential matcher) of the last batch."""
        a, b = C.c_int64(0), C.c_int64(0)
        self.lib.L.tkz_encoder_pretok_leftovers(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def workspace_bytes(self):
        return self.lib.L.tkz_encoder_workspace_bytes(self._h)

    @property
    def engine_downloads(self):
        """Copies of host-call results that left the device on a copy engine of their own (tkz_encoder_engine_downloads)."""
        return self.lib.L.tkz_encoder_engine_downloads(self._h)

    @property
    def side_by_side_batches(self):
        """Batches whose long-piece kernels ran beside k_merge_short on streams of their own (tkz_encoder_side_by_side_batches)."""
        return self.lib.L.tkz_encoder_side_by_side_batches(self._h)

    # -- host buffers --
    def encode_batch(self, data: np.ndarray, offsets: np.ndarray, out_cap=None, out=None):
        """EncodeBatch: (ids int32[total_tokens], out_offsets int64[n+1]).  `out` = (ids, out_offsets) arrays to fill
        (e.g. views of page-locked memory: the copies over PCIe then run at link speed instead of through a bounce buffer)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        cap = len(data) if out_cap is None else out_cap
        if out is not None:
            ids, ooff = out
            assert ids.dtype == np.int32 and ooff.dtype == np.int64 and ids.flags.c_contiguous and ooff.flags.c_contiguous
            assert len(ooff) >= n + 1
            cap = len(ids)
        else:
            ids = np.empty(max(1, cap), np.int32)
            ooff = np.empty(n + 1, np.int64)
        needed = C.c_int64(0)
        self.lib.check(self.lib.L.tkz_encode_batch_utf8(self._h, _ptr(data), _ptr(offsets), n, _ptr(ids), cap, _ptr(ooff), C.byref(needed)))
        return ids[:needed.value], ooff[:n + 1]

    def encode_batch_utf16(self, units: np.ndarray, offsets: np.ndarray, out=None):
        """EncodeBatch on UTF-16 code units (uint16[total], offsets int64[n+1] in units); transcoded on the device.
        `out` = (ids, out_offsets) arrays to fill, as in encode_batch."""
        units = np.ascontiguousarray(units, dtype=np.uint16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        if out is not None:
            ids, ooff = out
            assert ids.dtype == np.int32 and ooff.dtype == np.int64 and ids.flags.c_contiguous and ooff.flags.c_contiguous and len(ooff) >= n + 1
            cap = len(ids)
        else:
            cap = max(1, 3 * len(units))
            ids = np.empty(cap, np.int32)
            ooff = np.empty(n + 1, np.int64)
        needed = C.c_int64(0)
        buf = units if len(units) else np.zeros(1, np.uint16)
        self.lib.check(self.lib.L.tkz_encode_batch_utf16(self._h, _ptr(buf), _ptr(offsets), n, _ptr(ids), cap, _ptr(ooff), C.byref(needed)))
        return ids[:needed.value], ooff[:n + 1]

    def encode_pieces(self, data: np.ndarray, offsets: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        ids = np.empty(max(1, len(data)), np.int32)
        ooff = np.empty(n + 1, np.int64)
        needed = C.c_int64(0)
        self.lib.check(self.lib.L.tkz_encode_pieces(self._h, _ptr(data), _ptr(offsets), n, _ptr(ids), len(data), _ptr(ooff), C.byref(needed)))
        return ids[:needed.value], ooff

    def encode_batch_pieces(self, data: np.ndarray, offsets: np.ndarray):
        """EncodeBatch with piece granularity: (ids, doc_piece_offsets[n+1], piece_byte_offsets[np+1], piece_token_offsets[np+1])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        cap = max(1, len(data))
        ids = np.empty(cap, np.int32)
        dpo = np.empty(n + 1, np.int64)
        pbo = np.empty(cap + 1, np.int64)
        pto = np.empty(cap + 1, np.int64)
        npieces, needed = C.c_int64(0), C.c_int64(0)
        self.lib.check(self.lib.L.tkz_encode_batch_pieces_utf8(self._h, _ptr(data), _ptr(offsets), n, _ptr(ids), cap, _ptr(dpo), _ptr(pbo),
                                                               _ptr(pto), cap, C.byref(npieces), C.byref(needed)))
        k = npieces.value
        return ids[:needed.value], dpo, pbo[:k + 1], pto[:k + 1]

    def pretokenize(self, data: np.ndarray, offsets: np.ndarray):
        """Piece-start bitmap as a bool array of len(data) + 1 (the last entry is the sentinel)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        words = np.zeros(len(data) // 64 + 1, np.uint64)
        self.lib.check(self.lib.L.tkz_pretokenize_utf8(self._h, _ptr(data), _ptr(offsets), n, _ptr(words)))
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")
        return bits[:len(data) + 1].astype(bool)

    def encode_utf8(self, text: bytes):
        ids = np.empty(max(1, len(text)), np.int32)
        n = C.c_int64(0)
        buf = np.frombuffer(text, np.uint8) if text else np.zeros(1, np.uint8)
        self.lib.check(self.lib.L.tkz_encode_utf8(self._h, _ptr(buf), len(text), _ptr(ids), len(text), C.byref(n)))
        return ids[:n.value].tolist()

    def encode_utf16(self, units):
        u = np.ascontiguousarray(np.asarray(list(units) + [0], dtype=np.uint16))
        n_units = len(u) - 1
        ids = np.empty(max(1, 3 * n_units), np.int32)
        n = C.c_int64(0)
        self.lib.check(self.lib.L.tkz_encode_utf16(self._h, _ptr(u), n_units, _ptr(ids), 3 * n_units, C.byref(n)))
        return ids[:n.value].tolist()

    # -- Decode --
    def set_special_tokens(self, specials):
        """specials: {literal str: id}.  Registers SpecialTokensDecoder for Decode (TikTokenizer.cs:79)."""
        items = list(specials.items())
        lits = [k.encode("utf-8") for k, _ in items]
        ids = np.asarray([v for _, v in items], dtype=np.int32)
        blob = np.frombuffer(b"".join(lits), np.uint8) if sum(map(len, lits)) else np.zeros(1, np.uint8)
        offs = np.cumsum([0] + [len(x) for x in lits]).astype(np.int64)
        self.lib.check(self.lib.L.tkz_encoder_set_special_tokens(self._h, _ptr(ids) if len(ids) else None, _ptr(blob), _ptr(offs), len(ids)))

    def decode_batch(self, ids: np.ndarray, id_offsets: np.ndarray, out_cap=None):
        """Batch Decode on the device: (bytes uint8[total], byte_offsets int64[n+1])."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        id_offsets = np.ascontiguousarray(id_offsets, dtype=np.int64)
        n = len(id_offsets) - 1
        cap = out_cap if out_cap is not None else max(16, 8 * len(ids))
        while True:
            out = np.empty(max(1, cap), np.uint8)
            ooff = np.empty(n + 1, np.int64)
            needed = C.c_int64(0)
            st = self.lib.L.tkz_decode_batch(self._h, _ptr(ids) if len(ids) else None, _ptr(id_offsets), n, _ptr(out), cap, _ptr(ooff), C.byref(needed))
            if st == E_CAPACITY and out_cap is None:
                cap = needed.value
                continue
            self.lib.check(st)
            return out[:needed.value], ooff

    def decode_batch_device(self, d_ids, d_id_offsets, n_docs, total_ids, d_out, out_cap, d_out_offsets, stream=0):
        tot = C.c_int64(0)
        self.lib.check(self.lib.L.tkz_decode_batch_device(self._h, d_ids, d_id_offsets, n_docs, total_ids, d_out, out_cap, d_out_offsets,
                                                          stream or None, C.byref(tot)))
        return tot.value

    @property
    def counts_device(self):
        """Device pointer to {n_docs, n_bytes, n_tokens} (3 int64) of the last batch."""
        return self.lib.L.tkz_encoder_counts_device(self._h)

    # -- device buffers (raw pointers, e.g. torch tensors' data_ptr()) --
    def encode_batch_device(self, d_bytes, d_offsets, n_docs, total_bytes, d_out_ids, out_cap, d_out_offsets, stream=0):
        tot = C.c_int64(0)
        self.lib.check(self.lib.L.tkz_encode_batch_device(self._h, d_bytes, d_offsets, n_docs, total_bytes, d_out_ids, out_cap,
                                                          d_out_offsets, stream or None, C.byref(tot)))
        return tot.value

    def encode_batch_device_begin(self, d_bytes, d_offsets, n_docs, total_bytes, d_out_ids, out_cap, d_out_offsets, stream=0, d_counts3=0):
        """Enqueues the batch and returns a handle for encode_batch_device_end (several may be in flight).  d_counts3: device pointer to
        3 int64 of the caller's that receive THIS batch's {n_docs, n_bytes, n_tokens} (final once _end has returned)."""
        h = C.c_void_p()
        self.lib.check(self.lib.L.tkz_encode_batch_device_begin_counts(self._h, d_bytes, d_offsets, n_docs, total_bytes, d_out_ids, out_cap,
                                                                       d_out_offsets, stream or None, d_counts3 or None, C.byref(h)))
        return h

    def pending_counts_device(self, pending):
        """Device pointer to the {n_docs, n_bytes, n_tokens} block of a batch in flight (tkz_pending_counts_device)."""
        return self.lib.L.tkz_pending_counts_device(pending)

    def encode_batch_device_end(self, pending):
        tot = C.c_int64(0)
        self.lib.check(self.lib.L.tkz_encode_batch_device_end(pending, C.byref(tot)))
        return tot.value


class Comm:
    """The direct-RCCL communicator of the C ABI (tkz_comm_*): one all-gather of {n_docs, n_bytes, n_tokens} per batch.
    `exchange(id_or_None) -> id`: how the 128-byte id travels from rank 0 to the others (any broadcast the host has)."""

    ID_BYTES = 128

    def __init__(self, rank: int, world: int, device: int, exchange, lib: Library = None):
        self.lib = lib or default_library()
        if not self.lib.has_comm:
            raise RuntimeError("this build of libtkz has no communicator")
        idbuf = (C.c_uint8 * self.ID_BYTES)()
        if rank == 0:
            self.lib.check(self.lib.L.tkz_comm_unique_id(idbuf))
        got = exchange(bytes(idbuf) if rank == 0 else None)
        assert len(got) == self.ID_BYTES
        idbuf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(got)
        h = C.c_void_p()
        self.lib.check(self.lib.L.tkz_comm_create(idbuf, rank, world, device, C.byref(h)))
        self._h = h

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.L.tkz_comm_destroy(self._h)
            self._h = None

    @property
    def world(self):
        return self.lib.L.tkz_comm_world(self._h)

    @property
    def rank(self):
        return self.lib.L.tkz_comm_rank(self._h)

    @property
    def backend(self):
        return self.lib.L.tkz_comm_backend(self._h).decode()

    def allgather_counts_device(self, d_mine, d_table, stream=0):
        """d_mine: device pointer to 3 int64 (Encoder.counts_device); d_table: device pointer to world*3 int64.  Asynchronous."""
        self.lib.check(self.lib.L.tkz_comm_allgather_counts_device(self._h, d_mine, d_table, stream or None))

    def allgather_counts(self, n_docs, n_bytes, n_tokens):
        table = np.zeros(self.world * 3, np.int64)
        self.lib.check(self.lib.L.tkz_comm_allgather_counts(self._h, n_docs, n_bytes, n_tokens, table.ctypes.data))
        return table.reshape(self.world, 3)


def shard_range(n_docs_total, rank, world, lib: Library = None):
    lib = lib or default_library()
    lo, hi = C.c_int64(0), C.c_int64(0)
    lib.L.tkz_shard_range(n_docs_total, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_bases(table, rank, lib: Library = None):
    """table: int64[world, 3] as gathered.  Returns (bases3, totals3) as lists {docs, bytes, tokens}."""
    lib = lib or default_library()
    t = np.ascontiguousarray(table, dtype=np.int64).reshape(-1, 3)
    b, tot = np.zeros(3, np.int64), np.zeros(3, np.int64)
    lib.check(lib.L.tkz_shard_bases(t.ctypes.data, len(t), rank, b.ctypes.data, tot.ctypes.data))
    return b.tolist(), tot.tolist()


def shard_write(path, ids, offsets, doc_base=0, token_base=0, lib: Library = None):
    """tkz_shard_write on host arrays."""
    lib = lib or default_library()
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    lib.check(lib.L.tkz_shard_write(os.fsencode(path), _ptr(ids) if len(ids) else None, len(ids), _ptr(offsets), len(offsets) - 1, doc_base, token_base))


def shard_write_device(path, d_ids, n_tokens, d_offsets, n_docs, doc_base=0, token_base=0, device=0, lib: Library = None):
    """tkz_shard_write_device: straight from HBM (raw device pointers)."""
    lib = lib or default_library()
    lib.check(lib.L.tkz_shard_write_device(os.fsencode(path), device, d_ids, n_tokens, d_offsets, n_docs, doc_base, token_base))


def shard_read_header(path, lib: Library = None):
    lib = lib or default_library()
    v = [C.c_int64(0) for _ in range(4)]
    lib.check(lib.L.tkz_shard_read_header(os.fsencode(path), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)      # n_docs, n_tokens, doc_base, token_base


def corpus_doc_host(kind, seed, doc_index, min_len, max_len, lib: Library = None) -> bytes:
    lib = lib or default_library()
    n = lib.L.tkz_corpus_generate_doc_host(kind, seed, doc_index, min_len, max_len, None, 0)
    buf = np.zeros(max(1, n), np.uint8)
    lib.L.tkz_corpus_generate_doc_host(kind, seed, doc_index, min_len, max_len, buf.ctypes.data, n)
    return buf[:n].tobytes()


def corpus_generate_device(device, kind, seed, first_doc, n_docs, min_len, max_len, d_offsets, d_bytes, cap_bytes, stream=0,
                           lib: Library = None):
    lib = lib or default_library()
    tot = C.c_int64(0)
    lib.check(lib.L.tkz_corpus_generate_device(device, kind, seed, first_doc, n_docs, min_len, max_len, d_offsets, d_bytes or None,
                                               cap_bytes, stream or None, C.byref(tot)))
    return tot.value
