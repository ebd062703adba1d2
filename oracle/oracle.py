"""ctypes binding of the CPU oracle (oracle/tkz_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under tokenizer_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtkz_oracle.so")

P1, CL100K, O200K, O200K_DOTNET = 1, 2, 3, 4   # O200K: ECMAScript engine (TS reference); O200K_DOTNET: the same string through .NET Regex
E_FORMAT, E_DUP_RANK, E_KEY_NOT_FOUND, E_CAPACITY, E_UTF8, E_ARG = -1, -2, -3, -4, -5, -6


def build(force=False):
    src = os.path.join(_HERE, "tkz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, i64, i32, u8p = C.c_void_p, C.c_int64, C.c_int32, C.c_void_p
        L.tkzo_vocab_load.restype = vp
        L.tkzo_vocab_load.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_int)]
        L.tkzo_vocab_free.argtypes = [vp]
        L.tkzo_vocab_size.restype = i64
        L.tkzo_vocab_size.argtypes = [vp]
        L.tkzo_vocab_max_key_len.argtypes = [vp]
        L.tkzo_vocab_rank.restype = i32
        L.tkzo_vocab_rank.argtypes = [vp, u8p, C.c_int]
        L.tkzo_vocab_entry.argtypes = [vp, i64, u8p, C.c_int, C.POINTER(i32)]
        L.tkzo_encoder_create.restype = vp
        L.tkzo_encoder_create.argtypes = [vp, C.c_int, C.c_int]
        L.tkzo_encoder_free.argtypes = [vp]
        L.tkzo_encoder_add_special.argtypes = [vp, u8p, C.c_int, i32]
        L.tkzo_bpe.restype = i64
        L.tkzo_bpe.argtypes = [vp, u8p, i64, vp, i64]
        L.tkzo_split_utf8.restype = i64
        L.tkzo_split_utf8.argtypes = [C.c_int, u8p, i64, vp, vp, i64]
        L.tkzo_split_utf16.restype = i64
        L.tkzo_split_utf16.argtypes = [C.c_int, vp, i64, vp, vp, i64]
        L.tkzo_encode_utf8.restype = i64
        L.tkzo_encode_utf8.argtypes = [vp, u8p, i64, vp, i64]
        L.tkzo_encode_utf16.restype = i64
        L.tkzo_encode_utf16.argtypes = [vp, vp, i64, vp, i64]
        L.tkzo_encode_special_utf8.restype = i64
        L.tkzo_encode_special_utf8.argtypes = [vp, u8p, i64, vp, C.c_int, vp, i64]
        L.tkzo_encode_batch.restype = i64
        L.tkzo_encode_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, i64, vp, vp, C.c_int]
        L.tkzo_parallel_probe.restype = i64
        L.tkzo_parallel_probe.argtypes = [C.c_int]
        L.tkzo_check_batch.restype = i64
        L.tkzo_check_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, i64, vp, vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


def _buf(b):
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))


class Vocab:
    """LoadTikTokenBpe + Init's duplicate-rank check."""

    def __init__(self, data: bytes):
        err = C.c_int(0)
        self._h = lib().tkzo_vocab_load(_buf(data), len(data), C.byref(err))
        if not self._h:
            raise OracleError(err.value)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # (module globals may be gone at interpreter shutdown)
            lib().tkzo_vocab_free(self._h)
            self._h = None

    def __len__(self):
        return lib().tkzo_vocab_size(self._h)

    @property
    def max_key_len(self):
        return lib().tkzo_vocab_max_key_len(self._h)

    def rank(self, key: bytes):
        return lib().tkzo_vocab_rank(self._h, _buf(key), len(key))

    def entries(self):
        n = len(self)
        buf = (C.c_uint8 * 4096)()
        r = C.c_int32(0)
        out = []
        for i in range(n):
            k = lib().tkzo_vocab_entry(self._h, i, buf, 4096, C.byref(r))
            out.append((bytes(buf[:k]), r.value))
        return out

    def bpe(self, piece: bytes):
        out = np.empty(max(1, len(piece)), dtype=np.int32)
        k = lib().tkzo_bpe(self._h, _buf(piece), len(piece), out.ctypes.data, len(out))
        if k < 0:
            raise OracleError(int(k))
        return out[:k].tolist()


_cls_keep = None


def set_unicode_classes(classes):
    """The checker with a caller-supplied Unicode classification (uint8[65536] or uint8[1114112], codes 0..8) instead of the built-in 13.0 data;
    None: the built-in data again.  Process-wide."""
    global _cls_keep
    L = lib()
    L.tkzo_set_unicode_classes.argtypes = [C.c_void_p, C.c_int64]
    L.tkzo_set_unicode_classes.restype = None
    if classes is None:
        L.tkzo_set_unicode_classes(None, 0)
        _cls_keep = None
        return
    a = np.ascontiguousarray(classes, dtype=np.uint8)
    _cls_keep = a                                    # (the C side keeps a reference)
    L.tkzo_set_unicode_classes(a.ctypes.data, len(a))


def set_case_equivalence(on: bool):
    """cl100k's (?i:...) with .NET >= 7's case-equivalence tables (U+017F is an s).  Process-wide."""
    L = lib()
    L.tkzo_set_case_equivalence.argtypes = [C.c_int]
    L.tkzo_set_case_equivalence.restype = None
    L.tkzo_set_case_equivalence(1 if on else 0)


def split_utf8(pattern: int, text: bytes):
    """Regex.Matches: list of (byte_start, byte_len)."""
    cap = len(text) + 1
    st = np.empty(cap, dtype=np.int64)
    ln = np.empty(cap, dtype=np.int64)
    k = lib().tkzo_split_utf8(pattern, _buf(text), len(text), st.ctypes.data, ln.ctypes.data, cap)
    if k < 0:
        raise OracleError(int(k))
    return list(zip(st[:k].tolist(), ln[:k].tolist()))


def split_utf16(pattern: int, units):
    u = np.asarray(units, dtype=np.uint16)
    cap = len(u) + 1
    st = np.empty(cap, dtype=np.int64)
    ln = np.empty(cap, dtype=np.int64)
    uu = np.ascontiguousarray(np.concatenate([u, np.zeros(1, np.uint16)]))
    k = lib().tkzo_split_utf16(pattern, uu.ctypes.data, len(u), st.ctypes.data, ln.ctypes.data, cap)
    if k < 0:
        raise OracleError(int(k))
    return list(zip(st[:k].tolist(), ln[:k].tolist()))


class Encoder:
    """TikTokenizer restated: plain Encode, Encode with allowed specials, UTF-16 entry."""

    def __init__(self, vocab: Vocab, pattern: int, cache_size: int = 8192, specials=None):
        self.vocab = vocab
        self._h = lib().tkzo_encoder_create(vocab._h, pattern, cache_size)
        if not self._h:
            raise OracleError(E_ARG)
        self.specials = []
        for lit, tid in (specials or {}).items():
            b = lit.encode("utf-8")
            lib().tkzo_encoder_add_special(self._h, _buf(b), len(b), tid)
            self.specials.append(lit)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().tkzo_encoder_free(self._h)
            self._h = None

    def encode_bytes(self, text: bytes):
        out = np.empty(max(1, len(text)), dtype=np.int32)
        k = lib().tkzo_encode_utf8(self._h, _buf(text), len(text), out.ctypes.data, len(out))
        if k < 0:
            raise OracleError(int(k))
        return out[:k].tolist()

    def encode(self, text: str, allowed_special=None):
        b = text.encode("utf-8")
        if not allowed_special:
            return self.encode_bytes(b)
        idx = np.asarray([self.specials.index(s) for s in allowed_special if s in self.specials], dtype=np.int32)
        out = np.empty(max(1, len(b)), dtype=np.int32)
        k = lib().tkzo_encode_special_utf8(self._h, _buf(b), len(b), idx.ctypes.data, len(idx),
                                           out.ctypes.data, len(out))
        if k < 0:
            raise OracleError(int(k))
        return out[:k].tolist()

    def encode_utf16(self, units):
        u = np.ascontiguousarray(np.concatenate([np.asarray(units, dtype=np.uint16), np.zeros(1, np.uint16)]))
        n = len(u) - 1
        out = np.empty(max(1, 3 * n), dtype=np.int32)
        k = lib().tkzo_encode_utf16(self._h, u.ctypes.data, n, out.ctypes.data, len(out))
        if k < 0:
            raise OracleError(int(k))
        return out[:k].tolist()


def encode_batch(vocab: Vocab, pattern: int, data: np.ndarray, offsets: np.ndarray, threads=1, cache_size=8192, timing=None):
    """CPU-baseline batch: returns (ids_flat, counts). data uint8[total], offsets int64[n+1].
    timing (a dict), if given, receives "seconds": the wall time of the C call alone (the threads' encode work, without
    the numpy packing of the result below)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    # zeros, not empty: the pages are touched here, outside the timed call (256 threads first-touching 4 bytes per input byte
    # serialise on the kernel's mm lock and the figure would measure page faults)
    out = np.zeros(max(1, len(data)), dtype=np.int32)
    counts = np.zeros(max(1, n), dtype=np.int32)
    import time
    t0 = time.perf_counter()
    tot = lib().tkzo_encode_batch(vocab._h, pattern, cache_size, data.ctypes.data, offsets.ctypes.data, n,
                                  out.ctypes.data, counts.ctypes.data, threads)
    if timing is not None:
        timing["seconds"] = time.perf_counter() - t0
    if tot < 0:
        raise OracleError(int(tot))
    counts = counts[:n]
    if n == 0:
        return np.empty(0, np.int32), counts
    idx = np.concatenate([out[offsets[d]:offsets[d] + counts[d]] for d in range(n)]) if n < 100000 else _gather(out, offsets, counts)
    return idx, counts


def host_parallelism(thread_counts):
    """{threads: speed-up over one thread} of a register-only loop run on that many threads at once: what the host really grants."""
    probe = lambda n: min(lib().tkzo_parallel_probe(int(n)) for _ in range(2))     # (the better of two runs)
    probe(1)
    t1 = probe(1)
    return {int(n): round(n * t1 / max(1, probe(n)), 1) for n in thread_counts}


def check_batch(vocab: Vocab, pattern: int, data: np.ndarray, offsets: np.ndarray, want_ids: np.ndarray, want_offsets: np.ndarray,
                threads=1, cache_size=8192, timing=None):
    """The checker at full size: every document is encoded on `threads` threads and compared IN PLACE with
    want_ids[want_offsets[d]:want_offsets[d+1]] -- no result arrays, so 10 M documents need nothing beyond the inputs.
    Returns (documents that differ, first such document or -1, tokens encoded); timing["seconds"] = wall time of the C call."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    want_ids = np.ascontiguousarray(want_ids, dtype=np.int32)
    want_offsets = np.ascontiguousarray(want_offsets, dtype=np.int64)
    assert len(want_offsets) == len(offsets)
    n = len(offsets) - 1
    first, tokens = C.c_int64(-1), C.c_int64(0)
    import time
    t0 = time.perf_counter()
    bad = lib().tkzo_check_batch(vocab._h, pattern, cache_size, data.ctypes.data, offsets.ctypes.data, n,
                                 want_ids.ctypes.data if len(want_ids) else None, want_offsets.ctypes.data, threads, C.byref(first), C.byref(tokens))
    if timing is not None:
        timing["seconds"] = time.perf_counter() - t0
    if bad < 0:
        raise OracleError(int(bad))
    return int(bad), int(first.value), int(tokens.value)


def _gather(out, offsets, counts):
    starts = offsets[:-1]
    total = int(counts.sum())
    pos = np.repeat(starts - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts) + np.arange(total)
    return out[pos]


# ---- EncodeTrimSuffix / EncodeTrimPrefix restated -------------------------------------------------
# Pure Python over the C primitives above (split on UTF-16 units, whole-piece rank, bpe); small inputs only.
# Follows Tokenizer_C#/TokenizerLib/TikTokenizer.cs function by function; all indices are UTF-16 code units.
class TrimOracle:
    def __init__(self, vocab: Vocab, pattern: int, specials=None):
        import re
        self.vocab, self.pattern = vocab, pattern
        self.specials = dict(specials or {})
        # new Regex(string.Join("|", specials.Keys.Select(Regex.Escape)))          TikTokenizer.cs:78
        self._re = re.compile("|".join(re.escape(k) for k in self.specials)) if self.specials else None

    @staticmethod
    def _units(text: str):
        return np.frombuffer(text.encode("utf-16-le", "surrogatepass"), dtype=np.uint16)

    @staticmethod
    def _str(units) -> str:
        return np.asarray(units, dtype=np.uint16).tobytes().decode("utf-16-le", "surrogatepass")

    def _piece_tokens(self, units):
        # Encoding.UTF8.GetBytes(piece) (:261 -- a lone surrogate becomes U+FFFD), Encoder.TryGetValue (:262), BytePairEncode (:268)
        b = self._str(units).encode("utf-16-le", "surrogatepass").decode("utf-16-le", "replace").encode("utf-8")
        r = self.vocab.rank(b)
        return [r] if r >= 0 else self.vocab.bpe(b)

    def _find_next_special(self, text: str, allowed, start: int):
        """FindNextSpecialToken (:230-241) on a Python str whose indices we keep in UTF-16 units through `u2s`."""
        # special literals are ASCII, so search on the str and convert the index
        find = start
        while True:
            m = self._re.search(self._s, self._u2s(find)) if self._re else None
            if m is None:
                return None, len(self._u)
            if m.group(0) in allowed:
                return m, self._s2u(m.start())
            find = self._s2u(m.start()) + 1                    # startFind = nextSpecial.Index + 1 (:238)

    def _bind(self, text: str):
        self._s = text
        self._u = self._units(text)
        # maps between str indices (code points) and UTF-16 unit indices
        s2u = [0]
        for ch in text:
            s2u.append(s2u[-1] + (2 if ord(ch) >= 0x10000 else 1))
        self._s2u_tab = s2u
        u2s = {}
        for si, ui in enumerate(s2u):
            u2s[ui] = si
        self._u2s_tab = u2s

    def _s2u(self, si):
        return self._s2u_tab[si]

    def _u2s(self, ui):
        while ui not in self._u2s_tab:                         # inside a surrogate pair: the search cannot match there anyway
            ui += 1
        return self._u2s_tab[ui]

    # private (int, int) EncodeTrimSuffix(text, tokenIds, start, end, maxTokenCount, tokenCount, encodeLength)   (:288-341)
    def _trim_suffix_segment(self, token_ids, start, end, max_tokens, token_count, encode_length):
        seg = self._u[start:end]
        for st, ln in split_utf16(self.pattern, seg):
            toks = self._piece_tokens(seg[st:st + ln])
            token_count += len(toks)
            if token_count <= max_tokens:
                encode_length += ln
                token_ids.extend(toks)
            else:
                break
            if token_count >= max_tokens:
                break
        return token_count, encode_length

    # EncodeTrimSuffixInternal (:343-392) and the public overloads (:394-429)
    def encode_trim_suffix(self, text: str, allowed, max_tokens: int):
        self._bind(text)
        n = len(self._u)
        token_ids = []
        if not allowed or self._re is None:
            _, enc_len = self._trim_suffix_segment(token_ids, 0, n, max_tokens, 0, 0)
            return token_ids, (text if enc_len == n else self._str(self._u[:enc_len]))
        allowed = set(allowed)
        start = token_count = enc_len = 0
        while True:
            m, end = self._find_next_special(text, allowed, start)
            if end > start:
                token_count, enc_len = self._trim_suffix_segment(token_ids, start, end, max_tokens, token_count, enc_len)
                if token_count >= max_tokens:
                    break
            if m is not None:
                token_count += 1
                if token_count <= max_tokens:
                    token_ids.append(self.specials[m.group(0)])             # EncodeSpecialToken (:215-220)
                    start = end + len(m.group(0))
                    enc_len += len(m.group(0))
                    if start >= n:
                        break
                if token_count >= max_tokens:
                    break
            else:
                break
        return token_ids, (text if enc_len == n else self._str(self._u[:enc_len]))

    # private void Encode(text, tokenIds, start, ref tokenCount, ref encodeLength, tokenCountMap, end)   (:485-519)
    def _encode_segment_mapped(self, token_ids, start, end, state, count_map):
        seg = self._u[start:end]
        for st, ln in split_utf16(self.pattern, seg):
            toks = self._piece_tokens(seg[st:st + ln])
            state[0] += len(toks)
            state[1] += ln
            token_ids.extend(toks)
            count_map[state[0]] = state[1]

    # EncodeTrimPrefixInternal (:431-468), TrimPrefix (:470-483) and the public overloads (:529-564)
    def encode_trim_prefix(self, text: str, allowed, max_tokens: int):
        self._bind(text)
        n = len(self._u)
        token_ids = []
        state = [0, 0]
        count_map = {0: 0}
        if not allowed or self._re is None:
            self._encode_segment_mapped(token_ids, 0, n, state, count_map)
        else:
            allowed = set(allowed)
            start = 0
            while True:
                m, end = self._find_next_special(text, allowed, start)
                if end > start:
                    self._encode_segment_mapped(token_ids, start, end, state, count_map)
                if m is not None:
                    token_ids.append(self.specials[m.group(0)])
                    start = end + len(m.group(0))
                    state[0] += 1
                    state[1] += len(m.group(0))
                    count_map[state[0]] = state[1]
                    if start >= n:
                        break
                else:
                    break
        token_count = state[0]
        if token_count <= max_tokens:
            return token_ids, text
        prefix = token_count - max_tokens
        cut_tokens = cut_len = 0
        for key in sorted(count_map):                          # SortedDictionary enumeration
            if key >= prefix:
                cut_tokens, cut_len = key, count_map[key]
                break
        return token_ids[cut_tokens:], self._str(self._u[cut_len:])
