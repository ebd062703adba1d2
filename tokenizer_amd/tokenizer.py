"""Host-side mirror of the reference's tokenizer interface for the Encode path, over libtkz.

Names, argument meaning and error behaviour follow
  ITokenizer                     Tokenizer_C#/TokenizerLib/ITokenizer.cs:7-46
  TikTokenizer                   Tokenizer_C#/TokenizerLib/TikTokenizer.cs:20-605
  TokenizerBuilder               Tokenizer_C#/TokenizerLib/TokenizerBuilder.cs:14-214
so that the parity tests read like the reference's own (TikTokenizerUnitTest.cs).  `EncodeBatch` is the
one addition (the reference has no batch API).  What runs where:

  * special-token segmentation (EncodeInternal / FindNextSpecialToken, TikTokenizer.cs:141-170,230-241)
    is host work: every text is cut into plain segments + literal special ids, and ALL plain segments of
    the batch go to the GPU as one document batch (a segment is matched in isolation by the reference
    too: `Regex.Matches(text[start..end])`, TikTokenizer.cs:252);
  * the plain path (TikTokenizer.cs:250-274 + BytePairEncoder.cs:13-76) is the HIP path of libtkz;
  * Decode / DecodeBatch (TikTokenizer.cs:586-604) run on the device too: an id -> bytes gather through the decoder table + a scan;
  * EncodeTrimSuffix / EncodeTrimPrefix (TikTokenizer.cs:288-579) cut at piece granularity: the GPU
    encodes every plain segment with piece granularity (tkz_encode_batch_pieces_utf8: token count and byte span of each
    regex match), the walk over pieces / special tokens that decides where to cut is host work.
"""
import os
import re
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np

from . import _native as N

ENDOFTEXT = "<|endoftext|>"
FIM_PREFIX = "<|fim_prefix|>"
FIM_MIDDLE = "<|fim_middle|>"
FIM_SUFFIX = "<|fim_suffix|>"
ENDOFPROMPT = "<|endofprompt|>"

# the split regexes, verbatim (TokenizerBuilder.cs:112,128; tokenizer_ts/src/tokenizerBuilder.ts:79-89)
REGEX_PATTERN_1 = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
REGEX_CL100K = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*"
                r"|\s*[\r\n]+|\s+(?!\S)|\s+")
_O2_SUFFIX = r"(?:'s|'S|'t|'T|'re|'RE|'Re|'eR|'ve|'VE|'vE|'Ve|'m|'M|'ll|'lL|'Ll|'LL|'d|'D)?"
REGEX_O200K = "|".join([
    "[^\r\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+" + _O2_SUFFIX,
    "[^\r\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]+[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]*" + _O2_SUFFIX,
    r"\p{N}{1,3}", r" ?[^\s\p{L}\p{N}]+[\r\n/]*", r"\s*[\r\n]+", r"\s+(?!\S)", r"\s+"])

# TokenizerBuilder.cs:17-66 (and the o200k models of tokenizer_ts/src/tokenizerBuilder.ts)
MODEL_PREFIX_TO_ENCODING = {"gpt-4-": "cl100k_base", "gpt-3.5-turbo-": "cl100k_base", "gpt-4o-": "o200k_base"}
MODEL_TO_ENCODING = {
    "gpt-4o": "o200k_base",
    "gpt-4": "cl100k_base", "gpt-3.5-turbo": "cl100k_base",
    "text-davinci-003": "p50k_base", "text-davinci-002": "p50k_base", "text-davinci-001": "r50k_base",
    "text-curie-001": "r50k_base", "text-babbage-001": "r50k_base", "text-ada-001": "r50k_base",
    "davinci": "r50k_base", "curie": "r50k_base", "babbage": "r50k_base", "ada": "r50k_base",
    "code-davinci-002": "p50k_base", "code-davinci-001": "p50k_base", "code-cushman-002": "p50k_base",
    "code-cushman-001": "p50k_base", "davinci-codex": "p50k_base", "cushman-codex": "p50k_base",
    "text-davinci-edit-001": "p50k_edit", "code-davinci-edit-001": "p50k_edit",
    "text-embedding-ada-002": "cl100k_base",
    "text-similarity-davinci-001": "r50k_base", "text-similarity-curie-001": "r50k_base",
    "text-similarity-babbage-001": "r50k_base", "text-similarity-ada-001": "r50k_base",
    "text-search-davinci-doc-001": "r50k_base", "text-search-curie-doc-001": "r50k_base",
    "text-search-babbage-doc-001": "r50k_base", "text-search-ada-doc-001": "r50k_base",
    "code-search-babbage-code-001": "r50k_base", "code-search-ada-code-001": "r50k_base",
    "gpt2": "gpt2",
}
# encoder -> (regex, rank file name, special tokens)          TokenizerBuilder.cs:109-181
ENCODERS = {
    "cl100k_base": (REGEX_CL100K, "cl100k_base.tiktoken",
                    {ENDOFTEXT: 100257, FIM_PREFIX: 100258, FIM_MIDDLE: 100259, FIM_SUFFIX: 100260, ENDOFPROMPT: 100276}),
    "p50k_base": (REGEX_PATTERN_1, "p50k_base.tiktoken", {ENDOFTEXT: 50256}),
    "p50k_edit": (REGEX_PATTERN_1, "p50k_base.tiktoken", {ENDOFTEXT: 50256, FIM_PREFIX: 50281, FIM_MIDDLE: 50282, FIM_SUFFIX: 50283}),
    "r50k_base": (REGEX_PATTERN_1, "r50k_base.tiktoken", {ENDOFTEXT: 50256}),
    "gpt2": (REGEX_PATTERN_1, "gpt2.tiktoken", {ENDOFTEXT: 50256}),
    "o200k_base": (REGEX_O200K, "o200k_base.tiktoken", {ENDOFTEXT: 199999, ENDOFPROMPT: 200018}),
}
# the engine of the reference that defines each encoder: o200k_base ships only with the TypeScript reference (tokenizer_ts/src/tokenizerBuilder.ts:79-89,
# compiled by `new RegExp(pattern, "gu")`, tikTokenizer.ts:100); everything else is TokenizerBuilder.cs's.  By-name construction follows this table;
# CreateTokenizer with an explicit pattern string keeps the .NET reading (it replaces `new Regex(pattern)`, TikTokenizer.cs:77).
ENCODER_ENGINE = {"o200k_base": "js"}


def host_unicode_classes(ucd=None, n_code_points: int = 0x110000, engine: str = "dotnet"):
    """The class table a host hands to TikTokenizer(unicode_classes=...): class 0..8 of every code point below n_code_points (65536: code units) from
    a `unicodedata`-like module (default: this Python's -- another Unicode version than the 13.0 libtkz is built with).  \\s is .NET's
    ([\\f\\n\\r\\t\\v\\x85\\p{Z}]) or, engine="js", ECMAScript's ([\\t\\n\\v\\f\\r\\ufeff\\p{Zs}\\u2028\\u2029])."""
    import unicodedata
    ucd = ucd or unicodedata
    out = np.zeros(n_code_points, np.uint8)
    code = {"Lu": 1, "Ll": 2, "Lt": 3, "Lm": 4, "Lo": 5, "Mn": 6, "Mc": 6, "Me": 6, "Nd": 7, "Nl": 7, "No": 7, "Zs": 8, "Zl": 8, "Zp": 8}
    for cp in range(n_code_points):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        out[cp] = code.get(ucd.category(chr(cp)), 0)
    # \s beyond \p{Z}: .NET adds [\f\n\r\t\v\x85]; ECMAScript adds [\t\n\v\f\r\ufeff] (U+0085 is NOT white space there)
    for cp in ((9, 10, 11, 12, 13, 0xFEFF) if engine == "js" else (9, 10, 11, 12, 13, 0x85)):
        if cp < n_code_points:
            out[cp] = 8
    return out


def _utf8_like_dotnet(s: str) -> bytes:
    """Encoding.UTF8.GetBytes: a lone surrogate becomes U+FFFD (TikTokenizer.cs:261)."""
    try:
        return s.encode("utf-8")
    except UnicodeEncodeError:
        return s.encode("utf-16-le", "surrogatepass").decode("utf-16-le", "replace").encode("utf-8")


class TikTokenizer:
    """ITokenizer over the MI355X encode path.  Construct through TokenizerBuilder, or directly with the
    bytes of a .tiktoken rank file (the reference takes a Stream, TikTokenizer.cs:60-65)."""

    def __init__(self, tikTokenBpeFile: bytes, specialTokensEncoder: Optional[Dict[str, int]], pattern: str,
                 cacheSize: int = 8192, device: int = 0, lib: Optional[N.Library] = None, host: str = "dotnet",
                 unicode_classes=None, case_equivalence: bool = False, reserve_bytes: int = 0, reserve_docs: int = 0):
        """`host` names the regex engine whose reading of `pattern` is wanted: "dotnet" (the default: `new Regex(pattern,
        RegexOptions.Compiled)`, TikTokenizer.cs:77 -- UTF-16 code units, .NET's \\s) or "js" (`new RegExp(pattern, "gu")`,
        tokenizer_ts/src/tikTokenizer.ts:100 -- code points, ECMAScript's \\s; implemented for the o200k string only, the one
        pattern that exists only in the TypeScript reference).  The two differ on supplementary-plane chars, U+0085 and U+FEFF."""
        self._lib = lib or N.default_library()
        if host not in ("dotnet", "js"):
            raise ValueError("host must be 'dotnet' or 'js'")
        pat = self._lib.L.tkz_pattern_from_regex_engine  # maps the reference's regex text to a scanner, refuses anything else
        import ctypes as C
        out = C.c_int32(0)
        self._lib.check(pat(pattern.encode("utf-8"), N.ENGINE_ECMASCRIPT if host == "js" else N.ENGINE_DOTNET, C.byref(out)))
        self._vocab = N.Vocab(tikTokenBpeFile, self._lib)     # FormatError / DuplicateRankError as in LoadTikTokenBpe + Init
        self._encoder = N.Encoder(self._vocab, out.value, device)
        # The split is whatever the HOST's regex engine makes of the pattern (TikTokenizer.cs:77 compiles it in the running process): `unicode_classes`
        # hands that runtime's classification over (uint8[65536] per code unit, or uint8[1114112] per code point; host_unicode_classes() builds
        # one from a `unicodedata`-like module), `case_equivalence` is .NET >= 7's reading of cl100k's (?i:...) ('ſ is 's).  Default: net6.0's.
        if unicode_classes is not None:
            if host == "js" and len(unicode_classes) == 65536:
                # the ECMAScript reading classifies by CODE POINT: a table of code units leaves the supplementary planes on the built-in Unicode 13.0 data
                import warnings
                warnings.warn("host='js' classifies by code point: a 65536-entry unicode_classes table covers the BMP only, the supplementary planes keep "
                              "libtkz's built-in Unicode 13.0 classes (pass 1114112 entries to override them too)")
            self._encoder.set_unicode_classes(unicode_classes)
        if case_equivalence:
            self._encoder.set_option(N.OPT_CASE_EQUIVALENCE, 1)
        # construction pays, not the first Encode (TokenizerBuilder.cs:210-213): the device workspace of batches of up to reserve_bytes / reserve_docs
        if reserve_bytes > 0:
            self._encoder.reserve(reserve_bytes, max(1, reserve_docs))
        # the reference's LRU piece memo (LRUCache.cs; no effect on results) lives on the device with a fixed size: cacheSize only says
        # whether it is used (the reference's LRUCache of size 0 keeps nothing)
        if cacheSize <= 0:
            self._encoder.set_option(N.OPT_PIECE_MEMO, 0)
        self.SpecialTokensEncoder: Dict[str, int] = dict(specialTokensEncoder or {})
        self.SpecialTokens = set(self.SpecialTokensEncoder)
        # alternation of the escaped literals in registration order (TikTokenizer.cs:78): leftmost match, first alternative wins
        self._special_re = re.compile("|".join(re.escape(k) for k in self.SpecialTokensEncoder)) if self.SpecialTokensEncoder else None
        if self.SpecialTokensEncoder:
            self._encoder.set_special_tokens(self.SpecialTokensEncoder)      # SpecialTokensDecoder (TikTokenizer.cs:79), for Decode

    # ---- segmentation (host) ---------------------------------------------------------------------
    def _segments(self, text: str, allowed: Optional[Iterable[str]]):
        """EncodeInternal + FindNextSpecialToken: list of ('t', plain_text, None) / ('s', id, literal)."""
        allowed = set(allowed) if allowed else set()
        if not allowed or self._special_re is None:
            return [("t", text, None)] if text else []
        out = []
        start = 0
        while True:
            find = start
            m = None
            while True:                                       # FindNextSpecialToken (TikTokenizer.cs:230-241)
                m = self._special_re.search(text, find)
                if m is None or m.group(0) in allowed:
                    break
                find = m.start() + 1
            end = m.start() if m else len(text)
            if end > start:
                out.append(("t", text[start:end], None))
            if m is None:
                break
            out.append(("s", self.SpecialTokensEncoder[m.group(0)], m.group(0)))   # EncodeSpecialToken (:215-220)
            start = m.end()
            if start >= len(text):
                break
        return out

    def _resolve_allowed(self, allowedSpecialOrApply):
        if isinstance(allowedSpecialOrApply, bool):           # Encode(string, bool applySpecialTokens = true)  (:193-207)
            return self.SpecialTokens if (allowedSpecialOrApply and self.SpecialTokens) else None
        return allowedSpecialOrApply                          # Encode(string, IReadOnlyCollection<string>)     (:178-185)

    # ---- ITokenizer --------------------------------------------------------------------------------
    def Encode(self, text: str, allowedSpecialOrApply: Union[bool, Sequence[str], None] = True) -> List[int]:
        return self.EncodeBatch([text], allowedSpecialOrApply)[0]

    def EncodeBatch(self, texts: Sequence[str], allowedSpecialOrApply: Union[bool, Sequence[str], None] = True) -> List[List[int]]:
        ids, offs = self.EncodeBatchFlat(texts, allowedSpecialOrApply)
        flat = ids.tolist()
        return [flat[offs[d]:offs[d + 1]] for d in range(len(texts))]

    def EncodeBatchFlat(self, texts: Sequence[str], allowedSpecialOrApply: Union[bool, Sequence[str], None] = True):
        """EncodeBatch without a list per text: (ids int32[total], offsets int64[len(texts) + 1]); text d is ids[offsets[d]:offsets[d+1]].
        When no special token applies (the reference's plain path, TikTokenizer.cs:180-183,196-199) the arrays are the device call's own
        output, untouched; otherwise the special ids are spliced in between the plain segments' ids with array copies."""
        allowed = self._resolve_allowed(allowedSpecialOrApply)
        if not allowed or self._special_re is None:
            segs = [_utf8_like_dotnet(t) for t in texts]
            if not segs:
                return np.zeros(0, np.int32), np.zeros(1, np.int64)
            lens = np.fromiter(map(len, segs), np.int64, len(segs))
            offs = np.zeros(len(segs) + 1, np.int64)
            np.cumsum(lens, out=offs[1:])
            data = np.frombuffer(b"".join(segs), np.uint8) if offs[-1] else np.zeros(0, np.uint8)
            return self._encoder.encode_batch(data, offs)
        plans = [self._segments(t, allowed) for t in texts]
        segs = [_utf8_like_dotnet(s) for plan in plans for kind, s, _ in plan if kind == "t"]
        if segs:
            data = np.frombuffer(b"".join(segs), np.uint8) if sum(map(len, segs)) else np.zeros(0, np.uint8)
            offs = np.cumsum([0] + [len(s) for s in segs]).astype(np.int64)
            ids, ooff = self._encoder.encode_batch(data, offs)
        else:
            ids, ooff = np.zeros(0, np.int32), np.zeros(1, np.int64)
        n_special = sum(1 for plan in plans for kind, _, _ in plan if kind == "s")
        out = np.empty(len(ids) + n_special, np.int32)
        out_offs = np.zeros(len(texts) + 1, np.int64)
        w, k = 0, 0
        for d, plan in enumerate(plans):
            for kind, v, _ in plan:
                if kind == "s":
                    out[w] = v
                    w += 1
                else:
                    n = int(ooff[k + 1] - ooff[k])
                    out[w:w + n] = ids[ooff[k]:ooff[k + 1]]
                    w += n
                    k += 1
            out_offs[d + 1] = w
        return out, out_offs

    # ---- trim variants (TikTokenizer.cs:288-579) --------------------------------------------------
    def _piece_items(self, text: str, allowed):
        """The walk both trim variants share: the text as a list of (n_tokens, ids, utf16_length) items in order --
        one item per regex piece of every plain segment (Regex.Matches(text[start..end]), :290,:485) and one per special
        token (:215-220).  All plain segments go to the GPU in one piece-granular call."""
        plan = self._segments(text, allowed)
        segs = [_utf8_like_dotnet(s) for kind, s, _ in plan if kind == "t"]
        items = []
        if segs:
            data = np.frombuffer(b"".join(segs), np.uint8)
            offs = np.cumsum([0] + [len(s) for s in segs]).astype(np.int64)
            ids, dpo, pbo, pto = self._encoder.encode_batch_pieces(data, offs)
            # UTF-16 length of a piece = chars + 4-byte chars (piece.Length, :295): count lead bytes, 0xF0.. twice
            lead = ((data & 0xC0) != 0x80).astype(np.int64) + (data >= 0xF0).astype(np.int64)
            cum = np.concatenate([[0], np.cumsum(lead)])
            u16 = cum[pbo[1:]] - cum[pbo[:-1]]
        k = 0
        for kind, v, lit in plan:
            if kind == "s":
                items.append((1, [v], len(lit.encode("utf-16-le")) // 2, True))
            else:
                for p in range(int(dpo[k]), int(dpo[k + 1])):
                    items.append((int(pto[p + 1] - pto[p]), ids[pto[p]:pto[p + 1]].tolist(), int(u16[p]), False))
                k += 1
        return items

    @staticmethod
    def _utf16_prefix(text: str, n_units: int, drop: bool = False) -> str:
        """text[..n] / text[n..] with n in UTF-16 code units, as the reference slices a .NET string."""
        raw = text.encode("utf-16-le", "surrogatepass")
        if n_units * 2 >= len(raw):
            return "" if drop else text
        part = raw[2 * n_units:] if drop else raw[:2 * n_units]
        return part.decode("utf-16-le", "surrogatepass")

    def _trim_args(self, a, b):
        # EncodeTrimX(string, IReadOnlyCollection<string> allowedSpecial, int maxTokenCount)       (:394-403, :529-536)
        # EncodeTrimX(string, int maxTokenCount, bool applySpecialTokens = true)                   (:412-429, :545-564)
        if isinstance(a, int) and not isinstance(a, bool):
            apply = True if b is None else bool(b)
            return (self.SpecialTokens if (apply and self.SpecialTokens) else None), int(a)
        if b is None:
            raise TypeError("maxTokenCount is required")
        return (a if a else None), int(b)

    def EncodeTrimSuffix(self, text: str, a, b=None):
        """Token ids and the text they cover, cut after the last piece / special token that still fits maxTokenCount."""
        allowed, max_tokens = self._trim_args(a, b)
        token_ids: List[int] = []
        token_count = 0
        encode_length = 0
        for n, ids, ulen, is_special in self._piece_items(text, allowed):
            token_count += n                                   # (:298,:313,:328; special: :364)
            if token_count <= max_tokens:
                token_ids.extend(ids)
                encode_length += ulen
            else:
                break                                          # the piece that overflows is dropped with everything after it
            if token_count >= max_tokens:
                break                                          # (:340, :356-359, :375-378)
        return token_ids, self._utf16_prefix(text, encode_length)

    def EncodeTrimPrefix(self, text: str, a, b=None):
        """Token ids and the text they cover, cut before the first piece boundary that leaves at most maxTokenCount tokens."""
        allowed, max_tokens = self._trim_args(a, b)
        token_ids: List[int] = []
        token_count = 0
        encode_length = 0
        boundaries = [(0, 0)]                                  # tokenCountMap: cumulative tokens -> cumulative UTF-16 length (:438-441)
        for n, ids, ulen, is_special in self._piece_items(text, allowed):
            token_count += n
            encode_length += ulen
            token_ids.extend(ids)
            boundaries.append((token_count, encode_length))
        if token_count <= max_tokens:                          # TrimPrefix (:463-483)
            return token_ids, text
        prefix_tokens = token_count - max_tokens
        cut_tokens, cut_len = 0, 0
        for tc, el in boundaries:
            if tc >= prefix_tokens:
                cut_tokens, cut_len = tc, el
                break
        return token_ids[cut_tokens:], self._utf16_prefix(text, cut_len, drop=True)

    def Decode(self, tokens: Sequence[int]) -> str:
        """TikTokenizer.cs:586-604: ids that are neither in the vocabulary nor special tokens are dropped; the bytes are decoded as
        UTF-8 (Encoding.UTF8.GetString: malformed sequences become U+FFFD)."""
        return self.DecodeBatch([tokens])[0]

    def DecodeBatch(self, batches: Sequence[Sequence[int]]) -> List[str]:
        """Decode for a batch, on the device (tkz_decode_batch: id -> bytes gather through the decoder table + scan)."""
        flat = np.asarray([t for b in batches for t in b], dtype=np.int64)
        flat = np.where((flat < -2**31) | (flat >= 2**31), -1, flat).astype(np.int32)      # (an id outside int is in no table)
        offs = np.cumsum([0] + [len(b) for b in batches]).astype(np.int64)
        data, boffs = self._encoder.decode_batch(flat, offs)
        raw = data.tobytes()
        return [raw[boffs[d]:boffs[d + 1]].decode("utf-8", "replace") for d in range(len(batches))]

    # the raw device encoder, for callers that hold documents in HBM (bench.py)
    @property
    def native(self) -> N.Encoder:
        return self._encoder


class TokenizerBuilder:
    """TokenizerBuilder.cs:14-214 without the HTTP download: rank files are read from a directory
    (`vocab_dir`, default $TKZ_VOCAB_DIR), because the build has no network."""

    @staticmethod
    def _encoder_for_model(modelName: str) -> str:
        enc = MODEL_TO_ENCODING.get(modelName)
        if enc is None:
            for prefix, e in MODEL_PREFIX_TO_ENCODING.items():
                if modelName.startswith(prefix):
                    enc = e
                    break
        if enc is None:
            raise NotImplementedError("Doesn't support this model [%s]" % modelName)       # TokenizerBuilder.cs:96
        return enc

    @staticmethod
    def CreateByModelName(modelName: str, extraSpecialTokens: Optional[Dict[str, int]] = None, vocab_dir: Optional[str] = None, device: int = 0,
                          host: Optional[str] = None, lib: Optional[N.Library] = None):
        return TokenizerBuilder.CreateByEncoderName(TokenizerBuilder._encoder_for_model(modelName), extraSpecialTokens, vocab_dir, device, host, lib)

    @staticmethod
    def CreateByEncoderName(encoderName: str, extraSpecialTokens: Optional[Dict[str, int]] = None, vocab_dir: Optional[str] = None, device: int = 0,
                            host: Optional[str] = None, lib: Optional[N.Library] = None):
        """`host=None` picks the engine of the reference that DEFINES the encoder (ENCODER_ENGINE): o200k_base / gpt-4o exist only in the
        TypeScript reference, whose `new RegExp(pattern, "gu")` matches by code point with ECMAScript's \\s ("js"); every other encoder is the
        C# reference's ("dotnet").  An explicit `host` overrides that (e.g. "dotnet" for a C# host that hands the o200k string to
        TokenizerBuilder.CreateTokenizer(stream, specials, pattern) itself)."""
        if encoderName not in ENCODERS:
            raise NotImplementedError("Doesn't support this encoder [%s]" % encoderName)  # TokenizerBuilder.cs:179
        if host is None:
            host = ENCODER_ENGINE.get(encoderName, "dotnet")
        regex, fname, specials = ENCODERS[encoderName]
        specials = dict(specials)
        if extraSpecialTokens:
            specials.update(extraSpecialTokens)
        d = vocab_dir or os.environ.get("TKZ_VOCAB_DIR") or "."
        path = os.path.join(d, fname)
        if not os.path.exists(path):
            raise FileNotFoundError("%s not found: the reference downloads it at run time (TokenizerBuilder.cs:113,195); "
                                    "place it in vocab_dir / $TKZ_VOCAB_DIR" % path)
        with open(path, "rb") as f:
            return TokenizerBuilder.CreateTokenizer(f.read(), specials, regex, device=device, lib=lib, host=host)

    @staticmethod
    def CreateTokenizer(tikTokenBpeFile: bytes, specialTokensEncoder: Optional[Dict[str, int]], pattern: str, cacheSize: int = 8192,
                        device: int = 0, lib: Optional[N.Library] = None, host: str = "dotnet", unicode_classes=None,
                        case_equivalence: bool = False) -> TikTokenizer:
        return TikTokenizer(tikTokenBpeFile, specialTokensEncoder, pattern, cacheSize, device, lib, host, unicode_classes, case_equivalence)
