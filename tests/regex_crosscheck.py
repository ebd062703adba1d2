"""Independent cross-check of the oracle's split against the third-party `regex` engine.

.NET matches one UTF-16 code unit per class test.  We emulate that with `regex` by feeding it a
str whose "characters" ARE the UTF-16 code units (surrogate halves become lone-surrogate code
points, category Cs), and by writing \\s as the explicit .NET set.  `regex` ships newer Unicode
tables than .NET 6 (13.0), so the random alphabet is restricted to characters whose L/N/M/Z
classification agrees between `unicodedata` 13.0 and `regex`; U+017F / U+212A are excluded because
`regex` folds them onto s / k under (?i) while net6.0 does not.
"""
import random
import unicodedata

import regex

WS = r"[\t-\r \x85\p{Z}]"
NWS = r"[^\t-\r \x85\p{Z}]"


P1_SRC = r"'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"
CL_SRC = r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"
_C = r"(?:'s|'S|'t|'T|'re|'RE|'Re|'eR|'ve|'VE|'vE|'Ve|'m|'M|'ll|'lL|'Ll|'LL|'d|'D)?"
O2_SRC = "|".join([
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+" + _C,
    r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*" + _C,
    r"\p{N}{1,3}",
    r" ?[^\s\p{L}\p{N}]+[\r\n/]*",
    r"\s*[\r\n]+",
    r"\s+(?!\S)",
    r"\s+",
])


# o200k is matched by CODE POINT with ECMAScript's \s (tokenizer_ts/src/tikTokenizer.ts:100: `new RegExp(pattern, "gu")`):
# WhiteSpace + LineTerminator = TAB VT FF SP NBSP ZWNBSP(U+FEFF) \p{Zs} LF CR LS PS.  U+0085 is not in it.
JS_WS_IN = r"\t-\r \xa0\ufeff\p{Zs}\u2028\u2029"


def _compile(src, ws_in=r"\t-\r \x85\p{Z}"):
    # write \s inside and outside classes as an explicit set (.NET's by default)
    WS, NWS = "[" + ws_in + "]", "[^" + ws_in + "]"
    out, i, depth = [], 0, 0
    while i < len(src):
        ch = src[i]
        if ch == "\\" and i + 1 < len(src):
            nx = src[i + 1]
            if nx == "s":
                out.append(ws_in if depth else WS)
                i += 2
                continue
            if nx == "S":
                assert not depth
                out.append(NWS)
                i += 2
                continue
            out.append(src[i:i + 2])
            i += 2
            continue
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        out.append(ch)
        i += 1
    return regex.compile("".join(out), regex.V0)


# pattern 4 = the o200k string as the C# reference would run it (TokenizerBuilder.cs:210-213 -> `new Regex(pattern, Compiled)`,
# TikTokenizer.cs:77): the engine is fed code UNITS and \s is .NET's, exactly as for patterns 1 / cl100k
PATTERNS = {1: _compile(P1_SRC), 2: _compile(CL_SRC), 3: _compile(O2_SRC, JS_WS_IN), 4: _compile(O2_SRC)}


def to_units(s: str):
    """str -> list of UTF-16 code units."""
    b = s.encode("utf-16-le", "surrogatepass")
    return [b[i] | (b[i + 1] << 8) for i in range(0, len(b), 2)]


def split_units_regex(pattern_id: int, units):
    """(start, length) in UTF-16 units of every match.  Patterns 1 / cl100k / 4 (o200k through .NET): the engine is fed the code units
    themselves; 3 (o200k through ECMAScript): code points -- a well-formed surrogate pair is one character, a lone surrogate stays a lone
    (Cs) code point."""
    if pattern_id != 3:
        s = "".join(map(chr, units))
        return [(m.start(), m.end() - m.start()) for m in PATTERNS[pattern_id].finditer(s)]
    chars, at = [], []
    i = 0
    while i < len(units):
        u = units[i]
        at.append(i)
        if 0xD800 <= u <= 0xDBFF and i + 1 < len(units) and 0xDC00 <= units[i + 1] <= 0xDFFF:
            chars.append(chr(0x10000 + ((u - 0xD800) << 10) + (units[i + 1] - 0xDC00)))
            i += 2
        else:
            chars.append(chr(u))
            i += 1
    at.append(len(units))
    return [(at[m.start()], at[m.end()] - at[m.start()]) for m in PATTERNS[3].finditer("".join(chars))]


def _agree(cp):
    ch = chr(cp)
    cat = unicodedata.category(ch)
    if cat in ("Cn", "Co"):
        return False
    maj = cat[0]
    for cls, pat in (("L", r"\p{L}"), ("N", r"\p{N}"), ("M", r"\p{M}"), ("Z", r"\p{Z}")):
        if bool(regex.fullmatch(pat, ch)) != (maj == cls):
            return False
    if maj == "L":
        for sub in ("Lu", "Ll", "Lt", "Lm", "Lo"):
            if bool(regex.fullmatch(r"\p{%s}" % sub, ch)) != (cat == sub):
                return False
    return True


def alphabet():
    base = list("abcdeflmrstvDELMRSTVxyzXYZ") * 3 + list("0123456789") + list("  \t\n\r\x0b\x0c") + \
        list("''''.,;:!?()[]{}<>=+-*/\\_#@&|\"`~^%$") + ["\x85", "\xa0", "　", " ", " ", "\x1c", "\x1f"]
    extra = [0xE9, 0xC9, 0x4E2D, 0x6587, 0x3042, 0x30AB, 0xAC00, 0x0416, 0x0436, 0x05D0, 0x0627, 0x0660, 0x0969,
             0x2B50, 0x2764, 0xFE0F, 0x200D, 0x0301, 0x0300, 0x093F, 0x01C5, 0x02B0, 0x00AA, 0x00B2, 0x2160, 0x3007,
             0x00DF, 0x0130, 0x0131, 0xFF21, 0xFF41, 0xFF10, 0x1F600, 0x1F468, 0x1F3FD, 0x20000, 0x1D7D8, 0x10400,
             0xFFFD, 0x00B7, 0x2019, 0x201C, 0x00AD, 0x0E01, 0x0E31]
    # supplementary-plane letters / digits / marks (o200k sees their classes; .NET sees two OTHER units) and ECMAScript's \s edge cases
    extra += [0x1D400, 0x1D7CE, 0x1D7CF, 0x10428, 0x1E900, 0x1E922, 0x16F93, 0x2A700, 0x1F1E6, 0x101FD, 0x1D165, 0x10107, 0xFEFF, 0x2028, 0x1680]
    for cp in extra:
        if _agree(cp):
            base.append(chr(cp))
    return base


def random_text(rng: random.Random, alpha, n):
    # bursty: runs of the same "kind" are common, which is what exercises the run logic
    out = []
    while len(out) < n:
        ch = rng.choice(alpha)
        rep = 1 if rng.random() < 0.6 else rng.randint(1, 6)
        if rng.random() < 0.15:
            out.extend(rng.choice(alpha) for _ in range(rep))
        else:
            out.extend([ch] * rep if rng.random() < 0.3 else [rng.choice(alpha) for _ in range(rep)])
    return "".join(out[:n])
