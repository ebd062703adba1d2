// tkz_classes.h -- character classes of the split regexes, evaluated on UTF-8 bytes.
//
// Pattern 1 and cl100k: the reference matches with System.Text.RegularExpressions over UTF-16 code units
// (Tokenizer_C#/TokenizerLib/TikTokenizer.cs:77,252): every class test looks at ONE unit, a surrogate
// half is category Cs.  A code point >= 0x10000 is therefore two "other" units; on UTF-8 data that
// is a 4-byte sequence whose char is class OTHER and which can never be the one-unit optional
// prefix `[^\r\n\p{L}\p{N}]?` of a word (the \p{L} that must follow would have to match the low
// surrogate).  BMP units are classified by a 64 Ki table generated from Unicode 13.0 (net6.0's
// data, see tools/gen_unicode_tables.py).
#pragma once
#include <stdint.h>

#include "tkz_simt.h"

// Unicode classes stored in TkzTables::bmp_class
enum : uint8_t { UC_OTHER = 0, UC_LU = 1, UC_LL = 2, UC_LT = 3, UC_LM = 4, UC_LO = 5, UC_M = 6, UC_N = 7, UC_WS = 8 };

// What one decoded char looks like to the scanners.
struct TkzChar {
    uint32_t cp;      // code point (0xFFFFFFFF when the sequence is malformed)
    uint8_t len;      // bytes (1..4); 1 for a malformed byte
    uint8_t uc;       // UC_* (supplementary plane => UC_OTHER)
    uint8_t units;    // UTF-16 units: 1 or 2
    uint8_t bad;      // malformed UTF-8
};

TKZ_HD uint8_t tkz_ascii_class(uint32_t c) {
    if (c - 'a' < 26u) return UC_LL;
    if (c - 'A' < 26u) return UC_LU;
    if (c - '0' < 10u) return UC_N;
    if (c == ' ' || c - 9u < 5u) return UC_WS;   // \t \n \v \f \r and space (U+001C..1F are not \s in .NET)
    return UC_OTHER;
}

// Decode the char whose LEAD byte is b0; b1..b3 are the following bytes (0 when past the end of the
// document: 0 is never a continuation byte, so truncation shows up as malformed).  tkz_decode_raw leaves the class of a
// non-ASCII char to the caller (uc = UC_OTHER), so that several table lookups can be in flight together.
TKZ_HD TkzChar tkz_decode_raw(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    TkzChar c;
    c.bad = 0; c.units = 1; c.uc = UC_OTHER;
    if (b0 < 0x80) { c.cp = b0; c.len = 1; c.uc = tkz_ascii_class(b0); return c; }
    const bool c1 = (b1 & 0xC0) == 0x80, c2 = (b2 & 0xC0) == 0x80, c3 = (b3 & 0xC0) == 0x80;
    if (b0 >= 0xC2 && b0 <= 0xDF && c1) {
        c.cp = ((b0 & 0x1F) << 6) | (b1 & 0x3F); c.len = 2;
    } else if (b0 >= 0xE0 && b0 <= 0xEF && c1 && c2) {
        c.cp = ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F); c.len = 3;
        if (c.cp < 0x800 || (c.cp >= 0xD800 && c.cp <= 0xDFFF)) c.bad = 1;
    } else if (b0 >= 0xF0 && b0 <= 0xF4 && c1 && c2 && c3) {
        c.cp = ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F); c.len = 4;
        if (c.cp < 0x10000 || c.cp > 0x10FFFF) c.bad = 1;
        c.units = 2;
    } else {
        c.cp = 0xFFFFFFFFu; c.len = 1; c.bad = 1;
    }
    return c;
}
TKZ_HD TkzChar tkz_decode(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, const uint8_t* bmp_class) {
    TkzChar c = tkz_decode_raw(b0, b1, b2, b3);
    if (b0 >= 0x80 && c.units == 1 && !c.bad) c.uc = bmp_class[c.cp];
    return c;
}

TKZ_HD bool tkz_uc_is_letter(uint8_t uc) { return uc >= UC_LU && uc <= UC_LO; }

// ---- o200k: CODE-POINT semantics ---------------------------------------------------------------
// o200k exists only in the TypeScript reference, which compiles the pattern with `new RegExp(pattern, "gu")`
// (tokenizer_ts/src/tikTokenizer.ts:100; the C# builder does not know the encoding, TokenizerBuilder.cs:109-181): one class test
// per CHARACTER -- a supplementary-plane char has its own Unicode class, can be the optional one-char prefix and counts once
// toward \p{N}{1,3} -- and \s is ECMAScript's WhiteSpace + LineTerminator: U+FEFF is white space, U+0085 is not.
// The class table in HBM holds one byte per code point below TKZ_UCD_DIRECT (BMP + planes 1-3, Unicode 13.0: one gather per char),
// followed by {uint32 n, n x {first, last | class << 24}} for the few assigned ranges above (plane 14's variation selectors).
#define TKZ_UCD_DIRECT 0x40000u
TKZ_HD uint8_t tkz_supp_class(const uint8_t* ucd, uint32_t cp) {
    if (cp < TKZ_UCD_DIRECT) return ucd[cp];
    const uint32_t* t = reinterpret_cast<const uint32_t*>(ucd + TKZ_UCD_DIRECT);
    for (uint32_t i = 0; i < t[0]; ++i)
        if (cp >= t[1 + 2 * i] && cp <= (t[2 + 2 * i] & 0xFFFFFFu)) return (uint8_t)(t[2 + 2 * i] >> 24);
    return (uint8_t)UC_OTHER;
}
// what the o200k matcher sees: `units` becomes 1 for every well-formed char ("one class test per char")
TKZ_HD void tkz_char_to_code_point_semantics(TkzChar& c, const uint8_t* ucd) {
    if (c.bad) return;
    if (c.units == 2) { c.uc = tkz_supp_class(ucd, c.cp); c.units = 1; }
    else if (c.cp == 0xFEFFu) c.uc = UC_WS;
    else if (c.cp == 0x85u) c.uc = UC_OTHER;
}

// ---- classes of pattern 1 / cl100k (TokenizerBuilder.cs:112,128): one small code per char --------
//   PC_O1   [^\s\p{L}\p{N}], one UTF-16 unit          PC_O2  the same, two units (supplementary plane)
//   PC_L    \p{L}     PC_N  \p{N}     PC_CRLF  \r or \n     PC_SP  ' '     PC_WS  any other \s
//   PC_NONE no char (before the start / after the end of the document)
enum : int { PC_NONE = 0, PC_O1 = 1, PC_O2 = 2, PC_L = 3, PC_N = 4, PC_CRLF = 5, PC_SP = 6, PC_WS = 7 };

TKZ_HD int tkz_pc_of(const TkzChar& c) {
    if (c.units == 2) return PC_O2;
    if (tkz_uc_is_letter(c.uc)) return PC_L;
    if (c.uc == UC_N) return PC_N;
    if (c.uc == UC_WS) return (c.cp == '\r' || c.cp == '\n') ? PC_CRLF : (c.cp == ' ' ? PC_SP : PC_WS);
    return PC_O1;
}
TKZ_HD bool tkz_pc_is_other(int pc) { return pc == PC_O1 || pc == PC_O2; }
TKZ_HD bool tkz_pc_is_ws(int pc) { return pc >= PC_CRLF; }

// Contraction literal after an apostrophe: b1, b2 are the two bytes that follow it.
// Returns the length in bytes of `'` + literal (2 or 3), or 0.
//   case-sensitive list  's|'t|'re|'ve|'m|'ll|'d            (pattern 1, TokenizerBuilder.cs:128)
//   (?i:...)             ASCII case pairs only (net6.0)     (cl100k,   TokenizerBuilder.cs:112)
TKZ_HD int tkz_contraction_len(uint32_t b1, uint32_t b2, bool ignore_case) {
    if (ignore_case) {
        if (b1 - 'A' < 26u) b1 |= 0x20;
        if (b2 - 'A' < 26u) b2 |= 0x20;
    }
    if (b1 == 's' || b1 == 't' || b1 == 'm' || b1 == 'd') return 2;
    if ((b1 == 'r' && b2 == 'e') || (b1 == 'v' && b2 == 'e') || (b1 == 'l' && b2 == 'l')) return 3;
    return 0;
}
// The explicit o200k list (tokenizer_ts/src/tokenizerBuilder.ts:80-81):
//   's|'S|'t|'T|'re|'RE|'Re|'eR|'ve|'VE|'vE|'Ve|'m|'M|'ll|'lL|'Ll|'LL|'d|'D     ('eR sits where 'rE would be)
TKZ_HD int tkz_contraction_len_o200k(uint32_t a, uint32_t b) {
    if (a == 's' || a == 'S' || a == 't' || a == 'T' || a == 'm' || a == 'M' || a == 'd' || a == 'D') return 2;
    if ((a == 'r' && b == 'e') || (a == 'R' && b == 'E') || (a == 'R' && b == 'e') || (a == 'e' && b == 'R')) return 3;
    if ((a == 'v' || a == 'V') && (b == 'e' || b == 'E')) return 3;
    if ((a == 'l' || a == 'L') && (b == 'l' || b == 'L')) return 3;
    return 0;
}
