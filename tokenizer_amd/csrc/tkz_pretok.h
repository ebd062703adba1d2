// tkz_pretok.h -- Regex.Matches(text) of the three split patterns the reference defines
// (Tokenizer_C#/TokenizerLib/TikTokenizer.cs:77,252; patterns TokenizerBuilder.cs:112,128 and
// tokenizer_ts/src/tokenizerBuilder.ts:79-89), as scanners over UTF-8 bytes.
//
// Two formulations, which must agree bit for bit:
//
//  (1) tkz_match_*: the sequential matcher -- from a match start p, the end of the leftmost-first
//      match.  Char-level restatement of each alternation with its backtracking resolved by hand.
//      Used by the per-document fallback kernel (o200k today, and the cross-check of (2) in tests).
//
//  (2) TkzRowScan: the position-parallel formulation for pattern 1 and cl100k -- one wavefront
//      takes 64 consecutive bytes ("a row"), every lane classifies its byte, class masks come from
//      64-lane ballots, and "does a piece start here" is a function of the neighbouring chars plus
//      three run scans (index in a digit run mod 3; CR/LF absorbed by the preceding
//      `[^\s\p{L}\p{N}]+[\r\n]*`; a CR/LF further on in the same white-space run for `\s*[\r\n]+`).
//      Rows are processed in order by a wave, carrying the scan state in scalars.
//      Derivation of the local rules: DESIGN.md "K1".
#pragma once
#include <stdint.h>

#include "tkz_classes.h"
#include "tkz_simt.h"

// o200k twice: TKZ_PAT_O200K is the TypeScript reference's engine (code points, ECMAScript \s), TKZ_PAT_O200K_NET what .NET's Regex
// makes of the same string (code units: a supplementary-plane char is two OTHER units; .NET \s) -- tkz_classes.h.  One set of scanners:
// the ASCII block scanner cannot tell them apart, the char-level one and the sequential matcher take `by_code_point`.
enum { TKZ_PAT_P1 = 1, TKZ_PAT_CL100K = 2, TKZ_PAT_O200K = 3, TKZ_PAT_O200K_NET = 4 };
TKZ_HD bool tkz_pat_is_o200k(int pattern) { return pattern >= TKZ_PAT_O200K; }

// =================================================================================================
// (1) sequential matcher
// =================================================================================================
struct TkzDoc { const uint8_t* b; int64_t n; const uint8_t* bmp; int by_code_point; };   // by_code_point: o200k (tkz_classes.h)

TKZ_HD uint32_t tkz_doc_byte(const TkzDoc& d, int64_t p) { return p < d.n ? d.b[p] : 0u; }
TKZ_HD TkzChar tkz_doc_char(const TkzDoc& d, int64_t p) {
    TkzChar c = tkz_decode(d.b[p], tkz_doc_byte(d, p + 1), tkz_doc_byte(d, p + 2), tkz_doc_byte(d, p + 3), d.bmp);
    if (d.by_code_point) tkz_char_to_code_point_semantics(c, d.bmp);
    return c;
}
TKZ_HD int64_t tkz_prev_char(const TkzDoc& d, int64_t p, int64_t lo) {   // start of the char that ends at p
    int64_t s = p - 1;
    int k = 0;
    while (s > lo && k < 3 && (d.b[s] & 0xC0) == 0x80) { --s; ++k; }
    // if that is not a lead covering p-1 (malformed input), fall back to the single byte
    const TkzChar c = tkz_doc_char(d, s);
    return (s + c.len == p) ? s : p - 1;
}

// white-space alternatives shared by the patterns.  crlf_alt: `\s*[\r\n]+` is present (cl100k, o200k).
TKZ_HD int64_t tkz_match_ws(const TkzDoc& d, int64_t p, bool crlf_alt) {
    int64_t k = p, last = p, after_crlf = -1;
    while (k < d.n) {
        const TkzChar c = tkz_doc_char(d, k);
        if (c.uc != UC_WS || c.units == 2) break;
        last = k; k += c.len;
        if (c.cp == '\r' || c.cp == '\n') after_crlf = k;
    }
    if (crlf_alt && after_crlf >= 0) return after_crlf;   // \s*[\r\n]+ : backtracks to the LAST CR/LF of the run
    if (k == d.n) return k;                               // \s+(?!\S) holds at the end of the text
    if (last > p) return last;                            // \s+(?!\S) gives back one char
    return k;                                             // \s+
}

TKZ_HD int64_t tkz_match_p1(const TkzDoc& d, int64_t p) {
    const TkzChar c0 = tkz_doc_char(d, p);
    if (c0.cp == '\'') {
        const int k = tkz_contraction_len(tkz_doc_byte(d, p + 1), tkz_doc_byte(d, p + 2), false);
        if (k) return p + k;
    }
    int64_t q = p;
    int pc = tkz_pc_of(c0);
    if (pc == PC_SP && p + 1 < d.n) {                     // ' ?' is greedy: try the space first
        const int pc1 = tkz_pc_of(tkz_doc_char(d, p + 1));
        if (pc1 == PC_L || pc1 == PC_N || tkz_pc_is_other(pc1)) { q = p + 1; pc = pc1; }
    }
    if (pc == PC_L || pc == PC_N || tkz_pc_is_other(pc)) {
        const bool oth = tkz_pc_is_other(pc);
        int64_t k = q;
        while (k < d.n) {
            const TkzChar c = tkz_doc_char(d, k);
            const int x = tkz_pc_of(c);
            if (oth ? !tkz_pc_is_other(x) : x != pc) break;
            k += c.len;
        }
        return k;
    }
    return tkz_match_ws(d, p, false);
}

TKZ_HD int64_t tkz_match_cl100k(const TkzDoc& d, int64_t p) {
    const TkzChar c0 = tkz_doc_char(d, p);
    if (c0.cp == '\'') {
        const int k = tkz_contraction_len(tkz_doc_byte(d, p + 1), tkz_doc_byte(d, p + 2), true);
        if (k) return p + k;
    }
    const int pc0 = tkz_pc_of(c0);
    {   // [^\r\n\p{L}\p{N}]?\p{L}+
        int64_t q = -1;
        if (pc0 == PC_L) q = p;
        else if ((pc0 == PC_O1 || pc0 == PC_SP || pc0 == PC_WS) && p + c0.len < d.n &&
                 tkz_pc_of(tkz_doc_char(d, p + c0.len)) == PC_L) q = p + c0.len;
        if (q >= 0) {
            int64_t k = q;
            while (k < d.n) { const TkzChar c = tkz_doc_char(d, k); if (tkz_pc_of(c) != PC_L) break; k += c.len; }
            return k;
        }
    }
    if (pc0 == PC_N) {   // \p{N}{1,3}
        int64_t k = p; int cnt = 0;
        while (k < d.n && cnt < 3) { const TkzChar c = tkz_doc_char(d, k); if (tkz_pc_of(c) != PC_N) break; k += c.len; ++cnt; }
        return k;
    }
    {   //  ?[^\s\p{L}\p{N}]+[\r\n]*
        int64_t q = -1;
        if (tkz_pc_is_other(pc0)) q = p;
        else if (pc0 == PC_SP && p + 1 < d.n && tkz_pc_is_other(tkz_pc_of(tkz_doc_char(d, p + 1)))) q = p + 1;
        if (q >= 0) {
            int64_t k = q;
            while (k < d.n) { const TkzChar c = tkz_doc_char(d, k); if (!tkz_pc_is_other(tkz_pc_of(c))) break; k += c.len; }
            while (k < d.n && (d.b[k] == '\r' || d.b[k] == '\n')) ++k;
            return k;
        }
    }
    return tkz_match_ws(d, p, true);
}

// o200k: A = [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}], B = [\p{Ll}\p{Lm}\p{Lo}\p{M}]
TKZ_HD bool tkz_o2_A(const TkzChar& c) { return c.units == 1 && (c.uc == UC_LU || c.uc == UC_LT || c.uc == UC_LM || c.uc == UC_LO || c.uc == UC_M); }
TKZ_HD bool tkz_o2_B(const TkzChar& c) { return c.units == 1 && (c.uc == UC_LL || c.uc == UC_LM || c.uc == UC_LO || c.uc == UC_M); }
TKZ_HD int64_t tkz_o2_suffix(const TkzDoc& d, int64_t e) {   // (?:'s|'S|...)?
    if (e < d.n && d.b[e] == '\'') e += tkz_contraction_len_o200k(tkz_doc_byte(d, e + 1), tkz_doc_byte(d, e + 2));
    return e;
}
TKZ_HD int64_t tkz_match_o200k(const TkzDoc& d, int64_t p) {
    const TkzChar c0 = tkz_doc_char(d, p);
    const bool is_crlf = c0.cp == '\r' || c0.cp == '\n';
    const bool prefixable = c0.units == 1 && !is_crlf && !tkz_uc_is_letter(c0.uc) && c0.uc != UC_N;
    // alt 1: prefix? A* B+ suffix?
    for (int pre = 1; pre >= 0; --pre) {
        if (pre && !prefixable) continue;
        const int64_t q = p + (pre ? c0.len : 0);
        int64_t t = q;
        while (t < d.n) { const TkzChar c = tkz_doc_char(d, t); if (!tkz_o2_A(c)) break; t += c.len; }
        if (t < d.n && tkz_o2_B(tkz_doc_char(d, t))) {            // B+ continues after the A run
            int64_t k = t;
            while (k < d.n) { const TkzChar c = tkz_doc_char(d, k); if (!tkz_o2_B(c)) break; k += c.len; }
            return tkz_o2_suffix(d, k);
        }
        // A* gives chars back until B+ can start: the last char of the A run that is also in B
        int64_t s = t;
        while (s > q) {
            const int64_t s0 = tkz_prev_char(d, s, q);
            const TkzChar c = tkz_doc_char(d, s0);
            if (tkz_o2_B(c)) return tkz_o2_suffix(d, s0 + c.len);   // what follows (up to t) is in A only, so B+ stops here
            s = s0;
        }
    }
    // alt 2: prefix? A+ B* suffix?
    for (int pre = 1; pre >= 0; --pre) {
        if (pre && !prefixable) continue;
        const int64_t q = p + (pre ? c0.len : 0);
        int64_t t = q;
        while (t < d.n) { const TkzChar c = tkz_doc_char(d, t); if (!tkz_o2_A(c)) break; t += c.len; }
        if (t > q) {
            while (t < d.n) { const TkzChar c = tkz_doc_char(d, t); if (!tkz_o2_B(c)) break; t += c.len; }
            return tkz_o2_suffix(d, t);
        }
    }
    if (c0.uc == UC_N && c0.units == 1) {   // \p{N}{1,3}
        int64_t k = p; int cnt = 0;
        while (k < d.n && cnt < 3) { const TkzChar c = tkz_doc_char(d, k); if (c.uc != UC_N || c.units != 1) break; k += c.len; ++cnt; }
        return k;
    }
    {   //  ?[^\s\p{L}\p{N}]+[\r\n/]*   (the class contains \p{M})
        auto oth = [](const TkzChar& c) { return c.units == 2 || c.uc == UC_OTHER || c.uc == UC_M; };
        int64_t q = -1;
        if (oth(c0)) q = p;
        else if (c0.cp == ' ' && p + 1 < d.n && oth(tkz_doc_char(d, p + 1))) q = p + 1;
        if (q >= 0) {
            int64_t k = q;
            while (k < d.n) { const TkzChar c = tkz_doc_char(d, k); if (!oth(c)) break; k += c.len; }
            while (k < d.n && (d.b[k] == '\r' || d.b[k] == '\n' || d.b[k] == '/')) ++k;
            return k;
        }
    }
    return tkz_match_ws(d, p, true);
}

TKZ_HD int64_t tkz_match_at(int pattern, const TkzDoc& d, int64_t p) {
    return pattern == TKZ_PAT_P1 ? tkz_match_p1(d, p) : pattern == TKZ_PAT_CL100K ? tkz_match_cl100k(d, p) : tkz_match_o200k(d, p);   // (both o200k variants: d.by_code_point)
}

// =================================================================================================
// (2) position-parallel rows (pattern 1, cl100k)
// =================================================================================================
#ifndef TKZ_NO_SIMT

struct TkzRowLane {      // what a lane knows about ITS byte of a row
    int pc;              // PC_* of the char the byte belongs to (PC_NONE outside the corpus)
    int len;             // byte length of that char
    int off;             // distance back to its lead byte (0 for a lead)
    uint32_t b;          // the byte
    int bad;             // malformed UTF-8 seen by this lane
};

// Where the scanners read text from: the corpus in HBM, with an optional window [lo, hi) of it staged in
// LDS (k_pretok_rows stages each wavefront's rows once, 16 B per lane, instead of paying an HBM round trip
// per 64-byte row).  Positions outside [0, total) read as 0.
struct TkzSrc { const uint8_t* bytes; int64_t total; const uint8_t* stage; int64_t lo, hi; };

TKZ_DEV uint32_t tkz_gbyte(const TkzSrc& S, int64_t pos) {
    if (pos >= S.lo && pos < S.hi) return S.stage[pos - S.lo];
    return (pos >= 0 && pos < S.total) ? S.bytes[pos] : 0u;
}

TKZ_DEV TkzRowLane tkz_classify_byte(const TkzSrc& S, int64_t pos, const uint8_t* bmp) {
    TkzRowLane r;
    r.pc = PC_NONE; r.len = 1; r.off = 0; r.b = 0; r.bad = 0;
    if (pos < 0 || pos >= S.total) return r;
    uint32_t m3 = tkz_gbyte(S, pos - 3), m2 = tkz_gbyte(S, pos - 2), m1 = tkz_gbyte(S, pos - 1);
    const uint32_t b0 = tkz_gbyte(S, pos);
    const uint32_t p1 = tkz_gbyte(S, pos + 1), p2 = tkz_gbyte(S, pos + 2), p3 = tkz_gbyte(S, pos + 3);
    r.b = b0;
    int off = 0;
    if ((b0 & 0xC0) == 0x80) {   // continuation byte: find the lead within 3 bytes
        if (m1 >= 0xC2 && pos >= 1) off = 1;
        else if ((m1 & 0xC0) == 0x80 && m2 >= 0xE0 && pos >= 2) off = 2;
        else if ((m1 & 0xC0) == 0x80 && (m2 & 0xC0) == 0x80 && m3 >= 0xF0 && pos >= 3) off = 3;
        else r.bad = 1;
    }
    TkzChar c;
    if (off == 0) c = tkz_decode(b0, p1, p2, p3, bmp);
    else if (off == 1) c = tkz_decode(m1, b0, p1, p2, bmp);
    else if (off == 2) c = tkz_decode(m2, m1, b0, p1, bmp);
    else c = tkz_decode(m3, m2, m1, b0, bmp);
    if (c.bad || off >= c.len) { r.bad = 1; c = tkz_decode(b0, 0, 0, 0, bmp); c.bad = 0; c.len = 1; c.units = 1; c.uc = UC_OTHER; off = 0; }
    r.pc = tkz_pc_of(c); r.len = c.len; r.off = off;
    return r;
}

// Per-row class masks (one bit per byte; a multi-byte char marks all of its bytes):
//   nb \p{N} bytes, nl \p{N} lead bytes, cr CR/LF, wsb \s, L \p{L}, O [^\s\p{L}\p{N}], sp ' ', ap '\'',
//   s1 / rv / e / ll  the letters of the contraction literals (s|t|m|d, r|v, e, l; case-folded for cl100k)
struct TkzRowMasks { uint64_t nb, nl, cr, wsb, L, O, sp, ap, s1, rv, e, ll; };

enum : uint32_t { AF_L = 1, AF_N = 2, AF_CRLF = 4, AF_SP = 8, AF_WSO = 16, AF_AP = 32, AF_S1 = 64, AF_RV = 128, AF_E = 256, AF_LL = 512 };
// class + literal flags of an ASCII byte (b < 128)
TKZ_HD uint32_t tkz_ascii_flags(uint32_t b, bool ignore_case) {
    uint32_t f = 0;
    const uint8_t uc = tkz_ascii_class(b);
    if (tkz_uc_is_letter(uc)) f |= AF_L;
    else if (uc == UC_N) f |= AF_N;
    else if (uc == UC_WS) f |= (b == '\r' || b == '\n') ? AF_CRLF : (b == ' ' ? AF_SP : AF_WSO);
    if (b == '\'') f |= AF_AP;
    const uint32_t lc = (ignore_case && b - 'A' < 26u) ? (b | 0x20u) : b;
    if (lc == 's' || lc == 't' || lc == 'm' || lc == 'd') f |= AF_S1;
    if (lc == 'r' || lc == 'v') f |= AF_RV;
    if (lc == 'e') f |= AF_E;
    if (lc == 'l') f |= AF_LL;
    return f;
}
TKZ_HD int tkz_pc_of_flags(uint32_t f) {
    return (f & AF_L) ? PC_L : (f & AF_N) ? PC_N : (f & AF_CRLF) ? PC_CRLF : (f & AF_SP) ? PC_SP : (f & AF_WSO) ? PC_WS : PC_O1;
}
// masks of a row whose 64 bytes are all ASCII, from the per-lane flags
TKZ_DEV TkzRowMasks tkz_row_masks_ascii(uint32_t f) {
    TkzRowMasks m;
    m.nb = simt::ballot(f & AF_N); m.nl = m.nb;
    m.cr = simt::ballot(f & AF_CRLF);
    m.wsb = simt::ballot(f & (AF_CRLF | AF_SP | AF_WSO));
    m.L = simt::ballot(f & AF_L);
    m.O = ~(m.nb | m.wsb | m.L);
    m.sp = simt::ballot(f & AF_SP);
    m.ap = simt::ballot(f & AF_AP);
    m.s1 = simt::ballot(f & AF_S1); m.rv = simt::ballot(f & AF_RV); m.e = simt::ballot(f & AF_E); m.ll = simt::ballot(f & AF_LL);
    return m;
}
// masks of any row, from the general per-byte classification
TKZ_DEV TkzRowMasks tkz_row_masks(const TkzRowLane& L, bool ignore_case) {
    TkzRowMasks m;
    m.nb = simt::ballot(L.pc == PC_N);
    m.nl = simt::ballot(L.pc == PC_N && L.off == 0);
    m.cr = simt::ballot(L.pc == PC_CRLF);
    m.wsb = simt::ballot(tkz_pc_is_ws(L.pc));
    m.L = simt::ballot(L.pc == PC_L);
    m.O = simt::ballot(tkz_pc_is_other(L.pc));
    m.sp = simt::ballot(L.pc == PC_SP);
    const uint32_t f = (L.pc != PC_NONE && L.b < 128u) ? tkz_ascii_flags(L.b, ignore_case) : 0u;
    m.ap = simt::ballot(f & AF_AP);
    m.s1 = simt::ballot(f & AF_S1); m.rv = simt::ballot(f & AF_RV); m.e = simt::ballot(f & AF_E); m.ll = simt::ballot(f & AF_LL);
    return m;
}

// value of lane `rel` in the three-row window (rel in [-64, 128))
TKZ_DEV int tkz_win3(int vP, int vC, int vN, int rel) {
    const int a = simt::shfl(vP, rel & 63), b = simt::shfl(vC, rel & 63), c = simt::shfl(vN, rel & 63);
    return rel < 0 ? a : (rel < 64 ? b : c);
}
TKZ_DEV int tkz_win2(int vP, int vC, int rel) {   // rel in [-64, 64)
    const int a = simt::shfl(vP, rel & 63), b = simt::shfl(vC, rel & 63);
    return rel < 0 ? a : b;
}
TKZ_DEV int tkz_bit3(uint64_t mP, uint64_t mC, uint64_t mN, int rel) {
    const uint64_t m = rel < 0 ? mP : (rel < 64 ? mC : mN);
    return (int)((m >> (rel & 63)) & 1ull);
}

// T = S | (R & (T << 1)) : propagate seeds S (subset of R) upward through runs of R.  64-bit.
TKZ_HD uint64_t tkz_fill_up64(uint64_t S, uint64_t R) { return ((~(R + S)) & R) | S; }
// the same over 128 bits (lo = bits 0..63)
TKZ_HD void tkz_fill_up128(uint64_t Slo, uint64_t Shi, uint64_t Rlo, uint64_t Rhi, uint64_t* Tlo, uint64_t* Thi) {
    const uint64_t lo = Rlo + Slo;
    const uint64_t carry = lo < Rlo ? 1ull : 0ull;
    const uint64_t hi = Rhi + Shi + carry;
    *Tlo = ((~lo) & Rlo) | Slo;
    *Thi = ((~hi) & Rhi) | Shi;
}

struct TkzScanCarry {        // wave-uniform state carried from row to row
    int nb63;                // previous row: its last byte is an N byte
    int carryN;              // N chars of the running digit run before this row, mod 3
    int abs63;               // previous row: its last byte is an absorbed CR/LF
    int64_t sa_from, sa_end, sa_lastcr;   // scan-ahead cache: a connected \s run covers [sa_from, sa_end), last CR/LF at sa_lastcr
};

// Does a connected white-space run that starts at absolute position `from` (a row boundary) contain a
// CR/LF?  Only needed when a white-space run covers the whole look-ahead row.  Wave-uniform.
TKZ_DEV bool tkz_scan_ahead_crlf(const TkzSrc& S, const uint64_t* docbits, int64_t nrows,
                                 const uint8_t* bmp, int64_t from, TkzScanCarry& cy) {
    if (cy.sa_from >= 0 && from >= cy.sa_from && from < cy.sa_end) return cy.sa_lastcr >= from;
    int64_t row = from >> 6;
    int64_t lastcr = -1, end = from;
    for (;; ++row) {
        if (row >= nrows) { end = row << 6; break; }
        const TkzRowLane L = tkz_classify_byte(S, (row << 6) + simt::lane(), bmp);
        const uint64_t wsb = simt::ballot(tkz_pc_is_ws(L.pc));
        const uint64_t cr = simt::ballot(L.pc == PC_CRLF);
        const uint64_t conn = wsb & ~docbits[row];
        const uint64_t brk = ~conn;
        const int m = brk ? tkz_ctz64(brk) : 64;
        const uint64_t c = cr & tkz_lowmask(m);
        if (c) lastcr = (row << 6) + tkz_msb64(c);
        if (m < 64) { end = (row << 6) + m; break; }
    }
    cy.sa_from = from; cy.sa_end = end; cy.sa_lastcr = lastcr;
    return lastcr >= from;
}

// Evaluate one row.  P/C/N = lane values of the previous / current / next row; ds* = document-start
// bits; mC/mN = class masks of the current / next row; fP = per-lane flags of the previous row
// (clen, o1ms), fC receives the current row's.  Returns the piece-start mask of the current row.
template <int PATTERN>
TKZ_DEV uint64_t tkz_row_eval(const TkzRowLane& P, const TkzRowLane& C, const TkzRowLane& N,
                              uint64_t dsP, uint64_t dsC, uint64_t dsN,
                              const TkzRowMasks& mC, const TkzRowMasks& mN,
                              int clenP, int o1msP, int* clenC_out, int* o1msC_out,
                              uint64_t* c2C, uint64_t* c3C, uint64_t* o1C,
                              TkzScanCarry& cy,
                              const TkzSrc& S, const uint64_t* docbits, int64_t nrows,
                              const uint8_t* bmp, int64_t row) {
    const int lane = simt::lane();
    const int lead = lane - C.off;                       // >= -3
    const int dsLead = tkz_bit3(dsP, dsC, dsN, lead);
    int p = tkz_win3(P.pc, C.pc, N.pc, lead - 1);
    if (dsLead) p = PC_NONE;
    const int nx = lead + C.len;                         // <= 66
    int n = tkz_win3(P.pc, C.pc, N.pc, nx);
    if (tkz_bit3(dsP, dsC, dsN, nx)) n = PC_NONE;
    const bool isLead = C.off == 0 && C.pc != PC_NONE;

    // contraction at an apostrophe that is a match start
    int clen = 0;
    if (C.b == '\'' && C.pc == PC_O1 && !tkz_pc_is_other(p) && p != PC_SP) {
        const int64_t pos = (row << 6) + lane;
        const int k = tkz_contraction_len(tkz_gbyte(S, pos + 1), tkz_gbyte(S, pos + 2), PATTERN == TKZ_PAT_CL100K);
        if (k && !tkz_bit3(dsP, dsC, dsN, lane + 1) && (k == 2 || !tkz_bit3(dsP, dsC, dsN, lane + 2))) clen = k;
    }
    const int o1ms = (C.pc == PC_O1 && !tkz_pc_is_other(p) && p != PC_SP && !tkz_pc_is_other(n)) ? 1 : 0;
    *clenC_out = clen; *o1msC_out = o1ms;
    *c2C = simt::ballot(clen == 2); *c3C = simt::ballot(clen == 3); *o1C = simt::ballot(o1ms != 0);
    const int cback2 = tkz_win2(clenP, clen, lane - 2), cback3 = tkz_win2(clenP, clen, lane - 3);   // (collectives: evaluate both)
    const bool contrEnd = cback2 == 2 || cback3 == 3;
    const int clenPrev = tkz_win2(clenP, clen, lead - 1);
    const int o1msPrev = tkz_win2(o1msP, o1ms, lead - 1);

    bool start = false;
    if (PATTERN == TKZ_PAT_P1) {
        if (C.pc == PC_L) start = p != PC_L && p != PC_SP && clenPrev == 0;
        else if (C.pc == PC_N) start = p != PC_N && p != PC_SP;
        else if (tkz_pc_is_other(C.pc)) start = !tkz_pc_is_other(p) && p != PC_SP;
        else if (tkz_pc_is_ws(C.pc)) start = !tkz_pc_is_ws(p) || (!tkz_pc_is_ws(n) && n != PC_NONE);
    } else {
        // ---- run scans (wave-uniform mask algebra) ----
        // (a) index of an N char inside its digit run
        const uint64_t startsN = mC.nb & (dsC | ~((mC.nb << 1) | (uint64_t)(cy.nb63 & 1)));
        int idx = 0;
        {
            const uint64_t below = tkz_lowmask(lane);                 // bits < lane
            const uint64_t m = startsN & (below | (1ull << lane));    // run starts at or below me
            if (m) { const int rs = tkz_msb64(m); idx = tkz_popc64(mC.nl & below & ~tkz_lowmask(rs)); }
            else idx = cy.carryN + tkz_popc64(mC.nl & below);
        }
        // (b) CR/LF absorbed by a preceding ` ?[^\s\p{L}\p{N}]+[\r\n]*`
        const uint64_t R = mC.cr & ~dsC;
        uint64_t seeds = simt::ballot(C.pc == PC_CRLF && tkz_pc_is_other(p));
        if (cy.abs63) seeds |= R & 1ull;
        const uint64_t ABS = tkz_fill_up64(seeds & R, R);
        // (c) T(j) = CR(j) | (CONN(j+1) & T(j+1)): a CR/LF at or after j inside the same \s run
        uint64_t Tcur;
        {
            const uint64_t connC = mC.wsb & ~dsC, connN = mN.wsb & ~dsN;
            uint64_t crN = mN.cr;
            if (connN == ~0ull && (mC.wsb >> 63)) {      // the run may continue past the look-ahead row
                const int64_t from = (row + 2) << 6;
                const bool conn128 = from < (nrows << 6) && !(docbits[row + 2] & 1ull) &&
                                     tkz_pc_is_ws(simt::first_lane(tkz_classify_byte(S, from, bmp).pc));
                if (conn128 && tkz_scan_ahead_crlf(S, docbits, nrows, bmp, from, cy)) crN |= 1ull << 63;
            }
            // reversed positions k = 127 - j : S'(k) = CR(127-k), G'(k) = CONN(128-k)
            const uint64_t Slo = tkz_brev64(crN), Shi = tkz_brev64(mC.cr);
            // CONN as a 128-bit value (lo = connC, hi = connN); G' = brev128(CONN) << 1
            const uint64_t rlo = tkz_brev64(connN), rhi = tkz_brev64(connC);
            const uint64_t Glo = rlo << 1, Ghi = (rhi << 1) | (rlo >> 63);
            const uint64_t S2lo = (Slo << 1) & Glo, S2hi = ((Shi << 1) | (Slo >> 63)) & Ghi;
            uint64_t Plo, Phi;
            tkz_fill_up128(S2lo, S2hi, Glo, Ghi, &Plo, &Phi);
            (void)Plo;
            Tcur = tkz_brev64(Shi | Phi);
        }
        const bool absorbed = (ABS >> lane) & 1ull;
        const bool absorbedPrev = lane == 0 ? (cy.abs63 != 0) : (((ABS >> ((lane - 1) & 63)) & 1ull) != 0);
        const bool crlfAhead = (Tcur >> lane) & 1ull;

        if (C.pc == PC_L) start = p != PC_L && p != PC_SP && p != PC_WS && !o1msPrev;
        else if (C.pc == PC_N) start = (idx % 3) == 0;
        else if (tkz_pc_is_other(C.pc)) start = !tkz_pc_is_other(p) && p != PC_SP;
        else if (tkz_pc_is_ws(C.pc))
            start = !absorbed && ((!tkz_pc_is_ws(p) || absorbedPrev) || (p == PC_CRLF && !crlfAhead) ||
                                  (C.pc != PC_CRLF && !tkz_pc_is_ws(n) && n != PC_NONE));

        // carries for the next row
        int carryN = 0;
        if (mC.nb >> 63) carryN = (startsN ? tkz_popc64(mC.nl & ~tkz_lowmask(tkz_msb64(startsN))) : cy.carryN + tkz_popc64(mC.nl)) % 3;
        cy.carryN = carryN;
        cy.nb63 = (int)(mC.nb >> 63);
        cy.abs63 = (int)(ABS >> 63);
    }
    start = start || contrEnd;
    return simt::ballot(start && isLead) | dsC;
}

// The same for a row whose own 64 bytes and whose successor row are pure ASCII: every quantity is a
// 64-bit mask and the rules of tkz_row_eval become wave-uniform bit algebra (no per-lane work at all).
// mP is only consulted for bit 63 of its masks (the char before this row).
template <int PATTERN>
TKZ_DEV uint64_t tkz_row_eval_ascii(const TkzRowMasks& mP, const TkzRowMasks& mC, const TkzRowMasks& mN,
                                    uint64_t dsC, uint64_t dsN, uint64_t c2P, uint64_t c3P, uint64_t o1P,
                                    uint64_t* c2C, uint64_t* c3C, uint64_t* o1C, TkzScanCarry& cy,
                                    const TkzSrc& S, const uint64_t* docbits, int64_t nrows,
                                    const uint8_t* bmp, int64_t row) {
    const uint64_t nds = ~dsC;
    const uint64_t KN = (dsC >> 1) | (dsN << 63);                 // the NEXT position starts a document
    const uint64_t L = mC.L, N = mC.nb, O = mC.O, CR = mC.cr, SP = mC.sp, W = mC.wsb, AP = mC.ap;
#define TKZ_PREV(xC, xP) ((((xC) << 1) | ((xP) >> 63)) & nds)
#define TKZ_NEXT(xC, xN) ((((xC) >> 1) | ((xN) << 63)) & ~KN)
    const uint64_t pL = TKZ_PREV(L, mP.L), pN = TKZ_PREV(N, mP.nb), pO = TKZ_PREV(O, mP.O), pSP = TKZ_PREV(SP, mP.sp);
    const uint64_t pW = TKZ_PREV(W, mP.wsb), pCR = TKZ_PREV(CR, mP.cr);
    const uint64_t nO = TKZ_NEXT(O, mN.O);
    const uint64_t nReal = TKZ_NEXT(L | N | O, mN.L | mN.nb | mN.O);   // a next char exists in the document and is not \s
    // contractions: an apostrophe that is a match start, followed by a literal inside the document
    uint64_t c2 = 0, c3 = 0;
    if (AP) {
        const uint64_t KN2 = (dsC >> 2) | (dsN << 62);
        const uint64_t n1S1 = (mC.s1 >> 1) | (mN.s1 << 63), n1RV = (mC.rv >> 1) | (mN.rv << 63), n1LL = (mC.ll >> 1) | (mN.ll << 63);
        const uint64_t n2E = (mC.e >> 2) | (mN.e << 62), n2LL = (mC.ll >> 2) | (mN.ll << 62);
        const uint64_t ms = AP & ~pO & ~pSP & ~KN;
        c2 = ms & n1S1;
        c3 = ms & ~KN2 & ((n1RV & n2E) | (n1LL & n2LL));
    }
    *c2C = c2; *c3C = c3;
    const uint64_t contrEnd = (c2 << 2) | (c3 << 3) | (c2P >> 62) | (c3P >> 61);
    uint64_t start;
    if (PATTERN == TKZ_PAT_P1) {
        const uint64_t clenPrev = ((c2 | c3) << 1) | ((c2P | c3P) >> 63);
        *o1C = 0;
        start = (L & ~pL & ~pSP & ~clenPrev) | (N & ~pN & ~pSP) | (O & ~pO & ~pSP) | (W & (~pW | nReal));
    } else {
        const uint64_t o1 = O & ~pO & ~pSP & ~nO;
        *o1C = o1;
        const uint64_t o1Prev = (o1 << 1) | (o1P >> 63);
        const uint64_t pWSo = pW & ~pCR & ~pSP;
        const uint64_t sL = L & ~pL & ~pSP & ~pWSo & ~o1Prev;
        const uint64_t sO = O & ~pO & ~pSP;
        // \p{N}{1,3}: every third digit of a run
        uint64_t sN = 0;
        int carryN = 0;
        if (N) {
            const uint64_t Q = N & pN;                            // continues the run of the previous byte
            uint64_t S = N & ~pN;                                 // runs that start in this row
            if (Q & 1ull) {                                       // a run entering from the previous row
                const int d = (3 - cy.carryN) % 3, lead = (~Q) ? tkz_ctz64(~Q) : 64;
                if (d < lead) S |= 1ull << d;
            }
            const uint64_t Q3 = Q & (Q << 1) & (Q << 2);
            uint64_t T = S | ((S << 3) & Q3);
            const uint64_t Q6 = Q3 & (Q3 << 3);
            if (Q6) {
                T |= (T << 6) & Q6;
                const uint64_t Q12 = Q6 & (Q6 << 6);
                if (Q12) {
                    T |= (T << 12) & Q12;
                    const uint64_t Q24 = Q12 & (Q12 << 12);
                    if (Q24) { T |= (T << 24) & Q24; const uint64_t Q48 = Q24 & (Q24 << 24); T |= (T << 48) & Q48; }
                }
            }
            sN = T;
            if (N >> 63) carryN = (~Q) ? (64 - tkz_msb64(~Q)) % 3 : (cy.carryN + 64) % 3;
        }
        // white space
        uint64_t ABS = 0;
        if (CR) {
            const uint64_t R = CR & nds;
            uint64_t seeds = CR & pO;
            if (cy.abs63) seeds |= R & 1ull;
            ABS = tkz_fill_up64(seeds & R, R);
        }
        uint64_t Tcur = 0;
        {
            const uint64_t connC = W & nds, connN = mN.wsb & ~dsN;
            const bool beyond = connN == ~0ull && (W >> 63);
            if ((CR | mN.cr) || beyond) {
                uint64_t crN = mN.cr;
                if (beyond) {
                    const int64_t from = (row + 2) << 6;
                    const bool conn128 = from < (nrows << 6) && !(docbits[row + 2] & 1ull) &&
                                         tkz_pc_is_ws(simt::first_lane(tkz_classify_byte(S, from, bmp).pc));
                    if (conn128 && tkz_scan_ahead_crlf(S, docbits, nrows, bmp, from, cy)) crN |= 1ull << 63;
                }
                const uint64_t Slo = tkz_brev64(crN), Shi = tkz_brev64(CR);
                const uint64_t rlo = tkz_brev64(connN), rhi = tkz_brev64(connC);
                const uint64_t Glo = rlo << 1, Ghi = (rhi << 1) | (rlo >> 63);
                const uint64_t S2lo = (Slo << 1) & Glo, S2hi = ((Shi << 1) | (Slo >> 63)) & Ghi;
                uint64_t Plo, Phi;
                tkz_fill_up128(S2lo, S2hi, Glo, Ghi, &Plo, &Phi);
                (void)Plo;
                Tcur = tkz_brev64(Shi | Phi);
            }
        }
        const uint64_t pABS = (ABS << 1) | (uint64_t)(cy.abs63 & 1);
        const uint64_t sW = W & ~ABS & ((~pW | pABS) | (pCR & ~Tcur) | (~CR & nReal));
        start = sL | sN | sO | sW;
        cy.carryN = carryN;
        cy.nb63 = (int)(N >> 63);
        cy.abs63 = (int)(ABS >> 63);
    }
#undef TKZ_PREV
#undef TKZ_NEXT
    return start | contrEnd | dsC;
}

// =================================================================================================
// (2b) one ROW PER LANE: 64 consecutive ASCII rows evaluated at once.
//
// Lane j owns row (first_row + j): its 64 bytes are classified with SWAR arithmetic on 16 dwords (no
// ballots, no per-byte work), giving the same 64-bit class masks as tkz_row_masks_ascii -- but in that
// lane's VGPRs -- and the rules of tkz_row_eval_ascii are then ordinary per-lane 64-bit arithmetic, 64 rows
// per instruction instead of one row per scalar instruction.  Every cross-row dependency is brought
// down to the nearest neighbour (two lane-shift exchanges) by refusing blocks in which a digit run or a
// white-space run covers a whole 64-byte row; those blocks, blocks with a non-ASCII byte and blocks that
// are not entirely inside the corpus are left to the sequential row loop.  Lanes 0 and 63 are context:
// only lanes 1..62 produce output rows.
// =================================================================================================
constexpr int kBlockRowStride = 80;       // bytes between rows in LDS: keeps 16-byte alignment, spreads the 4 x b128 row reads over all banks

// bit 7 of every byte of x that lies in [lo, hi]; all bytes of x are < 0x80
TKZ_HD uint32_t tkz_swar_range(uint32_t x, uint32_t lo, uint32_t hi) {
    return (x + (0x80u - lo) * 0x01010101u) & ~(x + (0x7Fu - hi) * 0x01010101u) & 0x80808080u;
}
TKZ_HD uint32_t tkz_swar_eq(uint32_t x, uint32_t c) {
    const uint32_t z = x ^ (c * 0x01010101u);
    return ~(z + 0x7F7F7F7Fu) & 0x80808080u;
}
// the four bit-7 flags of r as a nibble (byte 0 -> bit 0)
TKZ_HD uint32_t tkz_swar_nibble(uint32_t r) { return (((r >> 7) & 0x01010101u) * 0x01020408u) >> 24; }

struct TkzBlockMasks { uint64_t L, N, O, W, CR, SP, AP, UP, SL, HI, CONT; uint32_t hi; };   // HI: bytes >= 0x80, CONT: 10xxxxxx

// classify the 64 bytes of this lane's row (16 dwords at `row`, 16-byte aligned); CASES: also upper-case letters and '/'
template <bool CASES>
TKZ_HD TkzBlockMasks tkz_block_classify(const uint4* row) {
    uint32_t mL[2] = {0, 0}, mN[2] = {0, 0}, mW[2] = {0, 0}, mC[2] = {0, 0}, mS[2] = {0, 0}, mA[2] = {0, 0}, mU[2] = {0, 0}, mX[2] = {0, 0};
    uint32_t mH[2] = {0, 0}, mT[2] = {0, 0};
    uint32_t hi = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = row[q];
        const uint32_t xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 4 * q + t;                       // dword index: bytes 4k .. 4k+3
            const uint32_t xr = xs[t];
            hi |= xr;
            // (the range tests add per byte and must not carry into the next one: they run on the low 7 bits; what they say
            //  about a byte >= 0x80 is meaningless and masked by the caller with HI)
            const uint32_t x = xr & 0x7F7F7F7Fu;
            const uint32_t y = x | 0x20202020u;
            const int h = k >> 3, sh = 4 * (k & 7);
            mL[h] |= tkz_swar_nibble(tkz_swar_range(y, 'a', 'z')) << sh;
            mN[h] |= tkz_swar_nibble(tkz_swar_range(x, '0', '9')) << sh;
            mW[h] |= tkz_swar_nibble(tkz_swar_range(x, 9, 13) | tkz_swar_eq(x, ' ')) << sh;
            mC[h] |= tkz_swar_nibble(tkz_swar_eq(x, '\n') | tkz_swar_eq(x, '\r')) << sh;
            mS[h] |= tkz_swar_nibble(tkz_swar_eq(x, ' ')) << sh;
            mA[h] |= tkz_swar_nibble(tkz_swar_eq(x, '\'')) << sh;
            mH[h] |= tkz_swar_nibble(xr & 0x80808080u) << sh;
            mT[h] |= tkz_swar_nibble(xr & ~(xr << 1) & 0x80808080u) << sh;
            if (CASES) {
                mU[h] |= tkz_swar_nibble(tkz_swar_range(x, 'A', 'Z')) << sh;
                mX[h] |= tkz_swar_nibble(tkz_swar_eq(x, '/')) << sh;
            }
        }
    }
    TkzBlockMasks m;
    m.L = ((uint64_t)mL[1] << 32) | mL[0]; m.N = ((uint64_t)mN[1] << 32) | mN[0]; m.W = ((uint64_t)mW[1] << 32) | mW[0];
    m.CR = ((uint64_t)mC[1] << 32) | mC[0]; m.SP = ((uint64_t)mS[1] << 32) | mS[0]; m.AP = ((uint64_t)mA[1] << 32) | mA[0];
    m.UP = ((uint64_t)mU[1] << 32) | mU[0]; m.SL = ((uint64_t)mX[1] << 32) | mX[0];
    m.HI = ((uint64_t)mH[1] << 32) | mH[0]; m.CONT = ((uint64_t)mT[1] << 32) | mT[0];
    m.O = ~(m.L | m.N | m.W);
    m.hi = hi & 0x80808080u;
    return m;
}

// ---- rows with multi-byte chars: the same algebra on CHARS -------------------------------------------
// The rules are stated on chars.  On an ASCII row a char is a byte and the masks above are already char masks; on
// a row with multi-byte chars every lane packs its masks down to one bit per char (the continuation bytes are
// squeezed out: compress / expand of Hacker's Delight 7-4, 7-5), evaluates the same algebra on n <= 64 chars
// instead of 64, and spreads the piece starts back onto the lead bytes.  A char belongs to the row its lead
// byte is in.
struct TkzPext { uint64_t mv[6]; };
TKZ_HD TkzPext tkz_pext_prepare(uint64_t m) {
    TkzPext p;
    uint64_t mk = ~m << 1;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        uint64_t mp = mk ^ (mk << 1);
        mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16; mp ^= mp << 32;
        const uint64_t mv = mp & m;
        p.mv[i] = mv;
        m = (m ^ mv) | (mv >> (1 << i));
        mk &= ~mp;
    }
    return p;
}
TKZ_HD uint64_t tkz_pext(uint64_t x, uint64_t m, const TkzPext& p) {      // the bits of x under m, packed to the low end
    x &= m;
#pragma unroll
    for (int i = 0; i < 6; ++i) { const uint64_t t = x & p.mv[i]; x = (x ^ t) | (t >> (1 << i)); }
    return x;
}
TKZ_HD uint64_t tkz_pdep(uint64_t x, uint64_t m, const TkzPext& p) {      // bit k of x to the k-th set bit of m
#pragma unroll
    for (int i = 5; i >= 0; --i) { const uint64_t t = x << (1 << i); x = (x & ~p.mv[i]) | (t & p.mv[i]); }
    return x & m;
}

// ---- states that can cross a whole row ----------------------------------------------------------------
// Two of the run states are not nearest-neighbour matters: the phase of a digit run (`\p{N}{1,3}` counts from the run's
// first digit) flows through rows that are all digits, and "is there a CR/LF further on in this white-space run"
// (`\s*[\r\n]+`) flows backwards through rows that are all white space.  Inside the block both are resolved by a scan over
// the lanes of (propagate | generate) functions; what flows in from beyond the block -- only when the block's first row is
// all digits or its last row all white space -- is found by a wavefront-wide search over the rows outside (64 rows per step).
struct TkzBlockCtx { const uint8_t* bytes; const uint64_t* docbits; int64_t total, nrows, row0; };   // row0 = row of lane 0

// class masks of corpus row `row` (any row index; bytes outside the corpus read as 0 = "other")
TKZ_DEV TkzBlockMasks tkz_corpus_row_masks(const TkzBlockCtx& X, int64_t row) {
    uint4 v[4];
    const int64_t pos = row << 6;
    if (row >= 0 && pos + 64 <= X.total) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = tkz_load16(X.bytes + pos + 16 * q);
    } else {
        uint32_t w[16];
        for (int k = 0; k < 16; ++k) {
            w[k] = 0;
            for (int j = 0; j < 4; ++j) { const int64_t q = pos + 4 * k + j; if (row >= 0 && q < X.total) w[k] |= (uint32_t)X.bytes[q] << (8 * j); }
        }
        for (int q = 0; q < 4; ++q) { v[q].x = w[4 * q]; v[q].y = w[4 * q + 1]; v[q].z = w[4 * q + 2]; v[q].w = w[4 * q + 3]; }
    }
    return tkz_block_classify<false>(v);
}
// digits of the run that ends right before row X.row0, mod 3 (0 when the char before is not a digit); -1 when the run is
// preceded by a multi-byte char (which may itself be a digit: the caller leaves the block to the sequential path).  Wave-uniform.
TKZ_DEV int tkz_digits_before(const TkzBlockCtx& X) {
    const int lane = simt::lane();
    int64_t count = 0;
    for (int64_t base = X.row0 - 1;; base -= 64) {
        const int64_t row = base - lane;
        int tail = 0, unknown = 0;
        bool full = false;
        if (row >= 0) {
            const TkzBlockMasks m = tkz_corpus_row_masks(X, row);
            const uint64_t ds = X.docbits[row];
            const uint64_t notN = ~(m.N & ~m.HI);
            tail = notN ? 63 - tkz_msb64(notN) : 64;                          // ASCII digits at the end of the row
            const bool capped = ds != 0 && tail >= 64 - tkz_msb64(ds);         // a document start begins a new run
            if (capped) tail = 64 - tkz_msb64(ds);
            else if (tail < 64 && ((m.HI >> (63 - tail)) & 1ull)) unknown = 1;   // stopped by a multi-byte char
            full = tail == 64 && ds == 0;
        }
        const uint64_t f = simt::ballot(full);
        const int idx = (~f) ? tkz_ctz64(~f) : 64;
        const int t = simt::shfl(tail, idx & 63), u = simt::shfl(unknown, idx & 63);
        if (idx < 64 && u) return -1;
        if (idx < 64) { count += 64 * (int64_t)idx + t; break; }
        count += 64 * 64;
    }
    return (int)(count % 3);
}
// does the connected white-space run that continues past the block's last row contain a CR/LF?  1 / 0; -1 when multi-byte
// text is met before the run ends (the caller leaves the block to the sequential path).  Wave-uniform.
TKZ_DEV int tkz_crlf_ahead(const TkzBlockCtx& X) {
    const int lane = simt::lane();
    for (int64_t base = X.row0 + 64;; base += 64) {
        const int64_t row = base + lane;
        bool full = false, hi = false, crlead = false, crany = false;
        if (row < X.nrows) {
            const TkzBlockMasks m = tkz_corpus_row_masks(X, row);
            const uint64_t conn = m.W & ~X.docbits[row];
            hi = m.hi != 0;
            full = !hi && conn == ~0ull;
            const int lead = (~conn) ? tkz_ctz64(~conn) : 64;
            crlead = (m.CR & tkz_lowmask(lead)) != 0;
            crany = m.CR != 0;
        }
        const uint64_t f = simt::ballot(full);
        const int idx = (~f) ? tkz_ctz64(~f) : 64;
        const uint64_t crs = simt::ballot(full && crany) & tkz_lowmask(idx);
        const int hi_at = simt::shfl(hi ? 1 : 0, idx & 63), cr_at = simt::shfl(crlead ? 1 : 0, idx & 63);
        if (crs) return 1;
        if (idx < 64) return hi_at ? -1 : cr_at;
    }
}
// digit phase: every lane contributes f(x) = prop ? (x + v) % 3 : v; returns the phase flowing INTO this lane's row (lane 0: in0)
TKZ_DEV int tkz_scan_phase(bool prop, int v, int in0) {
    const int lane = simt::lane();
    int p = prop ? 1 : 0, val = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int pl = simt::shfl(p, (lane - d) & 63), vl = simt::shfl(val, (lane - d) & 63);
        if (lane >= d && p) { val = (vl + val) % 3; p = pl; }
    }
    const int out = p ? (in0 + val) % 3 : val;
    const int prev = simt::shfl(out, (lane + 63) & 63);
    return lane == 0 ? in0 : prev;
}
// CR/LF ahead: every lane contributes f(x) = prop ? (v | x) : v, composed from the top lane down; returns the value of the
// row AFTER this lane's (lane 63: in64)
TKZ_DEV uint32_t tkz_scan_head(bool prop, uint32_t v, uint32_t in64) {
    const int lane = simt::lane();
    int p = prop ? 1 : 0;
    uint32_t val = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int pr = simt::shfl(p, (lane + d) & 63);
        const uint32_t vr = simt::shflu(val, (lane + d) & 63);
        if (lane + d < 64 && p) { val |= vr; p = pr; }
    }
    const uint32_t out = p ? (val | in64) : val;
    const uint32_t next = simt::shflu(out, (lane + 1) & 63);
    return lane == 63 ? in64 : next;
}

// one bit per CHAR of the lane's row (n chars, bits >= n clear): classes, raw contraction candidates ('x followed by a
// literal of 2 / 3 chars, conditions not yet applied), document starts
struct TkzCharMasks { uint64_t L, N, O, O2, W, CR, SP, k2, k3, ds; int n; bool ascii; };   // ascii: the row has no multi-byte char

// The rules on char masks; every lane of the wavefront calls it (two lane-shift exchanges inside).  Returns false when
// the nearest-neighbour scheme cannot do the block: a digit run or a white-space run covering a whole row.
template <int PATTERN>
TKZ_DEV bool tkz_block_core(const TkzCharMasks& c, uint64_t* start_out) {
    const int lane = simt::lane();
    const int n = c.n, top = n - 1;
    const uint64_t all = tkz_lowmask(n);
    const uint64_t L = c.L, N = c.N, O = c.O, W = c.W, CR = c.CR, SP = c.SP, ds = c.ds;
    const uint64_t nds = ~ds;
    // ---- exchange 1: the class of the char before my row (the last char of the previous lane's row), the first chars of the next row
    const uint32_t up_bits = (uint32_t)((L >> top) & 1) | ((uint32_t)((N >> top) & 1) << 1) | ((uint32_t)((O >> top) & 1) << 2) |
                             ((uint32_t)((SP >> top) & 1) << 3) | ((uint32_t)((W >> top) & 1) << 4) | ((uint32_t)((CR >> top) & 1) << 5);
    // head: a CR/LF inside the leading connected white-space run of my row (what the row above needs for `\s*[\r\n]+`)
    const uint64_t conn = W & nds;
    const int lead_ws = (~conn) ? tkz_ctz64(~conn) : 64;
    const uint32_t head = (CR & tkz_lowmask(lead_ws)) ? 1u : 0u;
    const uint32_t dn_bits = (uint32_t)(L & 1) | ((uint32_t)(N & 1) << 1) | ((uint32_t)(O & 1) << 2) | ((uint32_t)(ds & 3) << 3) | (head << 5);
    uint32_t pb = simt::shflu(up_bits, (lane + 63) & 63), nb = simt::shflu(dn_bits, (lane + 1) & 63);
    if (lane == 0) pb = 0;
    if (lane == 63) nb = 0;
    const uint64_t pN = ((N << 1) | ((pb >> 1) & 1)) & nds & all;
    const uint64_t Q = N & pN;
    // ---- the two states that can cross whole rows (pattern 1 has neither) ----
    int carry_in = 0;                                      // digits of the open run before my row, mod 3
    uint32_t head_next = (nb >> 5) & 1;                    // a CR/LF inside the leading connected white-space run of the next row
    if (PATTERN != TKZ_PAT_P1) {
        // a row of CR/LF only: the absorbed state would cross it -- left to the sequential path
        if (simt::ballot((CR & nds & all) == all)) return false;
        // (lane 0 is context and cannot see the row before it: only an all-digit row needs to know what came before)
        const bool first_all = lane == 0 && N == all && ds == 0;
        const bool propN = Q == all || first_all, propW = conn == all;
        int gen = 0;
        if (!propN && ((N >> top) & 1)) gen = (n - tkz_msb64(~Q & all)) % 3;     // the open run started inside my row
        if (simt::ballot(propN)) {
            int in0 = 0;
            // (what flows in from beyond the block is looked up only by the o200k scanner, which has no other fast path; here
            //  the row-sequential evaluator takes such a block, and the searches' registers stay out of this kernel)
            if (simt::ballot(first_all)) return false;
            carry_in = tkz_scan_phase(propN, propN ? n % 3 : gen, in0);
        } else {
            carry_in = simt::shfl(gen, (lane + 63) & 63);
            if (lane == 0) carry_in = 0;
        }
        if (simt::ballot(propW)) {
            uint32_t in64 = 0;
            if (simt::ballot(lane == 63 && propW)) return false;
            head_next = tkz_scan_head(propW, propW ? ((CR & all) ? 1u : 0u) : head, in64);
        }
    }
    const uint64_t pL = ((L << 1) | (pb & 1)) & nds, pO = ((O << 1) | ((pb >> 2) & 1)) & nds, pSP = ((SP << 1) | ((pb >> 3) & 1)) & nds;
    const uint64_t pW = ((W << 1) | ((pb >> 4) & 1)) & nds, pCR = ((CR << 1) | ((pb >> 5) & 1)) & nds;
    const uint64_t dsn = (uint64_t)((nb >> 3) & 3);                   // document-start bits of chars 0,1 of the next row
    const uint64_t KN = (ds >> 1) | (dsn << top), KN2 = (ds >> 2) | (dsn << (top - 1));
    const uint64_t nO = ((O >> 1) | ((uint64_t)((nb >> 2) & 1) << top)) & ~KN;
    const uint64_t nReal = (((L | N | O) >> 1) | ((uint64_t)((nb & 7) ? 1 : 0) << top)) & ~KN;
    // ---- contractions: an apostrophe that is a match start, followed by a literal inside the document
    const uint64_t apOk = ~pO & ~pSP & ~KN;
    const uint64_t c2 = c.k2 & apOk, c3 = c.k3 & apOk & ~KN2;
    const uint64_t o1 = (PATTERN == TKZ_PAT_P1) ? 0ull : (O & ~c.O2 & ~pO & ~pSP & ~nO);     // (a two-unit char is never the one-unit prefix)
    // local carries out of my row (exact because no run covers a whole row)
    uint32_t abs_out = 0;
    uint64_t ABS0 = 0, R = 0, seeds = 0;
    if (PATTERN != TKZ_PAT_P1) {
        R = CR & nds; seeds = CR & pO;
        ABS0 = tkz_fill_up64(seeds & R, R);
        abs_out = (uint32_t)((ABS0 >> top) & 1);
    }
    // ---- exchange 2: contraction ends, o1, absorbed state and digit phase flowing in from the previous row
    // (a literal that ends k chars past my last char ends at char k - 1 of the next row)
    const uint32_t c2out = (uint32_t)((c2 >> (n - 2)) & 3), c3out = (uint32_t)((c3 >> (n - 3)) & 7);      // (n >= 16: a char is at most 4 bytes)
    const uint32_t up2 = c2out | (c3out << 2) | ((uint32_t)((o1 >> top) & 1) << 5) | (abs_out << 6) |
                         ((uint32_t)(((c2 | c3) >> top) & 1) << 9);
    uint32_t p2b = simt::shflu(up2, (lane + 63) & 63);
    if (lane == 0) p2b = 0;
    const uint64_t contrEnd = (c2 << 2) | (c3 << 3) | (uint64_t)(p2b & 3) | (uint64_t)((p2b >> 2) & 7);
    uint64_t start;
    if (PATTERN == TKZ_PAT_P1) {
        const uint64_t clenPrev = ((c2 | c3) << 1) | (uint64_t)((p2b >> 9) & 1);
        start = (L & ~pL & ~pSP & ~clenPrev) | (N & ~pN & ~pSP) | (O & ~pO & ~pSP) | (W & (~pW | nReal));
    } else {
        const uint64_t o1Prev = (o1 << 1) | (uint64_t)((p2b >> 5) & 1);
        const uint64_t pWSo = pW & ~pCR & ~pSP;
        const uint64_t sL = L & ~pL & ~pSP & ~pWSo & ~o1Prev;
        const uint64_t sO = O & ~pO & ~pSP;
        // \p{N}{1,3}
        uint64_t S = N & ~pN;
        if (Q & 1ull) {
            const int d = (3 - carry_in) % 3, lead = (~Q) ? tkz_ctz64(~Q) : 64;
            if (d < lead) S |= 1ull << d;
        }
        const uint64_t Q3 = Q & (Q << 1) & (Q << 2);
        uint64_t T = S | ((S << 3) & Q3);
        const uint64_t Q6 = Q3 & (Q3 << 3);
        T |= (T << 6) & Q6;
        const uint64_t Q12 = Q6 & (Q6 << 6);
        T |= (T << 12) & Q12;
        const uint64_t Q24 = Q12 & (Q12 << 12);
        T |= (T << 24) & Q24;
        T |= (T << 48) & (Q24 & (Q24 << 24));
        const uint64_t sN = T;
        // white space
        const uint32_t abs_in = (p2b >> 6) & 1;
        const uint64_t ABS = abs_in ? tkz_fill_up64((seeds | (R & 1ull)) & R, R) : ABS0;
        // T(i) = CR(i) | (CONN(i+1) & T(i+1)),  T(n) = head of the next row (which includes CONN(n))
        const uint64_t Scr = CR | ((uint64_t)head_next << top);
        const uint64_t Srev = tkz_brev64(Scr), Grev = tkz_brev64(conn) << 1;   // reversed positions k = 63 - i: G'(k) = CONN(64 - k)
        const uint64_t Tcur = tkz_brev64(Srev | tkz_fill_up64((Srev << 1) & Grev, Grev));
        const uint64_t pABS = (ABS << 1) | (uint64_t)abs_in;
        const uint64_t sW = W & ~ABS & ((~pW | pABS) | (pCR & ~Tcur) | (~CR & nReal));
        start = sL | sN | sO | sW;
    }
    *start_out = (start | contrEnd | ds) & all;
    return true;
}

// raw contraction candidates at BYTE positions: an apostrophe followed by a literal (the bytes may sit in the next row)
template <int PATTERN>
TKZ_DEV void tkz_block_contractions(const uint8_t* stage, uint64_t AP, uint64_t* k2, uint64_t* k3) {
    const int lane = simt::lane();
    uint64_t c2 = 0, c3 = 0;
    for (uint64_t ap = AP; ap; ap &= ap - 1) {
        const int pos = tkz_ctz64(ap);
        const int p1 = pos + 1, p2 = pos + 2;
        const uint32_t b1 = stage[(lane + (p1 >> 6)) * kBlockRowStride + (p1 & 63)];
        const uint32_t b2 = stage[(lane + (p2 >> 6)) * kBlockRowStride + (p2 & 63)];
        const int k = tkz_contraction_len(b1, b2, PATTERN == TKZ_PAT_CL100K);
        if (k == 2) c2 |= 1ull << pos;
        else if (k == 3) c3 |= 1ull << pos;
    }
    *k2 = c2; *k3 = c3;
}

// Evaluates rows first_row .. first_row+63 (lane = row - first_row) staged at `stage` (kBlockRowStride bytes per
// row, one extra row of zeros after the last).  Returns false -- and leaves `out` unset -- when the block has to be
// done by the sequential row loop (a run covering a whole row, malformed UTF-8, a document that starts inside a char);
// otherwise *out is this lane's piece-start word (valid for lanes 1..62).
template <int PATTERN>
TKZ_DEV bool tkz_block_eval(const uint8_t* stage, uint64_t ds, const uint8_t* bmp, uint64_t* out) {
    const int lane = simt::lane();
    const uint8_t* myrow = stage + lane * kBlockRowStride;
    const TkzBlockMasks m = tkz_block_classify<false>(reinterpret_cast<const uint4*>(myrow));
    TkzCharMasks c;
    if (simt::ballot(m.hi != 0) == 0) {                    // ---- an ASCII block: bytes are chars ----
        c.L = m.L; c.N = m.N; c.O = m.O; c.O2 = 0; c.W = m.W; c.CR = m.CR; c.SP = m.SP; c.ds = ds; c.n = 64; c.ascii = true;
        tkz_block_contractions<PATTERN>(stage, m.AP, &c.k2, &c.k3);
        uint64_t start;
        if (!tkz_block_core<PATTERN>(c, &start)) return false;
        *out = start;
        return true;
    }
    // ---- multi-byte chars: the ASCII classes are valid on the ASCII bytes only; lead / continuation structure ----
    const uint64_t HI = m.HI, CONT = m.CONT;
    const uint64_t A = ~HI;                                // ASCII bytes
    uint64_t L = m.L & A, N = m.N & A, W = m.W & A, O1 = ~(m.L | m.N | m.W) & A, O2 = 0;
    const uint64_t CR = m.CR & A, SP = m.SP & A;
    int bad = 0;
    uint64_t E = 0;                                        // where continuation bytes are expected, from the leads of my row
    uint32_t spill = 0;                                    // ... and in the first three bytes of the next row
    for (uint64_t t = HI & ~CONT; t; t &= t - 1) {         // every non-ASCII lead of my row: decode, class from the table
        const int pos = tkz_ctz64(t);
        uint32_t b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int q = pos + k; b[k] = stage[(lane + (q >> 6)) * kBlockRowStride + (q & 63)]; }
        const TkzChar ch = tkz_decode(b[0], b[1], b[2], b[3], bmp);
        bad |= ch.bad;
        const int pc = tkz_pc_of(ch);
        const uint64_t bit = 1ull << pos;
        if (pc == PC_L) L |= bit; else if (pc == PC_N) N |= bit; else if (pc == PC_WS) W |= bit; else if (pc == PC_O2) O2 |= bit; else O1 |= bit;
        for (int k = 1; k < ch.len; ++k) { const int q = pos + k; if (q < 64) E |= 1ull << q; else spill |= 1u << (q - 64); }
    }
    uint32_t spill_in = simt::shflu(spill, (lane + 63) & 63);
    if (lane == 0) { const int lc = (~CONT) ? tkz_ctz64(~CONT) : 64; spill_in = lc <= 3 ? (uint32_t)tkz_lowmask(lc) : 0u; }   // lane 0 is context: its leading continuation bytes belong to a char of the row before
    E |= spill_in;
    // malformed text and a document that starts inside a char are reported by the sequential row loop
    if (simt::ballot(bad != 0 || E != CONT || (ds & CONT) != 0)) return false;
    const uint64_t LEAD = ~CONT;
    const TkzPext px = tkz_pext_prepare(LEAD);
    uint64_t k2b, k3b;
    tkz_block_contractions<PATTERN>(stage, m.AP & A, &k2b, &k3b);
    c.n = tkz_popc64(LEAD);
    c.L = tkz_pext(L, LEAD, px); c.N = tkz_pext(N, LEAD, px); c.W = tkz_pext(W, LEAD, px);
    c.O2 = tkz_pext(O2, LEAD, px); c.O = tkz_pext(O1, LEAD, px) | c.O2;
    c.CR = tkz_pext(CR, LEAD, px); c.SP = tkz_pext(SP, LEAD, px);
    c.k2 = tkz_pext(k2b, LEAD, px); c.k3 = tkz_pext(k3b, LEAD, px);
    c.ds = tkz_pext(ds, LEAD, px);
    c.ascii = HI == 0;
    uint64_t start;
    if (!tkz_block_core<PATTERN>(c, &start)) return false;
    *out = tkz_pdep(start, LEAD, px) | ds;
    return true;
}

// The same for o200k (tokenizer_ts/src/tokenizerBuilder.ts:79-89) on ASCII rows.  Without \p{Lm}\p{Lo}\p{M} the two word
// alternatives reduce to U*l* (upper-case run, then lower-case run): inside a letter run a piece starts exactly at a
// lower->upper transition; the one-char prefix rule is cl100k's; a contraction is not an alternative of its own but an
// optional SUFFIX of a word piece (the apostrophe is glued when the char before it ends a word piece -- not when that
// char is itself the tail of a glued contraction: `it's's` = `it's` + `'s`, resolved by a short fixed-point iteration);
// ` ?[^\s\p{L}\p{N}]+[\r\n/]*` also absorbs '/' after the CR/LF it absorbed.  Refuses (returns false) what the
// nearest-neighbour scheme cannot do, as tkz_block_eval does.
TKZ_DEV bool tkz_block_eval_o200k(const uint8_t* stage, uint64_t ds, const TkzBlockCtx& X, uint64_t* out) {
    const int lane = simt::lane();
    const TkzBlockMasks m = tkz_block_classify<true>(reinterpret_cast<const uint4*>(stage + lane * kBlockRowStride));
    const uint64_t L = m.L, N = m.N, O = m.O, W = m.W, CR = m.CR, SP = m.SP, AP = m.AP, UP = m.UP, LOW = m.L & ~m.UP;
    const uint64_t nds = ~ds;
    const uint32_t up_bits = (uint32_t)(L >> 63) | ((uint32_t)(N >> 63) << 1) | ((uint32_t)(O >> 63) << 2) | ((uint32_t)(SP >> 63) << 3) |
                             ((uint32_t)(W >> 63) << 4) | ((uint32_t)(CR >> 63) << 5) | ((uint32_t)(LOW >> 63) << 6);
    const uint64_t conn = W & nds;
    const int lead_ws = (~conn) ? tkz_ctz64(~conn) : 64;
    const uint32_t head = (CR & tkz_lowmask(lead_ws)) ? 1u : 0u;
    const uint32_t dn_bits = (uint32_t)(L & 1) | ((uint32_t)(N & 1) << 1) | ((uint32_t)(O & 1) << 2) | ((uint32_t)(ds & 3) << 3) | (head << 5);
    uint32_t pb = simt::shflu(up_bits, (lane + 63) & 63), nb = simt::shflu(dn_bits, (lane + 1) & 63);
    if (lane == 0) pb = 0;
    if (lane == 63) nb = 0;
    const uint64_t pN = ((N << 1) | ((pb >> 1) & 1)) & nds;
    const uint64_t Q = N & pN;
    const uint64_t R = (CR | m.SL) & nds;                              // what `[\r\n/]*` can run through
    const bool bad = m.hi != 0 || (R & tkz_lowmask(60)) == tkz_lowmask(60);
    if (simt::ballot(bad)) return false;
    // the two states that can cross whole rows: digit phase, CR/LF further on in the white-space run (see tkz_block_core)
    int carry_in;
    uint32_t head_next = (nb >> 5) & 1;
    {
        const bool first_all = lane == 0 && N == ~0ull && ds == 0;
        const bool propN = Q == ~0ull || first_all, propW = conn == ~0ull;
        int gen = 0;
        if (!propN && (N >> 63)) gen = (64 - tkz_msb64(~Q)) % 3;
        if (simt::ballot(propN)) {
            int in0 = 0;
            if (simt::ballot(first_all)) { in0 = tkz_digits_before(X); if (in0 < 0) return false; }
            carry_in = tkz_scan_phase(propN, propN ? 64 % 3 : gen, in0);
        } else {
            carry_in = simt::shfl(gen, (lane + 63) & 63);
            if (lane == 0) carry_in = 0;
        }
        if (simt::ballot(propW)) {
            uint32_t in64 = 0;
            if (simt::ballot(lane == 63 && propW)) { const int a = tkz_crlf_ahead(X); if (a < 0) return false; in64 = (uint32_t)a; }
            head_next = tkz_scan_head(propW, propW ? (CR ? 1u : 0u) : head, in64);
        }
    }
    const uint64_t pL = ((L << 1) | (pb & 1)) & nds, pO = ((O << 1) | ((pb >> 2) & 1)) & nds, pSP = ((SP << 1) | ((pb >> 3) & 1)) & nds;
    const uint64_t pW = ((W << 1) | ((pb >> 4) & 1)) & nds, pCR = ((CR << 1) | ((pb >> 5) & 1)) & nds;
    const uint64_t pLOW = ((LOW << 1) | ((pb >> 6) & 1)) & nds;
    const uint64_t dsn = (uint64_t)((nb >> 3) & 3);
    const uint64_t KN = (ds >> 1) | (dsn << 63), KN2 = (ds >> 2) | (dsn << 62);
    const uint64_t nO = ((O >> 1) | ((uint64_t)((nb >> 2) & 1) << 63)) & ~KN;
    const uint64_t nReal = (((L | N | O) >> 1) | ((uint64_t)((nb & 7) ? 1 : 0) << 63)) & ~KN;
    // ---- contraction candidates: an apostrophe right after a letter, followed by a literal inside the document
    uint64_t k2 = 0, k3 = 0;
    {
        uint64_t ap = AP & pL & ~KN;
        while (simt::ballot(ap != 0)) {
            if (ap) {
                const int pos = tkz_ctz64(ap);
                ap &= ap - 1;
                const int p1 = pos + 1, p2 = pos + 2;
                const uint32_t b1 = stage[(lane + (p1 >> 6)) * kBlockRowStride + (p1 & 63)];
                const uint32_t b2 = stage[(lane + (p2 >> 6)) * kBlockRowStride + (p2 & 63)];
                const int k = tkz_contraction_len_o200k(b1, b2);
                if (k == 2) k2 |= 1ull << pos;
                else if (k == 3 && !((KN2 >> pos) & 1ull)) k3 |= 1ull << pos;
            }
        }
    }
    // glued = candidates whose preceding letter is not the tail of a glued contraction (fixed point, left to right)
    auto resolve = [&](uint64_t blk_in, uint64_t* g2o, uint64_t* g3o) -> bool {
        uint64_t g2 = k2, g3 = k3, blk = blk_in;
        for (int it = 0; it < 6; ++it) {
            g2 = k2 & ~blk; g3 = k3 & ~blk;
            const uint64_t nblk = (g2 << 2) | (g3 << 3) | blk_in;
            if (nblk == blk) { *g2o = g2; *g3o = g3; return true; }
            blk = nblk;
        }
        *g2o = g2; *g3o = g3;
        return false;
    };
    uint64_t g2, g3;
    bool conv = resolve(0, &g2, &g3);
    // local state that flows to the next row (computed with no inflow: exact unless a run covers the row, refused above)
    const uint64_t seeds = CR & pO;
    const uint64_t ABS0 = tkz_fill_up64(seeds & R, R);
    const uint64_t o1_0 = O & ~ABS0 & (~pO | (ABS0 << 1)) & ~pSP & ~nO & ~(g2 | g3);
    const uint32_t up2 = (uint32_t)(g2 >> 62) | ((uint32_t)(g3 >> 61) << 2) | ((uint32_t)(o1_0 >> 63) << 5) | ((uint32_t)(ABS0 >> 63) << 6);
    uint32_t p2b = simt::shflu(up2, (lane + 63) & 63);
    if (lane == 0) p2b = 0;
    // with the real inflow: positions 0..2 may be the fresh start after a contraction of the previous row
    const uint64_t blk_in = (uint64_t)(p2b & 3) | (uint64_t)((p2b >> 2) & 7);
    uint64_t h2, h3;
    conv = resolve(blk_in, &h2, &h3) && conv;
    // if the inflow changed what I told the next row, that row worked from wrong data: leave the block to the sequential path
    if (simt::ballot(!conv || (h2 >> 62) != (g2 >> 62) || (h3 >> 61) != (g3 >> 61))) return false;
    g2 = h2; g3 = h3;
    const uint64_t glued = g2 | g3;
    const uint64_t contrEnd = (g2 << 2) | (g3 << 3) | blk_in;
    // letters covered by a glued literal (never piece starts): from this row and from the previous one
    const uint32_t c2p = p2b & 3, c3p = (p2b >> 2) & 7;                 // previous row: g2 bits 62,63 ; g3 bits 61,62,63
    const uint64_t cover = (g2 << 1) | (g3 << 1) | (g3 << 2) |
                           (uint64_t)(((c2p >> 1) | (c3p >> 1) | (c3p >> 2)) & 1) | ((uint64_t)((c3p >> 2) & 1) << 1);
    const uint32_t abs_in = (p2b >> 6) & 1;
    const uint64_t ABS = abs_in ? tkz_fill_up64((seeds | (R & 1ull)) & R, R) : ABS0;
    const uint64_t pABS = (ABS << 1) | (uint64_t)abs_in;
    const uint64_t o1 = O & ~ABS & (~pO | pABS) & ~pSP & ~nO & ~glued;
    const uint64_t o1Prev = (o1 << 1) | (uint64_t)((p2b >> 5) & 1);
    const uint64_t pWSo = pW & ~pCR & ~pSP;
    const uint64_t sL = (L & ~pL & ~pSP & ~pWSo & ~o1Prev) | (UP & pLOW);
    const uint64_t sO = O & ~ABS & (~pO | pABS) & ~pSP;
    uint64_t S = N & ~pN;
    if (Q & 1ull) {
        const int d = (3 - carry_in) % 3, lead = (~Q) ? tkz_ctz64(~Q) : 64;
        if (d < lead) S |= 1ull << d;
    }
    const uint64_t Q3 = Q & (Q << 1) & (Q << 2);
    uint64_t T = S | ((S << 3) & Q3);
    const uint64_t Q6 = Q3 & (Q3 << 3);
    T |= (T << 6) & Q6;
    const uint64_t Q12 = Q6 & (Q6 << 6);
    T |= (T << 12) & Q12;
    const uint64_t Q24 = Q12 & (Q12 << 12);
    T |= (T << 24) & Q24;
    T |= (T << 48) & (Q24 & (Q24 << 24));
    const uint64_t Scr = CR | ((uint64_t)head_next << 63);
    const uint64_t Srev = tkz_brev64(Scr), Grev = tkz_brev64(conn) << 1;
    const uint64_t Tcur = tkz_brev64(Srev | tkz_fill_up64((Srev << 1) & Grev, Grev));
    const uint64_t sW = W & ~ABS & ((~pW | pABS) | (pCR & ~Tcur) | (~CR & nReal));
    *out = ((sL | T | sO | sW) & ~cover & ~glued) | contrEnd | ds;
    return true;
}

// =================================================================================================
// (2c) o200k on rows WITH multi-byte chars (CJK, kana, hangul, emoji, combining marks ...).
//
// Char-level form of tkz_block_eval_o200k.  With \p{Lm}\p{Lo}\p{M} in BOTH word alternatives
//     alt 1  [^\r\n\p{L}\p{N}]? A* B+ suffix?      A = [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]
//     alt 2  [^\r\n\p{L}\p{N}]? A+ B* suffix?      B = [\p{Ll}\p{Lm}\p{Lo}\p{M}]
// a run of word chars (letters and marks) is cut by a two-state automaton.  Classes: U = Lu|Lt (A only), l = Ll (B only),
// Y = Lm|Lo|M (both).  State S0 = inside A*, S1 = inside B+.  S0: U,Y stay, l -> S1.  S1: l,Y stay, U -> a new piece.  So
//   (a) an U starts a piece when the nearest preceding non-Y char of its run is an l  -- a forward flow through runs of Y;
//   (b) when the A run is NOT followed by an l, A* has to give chars back until B+ can match: the piece ends right after the
//       LAST Y of the run and the U's behind it are a piece of their own (alt 2): an U whose predecessor is a Y in state S0
//       starts a piece when only U's follow up to the end of the word run -- a backward flow through runs of U.
// \p{M} is also in the class of ` ?[^\s\p{L}\p{N}]+[\r\n/]*` ("R4" below): a mark is part of such a piece when the char before
// it is, of a word otherwise.  An O char (other, not a mark) opens an R4 piece when it follows a blank or is not followed by a
// word char (then it would be the one-char prefix of the word), and every O / mark behind one that is in R4 is in R4: a
// forward flow through runs of O|M.  The tail `[\r\n/]*` (ABS) follows an R4 char; a '/' swallowed by it is NOT an R4 char
// for what follows (`;\n/*`: the '*' stands on its own again), while the tail itself begins behind an R4 char: the two flows
// depend on each other from left to right and are iterated (tkz_block_core_o200k), both followed through whole rows.
// Everything else (digits, white space, contraction suffixes) is tkz_block_eval_o200k's algebra on n <= 64 chars.
// Refused blocks (return false) are matched sequentially; that is the definition, so refusing is always safe.
// =================================================================================================
// forward flow over the lanes: every lane contributes f(x) = prop ? x : gen; returns the value flowing INTO this lane (lane 0: in0)
TKZ_DEV uint32_t tkz_scan_flow(bool prop, uint32_t gen, uint32_t in0) {
    const int lane = simt::lane();
    int p = prop ? 1 : 0;
    uint32_t val = prop ? 0u : gen;
    for (int d = 1; d < 64; d <<= 1) {
        const int pl = simt::shfl(p, (lane - d) & 63);
        const uint32_t vl = simt::shflu(val, (lane - d) & 63);
        if (lane >= d && p) { val = vl; p = pl; }
    }
    const uint32_t outv = p ? in0 : val;
    const uint32_t prev = simt::shflu(outv, (lane + 63) & 63);
    return lane == 0 ? in0 : prev;
}

struct TkzO2Chars { uint64_t U, l, X, M, N, W, O, O2, CR, SP, SL, AP, k2, k3, ds; int n; };   // one bit per CHAR of the lane's row (O excludes marks; O2: the O chars of two UTF-16 units, .NET semantics only)

// (development aid of the CPU-emulated build: which rule refused a block, counted in tkz_o2_refusals[])
#ifdef TKZ_HOSTEMU
extern "C" long long tkz_o2_refusals[32];
inline bool tkz_o2_refuse(int why) { if (simt::lane() == 0) ++tkz_o2_refusals[why & 31]; return false; }
#else
TKZ_DEV bool tkz_o2_refuse(int) { return false; }
#endif
TKZ_DEV bool tkz_block_core_o200k(const TkzO2Chars& c, const TkzBlockCtx& X, uint64_t* start_out) {
    const int lane = simt::lane();
    const int n = c.n, top = n - 1;
    const uint64_t all = tkz_lowmask(n);
    const uint64_t U = c.U, l = c.l, Xl = c.X, M = c.M, N = c.N, W = c.W, O = c.O, CR = c.CR, SP = c.SP, SL = c.SL, AP = c.AP, ds = c.ds;
    const uint64_t nds = ~ds;
    const uint64_t Lw = U | l | Xl, Wd = Lw | M, Oc = O | M;
    auto bit = [](uint64_t x, int i) -> uint32_t { return (uint32_t)((x >> i) & 1ull); };
    // ---- exchange 1: class of the last char of the previous row, of the first char of the next row ----
    const uint32_t up_bits = bit(U, top) | (bit(l, top) << 1) | (bit(Xl, top) << 2) | (bit(M, top) << 3) | (bit(N, top) << 4) | (bit(O, top) << 5) |
                             (bit(SP, top) << 6) | (bit(W, top) << 7) | (bit(CR, top) << 8) | (bit(SL, top) << 9);
    const uint64_t conn = W & nds;
    const int lead_ws = tkz_ctz64z(~conn);
    const uint32_t head = (CR & tkz_lowmask(lead_ws)) ? 1u : 0u;
    const uint32_t dn_bits = bit(U, 0) | (bit(l, 0) << 1) | (bit(Xl, 0) << 2) | (bit(M, 0) << 3) | (bit(N, 0) << 4) | (bit(O, 0) << 5) | (bit(W, 0) << 6) |
                             ((uint32_t)(ds & 3) << 7) | (head << 9) | (bit(SL, 0) << 10);
    uint32_t pb = simt::shflu(up_bits, (lane + 63) & 63), nb = simt::shflu(dn_bits, (lane + 1) & 63);
    if (lane == 0) pb = 0;
    if (lane == 63) nb = 0;
    const uint64_t dsn = (uint64_t)((nb >> 7) & 3);
    const uint64_t KN = ((ds >> 1) | (dsn << top)) & all, KN2 = ((ds >> 2) | (dsn << (top - 1))) & all;
    const uint64_t pSP = ((SP << 1) | ((pb >> 6) & 1)) & nds & all;
    const uint64_t pW = ((W << 1) | ((pb >> 7) & 1)) & nds & all, pCR = ((CR << 1) | ((pb >> 8) & 1)) & nds & all;
    const uint64_t pN = ((N << 1) | ((pb >> 4) & 1)) & nds & all;
    const uint64_t nWd = ((Wd >> 1) | ((uint64_t)((nb & 15) ? 1 : 0) << top)) & ~KN & all;
    // ---- R4 and ABS: which O / mark chars belong to a ` ?[^\s\p{L}\p{N}]+` piece (T), which CR / LF / '/' its tail `[\r\n/]*`
    // swallows (ABS).  Left to right:   ABS(i) = Rabs(i) & (ABS(i-1) | (CR(i) & T(i-1)))      Rabs = CR | LF | '/'
    //                                   T(i)   = Oc(i) & ~ABS(i) & (gR4(i) | T(i-1))
    // -- a swallowed '/' is not an R4 char for what follows (`;\n/*`: the '*' is on its own again) while ABS needs T at its seeds.
    // Both are fills along runs, so they are iterated: round 1 ignores the cuts (exact unless a swallowed '/' is followed by an
    // O / mark that is not a '/'), every further round is exact one more such link to the right; kO2Rounds of them, then the block is
    // refused.  Both states cross rows (a row of nothing but O / marks, of nothing but CR / LF / '/'): lane scans. ----
    // (.NET: a char of two units is never the one-unit prefix of a word -- the \p{L} behind the prefix would have to match its low surrogate --,
    //  so it always belongs to a ` ?[^\s\p{L}\p{N}]+` piece)
    const uint64_t gR4all = O & (pSP | ~nWd | c.O2) & all;
    const uint64_t PRall = Oc & nds & all;
    const uint64_t Rabs = (CR | SL) & nds & all;
    const uint64_t nOcNS = (((Oc & ~SL) >> 1) | ((uint64_t)((((nb >> 3) & 1) | ((nb >> 5) & 1)) & (((nb >> 10) & 1) ^ 1)) << top)) & ~KN & all;
    constexpr int kO2Rounds = 4;
    {   // lane 0 is context: its leading run of O / marks that open no piece of their own is in R4 or not as the (unknown) char before the row
        // says; a CR / LF right behind that run followed by CR / LF / '/' up to the end of the row would carry the guess into row 1
        const int lead = tkz_ctz64z(~(PRall & ~gR4all));
        const bool carried = lane == 0 && lead > 0 && lead < n && bit(CR & Rabs, lead) && (Rabs >> lead) == (all >> lead);
        if (simt::ballot(carried)) return tkz_o2_refuse(12);
    }
    uint64_t T = 0, ABS = 0, seeds = 0;
    uint32_t cinR = 0, abs_in = 0;
    for (int round = 0;; ++round) {
        const uint64_t gR4 = gR4all & ~ABS, PR = PRall & ~ABS;          // (round 0: ABS = 0, no cuts)
        const bool propR = PR == all && gR4 == 0;
        const uint32_t genR = bit(tkz_fill_up64(gR4, PR | gR4), top);
        {
            const uint64_t pm = simt::ballot(propR);
            if (pm) {
                if (pm & 1ull) {                                   // lane 0 is context: what flows out of it is unknown
                    const int lp = tkz_ctz64z(~pm);
                    if (lp >= 2) return tkz_o2_refuse(1);
                    if (simt::ballot(lane == 1 && (PR & ~gR4 & 1ull))) return tkz_o2_refuse(2);
                }
                cinR = tkz_scan_flow(propR, genR, 0u);
            } else {
                cinR = simt::shflu(genR, (lane + 63) & 63);
                if (lane == 0) cinR = 0;
            }
        }
        T = tkz_fill_up64(gR4 | (cinR ? (PR & 1ull) : 0ull), PR | gR4);     // chars in R4
        seeds = CR & ((T << 1) | cinR) & nds & all;                       // a CR / LF right behind an R4 char opens the tail
        const uint32_t genA = bit(tkz_fill_up64(seeds & Rabs, Rabs), top);
        const bool propA = Rabs == all && !genA;                           // a row of nothing but CR / LF / '/' (and no tail of its own): the absorbed state crosses it
        {
            const uint64_t pm = simt::ballot(propA);
            if (pm) {
                if (pm & 1ull) return tkz_o2_refuse(3);             // lane 0 is context: what flows out of it is unknown
                abs_in = tkz_scan_flow(propA, genA, 0u);
            } else {
                abs_in = simt::shflu(genA, (lane + 63) & 63);
                if (lane == 0) abs_in = 0;
            }
        }
        const uint64_t ABSn = tkz_fill_up64((seeds | (abs_in ? (Rabs & 1ull) : 0ull)) & Rabs, Rabs);
        const bool changed = ABSn != ABS;
        ABS = ABSn;
        // (round 0: T ran through the swallowed '/' chars; that is wrong only where such a char is followed by an O / mark that is not a '/')
        if (round == 0 ? !simt::ballot((SL & ABS & nOcNS) != 0) : !simt::ballot(changed)) break;
        if (round == kO2Rounds) return tkz_o2_refuse(8);
    }
    const uint64_t ABS0 = ABS;
    const uint64_t pR4 = ((T << 1) | cinR) & nds & all;
    // ---- the two states that can cross whole rows: digit phase, CR/LF further on in the white-space run (tkz_block_core) ----
    const uint64_t Q = N & pN;
    int carry_in;
    uint32_t head_next = (nb >> 9) & 1;
    {
        const bool first_all = lane == 0 && N == all && ds == 0;
        const bool propN = Q == all || first_all, propW = conn == all;
        int gen = 0;
        if (!propN && bit(N, top)) gen = (n - tkz_msb64(~Q & all)) % 3;
        if (simt::ballot(propN)) {
            int in0 = 0;
            if (simt::ballot(first_all)) { in0 = tkz_digits_before(X); if (in0 < 0) return tkz_o2_refuse(4); }
            carry_in = tkz_scan_phase(propN, propN ? n % 3 : gen, in0);
        } else {
            carry_in = simt::shfl(gen, (lane + 63) & 63);
            if (lane == 0) carry_in = 0;
        }
        if (simt::ballot(propW)) {
            uint32_t in64 = 0;
            if (simt::ballot(lane == 63 && propW)) { const int a = tkz_crlf_ahead(X); if (a < 0) return tkz_o2_refuse(5); in64 = (uint32_t)a; }
            head_next = tkz_scan_head(propW, propW ? ((CR & all) ? 1u : 0u) : head, in64);
        }
    }
    // ---- contraction suffixes: an apostrophe right after the last char of a word piece, followed by a literal inside the document ----
    const uint32_t pbLw = (pb & 7) ? 1u : 0u, pbMw = (((pb >> 3) & 1) && !cinR) ? 1u : 0u;
    const uint64_t pLw = ((Lw << 1) | pbLw) & nds & all, pMw = (((M & ~T) << 1) | pbMw) & nds & all;
    const uint64_t cand = AP & (pLw | pMw) & ~KN;
    const uint64_t k2 = c.k2 & cand, k3 = c.k3 & cand & ~KN2;
    auto resolve = [&](uint64_t blk_in, uint64_t* g2o, uint64_t* g3o) -> bool {
        uint64_t g2 = k2, g3 = k3, blk = blk_in;
        for (int it = 0; it < 6; ++it) {
            g2 = k2 & ~blk; g3 = k3 & ~blk;
            const uint64_t nblk = (g2 << 2) | (g3 << 3) | blk_in;
            if (nblk == blk) { *g2o = g2; *g3o = g3; return true; }
            blk = nblk;
        }
        *g2o = g2; *g3o = g3;
        return false;
    };
    uint64_t g2, g3;
    bool conv = resolve(0, &g2, &g3);
    // ---- exchange 2: what flows into the next row (computed with no inflow: exact unless a run covers the row, refused above) ----
    const uint64_t o1_0 = O & ~T & ~(g2 | g3) & ~ABS0 & all;
    const uint32_t up2 = (uint32_t)((g2 >> (n - 2)) & 3) | ((uint32_t)((g3 >> (n - 3)) & 7) << 2) | (bit(o1_0, top) << 5) | (bit(ABS0, top) << 6);
    uint32_t p2b = simt::shflu(up2, (lane + 63) & 63);
    if (lane == 0) p2b = 0;
    const uint64_t blk_in = (uint64_t)(p2b & 3) | (uint64_t)((p2b >> 2) & 7);
    uint64_t h2, h3;
    conv = resolve(blk_in, &h2, &h3) && conv;
    if (simt::ballot(!conv || ((h2 >> (n - 2)) & 3) != ((g2 >> (n - 2)) & 3) || ((h3 >> (n - 3)) & 7) != ((g3 >> (n - 3)) & 7))) return tkz_o2_refuse(7);
    g2 = h2; g3 = h3;
    const uint64_t glued = g2 | g3;
    const uint64_t contrEnd = ((g2 << 2) | (g3 << 3) | blk_in) & all;
    const uint32_t c2p = p2b & 3, c3p = (p2b >> 2) & 7;
    const uint64_t cover = ((g2 << 1) | (g3 << 1) | (g3 << 2) |
                            (uint64_t)(((c2p >> 1) | (c3p >> 1) | (c3p >> 2)) & 1) | ((uint64_t)((c3p >> 2) & 1) << 1)) & all;
    const uint64_t pABS = ((ABS << 1) | (uint64_t)abs_in) & all;
    // ---- word runs: chars, run starts, the S1 flow (rule a) and the trailing-U flow (rule b) ----
    const uint64_t rs = (ds | contrEnd) & all;
    const uint64_t Up = U & ~cover, lp = l & ~cover, Yp = (Xl | (M & ~T)) & all, Wdp = Up | lp | Yp;
    const uint64_t Yc = Yp & ~rs;                               // a Y char that continues its run
    const uint64_t Rst = Yc | lp;
    const bool propS = Yc == all;
    const uint32_t genS = bit(tkz_fill_up64(lp, Rst), top);
    uint32_t cinS;
    {
        const uint64_t pm = simt::ballot(propS);
        if (pm) {
            if (pm & 1ull) {                                   // lane 0's state is unknown: it matters to the first U behind the Y run that carries it
                const int lp0 = tkz_ctz64z(~pm);               // first row that is not all Y
                const int leadY = tkz_ctz64z(~Yc);
                const bool hit = lane == lp0 && leadY < n && bit(Up, leadY) && !bit(rs, leadY);
                if (simt::ballot(hit)) return tkz_o2_refuse(9);
            }
            cinS = tkz_scan_flow(propS, genS, 0u);
        } else {
            cinS = simt::shflu(genS, (lane + 63) & 63);
            if (lane == 0) cinS = 0;
        }
    }
    const uint64_t st1 = tkz_fill_up64(lp | (cinS ? (Yc & 1ull) : 0ull), Rst);
    const uint64_t p_st1 = ((st1 << 1) | cinS) & ~rs & all;
    // backward: E(i) = U(i) & (the next char ends the word run | (the next char is an U of the same run & E(i+1)))
    const uint64_t Uc = Up & ~(rs & ~1ull);                     // leading-run test: U chars with no run start behind position 0
    const int leadU = tkz_ctz64z(~Uc);
    const bool propE = leadU >= n && !bit(rs, 0);               // nothing but upper-case letters of one run: the flow crosses the row
    uint32_t e_head;                                            // E of the virtual char behind the previous row's last one
    if (bit(rs, 0) || !bit(Wdp, 0)) e_head = 1;
    else if (propE) e_head = 0;                                 // (placeholder: the lane scan below passes the inflow through)
    else if (bit(Up, 0)) e_head = (!bit(Wdp, leadU) || bit(rs, leadU)) ? 1u : 0u;
    else e_head = 0;
    uint32_t et;
    if (simt::ballot(propE)) {
        if (simt::ballot(lane == 63 && propE)) return tkz_o2_refuse(10);   // lane 63 is context: what flows into it is unknown
        et = tkz_scan_head(propE, propE ? 0u : e_head, 1u);
    } else {
        et = simt::shflu(e_head, (lane + 1) & 63);
        if (lane == 63) et = 1;
    }
    const uint64_t below = tkz_lowmask(top);
    const uint64_t stopm = ((~Wdp | rs) >> 1) & below, contm = ((Up & ~rs) >> 1) & below;
    const uint64_t Sfull = (Up & stopm) | ((et || bit(KN, top)) ? (Up & (1ull << top)) : 0ull);
    const uint64_t Rfull = (Up & contm) | Sfull;
    const int rsh = 64 - n;
    const uint64_t Erev = tkz_fill_up64(tkz_brev64(Sfull) >> rsh, tkz_brev64(Rfull) >> rsh);
    const uint64_t E = (tkz_brev64(Erev) >> rsh) & all;
    const uint32_t pbYp = (((pb >> 2) & 1) || (((pb >> 3) & 1) && !cinR)) ? 1u : 0u;
    const uint64_t pYp = ((Yp << 1) | pbYp) & ~rs & all;
    const uint32_t pbWdp = ((pb & 7) || (((pb >> 3) & 1) && !cinR)) ? 1u : 0u;
    const uint64_t pWdp = ((Wdp << 1) | pbWdp) & ~rs & all;
    // ---- piece starts ----
    const uint64_t o1 = O & ~T & ~glued & ~ABS & all;
    const uint64_t o1Prev = ((o1 << 1) | (uint64_t)((p2b >> 5) & 1)) & all;
    const uint64_t pWSo = pW & ~pCR & ~pSP;
    const uint64_t sWd = Wdp & ~pWdp & ~pSP & ~pWSo & ~o1Prev;
    const uint64_t sa = Up & p_st1;
    const uint64_t sb = Up & E & pYp & ~p_st1 & ~rs;
    const uint64_t sO = O & ~ABS & ~pSP & ~pR4;
    uint64_t S = N & ~pN;
    if (Q & 1ull) {
        const int d = (3 - carry_in) % 3, lead = tkz_ctz64z(~Q);
        if (d < lead) S |= 1ull << d;
    }
    const uint64_t Q3 = Q & (Q << 1) & (Q << 2);
    uint64_t Tn = S | ((S << 3) & Q3);
    const uint64_t Q6 = Q3 & (Q3 << 3);
    Tn |= (Tn << 6) & Q6;
    const uint64_t Q12 = Q6 & (Q6 << 6);
    Tn |= (Tn << 12) & Q12;
    const uint64_t Q24 = Q12 & (Q12 << 12);
    Tn |= (Tn << 24) & Q24;
    Tn |= (Tn << 48) & (Q24 & (Q24 << 24));
    const uint64_t nReal = ((((~W) & all) >> 1) | ((uint64_t)((nb & 0x3F) ? 1 : 0) << top)) & ~KN & all;
    const uint64_t Scr = CR | ((uint64_t)head_next << top);
    const uint64_t Srev = tkz_brev64(Scr), Grev = tkz_brev64(conn) << 1;
    const uint64_t Tcur = tkz_brev64(Srev | tkz_fill_up64((Srev << 1) & Grev, Grev));
    const uint64_t sW = W & ~ABS & ((~pW | pABS) | (pCR & ~Tcur) | (~CR & nReal));
    *start_out = ((((sWd | sa | sb | Tn | sO | sW) & ~cover & ~glued) | contrEnd | ds)) & all;
    return true;
}

// Evaluates rows first_row .. first_row+63 (lane = row - first_row) staged at `stage` as tkz_block_eval does, for o200k, any UTF-8.
// by_code_point: the ECMAScript engine's view (tkz_char_to_code_point_semantics); otherwise .NET's (tkz_classes.h: a char of four bytes is two
// OTHER units whatever its Unicode class, \s as the table has it).
TKZ_DEV bool tkz_block_eval_o200k_mb(const uint8_t* stage, uint64_t dsb, const TkzBlockCtx& X, const uint8_t* ucd, bool by_code_point, uint64_t* out) {
    const int lane = simt::lane();
    const TkzBlockMasks m = tkz_block_classify<true>(reinterpret_cast<const uint4*>(stage + lane * kBlockRowStride));
    const uint64_t HI = m.HI, CONT = m.CONT, A = ~HI;
    uint64_t U = m.UP & A, l = m.L & ~m.UP & A, Xl = 0, M = 0, N = m.N & A, W = m.W & A, O = ~(m.L | m.N | m.W) & A;
    const uint64_t CR = m.CR & A, SP = m.SP & A, SL = m.SL & A, AP = m.AP & A;
    int bad = 0;
    uint64_t O2 = 0;
    uint64_t E = 0;                                        // where continuation bytes are expected, from the leads of my row
    uint32_t spill = 0;                                    // ... and in the first three bytes of the next row
    // every non-ASCII lead of my row: decode, class by code point from the table -- four chars per step, their gathers in flight
    // together (a row of CJK text has ~21 leads: the loop is a chain of table round trips otherwise)
    const uint32_t* srow = reinterpret_cast<const uint32_t*>(stage);
    for (uint64_t t = HI & ~CONT; t;) {
        int pos[4];
        uint32_t cp[4], cls[4];
        int len[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pos[j] = -1; cp[j] = 0; len[j] = 1;
            if (t) {
                pos[j] = tkz_ctz64(t);
                t &= t - 1;
                // the four bytes at `pos` (they may run into the next row): two aligned dwords of the staged block
                const int q = pos[j], r = lane + (q >> 6), o = q & 63;
                const int w0 = (r * kBlockRowStride + (o & ~3)) >> 2;
                const int q4 = q + 4 - (q & 3), r1 = lane + (q4 >> 6), o1 = q4 & 63;
                const int w1 = (r1 * kBlockRowStride + o1) >> 2;
                const uint32_t x = (uint32_t)((((uint64_t)srow[w1] << 32) | srow[w0]) >> (8 * (o & 3)));
                const TkzChar ch = tkz_decode_raw(x & 0xFFu, (x >> 8) & 0xFFu, (x >> 16) & 0xFFu, x >> 24);
                bad |= ch.bad;
                cp[j] = ch.bad ? 0u : ch.cp; len[j] = ch.len;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cls[j] = cp[j] < TKZ_UCD_DIRECT ? (uint32_t)ucd[cp[j]] : (uint32_t)tkz_supp_class(ucd, cp[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (pos[j] < 0) continue;
            uint32_t uc = cls[j];
            const uint64_t bitp = 1ull << pos[j];
            if (by_code_point) { if (cp[j] == 0xFEFFu) uc = UC_WS; else if (cp[j] == 0x85u) uc = UC_OTHER; }   // ECMAScript \s (tkz_char_to_code_point_semantics)
            else if (cp[j] >= 0x10000u) { uc = UC_OTHER; O2 |= bitp; }                                       // .NET: two Cs units
            if (uc == UC_LU || uc == UC_LT) U |= bitp;
            else if (uc == UC_LL) l |= bitp;
            else if (uc == UC_LM || uc == UC_LO) Xl |= bitp;
            else if (uc == UC_M) M |= bitp;
            else if (uc == UC_N) N |= bitp;
            else if (uc == UC_WS) W |= bitp;
            else O |= bitp;
            for (int k = 1; k < len[j]; ++k) { const int q = pos[j] + k; if (q < 64) E |= 1ull << q; else spill |= 1u << (q - 64); }
        }
    }
    uint32_t spill_in = simt::shflu(spill, (lane + 63) & 63);
    if (lane == 0) { const int lc = (~CONT) ? tkz_ctz64(~CONT) : 64; spill_in = lc <= 3 ? (uint32_t)tkz_lowmask(lc) : 0u; }
    E |= spill_in;
    if (simt::ballot(bad != 0 || E != CONT || (dsb & CONT) != 0)) return tkz_o2_refuse(11);   // (reported by the sequential path)
    // raw contraction candidates at BYTE positions: an apostrophe followed by a literal of the o200k list
    uint64_t k2b = 0, k3b = 0;
    for (uint64_t ap = AP; ap; ap &= ap - 1) {
        const int pos = tkz_ctz64(ap);
        const int p1 = pos + 1, p2 = pos + 2;
        const uint32_t b1 = stage[(lane + (p1 >> 6)) * kBlockRowStride + (p1 & 63)];
        const uint32_t b2 = stage[(lane + (p2 >> 6)) * kBlockRowStride + (p2 & 63)];
        const int k = tkz_contraction_len_o200k(b1, b2);
        if (k == 2) k2b |= 1ull << pos;
        else if (k == 3) k3b |= 1ull << pos;
    }
    const uint64_t LEAD = ~CONT;
    const TkzPext px = tkz_pext_prepare(LEAD);
    TkzO2Chars c;
    c.n = tkz_popc64(LEAD);
    c.U = tkz_pext(U, LEAD, px); c.l = tkz_pext(l, LEAD, px); c.X = tkz_pext(Xl, LEAD, px); c.M = tkz_pext(M, LEAD, px);
    c.N = tkz_pext(N, LEAD, px); c.W = tkz_pext(W, LEAD, px); c.O = tkz_pext(O, LEAD, px); c.O2 = tkz_pext(O2, LEAD, px);
    c.CR = tkz_pext(CR, LEAD, px); c.SP = tkz_pext(SP, LEAD, px); c.SL = tkz_pext(SL, LEAD, px); c.AP = tkz_pext(AP, LEAD, px);
    c.k2 = tkz_pext(k2b, LEAD, px); c.k3 = tkz_pext(k3b, LEAD, px);
    c.ds = tkz_pext(dsb, LEAD, px);
    uint64_t start;
    if (!tkz_block_core_o200k(c, X, &start)) return false;
    *out = tkz_pdep(start, LEAD, px) | dsb;
    return true;
}
#endif  // TKZ_NO_SIMT
