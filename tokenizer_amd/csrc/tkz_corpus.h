// tkz_corpus.h -- deterministic synthetic corpora of BASELINE.json's configs.
// Counter-based: document i depends only on (kind, seed, i, min_len, max_len), so the device can
// generate a 10 M-document batch in place and the host can regenerate any single document for a spot
// check with the same function.  Integer arithmetic only (bit-identical on host and device).
//
//   kind 1  ASCII English/code-like: Zipf-ranked (s = 1) words from a fixed 4096-word table (4 % of it longer than
//           16 bytes, all in the rarest octave) joined by single spaces, sentence punctuation, newlines, double
//           spaces, 1..6-digit numbers, contractions ('s 're 'll and upper-case 'S), code-like punctuation
//           runs, identifiers (snake_case, camelCase, PascalCase, dotted.paths, calls), URLs and hex blobs
//                                                                                  (configs 1, 2, 4)
//   kind 2  mixed UTF-8: ASCII words, BMP CJK / kana / hangul runs, emoji incl. supplementary-plane
//           ZWJ / VS-16 / skin-tone sequences placed directly against letters and CJK, and
//           U+3000 / NBSP / NEL white space                                        (config 3)
//   kind 3  kind 1 plus, in 1 document out of 100, long single-class runs (4 Ki..32 Ki of one
//           letter, '=', spaces, digits, or a camelCase chain)                     (config 5)
// A document is exactly its drawn length: the last item is cut (ASCII) or the tail is filled with
// ASCII letters when a multi-byte item no longer fits.
#pragma once
#include <stdint.h>

#include "tkz_simt.h"

constexpr int kTkzCorpusWordCount = 4096, kTkzCorpusWordWidth = 24;
static constexpr char kTkzCorpusWords[kTkzCorpusWordCount * kTkzCorpusWordWidth + 1] =
#include "tkz_corpus_words.inc"
    ;

struct TkzRng {
    uint64_t s;
    TKZ_HD uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    TKZ_HD uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }   // uniform in [0, n)
};

struct TkzEmit {
    uint8_t* buf; int64_t cap; int64_t n;     // buf may be null (length query never needs content)
    TKZ_HD bool full() const { return n >= cap; }
    TKZ_HD void put(uint32_t b) { if (n < cap) { if (buf) buf[n] = (uint8_t)b; ++n; } }
    TKZ_HD int64_t room() const { return cap - n; }
    TKZ_HD void cp(uint32_t c) {              // UTF-8 encode; caller checked room() >= 4
        if (c < 0x80) put(c);
        else if (c < 0x800) { put(0xC0 | (c >> 6)); put(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { put(0xE0 | (c >> 12)); put(0x80 | ((c >> 6) & 0x3F)); put(0x80 | (c & 0x3F)); }
        else { put(0xF0 | (c >> 18)); put(0x80 | ((c >> 12) & 0x3F)); put(0x80 | ((c >> 6) & 0x3F)); put(0x80 | (c & 0x3F)); }
    }
};

// log-uniform rank in [0, 4095) (Zipf s = 1): octave k (uniform), then uniform inside [2^k - 1, 2^(k+1) - 1)
TKZ_HD uint32_t tkz_zipf_rank(TkzRng& r) {
    const uint32_t k = r.below(12);
    return ((1u << k) - 1u) + r.below(1u << k);
}
TKZ_HD void tkz_emit_word_rank(TkzEmit& e, uint32_t rank, int casing);
TKZ_HD void tkz_emit_word(TkzEmit& e, TkzRng& r, int casing) { tkz_emit_word_rank(e, tkz_zipf_rank(r), casing); }   // casing: 0 lower, 1 Capitalised, 2 UPPER
TKZ_HD void tkz_emit_word_rank(TkzEmit& e, uint32_t rank, int casing) {
    const char* w = &kTkzCorpusWords[rank * kTkzCorpusWordWidth];
    for (int i = 0; i < kTkzCorpusWordWidth && w[i]; ++i) {
        uint32_t c = (uint8_t)w[i];
        if (casing == 2 || (casing == 1 && i == 0)) c -= 32;
        e.put(c);
    }
}
TKZ_HD void tkz_emit_number(TkzEmit& e, TkzRng& r) {
    const int nd = 1 + (int)r.below(6);
    for (int i = 0; i < nd; ++i) e.put('0' + r.below(10));
}
TKZ_HD void tkz_emit_punct_run(TkzEmit& e, TkzRng& r) {
    const char* const runs[] = {"();", "==", "->", " {", "}", "//", "/*", "*/", "[]", "::", "!=", "<=", "&&", "||", "+=", "...", "\"\"", "<>", "#", "$(", ");", "},", "=>", "--"};
    const char* s = runs[r.below(24)];
    for (int i = 0; s[i]; ++i) e.put((uint8_t)s[i]);
}

// identifiers as source code has them: snake_case, camelCase, PascalCase, a dotted path, a call -- two or three of the
// 511 commonest (short) words
TKZ_HD void tkz_emit_identifier(TkzEmit& e, TkzRng& r) {
    const uint32_t style = r.below(5);
    const int n = 2 + (int)r.below(2);
    for (int i = 0; i < n; ++i) {
        if (i && style == 0) e.put('_');
        if (i && style == 3) e.put('.');
        const uint32_t k = r.below(9);
        tkz_emit_word_rank(e, ((1u << k) - 1u) + r.below(1u << k), (style == 1 && i > 0) || style == 2 ? 1 : 0);
    }
    if (style == 4) { e.put('('); if (r.below(2)) e.put(')'); }
}
TKZ_HD void tkz_emit_hex(TkzEmit& e, TkzRng& r, int n) {
    for (int i = 0; i < n; ++i) { const uint32_t d = r.below(16); e.put(d < 10 ? '0' + d : 'a' + d - 10); }
}
TKZ_HD void tkz_emit_url(TkzEmit& e, TkzRng& r) {
    const char* const head = r.below(4) ? "https://www." : "http://";
    for (int i = 0; head[i]; ++i) e.put((uint8_t)head[i]);
    tkz_emit_word(e, r, 0);
    const char* const tld[] = {".com/", ".org/", ".net/", ".io/"};
    const char* t = tld[r.below(4)];
    for (int i = 0; t[i]; ++i) e.put((uint8_t)t[i]);
    const int n = 1 + (int)r.below(3);
    for (int i = 0; i < n; ++i) { if (i) e.put('/'); tkz_emit_word(e, r, 0); }
    if (r.below(3) == 0) { const char* q = "?id="; for (int i = 0; q[i]; ++i) e.put((uint8_t)q[i]); tkz_emit_number(e, r); }
}

TKZ_HD void tkz_corpus_ascii_item(TkzEmit& e, TkzRng& r, bool first) {
    const uint32_t u = r.below(1000);
    if (!first) {
        if (u < 20) { e.put('\n'); if (r.below(3) == 0) e.put('\n'); if (r.below(2) == 0) { e.put(' '); e.put(' '); e.put(' '); e.put(' '); } }
        else if (u < 45) { e.put(' '); e.put(' '); }
        else if (u < 60) { e.put('\t'); }
        else if (u < 75) { /* no separator: item directly against the previous one */ }
        else e.put(' ');
    }
    const uint32_t v = r.below(1000);
    if (v < 60) tkz_emit_number(e, r);
    else if (v < 110) tkz_emit_punct_run(e, r);
    else if (v < 130) tkz_emit_identifier(e, r);
    else if (v < 134) tkz_emit_url(e, r);
    else if (v < 137) { if (r.below(2)) { e.put('0'); e.put('x'); tkz_emit_hex(e, r, r.below(2) ? 8 : 16); } else tkz_emit_hex(e, r, r.below(2) ? 32 : 40); }
    else {
        const uint32_t c = r.below(100);
        tkz_emit_word(e, r, c < 86 ? 0 : (c < 97 ? 1 : 2));
        const uint32_t t = r.below(1000);
        if (t < 18) { e.put('\''); e.put('s'); }
        else if (t < 24) { e.put('\''); e.put('r'); e.put('e'); }
        else if (t < 30) { e.put('\''); e.put('l'); e.put('l'); }
        else if (t < 34) { e.put('\''); e.put('S'); }
        else if (t < 38) { e.put('\''); e.put('t'); }
        else if (t < 100) e.put(','); 
        else if (t < 170) e.put('.');
        else if (t < 180) e.put('?');
        else if (t < 188) e.put('!');
        else if (t < 196) e.put(';');
        else if (t < 204) e.put(':');
    }
}

TKZ_HD void tkz_corpus_utf8_item(TkzEmit& e, TkzRng& r, bool first) {
    const uint32_t u = r.below(100);
    if (e.room() < 40) { while (!e.full()) e.put('a' + r.below(26)); return; }   // tail: ASCII fill, never a cut char
    if (u < 50) { tkz_corpus_ascii_item(e, r, first); return; }
    if (u < 85) {   // CJK / kana / hangul run of 1..8 chars, usually directly against what precedes
        if (!first && r.below(4) == 0) e.put(' ');
        const int n = 1 + (int)r.below(8);
        const uint32_t scr = r.below(10);
        for (int i = 0; i < n; ++i) {
            if (scr < 6) e.cp(0x4E00 + r.below(0x9FFC - 0x4E00 + 1));
            else if (scr < 8) e.cp(0x3041 + r.below(0x30FA - 0x3041 + 1));
            else e.cp(0xAC00 + r.below(0xD7A3 - 0xAC00 + 1));
        }
        if (r.below(5) == 0) e.cp(r.below(2) ? 0x3002 : 0xFF0C);   // ideographic full stop / fullwidth comma
        return;
    }
    if (u < 95) {   // emoji, with ZWJ / VS-16 / skin-tone sequences, glued to neighbours
        const uint32_t k = r.below(6);
        if (k == 0) { e.cp(0x1F468); e.cp(0x200D); e.cp(0x1F469); e.cp(0x200D); e.cp(0x1F467); }
        else if (k == 1) { e.cp(0x2764); e.cp(0xFE0F); }
        else if (k == 2) { e.cp(0x1F44D); e.cp(0x1F3FB + r.below(5)); }
        else if (k == 3) { e.cp(0x2B50); }
        else e.cp(0x1F300 + r.below(0x1FAFF - 0x1F300 + 1));
        return;
    }
    // exotic white space: U+3000, NBSP, NEL, sometimes doubled or before a newline
    const uint32_t k = r.below(3);
    e.cp(k == 0 ? 0x3000 : (k == 1 ? 0xA0 : 0x85));
    if (r.below(3) == 0) e.cp(0x3000);
    if (r.below(4) == 0) e.put('\n');
}

TKZ_HD void tkz_corpus_long_run(TkzEmit& e, TkzRng& r) {
    const int64_t n = 4096 + (int64_t)r.below(28673);          // 4 Ki .. 32 Ki
    const uint32_t k = r.below(5);
    if (k == 0) { const uint32_t c = 'a' + r.below(26); for (int64_t i = 0; i < n; ++i) e.put(c); }
    else if (k == 1) for (int64_t i = 0; i < n; ++i) e.put('=');
    else if (k == 2) for (int64_t i = 0; i < n; ++i) e.put(' ');
    else if (k == 3) for (int64_t i = 0; i < n; ++i) e.put('0' + r.below(10));
    else { const int64_t end = e.n + n; while (e.n < end && !e.full()) tkz_emit_word(e, r, 1); }
}

// Generates document `doc` into buf (cap bytes; pass the document's own length) and returns its length.
// With buf == nullptr only the length is computed (O(1)).
TKZ_HD int64_t tkz_corpus_doc(int kind, uint64_t seed, int64_t doc, int min_len, int max_len, uint8_t* buf, int64_t cap) {
    TkzRng r;
    r.s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(doc + 1)) ^ ((uint64_t)kind << 56);
    r.next();
    const int64_t len = (int64_t)min_len + (int64_t)r.below((uint32_t)(max_len - min_len + 1));
    if (!buf) return len;
    TkzEmit e; e.buf = buf; e.cap = cap < len ? cap : len; e.n = 0;
    const bool long_runs = kind == 3 && r.below(100) == 0;
    bool first = true;
    while (!e.full()) {
        if (long_runs && r.below(64) == 0) tkz_corpus_long_run(e, r);
        else if (kind == 5) { e.put(' '); tkz_emit_word_rank(e, r.below(4096), 0); }     // words of the table, uniformly, each behind a single space
        else if (kind == 2) tkz_corpus_utf8_item(e, r, first);
        else tkz_corpus_ascii_item(e, r, first);
        first = false;
    }
    return len;
}
