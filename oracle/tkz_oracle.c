/*
 * tkz_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See tkz_oracle.h.
 *
 * Every function cites the reference lines it restates.  The code favours being a literal
 * reading of the reference (lists, RemoveAt, a rank lookup per GetRank, regex alternatives
 * tried in order with explicit backtracking loops) over speed: it is the checker.
 */
#define _GNU_SOURCE
#include "tkz_oracle.h"

#include <limits.h>
#include <pthread.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Unicode 13.0 BMP code-unit classes (generated; see tools/gen_unicode_tables.py)             */
/* ------------------------------------------------------------------------------------------ */
enum { C_OTHER = 0, C_LU = 1, C_LL = 2, C_LT = 3, C_LM = 4, C_LO = 5, C_M = 6, C_N = 7, C_WS = 8 };

static const struct { uint16_t a, b; uint8_t c; } k_ranges[] = {
#include "unicode13_classes.inc"
};
static uint8_t g_cls[65536];
static pthread_once_t g_cls_once = PTHREAD_ONCE_INIT;
static void cls_init(void) {
    memset(g_cls, 0, sizeof g_cls);
    for (size_t i = 0; i < sizeof k_ranges / sizeof k_ranges[0]; ++i)
        for (unsigned u = k_ranges[i].a; u <= k_ranges[i].b; ++u) g_cls[u] = k_ranges[i].c;
}
/* Supplementary planes, by code point: used ONLY for o200k, the one pattern that exists only in the TypeScript reference and is
 * compiled there with `new RegExp(pattern, "gu")` (tokenizer_ts/src/tikTokenizer.ts:100): code-point matching, one class test per
 * CHARACTER.  The two patterns the C# reference defines keep .NET's code-unit matching (a supplementary char = two OTHER units). */
static const struct { uint32_t a, b; uint8_t c; } k_supp[] = {
#include "unicode13_supp.inc"
};
/* Overrides (tests of tkz_encoder_set_unicode_classes / TKZ_OPT_CASE_EQUIVALENCE: the host's runtime defines the split, TikTokenizer.cs:77): a class
 * per code point handed over by the caller instead of the built-in Unicode 13.0 data, and U+017F as an `s` in cl100k's (?i:...) (.NET >= 7).  Process-wide;
 * the checker is not used from several threads with different settings. */
static const uint8_t* g_cls_override = NULL;      /* 65536 or 0x110000 entries */
static int64_t g_cls_override_n = 0;
static int g_case_equiv = 0;
static uint8_t supp_class(uint32_t cp) {
    if (g_cls_override && (int64_t)cp < g_cls_override_n) return g_cls_override[cp];
    size_t lo = 0, hi = sizeof k_supp / sizeof k_supp[0];
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (k_supp[mid].b < cp) lo = mid + 1; else hi = mid; }
    return (lo < sizeof k_supp / sizeof k_supp[0] && k_supp[lo].a <= cp) ? k_supp[lo].c : C_OTHER;
}
/* class of a code point under the ECMAScript `u`-flag semantics of the o200k pattern: Unicode categories as above, but \s is
 * ECMAScript's WhiteSpace + LineTerminator = [\t\n\v\f\r \u00a0\ufeff\p{Zs}\u2028\u2029]: U+FEFF IS white space, U+0085 is NOT
 * (it is Cc, hence [^\s\p{L}\p{N}]).  A lone surrogate (possible only through the UTF-16 entry) is Cs: OTHER. */
static uint8_t js_class(uint32_t cp) {
    if (cp == 0xFEFF) return C_WS;
    if (cp == 0x85) return C_OTHER;
    return cp < 0x10000 ? g_cls[cp] : supp_class(cp);
}
void tkzo_set_unicode_classes(const uint8_t* classes, int64_t n) {
    pthread_once(&g_cls_once, cls_init);
    cls_init();                                      /* the built-in table again ... */
    g_cls_override = NULL; g_cls_override_n = 0;
    if (!classes) return;
    /* ... then the caller's classes for everything but ASCII and the surrogate code units, as the library takes them */
    for (int64_t cp = 128; cp < 65536 && cp < n; ++cp) g_cls[cp] = (cp >= 0xD800 && cp <= 0xDFFF) ? 0 : classes[cp];
    if (n > 65536) { g_cls_override = classes; g_cls_override_n = n; }      /* (kept by reference: the caller keeps the array alive) */
}
void tkzo_set_case_equivalence(int on) { g_case_equiv = on; }
static inline int isL(uint8_t c) { return c >= C_LU && c <= C_LO; }
static inline int isN(uint8_t c) { return c == C_N; }
static inline int isWS(uint8_t c) { return c == C_WS; }

/* ------------------------------------------------------------------------------------------ */
/* Rank dictionary: exact byte string -> rank.  Stands in for                                  */
/* Dictionary<byte[],int>(ByteArrayComparer)  (TikTokenizer.cs:101, BytePairComparer.cs:8-43). */
/* Only exact-match semantics matter; the reference's hash function is not observable.         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t off; int32_t len; int32_t rank; int32_t used; } dslot;
struct tkzo_vocab {
    dslot* slots; uint64_t nslots; /* power of two */
    uint8_t* arena; size_t arena_len, arena_cap;
    int64_t count; int max_key_len;
    /* insertion-ordered list for enumeration */
    uint64_t* order; int64_t order_cap;
};

static uint64_t fnv1a(const uint8_t* p, int64_t n) {
    uint64_t h = 1469598103934665603ULL;
    for (int64_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
    return h ^ (h >> 29);
}
static dslot* dict_find(const tkzo_vocab* v, const uint8_t* key, int len) {
    uint64_t m = v->nslots - 1, i = fnv1a(key, len) & m;
    for (;;) {
        dslot* s = &v->slots[i];
        if (!s->used) return s;
        if (s->len == len && memcmp(v->arena + s->off, key, (size_t)len) == 0) return s;
        i = (i + 1) & m;
    }
}
static int dict_grow(tkzo_vocab* v) {
    uint64_t on = v->nslots; dslot* os = v->slots;
    v->nslots = on ? on * 2 : 1024;
    v->slots = (dslot*)calloc(v->nslots, sizeof(dslot));
    if (!v->slots) return -1;
    for (uint64_t i = 0; i < on; ++i) if (os[i].used) {
        dslot* s = dict_find(v, v->arena + os[i].off, os[i].len); *s = os[i];
    }
    /* order[] stores arena offsets (stable), so nothing to fix up */
    free(os);
    return 0;
}
/* bpeDict[tokenBytes] = rank  (TikTokenizer.cs:125): a repeated key overwrites. */
static int dict_set(tkzo_vocab* v, const uint8_t* key, int len, int32_t rank) {
    if ((uint64_t)(v->count + 1) * 2 > v->nslots) if (dict_grow(v)) return -1;
    dslot* s = dict_find(v, key, len);
    if (s->used) { s->rank = rank; return 0; }
    if (v->arena_len + (size_t)len + 1 > v->arena_cap) {
        size_t nc = v->arena_cap ? v->arena_cap * 2 : (1u << 20);
        while (nc < v->arena_len + (size_t)len + 1) nc *= 2;
        uint8_t* na = (uint8_t*)realloc(v->arena, nc); if (!na) return -1;
        v->arena = na; v->arena_cap = nc;
    }
    memcpy(v->arena + v->arena_len, key, (size_t)len);
    s->off = (uint32_t)v->arena_len; s->len = len; s->rank = rank; s->used = 1;
    v->arena_len += (size_t)len;
    if (v->count == v->order_cap) {
        int64_t nc = v->order_cap ? v->order_cap * 2 : 65536;
        uint64_t* no = (uint64_t*)realloc(v->order, (size_t)nc * sizeof(uint64_t)); if (!no) return -1;
        v->order = no; v->order_cap = nc;
    }
    v->order[v->count++] = ((uint64_t)s->off << 32) | (uint32_t)len;
    if (len > v->max_key_len) v->max_key_len = len;
    return 0;
}
int32_t tkzo_vocab_rank(const tkzo_vocab* v, const uint8_t* key, int len) {
    if (!v->nslots) return -1;
    dslot* s = dict_find(v, key, len);
    return s->used ? s->rank : -1;
}
/* TryGetValue */
static inline int dict_get(const tkzo_vocab* v, const uint8_t* key, int64_t len, int32_t* rank) {
    if (len > INT_MAX || !v->nslots) return 0;
    dslot* s = dict_find(v, key, (int)len);
    if (!s->used) return 0;
    *rank = s->rank; return 1;
}
/* ranks.TryGetValue(slice, out rank) returning int.MaxValue on miss (BytePairEncoder.cs:30-35) */
static inline int32_t rank_or_max(const tkzo_vocab* v, const uint8_t* key, int64_t len) {
    if (len > INT_MAX || !v->nslots) return INT_MAX;
    dslot* s = dict_find(v, key, (int)len);
    return s->used ? s->rank : INT_MAX;
}
int64_t tkzo_vocab_size(const tkzo_vocab* v) { return v->count; }
int tkzo_vocab_max_key_len(const tkzo_vocab* v) { return v->max_key_len; }
int tkzo_vocab_entry(const tkzo_vocab* v, int64_t i, uint8_t* buf, int cap, int32_t* rank) {
    if (i < 0 || i >= v->count) return -1;
    uint32_t off = (uint32_t)(v->order[i] >> 32); int len = (int)(uint32_t)v->order[i];
    if (len > cap) return -1;
    memcpy(buf, v->arena + off, (size_t)len);
    if (rank) *rank = tkzo_vocab_rank(v, v->arena + off, len);
    return len;
}
void tkzo_vocab_free(tkzo_vocab* v) {
    if (!v) return;
    free(v->slots); free(v->arena); free(v->order); free(v);
}

/* ------------------------------------------------------------------------------------------ */
/* LoadTikTokenBpe (TikTokenizer.cs:99-139) + the duplicate-rank check of Init (:80-87)        */
/* ------------------------------------------------------------------------------------------ */
static int b64val(int c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}
/* Convert.FromBase64String: standard alphabet, length % 4 == 0, '=' padding only at the end. */
static int b64decode(const uint8_t* s, int n, uint8_t* out, int cap) {
    if (n % 4) return -1;
    int o = 0;
    for (int i = 0; i < n; i += 4) {
        int v[4], pad = 0;
        for (int k = 0; k < 4; ++k) {
            int c = s[i + k];
            if (c == '=') {
                if (i + 4 != n || k < 2) return -1;
                v[k] = 0; ++pad;
            } else {
                if (pad) return -1;
                v[k] = b64val(c); if (v[k] < 0) return -1;
            }
        }
        uint32_t w = ((uint32_t)v[0] << 18) | ((uint32_t)v[1] << 12) | ((uint32_t)v[2] << 6) | (uint32_t)v[3];
        if (o + 3 - pad > cap) return -1;
        out[o++] = (uint8_t)(w >> 16);
        if (pad < 2) out[o++] = (uint8_t)(w >> 8);
        if (pad < 1) out[o++] = (uint8_t)w;
    }
    return o;
}
/* the file is read through a StreamReader (UTF-8): white space is decided on the decoded chars.  A lone byte 0x85 / 0xA0 decodes to
 * U+FFFD (not white space); C2 85, C2 A0 and the other char.IsWhiteSpace code points are. */
static int is_dotnet_ws_ascii(int c) { return c == ' ' || (c >= 9 && c <= 13); }
static size_t dotnet_ws_at(const uint8_t* p, size_t n) {   /* bytes of the white-space char that starts at p, 0 if none */
    if (n >= 1 && is_dotnet_ws_ascii(p[0])) return 1;
    if (n >= 2 && p[0] == 0xC2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
    if (n >= 3 && p[0] == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;                                   /* U+1680 */
    if (n >= 3 && p[0] == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3;
    if (n >= 3 && p[0] == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;                                   /* U+205F */
    if (n >= 3 && p[0] == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;                                   /* U+3000 */
    return 0;
}

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y);
}

tkzo_vocab* tkzo_vocab_load(const uint8_t* file, size_t n, int* err) {
    pthread_once(&g_cls_once, cls_init);
    int e = TKZO_OK;
    tkzo_vocab* v = (tkzo_vocab*)calloc(1, sizeof *v);
    if (!v) { if (err) *err = TKZO_E_ARG; return NULL; }
    size_t pos = 0;
    if (n >= 3 && file[0] == 0xEF && file[1] == 0xBB && file[2] == 0xBF) pos = 3; /* StreamReader skips a BOM */
    uint8_t* key = (uint8_t*)malloc(n + 4);
    while (pos < n && e == TKZO_OK) {
        /* StreamReader.ReadLine: a line ends at \n, \r or \r\n  (:107) */
        size_t ls = pos; while (pos < n && file[pos] != '\n' && file[pos] != '\r') ++pos;
        size_t le = pos;
        if (pos < n) { if (file[pos] == '\r' && pos + 1 < n && file[pos + 1] == '\n') pos += 2; else ++pos; }
        /* string.IsNullOrWhiteSpace(line) -> continue  (:109-112) */
        int blank = 1;
        for (size_t i = ls; i < le;) { const size_t w = dotnet_ws_at(file + i, le - i); if (!w) { blank = 0; break; } i += w; }
        if (blank) continue;
        /* line.Split(' ') must give exactly two fields  (:114-118) */
        int nsp = 0; size_t sp = 0;
        for (size_t i = ls; i < le; ++i) if (file[i] == ' ') { if (!nsp) sp = i; ++nsp; }
        if (nsp != 1) { e = TKZO_E_FORMAT; break; }
        int klen = b64decode(file + ls, (int)(sp - ls), key, (int)(n + 4)); /* (:120) */
        if (klen < 0) { e = TKZO_E_FORMAT; break; }
        /* int.TryParse(tokens[1]) (:122): optional white, optional sign, digits, optional white */
        size_t a = sp + 1, b = le;
        while (a < b && is_dotnet_ws_ascii(file[a])) ++a;
        while (b > a && is_dotnet_ws_ascii(file[b - 1])) --b;
        int neg = 0;
        if (a < b && (file[a] == '+' || file[a] == '-')) { neg = file[a] == '-'; ++a; }
        if (a >= b) { e = TKZO_E_FORMAT; break; }
        int64_t val = 0;
        for (size_t i = a; i < b; ++i) {
            if (file[i] < '0' || file[i] > '9') { e = TKZO_E_FORMAT; break; }
            val = val * 10 + (file[i] - '0');
            if (val > (int64_t)INT_MAX + 1) { e = TKZO_E_FORMAT; break; }
        }
        if (e) break;
        if (neg) val = -val;
        if (val > INT_MAX || val < INT_MIN) { e = TKZO_E_FORMAT; break; }
        if (dict_set(v, key, klen, (int32_t)val)) { e = TKZO_E_ARG; break; }
    }
    free(key);
    if (e == TKZO_OK && v->count > 0) {
        /* Decoder = Encoder.ToDictionary(value -> key): duplicate ranks throw ArgumentException (:82-87) */
        int32_t* r = (int32_t*)malloc((size_t)v->count * sizeof(int32_t));
        for (int64_t i = 0; i < v->count; ++i) {
            uint32_t off = (uint32_t)(v->order[i] >> 32); int len = (int)(uint32_t)v->order[i];
            r[i] = tkzo_vocab_rank(v, v->arena + off, len);
        }
        qsort(r, (size_t)v->count, sizeof(int32_t), cmp_i32);
        for (int64_t i = 1; i < v->count; ++i) if (r[i] == r[i - 1]) { e = TKZO_E_DUP_RANK; break; }
        free(r);
    }
    if (e != TKZO_OK) { tkzo_vocab_free(v); v = NULL; }
    if (err) *err = e;
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* BytePairEncoder.BytePairEncode  (Utils/BytePairEncoder.cs:13-76), line by line.             */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int64_t* idx; int32_t* rk; int64_t cap; } bpe_scratch;

static int32_t bpe_get_rank(const tkzo_vocab* v, const uint8_t* bytes, const int64_t* idx,
                            int64_t count, int64_t start, int64_t skip) {
    /* GetRank(startIndex, skip)  (:25-36) */
    if (start + skip + 2 < count) {
        int64_t a = idx[start], b = idx[start + skip + 2];
        return rank_or_max(v, bytes + a, b - a);
    }
    return INT_MAX;
}

static int64_t bpe_run(const tkzo_vocab* v, const uint8_t* bytes, int64_t n, int32_t* out,
                       int64_t cap, bpe_scratch* sc) {
    if (n == 1) { /* (:15-18) ranks[mergingBytes] -> KeyNotFoundException when absent */
        int32_t r;
        if (!dict_get(v, bytes, 1, &r)) return TKZO_E_KEY_NOT_FOUND;
        if (cap < 1) return TKZO_E_CAPACITY;
        out[0] = r;
        return 1;
    }
    if (sc->cap < n + 1) {
        free(sc->idx); free(sc->rk);
        sc->cap = (n + 1) * 2;
        sc->idx = (int64_t*)malloc((size_t)sc->cap * sizeof(int64_t));
        sc->rk = (int32_t*)malloc((size_t)sc->cap * sizeof(int32_t));
    }
    int64_t* idx = sc->idx; int32_t* rk = sc->rk;
    int64_t count = n + 1;                                   /* (:20-24) */
    for (int64_t i = 0; i < count; ++i) { idx[i] = i; rk[i] = INT_MAX; }
    for (int64_t i = 0; i < count - 2; ++i) {                /* (:37-44) */
        int32_t r = bpe_get_rank(v, bytes, idx, count, i, 0);
        if (r != INT_MAX) rk[i] = r;
    }
    while (count > 1) {                                      /* (:45) */
        int64_t mi = 0; int32_t mr = INT_MAX;                /* (:47) */
        for (int64_t i = 0; i < count - 1; ++i)              /* (:48-54) strict <  => leftmost min */
            if (rk[i] < mr) { mi = i; mr = rk[i]; }
        if (mr != INT_MAX) {                                 /* (:55) */
            int64_t j = mi;
            rk[j] = bpe_get_rank(v, bytes, idx, count, j, 1);               /* (:58) */
            if (j > 0) rk[j - 1] = bpe_get_rank(v, bytes, idx, count, j - 1, 1); /* (:59-62) */
            memmove(idx + j + 1, idx + j + 2, (size_t)(count - j - 2) * sizeof(int64_t)); /* RemoveAt(j+1) (:63) */
            memmove(rk + j + 1, rk + j + 2, (size_t)(count - j - 2) * sizeof(int32_t));
            --count;
        } else break;                                        /* (:65-68) */
    }
    if (count - 1 > cap) return TKZO_E_CAPACITY;
    for (int64_t i = 0; i < count - 1; ++i) {                /* (:70-75) ranks[...] throws when absent */
        int32_t r;
        if (!dict_get(v, bytes + idx[i], idx[i + 1] - idx[i], &r)) return TKZO_E_KEY_NOT_FOUND;
        out[i] = r;
    }
    return count - 1;
}

int64_t tkzo_bpe(const tkzo_vocab* v, const uint8_t* bytes, int64_t n, int32_t* out, int64_t cap) {
    bpe_scratch sc = {0};
    int64_t r = bpe_run(v, bytes, n, out, cap, &sc);
    free(sc.idx); free(sc.rk);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* The split regexes, as a leftmost-first backtracking matcher over UTF-16 code units.         */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint32_t* u;   /* the text as the regex engine sees it: UTF-16 code units (pattern 1, cl100k: .NET) or code points (o200k: JS /u) */
    const uint8_t* c;    /* class per unit / code point */
    int64_t n;
} utext;

#define U(i) (t->u[i])
#define CL(i) (t->c[i])
static inline int u_crlf(const utext* t, int64_t i) { return U(i) == '\r' || U(i) == '\n'; }
/* [^\s\p{L}\p{N}] */
static inline int u_other(const utext* t, int64_t i) { uint8_t c = CL(i); return !isWS(c) && !isL(c) && !isN(c); }
/* [^\r\n\p{L}\p{N}] */
static inline int u_prefix(const utext* t, int64_t i) { uint8_t c = CL(i); return !u_crlf(t, i) && !isL(c) && !isN(c); }

static inline int lower_ascii(int c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }

/* 's|'t|'re|'ve|'m|'ll|'d at p, alternatives in the written order.
 * mode 0: case-sensitive lower only (pattern 1, TokenizerBuilder.cs:128)
 * mode 1: (?i:...) (cl100k, TokenizerBuilder.cs:112) -- ASCII case pairs only (net6.0)
 * mode 2: the explicit o200k list 's|'S|'t|'T|'re|'RE|'Re|'eR|'ve|'VE|'vE|'Ve|'m|'M|'ll|'lL|'Ll|'LL|'d|'D
 *         (tokenizer_ts/src/tokenizerBuilder.ts:80-81).  Note 'eR is listed where 'rE would be. */
static int64_t m_contraction(const utext* t, int64_t p, int mode) {
    if (p >= t->n || U(p) != '\'') return -1;
    int64_t r = t->n - p - 1;
    if (r < 1) return -1;
    int a = (int)U(p + 1), b = r >= 2 ? (int)U(p + 2) : -1;
    if (mode == 1 && g_case_equiv && a == 0x17F) return p + 2;      /* .NET >= 7: U+017F is case-equivalent to s / S */
    if (a > 127) return -1;
    if (mode == 0) {
        if (a == 's' || a == 't') return p + 2;
        if (a == 'r' && b == 'e') return p + 3;
        if (a == 'v' && b == 'e') return p + 3;
        if (a == 'm') return p + 2;
        if (a == 'l' && b == 'l') return p + 3;
        if (a == 'd') return p + 2;
        return -1;
    }
    if (mode == 1) {
        int la = lower_ascii(a), lb = (b >= 0 && b <= 127) ? lower_ascii(b) : -1;
        if (la == 's' || la == 't') return p + 2;
        if (la == 'r' && lb == 'e') return p + 3;
        if (la == 'v' && lb == 'e') return p + 3;
        if (la == 'm') return p + 2;
        if (la == 'l' && lb == 'l') return p + 3;
        if (la == 'd') return p + 2;
        return -1;
    }
    /* mode 2: explicit list, in order */
    if (a == 's' || a == 'S' || a == 't' || a == 'T') return p + 2;
    if ((a == 'r' && b == 'e') || (a == 'R' && b == 'E') || (a == 'R' && b == 'e') || (a == 'e' && b == 'R')) return p + 3;
    if ((a == 'v' && b == 'e') || (a == 'V' && b == 'E') || (a == 'v' && b == 'E') || (a == 'V' && b == 'e')) return p + 3;
    if (a == 'm' || a == 'M') return p + 2;
    if ((a == 'l' && b == 'l') || (a == 'l' && b == 'L') || (a == 'L' && b == 'l') || (a == 'L' && b == 'L')) return p + 3;
    if (a == 'd' || a == 'D') return p + 2;
    return -1;
}

/* \s+(?!\S): greedy run, give back units until the lookahead holds. */
static int64_t m_ws_not_before_nonws(const utext* t, int64_t p) {
    int64_t k = 0; while (p + k < t->n && isWS(CL(p + k))) ++k;
    for (int64_t q = k; q >= 1; --q) {
        int64_t e = p + q;
        if (e == t->n || isWS(CL(e))) return e;   /* (?!\S): at end of text, or next unit is \s */
    }
    return -1;
}
/* \s+ */
static int64_t m_ws(const utext* t, int64_t p) {
    int64_t k = 0; while (p + k < t->n && isWS(CL(p + k))) ++k;
    return k ? p + k : -1;
}
/* \s*[\r\n]+ : greedy \s*, backtrack until [\r\n]+ can match */
static int64_t m_ws_then_newlines(const utext* t, int64_t p) {
    int64_t k = 0; while (p + k < t->n && isWS(CL(p + k))) ++k;
    for (int64_t q = k; q >= 0; --q) {
        int64_t s = p + q, e = s;
        while (e < t->n && u_crlf(t, e)) ++e;
        if (e > s) return e;
    }
    return -1;
}

/* pattern 1:  's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+ */
static int64_t match_p1(const utext* t, int64_t p) {
    int64_t e = m_contraction(t, p, 0);
    if (e >= 0) return e;
    for (int kind = 0; kind < 3; ++kind) {           /*  ?\p{L}+ ,  ?\p{N}+ ,  ?[^\s\p{L}\p{N}]+  */
        for (int sp = 1; sp >= 0; --sp) {            /* ' ?' is greedy: try the space first */
            if (sp && U(p) != ' ') continue;
            int64_t q = p + sp, k = q;
            while (k < t->n && (kind == 0 ? isL(CL(k)) : kind == 1 ? isN(CL(k)) : u_other(t, k))) ++k;
            if (k > q) return k;
        }
    }
    e = m_ws_not_before_nonws(t, p);
    if (e >= 0) return e;
    return m_ws(t, p);
}

/* cl100k: (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+ */
static int64_t match_cl100k(const utext* t, int64_t p) {
    int64_t e = m_contraction(t, p, 1);
    if (e >= 0) return e;
    for (int pre = 1; pre >= 0; --pre) {             /* [^\r\n\p{L}\p{N}]?\p{L}+ */
        if (pre && !u_prefix(t, p)) continue;
        int64_t q = p + pre, k = q;
        while (k < t->n && isL(CL(k))) ++k;
        if (k > q) return k;
    }
    {                                                /* \p{N}{1,3} */
        int64_t k = p; while (k < t->n && k < p + 3 && isN(CL(k))) ++k;
        if (k > p) return k;
    }
    for (int sp = 1; sp >= 0; --sp) {                /*  ?[^\s\p{L}\p{N}]+[\r\n]* */
        if (sp && U(p) != ' ') continue;
        int64_t q = p + sp, k = q;
        while (k < t->n && u_other(t, k)) ++k;
        if (k > q) { while (k < t->n && u_crlf(t, k)) ++k; return k; }
    }
    e = m_ws_then_newlines(t, p);
    if (e >= 0) return e;
    e = m_ws_not_before_nonws(t, p);
    if (e >= 0) return e;
    return m_ws(t, p);
}

/* o200k (tokenizer_ts/src/tokenizerBuilder.ts:79-89), A=[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}], B=[\p{Ll}\p{Lm}\p{Lo}\p{M}]:
 *  [^\r\n\p{L}\p{N}]?A*B+(?:contr)? | [^\r\n\p{L}\p{N}]?A+B*(?:contr)? | \p{N}{1,3}
 *  |  ?[^\s\p{L}\p{N}]+[\r\n/]* | \s*[\r\n]+ | \s+(?!\S) | \s+ */
static inline int o2_A(uint8_t c) { return c == C_LU || c == C_LT || c == C_LM || c == C_LO || c == C_M; }
static inline int o2_B(uint8_t c) { return c == C_LL || c == C_LM || c == C_LO || c == C_M; }
static int64_t match_o200k(const utext* t, int64_t p) {
    for (int pre = 1; pre >= 0; --pre) {             /* alt 1: prefix? A* B+ contr? */
        if (pre && !u_prefix(t, p)) continue;
        int64_t q = p + pre, a = 0;
        while (q + a < t->n && o2_A(CL(q + a))) ++a;
        for (int64_t aa = a; aa >= 0; --aa) {        /* A* greedy, give back one unit at a time */
            int64_t s = q + aa, k = s;
            while (k < t->n && o2_B(CL(k))) ++k;
            if (k > s) { int64_t c = m_contraction(t, k, 2); return c >= 0 ? c : k; }
        }
    }
    for (int pre = 1; pre >= 0; --pre) {             /* alt 2: prefix? A+ B* contr? */
        if (pre && !u_prefix(t, p)) continue;
        int64_t q = p + pre, k = q;
        while (k < t->n && o2_A(CL(k))) ++k;
        if (k > q) {
            while (k < t->n && o2_B(CL(k))) ++k;
            int64_t c = m_contraction(t, k, 2); return c >= 0 ? c : k;
        }
    }
    {                                                /* \p{N}{1,3} */
        int64_t k = p; while (k < t->n && k < p + 3 && isN(CL(k))) ++k;
        if (k > p) return k;
    }
    for (int sp = 1; sp >= 0; --sp) {                /*  ?[^\s\p{L}\p{N}]+[\r\n/]* */
        if (sp && U(p) != ' ') continue;
        int64_t q = p + sp, k = q;
        while (k < t->n && u_other(t, k)) ++k;
        if (k > q) { while (k < t->n && (u_crlf(t, k) || U(k) == '/')) ++k; return k; }
    }
    int64_t e = m_ws_then_newlines(t, p);
    if (e >= 0) return e;
    e = m_ws_not_before_nonws(t, p);
    if (e >= 0) return e;
    return m_ws(t, p);
}

static int64_t match_at(int pattern, const utext* t, int64_t p) {
    switch (pattern) {
        case TKZO_PATTERN_P1: return match_p1(t, p);
        case TKZO_PATTERN_CL100K: return match_cl100k(t, p);
        case TKZO_PATTERN_O200K: return match_o200k(t, p);          /* by code point, ECMAScript \s: what `t` holds decides */
        case TKZO_PATTERN_O200K_DOTNET: return match_o200k(t, p);   /* by code unit, .NET \s (TikTokenizer.cs:77 given the o200k string) */
    }
    return -1;
}

/* Regex.Matches(text): scan left to right; where no alternative matches, advance one unit. */
typedef void (*piece_fn)(void* ctx, int64_t ustart, int64_t ulen);
static void split_units(int pattern, const utext* t, piece_fn fn, void* ctx) {
    int64_t p = 0;
    while (p < t->n) {
        int64_t e = match_at(pattern, t, p);
        if (e <= p) { ++p; continue; }
        fn(ctx, p, e - p);
        p = e;
    }
}

/* UTF-8 (valid) -> UTF-16 units + classes + byte offset of each unit (the low half of a pair maps
 * to the same byte offset as the high half; no shipped pattern can split a pair). */
typedef struct { uint32_t* u; uint8_t* c; int64_t* off; int64_t n; int64_t cap; } u16buf;
/* (cap: a buffer that outlives the call -- the per-thread scratch of the batch baseline -- is grown, not re-allocated per document:
 *  three malloc/free pairs per 500-byte document were a third of the CPU baseline's time on 256 threads) */
static void u16buf_reserve(u16buf* b, int64_t n) {
    if (b->cap >= n + 2) return;
    free(b->u); free(b->c); free(b->off);
    b->cap = (n + 2) * 2;
    b->u = (uint32_t*)malloc((size_t)b->cap * sizeof(uint32_t));
    b->c = (uint8_t*)malloc((size_t)b->cap);
    b->off = (int64_t*)malloc((size_t)b->cap * sizeof(int64_t));
}
static int utf8_to_units(const uint8_t* s, int64_t n, u16buf* b, int by_code_point) {
    u16buf_reserve(b, n);
    int64_t i = 0, k = 0;
    while (i < n) {
        uint32_t c = s[i]; int len;
        if (c < 0x80) len = 1;
        else if (c >= 0xC2 && c <= 0xDF) len = 2;
        else if (c >= 0xE0 && c <= 0xEF) len = 3;
        else if (c >= 0xF0 && c <= 0xF4) len = 4;
        else return TKZO_E_UTF8;
        if (i + len > n) return TKZO_E_UTF8;
        uint32_t cp = c;
        if (len > 1) {
            cp = c & (0xFFu >> (len + 1));
            for (int j = 1; j < len; ++j) {
                uint32_t d = s[i + j];
                if ((d & 0xC0) != 0x80) return TKZO_E_UTF8;
                cp = (cp << 6) | (d & 0x3F);
            }
            if ((len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) ||
                (len == 4 && (cp < 0x10000 || cp > 0x10FFFF))) return TKZO_E_UTF8;
        }
        if (by_code_point) {
            b->u[k] = cp; b->c[k] = js_class(cp); b->off[k] = i; ++k;
        } else if (cp < 0x10000) {
            b->u[k] = cp; b->c[k] = g_cls[cp]; b->off[k] = i; ++k;
        } else {
            cp -= 0x10000;
            b->u[k] = 0xD800 + (cp >> 10); b->c[k] = C_OTHER; b->off[k] = i; ++k;
            b->u[k] = 0xDC00 + (cp & 0x3FF); b->c[k] = C_OTHER; b->off[k] = i; ++k;
        }
        i += len;
    }
    b->off[k] = n; b->n = k;
    return TKZO_OK;
}
static void u16buf_free(u16buf* b) { free(b->u); free(b->c); free(b->off); b->u = NULL; b->c = NULL; b->off = NULL; b->cap = 0; }

/* UTF-16 units -> what the engine sees + the unit offset of each entry: the units themselves (.NET), or -- by code point -- a
 * well-formed surrogate pair as ONE entry and a lone surrogate as one entry of class OTHER (Cs). */
static void units_to_text(const uint16_t* text, int64_t n, u16buf* b, int by_code_point) {
    u16buf_reserve(b, n);
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t x = text[i];
        b->off[k] = i;
        if (by_code_point && x >= 0xD800 && x <= 0xDBFF && i + 1 < n && text[i + 1] >= 0xDC00 && text[i + 1] <= 0xDFFF) {
            x = 0x10000 + ((x - 0xD800) << 10) + (text[i + 1] - 0xDC00); ++i;
            b->u[k] = x; b->c[k] = js_class(x);
        } else if (x >= 0xD800 && x <= 0xDFFF) { b->u[k] = x; b->c[k] = C_OTHER; }
        else { b->u[k] = x; b->c[k] = by_code_point ? js_class(x) : g_cls[x]; }
        ++k;
    }
    b->off[k] = n; b->n = k;
}

typedef struct { int64_t* starts; int64_t* lens; int64_t cap, count; const int64_t* off; } collect_ctx;
static void collect_piece(void* vc, int64_t us, int64_t ul) {
    collect_ctx* c = (collect_ctx*)vc;
    if (c->count < c->cap) {
        if (c->off) { c->starts[c->count] = c->off[us]; if (c->lens) c->lens[c->count] = c->off[us + ul] - c->off[us]; }
        else { c->starts[c->count] = us; if (c->lens) c->lens[c->count] = ul; }
    }
    ++c->count;
}
int64_t tkzo_split_utf8(int pattern, const uint8_t* text, int64_t n, int64_t* starts,
                        int64_t* lens, int64_t cap) {
    pthread_once(&g_cls_once, cls_init);
    u16buf b; memset(&b, 0, sizeof b);
    int r = utf8_to_units(text, n, &b, pattern == TKZO_PATTERN_O200K);
    if (r) { u16buf_free(&b); return r; }
    utext t = { b.u, b.c, b.n };
    collect_ctx c = { starts, lens, cap, 0, b.off };
    split_units(pattern, &t, collect_piece, &c);
    u16buf_free(&b);
    return c.count;
}
int64_t tkzo_split_utf16(int pattern, const uint16_t* text, int64_t n, int64_t* starts,
                         int64_t* lens, int64_t cap) {
    pthread_once(&g_cls_once, cls_init);
    u16buf b; memset(&b, 0, sizeof b);
    units_to_text(text, n, &b, pattern == TKZO_PATTERN_O200K);
    utext t = { b.u, b.c, b.n };
    collect_ctx c = { starts, lens, cap, 0, b.off };
    split_units(pattern, &t, collect_piece, &c);
    u16buf_free(&b);
    return c.count;
}

/* ------------------------------------------------------------------------------------------ */
/* LruCache<string,int[]>  (Utils/LRUCache.cs:7-136): hash + recency list, capacity bound.      */
/* A pure memo: it cannot change results, it only keeps the CPU baseline from being handicapped. */
/* ------------------------------------------------------------------------------------------ */
#define MEMO_KEY_INLINE 24
#define MEMO_TOK_INLINE 8
typedef struct memo_node {
    uint8_t* key; int32_t klen; int32_t* toks; int32_t ntok;
    uint8_t key_in[MEMO_KEY_INLINE]; int32_t toks_in[MEMO_TOK_INLINE];   /* (short entries live in the node: no allocation per insert) */
    int32_t prev, next;   /* recency list */
    int32_t hnext;        /* hash chain */
    uint64_t h;
} memo_node;
typedef struct {
    memo_node* nodes; int32_t cap, count, head, tail;
    int32_t* buckets; uint32_t nb;
} memo;
static void memo_init(memo* m, int cap) {
    memset(m, 0, sizeof *m); m->cap = cap; m->head = m->tail = -1;
    if (cap <= 0) return;
    m->nodes = (memo_node*)calloc((size_t)cap, sizeof(memo_node));
    m->nb = 1; while (m->nb < (uint32_t)cap * 2) m->nb <<= 1;
    m->buckets = (int32_t*)malloc(m->nb * sizeof(int32_t));
    for (uint32_t i = 0; i < m->nb; ++i) m->buckets[i] = -1;
}
static void memo_node_release(memo_node* x) {
    if (x->key != x->key_in) free(x->key);
    if (x->toks != x->toks_in) free(x->toks);
}
static void memo_free(memo* m) {
    for (int i = 0; i < m->count; ++i) memo_node_release(&m->nodes[i]);
    free(m->nodes); free(m->buckets);
}
static void memo_unlink(memo* m, int i) {
    memo_node* x = &m->nodes[i];
    if (x->prev >= 0) m->nodes[x->prev].next = x->next; else m->head = x->next;
    if (x->next >= 0) m->nodes[x->next].prev = x->prev; else m->tail = x->prev;
}
static void memo_push_front(memo* m, int i) {
    memo_node* x = &m->nodes[i]; x->prev = -1; x->next = m->head;
    if (m->head >= 0) m->nodes[m->head].prev = i;
    m->head = i;
    if (m->tail < 0) m->tail = i;
}
/* Lookup (LRUCache.cs:59-75): a hit moves the entry to the front */
static memo_node* memo_lookup(memo* m, const uint8_t* key, int klen, uint64_t h) {
    if (m->cap <= 0) return NULL;
    for (int i = m->buckets[h & (m->nb - 1)]; i >= 0; i = m->nodes[i].hnext) {
        memo_node* x = &m->nodes[i];
        if (x->h == h && x->klen == klen && memcmp(x->key, key, (size_t)klen) == 0) {
            memo_unlink(m, i); memo_push_front(m, i); return x;
        }
    }
    return NULL;
}
static void memo_hash_remove(memo* m, int i) {
    int32_t* p = &m->buckets[m->nodes[i].h & (m->nb - 1)];
    while (*p != i) p = &m->nodes[*p].hnext;
    *p = m->nodes[i].hnext;
}
/* Add (LRUCache.cs:95-121): evict the least recently used entry when full */
static void memo_add(memo* m, const uint8_t* key, int klen, uint64_t h, const int32_t* toks, int ntok) {
    if (m->cap <= 0) return;
    int i;
    if (m->count < m->cap) i = m->count++;
    else { i = m->tail; memo_unlink(m, i); memo_hash_remove(m, i); memo_node_release(&m->nodes[i]); }
    memo_node* x = &m->nodes[i];
    x->key = klen <= MEMO_KEY_INLINE ? x->key_in : (uint8_t*)malloc((size_t)klen); memcpy(x->key, key, (size_t)klen); x->klen = klen;
    x->toks = ntok <= MEMO_TOK_INLINE ? x->toks_in : (int32_t*)malloc(sizeof(int32_t) * (size_t)ntok); memcpy(x->toks, toks, sizeof(int32_t) * (size_t)ntok); x->ntok = ntok;
    x->h = h; x->hnext = m->buckets[h & (m->nb - 1)]; m->buckets[h & (m->nb - 1)] = i;
    memo_push_front(m, i);
}

/* ------------------------------------------------------------------------------------------ */
/* Encoder object and the per-piece driver  (TikTokenizer.cs:250-274)                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t* lit; int len; int32_t id; } special;
struct tkzo_encoder {
    const tkzo_vocab* v; int pattern;
    memo cache; bpe_scratch sc;
    u16buf ub;            /* the text as the regex engine sees it: reused from document to document */
    special* sp; int nsp;
};
tkzo_encoder* tkzo_encoder_create(const tkzo_vocab* v, int pattern, int cache_size) {
    pthread_once(&g_cls_once, cls_init);
    if (!v || pattern < TKZO_PATTERN_P1 || pattern > TKZO_PATTERN_O200K_DOTNET) return NULL;
    tkzo_encoder* e = (tkzo_encoder*)calloc(1, sizeof *e);
    e->v = v; e->pattern = pattern; memo_init(&e->cache, cache_size);
    return e;
}
void tkzo_encoder_free(tkzo_encoder* e) {
    if (!e) return;
    memo_free(&e->cache); free(e->sc.idx); free(e->sc.rk); u16buf_free(&e->ub);
    for (int i = 0; i < e->nsp; ++i) free(e->sp[i].lit);
    free(e->sp); free(e);
}
int tkzo_encoder_add_special(tkzo_encoder* e, const uint8_t* lit, int len, int32_t id) {
    e->sp = (special*)realloc(e->sp, sizeof(special) * (size_t)(e->nsp + 1));
    e->sp[e->nsp].lit = (uint8_t*)malloc((size_t)len); memcpy(e->sp[e->nsp].lit, lit, (size_t)len);
    e->sp[e->nsp].len = len; e->sp[e->nsp].id = id; ++e->nsp;
    return TKZO_OK;
}

typedef struct { tkzo_encoder* e; const uint8_t* bytes; const int64_t* off; int32_t* out; int64_t cap, n; int64_t err; } enc_ctx;
/* one regex match: memo -> whole-piece rank -> BytePairEncode + memo insert  (TikTokenizer.cs:254-271) */
static void encode_piece_bytes(enc_ctx* c, const uint8_t* pb, int64_t plen) {
    if (c->err) return;
    tkzo_encoder* e = c->e;
    uint64_t h = fnv1a(pb, plen);
    memo_node* hit = plen <= INT_MAX ? memo_lookup(&e->cache, pb, (int)plen, h) : NULL;
    if (hit) {                                                        /* (:254-257) */
        if (c->n + hit->ntok > c->cap) { c->err = TKZO_E_CAPACITY; return; }
        memcpy(c->out + c->n, hit->toks, sizeof(int32_t) * (size_t)hit->ntok); c->n += hit->ntok;
        return;
    }
    int32_t r;
    if (dict_get(e->v, pb, plen, &r)) {                               /* Encoder.TryGetValue (:262) */
        if (c->n + 1 > c->cap) { c->err = TKZO_E_CAPACITY; return; }
        c->out[c->n++] = r; return;                                   /* (:264) */
    }
    int64_t k = bpe_run(e->v, pb, plen, c->out + c->n, c->cap - c->n, &e->sc); /* (:268) */
    if (k < 0) { c->err = k; return; }
    if (plen <= INT_MAX && k <= INT_MAX) memo_add(&e->cache, pb, (int)plen, h, c->out + c->n, (int)k); /* (:270) */
    c->n += k;
}
static void encode_piece_u8(void* vc, int64_t us, int64_t ul) {
    enc_ctx* c = (enc_ctx*)vc;
    encode_piece_bytes(c, c->bytes + c->off[us], c->off[us + ul] - c->off[us]);
}

/* Encode(text, tokenIds, start, end) over a UTF-8 segment */
static int64_t encode_segment_utf8(tkzo_encoder* e, const uint8_t* text, int64_t n, int32_t* out, int64_t cap) {
    if (n == 0) return 0;
    u16buf* b = &e->ub;
    int r = utf8_to_units(text, n, b, e->pattern == TKZO_PATTERN_O200K);
    if (r) return r;
    utext t = { b->u, b->c, b->n };
    enc_ctx c = { e, text, b->off, out, cap, 0, 0 };
    split_units(e->pattern, &t, encode_piece_u8, &c);
    return c.err ? c.err : c.n;
}
int64_t tkzo_encode_utf8(tkzo_encoder* e, const uint8_t* text, int64_t n, int32_t* out, int64_t cap) {
    return encode_segment_utf8(e, text, n, out, cap);
}

/* UTF-16 entry: the regex sees the units as they are; each piece goes through
 * Encoding.UTF8.GetBytes (TikTokenizer.cs:261), which writes EF BF BD for a lone surrogate. */
typedef struct { enc_ctx base; const uint16_t* u; uint8_t* tmp; int64_t tmpcap; } enc16_ctx;
static void encode_piece_u16(void* vc, int64_t es, int64_t el) {
    enc16_ctx* c = (enc16_ctx*)vc;
    const int64_t us = c->base.off[es], ul = c->base.off[es + el] - us;      /* entries -> code units */
    if (c->tmpcap < ul * 3 + 4) { c->tmpcap = ul * 6 + 64; c->tmp = (uint8_t*)realloc(c->tmp, (size_t)c->tmpcap); }
    int64_t o = 0;
    for (int64_t i = us; i < us + ul; ++i) {
        uint32_t x = c->u[i];
        if (x >= 0xD800 && x <= 0xDBFF && i + 1 < us + ul && c->u[i + 1] >= 0xDC00 && c->u[i + 1] <= 0xDFFF) {
            x = 0x10000 + ((x - 0xD800) << 10) + (c->u[i + 1] - 0xDC00); ++i;
        } else if (x >= 0xD800 && x <= 0xDFFF) x = 0xFFFD;
        if (x < 0x80) c->tmp[o++] = (uint8_t)x;
        else if (x < 0x800) { c->tmp[o++] = (uint8_t)(0xC0 | (x >> 6)); c->tmp[o++] = (uint8_t)(0x80 | (x & 0x3F)); }
        else if (x < 0x10000) { c->tmp[o++] = (uint8_t)(0xE0 | (x >> 12)); c->tmp[o++] = (uint8_t)(0x80 | ((x >> 6) & 0x3F)); c->tmp[o++] = (uint8_t)(0x80 | (x & 0x3F)); }
        else { c->tmp[o++] = (uint8_t)(0xF0 | (x >> 18)); c->tmp[o++] = (uint8_t)(0x80 | ((x >> 12) & 0x3F)); c->tmp[o++] = (uint8_t)(0x80 | ((x >> 6) & 0x3F)); c->tmp[o++] = (uint8_t)(0x80 | (x & 0x3F)); }
    }
    encode_piece_bytes(&c->base, c->tmp, o);
}
int64_t tkzo_encode_utf16(tkzo_encoder* e, const uint16_t* text, int64_t n, int32_t* out, int64_t cap) {
    if (n == 0) return 0;
    u16buf b; memset(&b, 0, sizeof b);
    units_to_text(text, n, &b, e->pattern == TKZO_PATTERN_O200K);
    utext t = { b.u, b.c, b.n };
    enc16_ctx c; memset(&c, 0, sizeof c);
    c.base.e = e; c.base.out = out; c.base.cap = cap; c.base.off = b.off; c.u = text;
    split_units(e->pattern, &t, encode_piece_u16, &c);
    u16buf_free(&b); free(c.tmp);
    return c.base.err ? c.base.err : c.base.n;
}

/* EncodeInternal + FindNextSpecialToken + EncodeSpecialToken  (TikTokenizer.cs:141-170,215-241).
 * SpecialTokensRegex is an alternation of the escaped literals in dictionary (registration)
 * order: leftmost match wins, and at one position the FIRST listed literal that matches wins. */
static int special_match_at(const tkzo_encoder* e, const uint8_t* text, int64_t n, int64_t p) {
    for (int i = 0; i < e->nsp; ++i)
        if (e->sp[i].len > 0 && p + e->sp[i].len <= n && memcmp(text + p, e->sp[i].lit, (size_t)e->sp[i].len) == 0) return i;
    return -1;
}
int64_t tkzo_encode_special_utf8(tkzo_encoder* e, const uint8_t* text, int64_t n,
                                 const int32_t* allowed, int n_allowed, int32_t* out, int64_t cap) {
    if (n_allowed == 0) return encode_segment_utf8(e, text, n, out, cap);   /* (:180-183) */
    int64_t cnt = 0, start = 0;
    for (;;) {
        int64_t find = start, hit_pos = -1; int hit = -1;
        for (;;) {                                   /* FindNextSpecialToken (:230-241) */
            int64_t p = find; hit = -1;
            for (; p < n; ++p) { hit = special_match_at(e, text, n, p); if (hit >= 0) break; }
            if (hit < 0) break;
            int ok = 0; for (int k = 0; k < n_allowed; ++k) if (allowed[k] == hit) { ok = 1; break; }
            if (ok) { hit_pos = p; break; }
            find = p + 1;                            /* startFind = nextSpecial.Index + 1 (:238) */
            while (find < n && (text[find] & 0xC0) == 0x80) ++find;
            hit = -1;
        }
        int64_t end = hit >= 0 ? hit_pos : n;
        if (end > start) {                           /* (:150-153) */
            int64_t k = encode_segment_utf8(e, text + start, end - start, out + cnt, cap - cnt);
            if (k < 0) return k;
            cnt += k;
        }
        if (hit >= 0) {                              /* (:155-162), EncodeSpecialToken (:215-220) */
            if (cnt + 1 > cap) return TKZO_E_CAPACITY;
            out[cnt++] = e->sp[hit].id;
            start = hit_pos + e->sp[hit].len;
            if (start >= n) break;
        } else break;
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* Batch helper for the CPU baseline                                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const tkzo_vocab* v; int pattern, cache; const uint8_t* bytes; const int64_t* offs;
    int64_t d0, d1; int32_t* out; int32_t* counts; int64_t total; int64_t err;
} batch_job;
static void* batch_worker(void* vj) {
    batch_job* j = (batch_job*)vj;
    tkzo_encoder* e = tkzo_encoder_create(j->v, j->pattern, j->cache);
    int64_t total = 0;                     /* (kept local: the jobs of neighbouring threads share cache lines) */
    for (int64_t d = j->d0; d < j->d1; ++d) {
        int64_t a = j->offs[d], b = j->offs[d + 1];
        int64_t k = encode_segment_utf8(e, j->bytes + a, b - a, j->out + a, b - a);
        if (k < 0) { j->err = k; break; }
        j->counts[d] = (int32_t)k; total += k;
    }
    j->total = total;
    tkzo_encoder_free(e);
    return NULL;
}
/* The checker at full size: every document is encoded and compared, in place, with the ids another implementation produced for it
 * (want_ids[want_offsets[d] .. want_offsets[d+1])) -- no result arrays, so 10 M documents need nothing beyond the inputs.
 * Returns the number of documents that differ (or a negative error); *first_bad = the lowest such document, *tokens = tokens encoded. */
typedef struct {
    const tkzo_vocab* v; int pattern, cache; const uint8_t* bytes; const int64_t* offs;
    const int32_t* want; const int64_t* want_offs;
    int64_t d0, d1; int64_t bad, first_bad, total, err;
} check_job;
static void* check_worker(void* vj) {
    check_job* j = (check_job*)vj;
    tkzo_encoder* e = tkzo_encoder_create(j->v, j->pattern, j->cache);
    int32_t* buf = NULL; int64_t cap = 0, total = 0, bad = 0, first = -1;
    for (int64_t d = j->d0; d < j->d1; ++d) {
        int64_t a = j->offs[d], b = j->offs[d + 1];
        if (b - a > cap) { free(buf); cap = (b - a) * 2 + 64; buf = (int32_t*)malloc((size_t)cap * sizeof(int32_t)); }
        int64_t k = encode_segment_utf8(e, j->bytes + a, b - a, buf, cap);
        if (k < 0) { j->err = k; break; }
        int64_t wa = j->want_offs[d], wb = j->want_offs[d + 1];
        if (wb - wa != k || memcmp(buf, j->want + wa, (size_t)k * sizeof(int32_t)) != 0) { ++bad; if (first < 0) first = d; }
        total += k;
    }
    free(buf);
    j->total = total; j->bad = bad; j->first_bad = first;
    tkzo_encoder_free(e);
    return NULL;
}
int64_t tkzo_check_batch(const tkzo_vocab* v, int pattern, int cache_size, const uint8_t* bytes, const int64_t* doc_offsets, int64_t n_docs,
                         const int32_t* want_ids, const int64_t* want_offsets, int threads, int64_t* first_bad, int64_t* tokens) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    check_job* jobs = (check_job*)calloc((size_t)threads, sizeof(check_job));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; ++t) {
        check_job* j = &jobs[t];
        j->v = v; j->pattern = pattern; j->cache = cache_size; j->bytes = bytes; j->offs = doc_offsets; j->want = want_ids; j->want_offs = want_offsets;
        j->d0 = n_docs * t / threads; j->d1 = n_docs * (t + 1) / threads; j->first_bad = -1;
    }
    if (threads == 1) check_worker(&jobs[0]);
    else {
        for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, check_worker, &jobs[t]);
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    int64_t bad = 0, total = 0, first = -1, err = 0;
    for (int t = 0; t < threads; ++t) {
        if (jobs[t].err && !err) err = jobs[t].err;
        bad += jobs[t].bad; total += jobs[t].total;
        if (jobs[t].first_bad >= 0 && (first < 0 || jobs[t].first_bad < first)) first = jobs[t].first_bad;
    }
    free(jobs); free(th);
    if (first_bad) *first_bad = first;
    if (tokens) *tokens = total;
    return err ? err : bad;
}

/* How much parallel throughput the host really grants: `threads` threads each run the same register-only loop; returns the wall time in
 * nanoseconds.  (The CPU baseline is reported beside this: on a box whose 256 hardware threads are capped by a CPU quota or shared with
 * other tenants, the all-core figure says more about the quota than about the algorithm.) */
static void* spin_worker(void* p) {
    volatile uint64_t x = (uint64_t)(uintptr_t)p | 1u;
    uint64_t y = x;
    for (uint64_t i = 0; i < 200000000ull; ++i) y = y * 6364136223846793005ull + 1442695040888963407ull;
    x = y;
    return NULL;
}
int64_t tkzo_parallel_probe(int threads) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, spin_worker, (void*)(uintptr_t)(t + 1));
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(th);
    return (int64_t)(b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec);
}

int64_t tkzo_encode_batch(const tkzo_vocab* v, int pattern, int cache_size, const uint8_t* bytes,
                          const int64_t* doc_offsets, int64_t n_docs, int32_t* out,
                          int32_t* out_counts, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    batch_job jobs[256]; pthread_t th[256];
    for (int t = 0; t < threads; ++t) {
        batch_job* j = &jobs[t]; memset(j, 0, sizeof *j);
        j->v = v; j->pattern = pattern; j->cache = cache_size; j->bytes = bytes; j->offs = doc_offsets;
        j->d0 = n_docs * t / threads; j->d1 = n_docs * (t + 1) / threads; j->out = out; j->counts = out_counts;
    }
    if (threads == 1) batch_worker(&jobs[0]);
    else {
        for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
        for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    }
    int64_t total = 0;
    for (int t = 0; t < threads; ++t) { if (jobs[t].err) return jobs[t].err; total += jobs[t].total; }
    return total;
}
