"""Parity checks shared by the emulated (CPU) and the real (GPU) test modules: every function takes
the library under test (`lib`, a tokenizer_amd._native.Library) and the oracle, and compares the two on
the same seeded inputs.  Integer work: the bar is bit-exact."""
import ctypes as C
import os
import random

import pytest

import numpy as np

import regex_crosscheck as RC
from tokenizer_amd import _native as N

SMALL_ALPHAS = {
    "ws": list(" \n\r\t") + ["　", "\x85", "\xa0"] + list("a.1"),
    "dig": list("0123456789") + ["١", "１", " ", "a", ".", "\n"],
    "apo": list("'''sStTrReEvVmMlLdD x.\n 1"),
    "oth": list(".,;!\n\r  a1'") + ["\U0001F600", "⭐", "中"],
    "case": list("aAbB") + ["中", "́", "ǅ", "ʰ", "'", "s", " ", ".", "1", "\n", "/"],
    # pure-ASCII variants: these rows take the mask-algebra fast path of k_pretok_rows
    "a_ws": list(" \n\r\t\x0b\x0c") + list("a.1") + ["\x1c", "\x00"],
    "a_dig": list("0123456789") + list(" a.\n"),
    "a_apo": list("'''sStTrReEvVmMlLdD x.\n 1"),
    "a_oth": list(".,;!\n\r  a1'/\\"),
    "a_mix": list("abcdeflmrstvDELMRSTVxyzXYZ") * 2 + list("0123456789") + list("  \t\n\r") + list("''.,;:!?()[]{}<>=+-*/_#"),
    # long ASCII stretches broken by the occasional multi-byte char: fast <-> general path hand-over
    "a_brk": list("abc 12's.\n") * 6 + ["中", "\U0001F600", "é", "　"],
    # o200k: case transitions, contraction suffixes and their chains, '/' after CR/LF
    "a_case": list("aAbBsStTlLrReEvVdDmM") + list("") + list(" .\n/1"),
}


def gen_runs(rng, n):
    """Single-class runs of a few bytes up to several 4 KiB blocks (digits, spaces, mixed white space with and without
    CR/LF, newlines, letters, punctuation), glued together: what makes the digit phase and the CR/LF look-ahead of the
    position-parallel scanners cross rows and whole blocks."""
    out = []
    total = 0
    while total < n:
        k = rng.randrange(10)
        ln = rng.choice([1, 2, 3, 5, 63, 64, 65, 130, 700, 4096, 4100, 9000, 20000])
        ln = min(ln, n - total)
        if k == 0:
            seg = "".join(rng.choice("0123456789") for _ in range(ln))
        elif k == 1:
            seg = " " * ln
        elif k == 2:
            seg = "".join(rng.choice(" \t") for _ in range(ln))
        elif k == 3:
            seg = " " * (ln - 1) + rng.choice("\n\r")                 # a CR/LF at the very end of a long run
        elif k == 4:
            seg = "".join(rng.choice(" \n") if rng.random() < 0.02 else " " for _ in range(ln))
        elif k == 5:
            seg = "\n" * min(ln, 200)
        elif k == 6:
            seg = rng.choice("aZ") * ln
        elif k == 7:
            seg = rng.choice(["=", "/", ";"]) * ln
        elif k == 8:
            seg = rng.choice(["x", ".", "'s", "\u3000", "\uff11", "9"]) * min(ln, 100)
        else:
            seg = "".join(rng.choice("ab 12.\n") for _ in range(min(ln, 50)))
        out.append(seg)
        total += len(seg)
    return "".join(out)[:n]


def gen_text(rng, kind, n, alpha):
    if kind == "mix":
        return RC.random_text(rng, alpha, n)
    if kind == "runs":
        return gen_runs(rng, n)
    a = SMALL_ALPHAS[kind]
    out = []
    while len(out) < n:
        ch = rng.choice(a)
        rep = 1 if rng.random() < 0.5 else rng.choice([2, 3, 5, 70, 130, 300])
        rep = min(rep, n - len(out))
        if rng.random() < 0.5:
            out.extend([ch] * rep)
        else:
            out.extend(rng.choice(a) for _ in range(rep))
    return "".join(out[:n])


def pack(docs):
    data = np.frombuffer(b"".join(docs), np.uint8) if sum(map(len, docs)) else np.zeros(0, np.uint8)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
    return data, offs


def oracle_bitmap(O, pattern, docs):
    total = sum(len(d) for d in docs)
    bm = np.zeros(total + 1, bool)
    bm[total] = True
    pos = 0
    for d in docs:
        for (a, _n) in O.split_utf8(pattern, d):
            bm[pos + a] = True
        bm[pos] = True
        pos += len(d)
    return bm


def oracle_encode_docs(oenc, docs):
    ids, offs = [], [0]
    for d in docs:
        ids += oenc.encode_bytes(d)
        offs.append(len(ids))
    return ids, offs


def explain_bitmap_diff(got, exp, docs, offs):
    bad = np.nonzero(got != exp)[0]
    p = int(bad[0])
    d = int(np.searchsorted(offs, p, side="right") - 1)
    d = min(d, len(docs) - 1)
    q = p - int(offs[d])
    doc = docs[d]
    return "first of %d differing bits at byte %d (doc %d + %d): got %s expected %s; context %r | %r" % (
        len(bad), p, d, q, bool(got[p]), bool(exp[p]), doc[max(0, q - 16):q], doc[q:q + 16])


def check_pretok(lib, O, vocab, pattern, sequential, seeds, kinds, doc_lens, n_docs_choices=(1, 3, 20)):
    alpha = RC.alphabet()
    enc = N.Encoder(vocab, pattern)
    if sequential:
        enc.set_option(N.OPT_PRETOK_SEQUENTIAL, 1)
    for kind in kinds:
        for seed in seeds:
            rng = random.Random(seed * 7919 + sum(map(ord, kind)) + 31 * pattern)
            docs = [gen_text(rng, kind, rng.choice(doc_lens), alpha).encode("utf-8") for _ in range(rng.choice(n_docs_choices))]
            data, offs = pack(docs)
            got = enc.pretokenize(data, offs)
            exp = oracle_bitmap(O, pattern, docs)
            assert np.array_equal(got, exp), "pattern %d seq=%d kind=%s seed=%d: %s" % (
                pattern, sequential, kind, seed, explain_bitmap_diff(got, exp, docs, offs))


# ---- o200k: alphabets that exercise every flow of the multi-byte block scanner (tkz_block_core_o200k): case transitions through
# runs of Lo/Lm/M chars, marks after punctuation, swallowed '/' runs, contraction suffixes, supplementary-plane letters / digits /
# symbols, ECMAScript white space
O200K_ALPHAS = {
    "cjk": list("\u4e2d\u6587\u65e5\u672c\u8a9e\u304b\u306a\u30ab\u30ca\ud55c\uae00abcXYZ  .,!'s\n"),
    "case": list("aAbB\u4e2d\u02b0\u0301\u00e9 \u00c9's'S ."),
    "mark": list("a\u0301\u0301!!/ \n\u4e2dA'sx"),
    "emoji": ["\U0001F600", "\u200d", "\ufe0f", "\u2764", "a", "A", " ", "\u4e2d", "!", "\U0001D400", "\U00020000", "1", "\uff11"],
    "slash": list("/\n!\u0301a A;*"),
    "chain": list(";\n/*\u0301\u4e2d\r/\n"),          # `;\n/*\n/*...`: every link of such a chain costs the R4 / ABS iteration a round
    "upper": list("AB\u4e2d\u0301. a"),
    "all": list("aAzZ\u4e2d\u6587\u304b\u02b0\u0301\u0300\u00e9 \u00c9's'S'll'LL .,!/\n\r\t 12\uff13\u3000\u00a0\ufeff\u0085\u2028") +
           ["\U0001F600", "\u200d", "\ufe0f", "\U0001D400", "\U00020000", "\U0001D7D8"],
}


def o200k_gen(rng, a, n):
    out = []
    while len(out) < n:
        ch = rng.choice(a)
        r = rng.random()
        rep = 1 if r < 0.6 else (rng.choice([2, 3, 5, 8]) if r < 0.97 else rng.choice([20, 70, 130]))
        if rng.random() < 0.5:
            out.extend([ch] * rep)
        else:
            out.extend(rng.choice(a) for _ in range(rep))
    return "".join(out[:n])


def check_o200k_no_sync_points(lib, O, vocab, pattern=N.O200K):
    """Blocks the o200k block scanners hand on inside text without blanks, digits or line breaks: the sequential kernel finds no sync point
    within its window and takes the block from HBM; next to a document where it does find them.  What makes the char-level scanner hand a
    block on here: a run of '/' that covers the block's CONTEXT row (the last 64 bytes before the block: what flows out of it is unknown);
    the same run inside a block is followed through the rows (the second document)."""
    rng = random.Random(5)
    cjk = "".join(chr(0x4E00 + rng.randrange(2000)) for _ in range(7000))
    # blocks start every 3968 bytes: the runs cover bytes [3904, 3968) and [7872, 7936)
    d0 = cjk[:1290] + "/" * 200 + cjk[1290:1290 + 1256] + "/" * 200 + cjk[3000:4000]
    assert len(d0[:1290].encode("utf-8")) == 3870
    docs = [d0.encode("utf-8"), ("A" + cjk[:2500] + "/" * 130 + "x y 1" + cjk[100:2000]).encode("utf-8")]
    data, offs = pack(docs)
    enc = N.Encoder(vocab, pattern)
    got = enc.pretokenize(data, offs)
    exp = oracle_bitmap(O, pattern, docs)
    assert np.array_equal(got, exp), explain_bitmap_diff(got, exp, docs, offs)
    assert enc.pretok_leftovers()[1] >= 2


def check_o200k_blocks(lib, O, vocab, kinds, seeds, doc_lens=(3000, 9000, 20000), min_handled=None, pattern=N.O200K):
    """Block scanners of o200k vs the oracle's sequential matcher on documents long enough to hold whole 4 KiB blocks.
    Returns (blocks, left over by the ASCII scanner, left over by the multi-byte scanner)."""
    enc = N.Encoder(vocab, pattern)
    tblk = ta = tb = 0
    for kind in kinds:
        for seed in seeds:
            rng = random.Random(seed * 1000 + len(kind))
            docs = [o200k_gen(rng, O200K_ALPHAS[kind], rng.choice(doc_lens)).encode("utf-8") for _ in range(rng.choice([1, 2, 5]))]
            data, offs = pack(docs)
            got = enc.pretokenize(data, offs)
            exp = oracle_bitmap(O, pattern, docs)
            assert np.array_equal(got, exp), "o200k (pattern %d) kind=%s seed=%d: %s" % (pattern, kind, seed, explain_bitmap_diff(got, exp, docs, offs))
            a, b = enc.pretok_leftovers()
            ta += a
            tb += b
            tblk += (len(data) + 3967) // 3968
    return tblk, ta, tb


def check_vocab_keys(lib, O, vocab, ovocab, pattern=N.CL100K):
    """V1/K2: every vocabulary key, presented as one piece, must come back as exactly [rank]."""
    enc = N.Encoder(vocab, pattern)
    ents = ovocab.entries()
    keys = [k for k, _ in ents]
    data, offs = pack(keys)
    ids, ooff = enc.encode_pieces(data, offs)
    assert ids.tolist() == [r for _, r in ents]
    assert np.array_equal(ooff, np.arange(len(keys) + 1))


def random_piece(rng, keys, lens, p_listed=0.5):
    n = rng.choice(lens) if rng.random() < p_listed else rng.randint(1, 24)
    m = rng.random()
    if m < 0.3:
        return bytes(rng.randrange(256) for _ in range(n))
    if m < 0.6:
        return bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ETAOIN") for _ in range(n))
    if m < 0.7:
        return bytes([rng.choice(b"a= 0x")]) * n
    if m < 0.85:
        out = b""
        while len(out) < n:
            out += rng.choice(keys)
        return out[:n]
    return (rng.choice(keys) * (n // 2 + 1))[:n]


def check_pieces(lib, O, vocab, ovocab, seed, rounds, lens, counts, p_listed=0.5):
    """K3: BytePairEncode (+ whole-piece lookup) on arbitrary byte strings, vs the oracle's bpe()."""
    enc = N.Encoder(vocab, N.CL100K)
    keys = [k for k, _ in ovocab.entries()]
    rng = random.Random(seed)
    for it in range(rounds):
        pcs = [random_piece(rng, keys, lens, p_listed) for _ in range(rng.choice(counts))]
        data, offs = pack(pcs)
        ids, ooff = enc.encode_pieces(data, offs)
        exp, eoff = [], [0]
        for p in pcs:
            r = ovocab.rank(p)
            exp += [r] if r >= 0 else ovocab.bpe(p)
            eoff.append(len(exp))
        if ids.tolist() != exp or ooff.tolist() != eoff:
            for i, p in enumerate(pcs):
                g = ids[ooff[i]:ooff[i + 1]].tolist()
                x = exp[eoff[i]:eoff[i + 1]]
                assert g == x and ooff[i] == eoff[i], "round %d piece %d (len %d) %r: got %r expected %r" % (it, i, len(p), p[:40], g[:12], x[:12])
            raise AssertionError("offset mismatch")


def check_batch(lib, O, vocab, ovocab, pattern, seed, rounds, doc_lens, n_docs_choices, kinds=("mix",)):
    """End to end: EncodeBatch vs TikTokenizer.Encode(text, false) restated by the oracle, ids and offsets."""
    alpha = RC.alphabet()
    enc = N.Encoder(vocab, pattern)
    oenc = O.Encoder(ovocab, pattern)
    rng = random.Random(seed)
    for it in range(rounds):
        kind = rng.choice(kinds)
        docs = [gen_text(rng, kind, rng.choice(doc_lens), alpha).encode("utf-8") for _ in range(rng.choice(n_docs_choices))]
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, docs)
        if ids.tolist() != exp or ooff.tolist() != eoff:
            for i in range(len(docs)):
                g = ids[ooff[i]:ooff[i + 1]].tolist()
                x = exp[eoff[i]:eoff[i + 1]]
                assert g == x and ooff[i] == eoff[i], "pattern %d round %d doc %d (len %d): got %r... expected %r..." % (
                    pattern, it, i, len(docs[i]), g[:12], x[:12])
            raise AssertionError("trailing mismatch")


def check_side_by_side(lib, O, vocab, ovocab, pattern=N.CL100K, seed=71, rounds=4):
    """Large batches (above TKZ_OPT_LATENCY_BYTES) on a workspace whose previous batch left few long misses run k_merge_long_q and k_merge_coop on streams of
    their own BESIDE k_merge_short (launch_encode): the sub-tiles' token counts are then summed with atomics from zero, by kernels in any order.  One encoder,
    several batches full of short, long (17..128 bytes), wavefront (129..1024) and giant missed pieces, every document against the oracle; the encoder must
    report that the later batches took that form.  (The small-batch form on the same text is the rest of the suite.)"""
    alpha = RC.alphabet()
    rng = random.Random(seed)
    enc = N.Encoder(vocab, pattern)
    enc.set_option(N.OPT_LATENCY_BYTES, 0)
    oenc = O.Encoder(ovocab, pattern)
    cons = "bcdfghjklmnpqrstvwxz"
    for it in range(rounds):
        docs = []
        for _ in range(rng.choice([3, 30, 120])):
            parts = []
            for _ in range(rng.randint(1, 12)):
                r = rng.random()
                if r < 0.4: parts.append(gen_text(rng, "mix", rng.choice([5, 80, 700]), alpha))
                elif r < 0.7: parts.append(" " + "".join(rng.choice(cons) for _ in range(rng.choice([3, 9, 17, 30, 64, 100, 128]))))
                elif r < 0.9: parts.append(" " + "".join(rng.choice(cons) for _ in range(rng.choice([129, 300, 1024]))))
                else: parts.append(" " + "".join(rng.choice("ab") for _ in range(rng.choice([1025, 2500]))))
            docs.append("".join(parts).encode("utf-8"))
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, docs)
        assert ooff.tolist() == eoff, "round %d: offsets" % it
        assert ids.tolist() == exp, "round %d: ids" % it
    assert enc.side_by_side_batches >= rounds - 1, (enc.side_by_side_batches, rounds)


def check_side_by_side_threads(lib, O, vocab, ovocab, pattern=N.CL100K, seed=73, threads=3, rounds=3):
    """Several callers at once on ONE encoder (a workspace each), every batch large enough for the side-by-side form: only one batch in the process takes it
    at a time (the others keep the serial form), none may hang, every document equals the oracle's."""
    import threading
    alpha = RC.alphabet()
    enc = N.Encoder(vocab, pattern)
    enc.set_option(N.OPT_LATENCY_BYTES, 0)
    oenc = O.Encoder(ovocab, pattern)
    cons = "bcdfghjklmnpqrstvwxz"
    jobs = []
    for t in range(threads):
        rng = random.Random(seed + t)
        mine = []
        for _ in range(rounds):
            docs = []
            for _ in range(rng.choice([8, 40])):
                parts = [gen_text(rng, "mix", rng.choice([20, 300]), alpha)]
                for _ in range(rng.randint(0, 5)):
                    parts.append(" " + "".join(rng.choice(cons) for _ in range(rng.choice([5, 20, 70, 140, 400]))))
                docs.append("".join(parts).encode("utf-8"))
            mine.append((docs, oracle_encode_docs(oenc, docs)))
        jobs.append(mine)
    errs = []

    def work(mine):
        try:
            for docs, (exp, eoff) in mine:
                data, offs = pack(docs)
                ids, ooff = enc.encode_batch(data, offs)
                if ids.tolist() != exp or ooff.tolist() != eoff:
                    errs.append("mismatch")
        except Exception as ex:      # noqa: BLE001
            errs.append(repr(ex))
    ths = [threading.Thread(target=work, args=(m,)) for m in jobs]
    for th in ths: th.start()
    for th in ths: th.join(timeout=300)
    assert not any(th.is_alive() for th in ths), "a caller hangs"
    assert not errs, errs
    assert enc.side_by_side_batches >= 1


def check_dense_region(lib, O, vocab, ovocab, pattern=N.CL100K, seed=17):
    """The packed region for the tokens of merged short pieces (4096 per group of 16 sub-tiles): groups that fill it exactly, overflow it
    (the rest waits in tmp) and stay far below it, next to each other, with long misses in between."""
    rng = random.Random(seed)
    enc = N.Encoder(vocab, pattern)
    oenc = O.Encoder(ovocab, pattern)
    cons = "bcdfghjklmnpqrstvwxz"

    def gibberish(n):                      # 2..9-letter words that no vocabulary holds: every piece is merged, ~1 token per 1.5 bytes
        out = []
        while sum(map(len, out)) < n:
            out.append(" " + "".join(rng.choice(cons) for _ in range(rng.randint(2, 9))))
        return "".join(out)[:n]

    def english(n):
        return ("the quick brown fox jumps over the lazy dog and runs away " * (n // 50 + 1))[:n]

    docs = []
    for it in range(6):
        parts = []
        for _ in range(rng.randint(2, 10)):
            k = rng.random()
            parts.append(gibberish(rng.choice([300, 1500, 4096, 6000])) if k < 0.5 else english(rng.choice([200, 3000, 5000])) if k < 0.85
                         else " " + "".join(rng.choice(cons) for _ in range(rng.choice([17, 40, 200]))))
        docs.append("".join(parts).encode("utf-8"))
    data, offs = pack(docs)
    ids, ooff = enc.encode_batch(data, offs)
    exp, eoff = oracle_encode_docs(oenc, docs)
    assert ooff.tolist() == eoff and ids.tolist() == exp


def check_reserve(lib, O, vocab, ovocab, pattern=N.CL100K, seed=13):
    """tkz_encoder_reserve: the workspace of a stated batch size is allocated by the call, and batches up to that size allocate nothing more (the
    reference pays construction costs in CreateTokenizer, TokenizerBuilder.cs:210-213, not in the first Encode)."""
    rng = random.Random(seed)
    alpha = RC.alphabet()
    enc = N.Encoder(vocab, pattern)
    enc.set_option(N.OPT_PROMOTE, 0)                        # (a promotion builds new key-table images: megabytes that are not workspace)
    w0 = enc.workspace_bytes
    with pytest.raises(N.TkzError) as ei:
        enc.reserve(-1, 10)
    assert ei.value.code == N.E_ARG
    enc.reserve(2_000_000, 4_000)
    w1 = enc.workspace_bytes
    assert w1 - w0 > 7 * 2_000_000, (w0, w1)                # ~7.8 bytes per input byte + staging
    oenc = O.Encoder(ovocab, pattern)
    for total in (300_000, 1_500_000):                      # (above the single-launch path; a learning window allocates its counters: reserved too)
        # (ordinary prose: a piece per ~5 bytes.  Text with a piece per byte needs more piece records than a workspace starts with -- a piece per 3 bytes --
        #  and lists of misses longer than 64: those two still grow, once, inside the call that meets such text)
        words = ["the", "and", "with", "from", "this", "that", "have", "will", "token", "encoder", "value", "return", "12", ",", "."]
        docs = []
        while sum(map(len, docs)) < total:
            docs.append(" ".join(rng.choice(words) for _ in range(rng.choice([40, 400, 1800]))).encode("utf-8"))
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        e, eo = oracle_encode_docs(oenc, docs)
        assert ids.tolist() == e and ooff.tolist() == eo
    assert enc.workspace_bytes == w1, (w1, enc.workspace_bytes)
    enc.reserve(1_000_000, 100)                             # smaller than what is there: nothing happens
    assert enc.workspace_bytes == w1


def check_adaptation(lib, O, vocab, ovocab, monkeypatch, pattern=N.CL100K, seed=71):
    """TKZ_OPT_ADAPT: the cache follows the text.  An encoder learns text A, the text drifts to B (other words: the share of pieces that miss the key
    tables rises), the encoder drops its promotions, empties the memo, learns B -- A's pieces are promoted no longer (eviction) --, and back again.
    Batches smaller than the learning window add up to one.  Replaced table images are freed once no call is in flight; the ids are the oracle's
    throughout.  (The thresholds are megabytes; $TKZ_ADAPT_* brings them down to the size of these batches.)"""
    monkeypatch.setenv("TKZ_ADAPT_SETTLE_BYTES", "100000")
    monkeypatch.setenv("TKZ_ADAPT_MIN_BYTES", "300000")
    cons, vow = "bcdfghjklmnpqrstvwxz", "aeiou"

    def lexicon(r, n):
        return ["".join(r.choice(cons) + r.choice(vow) for _ in range(r.randint(2, 6))) + r.choice(["", "s", "ed", "ing"]) for _ in range(n)]
    lex_a, lex_b = lexicon(random.Random(seed + 1), 300), lexicon(random.Random(seed + 2), 300)
    known = ["the", "and", "with", "from", "this", "that", "have", "will"]           # (vocabulary words: whole-piece hits whatever is promoted)

    def batch(lex, r, nbytes=160_000, hit_share=0.5):
        docs, size = [], 0
        while size < nbytes:
            words = []
            n = r.choice([300, 2000, 9000])
            while sum(map(len, words)) < n:
                words.append(r.choice([" ", " ", "\n", ", "]) + (r.choice(known) if r.random() < hit_share else r.choice(lex)))
            docs.append("".join(words).encode("utf-8"))
            size += len(docs[-1])
        return docs
    oenc = O.Encoder(ovocab, pattern)
    enc = N.Encoder(vocab, pattern)
    enc.set_option(N.OPT_PROMOTE_MIN_BYTES, 100_000)
    rr = random.Random(seed + 3)
    mem0 = None

    def run(docs):
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        e, eo = oracle_encode_docs(oenc, docs)
        assert ids.tolist() == e and ooff.tolist() == eo
        return enc.adapt_stats()
    st = run(batch(lex_a, rr))                            # the learning window (one batch): promoted at its end
    assert st["promotions"] == 1 and st["promoted_pieces"] > 50 and st["relearns"] == 0, st
    promoted_a = st["promoted_pieces"]
    for _ in range(3):
        st = run(batch(lex_a, rr))
    assert st["settled_miss_share"] is not None and st["relearns"] == 0 and st["retired_images"] == 0, st
    settled_a = st["settled_miss_share"]
    mem0 = enc.workspace_bytes
    # the text drifts: other words.  The miss share rises; once it has, the encoder learns again
    seen_relearn = None
    for i in range(6):
        st = run(batch(lex_b, rr))
        if st["relearns"] == 1 and seen_relearn is None:
            seen_relearn = i
            assert st["promoted_pieces"] == 0, st          # every promotion dropped: the next window starts from the vocabulary alone
    assert seen_relearn is not None and seen_relearn <= 3, st
    assert st["promotions"] >= 2 and st["promoted_pieces"] > 50, st
    # eviction: text A's pieces are not promoted any more -- they miss the key tables again
    enc.set_option(N.OPT_PROMOTE, 0)                      # (nothing automatic while the two texts are probed: text A would be a drift of its own)
    enc.set_option(N.OPT_PIECE_STATS, 1)
    enc.piece_stats(reset=True)
    probe_a = batch(lex_a, random.Random(seed + 9), 150_000)
    run(probe_a)
    ps_a = enc.piece_stats(reset=True)
    run(batch(lex_b, random.Random(seed + 10), 150_000))
    ps_b = enc.piece_stats(reset=True)
    enc.set_option(N.OPT_PIECE_STATS, 0)
    enc.set_option(N.OPT_PROMOTE, 1)
    assert ps_a["whole_piece_hit_rate"] < ps_b["whole_piece_hit_rate"] - 0.15, (ps_a, ps_b)
    # ... and back to A: a second re-learn
    for i in range(8):
        st = run(batch(lex_a, rr))
    assert st["relearns"] >= 2 and st["promoted_pieces"] > 50, st
    run(batch(lex_a, rr))
    assert enc.adapt_stats()["retired_images"] == 0       # freed when the call that followed the swap returned
    assert enc.workspace_bytes <= mem0 + (4 << 20), (mem0, enc.workspace_bytes)
    # a change of text that falls BETWEEN two installs (round 6): the second round's promotion lands on the change, so the miss share has no settled level to leave --
    # it settles again on text B, ABOVE the level before that install although a round only adds pieces: a drift all the same ($TKZ_ADAPT_ROUND_BYTES: the rounds'
    # spacing, a gigabyte by default)
    monkeypatch.setenv("TKZ_ADAPT_ROUND_BYTES", "600000")
    enc4 = N.Encoder(vocab, pattern)
    enc4.set_option(N.OPT_PROMOTE_MIN_BYTES, 100_000)
    r4 = random.Random(seed + 21)

    def run4(lex):
        docs = batch(lex, r4)
        data, offs = pack(docs)
        ids, ooff = enc4.encode_batch(data, offs)
        e, eo = oracle_encode_docs(oenc, docs)
        assert ids.tolist() == e and ooff.tolist() == eo
        return enc4.adapt_stats()
    st = run4(lex_a)
    while st["promotions"] < 2:                           # (batches of ~160 KB: the first window, the settled level, then the second round 600 KB after the first began)
        assert st["relearns"] == 0 and enc4.adapt_stats()["promotions"] >= 1, st
        st = run4(lex_a)
    # (the second install has just landed -- on its thread: as a rule no level has settled yet, st["settled_miss_share"] is None -- and the text changes here)
    for i in range(6):
        st = run4(lex_b)
        if st["relearns"]:
            break
    assert st["relearns"] == 1 and i <= 3, (i, st)
    monkeypatch.delenv("TKZ_ADAPT_ROUND_BYTES")
    # small batches add up to a learning window (the single-launch path -- up to 128 KiB -- keeps no statistics: these are just above it)
    enc2 = N.Encoder(vocab, pattern)
    enc2.set_option(N.OPT_PROMOTE_MIN_BYTES, 600_000)
    enc3 = N.Encoder(vocab, pattern)
    enc3.set_option(N.OPT_PROMOTE_MIN_BYTES, 600_000)
    enc3.set_option(N.OPT_ADAPT, 0)                       # round 5's rule: only a batch of that size learns
    r5 = random.Random(seed + 5)
    for i in range(5):
        docs = batch(lex_a, r5, 140_000)
        data, offs = pack(docs)
        e, eo = oracle_encode_docs(oenc, docs)
        for en in (enc2, enc3):
            ids, ooff = en.encode_batch(data, offs)
            assert ids.tolist() == e and ooff.tolist() == eo
        if i < 3:
            assert enc2.adapt_stats()["promotions"] == 0 and enc2.adapt_stats()["learning_window_bytes"] > 0
    assert enc2.adapt_stats()["promotions"] == 1 and enc2.adapt_stats()["promoted_pieces"] > 50, enc2.adapt_stats()
    assert enc3.adapt_stats()["promotions"] == 0 and enc3.adapt_stats()["promoted_pieces"] == 0


def check_memo_refresh(lib, O, vocab, ovocab, monkeypatch, pattern=N.CL100K, seed=77):
    """The memo does not evict -- an entry is never replaced --, so text with many one-off pieces fills it and whatever the text turns into finds it closed.
    TKZ_OPT_ADAPT (round 6): a learning round that reads the memo back three quarters full has it emptied at the start of the next learning window (the
    reference's LRUCache evicts, LRUCache.cs:79-88).  A memo of 4,096 slots, text A with 6,000 different missed pieces, then text B with 250 that repeat: with
    the rule B's pieces are answered by the memo (or promoted), without it (TKZ_OPT_ADAPT 0) nearly every occurrence is merged again.  Ids = the oracle's throughout."""
    monkeypatch.setenv("TKZ_MEMO_SLOTS_LOG2", "12")
    monkeypatch.setenv("TKZ_ADAPT_ROUND_BYTES", "300000")
    monkeypatch.setenv("TKZ_ADAPT_SETTLE_BYTES", "100000")
    monkeypatch.setenv("TKZ_ADAPT_MIN_BYTES", "300000")
    cons, vow = "bcdfghjklmnpqrstvwxz", "aeiou"
    r0 = random.Random(seed)

    def word(r):
        return "".join(r.choice(cons) + r.choice(vow) for _ in range(r.randint(3, 5))) + r.choice(["", "s", "ed"])
    one_off = [word(r0) for _ in range(6000)]
    lex_b = [word(random.Random(seed + 1)) + "q" for _ in range(250)]
    oenc = O.Encoder(ovocab, pattern)

    def batch(lex, r, nbytes=170_000):
        docs, size = [], 0
        while size < nbytes:
            words, n = [], r.choice([2000, 9000])
            while sum(map(len, words)) < n:
                words.append(r.choice([" ", " ", "\n"]) + (r.choice(["the", "and", "with"]) if r.random() < 0.3 else r.choice(lex)))
            docs.append("".join(words).encode("utf-8"))
            size += len(docs[-1])
        return docs
    merged = {}
    for adapt in (1, 0):
        enc = N.Encoder(vocab, pattern)
        assert enc.memo_slots == 4096
        enc.set_option(N.OPT_PROMOTE_MIN_BYTES, 100_000)
        enc.set_option(N.OPT_ADAPT, adapt)
        rr = random.Random(seed + 5)

        def run(docs):
            data, offs = pack(docs)
            ids, ooff = enc.encode_batch(data, offs)
            e, eo = oracle_encode_docs(oenc, docs)
            assert ids.tolist() == e and ooff.tolist() == eo
            return enc.adapt_stats()
        for _ in range(6):                                 # text A: the memo fills in the first batch; two rounds (the second one starts on an emptied memo and fills it again)
            st = run(batch(one_off, rr))
        assert st["promotions"] >= (2 if adapt else 1), st
        for _ in range(10):                                # text B: with TKZ_OPT_ADAPT its next window empties the memo once more -- and B's pieces get in
            st = run(batch(lex_b, rr))
        enc.set_option(N.OPT_PROMOTE, 0)
        enc.set_option(N.OPT_PIECE_STATS, 1)
        enc.piece_stats(reset=True)
        run(batch(lex_b, random.Random(seed + 9)))
        merged[adapt] = enc.piece_stats(reset=True)["merged_short"]
    print("merged_short with / without the rule:", merged)
    assert merged[1] * 2 < merged[0], merged


def check_piece_memo(lib, O, vocab, ovocab, pattern=N.CL100K, seed=23):
    """The piece memo (TKZ_OPT_PIECE_MEMO): the same ids with the memo off, empty, filled by an earlier batch of the same text (every
    short miss a hit), and filled by OTHER text (hits and misses mixed, slots already taken by other pieces)."""
    rng = random.Random(seed)
    cons, vow = "bcdfghjklmnpqrstvwxz", "aeiou"

    def words(n, r):           # pronounceable non-words: they miss the vocabulary, merge to 2..5 tokens, and repeat
        lex = ["".join(r.choice(cons) + r.choice(vow) for _ in range(r.randint(2, 4))) + r.choice(["", "s", "ed", "ing"]) for _ in range(300)]
        out = []
        while sum(map(len, out)) < n:
            out.append(r.choice([" ", " ", "\n", " the "]) + r.choice(lex))
        return "".join(out)[:n]

    docs_a = [words(rng.choice([500, 3000, 9000]), random.Random(seed + 1)).encode("utf-8") for _ in range(6)]
    docs_b = [words(rng.choice([500, 3000, 9000]), random.Random(seed + 2)).encode("utf-8") for _ in range(6)]
    oenc = O.Encoder(ovocab, pattern)
    exp_a, eoff_a = oracle_encode_docs(oenc, docs_a)
    exp_b, eoff_b = oracle_encode_docs(oenc, docs_b)
    enc = N.Encoder(vocab, pattern)

    def run(docs, exp, eoff, what):
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        assert ooff.tolist() == eoff and ids.tolist() == exp, what

    enc.set_option(N.OPT_PIECE_MEMO, 0)
    run(docs_a, exp_a, eoff_a, "memo off")
    enc.set_option(N.OPT_PIECE_MEMO, 2)
    run(docs_a, exp_a, eoff_a, "memo empty")
    run(docs_a, exp_a, eoff_a, "memo filled by the same text")
    run(docs_b, exp_b, eoff_b, "memo filled by other text")
    run(docs_a, exp_a, eoff_a, "memo filled by both")
    enc.set_option(N.OPT_PIECE_MEMO, 0)
    run(docs_b, exp_b, eoff_b, "memo off again")


def check_promotion(lib, O, vocab, ovocab, pattern=N.CL100K, seed=61):
    """Promoted pieces (TKZ_OPT_PROMOTE): hot memo entries moved into the SHORT / MID key tables with their <= 4 tokens in place of a rank.  Same ids
    as the oracle before, after, and after dropping them again -- on the batch path, the single-launch path, the piece-granular entry and pieces
    handed over one by one; by hand (value 2) and automatically (a learning batch counts the memo's hits, the batch's end promotes)."""
    rng = random.Random(seed)
    cons, vow = "bcdfghjklmnpqrstvwxz", "aeiou"
    r1 = random.Random(seed + 1)
    # pronounceable non-words of 4..16 bytes (the last ones go to the MID table: 13..16), repeated: they miss the vocabulary and fill the memo
    lex = ["".join(r1.choice(cons) + r1.choice(vow) for _ in range(r1.randint(2, 7))) + r1.choice(["", "s", "ed", "ing"]) for _ in range(400)]
    lex = [w for w in lex if len(w) <= 15] + ["zqxjkvbwpfyhgm", "qzqzqzqzqzqzqz", "xv", "ĳĳĳ", "żółć"]

    def words(n, r, lexicon):
        out = []
        while sum(map(len, out)) < n:
            out.append(r.choice([" ", " ", "\n", " the ", ", ", " 12 "]) + r.choice(lexicon))
        return "".join(out)
    # (both batches are too large for the single-launch path, which keeps no statistics)
    docs_a = [words(n, random.Random(seed + 10 + i), lex).encode("utf-8") for i, n in enumerate([400, 3000, 9000, 40000, 60000, 30000, 3000, 500])]
    docs_b = [words(n, random.Random(seed + 30 + i), lex[::2] + ["brandnewword", "anotherone"]).encode("utf-8") for i, n in enumerate([600, 5000, 30000, 50000, 40000, 20000])]
    small = [words(300, random.Random(seed + 50 + i), lex).encode("utf-8") for i in range(5)] + [b""]
    oenc = O.Encoder(ovocab, pattern)
    exp = {id(d): oracle_encode_docs(oenc, d) for d in (docs_a, docs_b, small)}
    enc = N.Encoder(vocab, pattern)
    enc.set_option(N.OPT_PROMOTE, 0)                     # (nothing automatic in the first half of the test)

    def run(docs, what):
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        e, eo = exp[id(docs)]
        assert ooff.tolist() == eo and ids.tolist() == e, what

    def stats(docs):
        enc.set_option(N.OPT_PIECE_STATS, 1)
        enc.piece_stats(reset=True)
        run(docs, "statistics run")
        st = enc.piece_stats(reset=True)
        enc.set_option(N.OPT_PIECE_STATS, 0)
        return st
    run(docs_a, "before: memo empty")
    before = stats(docs_a)
    assert before["promoted_pieces_in_tables"] == 0 and before["short_misses"] > 100
    enc.set_option(N.OPT_PROMOTE, 2)                     # by hand: whatever the memo holds
    after = stats(docs_a)
    assert after["promoted_pieces_in_tables"] > 50, after
    # the pieces the memo answered are whole-piece hits now (what is left are pieces of more than 4 tokens, which no memo entry holds)
    assert after["short_misses"] <= before["short_misses"] - int(0.9 * before["memo_hits"]) and before["memo_hits"] > 1000, (before, after)
    assert after["pieces"] == before["pieces"]
    run(docs_a, "promoted, same text")
    run(docs_b, "promoted, other text (promoted pieces beside pieces that are not)")
    small_before = enc.small_path_calls()
    run(small, "promoted, single-launch path")
    for d in small[:3]:
        data, offs = pack([d])
        ids, ooff = enc.encode_batch(data, offs)
        assert ids.tolist() == oenc.encode_bytes(d)
    assert enc.small_path_calls()[0] > small_before[0]
    # piece granularity, and promoted pieces handed over as pieces
    data, offs = pack(docs_a[:3])
    ids, dpo, pbo, pto = enc.encode_batch_pieces(data, offs)
    assert ids.tolist() == oracle_encode_docs(oenc, docs_a[:3])[0]
    pcs = [w.encode("utf-8") for w in lex[:200]] + [(" " + w).encode("utf-8") for w in lex[:200]]
    data, offs = pack(pcs)
    ids, ooff = enc.encode_pieces(data, offs)
    want = []
    for q in pcs:
        want += ovocab.bpe(q) if ovocab.rank(q) < 0 else [ovocab.rank(q)]
    assert ids.tolist() == want
    # a second promotion by hand adds what the memo has learnt since (docs_b's new words), never a piece twice
    n1 = after["promoted_pieces_in_tables"]
    run(docs_b, "fill")
    enc.set_option(N.OPT_PROMOTE, 2)
    n2 = stats(docs_b)["promoted_pieces_in_tables"]
    assert n2 >= n1
    run(docs_a, "after the second promotion")
    # the cap
    enc.set_option(N.OPT_PROMOTE, 3)                     # dropped: the vocabulary's own tables again
    dropped = stats(docs_a)
    assert dropped["promoted_pieces_in_tables"] == 0 and dropped["short_misses"] == before["short_misses"], (before, dropped)
    enc.set_option(N.OPT_PROMOTE_CAP, 7)
    enc.set_option(N.OPT_PROMOTE, 2)
    assert stats(docs_a)["promoted_pieces_in_tables"] == 7
    run(docs_b, "capped")
    # ---- automatic: a learning batch (the first one of at least PROMOTE_MIN_BYTES) counts hits, its end promotes the entries that were hit ----
    enc2 = N.Encoder(vocab, pattern)
    enc2.set_option(N.OPT_PROMOTE_MIN_BYTES, 20000)
    data, offs = pack(small)
    enc2.encode_batch(data, offs)                        # too small: not a learning batch
    enc2.set_option(N.OPT_PIECE_STATS, 1)
    big = docs_a + docs_b
    data, offs = pack(big)
    ids, ooff = enc2.encode_batch(data, offs)            # learns (its memo starts empty: the hits are those of this very batch), promotes at its end
    e_a, e_b = exp[id(docs_a)][0], exp[id(docs_b)][0]
    assert ids.tolist() == e_a + e_b
    st1 = enc2.piece_stats(reset=True)
    assert st1["promoted_pieces_in_tables"] > 20, st1
    ids, ooff = enc2.encode_batch(data, offs)
    assert ids.tolist() == e_a + e_b
    st2 = enc2.piece_stats(reset=True)
    assert st2["short_misses"] <= st1["short_misses"] - int(0.8 * st1["memo_hits"]) and st1["memo_hits"] > 1000, (st1, st2)      # (the hits are sampled: one group in eight)
    # pieces of 17..28 bytes that repeat and merge into at most 4 tokens (k_merge_long logs them during the learning batch) are promoted too: the
    # indentation runs of source code -- "\n" + 16..27 blanks -- under a table that holds runs of blanks as keys
    lines = []
    r4 = random.Random(seed + 4)
    for _ in range(7000):                                 # (more than the single-launch path takes: it keeps no statistics)
        lines.append("\n" + " " * r4.choice([16, 19, 20, 23, 24, 27, 31, 40]) + r4.choice(["return x", "if a:", "pass", "x = 1"]))
    code = "".join(lines).encode()
    cdocs = [code[i:i + 30000] for i in range(0, len(code), 30000)]
    cexp, ceoff = oracle_encode_docs(oenc, cdocs)
    enc4 = N.Encoder(vocab, pattern)
    enc4.set_option(N.OPT_PROMOTE_MIN_BYTES, 20000)
    enc4.set_option(N.OPT_PIECE_STATS, 1)
    cdata, coffs = pack(cdocs)
    long_misses = []
    for rep in range(2):
        enc4.piece_stats(reset=True)
        ids, ooff = enc4.encode_batch(cdata, coffs)
        assert ids.tolist() == cexp and ooff.tolist() == ceoff, rep
        long_misses.append(enc4.piece_stats(reset=True)["long_misses"])
    # (pattern 1 cuts "\n" + n - 1 blanks -- the last blank goes with the word --; cl100k cuts the "\n" off first, o200k too)
    few_tokens = sum(1 for n in (16, 19, 20, 23, 24, 27) if ovocab.rank(b"\n" + b" " * (n - 1)) < 0 and len(ovocab.bpe(b"\n" + b" " * (n - 1))) <= 4)
    if few_tokens >= 3 and pattern == N.P1:
        assert long_misses[1] < long_misses[0] * 0.8, long_misses
    # the second automatic round waits for a gigabyte more: no further promotion here, and the same ids
    ids, ooff = enc2.encode_batch(data, offs)
    assert ids.tolist() == e_a + e_b and enc2.piece_stats()["promoted_pieces_in_tables"] == st1["promoted_pieces_in_tables"]
    # automatic promotion switched off: nothing is promoted
    enc3 = N.Encoder(vocab, pattern)
    enc3.set_option(N.OPT_PROMOTE, 0)
    enc3.set_option(N.OPT_PROMOTE_MIN_BYTES, 20000)
    enc3.set_option(N.OPT_PIECE_STATS, 1)
    for _ in range(2):
        ids, ooff = enc3.encode_batch(data, offs)
        assert ids.tolist() == e_a + e_b
    assert enc3.piece_stats()["promoted_pieces_in_tables"] == 0


def check_runtime_overrides(lib, O, vocab, ovocab, seed=71):
    """The host's runtime defines the split (TikTokenizer.cs:77 compiles the pattern with the RUNNING process's regex engine): a Unicode class table
    handed over by the host (tkz_encoder_set_unicode_classes) and cl100k's (?i:...) with .NET >= 7's case-equivalence tables
    (TKZ_OPT_CASE_EQUIVALENCE).  The device (parallel and sequential scanners, whole encode) against the oracle given the same overrides; the case mode
    also against the `regex` engine, whose (?i) folds U+017F onto s like .NET >= 7 does."""
    import regex_crosscheck as RC
    rng = random.Random(seed)
    try:
        # ---- a perturbed class table: 60+ code points reclassified -- CJK, digits, white space, marks, symbols; supplementary-plane chars for o200k ----
        enc0 = N.Encoder(vocab, N.O200K)
        base = enc0.unicode_classes(0, 0x110000).copy()
        del enc0
        moved = {}
        for cp in range(0x4E00, 0x4E10):
            moved[cp] = 0                      # 16 CJK ideographs: letters -> other
        for cp in range(0x6F22, 0x6F32):
            moved[cp] = 7                      # ... and 16 more -> digits
        moved.update({0x0663: 5, 0x0664: 0, 0xFF11: 0, 0xFF12: 2, 0x00B2: 0,             # digits -> letter / other
                      0x3000: 0, 0x00A0: 5, 0x200B: 8, 0x2603: 8, 0x0085: 0,              # white space <-> other / letter
                      0x0301: 5, 0x0300: 0, 0x093F: 7,                                    # marks
                      0x2603 + 1: 5, 0x2764: 1, 0x2B50: 2, 0x00E9: 1, 0x00C9: 2, 0x0416: 2, 0x0436: 1, 0x01C5: 6, 0x02B0: 0, 0x3042: 7, 0x30AB: 0, 0xAC00: 6})
        for cp in (0x1F600, 0x1F468, 0x1D400, 0x1D7CE, 0x10428, 0x20000, 0x2A700, 0xE0100, 0xE0101):
            moved[cp] = {0: 5, 5: 0, 1: 7, 2: 7, 7: 5, 6: 1}.get(int(base[cp]), 0)         # supplementary plane (and plane 14: beyond the direct part of the table)
        assert len(moved) >= 60
        full = base.copy()
        for cp, c in moved.items():
            full[cp] = c
        chars = [chr(cp) for cp in moved] + list("abcXYZ019  \n\r\t'.,;-_/") + ["'s", "'S", "'re", " ", "\n\n", "12345", "é", "中", "あ"]

        def text(n):
            out = []
            while sum(map(len, out)) < n:
                ch = rng.choice(chars)
                out.append(ch * rng.choice([1, 1, 1, 2, 3, 5]))
            return "".join(out)
        docs = [text(rng.choice([40, 300, 2500, 9000])).encode("utf-8") for _ in range(14)] + [b"", text(70000).encode("utf-8")]
        data, offs = pack(docs)
        for pattern, table in ((N.P1, full[:65536]), (N.CL100K, full[:65536]), (N.O200K_DOTNET, full[:65536]), (N.O200K, full)):
            enc = N.Encoder(vocab, pattern)
            before = enc.pretokenize(data, offs)
            enc.set_unicode_classes(table)
            O.set_unicode_classes(table)
            # the device's table IS the host's (ASCII and the surrogate code units aside), read back from the device
            got = enc.unicode_classes(0, 0x110000)
            want = base.copy(); want[:len(table)] = table; want[:128] = base[:128]; want[0xD800:0xE000] = 0
            assert np.array_equal(got, want), pattern
            oenc = O.Encoder(ovocab, pattern)
            exp, eoff = oracle_encode_docs(oenc, docs)
            for seq in (0, 1):
                enc.set_option(N.OPT_PRETOK_SEQUENTIAL, seq)
                bits = enc.pretokenize(data, offs)
                want_bits = oracle_bitmap(O, pattern, docs)
                assert np.array_equal(bits[:len(want_bits)], want_bits), (pattern, seq, int(np.flatnonzero(bits[:len(want_bits)] != want_bits)[0]))
                ids, ooff = enc.encode_batch(data, offs)
                assert ooff.tolist() == eoff and ids.tolist() == exp, (pattern, seq)
            assert not np.array_equal(before, bits)          # (the perturbed table cuts this text differently)
            # small batches (the single-launch path reads the same table)
            for d in docs[:4]:
                assert enc.encode_utf8(d) == oenc.encode_bytes(d)
            # the built-in table again
            enc.set_unicode_classes(None)
            O.set_unicode_classes(None)
            assert np.array_equal(enc.pretokenize(data, offs), before) and np.array_equal(enc.unicode_classes(0, 0x110000), base)
        with pytest.raises(N.TkzError):
            N.Encoder(vocab, N.CL100K).set_unicode_classes(np.zeros(1000, np.uint8))          # neither the BMP nor every code point
        with pytest.raises(N.TkzError):
            N.Encoder(vocab, N.CL100K).set_unicode_classes(np.full(65536, 9, np.uint8))        # a class code that does not exist
        # ---- cl100k's (?i:...) with .NET >= 7's case-equivalence tables: 'ſ is 's ----
        O.set_case_equivalence(True)
        enc = N.Encoder(vocab, N.CL100K)
        oenc = O.Encoder(ovocab, N.CL100K)
        hand = ["it'ſ fine", "it'ſabc", "'ſ", "x'ſ", " 'ſa", "''ſa", "a'ſ'ſ's'S", "a'ſ", "a'ſ\n", "don'ſt", "a'ſ1", "a'ſ é", "é'ſé", "a'\u017f\u0301b", "a'ſ" + "z" * 200]
        alpha = RC.alphabet() + ["ſ", "'ſ", "'ſ", "K"]
        cdocs = [t.encode("utf-8") for t in hand] + [RC.random_text(rng, alpha, rng.choice([30, 400, 3000])).encode("utf-8") for _ in range(40)]
        cdata, coffs = pack(cdocs)
        plain_bits = enc.pretokenize(cdata, coffs)
        enc.set_option(N.OPT_CASE_EQUIVALENCE, 1)
        cexp, ceoff = oracle_encode_docs(oenc, cdocs)
        for seq in (0, 1):
            enc.set_option(N.OPT_PRETOK_SEQUENTIAL, seq)
            bits = enc.pretokenize(cdata, coffs)
            want_bits = oracle_bitmap(O, N.CL100K, cdocs)
            assert np.array_equal(bits[:len(want_bits)], want_bits), (seq, int(np.flatnonzero(bits[:len(want_bits)] != want_bits)[0]))
            ids, ooff = enc.encode_batch(cdata, coffs)
            assert ooff.tolist() == ceoff and ids.tolist() == cexp, seq
        assert not np.array_equal(plain_bits, bits)
        for d in cdocs[:12]:                                 # (single strings: the batch path, the single-launch kernel stands aside in this mode)
            assert enc.encode_utf8(d) == oenc.encode_bytes(d)
        # ... and the oracle in this mode against `regex`, whose (?i) folds U+017F onto s as .NET >= 7 does (code units, .NET's \\s)
        for d in cdocs:
            u = RC.to_units(d.decode("utf-8"))
            want = RC.split_units_regex(2, u)
            got = O.split_utf16(N.CL100K, u)
            assert [tuple(x) for x in got] == want, d
        # the piece after 'ſ really is a piece of its own now
        assert [bytes(cdocs[1][a:a + l]) for a, l in O.split_utf8(N.CL100K, cdocs[1])] == [b"it", "'ſ".encode(), b"abc"]
        enc.set_option(N.OPT_CASE_EQUIVALENCE, 0)
        O.set_case_equivalence(False)
        assert np.array_equal(enc.pretokenize(cdata, coffs), plain_bits)
        assert [bytes(cdocs[1][a:a + l]) for a, l in O.split_utf8(N.CL100K, cdocs[1])] == [b"it", "'ſabc".encode()]
        # the option does nothing to the other patterns
        for pattern in (N.P1, N.O200K_DOTNET):
            e2 = N.Encoder(vocab, pattern)
            b0 = e2.pretokenize(cdata, coffs)
            e2.set_option(N.OPT_CASE_EQUIVALENCE, 1)
            assert np.array_equal(e2.pretokenize(cdata, coffs), b0)
    finally:
        O.set_unicode_classes(None)
        O.set_case_equivalence(False)


def check_document_marks(lib, O, vocab, ovocab):
    """k_docmark writes every word of the document-start bitmap once, by the first document that starts in it: runs of empty documents (stepped over by
    bisection), 64 one-byte documents in a word, empty documents at both ends, on the batch path and through the single launch."""
    oenc = O.Encoder(ovocab, O.CL100K)
    enc = N.Encoder(vocab, N.CL100K)
    filler = [b"the quick brown fox jumps over the lazy dog. " * 40] * 80            # (beyond the single launch's 128 KiB)
    for docs in ([b""] * 3000 + [b"a"] * 200 + [b""] * 70 + [b"hello world"] + [b""] * 5,
                 [b""] * 3000 + [b"a"] * 200 + [b""] * 70 + filler + [b""] * 1500 + [b"x", b"", b"yz", b""] * 40 + filler[:3] + [b""] * 5,
                 [b""] * 9, filler + [b""]):
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, docs)
        assert ids.tolist() == exp and ooff.tolist() == eoff, len(docs)


def check_sizing_attempt(lib, O, vocab, ovocab, capfd, pattern=N.CL100K, seed=43):
    """A fresh workspace's first large batch probes a sample of its sub-tiles first (encode_device: the sizing attempt) and every attempt after the first
    starts behind the pre-tokenizer, on the bitmaps the first one left.  Same ids as the oracle when the sample predicts the lists the batch needs (two
    attempts), when the crowded text lies behind the sample (three: sample, overflow, again), and on ordinary text (two); the attempts are read off the
    library's TKZ_LOG_SLOW_MS line.  The threshold is 64 MB of text: TKZ_SIZING_MIN_SUB brings it down to what a test can afford."""
    import re
    rng = random.Random(seed)
    cons = "bcdfghjklmnpqrstvwxz"
    words = "the of and to in is that for it with as was on be at by this had not are but from or have an they which one you were her all".split()

    def gib(n, lo, hi):
        out = []
        while sum(map(len, out)) < n:
            out.append(" " + "".join(rng.choice(cons) for _ in range(rng.randint(lo, hi))))
        return "".join(out)[:n]

    def plain(n):
        out = []
        while sum(map(len, out)) < n:
            out.append(" " + rng.choice(words))
        return "".join(out)[:n]
    oenc = O.Encoder(ovocab, pattern)
    crowded_all = [gib(4000, 2, 3).encode() for _ in range(45)]                                     # every sub-tile overflows a fresh list
    crowded_tail = [plain(4000).encode() for _ in range(40)] + [gib(4000, 2, 2).encode() for _ in range(5)]
    ordinary = [(plain(3000) + gib(300, 4, 9)).encode() for _ in range(50)]
    old = {k: os.environ.get(k) for k in ("TKZ_SIZING_MIN_SUB", "TKZ_LOG_SLOW_MS")}
    os.environ["TKZ_SIZING_MIN_SUB"] = "128"; os.environ["TKZ_LOG_SLOW_MS"] = "0"
    try:
        for docs, attempts in ((crowded_all, 2), (crowded_tail, 3), (ordinary, 2)):
            enc = N.Encoder(vocab, pattern)
            data, offs = pack(docs)
            assert len(data) > 140000
            exp, eoff = oracle_encode_docs(oenc, docs)
            for rep, want in ((0, attempts), (1, 1)):              # (the second call: a sized workspace, one attempt)
                capfd.readouterr()
                ids, ooff = enc.encode_batch(data, offs)
                err = capfd.readouterr().err
                assert ooff.tolist() == eoff and ids.tolist() == exp, (attempts, rep)
                got = [int(m) for m in re.findall(r"(\d+) attempt", err)]
                assert got and got[-1] == want, (attempts, rep, err)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_miss_lists(lib, O, vocab, ovocab, pattern=N.CL100K, seed=41):
    """The per-sub-tile miss lists (k_probe -> merge kernels -> k_place): sub-tiles with more misses than a list starts with (the batch
    is redone with longer lists), short and long misses sharing one list from both ends, a sub-tile of 1024 one-byte pieces, and calls
    after the lists have grown."""
    rng = random.Random(seed)
    enc = N.Encoder(vocab, pattern)
    oenc = O.Encoder(ovocab, pattern)
    cons = "bcdfghjklmnpqrstvwxz"

    def gib(n, lo, hi):
        out = []
        while sum(map(len, out)) < n:
            out.append(" " + "".join(rng.choice(cons) for _ in range(rng.randint(lo, hi))))
        return "".join(out)[:n]

    docs_small = [("the quick brown fox " * 40 + gib(200, 5, 9)).encode()]                          # a few misses: the lists as they start
    docs_big = [gib(5000, 2, 2).encode(),                                                             # ~340 three-byte misses per KiB
                "".join(gib(12, 3, 5) + gib(30, 17, 40) for _ in range(150)).encode(),               # short and long misses interleaved
                gib(3000, 17, 30).encode(),                                                           # long misses only
                ("a\nb\nc\nd\n" * 600).encode(),                                                      # 1024 pieces per sub-tile, all hits
                "".join(rng.choice(cons) + "\n" for _ in range(2000)).encode(),
                gib(700, 2, 3).encode() + ("x" * 1500).encode() + gib(700, 2, 16).encode(),           # a giant piece between crowded sub-tiles
                # (round 6) a group of 16 sub-tiles with ~2,000 short misses of EVERY length 1..16: k_merge_short keeps the pieces the memo does not answer in two
                # lists by length (<= 8 bytes | 9..16) from 320 misses a group on -- both fill, overflow mid-round and leave leftovers that share the last batch
                gib(20000, 1, 16).encode(), (gib(9000, 1, 8) + gib(9000, 9, 16)).encode()]
    # k_place's two paths: sub-tiles whose lists straddle what it keeps in LDS (32 of each kind), whose token runs straddle kPlaceBig (8),
    # with more than 256 pieces, and batches of tiny documents (up to four document starts in the four records a lane holds)
    words = "the of and to in is that for it with as was on be at by this had not are but from or have an they which one you were her all".split()

    def mixed(n, every, lo, hi):
        out = []
        while sum(map(len, out)) < n:
            out.append(" " + rng.choice(words) if rng.randrange(every) else " " + "".join(rng.choice(cons) for _ in range(rng.randint(lo, hi))))
        return "".join(out)[:n]

    docs_place = [mixed(6000, e, lo, hi).encode() for (e, lo, hi) in ((8, 3, 9), (6, 4, 16), (7, 17, 30), (5, 2, 40), (9, 9, 12), (4, 17, 24))]
    # ... sub-tiles with 65 .. 128 misses (k_place keeps 128 list entries in LDS, two per lane, the long list from the top slot down: the
    # second half is only loaded when more than 64 are kept) and a little beyond 128 (the general path): one piece in three / in two misses
    docs_place += [mixed(9000, e, lo, hi).encode() for (e, lo, hi) in ((3, 3, 9), (3, 3, 30), (2, 4, 24), (2, 2, 40), (3, 17, 22), (2, 2, 4))]

    def crowded(n, hits):        # per `hits` common words: four short misses and one long one -- ~100 .. 135 list entries per sub-tile, a fifth of them long
        out = []
        while sum(map(len, out)) < n:
            out += [" " + rng.choice(words) for _ in range(hits)]
            out += [" " + rng.choice(cons) + rng.choice(cons) for _ in range(4)]
            out.append(" " + "".join(rng.choice(cons) for _ in range(17)))
        return "".join(out)[:n]
    docs_place += [crowded(7000, h).encode() for h in (5, 3, 2)]
    tiny = [rng.choice([b"a", b" b", b"\n", b"c d", b"qz", b" the", b"x\n\n", b" zqxj"]) for _ in range(5000)]
    for docs in (docs_small, docs_big, docs_small, docs_big[::-1], docs_place, tiny, docs_place + tiny[:700] + docs_big[:2]):
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, docs)
        assert ooff.tolist() == eoff and ids.tolist() == exp
    # the 128-slot form of k_place (launched when more than a fifth of the sub-tiles of the workspace's previous batch held more than 64 list
    # entries): a batch too large for the single launch,
    # most of its sub-tiles with 65 .. 128 list entries of both kinds, one document with far more; twice (the second call starts on the grown lists)
    enc3 = N.Encoder(vocab, pattern)
    docs128 = [crowded(40000, h).encode() for h in (5, 3, 4, 3)] + [gib(5000, 2, 2).encode(), mixed(9000, 3, 3, 9).encode()]
    data, offs = pack(docs128)
    exp, eoff = oracle_encode_docs(oenc, docs128)
    for rep in range(3):           # (the form is chosen from the batch before: the first call runs k_place<64>, the later ones k_place<128>)
        ids, ooff = enc3.encode_batch(data, offs)
        assert ooff.tolist() == eoff and ids.tolist() == exp, rep
    # piece granularity on the crowded text (every record marked)
    data, offs = pack(docs_big[:3])
    ids, dpo, pbo, pto = enc.encode_batch_pieces(data, offs)
    exp, _ = oracle_encode_docs(oenc, docs_big[:3])
    assert ids.tolist() == exp
    # lists that grew give their memory back -- with hysteresis (three consecutive batches that would have done with shorter lists, one halving
    # a time): a crowded batch on the batch path (more than the single launch takes), then plain text of the same size until the lists are back
    # at their smallest, then the crowded one again; and a workspace that ALTERNATES crowded and plain batches keeps its lists (no re-run, no
    # free / allocate on every other batch)
    enc2 = N.Encoder(vocab, pattern)
    crowded = [gib(50000, 2, 2).encode() for _ in range(4)]
    plain = [("the quick brown fox jumps over the lazy dog, it's 12345 o'clock\n" * 800).encode() for _ in range(4)]
    sizes = []
    exp_of = {id(d): oracle_encode_docs(oenc, d) for d in (crowded, plain)}
    seq = [crowded] + [plain] * 14 + [crowded] + [plain, crowded] * 3
    for docs in seq:
        data, offs = pack(docs)
        ids, ooff = enc2.encode_batch(data, offs)
        exp, eoff = exp_of[id(docs)]
        assert ooff.tolist() == eoff and ids.tolist() == exp
        sizes.append(enc2.workspace_bytes)
    # (sizes[1]: both kinds of batch have been through the workspace once -- every other buffer has its final size)
    assert sizes[2] == sizes[1], sizes                                    # two low batches: nothing moves yet
    assert min(sizes[1:15]) < sizes[1] and sizes[14] <= min(sizes[1:15]) + (1 << 20), sizes      # ... then the lists come down, step by step
    assert sizes[15] > sizes[14], sizes                                   # the crowded batch grows the lists again
    assert len(set(sizes[16:])) == 1, sizes                               # alternating batches: the lists stay as they are


def check_place_paths(lib, O):
    """k_place's fast path hands a chunk of 256 records back to the general path when its tokens would overflow the LDS stage: under a
    table of nothing but the 256 single bytes a 30-letter word is 30 tokens, so 24 of them among single-byte pieces (which ARE keys) make a
    chunk of 952 tokens -- with at most 32 long misses in the sub-tile, which keeps it on the fast path until then; the chunk behind it
    (hits only) is placed by the fast path again.  Also token runs of exactly 32 and 33 (the longest the fast path takes, the shortest it
    refuses)."""
    raw = random_vocab_bytes(random.Random(1), n_keys=0)
    vocab, ovocab = N.Vocab(raw, lib), O.Vocab(raw)
    enc = N.Encoder(vocab, N.CL100K)
    oenc = O.Encoder(ovocab, N.CL100K)
    docs = [(("x" * 30 + "1") * 24 + "a1" * 140).encode() * 3,
            (("y" * 32 + "2") * 10 + "b2" * 300).encode(), (("y" * 33 + "2") * 10 + "b2" * 300).encode(),
            ("q7" * 500 + ("z" * 17 + "3") * 30 + "c3" * 100).encode()]
    for dd in (docs, docs[::-1], [b"".join(docs)]):
        data, offs = pack(dd)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, dd)
        assert ooff.tolist() == eoff and ids.tolist() == exp


def check_memo_zero_bytes(lib, O, vocab, ovocab, seed=43):
    """Pieces that hold zero bytes never use the piece memo (that rule is what makes a hit exact whatever mixture of old and new slot
    words a reader is handed): runs of NUL of every length beside letter pieces of the same lengths, memo off / empty / filled."""
    rng = random.Random(seed)
    enc = N.Encoder(vocab, N.CL100K)
    pcs = []
    for n in range(1, 17):
        pcs.append(b"\0" * n)
        pcs.append(bytes(rng.choice(b"bcdfghjklmnpqrstvwxz") for _ in range(n)))
        pcs.append(bytes(rng.choice(b"qzx") for _ in range(n - 1)) + b"\0")
        pcs.append(b"\0" + bytes(rng.choice(b"qzx") for _ in range(n - 1)))
        assert enc.memo_bucket(pcs[-1]) == -1 and enc.memo_bucket(pcs[-3]) >= 0
    pcs = pcs * 20
    rng.shuffle(pcs)
    exp, eoff = [], [0]
    for p in pcs:
        r = ovocab.rank(p)
        exp += [r] if r >= 0 else ovocab.bpe(p)
        eoff.append(len(exp))
    data, offs = pack(pcs)
    for mode in (0, 2, 1, 1):
        enc.set_option(N.OPT_PIECE_MEMO, mode)
        ids, ooff = enc.encode_pieces(data, offs)
        assert ids.tolist() == exp and ooff.tolist() == eoff, mode


def check_memo_contention(lib, O, vocab, ovocab, candidates=400_000, threads=2, rounds=3, seed=47):
    """Pieces that all map to ONE bucket of the piece memo (found with tkz_encoder_memo_bucket), interleaved with NUL runs of the same
    lengths and with pieces of other buckets, encoded by several host threads at once on one encoder (each call on its own workspace
    and stream, all of them reading and claiming the same slots): every call must return the oracle's ids."""
    import threading
    rng = random.Random(seed)
    enc = N.Encoder(vocab, N.CL100K)
    enc.set_option(N.OPT_PIECE_MEMO, 2)
    cons, vow = "bcdfghjklmnpqrstvwxz", "aeiou"
    by_bucket = {}
    seen = set()
    for _ in range(candidates):
        w = " " + "".join(rng.choice(cons) + rng.choice(vow) for _ in range(rng.randint(2, 5))) + rng.choice(["", "x", "q"])
        b = w.encode()
        if b in seen or ovocab.rank(b) >= 0:
            continue
        seen.add(b)
        by_bucket.setdefault(enc.memo_bucket(b), []).append(b)
    crowd = max(by_bucket.values(), key=len)
    assert len(crowd) >= 2 * enc.memo_ways, len(crowd)       # more contenders than the bucket has slots
    others = [v[0] for k, v in list(by_bucket.items())[:500]]
    nul = [b"\0" * len(p) for p in crowd]
    oenc = O.Encoder(ovocab, N.CL100K)

    def make(r):
        rr = random.Random(seed + r)
        words = []
        for _ in range(6000):
            k = rr.random()
            words.append(rr.choice(crowd) if k < 0.6 else rr.choice(nul) if k < 0.75 else rr.choice(others))
        docs = [b"".join(words[i:i + 40]) for i in range(0, len(words), 40)]
        return docs, oracle_encode_docs(oenc, docs)
    jobs = [make(r) for r in range(threads)]
    errors = []

    def work(i):
        docs, (exp, eoff) = jobs[i]
        data, offs = pack(docs)
        for _ in range(rounds):
            ids, ooff = enc.encode_batch(data, offs)
            if ids.tolist() != exp or ooff.tolist() != eoff:
                errors.append(i)
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def check_host_alloc(lib, O, vocab, ovocab):
    """tkz_host_alloc / tkz_host_free: page-locked buffers for the host-buffer entry points, used as input and output of a batch call."""
    hp = [C.c_void_p() for _ in range(4)]
    docs = [("doc %d: the quick brown fox, it's 2024!\n" % i).encode() * (1 + i % 7) for i in range(2000)]
    data, offs = pack(docs)
    sizes = [len(data) + 64, 8 * (len(docs) + 1), 4 * len(data), 8 * (len(docs) + 1)]
    for h, n in zip(hp, sizes):
        lib.check(lib.L.tkz_host_alloc(n, C.byref(h)))
        assert h.value and h.value % 64 == 0
    offs64 = np.ascontiguousarray(offs, np.int64)
    C.memmove(hp[0], data.ctypes.data, len(data)); C.memmove(hp[1], offs64.ctypes.data, 8 * (len(docs) + 1))
    enc = N.Encoder(vocab, N.CL100K)
    needed = C.c_int64(0)
    lib.check(lib.L.tkz_encode_batch_utf8(enc._h, hp[0], hp[1], len(docs), hp[2], len(data), hp[3], C.byref(needed)))
    ids = np.ctypeslib.as_array(C.cast(hp[2], C.POINTER(C.c_int32)), (needed.value,)).copy()
    ooff = np.ctypeslib.as_array(C.cast(hp[3], C.POINTER(C.c_int64)), (len(docs) + 1,)).copy()
    exp, eoff = oracle_encode_docs(O.Encoder(ovocab, O.CL100K), docs)
    assert ids.tolist() == exp and ooff.tolist() == eoff
    # the text at a 16-byte aligned place of the block is fetched by the launch sequence's own first kernel (k_ingest), elsewhere by a copy command; a
    # batch that overflows a fresh workspace's miss lists is run again on the text that kernel left in the staging buffer
    rng = random.Random(5)
    crowded = [("".join(" " + rng.choice("bcdfghjklmnpqrstvwxz") + rng.choice("bcdfghjklmnpqrstvwxz") for _ in range(1200))).encode() for _ in range(60)]
    for shift, dd in ((16, docs), (3, docs), (0, crowded), (48, crowded)):
        data2, offs2 = pack(dd)
        assert len(data2) + 64 <= sizes[0] and len(dd) <= len(docs)
        offs64 = np.ascontiguousarray(offs2, np.int64)
        C.memmove(hp[0].value + shift, data2.ctypes.data, len(data2)); C.memmove(hp[1], offs64.ctypes.data, 8 * (len(dd) + 1))
        enc2 = N.Encoder(vocab, N.CL100K)
        lib.check(lib.L.tkz_encode_batch_utf8(enc2._h, C.c_void_p(hp[0].value + shift), hp[1], len(dd), hp[2], len(data2), hp[3], C.byref(needed)))
        ids = np.ctypeslib.as_array(C.cast(hp[2], C.POINTER(C.c_int32)), (needed.value,)).copy()
        ooff = np.ctypeslib.as_array(C.cast(hp[3], C.POINTER(C.c_int64)), (len(dd) + 1,)).copy()
        exp, eoff = oracle_encode_docs(O.Encoder(ovocab, O.CL100K), dd)
        assert ids.tolist() == exp and ooff.tolist() == eoff, shift
    for h in hp:
        lib.L.tkz_host_free(h)
    lib.L.tkz_host_free(None)
    z = C.c_void_p()
    lib.check(lib.L.tkz_host_alloc(0, C.byref(z)))          # (a zero-byte request still yields a pointer that can be freed)
    lib.L.tkz_host_free(z)


def check_device_unicode_table(lib, vocab):
    """The Unicode class table as the DEVICE holds it (downloaded through tkz_encoder_unicode_classes) against `unicodedata` 13.0, all
    1,114,112 code points: the oracle and the product share one generated table, so this -- not a comparison of the two -- is what would
    catch a wrong class (tests/test_unicode_tables.py does the same for the host copies)."""
    from test_unicode_tables import expected_class
    enc = N.Encoder(vocab, N.O200K)
    got = np.zeros(0x110000, np.uint8)
    lib.check(lib.L.tkz_encoder_unicode_classes(enc._h, 0, 0x110000, got.ctypes.data))
    exp = np.fromiter((expected_class(u) for u in range(0x110000)), np.uint8, 0x110000)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, [(hex(int(u)), int(got[u]), int(exp[u])) for u in bad[:10]]


def read_device_i64(lib, ptr, n):
    """n int64 at a device pointer of the library's: host memory on the CPU-emulated build, hipMemcpy (which also waits for the null
    stream's earlier work) on the GPU."""
    out = np.zeros(n, np.int64)
    if "hostemu" in lib.path:
        C.memmove(out.ctypes.data, ptr, 8 * n)
    else:
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemcpy(out.ctypes.data, ptr, 8 * n, 2) == 0
    return out


def check_small_path(lib, O, vocab, ovocab, seed=53, rounds=40):
    """The single-launch path for small batches (k_small: at most 64 KiB in at most 8192 documents; o200k: of at most 1 KiB each): same ids as the
    oracle on single prompts and small batches of all three patterns; the calls that must be handed back to the batch path (a giant
    piece, malformed text, more misses than the lists hold) still give the batch path's answer; larger batches do not take it."""
    rng = random.Random(seed)
    alpha = RC.alphabet()
    for pattern in (N.P1, N.CL100K, N.O200K, N.O200K_DOTNET):
        enc = N.Encoder(vocab, pattern)
        oenc = O.Encoder(ovocab, pattern)
        for it in range(rounds):
            kind = rng.choice(["mix", "ws", "oth", "dig", "apo", "case", "a_mix", "a_brk", "a_ws", "a_dig"])
            nd = rng.choice([1, 1, 1, 2, 7, 60, 400])
            docs = [gen_text(rng, kind, rng.choice([0, 1, 5, 20, 64, 100, 300, 1000]) if nd < 100 else rng.randint(0, 18), alpha).encode("utf-8")[:1024] for _ in range(nd)]
            docs = [d.decode("utf-8", "ignore").encode("utf-8") for d in docs]           # (the cut may have split a char)
            while sum(map(len, docs)) > (65536 if pattern in (N.O200K, N.O200K_DOTNET) else 131072):
                docs.pop()
            if not sum(map(len, docs)):
                continue
            before = enc.small_path_calls()
            data, offs = pack(docs)
            ids, ooff = enc.encode_batch(data, offs)
            exp, eoff = oracle_encode_docs(oenc, docs)
            assert ids.tolist() == exp and ooff.tolist() == eoff, (pattern, it, kind, nd)
            after = enc.small_path_calls()
            assert after[0] == before[0] + 1, "a batch of %d bytes in %d documents must take the single-launch path" % (len(data), len(docs))
            # the device-resident counts (what the count all-gather of a sharded job sends) are this batch's, whichever path it took
            assert read_device_i64(lib, enc.counts_device, 3).tolist() == [len(docs), len(data), len(exp)], (pattern, it)
        # single strings through both single-string entries
        for text in ("Hello World", "Hello World, this is a short prompt of sixty-four bytes, more or", "⭐ naïve café 漢字かな 😀👍🏽 it's 12345\n\n  x", "a", " ", "\n"):
            b = text.encode("utf-8")
            assert enc.encode_utf8(b) == oenc.encode_bytes(b)
            assert enc.encode_utf16(np.frombuffer(text.encode("utf-16-le"), np.uint16).tolist()) == oenc.encode_bytes(b)
        calls, back = enc.small_path_calls()
        # (a document full of missed pieces outgrows a fresh encoder's lists once; the generator's 300- and 1000-char documents of ONE class hold missed
        #  pieces of more than 256 bytes, which the batch path merges a wavefront a piece -- k_merge_coop --: handed back)
        assert calls > rounds // 2 and back <= calls // 3, (calls, back)
        # handed back: a piece of more than 1024 bytes inside a small batch, a missed piece of more than 256 ...
        for docs in ([b"q" * 1024, b"hello world"],           # (1024 letters: one piece as long as a document of this path can be)
                     [b"hello " + b"q" * 300 + b" world", b"x"]):
            data, offs = pack(docs)
            c0 = enc.small_path_calls()
            ids, ooff = enc.encode_batch(data, offs)
            exp, eoff = oracle_encode_docs(oenc, docs)
            assert ids.tolist() == exp and ooff.tolist() == eoff
            c1 = enc.small_path_calls()
            assert (c1[0] - c0[0], c1[1] - c0[1]) == (1, 1), (c0, c1)
        # ... more misses than a fresh encoder's lists hold (the batch path grows them; later small calls fit)
        cons = "bcdfghjklmnpqrstvwxz"
        gdocs = ["".join(" " + rng.choice(cons) + rng.choice(cons) for _ in range(300)).encode() for _ in range(3)]
        gdata, goffs = pack(gdocs)
        gexp, geoff = oracle_encode_docs(oenc, gdocs)
        fresh = N.Encoder(vocab, pattern)
        for rep in range(2):
            ids, ooff = fresh.encode_batch(gdata, goffs)
            assert ids.tolist() == gexp and ooff.tolist() == geoff
        assert fresh.small_path_calls() == (2, 1), fresh.small_path_calls()        # handed back once; the lists have grown, the second call fits
        # malformed text and a document boundary inside a char: the same errors as the batch path
        import pytest
        with pytest.raises(N.TkzError) as ei:
            enc.encode_utf8(b"abc\xff")
        assert ei.value.code == N.E_INVALID_UTF8
        # a document of several KiB: one launch for pattern 1 / cl100k (row and block evaluators), the batch path for o200k (its
        # single-launch form splits with the sequential matcher, one lane per document: documents of up to 1 KiB only)
        mid = ("lorem ipsum dolor sit amet, consectetur 12345 adipiscing elit; " * 300).encode()[:rng.choice([5000, 17000, 40000])]
        c0 = enc.small_path_calls()
        assert enc.encode_utf8(mid) == oenc.encode_bytes(mid)
        assert enc.small_path_calls()[0] == c0[0] + (0 if pattern in (N.O200K, N.O200K_DOTNET) else 1)
        mdocs = [("doc %d: the quick brown fox, it's 2024!\n" % i).encode() * rng.randint(1, 5) for i in range(300 if pattern in (N.O200K, N.O200K_DOTNET) else 800)]
        mdata, moffs = pack(mdocs)
        ids, ooff = enc.encode_batch(mdata, moffs)
        mexp, meoff = oracle_encode_docs(oenc, mdocs)
        assert ids.tolist() == mexp and ooff.tolist() == meoff and len(mdata) <= (65536 if pattern in (N.O200K, N.O200K_DOTNET) else 131072)
        # too large for the single launch: the batch path, untouched counters
        big = ("lorem ipsum dolor sit amet " * 6000).encode()
        c0 = enc.small_path_calls()
        assert enc.encode_utf8(big) == oenc.encode_bytes(big)
        assert enc.small_path_calls() == c0


def check_long_diverse_pieces(lib, O, vocab, ovocab, lens=(3000, 9000, 20000), seed=5):
    """Long pieces that merge a few pairs per rank (chains of capitalised words under cl100k: one `\\p{L}+` piece), alone and followed by a run of one letter:
    tkz_bpe_long_tail from its entry points -- k_merge_coop (257..1024 bytes), the giant pieces' workgroup with the state in LDS (<= 16384 parts) and with
    the ids in the pool (<= 32768 parts), after rounds in global memory beyond."""
    rng = random.Random(seed)
    enc = N.Encoder(vocab, N.CL100K)
    oenc = O.Encoder(ovocab, N.CL100K)
    words = [w.decode().strip() for w, _ in ovocab.entries() if w.strip().isalpha() and len(w.strip()) > 3][:3000]
    for n in lens:
        for text in ("".join(rng.choice(words).capitalize() for _ in range(n // 5))[:n].encode(),
                     ("".join(rng.choice(words).capitalize() for _ in range(n // 10))[:n // 2] + "q" * (n // 2)).encode()):
            assert enc.encode_utf8(text) == oenc.encode_bytes(text), n


def check_runs_with_words(lib, O, vocab, ovocab, seed=9):
    """Runs of one byte (letters whose doubles are keys and letters whose doubles are not, '=', spaces) with a word before, between and behind them: what the
    proposals of the tail serialise on and its rounds for chains of equal pairs are for; every length class of the three mergers, both o200k readings too."""
    rng = random.Random(seed)
    for pattern in (N.CL100K, N.O200K_DOTNET):
        enc, oenc = N.Encoder(vocab, pattern), O.Encoder(ovocab, pattern)
        docs = []
        for n in (260, 700, 1024, 1025, 5000, 17000, 31000):
            for ch in "neq= ":
                docs.append((" they" + ch * n).encode())
                docs.append((" delogicality" + ch * (n // 2) + "Word" + rng.choice("wst") * (n // 2) + "stop").encode())
            docs.append(("ab" * (n // 2)).encode()); docs.append(("the" * (n // 3) + "quick" * 40).encode())
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, docs)
        assert ids.tolist() == exp and ooff.tolist() == eoff, pattern


def check_long_pieces_entry_points(lib, O, vocab, ovocab, seed=13):
    """Missed pieces of 257..1024 bytes (k_merge_coop) and giant ones (k_giant_merge) through the OTHER entry points: the UTF-16 batch (transcoded on the
    device), the piece-granular batch (token ranges of every piece), single strings, with special tokens between them (the C# segmentation), and many of
    them in one chunk of sub-tiles beside short and 17..256-byte misses -- every one against the oracle."""
    rng = random.Random(seed)
    cons = "bcdfghjklmnpqrstvwxz"
    def run(n):
        k = rng.randrange(4)
        if k == 0: return rng.choice("nqe= ") * n
        if k == 1: return "".join(rng.choice(cons) for _ in range(n))
        if k == 2: return ("the" * n)[:n]
        return "".join(rng.choice(["Word", "Stop", "Quick", "Zq", "X"]) for _ in range(n))[:n]
    for pattern in (N.CL100K, N.O200K_DOTNET):
        enc, oenc = N.Encoder(vocab, pattern), O.Encoder(ovocab, pattern)
        docs = []
        for _ in range(24):
            parts = []
            for _ in range(rng.randrange(1, 6)):
                parts.append(run(rng.choice([20, 100, 257, 300, 600, 1024, 1025, 2000, 6000])))
                parts.append(rng.choice([" ", "\n", " hello world, it's 12345 ", "", " \u6f22\u5b57 "]))
            docs.append("".join(parts))
        b = [d.encode("utf-8") for d in docs]
        data, offs = pack(b)
        exp, eoff = oracle_encode_docs(oenc, b)
        ids, ooff = enc.encode_batch(data, offs)
        assert ids.tolist() == exp and ooff.tolist() == eoff, pattern
        # UTF-16 code units, transcoded on the device
        units = [np.frombuffer(d.encode("utf-16-le"), np.uint16) for d in docs]
        uoffs = np.cumsum([0] + [len(u) for u in units]).astype(np.int64)
        ids16, ooff16 = enc.encode_batch_utf16(np.concatenate(units) if units else np.zeros(0, np.uint16), uoffs)
        assert ids16.tolist() == exp and ooff16.tolist() == eoff, pattern
        # piece granularity: the token range of every piece
        pids, dpo, pbo, pto = enc.encode_batch_pieces(data, offs)
        assert pids.tolist() == exp, pattern
        e_pto, pos, k = [0], 0, 0
        for d in b:
            for (a, n) in O.split_utf8(pattern, d):
                p = d[a:a + n]
                r = ovocab.rank(p)
                k += 1 if r >= 0 else len(ovocab.bpe(p))
                e_pto.append(k)
        assert pto.tolist() == e_pto and dpo[-1] == len(e_pto) - 1, pattern
        # single strings (a batch of one: the single-launch path hands these back)
        for d in b[:4]:
            assert enc.encode_utf8(d) == oenc.encode_bytes(d)


def check_errors(lib, O, vocab):
    enc = N.Encoder(vocab, N.CL100K)
    import pytest
    # malformed UTF-8 (the reference never sees it: a C# string always converts to well-formed UTF-8)
    for bad in (b"abc\xff", b"\xe4\xb8", b"\x80abc", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80"):
        with pytest.raises(N.TkzError) as ei:
            enc.encode_utf8(bad)
        assert ei.value.code == N.E_INVALID_UTF8, bad
    # a document boundary inside a character
    data = np.frombuffer("a中b".encode("utf-8"), np.uint8)
    with pytest.raises(N.TkzError) as ei:
        enc.encode_batch(data, np.array([0, 2, len(data)]))
    assert ei.value.code == N.E_INVALID_UTF8
    # offsets
    data = np.frombuffer(b"hello world", np.uint8)
    for offs in ([0, 7, 5, 11], [0, 12, 3, 11]):
        with pytest.raises(N.TkzError) as ei:
            enc.encode_batch(data, np.array(offs))
        assert ei.value.code == N.E_ARG, offs
    # capacity: the needed count comes back
    ids = np.empty(2, np.int32)
    ooff = np.empty(2, np.int64)
    import ctypes as C
    needed = C.c_int64(0)
    st = lib.L.tkz_encode_batch_utf8(enc._h, data.ctypes.data, np.array([0, 11], np.int64).ctypes.data, 1, ids.ctypes.data, 1, ooff.ctypes.data, C.byref(needed))
    assert st == N.E_CAPACITY and needed.value == 2
    # piece-granular form with piece arrays that are too small: BOTH required sizes come back from the one call (tkz.h)
    text = np.frombuffer(b"hello world, hello there", np.uint8)
    tiny_cap = 2
    pids = np.empty(64, np.int32)
    dpo = np.empty(2, np.int64); pbo = np.empty(tiny_cap + 1, np.int64); pto = np.empty(tiny_cap + 1, np.int64)
    npieces, need_ids = C.c_int64(0), C.c_int64(0)
    st = lib.L.tkz_encode_batch_pieces_utf8(enc._h, text.ctypes.data, np.array([0, len(text)], np.int64).ctypes.data, 1, pids.ctypes.data, 64,
                                            dpo.ctypes.data, pbo.ctypes.data, pto.ctypes.data, tiny_cap, C.byref(npieces), C.byref(need_ids))
    full = enc.encode_batch_pieces(text, np.array([0, len(text)]))
    assert st == N.E_CAPACITY and npieces.value == len(full[2]) - 1 and need_ids.value == len(full[0]) > 0
    # empty inputs (TikTokenizerUnitTest.cs:103-109)
    assert enc.encode_utf8(b"") == []
    ids, ooff = enc.encode_batch(np.zeros(0, np.uint8), np.array([0, 0, 0]))
    assert len(ids) == 0 and ooff.tolist() == [0, 0, 0]
    # a byte that is not in the vocabulary: KeyNotFoundException in the reference (BytePairEncoder.cs:17,73)
    tiny = N.Vocab(b"YQ== 0\nYWI= 1\n", lib)          # 'a', 'ab'
    te = N.Encoder(tiny, N.P1)
    assert te.encode_utf8(b"ab") == [1]
    assert te.encode_utf8(b"aab") == [0, 1]
    with pytest.raises(N.KeyNotFoundError):
        te.encode_utf8(b"b")
    with pytest.raises(N.KeyNotFoundError):
        te.encode_utf8(b"abb")


def check_utf16(lib, O, vocab, ovocab):
    """tkz_encode_utf16 vs the oracle's UTF-16 entry (lone surrogates -> U+FFFD per piece, TikTokenizer.cs:261)."""
    rng = random.Random(77)
    alpha = RC.alphabet()
    for pattern in (N.P1, N.CL100K, N.O200K, N.O200K_DOTNET):
        enc = N.Encoder(vocab, pattern)
        oenc = O.Encoder(ovocab, pattern)
        for it in range(12):
            units = RC.to_units(RC.random_text(rng, alpha, rng.randint(0, 80)))
            for _ in range(rng.choice([0, 0, 1, 3])):   # sprinkle lone surrogates
                units.insert(rng.randint(0, len(units)), rng.choice([0xD800, 0xDBFF, 0xDC00, 0xDFFF]))
            assert enc.encode_utf16(units) == oenc.encode_utf16(units), (pattern, units)


def check_utf16_batch(lib, O, vocab, ovocab, seed=91, rounds=10, doc_counts=(1, 7, 60), max_units=400):
    """tkz_encode_batch_utf16 (code units transcoded ON THE DEVICE) vs the oracle's UTF-16 entry per document, and vs the UTF-8
    batch entry fed with the bytes Encoding.UTF8.GetBytes would produce.  Surrogate pairs, lone halves, halves that face each
    other across a document boundary, pairs straddling the 16-unit lane groups and the 1024-unit tiles, empty documents."""
    rng = random.Random(seed)
    alpha = RC.alphabet()
    for pattern in (N.P1, N.CL100K, N.O200K, N.O200K_DOTNET):
        enc = N.Encoder(vocab, pattern)
        oenc = O.Encoder(ovocab, pattern)
        for it in range(rounds):
            docs = []
            for _ in range(rng.choice(doc_counts)):
                k = rng.random()
                if k < 0.1:
                    units = []
                elif k < 0.3:                                   # dense surrogate soup
                    units = [rng.choice([0xD800, 0xDBFF, 0xDC00, 0xDFFF, 0xD83D, 0xDE00, 0x41, 0x4E2D, 0xE9]) for _ in range(rng.randint(1, max_units))]
                else:
                    units = RC.to_units(RC.random_text(rng, alpha, rng.randint(0, max_units)))
                    for _ in range(rng.choice([0, 0, 1, 3])):
                        units.insert(rng.randint(0, len(units)), rng.choice([0xD800, 0xDBFF, 0xDC00, 0xDFFF]))
                docs.append(units)
            if len(docs) >= 2 and rng.random() < 0.5:           # a high half ending one document, a low half starting the next
                i = rng.randrange(len(docs) - 1)
                docs[i] = docs[i] + [0xD83D]
                docs[i + 1] = [0xDE00] + docs[i + 1]
            flat = np.asarray([u for d in docs for u in d], dtype=np.uint16)
            offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
            ids, ooff = enc.encode_batch_utf16(flat, offs)
            for d, units in enumerate(docs):
                assert ids[ooff[d]:ooff[d + 1]].tolist() == oenc.encode_utf16(units), (pattern, it, d, units[:40])
            u8 = [np.asarray(d, dtype=np.uint16).tobytes().decode("utf-16-le", "replace").encode("utf-8") for d in docs]
            data, boffs = pack(u8)
            ids8, ooff8 = enc.encode_batch(data, boffs)
            assert ids.tolist() == ids8.tolist() and ooff.tolist() == ooff8.tolist()
    enc = N.Encoder(vocab, N.CL100K)
    ids, ooff = enc.encode_batch_utf16(np.zeros(0, np.uint16), np.zeros(4, np.int64))       # three empty documents
    assert len(ids) == 0 and ooff.tolist() == [0, 0, 0, 0]
    with pytest.raises(N.TkzError):
        enc.encode_batch_utf16(np.zeros(8, np.uint16) + 65, np.array([0, 5, 3, 8], dtype=np.int64))


def random_vocab_bytes(rng, alphabet=b"abc", n_keys=300, max_len=6, rank_step=1, rank_base=0):
    """A rank table that is NOT a trained BPE vocabulary: all 256 single bytes plus random strings over a tiny
    alphabet, ranks in random order.  The reference's merge loop is still well defined on it, and new pairs
    routinely rank BELOW the pair just merged -- the case the round-based merger has to cut its rounds for."""
    import base64
    keys = {bytes([b]) for b in range(256)}
    n_keys = min(n_keys, sum(len(alphabet) ** k for k in range(2, max_len + 1)) // 2)
    while len(keys) < 256 + n_keys:
        keys.add(bytes(rng.choice(alphabet) for _ in range(rng.randint(2, max_len))))
    keys = list(keys)
    rng.shuffle(keys)
    # (sparse / large ranks are legal in a .tiktoken file; ranks >= 2^22 switch the lane mergers to their unpacked form)
    return b"".join(base64.b64encode(k) + b" " + str(rank_base + rank_step * i).encode() + b"\n" for i, k in enumerate(keys))


def check_random_vocab(lib, O, seed, n_vocabs, lens, n_pieces, max_len=6):
    """Every merge path (lean lane, arena lane packed and unpacked, whole-wave rounds in the pool) on adversarial rank tables."""
    rng = random.Random(seed)
    for vi in range(n_vocabs):
        big = vi % 3 == 2                                        # every third vocabulary: sparse ranks up to ~2^26
        raw = random_vocab_bytes(rng, alphabet=rng.choice([b"ab", b"abc", b"abcd"]), n_keys=rng.choice([20, 100, 400]), max_len=max_len,
                                 rank_step=97_003 if big else 1, rank_base=4_200_000 if big else 0)
        vocab, ovocab = N.Vocab(raw, lib), O.Vocab(raw)
        enc = N.Encoder(vocab, N.CL100K)
        pcs = [bytes(rng.choice(b"abcd"[:rng.randint(1, 4)]) for _ in range(rng.choice(lens))) for _ in range(n_pieces)]
        data, offs = pack(pcs)
        ids, ooff = enc.encode_pieces(data, offs)
        for i, p in enumerate(pcs):
            r = ovocab.rank(p)
            x = [r] if r >= 0 else ovocab.bpe(p)
            g = ids[ooff[i]:ooff[i + 1]].tolist()
            assert g == x, "vocab %d piece %d (len %d) %r: got %r expected %r" % (vi, i, len(p), p[:60], g[:16], x[:16])


def check_decode(lib, O, vocab, ovocab, raw_vocab=None, seed=5, rounds=6):
    """Batch Decode on the device (TikTokenizer.cs:586-604) vs the vocabulary's own keys: decode(encode(x)) == x for every
    document (the round trip every reference test asserts, TikTokenizerUnitTest.cs:47-48), unknown ids dropped, special
    tokens decoded from their literals, capacity and offset errors."""
    alpha = RC.alphabet()
    rng = random.Random(seed)
    enc = N.Encoder(vocab, N.CL100K)
    ents = ovocab.entries()
    key_of = {r: k for k, r in ents}
    max_id = max(key_of)
    for it in range(rounds):
        docs = [gen_text(rng, rng.choice(["mix", "a_mix", "oth", "ws"]), rng.choice([0, 1, 9, 200, 3000]), alpha).encode("utf-8") for _ in range(rng.choice([1, 5, 60]))]
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        out, boffs = enc.decode_batch(ids, ooff)
        assert out.tobytes() == data.tobytes() and boffs.tolist() == offs.tolist(), "round %d" % it
    # ids in no table contribute nothing; specials decode to their literal; a vocabulary id is never shadowed by a special
    known = sorted(key_of)
    a, b, c = known[5], known[0], known[17]
    enc.set_special_tokens({"<|endoftext|>": max_id + 1, "<|x|>": max_id + 20, "shadow": a})
    seq = [a, max_id + 1, max_id + 7, -3, b, max_id + 20, 2**31 - 1, c]
    exp = key_of[a] + b"<|endoftext|>" + key_of[b] + b"<|x|>" + key_of[c]
    out, boffs = enc.decode_batch(np.asarray(seq, np.int32), np.array([0, 3, 3, len(seq)]))
    assert out.tobytes() == exp and boffs.tolist() == [0, len(key_of[a]) + 13, len(key_of[a]) + 13, len(exp)]
    out, boffs = enc.decode_batch(np.zeros(0, np.int32), np.array([0, 0, 0]))
    assert len(out) == 0 and boffs.tolist() == [0, 0, 0]
    # a long tile: 1024 ids of the longest key exceed the LDS stage of k_dec_write (direct copies)
    longest = max(ents, key=lambda e: len(e[0]))
    many = np.full(2500, longest[1], np.int32)
    out, boffs = enc.decode_batch(many, np.array([0, 1000, 2500]))
    assert out.tobytes() == longest[0] * 2500 and boffs.tolist() == [0, 1000 * len(longest[0]), 2500 * len(longest[0])]
    with pytest.raises(N.TkzError) as ei:
        enc.decode_batch(np.asarray([1, 2, 3], np.int32), np.array([0, 2, 1, 3]))
    assert ei.value.code == N.E_ARG
    with pytest.raises(N.TkzError) as ei:
        enc.decode_batch(many, np.array([0, 2500]), out_cap=10)
    assert ei.value.code == N.E_CAPACITY


def check_piece_granular(lib, O, vocab, ovocab, pattern, seed=3, rounds=6):
    """tkz_encode_batch_pieces_utf8 (piece offsets built on the device from the bitmap, token marks per piece) vs the oracle's
    split + per-piece encode: the pieces of every document, their byte spans and their token ranges."""
    alpha = RC.alphabet()
    rng = random.Random(seed + pattern)
    enc = N.Encoder(vocab, pattern)
    for it in range(rounds):
        docs = [gen_text(rng, rng.choice(["mix", "a_mix", "a_ws", "oth", "a_dig"]), rng.choice([0, 0, 1, 30, 700, 5000]), alpha).encode("utf-8")
                for _ in range(rng.choice([1, 4, 50]))]
        data, offs = pack(docs)
        ids, dpo, pbo, pto = enc.encode_batch_pieces(data, offs)
        e_dpo, e_pbo, e_pto, e_ids = [0], [], [0], []
        pos = 0
        for d in docs:
            for (a, n) in O.split_utf8(pattern, d):
                p = d[a:a + n]
                r = ovocab.rank(p)
                e_ids += [r] if r >= 0 else ovocab.bpe(p)
                e_pbo.append(pos + a)
                e_pto.append(len(e_ids))
            e_dpo.append(len(e_pbo))
            pos += len(d)
        e_pbo.append(pos)
        assert dpo.tolist() == e_dpo and pbo.tolist() == e_pbo and pto.tolist() == e_pto and ids.tolist() == e_ids, "pattern %d round %d" % (pattern, it)


def check_host_chunks(lib, O, vocab, ovocab, pattern=N.CL100K, seed=21):
    """The chunked, overlapped host path of tkz_encode_batch_utf8 (document ranges on three streams, two staging sets) gives the
    ids and offsets of the one-piece path: empty documents on chunk edges, a chunk that is one long document, the capacity error
    with the total requirement, bad offsets.  ($TKZ_HOST_CHUNK_BYTES must be small for the batch to be cut at all.)"""
    alpha = RC.alphabet()
    rng = random.Random(seed)
    enc = N.Encoder(vocab, pattern)
    oenc = O.Encoder(ovocab, pattern)
    for it in range(4):
        docs = []
        for _ in range(rng.choice([30, 120])):
            k = rng.random()
            docs.append(b"" if k < 0.15 else gen_text(rng, rng.choice(["mix", "a_mix", "ws"]), rng.choice([1, 50, 700, 3000, 9000]), alpha).encode("utf-8"))
        data, offs = pack(docs)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, docs)
        assert ids.tolist() == exp and ooff.tolist() == eoff, "round %d" % it
        # capacity: one id short -> TKZ_E_CAPACITY and the whole batch's requirement
        import ctypes as C
        small = np.empty(max(1, len(exp) - 1), np.int32)
        oo = np.empty(len(docs) + 1, np.int64)
        needed = C.c_int64(0)
        st = lib.L.tkz_encode_batch_utf8(enc._h, data.ctypes.data, offs.ctypes.data, len(docs), small.ctypes.data, len(exp) - 1, oo.ctypes.data, C.byref(needed))
        if len(exp) > 0:
            assert st == N.E_CAPACITY and needed.value == len(exp), (st, needed.value, len(exp))
    with pytest.raises(N.TkzError) as ei:
        enc.encode_batch(np.frombuffer(b"x" * 40000, np.uint8), np.array([0, 30000, 20000, 40000]))
    assert ei.value.code == N.E_ARG
    # an error in a LATER chunk while the chunks behind it are already enqueued (round 6: two launch sequences ahead): the call reports it, every chunk
    # that was begun is ended, and the encoder serves the next call as if nothing had happened -- on both workspaces of the pipeline
    good = [gen_text(rng, "mix", 3000, alpha).encode("utf-8") for _ in range(12)]
    for bad_at in (0, 3, 7, 11):
        docs = list(good)
        docs[bad_at] = docs[bad_at][:1500] + b"\xC3(" + docs[bad_at][1500:]            # a lead byte without its continuation
        data, offs = pack(docs)
        with pytest.raises(N.TkzError) as ei:
            enc.encode_batch(data, offs)
        assert ei.value.code == N.E_INVALID_UTF8, (bad_at, ei.value.code)
        data, offs = pack(good)
        ids, ooff = enc.encode_batch(data, offs)
        exp, eoff = oracle_encode_docs(oenc, good)
        assert ids.tolist() == exp and ooff.tolist() == eoff, bad_at
    # UTF-16 batches through the chunks (round 6: the chunk loop serves that entry too): a document that spans a chunk cut -- or several -- leaves chunks
    # WITHOUT any document or unit: no kernel may be launched on nothing (a grid of 0 workgroups is an invalid configuration on HIP; the emulator aborts on it)
    for shape in ("one", "last", "first", "empties"):
        long_doc = RC.to_units(RC.random_text(rng, alpha, 9000))
        small = [RC.to_units(RC.random_text(rng, alpha, rng.randint(0, 60))) for _ in range(6)]
        docs16 = {"one": [long_doc], "last": small + [long_doc], "first": [long_doc] + small, "empties": [[]] * 3 + [long_doc] + [[]] * 3 + small}[shape]
        flat = np.asarray([u for d in docs16 for u in d], dtype=np.uint16)
        offs16 = np.cumsum([0] + [len(d) for d in docs16]).astype(np.int64)
        ids, ooff = enc.encode_batch_utf16(flat, offs16)
        for d, units in enumerate(docs16):
            assert ids[ooff[d]:ooff[d + 1]].tolist() == oenc.encode_utf16(units), (shape, d)
    # page-locked caller buffers through the chunks: the ids and offsets of every chunk are copied into the caller's arrays at the token base of
    # its chunk
    import ctypes as C
    docs = [b"" if rng.random() < 0.1 else gen_text(rng, rng.choice(["mix", "a_mix", "ws"]), rng.choice([1, 50, 700, 3000]), alpha).encode("utf-8") for _ in range(90)]
    data, offs = pack(docs)
    hp = [C.c_void_p() for _ in range(4)]
    for h, n in zip(hp, [len(data) + 64, 8 * (len(docs) + 1), 4 * len(data) + 64, 8 * (len(docs) + 1)]):
        lib.check(lib.L.tkz_host_alloc(n, C.byref(h)))
    offs64 = np.ascontiguousarray(offs, np.int64)
    C.memmove(hp[0], data.ctypes.data, len(data)); C.memmove(hp[1], offs64.ctypes.data, 8 * (len(docs) + 1))
    needed = C.c_int64(0)
    lib.check(lib.L.tkz_encode_batch_utf8(enc._h, hp[0], hp[1], len(docs), hp[2], len(data), hp[3], C.byref(needed)))
    exp, eoff = oracle_encode_docs(oenc, docs)
    assert np.ctypeslib.as_array(C.cast(hp[2], C.POINTER(C.c_int32)), (needed.value,)).tolist() == exp
    assert np.ctypeslib.as_array(C.cast(hp[3], C.POINTER(C.c_int64)), (len(docs) + 1,)).tolist() == eoff
    for h in hp:
        lib.L.tkz_host_free(h)


def check_begin_end(lib, O, vocab, ovocab, pattern=N.CL100K, seed=61, upload=None, streams=(0,)):
    """tkz_encode_batch_device_begin / _end: several batches in flight on one encoder, ended in another order than they began; a batch whose
    first attempt overflows the miss lists is run again inside _end.  `upload(np_array) -> (owner, pointer)` puts an array where the
    device entry points can read it (identity on the CPU-emulated build); outputs are read back through owner.cpu() if it has one."""
    rng = random.Random(seed)
    alpha = RC.alphabet()
    enc = N.Encoder(vocab, pattern)
    # (the first begun batch of a few kilobytes is a LEARNING batch -- it counts the memo's hits -- and its _end, the last one below, promotes the
    #  hottest entries into the key tables while nothing else is in flight any more; the batches after that run on the promoted tables)
    enc.set_option(N.OPT_PROMOTE_MIN_BYTES, 3000)
    oenc = O.Encoder(ovocab, pattern)
    if upload is None:
        upload = lambda a: (a, a.ctypes.data)
    cons = "bcdfghjklmnpqrstvwxz"
    batches = []
    for k in range(4):
        if k == 2:      # ~340 three-byte misses per KiB: the lists (64 entries a sub-tile in a fresh workspace) overflow, _end runs the batch again
            docs = [("".join(" " + rng.choice(cons) + rng.choice(cons) for _ in range(1500))).encode() for _ in range(3)]
        else:
            docs = [gen_text(rng, rng.choice(["mix", "a_mix", "ws"]), rng.choice([1, 60, 900, 4000]), alpha).encode("utf-8") for _ in range(rng.choice([1, 9, 40]))]
        data, offs = pack(docs)
        padded = np.zeros(len(data) + 64, np.uint8); padded[:len(data)] = data
        ids = np.zeros(max(1, len(data)), np.int32); ooff = np.zeros(len(docs) + 1, np.int64)
        b = dict(docs=docs, n=len(docs), total=len(data), bytes=upload(padded), offs=upload(offs.astype(np.int64)), ids=upload(ids), ooff=upload(ooff),
                 counts=upload(np.full(3, -1, np.int64)))
        # every batch its own {n_docs, n_bytes, n_tokens} block (batch 3: none given, the handle's workspace block)
        b["h"] = enc.encode_batch_device_begin(b["bytes"][1], b["offs"][1], b["n"], b["total"], b["ids"][1], max(1, b["total"]), b["ooff"][1], streams[k % len(streams)],
                                               d_counts3=0 if k == 3 else b["counts"][1])
        assert enc.pending_counts_device(b["h"]) == (b["counts"][1] if k != 3 else enc.pending_counts_device(b["h"])) and enc.pending_counts_device(b["h"])
        batches.append(b)
    back = lambda o: (o.cpu().numpy() if hasattr(o, "cpu") else o)
    for b in batches[::-1]:
        ntok = enc.encode_batch_device_end(b["h"])
        exp, eoff = oracle_encode_docs(oenc, b["docs"])
        assert ntok == len(exp) and back(b["ids"][0])[:ntok].tolist() == exp and back(b["ooff"][0]).tolist() == eoff
        b["ntok"] = ntok
    # with all four ended (in reverse order, one of them run twice): every block holds ITS batch's counts
    for k, b in enumerate(batches):
        if k != 3:
            assert back(b["counts"][0]).tolist() == [b["n"], b["total"], b["ntok"]], k
    # the same four again, begun and ended in order: whatever the learning batch promoted is in the tables now
    for b in batches:
        h = enc.encode_batch_device_begin(b["bytes"][1], b["offs"][1], b["n"], b["total"], b["ids"][1], max(1, b["total"]), b["ooff"][1], streams[0])
        ntok = enc.encode_batch_device_end(h)
        exp, eoff = oracle_encode_docs(oenc, b["docs"])
        assert ntok == len(exp) and back(b["ids"][0])[:ntok].tolist() == exp and back(b["ooff"][0]).tolist() == eoff
    # an empty batch: _begin does not wait, the counts and offsets are there after _end
    ec, eo = upload(np.full(3, -1, np.int64)), upload(np.full(3, -1, np.int64))
    eb, ei_ = upload(np.zeros(64, np.uint8)), upload(np.zeros(3, np.int64))
    h = enc.encode_batch_device_begin(eb[1], ei_[1], 2, 0, 0, 0, eo[1], streams[0], d_counts3=ec[1])
    assert enc.encode_batch_device_end(h) == 0 and back(ec[0]).tolist() == [2, 0, 0] and back(eo[0]).tolist() == [0, 0, 0]
    # an error is reported by _end (and the handle is gone either way): offsets that do not end at the byte count
    data, offs = pack([b"abc", b"defg"])
    bad = offs.astype(np.int64).copy(); bad[-1] += 1
    padded = np.zeros(64, np.uint8); padded[:len(data)] = data
    bb, bo, bi, boo = upload(padded), upload(bad), upload(np.zeros(16, np.int32)), upload(np.zeros(3, np.int64))
    h = enc.encode_batch_device_begin(bb[1], bo[1], 2, len(data), bi[1], 16, boo[1], streams[0])
    with pytest.raises(N.TkzError) as ei:
        enc.encode_batch_device_end(h)
    assert ei.value.code == N.E_ARG
    # tkz_encoder_destroy with a handle outstanding is deferred: the handle's _end reports it and frees the encoder
    enc2 = N.Encoder(vocab, pattern)
    b = batches[0]
    h = enc2.encode_batch_device_begin(b["bytes"][1], b["offs"][1], b["n"], b["total"], b["ids"][1], max(1, b["total"]), b["ooff"][1], streams[0])
    enc2.close()
    tot = C.c_int64(0)
    assert lib.L.tkz_encode_batch_device_end(h, C.byref(tot)) == N.E_ARG
