"""The reference's own unit tests (Tokenizer_C#/TokenizerTest/TikTokenizerUnitTest.cs), restated against the
host mirror tokenizer_amd.TikTokenizer.  The C# suite builds a cl100k tokenizer from a downloaded rank file;
offline the same scenarios run with the gpt2 rank file (the vocabulary the reference ships), the expected ids
coming from the golden gpt2 vector and from the oracle.  With $TKZ_VOCAB_DIR/cl100k_base.tiktoken present the
literal cl100k assertions of the C# file run as written."""
import os

import numpy as np

from conftest import find_vocab_file, load_golden_json
from tokenizer_amd import REGEX_CL100K, REGEX_PATTERN_1, TokenizerBuilder

IM_START, IM_END = "<|im_start|>", "<|im_end|>"


def run_gpt2_suite(lib, gpt2_bytes, lib_rs_text, oracle_mod, oracle_vocab):
    specials = {"<|endoftext|>": 50256, IM_START: 50300, IM_END: 50301}
    tok = TokenizerBuilder.CreateTokenizer(gpt2_bytes, specials, REGEX_PATTERN_1, lib=lib)
    oenc = oracle_mod.Encoder(oracle_vocab, oracle_mod.P1, specials=specials)
    golden = load_golden_json("tokens_gpt2.json")
    # TestEncode0 / TestEncode4-style: plain text, round trip
    enc = tok.Encode("Hello World")
    assert enc == oenc.encode("Hello World") and tok.Decode(enc) == "Hello World"
    assert tok.Encode("") == []                                           # TikTokenizerUnitTest.cs:103-109
    # TestEncode1 / TestEncode3: specials honoured by default and by an explicit allow-set (:52-64, :89-101)
    text = IM_START + "Hello World" + IM_END
    e1 = tok.Encode(text)
    assert e1[0] == 50300 and e1[-1] == 50301 and e1[1:-1] == enc
    assert tok.Encode(text, [IM_START, IM_END]) == e1
    assert tok.Decode(e1) == text
    # an allow-set that leaves a registered special out: that literal is plain text (FindNextSpecialToken skips it, :233-239)
    only_end = tok.Encode(text, [IM_END])
    assert only_end == oenc.encode(text, [IM_END]) and only_end[-1] == 50301 and only_end[0] != 50300
    # applySpecialTokens = false: everything is plain text (:201-205)
    assert tok.Encode(text, False) == oenc.encode(text) and 50300 not in tok.Encode(text, False)
    # TestEncode2: the long document, through both overloads (:66-87)
    assert tok.Encode(lib_rs_text, [IM_START, IM_END]) == golden
    assert tok.Encode(lib_rs_text, False) == golden
    assert tok.Decode(golden) == lib_rs_text
    # TestEncode5: a multi-byte char between specials (:112-126)
    t5 = IM_START + "Hello ⭐ World" + IM_END
    e5 = tok.Encode(t5, [IM_START, IM_END])
    assert e5 == oenc.encode(t5, [IM_START, IM_END]) and tok.Decode(e5) == t5
    # overlapping / adjacent / trailing specials and a special as the whole text
    for t in (IM_START, IM_START + IM_END, "a" + IM_START, IM_END + "b" + IM_END + IM_END, "<|im_start", "<|endoftext|><|endoftext|>x"):
        assert tok.Encode(t) == oenc.encode(t, list(specials)), t
    # EncodeBatch == [Encode(t) for t in texts], mixed empty / special / long inputs
    texts = ["", "Hello World", text, lib_rs_text[:3000], t5, IM_END, " ", "x" * 300]
    assert tok.EncodeBatch(texts) == [tok.Encode(t) for t in texts]
    assert tok.EncodeBatch(texts, False) == [oenc.encode(t) for t in texts]
    assert tok.EncodeBatch([]) == []
    # EncodeBatchFlat: (ids, offsets) without a list per text -- the plain path hands the device call's arrays back, the special path splices
    for apply in (False, True, [IM_END]):
        fids, foffs = tok.EncodeBatchFlat(texts, apply)
        want = tok.EncodeBatch(texts, apply)
        assert foffs[0] == 0 and len(foffs) == len(texts) + 1 and foffs[-1] == sum(map(len, want)) and fids.dtype == np.int32
        assert [fids[foffs[d]:foffs[d + 1]].tolist() for d in range(len(texts))] == want
    fids, foffs = tok.EncodeBatchFlat([])
    assert len(fids) == 0 and foffs.tolist() == [0]
    # the o200k string through the two engines the reference runs it with (host="dotnet": TikTokenizer.cs:77; host="js": tikTokenizer.ts:100)
    from tokenizer_amd import REGEX_O200K
    import pytest
    net = TokenizerBuilder.CreateTokenizer(gpt2_bytes, {}, REGEX_O200K, lib=lib)
    js = TokenizerBuilder.CreateTokenizer(gpt2_bytes, {}, REGEX_O200K, lib=lib, host="js")
    o_net, o_js = oracle_mod.Encoder(oracle_vocab, oracle_mod.O200K_DOTNET), oracle_mod.Encoder(oracle_vocab, oracle_mod.O200K)
    two = ("\U0001d400bc fooBAR's", "a\x85\x85b", "\ufeffx a\ufeff\ufeffb", "\U0001F600\u4e2d x\U0001F600\u0301y", "1\U0001d7cf\U0001d7d0\U0001d7d1\U0001d7d2 \U0001F600abc", lib_rs_text[:2000])
    for t in two:
        assert net.Encode(t, False) == o_net.encode(t) and js.Encode(t, False) == o_js.encode(t), t
    # (the two readings cut such text into different pieces; with a byte-level table like gpt2's the ids often coincide all the same --
    #  no merge crosses from the bytes of a supplementary-plane char into a letter -- so the difference is asserted on the pieces)
    b = two[0].encode("utf-8")
    arr = np.frombuffer(b, np.uint8)
    assert not np.array_equal(net.native.pretokenize(arr, np.array([0, len(b)])), js.native.pretokenize(arr, np.array([0, len(b)])))
    with pytest.raises(NotImplementedError):          # patterns 1 / cl100k exist with the C# engine's semantics only
        TokenizerBuilder.CreateTokenizer(gpt2_bytes, {}, REGEX_CL100K, lib=lib, host="js")
    # lone surrogates: Encoding.UTF8.GetBytes semantics
    s = "a\ud800b"
    assert tok.Encode(s, False) == oenc.encode_bytes("a�b".encode("utf-8"))
    run_trim_suite(tok, oracle_mod.TrimOracle(oracle_vocab, oracle_mod.P1, specials), specials, lib_rs_text)


def run_by_name_suite(lib, vocab_bytes, oracle_mod, tmpdir):
    """CreateByEncoderName / CreateByModelName pick the engine of the reference that DEFINES the encoder (round-4 advisor): o200k_base / gpt-4o exist
    only in the TypeScript reference -- code points, ECMAScript \\s (tokenizer_ts/src/tikTokenizer.ts:100) --, cl100k_base and the pattern-1 encoders
    in the C# one.  Rank files: the stand-ins of the right size under the names the builders look for (the real ones are not available offline)."""
    from tokenizer_amd import tokenizer as TK
    d = str(tmpdir)
    raw200, raw100 = vocab_bytes("synth200k"), vocab_bytes("synth100k")
    open(os.path.join(d, "o200k_base.tiktoken"), "wb").write(raw200)
    open(os.path.join(d, "cl100k_base.tiktoken"), "wb").write(raw100)
    ov200 = oracle_mod.Vocab(raw200)
    o_js, o_net = oracle_mod.Encoder(ov200, oracle_mod.O200K), oracle_mod.Encoder(ov200, oracle_mod.O200K_DOTNET)
    # text on which the two engines cut differently: supplementary-plane letters, digits and marks, U+0085, U+FEFF
    texts = ["\U0001d400bc fooBAR's", "a\x85\x85b", "\ufeffx a\ufeff\ufeffb", "\U0001F600\u4e2d x\U0001F600\u0301y", "1\U0001d7cf\U0001d7d0\U0001d7d1\U0001d7d2 \U0001F600abc",
             "\U00010400\U00010428\U00010400 \U0001e900\U0001e922 x\U000e0100y", "plain ASCII text, it's 12345 fine\r\n"]
    by_enc = TK.TokenizerBuilder.CreateByEncoderName("o200k_base", vocab_dir=d, lib=lib)
    by_model = TK.TokenizerBuilder.CreateByModelName("gpt-4o", vocab_dir=d, lib=lib)
    by_prefix = TK.TokenizerBuilder.CreateByModelName("gpt-4o-mini", vocab_dir=d, lib=lib)
    as_net = TK.TokenizerBuilder.CreateByEncoderName("o200k_base", vocab_dir=d, lib=lib, host="dotnet")
    differ = 0
    for t in texts:
        want_js, want_net = o_js.encode(t), o_net.encode(t)
        assert by_enc.Encode(t, False) == want_js and by_model.Encode(t, False) == want_js and by_prefix.Encode(t, False) == want_js, t
        assert as_net.Encode(t, False) == want_net, t
        b = t.encode("utf-8")
        arr = np.frombuffer(b, np.uint8)
        differ += int(not np.array_equal(by_enc.native.pretokenize(arr, np.array([0, len(b)])), as_net.native.pretokenize(arr, np.array([0, len(b)]))))
    assert differ >= 5                      # the by-name default is not the .NET reading on such text
    assert by_enc.SpecialTokensEncoder == {"<|endoftext|>": 199999, "<|endofprompt|>": 200018}
    # cl100k by name: the C# engine's reading, the only one there is
    ov100 = oracle_mod.Vocab(raw100)
    c = TK.TokenizerBuilder.CreateByModelName("gpt-4", vocab_dir=d, lib=lib)
    oc = oracle_mod.Encoder(ov100, oracle_mod.CL100K)
    for t in texts:
        assert c.Encode(t, False) == oc.encode(t), t
    import pytest
    with pytest.raises(NotImplementedError):
        TK.TokenizerBuilder.CreateByEncoderName("cl100k_base", vocab_dir=d, lib=lib, host="js")
    with pytest.raises(NotImplementedError):
        TK.TokenizerBuilder.CreateByModelName("no-such-model", vocab_dir=d, lib=lib)


def run_host_runtime_suite(lib, gpt2_bytes, oracle_mod, oracle_vocab):
    """The mirror's two knobs for a host on another runtime than net6.0: this Python's own Unicode tables (a newer version than the 13.0 libtkz ships)
    as the class table, and .NET >= 7's (?i:...).  Against the oracle given the same."""
    import unicodedata
    from tokenizer_amd import host_unicode_classes
    classes = host_unicode_classes(n_code_points=65536)
    builtin = np.zeros(65536, np.uint8)
    lib.L.tkz_unicode_classes(0, 65536, builtin.ctypes.data)
    moved = np.flatnonzero(classes[128:] != builtin[128:]) + 128
    moved = moved[(moved < 0xD800) | (moved > 0xDFFF)]
    assert unicodedata.unidata_version >= "13.0.0"
    tok = TokenizerBuilder.CreateTokenizer(gpt2_bytes, {}, REGEX_CL100K, lib=lib, unicode_classes=classes, case_equivalence=True)
    try:
        oracle_mod.set_unicode_classes(classes)
        oracle_mod.set_case_equivalence(True)
        oenc = oracle_mod.Encoder(oracle_vocab, oracle_mod.CL100K)
        texts = ["it'ſ fine'ſabc", "Hello World it's 12345", "".join(chr(int(c)) + "a1 " for c in moved[:400]), "漢字かな交じり文 😀 naïve"]
        for t in texts:
            assert tok.Encode(t, False) == oenc.encode(t), t
        assert tok.EncodeBatch(texts, False) == [oenc.encode(t) for t in texts]
    finally:
        oracle_mod.set_unicode_classes(None)
        oracle_mod.set_case_equivalence(False)
    return int(len(moved))


def run_trim_suite(tok, trim_oracle, specials, long_text):
    """TestEncodeTrimSuffix/2, TestEncodeTrimPrefix/2 (TikTokenizerUnitTest.cs:128-225): the same scenarios, every
    maxTokenCount from 0 past the full length, all three overload shapes, against the oracle's restatement of
    TikTokenizer.cs:288-579; then seeded random texts with specials, CJK, emoji and lone-surrogate-free BMP symbols."""
    import random
    allow = list(specials)
    cases = [IM_START + "Hello World" + IM_END, IM_START + "Hello TempWorld" + IM_END, IM_START + "HelloTemp World" + IM_END,
             "", "a", IM_END, "Hello ⭐ World 😀😀 done" + IM_END + " tail", long_text[:700]]
    rng = random.Random(77)
    frag = ["Hello", " World", " the", "ing", " 1234567", "\n\n", "  ", "'s", " don't", IM_START, IM_END, "<|endoftext|>", "<|im_",
            "漢字かな", "😀", "⭐", " tokenization", "!!!", "\t", "x" * 40]
    for _ in range(25):
        cases.append("".join(rng.choice(frag) for _ in range(rng.randint(1, 14))))
    for text in cases:
        full = len(tok.Encode(text, allow))
        for mx in sorted(set([0, 1, 2, 3, 4, 5, 6, 7, full - 1, full, full + 1, 1000])):
            if mx < 0:
                continue
            for fn, ofn in ((tok.EncodeTrimSuffix, trim_oracle.encode_trim_suffix), (tok.EncodeTrimPrefix, trim_oracle.encode_trim_prefix)):
                assert fn(text, allow, mx) == ofn(text, allow, mx), (fn.__name__, text, mx, "allowed")
                assert fn(text, mx) == ofn(text, allow, mx), (fn.__name__, text, mx, "apply=True")
                assert fn(text, mx, False) == ofn(text, None, mx), (fn.__name__, text, mx, "apply=False")
                assert fn(text, [IM_END], mx) == ofn(text, [IM_END], mx), (fn.__name__, text, mx, "subset")
    # the properties the C# tests assert: the ids decode to the returned text, and never exceed maxTokenCount
    text = IM_START + "Hello World" + IM_END
    ids, cut = tok.EncodeTrimSuffix(text, allow, 3)
    assert cut == IM_START + "Hello World" and tok.Decode(ids) == cut and len(ids) == 3          # :148-152
    ids, cut = tok.EncodeTrimPrefix(text, allow, 3)
    assert cut == "Hello World" + IM_END and tok.Decode(ids) == cut and len(ids) == 3            # :196-200
    assert tok.EncodeTrimSuffix(text, allow, 4) == (tok.Encode(text, allow), text)                # :132-134
    assert tok.EncodeTrimPrefix(text, allow, 5) == (tok.Encode(text, allow), text)                # :192-194


def run_cl100k_suite(lib):
    p = find_vocab_file("cl100k_base.tiktoken")
    if not p:
        return False
    specials = {"<|endoftext|>": 100257, "<|fim_prefix|>": 100258, "<|fim_middle|>": 100259, "<|fim_suffix|>": 100260,
                "<|endofprompt|>": 100276, IM_START: 100264, IM_END: 100265}
    tok = TokenizerBuilder.CreateTokenizer(open(p, "rb").read(), specials, REGEX_CL100K, lib=lib)
    assert tok.Encode("Hello World") == [9906, 4435]                                              # :39-49
    assert tok.Encode(IM_START + "Hello World" + IM_END) == [100264, 9906, 4435, 100265]          # :52-64
    assert tok.Encode(IM_START + "Hello ⭐ World" + IM_END, [IM_START, IM_END]) == [100264, 9906, 2928, 99834, 4435, 100265]   # :112-126
    lib_rs = open(os.path.join(os.path.dirname(__file__), "golden", "lib.rs.txt"), encoding="utf-8").read()
    assert tok.Encode(lib_rs, False) == load_golden_json("tokens_cl100k.json")                    # :66-87
    # TestEncodeTrimSuffix / 2 (:128-173) and TestEncodeTrimPrefix / 2 (:177-225), as written
    allow = list(specials)
    text = IM_START + "Hello World" + IM_END
    for mx, n, cut in ((4, 4, text), (5, 4, text), (3, 3, IM_START + "Hello World")):
        ids, t = tok.EncodeTrimSuffix(text, allow, mx)
        assert (len(ids), t) == (n, cut)
    ids, t = tok.EncodeTrimSuffix(text, 4, False)
    assert (len(ids), t) == (4, "<|im_start")
    assert tok.EncodeTrimSuffix(text, 4)[1] == text
    text2 = IM_START + "Hello TempWorld" + IM_END
    for mx, n, cut in ((5, 5, text2), (6, 5, text2), (3, 2, IM_START + "Hello")):
        ids, t = tok.EncodeTrimSuffix(text2, allow, mx)
        assert (len(ids), t) == (n, cut) and (mx != 3 or tok.Decode(ids) == cut)
    for mx, n, cut in ((4, 4, text), (5, 4, text), (3, 3, "Hello World" + IM_END)):
        ids, t = tok.EncodeTrimPrefix(text, allow, mx)
        assert (len(ids), t) == (n, cut)
    ids, t = tok.EncodeTrimPrefix(text, 4, False)
    assert (len(ids), t) == (4, "im_end|>")
    text3 = IM_START + "HelloTemp World" + IM_END
    for mx, n, cut in ((5, 5, text3), (6, 5, text3), (3, 2, " World" + IM_END)):
        ids, t = tok.EncodeTrimPrefix(text3, allow, mx)
        assert (len(ids), t) == (n, cut) and (mx != 3 or tok.Decode(ids) == cut)
    return True
