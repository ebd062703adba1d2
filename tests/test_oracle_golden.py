"""Pins the CPU oracle to the reference's own known-answer vectors.

Closes offline: gpt2 vocab (the only vocabulary the reference ships) + pattern 1 on lib.rs.txt
-> tokens_gpt2.json (Tokenizer_C#/TokenizerTest/TikTokenizerUnitTest.cs:227-245, 288-305).
cl100k / p50k / o200k id vectors run only when the vocab file is supplied in $TKZ_VOCAB_DIR.
"""
import json

import pytest

from conftest import find_vocab_file, load_golden_json


def test_gpt2_lib_rs_ids(oracle_mod, oracle_gpt2, lib_rs_bytes):
    exp = load_golden_json("tokens_gpt2.json")
    enc = oracle_mod.Encoder(oracle_gpt2, oracle_mod.P1)
    assert enc.encode_bytes(lib_rs_bytes) == exp
    # memo disabled must not change anything (LRU is a pure memo, LRUCache.cs)
    enc0 = oracle_mod.Encoder(oracle_gpt2, oracle_mod.P1, cache_size=0)
    assert enc0.encode_bytes(lib_rs_bytes) == exp


def test_gpt2_utf16_entry_matches_utf8(oracle_mod, oracle_gpt2, lib_rs_bytes):
    enc = oracle_mod.Encoder(oracle_gpt2, oracle_mod.P1)
    s = lib_rs_bytes.decode("utf-8")
    units = list(s.encode("utf-16-le"))
    u16 = [units[i] | (units[i + 1] << 8) for i in range(0, len(units), 2)]
    assert enc.encode_utf16(u16) == load_golden_json("tokens_gpt2.json")


def test_empty_and_single(oracle_mod, oracle_gpt2):
    enc = oracle_mod.Encoder(oracle_gpt2, oracle_mod.P1)
    assert enc.encode_bytes(b"") == []           # TikTokenizerUnitTest.cs:103-109
    assert enc.encode_bytes(b"!") == [0]         # tokenizer_ts/test/tikTokenizer.test.ts:22-27 (same id in gpt2)


def test_vocab_loader_errors(oracle_mod):
    O = oracle_mod
    with pytest.raises(O.OracleError) as e:
        O.Vocab(b"YQ== 0\nYg== 0\n")            # duplicate rank -> ArgumentException (TikTokenizer.cs:84-87)
    assert e.value.code == O.E_DUP_RANK
    with pytest.raises(O.OracleError) as e:
        O.Vocab(b"YQ== 0 1\n")                   # three fields (TikTokenizer.cs:114-118)
    assert e.value.code == O.E_FORMAT
    with pytest.raises(O.OracleError) as e:
        O.Vocab(b"YQ== x\n")                     # rank not an int (:122-129)
    assert e.value.code == O.E_FORMAT
    with pytest.raises(O.OracleError) as e:
        O.Vocab(b"Y!== 1\n")                     # bad base64 (FormatException -> InvalidOperationException)
    assert e.value.code == O.E_FORMAT
    v = O.Vocab(b"\nYQ== 0\n\r\n  \nYg== 1\n")   # blank lines skipped (:109-112)
    assert len(v) == 2 and v.rank(b"a") == 0 and v.rank(b"b") == 1
    assert len(O.Vocab(b"YQ== 0\n\xc2\x85\xc2\xa0 \xe3\x80\x80\nYg== 1\n")) == 2   # UTF-8 StreamReader: these chars are white space
    with pytest.raises(O.OracleError) as e:
        O.Vocab(b"YQ== 0\n\x85\nYg== 1\n")     # ... a lone byte 0x85 is U+FFFD, which is not
    assert e.value.code == O.E_FORMAT


def test_missing_single_byte_is_key_not_found(oracle_mod):
    O = oracle_mod
    v = O.Vocab(b"YQ== 0\nYWI= 1\n")            # 'a', 'ab' only
    enc = O.Encoder(v, O.P1)
    assert enc.encode_bytes(b"ab") == [1]
    with pytest.raises(O.OracleError) as e:      # BytePairEncoder.cs:17 / :73
        enc.encode_bytes(b"b")
    assert e.value.code == O.E_KEY_NOT_FOUND


def test_golden_splits_regression(oracle_mod):
    # splits.json: oracle outputs of an earlier build (a regression net; cross-checked against `regex` in test_oracle_regex.py);
    # splits_o200k_dotnet.json: the o200k string read by code unit with .NET's \\s, computed by `regex`, not by the oracle (a pin)
    for rec in load_golden_json("splits.json") + load_golden_json("splits_o200k_dotnet.json"):
        got = oracle_mod.split_utf8(rec["pattern"], rec["text"].encode("utf-8"))
        assert [list(p) for p in got] == rec["pieces"], rec["text"]


@pytest.mark.parametrize("vocab,pattern,fixture", [
    ("cl100k_base.tiktoken", 2, "tokens_cl100k.json"),
    ("p50k_base.tiktoken", 1, "tokens_p50k.json"),
    ("o200k_base.tiktoken", 3, "tokens_o200k.json"),
    ("o200k_base.tiktoken", 4, "tokens_o200k.json"),        # (the test text is ASCII: both readings of the o200k string give the same ids)
])
def test_downloaded_vocab_vectors(oracle_mod, lib_rs_bytes, vocab, pattern, fixture):
    p = find_vocab_file(vocab)
    if not p:
        pytest.skip("%s not supplied (reference downloads it at run time); set TKZ_VOCAB_DIR" % vocab)
    v = oracle_mod.Vocab(open(p, "rb").read())
    enc = oracle_mod.Encoder(v, pattern)
    assert enc.encode_bytes(lib_rs_bytes) == load_golden_json(fixture)


def test_hand_derived_splits(oracle_mod):
    """Expected pieces written by hand from the regex semantics (tests/hand_splits.py), not produced by any implementation."""
    from hand_splits import HAND_SPLITS
    for pat, text, exp in HAND_SPLITS:
        b = text.encode("utf-8")
        got = [b[a:a + n].decode("utf-8") for a, n in oracle_mod.split_utf8(pat, b)]
        assert got == exp, (pat, text)
        units = list(text.encode("utf-16-le"))
        units = [units[i] | (units[i + 1] << 8) for i in range(0, len(units), 2)]
        got16 = [bytes(x for u in units[a:a + n] for x in (u & 255, u >> 8)).decode("utf-16-le") for a, n in oracle_mod.split_utf16(pat, units)]
        assert got16 == exp, (pat, text, "utf16")
