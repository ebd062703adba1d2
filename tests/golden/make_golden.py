#!/usr/bin/env python3
"""Collect the reference's own known-answer DATA into tests/golden/ (run in the build container only).

What is taken (data only -- inputs and expected outputs, no reference source code):
  * model/gpt2.tiktoken                       the one vocabulary file the reference ships (rank table, data);
                                              stored gzip'ed
  * Tokenizer_C#/TokenizerTest/testData/lib.rs.txt          the test INPUT text of TestEncode2 & co
  * .../tokens_gpt2.json (== tokens_r50k_base.json)         expected ids, pattern 1 + gpt2 vocab
  * .../tokens.json                                          expected ids, cl100k_base (needs the cl100k vocab)
  * .../tokens_p50k_base.json                                expected ids, p50k_base  (needs the p50k vocab)
  * tokenizer_ts/test/testdata/tokens_gpt_4o.json            expected ids, o200k_base (needs the o200k vocab)
and what is generated here with the oracle (oracle outputs, re-checked against an independent
regex engine in tests/test_oracle_regex.py):
  * splits.json   piece boundaries of adversarial strings under the three patterns
"""
import gzip
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def main():
    cs = os.path.join(REF, "Tokenizer_C#", "TokenizerTest", "testData")
    ts = os.path.join(REF, "tokenizer_ts", "test", "testdata")
    with open(os.path.join(REF, "model", "gpt2.tiktoken"), "rb") as f:
        raw = f.read()
    with open(os.path.join(HERE, "gpt2.tiktoken.gz"), "wb") as f:
        f.write(gzip.compress(raw, 9, mtime=0))
    shutil.copyfile(os.path.join(cs, "lib.rs.txt"), os.path.join(HERE, "lib.rs.txt"))
    for src, dst in ((os.path.join(cs, "tokens_gpt2.json"), "tokens_gpt2.json"),
                     (os.path.join(cs, "tokens.json"), "tokens_cl100k.json"),
                     (os.path.join(cs, "tokens_p50k_base.json"), "tokens_p50k.json"),
                     (os.path.join(ts, "tokens_gpt_4o.json"), "tokens_o200k.json")):
        ids = json.load(open(src))
        json.dump(ids, open(os.path.join(HERE, dst), "w"), separators=(",", ":"))

    from oracle import oracle as O
    cases = [
        "Hello World", "Hello ⭐ World", "it's don't I'LL we'Ve x'sx ''s 's' 'S a'ta \t's \n'd",
        "abc 123456789 1 22 333 4444 a1b22c333", "  leading\n\n  two\n \n   x \r\n\r\n", "tabs\t\tand  spaces   end   ",
        "p.\n\n  q..\r\nr ...\n", "😀abc 中文😀中 ⭐x .a ..a . a", "١٢٣٤٥ ２０２４年 Ⅻ ½",
        "camelCaseHTMLParser XMLHttpRequest FOO fooBAR 中A Aé", "a b c　d\x85e\x1cf", "",
        " ", "\n", "a", "'", "''''", "x'", "'re're'RE", "é ño", "\U0001F468‍\U0001F469‍\U0001F467 fam",
    ]
    out = []
    for pat in (O.P1, O.CL100K, O.O200K):
        for s in cases:
            out.append({"pattern": pat, "text": s, "pieces": O.split_utf8(pat, s.encode("utf-8"))})
    json.dump(out, open(os.path.join(HERE, "splits.json"), "w"), ensure_ascii=True, separators=(",", ":"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
