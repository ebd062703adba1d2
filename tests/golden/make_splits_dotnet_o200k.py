#!/usr/bin/env python3
"""tests/golden/splits_o200k_dotnet.json: piece boundaries of the o200k regex as the C# reference's engine reads it
(`new Regex(pattern, RegexOptions.Compiled)`, TikTokenizer.cs:77: one class test per UTF-16 code UNIT, .NET's \\s), computed by the third-party
`regex` engine fed code units (tests/regex_crosscheck.py, pattern 4) -- NOT by the oracle: the oracle, the emulated kernels and the GPU are
all held to this file.  Texts: the adversarial strings of splits.json plus strings on which the .NET and the ECMAScript readings differ
(supplementary-plane letters / digits / marks / symbols, U+0085, U+FEFF).  Pieces are [byte offset, byte length] of the UTF-8 form."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import regex_crosscheck as RC   # noqa: E402

CASES = [
    "Hello World", "Hello ⭐ World", "it's don't I'LL we'Ve x'sx ''s 's' 'S a'ta \t's \n'd",
    "abc 123456789 1 22 333 4444 a1b22c333", "  leading\n\n  two\n \n   x \r\n\r\n", "tabs\t\tand  spaces   end   ",
    "p.\n\n  q..\r\nr ...\n", "😀abc 中文😀中 ⭐x .a ..a . a", "١٢٣٤٥ ２０２４年 Ⅻ ½",
    "camelCaseHTMLParser XMLHttpRequest FOO fooBAR 中A Aé", "a b c　d\x85e\x1cf", "",
    " ", "\n", "a", "'", "''''", "x'", "'re're'RE", "é ño", "\U0001F468‍\U0001F469‍\U0001F467 fam",
    # where the two readings of the string part
    "\U0001d400bc \U0001d400\U0001d401 x\U0001d400", "a\U00020000b \U00020000\U00020001", "\U0001d7cf\U0001d7d0\U0001d7d1\U0001d7d2 1\U0001d7cf2",
    "a\x85b a\x85\x85b \x85\x85", "﻿x a﻿﻿b ﻿ ﻿", "x\U0001F600́y \U0001F600́́ á\U0001F600",
    "\U0001F600's it\U0001F600's A\U0001F600B", "//\U0001F600/\n/ ;\n/*\U0001F600", " \U0001F600 \U0001F600a  \U0001F600", "\U00010428\U00010400 \U0001E900\U0001E922",
    "\U000E0100a 中\U000E0100", "fooBAR's HTMLParser'S XMLHttp'll x'eR y'rE",
]


def main():
    out = []
    for s in CASES:
        units = RC.to_units(s)
        # byte offset of every unit boundary (a surrogate pair is one 4-byte char: its low half maps to the same offset as the high half)
        boff, b = [], 0
        i = 0
        while i < len(units):
            u = units[i]
            if 0xD800 <= u <= 0xDBFF and i + 1 < len(units) and 0xDC00 <= units[i + 1] <= 0xDFFF:
                boff += [b, b]; b += 4; i += 2
            else:
                boff.append(b); b += 1 if u < 0x80 else 2 if u < 0x800 else 3; i += 1
        boff.append(b)
        pieces = [[boff[a], boff[a + n] - boff[a]] for a, n in RC.split_units_regex(4, units)]
        out.append({"pattern": 4, "text": s, "pieces": pieces})
    json.dump(out, open(os.path.join(HERE, "splits_o200k_dotnet.json"), "w"), ensure_ascii=True, separators=(",", ":"))
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
