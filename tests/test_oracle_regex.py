"""The oracle's hand-written split (oracle/tkz_oracle.c) against an independent backtracking regex
engine (`regex`) fed UTF-16 code units -- see tests/regex_crosscheck.py for how .NET's code-unit
semantics are emulated.  This is what stands behind the split for cl100k / o200k, whose id-level
vectors need vocab files the reference downloads (parity "unpinned at id level" without them)."""
import random

import pytest

import regex_crosscheck as RC


@pytest.mark.parametrize("pattern", [1, 2, 3, 4])
def test_split_units_random(oracle_mod, pattern):
    rng = random.Random(1234 + pattern)
    alpha = RC.alphabet()
    for it in range(1500):
        s = RC.random_text(rng, alpha, rng.randint(0, 40))
        units = RC.to_units(s)
        assert oracle_mod.split_utf16(pattern, units) == RC.split_units_regex(pattern, units), repr(s)


@pytest.mark.parametrize("pattern", [1, 2, 3, 4])
def test_split_utf8_equals_utf16_on_valid_text(oracle_mod, pattern):
    rng = random.Random(99 + pattern)
    alpha = RC.alphabet()
    for it in range(300):
        s = RC.random_text(rng, alpha, rng.randint(0, 60))
        units = RC.to_units(s)
        b = s.encode("utf-8")
        pieces8 = oracle_mod.split_utf8(pattern, b)
        texts8 = [b[a:a + n] for a, n in pieces8]
        pieces16 = oracle_mod.split_utf16(pattern, units)
        texts16 = ["".join(map(chr, units[a:a + n])).encode("utf-16", "surrogatepass").decode("utf-16").encode("utf-8")
                   for a, n in pieces16]
        assert texts8 == texts16
