"""Soak of the oracle's hand-written split (oracle/tkz_oracle.c: all four patterns) against the independent backtracking engine of tests/regex_crosscheck.py
(Python `regex` fed UTF-16 code units with .NET's / ECMAScript's \\s): random texts over the wide alphabet and over the adversarial small alphabets of
tests/parity.py (white space, digits, apostrophes and contraction suffixes, case transitions, CR / LF / '/', long runs), 0..400 chars, for a wall-clock budget.
What stands behind the cl100k / o200k split while the reference's own id vectors cannot run offline.  usage: oracle_regex_soak.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
import regex_crosscheck as RC
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
alpha = RC.alphabet()
kinds = ["mix"] * 4 + [k for k in parity.SMALL_ALPHAS]
t0 = time.time(); n = 0; units_total = 0; per = {1: 0, 2: 0, 3: 0, 4: 0}
while time.time() - t0 < budget:
    kind = rng.choice(kinds)
    ln = rng.choice([0, 1, 2, 5, 12, 40, 40, 100, 400])
    s = RC.random_text(rng, alpha, ln) if kind == "mix" else parity.gen_text(rng, kind, ln, alpha)
    # (the small alphabets hold chars whose classes Python's tables and Unicode 13 may disagree on: keep what the cross-check's own filter accepts)
    s = "".join(ch for ch in s if ord(ch) < 0x80 or RC._agree(ord(ch)))
    units = RC.to_units(s)
    for pattern in (1, 2, 3, 4):
        a, b = O.split_utf16(pattern, units), RC.split_units_regex(pattern, units)
        if a != b:
            print("MISMATCH pattern", pattern, "kind", kind, repr(s)); print(" oracle", a[:20]); print(" regex ", b[:20]); sys.exit(1)
        per[pattern] += 1
    n += 1; units_total += len(units)
print("oracle vs regex ok: %d texts x 4 patterns, %d code units, seed %d" % (n, units_total, seed))
