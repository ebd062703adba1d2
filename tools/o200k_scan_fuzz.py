#!/usr/bin/env python3
"""Differential soak of the o200k block scanners (ASCII and multi-byte) against the oracle's sequential matcher, on the CPU-emulated
kernels (default) or on the GPU (--gpu).  Alphabets are chosen so that every flow of tkz_block_core_o200k is exercised: case
transitions through runs of Lo/Lm/M chars, marks after punctuation, swallowed '/' runs, contraction suffixes, supplementary-plane
letters, digits and symbols, JS white space.  Prints how many 4 KiB blocks each scanner handed on.

    python tools/o200k_scan_fuzz.py [--gpu] [--pattern 3|4] [--kinds cjk,case,...] [--seeds N]
"""
import argparse
import gzip
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--kinds", default="cjk,case,mark,emoji,slash,chain,upper,all")
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--pattern", type=int, default=3, help="3: the ECMAScript reading of the o200k string (code points), 4: .NET's (code units)")
    args = ap.parse_args()
    import oracle as O
    import parity
    ALPHAS, gen = parity.O200K_ALPHAS, parity.o200k_gen
    from tokenizer_amd import _native as N
    if args.gpu:
        lib = None
    else:
        import emu
        lib = emu.library()
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    vocab = N.Vocab(raw, lib) if lib else N.Vocab(raw)
    enc = N.Encoder(vocab, args.pattern)
    for kind in args.kinds.split(","):
        ta = tb = tblk = 0
        for seed in range(args.first_seed, args.first_seed + args.seeds):
            rng = random.Random(seed * 1000 + len(kind))
            docs = [gen(rng, ALPHAS[kind], rng.choice([3000, 9000, 20000])).encode("utf-8") for _ in range(rng.choice([1, 2, 5]))]
            data, offs = parity.pack(docs)
            got = enc.pretokenize(data, offs)
            exp = parity.oracle_bitmap(O, args.pattern, docs)
            a, b = enc.pretok_leftovers()
            ta += a
            tb += b
            tblk += (len(data) + 3967) // 3968
            if not np.array_equal(got, exp):
                print("FAIL kind=%s seed=%d: %s" % (kind, seed, parity.explain_bitmap_diff(got, exp, docs, offs)))
                return 1
        why = ""
        if lib is not None:                       # the emulated build counts which rule refused
            import ctypes
            arr = (ctypes.c_longlong * 32).in_dll(lib.L, "tkz_o2_refusals")
            why = "  refusals by rule: " + " ".join("%d:%d" % (i, arr[i]) for i in range(32) if arr[i])
            for i in range(32):
                arr[i] = 0
        print("pattern %d %-6s ok: %d blocks, %d left by the ASCII scanner, %d of them left by the multi-byte scanner%s" % (args.pattern, kind, tblk, ta, tb, why))
    return 0


if __name__ == "__main__":
    sys.exit(main())
