#!/usr/bin/env python3
"""Soak of the lane merger of the 17..128-byte pieces (tkz_bpe_lane_u up to 64 bytes, tkz_bpe_lane_varc beyond) on random rank tables that are NOT trained
vocabularies -- ties, new pairs ranked below the pair just merged, sparse ranks up to 2^26 (the form with an ids[] array) -- and on the trained tables, through
BOTH forms of the long-miss kernel: the class queue of the large batches (TKZ_LATENCY_BYTES=0: k_long_count / k_long_scatter / k_merge_long_q) and the chunk
form of the small ones.  Every piece against the oracle's literal loop.  CPU-emulated kernels by default; TKZ_EMU_LIB=tokenizer_amd/lib/libtkz.so on a GPU box.
usage: lane_fuzz.py [seconds] [first seed]"""
import gzip, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu
import parity
from tokenizer_amd import _native as N
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lib = N.Library(os.environ["TKZ_EMU_LIB"]) if os.environ.get("TKZ_EMU_LIB") else emu.library()
trained = {}
for name in ("gpt2", "synth100k"):
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", name + ".tiktoken.gz"), "rb").read())
    trained[name] = (N.Vocab(raw, lib), O.Vocab(raw))
t0 = time.time(); rounds = 0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    os.environ["TKZ_LATENCY_BYTES"] = rng.choice(["0", "0", str(16 << 20)])          # (read when an encoder is created: the queue form twice as often)
    lens = sorted(rng.sample(range(13, 131), 10)) + [16, 17, 32, 33, 64, 65, 128]
    parity.check_random_vocab(lib, O, seed, n_vocabs=3, lens=lens, n_pieces=rng.choice([40, 200, 700]), max_len=rng.choice([3, 6, 12]))
    v, ov = trained[rng.choice(list(trained))]
    parity.check_pieces(lib, O, v, ov, seed=seed, rounds=1, lens=lens, counts=[rng.choice([30, 300, 1500])], p_listed=rng.choice([0.3, 1.0]))
    seed += 1; rounds += 1
print("lane_fuzz: %d rounds (seeds %d..%d) in %.0f s, all pieces equal to the oracle's" % (rounds, seed - rounds, seed - 1, time.time() - t0))
