"""Soak of tkz_bpe_long_tail (batches of proposals with local bounds, rounds for chains of equal pairs) on the CPU-emulated kernels: random rank tables that are not trained vocabularies (new pairs rank below the pair just merged, ranks tie, sparse ranks), short and long
keys (the window of the local bound is the longest key; beyond 1024 bytes the bound is global), tiny alphabets, pieces on all entry points (k_merge_coop:
257..1024 bytes; the giant pieces' workgroup with the state in LDS <= 16 Ki parts, with the ids left in the pool beyond, after rounds in global memory
beyond 32 Ki), each piece against the oracle's literal loop.  usage: tail_fuzz.py [seconds] [first seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu
import parity
from tokenizer_amd import _native as N
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = N.Library(os.environ["TKZ_EMU_LIB"]) if os.environ.get("TKZ_EMU_LIB") else emu.library()       # (TKZ_EMU_LIB: another emulated build)
t0 = time.time(); rounds = 0; total = 0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    alphabet = rng.choice([b"ab", b"abc", b"abcd", b"abcdefgh"])
    max_len = rng.choice([2, 3, 4, 6, 9, 14, 40, 300, 1100])
    big = rng.random() < 0.3
    n_keys = rng.choice([20, 100, 400, 2000])
    raw = parity.random_vocab_bytes(rng, alphabet=alphabet, n_keys=n_keys, max_len=max_len,
                                    rank_step=(97_003 if n_keys <= 400 else 50_021) if big else 1, rank_base=4_200_000 if big else 0)     # (ranks stay below 2^27)
    vocab, ovocab = N.Vocab(raw, lib), O.Vocab(raw)
    enc = N.Encoder(vocab, N.CL100K)
    lens = [rng.choice([257, 300, 511, 777, 1024, 1030, 1100, 1500, 2300, 4000, 7000, 12000]) for _ in range(4)] + ([rng.choice([16500, 20000, 33000])] if rng.random() < 0.3 else [])
    used = alphabet[:rng.randint(1, len(alphabet))]
    pcs = [bytes(rng.choice(used) for _ in range(n)) for n in lens]
    data, offs = parity.pack(pcs)
    ids, ooff = enc.encode_pieces(data, offs)
    for i, p in enumerate(pcs):
        r = ovocab.rank(p)
        x = [r] if r >= 0 else ovocab.bpe(p)
        g = ids[ooff[i]:ooff[i + 1]].tolist()
        if g != x:
            print("MISMATCH seed", seed, "piece", i, "len", len(p), "alphabet", alphabet, "max_len", max_len); sys.exit(1)
    rounds += 1; total += len(data); seed += 1
    print("seed", seed - 1, "ok:", alphabet, "max_len", max_len, "keys", n_keys, "lens", lens, "%.0f s" % (time.time() - t0), flush=True)
print("tail fuzz ok: %d vocabularies, %.2f MB, next seed %d" % (rounds, total / 1e6, seed))
