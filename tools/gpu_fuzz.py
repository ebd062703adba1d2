"""Soak test on the GPU: random batches (wide Unicode mix, adversarial small alphabets, long single-class runs, corpus kinds)
through tkz_encode_batch_utf8 against the oracle, all four patterns (both readings of the o200k string), for a fixed wall-clock budget.  usage: gpu_fuzz.py [seconds] [seed]
(TKZ_FUZZ_EMU=1: the same soak on the CPU-emulated kernels, with batches a hundredth the size.)"""
import gzip, os, random, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
EMU = os.environ.get("TKZ_FUZZ_EMU") == "1"
if EMU:
    import emu
else:
    import torch  # noqa: F401  (initialises the HIP runtime before libtkz binds to it)
import parity
import regex_crosscheck as RC
from tokenizer_amd import _native as N
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
vocab, ovocab = (N.Vocab(raw, emu.library()) if EMU else N.Vocab(raw)), O.Vocab(raw)
encs = {p: N.Encoder(vocab, p) for p in (1, 2, 3, 4)}
# (two of the four encoders learn their promotions from the soak's own batches -- the default threshold, 8 MB, is above most of them --: the batches after that run
#  on key tables with promoted pieces in them, rebuilt behind a batch that is still being compared)
for p in (2, 4):
    encs[p].set_option(N.OPT_PROMOTE_MIN_BYTES, 2000 if EMU else 200000)
alpha = RC.alphabet()
rng = random.Random(seed)
kinds = ["mix", "runs"] + list(parity.SMALL_ALPHAS)
t0 = time.time(); rounds = 0; total = 0
while time.time() - t0 < budget:
    pattern = rng.choice((1, 2, 3, 4))
    k = rng.random()
    if k < 0.25:
        kind = rng.choice((1, 2, 3)); n = rng.choice((1, 50, 200) if EMU else (1, 50, 2000)); lo = rng.choice((0, 16, 256, 20000)); hi = lo + rng.choice((1, 100, 512, 9000))
        if EMU and lo == 20000: n = min(n, 3)
        docs = [N.corpus_doc_host(kind, rng.randrange(1 << 30), d, lo, hi, lib=vocab.lib) for d in range(n)]
    else:
        kind = rng.choice(kinds)
        docs = [parity.gen_text(rng, kind, rng.choice([0, 1, 63, 64, 65, 700, 5000, 40000] + ([] if EMU else [200000])), alpha).encode("utf-8") for _ in range(rng.choice([1, 3, 30] + ([] if EMU else [300])))]
    data, offs = parity.pack(docs)
    if len(data) > (400_000 if EMU else 40_000_000):
        continue
    ids, ooff = encs[pattern].encode_batch(data, offs)
    o_ids, o_counts = O.encode_batch(ovocab, pattern, data, offs, threads=32)
    if not (np.array_equal(ids, o_ids) and np.array_equal(np.diff(ooff), o_counts)):
        print("MISMATCH round", rounds, "pattern", pattern, "kind", kind, "docs", len(docs), "bytes", len(data)); sys.exit(1)
    rounds += 1; total += len(data)
print(("emulated fuzz" if EMU else "gpu fuzz") + " ok: %d rounds, %.1f MB, seed %d; promoted pieces in the tables of the four encoders: %s" %
      (rounds, total / 1e6, seed, {p: encs[p].piece_stats()["promoted_pieces_in_tables"] for p in encs}))
