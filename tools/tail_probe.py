"""Development probe (CPU, emulated kernels): how many batches and merges the giant-piece tail (tkz_bpe_long_tail) takes on ONE long diverse piece --
a camelCase chain of the corpus lexicon, one \\p{L}+ piece under cl100k -- in a build with the development counters (-DTKZ_DEVPROF, TKZ_DEV_ABLATE=16:
the library prints them on stderr), checked against the oracle.  usage: tail_probe.py <libtkz_hostemu*.so> [bytes] [vocab]"""
import gzip, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TKZ_DEV_ABLATE"] = "16"
import numpy as np
from tokenizer_amd import _native as N
from oracle import oracle as O

lib = N.Library(os.path.abspath(sys.argv[1]))
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
vname = sys.argv[3] if len(sys.argv) > 3 else "synth100k"
raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", vname + ".tiktoken.gz"), "rb").read())
words = []
d = 0
while sum(len(w) for w in words) < nbytes:
    words += [w.decode().capitalize() for w in re.findall(rb"[A-Za-z]+", N.corpus_doc_host(1, 12345, d, 4000, 4000, lib=lib))]
    d += 1
text = "".join(words)[:nbytes].encode()
data = np.frombuffer(text, np.uint8); offs = np.array([0, len(text)], np.int64)
v = N.Vocab(raw, lib)
print("vocab", vname, "max_key_len", v.max_key_len if hasattr(v, "max_key_len") else "?", "piece bytes", len(text), flush=True)
enc = N.Encoder(v, N.CL100K)
ids, ooff = enc.encode_batch(data, offs)
o_ids, _ = O.encode_batch(O.Vocab(raw), N.CL100K, data, offs, threads=1)
print("tokens", len(ids), "bit-exact" if np.array_equal(ids, o_ids) else "MISMATCH")
