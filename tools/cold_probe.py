#!/usr/bin/env python3
"""First batch on a fresh encoder (empty piece memo, workspace not yet sized) against later ones, on the bench corpus: what a job that encodes
ONE batch pays.  usage: cold_probe.py [vocab=synth100k_heldout] [docs=10000000] [kind=1] [pattern=2] [reserve=1]   -> one JSON line
(reserve = 1, the default since round 6: tkz_encoder_reserve(batch size) right after the encoder is created -- its allocations are timed apart, `reserve_ms`)"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tokenizer_amd import _native as N
vname = sys.argv[1] if len(sys.argv) > 1 else "synth100k_heldout"
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pattern = int(sys.argv[4]) if len(sys.argv) > 4 else 2
reserve = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
seed = 0x5EED0000 + {1: 2, 2: 3, 3: 5}[kind]
d_offs = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
total = N.corpus_generate_device(0, kind, seed, 0, n_docs, 256, 768, d_offs.data_ptr(), None, 0, st)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
N.corpus_generate_device(0, kind, seed, 0, n_docs, 256, 768, d_offs.data_ptr(), d_bytes.data_ptr(), total, st)
d_ids = torch.empty(total, dtype=torch.int32, device=dev)
d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", vname + ".tiktoken.gz"), "rb").read())
out = {"vocab": vname, "docs": n_docs, "bytes": total, "kind": kind, "pattern": pattern, "reserve": reserve}
for memo in (1, 0):
    enc = N.Encoder(N.Vocab(raw), pattern)
    if not memo:
        enc.set_option(N.OPT_PIECE_MEMO, 0)
    if reserve:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc.reserve(total, n_docs)
        torch.cuda.synchronize(); out["reserve_ms_memo_%s" % ("on" if memo else "off")] = round((time.perf_counter() - t0) * 1e3, 1)
    enc.set_profiling(True)
    ms = []
    for it in range(6):        # (the promotion learnt from call 1 is built on a thread behind it and lands a call or two later)
        enc.kernel_ms(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ntok = enc.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total, d_oo.data_ptr(), st)
        torch.cuda.synchronize(); ms.append(round((time.perf_counter() - t0) * 1e3, 2))
        if it in (0, 5):
            out["%s_call_kernels_ms_memo_%d" % ("first" if it == 0 else "sixth", memo)] = {k: round(v[0], 2) for k, v in enc.kernel_ms().items()}
    out["call_ms_memo_%s" % ("on" if memo else "off")] = ms
    out["tokens"] = ntok
    del enc
print(json.dumps(out))
