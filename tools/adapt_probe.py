#!/usr/bin/env python3
"""TKZ_OPT_ADAPT under a text that changes: per-step rate and the encoder's bookkeeping (tkz_encoder_adapt_stats) while ONE encoder goes from the
synthetic corpus to the box's real text and back.  usage: adapt_probe.py [vocab=gpt2] [pattern=1] [adapt=1] [fast=0]   -> JSON lines"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from tokenizer_amd import _native as N
vname = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
pattern = int(sys.argv[2]) if len(sys.argv) > 2 else 1
adapt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
n_docs = 2_000_000
d_offs = torch.empty(n_docs + 1, dtype=torch.int64, device=dev)
total = N.corpus_generate_device(0, 1, 0x5EED0002, 0, n_docs, 256, 768, d_offs.data_ptr(), None, 0, st)
d_bytes = torch.empty(total + 64, dtype=torch.uint8, device=dev)
N.corpus_generate_device(0, 1, 0x5EED0002, 0, n_docs, 256, 768, d_offs.data_ptr(), d_bytes.data_ptr(), total, st)
r_bytes, r_offs, meta = bench.real_text_corpus(256 << 20, 256, 768)
r_nd, r_total = len(r_offs) - 1, int(r_offs[-1])
rd_bytes = torch.zeros(r_total + 64, dtype=torch.uint8, device=dev); rd_bytes[:r_total] = torch.from_numpy(r_bytes).to(dev)
rd_offs = torch.from_numpy(r_offs).to(dev)
d_ids = torch.empty(max(total, r_total), dtype=torch.int32, device=dev)
d_oo = torch.empty(max(n_docs, r_nd) + 1, dtype=torch.int64, device=dev)
raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", vname + ".tiktoken.gz"), "rb").read())
enc = N.Encoder(N.Vocab(raw), pattern)
enc.set_option(N.OPT_ADAPT, adapt)


def run(which):
    b, o, nd, tot = (d_bytes, d_offs, n_docs, total) if which == "syn" else (rd_bytes, rd_offs, r_nd, r_total)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.encode_batch_device(b.data_ptr(), o.data_ptr(), nd, tot, d_ids.data_ptr(), tot, d_oo.data_ptr(), st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = enc.adapt_stats()
    print(json.dumps({"text": which, "GBps": round(tot / dt / 1e9, 1), "ms": round(dt * 1e3, 2), **{k: s[k] for k in ("promotions", "relearns", "promoted_pieces", "settled_miss_share", "recent_miss_share")}}), flush=True)


fast = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # 1: the three synthetic batches back to back, as bench.py's drift leg feeds them (15 ms: the first promotion is still being built when the text changes)
if fast:
    for _ in range(3):
        enc.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, total, d_ids.data_ptr(), total, d_oo.data_ptr(), st)
for which, n in ((("real", 24), ("syn", 6)) if fast else (("syn", 3), ("real", 16), ("syn", 6))):
    for _ in range(n):
        run(which)
# ... and an encoder that only ever sees the real text, step for step
enc = N.Encoder(N.Vocab(raw), pattern)
enc.set_option(N.OPT_ADAPT, adapt)
print(json.dumps({"text": "a fresh encoder on the real text"}), flush=True)
for _ in range(16):
    run("real")
