#!/usr/bin/env python3
"""Single-call latency of the drop-in path (BASELINE.json configs[0]'s shape): ITokenizer.Encode(text) on one short prompt and
EncodeBatch on 1,000 prompts of 16..128 bytes, through the C ABI on host buffers (tkz_encode_utf8 / tkz_encode_batch_utf8:
upload, kernels, download, one call at a time), beside the CPU restatement of the reference (oracle/, one thread) on the same calls.
Prints one JSON line.  usage: python tools/latency_probe.py [--reps 200]"""
import argparse
import gzip
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--vocab", default="gpt2")
    ap.add_argument("--pattern", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    from tokenizer_amd import _native as N
    from oracle import oracle as O
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", args.vocab + ".tiktoken.gz"), "rb").read())
    enc = N.Encoder(N.Vocab(raw), args.pattern, device=0)
    ov = O.Vocab(raw)
    oenc = O.Encoder(ov, args.pattern)
    prompts = [N.corpus_doc_host(1, 0x5EED0001, d, 16, 128) for d in range(1000)]
    one = b"Hello World, this is a short prompt of sixty-four bytes, more or"[:64]
    data = np.frombuffer(b"".join(prompts), np.uint8)
    offs = np.cumsum([0] + [len(p) for p in prompts]).astype(np.int64)

    def med_us(fn, reps):
        fn(); fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
        return round(statistics.median(ts), 1), round(min(ts), 1)
    assert enc.encode_utf8(one) == oenc.encode_bytes(one)
    ids, ooff = enc.encode_batch(data, offs)
    exp = []
    for p in prompts:
        exp += oenc.encode_bytes(p)
    assert ids.tolist() == exp
    out = {"what": "single-call latency through the C ABI on host buffers, microseconds per call (median, min)",
           "vocab": args.vocab, "pattern": args.pattern, "reps": args.reps,
           "encode_one_64B_prompt_us": med_us(lambda: enc.encode_utf8(one), args.reps),
           "encode_batch_1000_prompts_us": med_us(lambda: enc.encode_batch(data, offs), max(20, args.reps // 4)),
           "batch_bytes": int(len(data)),
           "oracle_1_thread_one_prompt_us": med_us(lambda: oenc.encode_bytes(one), args.reps),
           "oracle_1_thread_1000_prompts_us": med_us(lambda: [oenc.encode_bytes(p) for p in prompts], 10)}
    enc.encode_utf8(one)
    ph = enc.small_path_phases()
    out["single_launch_path"] = {"calls_handed_back": enc.small_path_calls(),
                                 "phase_cycles_one_prompt": [ph[i + 1] - ph[i] for i in range(len(ph) - 1) if ph[i + 1] and ph[i]],
                                 "phases": "input+zero, docmark, bitmap copy, pre-tokenizer, counts+scans, probe, merge_short, merge_long, scan, place, docoffs"}
    enc.encode_batch(data, offs)
    ph = enc.small_path_phases()
    out["single_launch_path"]["phase_cycles_1000_prompts"] = [ph[i + 1] - ph[i] for i in range(len(ph) - 1) if ph[i + 1] and ph[i]]
    # mid-size host batches (between the single-launch path's 128 KiB and the chunk-pipelined path's 96 MB): documents of ~512 bytes, MB/s of one
    # tkz_encode_batch_utf8 call on pageable and on page-locked buffers
    try:
        import torch
        mid = {}
        for mb in (0.25, 1, 4, 16, 64):
            nd = max(1, int(mb * (1 << 20) / 512))
            docs = [N.corpus_doc_host(1, 0x5EED0002, d, 256, 768) for d in range(min(nd, 4096))]
            reps_d = (nd + len(docs) - 1) // len(docs)
            docs = (docs * reps_d)[:nd]
            bdata = np.frombuffer(b"".join(docs), np.uint8)
            boffs = np.cumsum([0] + [len(x) for x in docs]).astype(np.int64)
            oi, oo = np.zeros(len(bdata), np.int32), np.zeros(nd + 1, np.int64)
            tb = torch.empty(len(bdata), dtype=torch.uint8).pin_memory(); tb.numpy()[:] = bdata
            to = torch.empty(nd + 1, dtype=torch.int64).pin_memory(); to.numpy()[:] = boffs
            ti = torch.zeros(len(bdata), dtype=torch.int32).pin_memory(); too = torch.zeros(nd + 1, dtype=torch.int64).pin_memory()
            r = max(5, min(50, int(200 / max(mb, 0.25))))
            us_pageable = med_us(lambda: enc.encode_batch(bdata, boffs, out=(oi, oo)), r)
            us_pinned = med_us(lambda: enc.encode_batch(tb.numpy(), to.numpy(), out=(ti.numpy(), too.numpy())), r)
            mid["%g MB" % mb] = {"bytes": int(len(bdata)), "pageable_us": us_pageable[0], "pageable_MBps": round(len(bdata) / us_pageable[0], 1),
                                 "pinned_us": us_pinned[0], "pinned_MBps": round(len(bdata) / us_pinned[0], 1)}
        out["mid_size_host_batches"] = mid
    except Exception as ex:
        out["mid_size_host_batches"] = "%s: %s" % (type(ex).__name__, ex)
    # the floor: one trivial kernel launch + stream synchronisation through torch, for comparison
    try:
        import torch
        x = torch.zeros(64, device="cuda")
        st = torch.cuda.Stream()
        def floor():
            with torch.cuda.stream(st):
                x.add_(1)
            st.synchronize()
        out["launch_plus_sync_floor_us"] = med_us(floor, args.reps)
    except Exception as ex:
        out["launch_plus_sync_floor_us"] = str(ex)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
