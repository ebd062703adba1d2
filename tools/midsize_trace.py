#!/usr/bin/env python3
"""The command timeline of ONE mid-size host call (tkz_encode_batch_utf8 on page-locked buffers), for rocprofv3 --kernel-trace --memory-copy-trace:
   run:        rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python tools/midsize_trace.py run [mb=1] [calls=12]
   summarise:  python tools/midsize_trace.py show <dir>      -> the last call's commands (start relative to its first, duration, gap to the previous), microseconds"""
import csv, glob, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mb, calls):
    import gzip
    import numpy as np, torch
    from tokenizer_amd import _native as N
    raw = gzip.decompress(open(os.path.join(ROOT, "tests", "golden", "gpt2.tiktoken.gz"), "rb").read())
    enc = N.Encoder(N.Vocab(raw), 1)
    nd = max(1, int(mb * (1 << 20) / 512))
    docs = [N.corpus_doc_host(1, 0x5EED0002, d, 256, 768) for d in range(min(nd, 4096))]
    docs = (docs * ((nd + len(docs) - 1) // len(docs)))[:nd]
    bdata = np.frombuffer(b"".join(docs), np.uint8)
    boffs = np.cumsum([0] + [len(x) for x in docs]).astype(np.int64)
    tb = torch.empty(len(bdata), dtype=torch.uint8).pin_memory(); tb.numpy()[:] = bdata
    to = torch.empty(nd + 1, dtype=torch.int64).pin_memory(); to.numpy()[:] = boffs
    ti = torch.zeros(len(bdata), dtype=torch.int32).pin_memory(); too = torch.zeros(nd + 1, dtype=torch.int64).pin_memory()
    us = []
    for _ in range(calls):
        t0 = time.perf_counter()
        enc.encode_batch(tb.numpy(), to.numpy(), out=(ti.numpy(), too.numpy()))
        us.append(round((time.perf_counter() - t0) * 1e6, 1))
        time.sleep(0.002)          # (a gap in the trace between calls)
    print(json.dumps({"mb": mb, "bytes": int(len(bdata)), "call_us": us}))


def show(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?"))))
    rows.sort()
    if not rows:
        print("no trace rows under", d); return
    # calls are separated by >= 1 ms of nothing
    calls, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur) > 1_000_000:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    last = calls[-1]
    t0 = last[0][0]
    print("%d calls in the trace; the last one: %d commands, %.1f us from the first start to the last end, %.1f us inside commands" %
          (len(calls), len(last), (max(x[1] for x in last) - t0) / 1e3, sum(x[1] - x[0] for x in last) / 1e3))
    prev_end = t0
    for s, e, n in last:
        print("  %8.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    else:
        show(sys.argv[2])
