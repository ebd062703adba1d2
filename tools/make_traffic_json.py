#!/usr/bin/env python3
"""<summary.txt of tools/gpu_profile.sh> -> traffic.json for bench.py's roofline.traffic.
HBM bytes per launch of every encode kernel, as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 / rocprofv3: FETCH_SIZE and
WRITE_SIZE are in KiB, collected in separate --pmc passes; FETCH_SIZE reads half the bytes of a wide coalesced stream, so it is
doubled.  The file records the identity of the kernel sources it was collected on (bench.kernel_sources_sha): bench.py reports the
figure only for exactly those sources."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha        # identity of the kernel sources the counters were collected on

# usage: make_traffic_json.py <summary.txt> <docs> <kind> <out.json>
summary, docs, kind, outpath = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[-1]
tag = os.path.basename(os.path.dirname(os.path.abspath(summary)))
text = open(summary).read()
by_kernel = {}
for name, bench_name in (("k_probe", "k_probe"), ("k_merge_short", "k_merge_short"), ("k_place", "k_place"), ("k_pretok_rows", "k_pretok"), ("k_merge_long", "k_merge_long_group")):
    fetch = write = None
    for blk in re.finditer(r"%s\s+launches \d+\n((?:\s+\S+\s+per-launch\s+[0-9.]+\n)+)" % re.escape(name), text):
        m = re.search(r"WRITE_SIZE\s+per-launch\s+([0-9.]+)", blk.group(1))
        if m:
            write = float(m.group(1))
        m = re.search(r"FETCH_SIZE\s+per-launch\s+([0-9.]+)", blk.group(1))
        if m:
            fetch = float(m.group(1))
    if fetch is not None and write is not None:
        by_kernel[bench_name] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
out = {"tag": tag, "src_sha": kernel_sources_sha(), "docs_per_gpu": docs, "kind": kind, "by_kernel": by_kernel,
       "note": "2 x FETCH_SIZE (gfx950 correction for wide coalesced reads; uncalibrated for the 16-byte table gathers, "
               "which are mostly Infinity-Cache hits that the fabric counters still count) + WRITE_SIZE"}
json.dump(out, open(outpath, "w"), indent=1)
print(out)
