#!/usr/bin/env python3
"""profiles/<tag>/summary.txt (tools/gpu_profile.sh) -> profiles/traffic_latest.json for bench.py's roofline.traffic.
HBM bytes per launch of the dominant kernel, as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 / rocprofv3:
FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate --pmc passes; FETCH_SIZE reads half the bytes of a wide
coalesced stream, so it is doubled."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha        # identity of the kernel sources the counters were collected on

# usage: make_traffic_json.py <summary.txt> <docs> <kind> <kernel> <out.json>
summary, docs, kind, kernel, outpath = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
tag = os.path.basename(os.path.dirname(os.path.abspath(summary)))
text = open(summary).read()


def counter(name):
    m = re.search(r"%s\s+launches \d+\n\s+%s\s+per-launch\s+([0-9.]+)" % (re.escape(kernel), name), text)
    return float(m.group(1)) if m else None


fetch, write = counter("FETCH_SIZE"), None
for blk in re.finditer(r"%s\s+launches \d+\n((?:\s+\S+\s+per-launch\s+[0-9.]+\n)+)" % re.escape(kernel), text):
    m = re.search(r"WRITE_SIZE\s+per-launch\s+([0-9.]+)", blk.group(1))
    if m:
        write = float(m.group(1))
    m = re.search(r"FETCH_SIZE\s+per-launch\s+([0-9.]+)", blk.group(1))
    if m:
        fetch = float(m.group(1))
out = {"tag": tag, "src_sha": kernel_sources_sha(), "docs_per_gpu": docs, "kind": kind, "kernel": kernel, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
       "note": "2 x FETCH_SIZE (gfx950 correction for wide coalesced reads; uncalibrated for the 16-byte table gathers, "
               "which are mostly Infinity-Cache hits that the fabric counters still count) + WRITE_SIZE"}
json.dump(out, open(outpath, "w"), indent=1)
print(out)
