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

# usage: make_traffic_json.py <summary.txt> <docs> <kind> <out.json> [<log that holds the bench line of the profiled command>]
summary, docs, kind, outpath = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
linelog = sys.argv[5] if len(sys.argv) > 5 else None
tag = os.path.basename(os.path.dirname(os.path.abspath(summary)))
text = open(summary).read()
by_kernel = {}


def counter(name, key):
    """per-launch value of counter `key` in the (last) block of kernel `name`, or None"""
    val = None
    for blk in re.finditer(r"%s\s+launches \d+\n((?:\s+\S+\s+per-launch\s+[0-9.]+\n)+)" % re.escape(name), text):
        m = re.search(r"%s\s+per-launch\s+([0-9.]+)" % re.escape(key), blk.group(1))
        if m:
            val = float(m.group(1))
    return val


# (k_merge_long_group: every kernel inside the K_HEAVY bracket of launch_encode -- the class queue's three kernels, the wavefront-a-piece and the giant mergers)
groups = (("k_probe", ("k_probe",)), ("k_merge_short", ("k_merge_short",)), ("k_place", ("k_place",)), ("k_pretok", ("k_pretok_rows",)),
          ("k_merge_long_group", ("k_merge_long", "k_long_count", "k_long_scatter", "k_merge_coop", "k_giant_merge", "k_giant_order")))
for bench_name, names in groups:
    fetch = write = 0.0
    have = False
    issue = {}
    for name in names:
        f, w = counter(name, "FETCH_SIZE"), counter(name, "WRITE_SIZE")
        if f is not None and w is not None:
            fetch += f; write += w; have = True
        for key in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_INSTS_LDS"):
            v = counter(name, key)
            if v is not None:
                issue[key] = issue.get(key, 0.0) + v
    if have:
        by_kernel[bench_name] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
        if issue.get("SQ_INSTS_VALU"):
            # the instruction side of the same passes: wave-instructions per launch, and the lanes a VALU instruction has switched on on average
            by_kernel[bench_name]["issue"] = {"valu_insts": issue["SQ_INSTS_VALU"], "salu_insts": issue.get("SQ_INSTS_SALU"), "lds_insts": issue.get("SQ_INSTS_LDS"),
                                              "lanes_active": round(issue["SQ_THREAD_CYCLES_VALU"] / issue["SQ_INSTS_VALU"], 2) if issue.get("SQ_THREAD_CYCLES_VALU") else None}
out = {"tag": tag, "src_sha": kernel_sources_sha(), "docs_per_gpu": docs, "kind": kind, "by_kernel": by_kernel,
       "note": "2 x FETCH_SIZE (gfx950 correction for wide coalesced reads; uncalibrated for the 16-byte table gathers, "
               "which are mostly Infinity-Cache hits that the fabric counters still count) + WRITE_SIZE"}
if linelog and os.path.exists(linelog):
    # the profiled command's own bench line: which text it was (kind 6: the sha256 of the real text), under which table and pattern
    for ln in open(linelog, errors="replace"):
        if ln.startswith("{") and '"metric"' in ln:
            try:
                d = json.loads(ln)
                rt = d.get("config", {}).get("real_text") or {}
                out["docs_per_gpu"] = d.get("config", {}).get("docs_per_gpu", out["docs_per_gpu"])
                if rt.get("sha256"):
                    out["corpus_sha256"] = rt["sha256"]
                out["vocab_pattern"] = d["config"].get("vocab_pattern_key")
                out["bench_value"] = d.get("value"); out["bench_ms_per_step"] = d.get("ms_per_step")
            except Exception:
                pass
json.dump(out, open(outpath, "w"), indent=1)
print(out)
