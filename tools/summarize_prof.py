#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output directory: per-kernel time (kernel-trace stats) and per-kernel
PMC counter sums / per-launch averages.  Writes plain text to stdout."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_merge_short", "k_merge_long", "k_long_count", "k_long_scatter", "k_merge_coop", "k_list_stats", "k_giant_order", "k_giant_merge", "k_giant_find", "k_probe", "k_place", "k_counts3", "k_pretok_rows", "k_pretok_seq", "k_docmark", "k_docoffs", "k_scan_partials", "k_scan_top",
              "k_scan_final", "k_corpus_fill", "k_corpus_lengths", "k_offsets_scan", "k_doccount"):
        if k in name:
            return k
    return name[:60]


def main(d):
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    for f in stats:
        print("== kernel-trace stats:", os.path.relpath(f, d))
        for row in csv.DictReader(open(f)):
            print("  %-18s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"),
                                                                            row.get("AverageNs"), row.get("Percentage")))
    for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
        if not os.path.isdir(p):
            continue
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(set)
            for row in csv.DictReader(open(f)):
                k = short(row.get("Kernel_Name", ""))
                acc[k][row.get("Counter_Name")] += float(row.get("Counter_Value") or 0)
                calls[k].add(row.get("Dispatch_Id"))
            print("== PMC", os.path.basename(p))
            for k in acc:
                n = max(1, len(calls[k]))
                if k.startswith("k_corpus") or k.startswith("k_offsets"):
                    continue
                print("  %-18s launches %d" % (k, n))
                for c, v in sorted(acc[k].items()):
                    print("      %-26s per-launch %16.1f" % (c, v / n))


if __name__ == "__main__":
    main(sys.argv[1])
