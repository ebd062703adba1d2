#!/bin/bash
# development job: ranges / overlap of k_probe and k_merge_short -- parity subset, then the bench line with 1, 4, 8, 16 ranges
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-f}; O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "o200k or dense or batch_vs_oracle or corpus_properties or host_path or eight_shards" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
rm -f $O/bench_ranges.jsonl
for r in 0 1 4 16; do
  TKZ_ENCODE_RANGES=$r timeout 600 python bench.py --no-cpu-baseline >> $O/bench_ranges.jsonl 2>> $O/bench_ranges.err; echo "ranges $r rc=$?"
done
for spec in "--kind 2 --pattern 3 --docs 2000000" "--kind 2 --docs 2000000" "--kind 4 --docs 4000000" "--kind 1 --pattern 3"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
python - $TAG <<'P'
import json,sys
for f in ("bench_ranges.jsonl","bench_shapes.jsonl"):
    for l in open("gpurun_out/%s/%s" % (sys.argv[1] if len(sys.argv)>1 else "f", f)):
        if l.startswith("{"):
            d=json.loads(l); print(f[:12], d["config"]["pattern"][:6], d["config"]["workload"][:30], d["value"], d["ms_per_step"], d["roofline"].get("launches_per_step"), d["roofline"]["kernels_ms"])
P
