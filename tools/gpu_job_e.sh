#!/bin/bash
# development job: the o200k multi-byte block scanner -- parity on the GPU, soak, and the o200k shapes
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-e}; O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q -k "pretok or o200k or splits or corpus_properties" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
timeout 300 python tools/o200k_scan_fuzz.py --gpu --seeds 25 > $O/o200k_fuzz.log 2>&1; echo "o200k fuzz rc=$?"; tail -8 $O/o200k_fuzz.log
rm -f $O/bench_shapes.jsonl
for spec in "--kind 2 --pattern 3 --docs 2000000" "--kind 3 --pattern 3 --docs 32768 --min-len 30000 --max-len 34000" "--kind 1 --pattern 3 --docs 4000000"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
