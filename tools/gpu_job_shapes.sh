#!/bin/bash
# the bench line and the other shapes only (no profiles): gpurun_out/<tag>/bench_n1.json, bench_shapes.jsonl
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-shapes}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-200 $O/bench_n1.json
rm -f $O/bench_shapes.jsonl
for spec in "--kind 4 --docs 4000000" "--kind 2 --docs 2000000" "--kind 3 --pattern 3 --docs 32768 --min-len 30000 --max-len 34000" "--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000" "--kind 2 --pattern 3 --docs 2000000" "--kind 1 --pattern 3" "--kind 4 --pattern 3 --docs 4000000" "--kind 1 --pattern 1" "--kind 1 --vocab gpt2" "--kind 1 --no-memo"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
