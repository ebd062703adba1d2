#!/bin/bash
# development: the long-run shapes (both patterns, every document against the oracle), the mixed shape, a short headline run, a kernel trace of the long-run shape
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-coop}; O=gpurun_out/$TAG; mkdir -p $O
LONGRUN="--kind 3 --docs 32768 --min-len 30000 --max-len 34000 --heldout-steps 0"
( time timeout 600 python -m pytest tests -m gpu -x -q -k "giant or adversarial or long_diverse or long_runs or small" ) > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_subset.log | tail -3
rm -f $O/bench_shapes.jsonl
for spec in "$LONGRUN --pattern 2" "$LONGRUN --pattern 4" "--kind 2 --docs 2000000 --heldout-steps 0"; do
  timeout 600 python bench.py $spec --parity-only --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --heldout-steps 0 --pipelined-steps 0 --no-memo-steps 0 --no-piece-stats >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "headline rc=$?"
python - $O/bench_shapes.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["pattern"][:12], d["config"]["workload"][:30], d["value"], d.get("value_no_memo"), d.get("value_two_in_flight"), d["ms_per_step"], d["parity"][:28], d["roofline"]["kernels_ms"])
P
bash tools/gpu_job_trace.sh ${TAG}_trace "$LONGRUN --pattern 2"
