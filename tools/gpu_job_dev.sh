#!/bin/bash
# development job: a parity subset on the GPU, the bench line, a few shapes
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-dev}; O=gpurun_out/$TAG; mkdir -p $O
KEXPR=${2:-"pieces or giant or arena or batch_vs_oracle or dense or memo or corpus_properties or adversarial or vocab_key or golden or errors or host_path or eight_shards"}
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "$KEXPR" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
timeout 900 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout 900 python bench.py --no-cpu-baseline --no-memo > $O/bench_nomemo.json 2>> $O/bench_n1.err; echo "bench no-memo rc=$?"
rm -f $O/bench_shapes.jsonl
for spec in "--kind 2 --docs 2000000" "--kind 4 --docs 4000000" "--kind 1 --vocab gpt2 --docs 4000000"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
python - $TAG <<'P'
import json,sys
for f in ("bench_n1.json", "bench_nomemo.json", "bench_shapes.jsonl"):
  for l in open("gpurun_out/%s/%s" % (sys.argv[1], f)):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["pattern"][:6], d["config"]["workload"][:34], d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"])
P
