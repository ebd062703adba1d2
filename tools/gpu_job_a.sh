#!/bin/bash
# first GPU job of round 2: parity suite, bench line, other shapes, stage breakdown (dev build), rocprof passes
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02f; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-1500 $O/bench_n1.json
for spec in "--kind 4 --docs 4000000" "--kind 2 --docs 2000000" "--kind 3 --pattern 3 --docs 32768 --min-len 30000 --max-len 34000" "--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000" "--kind 2 --pattern 3 --docs 2000000" "--kind 1 --vocab gpt2"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
python - <<'PY'
import json
for l in open('gpurun_out/r02f/bench_shapes.jsonl'):
    j=json.loads(l); print(j['config']['workload'][:60], j['config']['pattern'], j['value'], j['ms_per_step'], j['roofline']['kernels_ms'])
PY
bash tools/gpu_profile.sh r02f 10000000 > $O/profile.log 2>&1; tail -60 $O/profile.log | head -80
