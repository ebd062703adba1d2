#!/bin/bash
# the short form of gpu_job_final.sh (a few GPU-minutes): a parity subset, smoke, the rocprof passes of the bench workload, then the bench
# line (which reports the PMC traffic of these very sources) and the main shapes
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-r02}; O=gpurun_out/$TAG; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
( time timeout 600 python -m pytest tests -m gpu -x -q -k "memo or dense or batch_vs_oracle or corpus_properties or golden or host_path" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/gpu_profile.sh $TAG 10000000 > $O/profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/prof_$TAG/traffic.json profiles/traffic_latest.json 2>/dev/null
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-300 $O/bench_n1.json
rm -f $O/bench_shapes.jsonl
for spec in "--kind 4 --docs 4000000" "--kind 2 --docs 2000000" "--kind 3 --pattern 3 --docs 32768 --min-len 30000 --max-len 34000" "--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000" "--kind 2 --pattern 3 --docs 2000000" "--kind 1 --no-memo"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
