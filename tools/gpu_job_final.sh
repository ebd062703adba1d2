#!/bin/bash
# round-2 measurement job: parity suite, smoke, bench line, other shapes, soak, rocprof passes (bench workload and mixed text)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-r02}; O=gpurun_out/$TAG; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
# the rocprof passes FIRST: the bench line below then reports the PMC traffic of these very sources (bench.py compares src_sha)
bash tools/gpu_profile.sh $TAG 10000000 > $O/profile.log 2>&1; echo "profile rc=$?"
bash tools/gpu_profile.sh ${TAG}_mixed 2000000 "--kind 2" > $O/profile_mixed.log 2>&1; echo "profile mixed rc=$?"
cp gpurun_out/prof_$TAG/traffic.json profiles/traffic_latest.json 2>/dev/null
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-300 $O/bench_n1.json
rm -f $O/bench_shapes.jsonl
for spec in "--kind 4 --docs 4000000" "--kind 2 --docs 2000000" "--kind 3 --pattern 3 --docs 32768 --min-len 30000 --max-len 34000" "--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000" "--kind 2 --pattern 3 --docs 2000000" "--kind 1 --pattern 3" "--kind 4 --pattern 3 --docs 4000000" "--kind 1 --pattern 1" "--kind 1 --vocab gpt2"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
timeout 200 python tools/gpu_fuzz.py 60 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
timeout 200 python tools/o200k_scan_fuzz.py --gpu --seeds 20 > $O/o200k_fuzz.log 2>&1; echo "o200k fuzz rc=$?"; tail -7 $O/o200k_fuzz.log
