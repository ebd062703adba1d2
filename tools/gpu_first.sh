#!/bin/bash
# First GPU contact: smoke, the gpu-marked parity tests, a short bench, a kernel-trace profile.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -15 gpurun_out/pytest_gpu.log
echo "== bench 1M" ; timeout 600 python bench.py --docs 1000000 --steps 3 --warmup 1 > gpurun_out/bench_1m.log 2>&1 ; echo "bench rc=$?" ; tail -3 gpurun_out/bench_1m.log
echo "== bench 10M" ; timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_10m.log 2>&1 ; echo "bench rc=$?" ; tail -3 gpurun_out/bench_10m.log
echo "== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -- python "$OLDPWD/bench.py" --docs 2000000 --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1 ; echo "rocprof rc=$?")
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 "$f"; done
