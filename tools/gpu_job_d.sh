#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/r02g; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q -s ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|GB/s" $O/pytest_gpu.log | tail -4
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-600 $O/bench_n1.json
python -c "
import json
j=json.loads([l for l in open('$O/bench_n1.json') if l.startswith('{')][-1]); print(j['pcie_inclusive']); print(j['cpu_baseline'])"
