#!/bin/bash
# development job: o200k scanners on the GPU (parity subset, soak), o200k shapes with block leftovers, kernel trace of the mixed shape
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-g}; O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "pretok or o200k or splits or errors or corpus_properties" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
timeout 300 python tools/o200k_scan_fuzz.py --gpu --seeds 20 > $O/o200k_fuzz.log 2>&1; echo "o200k fuzz rc=$?"; tail -7 $O/o200k_fuzz.log
rm -f $O/bench_shapes.jsonl
for spec in "--kind 2 --pattern 3 --docs 2000000" "--kind 3 --pattern 3 --docs 32768 --min-len 30000 --max-len 34000" "--kind 1 --pattern 3 --docs 4000000" "--kind 4 --pattern 3 --docs 2000000"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
python - $TAG <<'P'
import json,sys
for l in open("gpurun_out/%s/bench_shapes.jsonl" % sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["pattern"][:6], d["config"]["workload"][:34], d["value"], d["ms_per_step"], d["roofline"].get("o200k_blocks"), d["roofline"]["kernels_ms"])
P
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -o trace -- python $REPO/bench.py --kind 2 --pattern 3 --docs 2000000 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$O/trace.log 2>&1; echo "trace rc=$?"
cd $REPO; find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -d, -f1-4' | cut -c1-150; find $O -name "*.db" -delete
