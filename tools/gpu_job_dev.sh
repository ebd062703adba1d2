#!/bin/bash
# development job: parity subset on the GPU, the o200k scanner soak, the bench line, a few shapes, a kernel trace of one shape
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-e}; O=gpurun_out/$TAG; mkdir -p $O
KEXPR=${2:-"pretok or o200k or splits or dense or batch_vs_oracle or corpus_properties or pieces"}
TRACE=${3:-"--kind 2 --pattern 3 --docs 2000000"}
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "$KEXPR" ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_gpu.log | tail -3
timeout 300 python tools/o200k_scan_fuzz.py --gpu --seeds 10 > $O/o200k_fuzz.log 2>&1; echo "o200k fuzz rc=$?"; tail -3 $O/o200k_fuzz.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-200 $O/bench_n1.json
rm -f $O/bench_shapes.jsonl
for spec in "--kind 2 --pattern 3 --docs 2000000" "--kind 2 --docs 2000000" "--kind 4 --docs 4000000"; do
  timeout 600 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
done
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace -o trace -- python $REPO/bench.py $TRACE --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$O/trace.log 2>&1; echo "trace rc=$?"
cd $REPO; find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -d, -f1-4,7' ; find $O -name "*.db" -delete
