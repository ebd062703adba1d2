#!/bin/bash
# development: rocprofv3 kernel trace of one bench shape.  usage: tools/gpu_job_trace.sh <tag> "<bench args>"
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-trace}; O=$REPO/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $REPO/bench.py $2 --no-cpu-baseline --steps 3 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --no-piece-stats > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -16 "$f" | cut -c1-150
rm -rf $O/trace
