#!/bin/bash
# Kernel-trace summary + PMC passes (separate runs, never combined with tracing domains) for bench.py.
# usage: tools/gpu_profile.sh <tag> [docs] [extra bench.py arguments, e.g. "--kind 2"]
set -u
TAG=${1:-r01}; DOCS=${2:-2000000}; EXTRA=${3:-}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"; mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
# (no sizing attempt under the profiler: a fresh workspace's first large batch otherwise launches k_probe once more on a sixteenth of the sub-tiles, and the
#  per-launch average of that kernel in the stats would be over five full launches and one short one)
export TKZ_SIZING_MIN_SUB=4000000000
# (--heldout-steps 0 --pipelined-steps 0 --no-piece-stats: nothing but the headline steps, their warm-up and the sizing pass run under the profiler -- per-launch averages of a kernel are over THOSE launches)
BENCH="python $REPO/bench.py --docs $DOCS --steps 3 --warmup 1 --no-cpu-baseline --heldout-steps 0 --pipelined-steps 0 --no-piece-stats --no-first-call $EXTRA"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG/trace -o trace -- $BENCH > $REPO/gpurun_out/prof_$TAG/trace.log 2>&1; echo "trace rc=$?"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
         "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/prof_$TAG/pmc$i -o pmc -- $BENCH > $REPO/gpurun_out/prof_$TAG/pmc$i.log 2>&1; echo "pmc$i rc=$?"
done
cd $REPO
python tools/summarize_prof.py gpurun_out/prof_$TAG > gpurun_out/prof_$TAG/summary.txt 2>&1
KIND=1; case "$EXTRA" in *"--kind 2"*) KIND=2;; *"--kind 3"*) KIND=3;; *"--kind 4"*) KIND=4;; *"--kind 6"*) KIND=6;; esac
python tools/make_traffic_json.py gpurun_out/prof_$TAG/summary.txt $DOCS $KIND gpurun_out/prof_$TAG/traffic.json gpurun_out/prof_$TAG/trace.log > /dev/null 2>&1
cat gpurun_out/prof_$TAG/summary.txt
# keep only the summaries small enough to merge back
find gpurun_out/prof_$TAG -name "*.db" -delete
