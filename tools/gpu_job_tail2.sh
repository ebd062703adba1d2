#!/bin/bash
# development: the giant-piece counters (lib_prof, TKZ_DEV_ABLATE=16) on the long-run shape, both patterns
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-tail2}; O=gpurun_out/$TAG; mkdir -p $O
LONGRUN="--kind 3 --docs 32768 --min-len 30000 --max-len 34000 --heldout-steps 0"
for pat in ${PATS:-2 4}; do
  TKZ_DEV_ABLATE=16 TKZ_LIBTKZ=$REPO/tokenizer_amd/lib_prof/libtkz.so timeout 300 python bench.py $LONGRUN --pattern $pat --no-cpu-baseline --steps 1 --warmup 0 --pipelined-steps 0 --no-memo-steps 0 --no-piece-stats > $O/b_prof_$pat.json 2> $O/devprof_$pat.txt
  echo "pattern $pat:"; grep "devprof" $O/devprof_$pat.txt | tail -4 | cut -c1-420
done
