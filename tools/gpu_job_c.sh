#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/devprof; mkdir -p $O
TKZ_LIBTKZ=$REPO/tokenizer_amd/lib_dev/libtkz.so TKZ_DEV_ABLATE=16 timeout 300 python bench.py --kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000 --no-cpu-baseline --steps 1 --warmup 1 > $O/out.json 2> $O/err.log; grep devprof $O/err.log | tail -2
