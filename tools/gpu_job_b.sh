#!/bin/bash
# development: time kernel-shape variants (libs built with make OUT=../lib_x EXTRA=...) on the bench workload
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/variants; mkdir -p $O; rm -f $O/*.jsonl
for v in lib lib_va lib_vb; do
  for spec in "--docs 4000000" "--docs 2000000 --kind 2" "--docs 4000000 --kind 4"; do
    TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 300 python bench.py $spec --no-cpu-baseline --steps 4 --warmup 1 2>>$O/err.log | grep "^{" | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); print('$v', '$spec', j['value'], j['ms_per_step'], j['roofline']['kernels_ms'], j['parity'][:20])"
  done
done

