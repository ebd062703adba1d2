#!/bin/bash
# development: the bench line with variant builds of libtkz (tokenizer_amd/lib_<name>/libtkz.so, made with `make OUT=../lib_<name> EXTRA=...`)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-var}; O=gpurun_out/$TAG; mkdir -p $O; rm -f $O/variants.txt
for v in lib lib_u3o8 lib_u4o7 lib_u4o8; do
  [ -f tokenizer_amd/$v/libtkz.so ] || continue
  TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/b_$v.json 2>> $O/err.txt
  python - $O/b_$v.json $v >> $O/variants.txt <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"])
P
done
cat $O/variants.txt
