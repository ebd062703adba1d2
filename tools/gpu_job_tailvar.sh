#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03o; mkdir -p $O; rm -f $O/v.txt
for v in lib lib_t0g lib_t0l; do for pat in 2 3; do
  TKZ_LIBTKZ=$PWD/tokenizer_amd/$v/libtkz.so timeout 300 python bench.py --kind 3 --pattern $pat --docs 32768 --min-len 30000 --max-len 34000 --no-cpu-baseline --steps 3 --warmup 1 --no-memo-steps 0 > $O/b.json 2>> $O/err.txt
  python - $O/b.json $v $pat >> $O/v.txt <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "pattern", sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"]["k_merge_long_group"])
P
done; done; cat $O/v.txt
