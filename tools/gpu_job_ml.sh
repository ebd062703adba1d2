#!/bin/bash
# development job of round 6's k_merge_long work: A/B of variant builds on the mixed and the real-text shapes + the development counters of the merge kernel
#   usage: tools/gpu_job_ml.sh <tag> "<variant dirs>" [prof]      (variants: tokenizer_amd/<dir>/libtkz.so; lib_prof: make DEVPROF=1 OUT=../lib_prof)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-ml}; VARS=${2:-lib}; O=gpurun_out/$TAG; mkdir -p $O
COMMON="--no-cpu-baseline --steps 4 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --real-text-mb 0 --heldout-steps 0"
for v in $VARS; do
  IFS='|' read -ra XS <<< "${EXTRA_SHAPES:-}"          # e.g. EXTRA_SHAPES='head:|heldout:--vocab synth100k_heldout'
  for shape in "mixed:--kind 2 --docs 2000000" "real:--kind 6 --vocab gpt2 --pattern 1" "${XS[@]}"; do
    [ -n "$shape" ] || continue
    name=${shape%%:*}; args=${shape#*:}
    TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 600 python bench.py $COMMON $args > $O/${name}_$v.json 2>> $O/err.txt
    python - $O/${name}_$v.json $v $name <<'P'
import json,sys
d=json.load(open(sys.argv[1])); k=d["roofline"]["kernels_ms"]
print(sys.argv[3], sys.argv[2], "GB/s", round(d["value"]/1000,1), "ms", d["ms_per_step"], d["parity"][:40], "long_group", k.get("k_merge_long_group"), "short", k.get("k_merge_short"), "probe", k.get("k_probe"), "place", k.get("k_place"))
P
  done
done
if [ "${3:-}" = prof ]; then
  for shape in "--kind 2 --docs 2000000" "--kind 6 --vocab gpt2 --pattern 1"; do
    TKZ_DEV_ABLATE=16 TKZ_LIBTKZ=$REPO/tokenizer_amd/lib_prof/libtkz.so timeout 600 python bench.py $COMMON --steps 2 $shape 2>&1 >/dev/null | grep 'k_merge_long' | tail -1 | tee -a $O/prof.txt
  done
fi
