#!/bin/bash
# development A/B of ONE build under two environments (e.g. TKZ_NO_FORK=1 against the default), every shape with --parity-only (the oracle compares every document)
#   usage: tools/gpu_job_ab_env.sh <tag> "<ENV=1 or ->" ...      shapes: mixed real head heldout (SHAPES_AB to choose)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-ab}; shift; O=gpurun_out/$TAG; mkdir -p $O
COMMON="--parity-only --steps 4 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --real-text-mb 0 --heldout-steps 0 --no-first-call"
for envset in "$@"; do
  for shape in ${SHAPES_AB:-mixed real head heldout}; do
    case $shape in mixed) args="--kind 2 --docs 2000000";; real) args="--kind 6 --vocab gpt2 --pattern 1 --real-text-mb 256";; head) args="";; heldout) args="--vocab synth100k_heldout";; esac
    name=${shape}_$(echo "$envset" | tr -c 'A-Za-z0-9_\n' '_')
    ( [ "$envset" != "-" ] && export $envset; timeout 900 python bench.py $COMMON $args > $O/$name.json 2>> $O/err.txt )
    python - $O/$name.json "$envset" $shape <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["roofline"]["kernels_ms"]
    print(sys.argv[3], sys.argv[2], "GB/s", round(d["value"]/1000,1), "ms", d["ms_per_step"], d["parity"][:44], "long_group", k.get("k_merge_long_group"), "short", k.get("k_merge_short"), "probe", k.get("k_probe"), "place", k.get("k_place"))
except Exception as ex: print(sys.argv[3], sys.argv[2], "FAILED", ex)
P
  done
done
tail -5 $O/err.txt
