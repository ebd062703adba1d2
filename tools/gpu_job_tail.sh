#!/bin/bash
# round-4 GPU job for the giant-piece merger (tkz_bpe_long_tail: several proposals a thread, local bounds, rounds for chains of equal pairs): the tests that reach
# it, the long-run shapes with variant builds (tokenizer_amd/lib_rf: TKZ_ROUNDS_FIRST, the routing before; lib_prof: the development counters), the shapes
# again with the oracle comparing every document.   usage: tools/gpu_job_tail.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-tail}; O=gpurun_out/$TAG; mkdir -p $O
LONGRUN="--kind 3 --docs 32768 --min-len 30000 --max-len 34000 --heldout-steps 0"
( time timeout 600 python -m pytest tests -m gpu -x -q -k "giant or adversarial or long_diverse or long_runs" ) > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_subset.log | tail -3
rm -f $O/variants.txt
for pat in 2 4; do
for v in lib lib_rf; do
  [ -f tokenizer_amd/$v/libtkz.so ] || continue
  TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 300 python bench.py $LONGRUN --pattern $pat --no-cpu-baseline --steps 4 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --no-piece-stats > $O/b_${v}_$pat.json 2>> $O/variants.err
  python - $O/b_${v}_$pat.json $v $pat >> $O/variants.txt <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "pattern", sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"])
P
done
done
cat $O/variants.txt
if [ -f tokenizer_amd/lib_prof/libtkz.so ]; then
  for pat in 2 4; do
    TKZ_DEV_ABLATE=16 TKZ_LIBTKZ=$REPO/tokenizer_amd/lib_prof/libtkz.so timeout 300 python bench.py $LONGRUN --pattern $pat --no-cpu-baseline --steps 1 --warmup 0 --pipelined-steps 0 --no-memo-steps 0 --no-piece-stats > $O/b_prof_$pat.json 2> $O/devprof_$pat.txt
    echo "pattern $pat:"; grep "devprof" $O/devprof_$pat.txt | tail -3 | cut -c1-420
  done
fi
rm -f $O/bench_shapes.jsonl
for pat in 2 4; do
  timeout 600 python bench.py $LONGRUN --pattern $pat --parity-only --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape pattern $pat rc=$?"
done
python - $O/bench_shapes.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["pattern"][:12], d["value"], d.get("value_no_memo"), d.get("value_two_in_flight"), d["ms_per_step"], d["parity"][:28], d["roofline"]["kernels_ms"])
P
