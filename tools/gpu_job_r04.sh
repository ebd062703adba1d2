#!/bin/bash
# round-4 GPU job: stages selected by name.  usage: tools/gpu_job_r04.sh <tag> "<stages>"   stages: tests subset smoke bench variants profile profile_mixed shapes fuzz
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-r04}; STAGES=${2:-"tests bench"}; O=gpurun_out/$TAG; mkdir -p $O
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
if has tests; then ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_gpu.log | tail -4; fi
if has subset; then ( time timeout 1200 python -m pytest tests -m gpu -x -q -k "${KEXPR:-pieces or giant or arena or batch_vs_oracle or dense or memo or miss or adversarial or vocab_key or golden or errors or host_path or leak}" ) > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_subset.log | tail -4; fi
if has smoke; then timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log; fi
if has profile; then bash tools/gpu_profile.sh $TAG 10000000 "--no-memo-steps 0" > $O/profile.log 2>&1; echo "profile rc=$?"; cp gpurun_out/prof_$TAG/traffic.json profiles/traffic_latest.json 2>/dev/null; fi
if has profile_mixed; then bash tools/gpu_profile.sh ${TAG}_mixed 2000000 "--kind 2 --no-memo-steps 0" > $O/profile_mixed.log 2>&1; echo "profile mixed rc=$?"; fi
if has bench; then timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-400 $O/bench_n1.json; fi
if has variants; then
  rm -f $O/variants.txt
  for v in ${VARIANTS:-lib lib_w1 lib_noxcd}; do
    [ -f tokenizer_amd/$v/libtkz.so ] || continue
    TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 ${VARGS:-} > $O/b_$v.json 2>> $O/variants.err
    python - $O/b_$v.json $v >> $O/variants.txt <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d.get("value_no_memo"), d["ms_per_step"], d["roofline"]["kernels_ms"])
P
  done
  cat $O/variants.txt
fi
if has shapes; then
  rm -f $O/bench_shapes.jsonl
  IFS='|' read -ra SPECS <<< "${SHAPES:---kind 2 --docs 2000000|--kind 4 --docs 4000000|--kind 5|--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000|--kind 3 --pattern 4 --docs 32768 --min-len 30000 --max-len 34000|--kind 2 --pattern 4 --docs 2000000|--kind 2 --pattern 3 --docs 2000000|--pattern 4|--vocab gpt2|--vocab gpt2 --pattern 1}"
  for spec in "${SPECS[@]}"; do
    # (--parity-only: the oracle compares EVERY document of the shape's batch; no thread sweep, no PCIe / host-API legs)
    timeout 900 python bench.py $spec --parity-only --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; echo "shape [$spec] rc=$?"
  done
  python - $O/bench_shapes.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["pattern"][:12], d["config"]["vocab"][:10], d["config"]["workload"][:40], d["value"], d.get("value_no_memo"), d.get("value_two_in_flight"), d["ms_per_step"], d["parity"][:28], d["roofline"]["kernels_ms"])
P
fi
if has latency; then timeout 300 python tools/latency_probe.py > $O/latency.json 2> $O/latency.err; echo "latency rc=$?"; cat $O/latency.json; fi
if has fuzz; then timeout 200 python tools/gpu_fuzz.py 60 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log; fi
if has o2fuzz; then ( timeout 300 python tools/o200k_scan_fuzz.py --gpu --pattern 4 --seeds 40 --first-seed 200; timeout 200 python tools/o200k_scan_fuzz.py --gpu --pattern 3 --seeds 20 --first-seed 300 ) > $O/o200k_fuzz.log 2>&1; echo "o2fuzz rc=$?"; tail -3 $O/o200k_fuzz.log; fi
if has cold; then timeout 300 python tools/cold_probe.py synth100k_heldout > $O/cold_heldout.json 2> $O/cold.err; timeout 300 python tools/cold_probe.py synth100k >> $O/cold_heldout.json 2>> $O/cold.err; echo "cold rc=$?"; cat $O/cold_heldout.json; fi
if has fuzzlong; then timeout 460 python tools/gpu_fuzz.py 420 7 > $O/fuzz_long.log 2>&1; echo "fuzzlong rc=$?"; tail -1 $O/fuzz_long.log; ( timeout 400 python tools/o200k_scan_fuzz.py --gpu --pattern 4 --seeds 60 --first-seed 1000; timeout 300 python tools/o200k_scan_fuzz.py --gpu --pattern 3 --seeds 40 --first-seed 2000 ) > $O/o200k_fuzz_long.log 2>&1; echo "o2fuzzlong rc=$?"; tail -2 $O/o200k_fuzz_long.log; fi
