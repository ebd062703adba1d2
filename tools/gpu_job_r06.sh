#!/bin/bash
# round-6 GPU job: stages selected by name.  usage: tools/gpu_job_r06.sh <tag> "<stages>"
#   stages: tests subset smoke bench kind6 variants profile profile_mixed profile_real shapes latency midtrace cold fuzz tailfuzz lanefuzz adapt
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; TAG=${1:-r06}; STAGES=${2:-"subset bench"}; O=gpurun_out/$TAG; mkdir -p $O
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
T0=$(date +%s); lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt
if has tests; then ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; lap "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_gpu.log | tail -4; fi
if has subset; then ( time timeout 1200 python -m pytest tests -m gpu -x -q -k "${KEXPR:-promoted or memo or miss or two_halves or small_batches or batch_vs_oracle or golden or by_name or long_pieces or host_path or utf16}" ) > $O/pytest_subset.log 2>&1; lap "pytest subset rc=$?"; grep -E "passed|failed|Error|error" $O/pytest_subset.log | tail -6; fi
if has smoke; then timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; lap "smoke rc=$?"; tail -1 $O/smoke.log; fi
# (the profiles first: the bench line then carries the counted traffic and the issue figures of THESE sources -- bench.py reads profiles/traffic_latest.json)
if has profile; then bash tools/gpu_profile.sh $TAG 10000000 "--no-memo-steps 0 --real-text-mb 0 --heldout-steps 0" > $O/profile.log 2>&1; lap "profile rc=$?"; cp gpurun_out/prof_$TAG/traffic.json profiles/traffic_latest.json 2>/dev/null; fi
if has profile_mixed; then bash tools/gpu_profile.sh ${TAG}_mixed 2000000 "--kind 2 --no-memo-steps 0 --real-text-mb 0" > $O/profile_mixed.log 2>&1; lap "profile mixed rc=$?"; fi
# (the 256 MB of real text the default run's `real_text` leg encodes, under the real gpt2 table: its counted traffic goes into that leg's roofline)
if has profile_real; then bash tools/gpu_profile.sh ${TAG}_real 0 "--kind 6 --real-text-mb 256 --vocab gpt2 --pattern 1 --no-memo-steps 0" > $O/profile_real.log 2>&1; lap "profile real rc=$?"; cp gpurun_out/prof_${TAG}_real/traffic.json profiles/traffic_real_latest.json 2>/dev/null; fi
if has bench; then timeout 1200 python bench.py ${BARGS:-} > $O/bench_n1.json 2> $O/bench_n1.err; lap "bench rc=$?"; cut -c1-300 $O/bench_n1.json; tail -3 $O/bench_n1.err
  python - $O/bench_n1.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("value", d["value"], "ms", d["ms_per_step"], "no_memo", d.get("value_no_memo"), "heldout", d.get("value_heldout_vocab"), "2fl", d.get("value_two_in_flight"), "host_api", d.get("value_host_api"), d.get("host_api",{}) and d["host_api"].get("utf16"))
    print("kernels", d["roofline"]["kernels_ms"]); print("parity", d["parity"][:90]); print("promoted", d["config"].get("promoted_pieces"))
    h=d.get("heldout_vocab") or {}; print("heldout", {k:h.get(k) for k in ("value","ms_per_step","parity")}, (h.get("piece_stats") or {}))
    rt=d.get("real_text") or {}
    print("real_text corpus", rt.get("corpus"), rt.get("error"))
    for k,v in (rt.get("by_vocab") or {}).items(): print("  ", k, v["value"], "warm", v["value_warm_memo"], "nomemo", v["value_no_memo"], "B/tok", v["bytes_per_token"], v["parity"][:40], v["piece_stats"], v["kernels_ms"])
    print("cpu", d.get("cpu_baseline") and {k:d["cpu_baseline"][k] for k in ("value","cores","threads","value_1_thread")}); print("pcie", d.get("pcie_inclusive"))
except Exception as ex: print("bench summary:", ex)
P
fi
if has kind6; then
  for spec in "--vocab gpt2 --pattern 1" "--vocab gpt2 --pattern 2" "" "--vocab synth100k_heldout" "--vocab gpt2 --pattern 1 --min-len 8192 --max-len 32768"; do
    timeout 900 python bench.py --kind 6 --real-text-mb 0 $spec --steps 5 --warmup 1 --pipelined-steps 0 ${K6ARGS:-} >> $O/bench_kind6.jsonl 2>> $O/bench_kind6.err; lap "kind6 [$spec] rc=$?"
  done
  python - $O/bench_kind6.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["vocab"][:24], d["config"]["pattern"][:12], "value", d["value"], "warm", d.get("value_warm_memo"), "nomemo", d.get("value_no_memo"), d["ms_per_step"], d["parity"][:40], d["piece_stats"], d["roofline"]["kernels_ms"], d["config"]["real_text"]["bytes"], d["config"]["real_text"]["seconds_to_read"], d["config"]["real_text"]["sha256"][:12])
P
fi
if has variants; then
  rm -f $O/variants.txt
  for v in ${VARIANTS:-lib lib_a2048 lib_a3584 lib_lane32}; do
    [ -f tokenizer_amd/$v/libtkz.so ] || continue
    TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --real-text-mb 0 --heldout-steps 0 ${VARGS:-} > $O/b_$v.json 2>> $O/variants.err
    python - $O/b_$v.json $v >> $O/variants.txt <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d["parity"][:60], d["roofline"]["kernels_ms"])
P
  done
  lap "variants"; cat $O/variants.txt
  if [ -n "${VARGS2:-}" ]; then
    rm -f $O/variants2.txt
    for v in ${VARIANTS:-lib}; do
      [ -f tokenizer_amd/$v/libtkz.so ] || continue
      TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --real-text-mb 0 --heldout-steps 0 ${VARGS2} > $O/b2_$v.json 2>> $O/variants.err
      python - $O/b2_$v.json $v >> $O/variants2.txt <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d["value"], d["ms_per_step"], d["parity"][:60], d["roofline"]["kernels_ms"])
P
    done
    lap "variants2"; cat $O/variants2.txt
  fi
fi
if has shapes; then
  rm -f $O/bench_shapes.jsonl
  IFS='|' read -ra SPECS <<< "${SHAPES:---kind 2 --docs 2000000|--kind 4 --docs 4000000|--kind 5|--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000|--kind 3 --pattern 4 --docs 32768 --min-len 30000 --max-len 34000|--kind 2 --pattern 4 --docs 2000000|--kind 2 --pattern 3 --docs 2000000|--pattern 4|--vocab gpt2|--vocab gpt2 --pattern 1|--vocab synth100k_heldout}"
  for spec in "${SPECS[@]}"; do
    timeout 900 python bench.py $spec --parity-only --steps 3 --warmup 1 >> $O/bench_shapes.jsonl 2>> $O/bench_shapes.err; lap "shape [$spec] rc=$?"
  done
  python - $O/bench_shapes.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["pattern"][:12], d["config"]["vocab"][:10], d["config"]["workload"][:40], d["value"], d.get("value_no_memo"), d.get("value_two_in_flight"), d["ms_per_step"], d["parity"][:28], d["roofline"]["kernels_ms"])
P
fi
if has latency; then timeout 300 python tools/latency_probe.py > $O/latency.json 2> $O/latency.err; lap "latency rc=$?"; cat $O/latency.json; fi
if has fuzz; then timeout 200 python tools/gpu_fuzz.py 60 > $O/fuzz.log 2>&1; lap "fuzz rc=$?"; tail -2 $O/fuzz.log; fi
if has midtrace; then
  for mb in ${MIDMB:-0.25 1 4}; do
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/$O/midtrace_$mb -- python $REPO/tools/midsize_trace.py run $mb 12 ) > $O/midtrace_$mb.json 2> $O/midtrace_$mb.err
    tail -1 $O/midtrace_$mb.json; python tools/midsize_trace.py show $O/midtrace_$mb | tee $O/midtrace_$mb.txt
    find $O/midtrace_$mb -name "*.csv" -size +2M -delete
  done
  lap "midtrace"
fi
if has cold; then export TKZ_LOG_SLOW_MS=300; timeout 300 python tools/cold_probe.py synth100k_heldout > $O/cold_probe.jsonl 2> $O/cold.err; timeout 300 python tools/cold_probe.py synth100k >> $O/cold_probe.jsonl 2>> $O/cold.err; lap "cold rc=$?"; cat $O/cold_probe.jsonl; fi
if has tailfuzz; then TKZ_EMU_LIB=$REPO/tokenizer_amd/lib/libtkz.so timeout 400 python tools/tail_fuzz.py 240 > $O/tail_fuzz_gpu.log 2>&1; lap "tail fuzz rc=$?"; tail -2 $O/tail_fuzz_gpu.log; fi
if has lanefuzz; then TKZ_EMU_LIB=$REPO/tokenizer_amd/lib/libtkz.so timeout 400 python tools/lane_fuzz.py 240 > $O/lane_fuzz_gpu.log 2>&1; lap "lane fuzz rc=$?"; tail -2 $O/lane_fuzz_gpu.log; fi
if has adapt; then timeout 300 python tools/adapt_probe.py gpt2 1 1 > $O/adapt_probe.jsonl 2> $O/adapt.err; lap "adapt rc=$?"; tail -3 $O/adapt_probe.jsonl; fi
lap done
# (the `lanepiece` stage of the round -- TKZ_LATENCY_LANE_PIECE=32..128 under the 1 MB trace -- went with the knob: profiles/r05/lanepiece_latency.txt has its result)
