#!/bin/bash
# After a full tools/gpu_job_r06.sh run: the summaries the judge reads go from gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/copy_profiles.sh [tag=r06]
set -u
T=${1:-r06}; G=gpurun_out; P=profiles
for sfx in "" _mixed _real; do
  d=$P/$T$sfx; mkdir -p $d
  cp $G/prof_$T$sfx/summary.txt $G/prof_$T$sfx/traffic.json $d/ 2>/dev/null
  cp $G/prof_$T$sfx/trace/trace_kernel_stats.csv $d/kernel_trace_stats.csv 2>/dev/null
done
for f in adapt_probe.jsonl bench_kind6.jsonl bench_n1.json bench_shapes.jsonl cold_probe.jsonl fuzz.log lane_fuzz_gpu.log latency.json tail_fuzz_gpu.log; do cp $G/$T/$f $P/$T/ 2>/dev/null; done
tail -6 $G/$T/pytest_gpu.log > $P/$T/pytest_gpu_tail.txt
cp $G/giant_100kb_ms.txt $P/$T/ 2>/dev/null
cp $G/prof_$T/traffic.json $P/traffic_latest.json; cp $G/prof_${T}_real/traffic.json $P/traffic_real_latest.json
python3 - <<'P'
import json,sys
sys.path.insert(0,'.')
import bench
print("sources", bench.kernel_sources_sha(), "traffic_latest", json.load(open('profiles/traffic_latest.json')).get('src_sha'), "real", json.load(open('profiles/traffic_real_latest.json')).get('src_sha'))
P
