#!/bin/bash
# A/B of the chunk pipeline of the host entry points on one box: the host-path tests, then tools/latency_probe.py (mid-size calls) and tools/pcie_probe.py (a 512 MB call)
# under TKZ_D2H_ENGINE / TKZ_HOST_CHUNK_BYTES / TKZ_HOST_CHUNK_MIN settings.  SPECS="engine:chunk_mb:min_mb ..." (engine -2: the library's choice; 0 MB: the default)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/${1:-sdma}; mkdir -p $O
if [ -z "${NO_TESTS:-}" ]; then ( timeout 900 python -m pytest tests -m gpu -x -q -k "${KEXPR:-host_path or utf16 or pinned or small_batches}" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log; fi
show() { python - "$1" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))["mid_size_host_batches"]; print({k:(v["pinned_us"],v["pinned_MBps"],v["pageable_MBps"]) for k,v in d.items()})
P
}
for spec in ${SPECS:-"-1:0:0" "-2:0:0" "-2:32:0" "-2:8:4" "-2:16:4"}; do
  IFS=: read eng mb mn <<< "$spec"
  if [ "$eng" != -2 ]; then export TKZ_D2H_ENGINE=$eng; else unset TKZ_D2H_ENGINE; fi
  if [ "$mb" != 0 ]; then export TKZ_HOST_CHUNK_BYTES=$((mb<<20)); else unset TKZ_HOST_CHUNK_BYTES; fi
  if [ "$mn" != 0 ]; then export TKZ_HOST_CHUNK_MIN=$((mn<<20)); else unset TKZ_HOST_CHUNK_MIN; fi
  T=e${eng}_c${mb}_m${mn}
  timeout 300 python tools/latency_probe.py > $O/latency_$T.json 2> $O/latency_$T.err; echo "engine $eng chunk $mb MB min $mn MB rc=$?"; show $O/latency_$T.json
  if [ -z "${NO_PCIE:-}" ]; then timeout 300 python tools/pcie_probe.py > $O/pcie_$T.txt 2>&1; grep -E "pinned call|pageable call" $O/pcie_$T.txt | tail -4 | tr '\n' ';'; echo; fi
done
if [ -n "${TRACE_MB:-}" ]; then
  # (under rocprofv3 the downloads have been seen to fall back to the runtime's copies -- profiles/r06/sdma_engines.txt is the untraced timing; TKZ_TRACE_HOST shows the host's side)
  for mb in $TRACE_MB; do
    ( cd /tmp && export TMPDIR=/tmp TKZ_TRACE_HOST=1 && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/$O/midtrace_${mb} -- python $REPO/tools/midsize_trace.py run $mb 12 ) > $O/midtrace_${mb}.json 2> $O/midtrace_${mb}.err
    tail -1 $O/midtrace_${mb}.json; tail -1 $O/midtrace_${mb}.err; python tools/midsize_trace.py show $O/midtrace_${mb} > $O/midtrace_${mb}.txt; find $O/midtrace_${mb} -name "*.csv" -size +2M -delete
  done
fi
