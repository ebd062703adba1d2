#!/bin/bash
# A/B of k_merge_short variants on one box: a libtkz.so per variant (VARIANTS = directories under tokenizer_amd/), four workloads each:
# the headline batch with its held-out leg, 436 MB of real text under the gpt2 table / pattern 1, mixed UTF-8.  K_MERGE_SHORT by HIP events.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/${1:-ms}; mkdir -p $O; rm -f $O/table.txt
for v in ${VARIANTS:-lib_r06base lib}; do
  [ -f tokenizer_amd/$v/libtkz.so ] || continue
  export TKZ_LIBTKZ=$REPO/tokenizer_amd/$v/libtkz.so
  timeout 600 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --pipelined-steps 0 --no-memo-steps 0 --real-text-mb 0 --heldout-steps 3 --no-first-call > $O/head_$v.json 2>> $O/err.txt
  timeout 600 python bench.py --kind 6 --real-text-mb 0 --vocab gpt2 --pattern 1 --steps 5 --warmup 1 --pipelined-steps 0 --no-cpu-baseline > $O/real_$v.json 2>> $O/err.txt
  timeout 600 python bench.py --kind 6 --real-text-mb 0 --vocab gpt2 --pattern 2 --steps 5 --warmup 1 --pipelined-steps 0 --no-cpu-baseline > $O/real2_$v.json 2>> $O/err.txt
  timeout 600 python bench.py --kind 2 --docs 2000000 --parity-only --steps 3 --warmup 1 > $O/mixed_$v.json 2>> $O/err.txt
  python - $O $v >> $O/table.txt <<'P'
import json,sys
O,v=sys.argv[1:3]
def ld(n):
    try: return json.load(open("%s/%s_%s.json"%(O,n,v)))
    except Exception as ex: return None
h=ld("head"); r=ld("real"); r2=ld("real2"); m=ld("mixed")
row=[v]
if h: row += ["head %.1f GB/s %.2f ms ms_short %.3f"%(h["value"]/1e3,h["ms_per_step"],h["roofline"]["kernels_ms"]["k_merge_short"]), "heldout %.1f %s"%((h.get("value_heldout_vocab") or 0)/1e3, (h.get("heldout_vocab") or {}).get("parity","")[:12]), h["parity"][:9]]
for nm,d in (("real",r),("real2",r2),("mixed",m)):
    if d: row += ["%s %.1f GB/s %.2f ms ms_short %.3f %s"%(nm,d["value"]/1e3,d["ms_per_step"],d["roofline"]["kernels_ms"]["k_merge_short"],d["parity"][:9])]
print(" | ".join(row))
P
  tail -1 $O/table.txt
done
tail -3 $O/err.txt
