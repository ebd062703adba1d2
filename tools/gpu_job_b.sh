#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; O=gpurun_out/variants; mkdir -p $O
for spec in "--kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000" "--kind 3 --pattern 1 --docs 32768 --min-len 30000 --max-len 34000"; do
    timeout 300 python bench.py $spec --no-cpu-baseline --steps 3 --warmup 1 2>>$O/err.log | grep "^{" | python -c "
import sys, json
j=json.loads(sys.stdin.readline()); print('$spec', j['value'], j['ms_per_step'], j['roofline']['kernels_ms'])"
done

timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "giant or adversarial or pieces_vs or device_corpus and 1500" 2>&1 | tail -3
TKZ_LIBTKZ=$REPO/tokenizer_amd/lib_dev/libtkz.so TKZ_DEV_ABLATE=16 timeout 300 python bench.py --kind 3 --pattern 2 --docs 32768 --min-len 30000 --max-len 34000 --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "giant" | tail -1
