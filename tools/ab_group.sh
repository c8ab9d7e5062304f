#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
for v in ${VS:-0 8 0 8 0 8}; do
  F5HIP_TUNING="gemm_group=$v" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gemm_group=$v', d['value'], {x['kernel']:round(x['ms']*1e3,1) for x in d['roofline']['kernels'][:5]})
"
done
