#!/usr/bin/env bash
# in-situ coordinate sweep of the L2 patch height per call site (qkv|out|ff1|ff2, two digits each)
set -u
cd "$(dirname "$0")/.."
for v in ${VS:-8080808 4080808 16080808 8040808 8160808 8080408 8081608 8080804 8080816 8080808}; do
  F5HIP_TUNING="gemm_group_sites=$v" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel']:round(x['ms']*1e3,1) for x in d['roofline']['kernels'][:5]}
print('sites=$v', d['value'], k, round(sum(k.values()),1))
"
done
