#!/usr/bin/env bash
# in-situ: which block GEMMs gain from walking their tiles in the opposite order to their producer (bit 0 qkv, 1 out, 2 ff1, 3 ff2)
set -u
cd "$(dirname "$0")/.."
for v in ${VS:-0 8 4 12 2 1 10 0 8}; do
  F5HIP_TUNING="gemm_reverse_sites=$v" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel']:round(x['ms']*1e3,1) for x in d['roofline']['kernels'][:7]}
print('reverse=$v', d['value'], k, round(sum(k.values()),1))
"
done
