#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
F5HIP_TUNING="conv31_tok=128" timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -k "conv or rows_are_independent or ragged_chunks" > gpurun_out/conv_forced.log 2>&1; tail -2 gpurun_out/conv_forced.log
for b in 1 2; do for v in 256 128 256 128; do
  F5HIP_TUNING="conv31_tok=$v" timeout -k 10 200 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel']:round(x['ms']*1e3,1) for x in d['roofline']['kernels']}
print('B=$b tok=$v', d['value'], d['ms_per_step'], k['conv31'])
"
done; done
