#!/bin/bash
set -o pipefail
OUT=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "w4_kernel" > $OUT/r4ae_tests.log 2>&1; rc=$?
tail -4 $OUT/r4ae_tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4ae_bench.json 2> $OUT/r4ae_bench.err || tail -5 $OUT/r4ae_bench.err
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r4ae_bench.json").read().strip().splitlines()[-1])
print("C2", j["value"], j["ms_per_step"], "roofline", j["roofline"]["achieved"], j["roofline"]["frac"])
print(" ".join(f"{x['kernel'][:12]} {x['ms']*1e3:.1f} ({x.get('frac')})" for x in j["roofline"]["kernels"][:8]))
PY
done
