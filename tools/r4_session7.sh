#!/bin/bash
set -o pipefail
OUT=gpurun_out
for ink in 1 0 1 0; do
  for b in 1 2; do
    F5HIP_TUNING=ln_fold_inkernel=$ink python bench.py --batch $b --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r4h_bench_b${b}_ink$ink.json 2> $OUT/r4h_bench_b${b}_ink$ink.err || { tail $OUT/r4h_bench_b${b}_ink$ink.err; exit 1; }
    python - <<PY
import json
j=json.loads(open("$OUT/r4h_bench_b${b}_ink$ink.json").read().strip().splitlines()[-1])
print("B=$b inkernel=$ink", j["value"], "mel-frames/s", j["ms_per_step"], "ms", {k["kernel"][:8]: k["ms"] for k in j["roofline"]["kernels"]})
PY
  done
done
