#!/bin/bash
# round 4, GPU session 6: in-kernel row statistics for small tiles -- parity / bit-equality suites, then A/B at the small shapes
set -o pipefail
OUT=gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_prompts.py -m gpu -q -x > $OUT/r4g_tests.log 2>&1 || { tail -40 $OUT/r4g_tests.log; exit 1; }
tail -3 $OUT/r4g_tests.log
for ink in 1 0 1 0; do
  for b in 4 1; do
    F5HIP_TUNING=ln_fold_inkernel=$ink python bench.py --batch $b --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r4g_bench_b${b}_ink$ink.json 2> $OUT/r4g_bench_b${b}_ink$ink.err || { tail $OUT/r4g_bench_b${b}_ink$ink.err; exit 1; }
    python - <<PY
import json
j=json.loads(open("$OUT/r4g_bench_b${b}_ink$ink.json").read().strip().splitlines()[-1])
print("B=$b inkernel=$ink", j["value"], "mel-frames/s", j["ms_per_step"], "ms", {k["kernel"][:8]: k["ms"] for k in j["roofline"]["kernels"]})
PY
  done
done
python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/r4g_bench_c2.json 2> $OUT/r4g_bench_c2.err || exit 1
python - <<PY
import json
j=json.loads(open("$OUT/r4g_bench_c2.json").read().strip().splitlines()[-1])
print("C2", j["value"], "mel-frames/s", {k["kernel"][:8]: k["ms"] for k in j["roofline"]["kernels"]})
PY
