#!/bin/bash
# round 4, GPU session 2: LayerNorm fold -- parity suites, then same-box A/B of the fold at C2 / shard / single-utterance shapes
set -o pipefail
OUT=gpurun_out
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_vocoder_wrapper.py -m gpu -q -x > $OUT/r4c_tests.log 2>&1 || { tail -40 $OUT/r4c_tests.log; exit 1; }
tail -3 $OUT/r4c_tests.log
for fold in 1 0 1 0; do
  F5HIP_TUNING=ln_fold=$fold python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/r4c_bench_c2_fold$fold.json 2> $OUT/r4c_bench_c2_fold$fold.err || { tail -20 $OUT/r4c_bench_c2_fold$fold.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("$OUT/r4c_bench_c2_fold$fold.json").read().strip().splitlines()[-1])
print("C2 fold=$fold", j["value"], "mel-frames/s", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
done
for fold in 1 0; do
  for b in 4 1; do
    F5HIP_TUNING=ln_fold=$fold python bench.py --batch $b --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r4c_bench_b${b}_fold$fold.json 2> $OUT/r4c_bench_b${b}_fold$fold.err || exit 1
    python - <<PY
import json
j=json.loads(open("$OUT/r4c_bench_b${b}_fold$fold.json").read().strip().splitlines()[-1])
print("B=$b fold=$fold", j["value"], "mel-frames/s", j["ms_per_step"], "ms", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
  done
done
