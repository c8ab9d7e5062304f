#!/bin/bash
# round 4, GPU session 3: e2e bf16 error with / without the LayerNorm fold (informational), bucketed-inference tests, fold A/B benches
set -o pipefail
OUT=gpurun_out
for fold in 0 1; do
  F5HIP_TUNING=ln_fold=$fold python -m pytest tests/test_gpu_vocoder_wrapper.py -m gpu -q -s -k "oracle_chain and bf16" > $OUT/r4d_e2e_fold$fold.log 2>&1
  echo "fold=$fold: $(grep -a 'vs oracle chain' $OUT/r4d_e2e_fold$fold.log | tail -1)"
done
python -m pytest tests/test_gpu_prompts.py -m gpu -q -x > $OUT/r4d_prompts.log 2>&1 || { tail -40 $OUT/r4d_prompts.log; exit 1; }
tail -2 $OUT/r4d_prompts.log
for fold in 1 0 1 0; do
  F5HIP_TUNING=ln_fold=$fold python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/r4d_bench_c2_fold$fold.json 2> $OUT/r4d_bench_c2_fold$fold.err || { tail -20 $OUT/r4d_bench_c2_fold$fold.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("$OUT/r4d_bench_c2_fold$fold.json").read().strip().splitlines()[-1])
print("C2 fold=$fold", j["value"], "mel-frames/s", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
done
for fold in 1 0; do
  for b in 4 1; do
    F5HIP_TUNING=ln_fold=$fold python bench.py --batch $b --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r4d_bench_b${b}_fold$fold.json 2> $OUT/r4d_bench_b${b}_fold$fold.err || exit 1
    python - <<PY
import json
j=json.loads(open("$OUT/r4d_bench_b${b}_fold$fold.json").read().strip().splitlines()[-1])
print("B=$b fold=$fold", j["value"], "mel-frames/s", j["ms_per_step"], "ms", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
  done
done
