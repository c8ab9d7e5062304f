#!/bin/bash
# round 4, GPU session 4: op-level LayerNorm-fold test, prompts test, e2e numbers, then the fold A/B benches
OUT=gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "layernorm_fold" > $OUT/r4e_fold_op.log 2>&1
grep -a "ln_fold M=\|passed\|failed\|Error\|assert" $OUT/r4e_fold_op.log | head -40
for fold in 0 1; do
  F5HIP_TUNING=ln_fold=$fold python -m pytest tests/test_gpu_vocoder_wrapper.py -m gpu -q -s -k "oracle_chain and bf16" > $OUT/r4e_e2e_fold$fold.log 2>&1
  echo "fold=$fold: $(grep -a 'vs oracle chain \[' $OUT/r4e_e2e_fold$fold.log | tail -1)"
done
python -m pytest tests/test_gpu_prompts.py -m gpu -q -x > $OUT/r4e_prompts.log 2>&1; tail -15 $OUT/r4e_prompts.log
for fold in 1 0 1 0; do
  F5HIP_TUNING=ln_fold=$fold python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > $OUT/r4e_bench_c2_fold$fold.json 2> $OUT/r4e_bench_c2_fold$fold.err || { tail -20 $OUT/r4e_bench_c2_fold$fold.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("$OUT/r4e_bench_c2_fold$fold.json").read().strip().splitlines()[-1])
print("C2 fold=$fold", j["value"], "mel-frames/s", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
done
for fold in 1 0; do
  for b in 4 1; do
    F5HIP_TUNING=ln_fold=$fold python bench.py --batch $b --no-extra --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r4e_bench_b${b}_fold$fold.json 2> $OUT/r4e_bench_b${b}_fold$fold.err || exit 1
    python - <<PY
import json
j=json.loads(open("$OUT/r4e_bench_b${b}_fold$fold.json").read().strip().splitlines()[-1])
print("B=$b fold=$fold", j["value"], "mel-frames/s", j["ms_per_step"], "ms", {k["kernel"]: k["ms"] for k in j["roofline"]["kernels"]})
PY
  done
done
