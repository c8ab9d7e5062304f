#!/bin/bash
set -o pipefail
OUT=gpurun_out
python -m pytest tests/test_gpu_frontend.py tests/test_gpu_prompts.py -m gpu -q -x > $OUT/r4l_tests.log 2>&1 || { tail -40 $OUT/r4l_tests.log; exit 1; }
tail -3 $OUT/r4l_tests.log
bash tools/r4_profiles.sh
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r4prof/bench_default.json").read().strip().splitlines()[-1])
print(j["workloads"]["bucketed_eval_coarse"])
PY
