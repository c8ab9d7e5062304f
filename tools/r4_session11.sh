#!/bin/bash
set -o pipefail
OUT=gpurun_out
python __graft_entry__.py smoke > $OUT/r4o_smoke.log 2>&1 || { tail -20 $OUT/r4o_smoke.log; exit 1; }
tail -5 $OUT/r4o_smoke.log
python -m pytest tests -m gpu -q -x --durations=8 > $OUT/r4o_tests.log 2>&1 || { tail -40 $OUT/r4o_tests.log; exit 1; }
tail -12 $OUT/r4o_tests.log
bash tools/r4_profiles.sh > $OUT/r4o_profiles.log 2>&1; tail -8 $OUT/r4o_profiles.log
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r4prof/bench_default.json").read().strip().splitlines()[-1])
print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in j["workloads"].items()})
PY
