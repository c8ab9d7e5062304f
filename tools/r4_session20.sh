#!/bin/bash
set -o pipefail
OUT=gpurun_out
python __graft_entry__.py smoke > $OUT/r4ac_smoke.log 2>&1 || { tail -20 $OUT/r4ac_smoke.log; exit 1; }
tail -3 $OUT/r4ac_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/r4ac_tests.log 2>&1 || { tail -40 $OUT/r4ac_tests.log; exit 1; }
tail -9 $OUT/r4ac_tests.log
python3 bench.py > $OUT/r4ac_bench.json 2> $OUT/r4ac_bench.err || tail -5 $OUT/r4ac_bench.err
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r4ac_bench.json").read().strip().splitlines()[-1])
print("C2", j["value"], j["ms_per_step"], "roofline", j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel"][:60])
print(" ".join(f"{x['kernel'][:12]} {x['ms']*1e3:.1f} ({x.get('frac')})" for x in j["roofline"]["kernels"][:8]))
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in j["workloads"].items()})
PY
