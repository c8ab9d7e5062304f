#!/bin/bash
set -o pipefail
OUT=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=3 > $OUT/r4an_tests.log 2>&1 || { tail -40 $OUT/r4an_tests.log; exit 1; }
tail -6 $OUT/r4an_tests.log
run() {  # batch knobs
  F5HIP_TUNING="$2" timeout -k 10 300 python bench.py --batch $1 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4an_ab.json 2>$OUT/r4an_ab.err || { tail -5 $OUT/r4an_ab.err; return 1; }
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4an_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:7])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run 1 "gemm_w4=0" && run 1 "" && run 2 "gemm_w4=0" && run 2 "" && run 3 "gemm_w4=0" && run 3 "" && run 4 "gemm_w4=0" && run 4 "" && run 6 "gemm_w4=0" && run 6 "") 2>&1 | tee $OUT/r4an_ab.log
