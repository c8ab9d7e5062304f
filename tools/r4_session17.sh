#!/bin/bash
set -o pipefail
OUT=gpurun_out
run() {  # knobs
  F5HIP_TUNING="$1" timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4w_ab.json 2>$OUT/r4w_ab.err || { tail -5 $OUT/r4w_ab.err; return 1; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4w_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:14]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:8])
print(f"[{sys.argv[1]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run "gemm_w4=0" && run "gemm_w4=1" && run "gemm_w4=0" && run "gemm_w4=1") 2>&1 | tee $OUT/r4w_ab.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/r4w_tests.log 2>&1; tail -12 $OUT/r4w_tests.log
