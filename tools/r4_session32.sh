#!/bin/bash
set -o pipefail
OUT=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py -m gpu -q -x -k "w4_kernel or range_guard" > $OUT/r4at_tests.log 2>&1; rc=$?
tail -6 $OUT/r4at_tests.log
[ $rc -eq 0 ] || exit 1
run() {  # batch knobs
  F5HIP_TUNING="$2" timeout -k 10 300 python bench.py --batch $1 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4at_ab.json 2>$OUT/r4at_ab.err || { tail -5 $OUT/r4at_ab.err; return 1; }
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4at_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:7])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run 4 "gemm_w4_ink=1" && run 4 "" && run 4 "gemm_w4_ink=1" && run 4 "" && run 3 "gemm_w4_ink=1" && run 3 "" && run 6 "gemm_w4_ink=1" && run 6 "" && run 32 "") 2>&1 | tee $OUT/r4at_ab.log
