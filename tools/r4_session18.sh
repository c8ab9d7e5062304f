#!/bin/bash
set -o pipefail
OUT=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "w4_kernel or in_place_residual or fold_site" > $OUT/r4x_tests.log 2>&1; rc=$?
tail -8 $OUT/r4x_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/gemm_ab.py 65536 base gemm_w4=0 2>&1 | tee $OUT/r4x_gemm_ab.log
run() {  # knobs
  F5HIP_TUNING="$1" timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4x_ab.json 2>$OUT/r4x_ab.err || { tail -5 $OUT/r4x_ab.err; return 1; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4x_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:14]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:8])
print(f"[{sys.argv[1]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run "gemm_w4=0" && run "gemm_w4=1" && run "gemm_w4=1") 2>&1 | tee $OUT/r4x_ab.log
