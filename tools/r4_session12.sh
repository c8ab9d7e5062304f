#!/bin/bash
# in-launch row statistics (ln_fold_fin): parity, then same-box A/B at the small shapes; vendor-library GEMM reference beside tools/gemm_ab.py
set -o pipefail
OUT=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "in_launch or in_kernel or range_guard or fold" > $OUT/r4p_tests.log 2>&1 || { tail -40 $OUT/r4p_tests.log; exit 1; }
tail -3 $OUT/r4p_tests.log
run() {  # batch, knobs
  F5HIP_TUNING="$2" timeout -k 10 200 python bench.py --batch $1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4p_ab.json 2>$OUT/r4p_ab.err || { tail -5 $OUT/r4p_ab.err; return 1; }
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4p_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel']} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:8])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
for b in 1 2 4; do
  run $b "ln_fold_fin=0" && run $b "ln_fold_fin=1" && run $b "ln_fold_fin=0" && run $b "ln_fold_fin=1" || exit 1
done 2>&1 | tee $OUT/r4p_ab.log
timeout -k 10 120 python tools/hipblaslt_ref.py 65536 2>&1 | tee $OUT/r4p_hipblaslt.log
timeout -k 10 120 python tools/gemm_ab.py 65536 base 2>&1 | tee $OUT/r4p_gemm_ab.log
