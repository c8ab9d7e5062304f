#!/bin/bash
set -o pipefail
OUT=gpurun_out
run() {  # batch knobs
  F5HIP_TUNING="$2" timeout -k 10 300 python bench.py --batch $1 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4ar_ab.json 2>$OUT/r4ar_ab.err || { tail -5 $OUT/r4ar_ab.err; return 1; }
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4ar_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:7])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run 4 "" && run 4 "gemm_w4_bm=128" && run 4 "" && run 4 "gemm_w4_bm=128" && run 8 "" && run 8 "gemm_w4_bm=128" && run 32 "" && run 32 "gemm_w4_bm=128") 2>&1 | tee $OUT/r4ar_ab.log
