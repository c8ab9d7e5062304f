#!/bin/bash
# timing-only ablations of the in-place residual epilogue of gemm_w4 (wrong results by construction): C = no statistics, D = no stream loads
set -o pipefail
OUT=gpurun_out
run() {  # label lib
  F5HIP_LIB="$2" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/r4ah_ab.json 2>$OUT/r4ah_ab.err || { tail -5 $OUT/r4ah_ab.err; return 1; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4ah_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:5])
print(f"[{sys.argv[1]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
L=$PWD/eraxvif5tts_amd/lib
(run "default" "" && run "C no statistics" $L/libf5hip_c.so && run "D no stream loads" $L/libf5hip_d.so && run "default" "") 2>&1 | tee $OUT/r4ah_ab.log
