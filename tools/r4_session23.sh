#!/bin/bash
# same-box A/B of two BUILDS of the library: F5HIP_LIB=<in-tree .so> against the default
set -o pipefail
OUT=gpurun_out
run() {  # label lib
  F5HIP_LIB="$2" timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4ag_ab.json 2>$OUT/r4ag_ab.err || { tail -5 $OUT/r4ag_ab.err; return 1; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4ag_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:5])
print(f"[{sys.argv[1]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
A=$PWD/eraxvif5tts_amd/lib/libf5hip_a.so
(run "build A" $A && run "build B" "" && run "build A" $A && run "build B" "") 2>&1 | tee $OUT/r4ag_ab.log
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "w4_kernel or in_place_residual" > $OUT/r4ag_tests.log 2>&1; tail -3 $OUT/r4ag_tests.log
