#!/bin/bash
set -o pipefail
OUT=gpurun_out
run() {  # knobs
  F5HIP_TUNING="$1" timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4af_ab.json 2>$OUT/r4af_ab.err || { tail -5 $OUT/r4af_ab.err; return 1; }
  python - "$1" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4af_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:5])
print(f"[{sys.argv[1]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run "" && run "gemm_group=4" && run "gemm_group=16" && run "gemm_group=32" && run "gemm_reverse_sites=0" && run "gemm_reverse_sites=15" && run "gemm_reverse_sites=4" && run "gemm_group_sites=16081608" && run "") 2>&1 | tee $OUT/r4af_ab.log
