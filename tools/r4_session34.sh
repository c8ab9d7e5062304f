#!/bin/bash
set -o pipefail
OUT=gpurun_out
run() {  # batch knobs
  F5HIP_TUNING="$2" timeout -k 10 300 python bench.py --batch $1 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4aw_ab.json 2>$OUT/r4aw_ab.err || { tail -5 $OUT/r4aw_ab.err; return 1; }
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4aw_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:5])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
(run 1 "" && run 1 "w_prefetch=0" && run 1 "" && run 1 "w_prefetch=0" && run 2 "" && run 2 "w_prefetch=0") 2>&1 | tee $OUT/r4aw_ab.log
