#!/bin/bash
# same-box A/B of 128-row-tile schedules (builds v1: requests every 3 MFMAs; v2: everything two slots later)
set -o pipefail
OUT=gpurun_out
run() {  # label lib batch
  F5HIP_LIB="$2" timeout -k 10 300 python bench.py --batch $3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $OUT/r4au_ab.json 2>$OUT/r4au_ab.err || { tail -5 $OUT/r4au_ab.err; return 1; }
  python - "$1" "$3" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r4au_ab.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel'][:8]} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:5])
print(f"B={sys.argv[2]} [{sys.argv[1]}]: {d['value']:.0f} mel-frames/s {d['ms_per_step']:.2f} ms | {k}", flush=True)
PY
}
L=$PWD/eraxvif5tts_amd/lib
(for b in 1 2 4; do run "default" "" $b && run "v1" $L/libf5hip_v1.so $b && run "v2" $L/libf5hip_v2.so $b && run "default" "" $b || exit 1; done) 2>&1 | tee $OUT/r4au_ab.log
