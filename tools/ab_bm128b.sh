#!/usr/bin/env bash
set -u
cd "$(dirname "$0")/.."
run() {
  F5HIP_TUNING="$2" timeout -k 10 200 python bench.py --batch $1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/abbm.json 2>/dev/null
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/abbm.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel']} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} {d['ms_per_step']:.2f} ms | {k}")
PY
}
for b in 1 2 3; do
  run $b "gemm_bm128=1"
  run $b "gemm_bm128=3"
  run $b "gemm_bm128=1"
  run $b "gemm_bm128=3"
done
