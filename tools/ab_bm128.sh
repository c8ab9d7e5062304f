#!/usr/bin/env bash
# 128-row token tiles: parity under forced use, then same-box A/B at the single-utterance and shard shapes
set -u
cd "$(dirname "$0")/.."
F5HIP_TUNING="gemm_bm128=2" timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q > gpurun_out/bm128_forced.log 2>&1; tail -3 gpurun_out/bm128_forced.log
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q > gpurun_out/bm128_default.log 2>&1; tail -3 gpurun_out/bm128_default.log
run() {
  F5HIP_TUNING="$2" timeout -k 10 200 python bench.py --batch $1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/abbm.json 2>/dev/null
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/abbm.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel']} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} {d['ms_per_step']:.2f} ms | {k}")
PY
}
for b in 1 2 4; do
  run $b "gemm_bm128=0"
  run $b "gemm_bm128=1"
  run $b "gemm_bm128=2"
  run $b "gemm_bm128=0"
  run $b "gemm_bm128=1"
done
