#!/usr/bin/env bash
# same-box A/B of tuning knobs at the C3 shard shape (4 x 1024) and at 8 x 1024
set -u
cd "$(dirname "$0")/.."
run() {  # batch, knobs
  F5HIP_TUNING="$2" timeout -k 10 200 python bench.py --batch $1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/abb4.json 2>/dev/null
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/abb4.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel']} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'][:7])
print(f"B={sys.argv[1]} [{sys.argv[2]}]: {d['value']:.0f} | {k}")
PY
}
for b in 4 8; do
  run $b ""
  run $b "w_prefetch=0"
  run $b "ln_rows_min=4096"
  run $b "ln_rows_min=4096,ln_rows=4"
  run $b "attn_variant=5"
  run $b ""
done
