#!/usr/bin/env bash
# throughput over (utterances, frames): looks for shapes where the tile / kernel choice falls off a cliff
set -u
cd "$(dirname "$0")/.."
for n in ${NS:-512 1024 1536 2048}; do
  for b in ${BS:-1 2 3 4 6 8 16}; do
    timeout -k 10 120 python bench.py --batch $b --seq-len $n --nfe 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={x['kernel']:round(x['ms']*1e3,1) for x in d['roofline']['kernels'][:7]}
print(f\"N=$n B=$b rows={2*$b*$n:6d}: {d['value']:8.0f} mel-frames/s  {d['ms_per_step']:8.2f} ms  {k}\")
"
  done
done
