#!/usr/bin/env bash
# per-GPU shard shapes of the strong-scaling run (32 utterances over N GPUs -> 32/N per GPU) and the single-utterance shape, final build
set -u
cd "$(dirname "$0")/.."
for b in 32 16 8 4 1; do
  timeout -k 10 200 python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/sb_$b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/sb_$b.json").read().strip().splitlines()[-1])
k=" ".join(f"{x['kernel']} {x['ms']*1e3:.1f}" for x in d['roofline']['kernels'])
print(f"B=$b: {d['value']:.0f} mel-frames/s, {d['ms_per_step']:.2f} ms per sample(), loop_mfma_frac {d.get('loop_mfma_frac')} | us per launch: {k}")
PY
done
