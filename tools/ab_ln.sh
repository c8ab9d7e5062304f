set -u
for v in 1 4 2 1 4; do
  F5HIP_TUNING="ln_rows=$v" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/ab_ln_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_ln_$v.json").read().strip().splitlines()[-1])
k={x['kernel']:round(x['ms']*1e3,1) for x in d['roofline']['kernels']}
print("ln_rows=$v", d['value'], k['ln1'], k['ln2'], k['qkv'], k['ff1'])
PY
done
