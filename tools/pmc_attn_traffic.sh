#!/usr/bin/env bash
# HBM-side traffic and L2 hit rate of the attention kernel (separate rocprofv3 --pmc passes; FETCH_SIZE reads 1/2 on gfx950).
# usage: tools/pmc_attn_traffic.sh <tag> <variant> <B*2> <N>
set -u
tag="$1"; v="$2"; B="$3"; N="$4"
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_t$i -- python3 tools/attn_one.py $v $B $N 16 3 > gpurun_out/${tag}_t$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/${tag}_t$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for i in (1, 2, 3):
    for f in glob.glob("gpurun_out/${tag}_t%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
B, N = $B, $N
alg = B * N * 16 * 64 * 2 * 4  # q, k, v read once + o written, bf16
print(f"attention variant $v, B*2 = {B}, N = {N}, H = 16: per launch (mean of {len(acc.get('FETCH_SIZE', []))})")
if "FETCH_SIZE" in m: print(f"  fetch {m['FETCH_SIZE'] * 2048 / 1e6:.1f} MB   write {m.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB   algorithmic (q, k, v once + o) {alg / 1e6:.1f} MB")
if "TCC_HIT_sum" in m: print(f"  L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}  (hits {m['TCC_HIT_sum']:.3e}, misses {m['TCC_MISS_sum']:.3e})")
PY
