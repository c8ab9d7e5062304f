#!/usr/bin/env bash
# Round-4 profile collection on the GPU box (final build): the default bench line, kernel stats of the C2 / C4 bench commands, in-situ HBM traffic
# per kernel (PMC, separate passes) and the QKV traffic record bench.py reads.
set -u
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r4prof
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/c2.log 2>&1 || tail -5 $O/c2.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > $O/c4.log 2>&1 || tail -5 $O/c4.log
timeout -k 10 500 python3 tools/pmc_insitu_traffic.py collect $O/insitu > $O/insitu.log 2>&1
python3 tools/pmc_insitu_traffic.py summarise $O/insitu $O/r4_qkv_traffic.json > $O/r4_insitu_hbm_traffic_per_kernel.txt 2>&1
for d in c2 c4; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${d}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O/insitu -name "*counter_collection.csv" -size +8M -delete
python3 tools/prof_summary.py $O/c2_kernel_stats.csv 4 12
python3 tools/prof_summary.py $O/c4_kernel_stats.csv 3 8
head -14 $O/r4_insitu_hbm_traffic_per_kernel.txt
python3 - <<PY
import json
j=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("C2", j["value"], j["ms_per_step"], "roofline", j["roofline"]["achieved"], j["roofline"]["frac"], "traffic", j["roofline"]["traffic"])
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in j["workloads"].items()})
print(j["workloads"]["bucketed_eval"])
PY
