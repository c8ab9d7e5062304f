#!/usr/bin/env bash
# Round-3 profile collection on the GPU box (final build): kernel stats of the C2 bench command, PMC traffic of the QKV GEMM, in-situ HBM traffic per kernel.
set -u
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out/r3prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/c2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3prof/c2.log 2>&1 || tail -5 gpurun_out/r3prof/c2.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3prof/c4 -- python3 bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3prof/c4.log 2>&1 || tail -5 gpurun_out/r3prof/c4.log
timeout -k 10 500 python3 tools/pmc_traffic.py collect gpurun_out/r3prof/traffic > gpurun_out/r3prof/traffic.log 2>&1
python3 tools/pmc_traffic.py summarise gpurun_out/r3prof/traffic gpurun_out/r3prof/r3_qkv_traffic.json >> gpurun_out/r3prof/traffic.log 2>&1
timeout -k 10 500 python3 tools/pmc_insitu_traffic.py collect gpurun_out/r3prof/insitu > gpurun_out/r3prof/insitu.log 2>&1
python3 tools/pmc_insitu_traffic.py summarise gpurun_out/r3prof/insitu > gpurun_out/r3prof/r3_insitu_hbm_traffic_per_kernel.txt 2>&1
for d in c2 c4; do f=$(find gpurun_out/r3prof/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3prof/${d}_kernel_stats.csv; done
ls -la gpurun_out/r3prof | head -30
