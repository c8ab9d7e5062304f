#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r4q_blaslt -o blaslt -- python3 tools/hipblaslt_ref.py 65536 > gpurun_out/r4q_blaslt.log 2>&1
tail -5 gpurun_out/r4q_blaslt.log
find gpurun_out/r4q_blaslt -name "*kernel_stats*" | head
f=$(find gpurun_out/r4q_blaslt -name "*kernel_stats.csv" | head -1)
cut -c1-400 "$f" | head -12
find gpurun_out/r4q_blaslt -name "*kernel_trace.csv" -delete
