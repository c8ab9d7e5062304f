#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 tools/bin/gemm_w4_probe 2>&1 | tee gpurun_out/r4r_w4_probe.log
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r4r_pmc -- tools/bin/gemm_w4_probe > gpurun_out/r4r_pmc.log 2>&1
f=$(find gpurun_out/r4r_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k]+=1
for k,v in agg.items(): print(k, dict(v))
PY
find gpurun_out/r4r_pmc -name "*.csv" -size +2M -delete
